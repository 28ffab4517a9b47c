"""Pin the CPU oracle (oracle/ref_cpu.py) to vectors recorded from the reference's
own TransformerDecoder (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import helpers
from helpers import DEFAULT_B8, GOLDEN_DIR, TINY
from oracle import ref_cpu


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _leaf_params(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def test_concat_known_answer():
    # img_text_concat.py:115-123: rows are patches, then sos, then caption; seq-first.
    B, P, L, d = 2, 3, 2, 4
    img = torch.arange(B * P * d, dtype=torch.float32).reshape(B, P, d)
    sos = 1000 + torch.arange(B * d, dtype=torch.float32).reshape(B, 1, d)
    cap = 2000 + torch.arange(B * L * d, dtype=torch.float32).reshape(B, L, d)
    x = ref_cpu.concat_prefix(img, sos, cap)
    assert x.shape == (P + 1 + L, B, d)
    for b in range(B):
        assert torch.equal(x[:P, b], img[b])
        assert torch.equal(x[P, b], sos[b, 0])
        assert torch.equal(x[P + 1:, b], cap[b])


def test_mask_reference_mode():
    g = _load("mask.npz")
    m = ref_cpu.caption_mask(229, "reference")
    assert np.array_equal(torch.unique(m).numpy(), g["m229_unique"])
    big = int(g["big_len"])
    mb = ref_cpu.caption_mask(big, "reference")
    assert np.array_equal(mb[-12:, -12:].numpy(), g["big_corner"])
    assert int(torch.isinf(mb).sum()) == int(g["big_neginf_count"])
    assert int((mb[:, :12288] != 0).sum()) == int(g["big_left_nonzero"])


def test_tiny_forward_backward_adam():
    g = _load("tiny.npz")
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=101)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=202)
    x = ref_cpu.concat_prefix(img, sos, cap)
    p = _leaf_params(sd)
    blocks, head = ref_cpu.split_params(p)
    logits = ref_cpu.decoder_forward(x, blocks, head, sh.nhead)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=2e-5, rtol=1e-5)
    loss, correct, pred = ref_cpu.caption_loss(logits, labels, sh.L)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert correct == int(g["correct"])
    assert np.array_equal(pred.numpy(), g["pred"])
    loss.backward()
    for k in ref_cpu.PARAM_KEYS:
        ref = g["grad." + k]
        np.testing.assert_allclose(p[k].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=1e-4)

    # two Adam steps (decoder.py:135-139,188-190) land on the same parameters
    w = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    for step in (1, 2):
        pp = _leaf_params(w)
        blocks, head = ref_cpu.split_params(pp)
        lg = ref_cpu.decoder_forward(x, blocks, head, sh.nhead)
        ls, _, _ = ref_cpu.caption_loss(lg, labels, sh.L)
        if step == 2:
            assert abs(ls.item() - float(g["loss_step2"])) < 1e-5
        ls.backward()
        for k in w:
            w[k], m[k], v2[k] = ref_cpu.adam_step(w[k], pp[k].grad, m[k], v2[k], step)
    for k in ref_cpu.PARAM_KEYS:
        np.testing.assert_allclose(w[k].numpy(), g["after2." + k], atol=3e-6, rtol=1e-6)


def test_default_b8():
    g = _load("default_b8.npz")
    sh = DEFAULT_B8
    sd = helpers.make_state_dict(sh, seed=303)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=404)
    x = ref_cpu.concat_prefix(img, sos, cap)
    p = _leaf_params(sd)
    blocks, head = ref_cpu.split_params(p)
    logits = ref_cpu.decoder_forward(x, blocks, head, sh.nhead)
    np.testing.assert_allclose(logits.detach()[-sh.L:, :, ::997].numpy(), g["logits_tail_strided"], atol=3e-5)
    np.testing.assert_allclose(logits.detach()[0, :, ::997].numpy(), g["logits_row0_strided"], atol=3e-5)
    loss, correct, _ = ref_cpu.caption_loss(logits, labels, sh.L)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert correct == int(g["correct"])
    loss.backward()
    for k in ref_cpu.PARAM_KEYS:
        assert abs(p[k].grad.norm().item() - float(g["gnorm." + k])) <= 2e-4 * float(g["gnorm." + k]) + 1e-9
    np.testing.assert_allclose(p["norm2.weight"].grad.numpy(), g["grad.norm2.weight"], atol=1e-7, rtol=2e-3)
    np.testing.assert_allclose(p["output_projection.weight"].grad[::1009].numpy(),
                               g["grad.output_projection.weight_rows"], atol=1e-8, rtol=2e-3)


def test_composed_modes():
    g = _load("composed.npz")
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=505)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=606)
    x = ref_cpu.concat_prefix(img, sos, cap)
    blocks, head = ref_cpu.split_params(sd)
    lg = ref_cpu.decoder_forward(x, blocks, head, sh.nhead, mask_mode="prefix_causal", prefix_len=sh.P)
    np.testing.assert_allclose(lg.numpy(), g["prefix_causal_logits"], atol=2e-5)

    pa = _leaf_params(helpers.make_state_dict(sh, seed=707))
    pb = _leaf_params(helpers.make_state_dict(sh, seed=808))
    ba, _ = ref_cpu.split_params(pa)
    bb, hb = ref_cpu.split_params(pb)
    with torch.no_grad():
        lg2 = ref_cpu.decoder_forward(x, ba + bb, hb, sh.nhead)
    np.testing.assert_allclose(lg2.numpy(), g["two_block_logits"], atol=3e-5)
    lg3 = ref_cpu.decoder_forward(x, ba + bb, hb, sh.nhead, mask_mode="prefix_causal", prefix_len=sh.P)
    loss, _, _ = ref_cpu.caption_loss(lg3, labels, sh.L)
    assert abs(loss.item() - float(g["two_block_pc_loss"])) < 1e-5
    loss.backward()
    for key, gk in (("linear1.weight", "two_block_pc_grad_a_linear1_weight"),):
        np.testing.assert_allclose(pa[key].grad.numpy(), g[gk], atol=1e-7, rtol=2e-3)
    np.testing.assert_allclose(pb["self_attn.in_proj_weight"].grad.numpy(),
                               g["two_block_pc_grad_b_in_proj_weight"], atol=1e-7, rtol=2e-3)


@pytest.mark.parametrize("name,sh,seeds,gain", [("tiny", TINY, (909, 1010, 1111), 4.0),
                                                ("default", DEFAULT_B8, (1212, 1313, 1414), 6.0)])
def test_greedy(name, sh, seeds, gain):
    g = _load("composed.npz")
    sd = helpers.make_state_dict(sh, seed=seeds[0], head_gain=gain)
    img, sos, _, _ = helpers.make_embeddings(sh, seed=seeds[1])
    emb = helpers.make_token_table(sh, seed=seeds[2])
    blocks, head = ref_cpu.split_params(sd)
    V = sh.vocab
    with torch.no_grad():
        ids, margins = ref_cpu.greedy_decode(img, sos, emb, blocks, head, sh.nhead, sh.L, V - 3, V - 2, V - 1,
                                             return_margins=True)
    assert np.array_equal(ids.numpy(), g[f"greedy_{name}_ids"])
    np.testing.assert_allclose(margins.numpy(), g[f"greedy_{name}_margins"], atol=5e-4)


def test_oracle_gelu_is_torch_gelu():
    """The one arithmetic of the oracle that no reference vector pins (the reference hard-codes ReLU, decoder.py:60):
    its erf-form GELU is torch.nn.functional.gelu's."""
    import torch
    from oracle import ref_cpu
    x = torch.linspace(-8.0, 8.0, 4001)
    assert torch.allclose(ref_cpu._activation(x, "gelu"), torch.nn.functional.gelu(x), rtol=0, atol=2e-6)      # a few f32 ulp at |x| = 8 (measured 9.5e-7)
