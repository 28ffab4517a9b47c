"""Parity of the HIP path (through the C ABI, via the drop-in module) with
  (a) golden vectors recorded from the reference's own TransformerDecoder, and
  (b) the CPU oracle on the same seeded inputs.

Stated tolerances:
  f32 validation mode : logits <= 1e-4 abs, loss <= 1e-5 rel, grads <= 1e-4 of the
                        tensor's max, greedy ids identical.
  bf16 MFMA mode      : logits <= 5e-2 abs (logit scale ~3), loss <= 1e-2 rel,
                        gradient cosine >= 0.999 per tensor.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from helpers import DEFAULT_B8, GOLDEN_DIR, TINY  # noqa: E402
from image_caption_amd import ops  # noqa: E402
from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions, block_key  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV = "cuda"
F32 = torch.float32
BF16 = torch.bfloat16


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def make_model(sh, sd, dtype, dropout=0.0, n_blocks=1, **opt):
    m = TransformerDecoder(sh.d, sh.nhead, 6, sh.ff, dropout, sh.vocab, n_blocks=n_blocks,
                           options=EngineOptions(compute_dtype=dtype, **opt))
    m.load_state_dict(sd)
    return m.to(DEV)


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


# Tensors whose gradient passes through the ReLU's on/off pattern before anything averages it.  A bf16 forward puts
# the pre-activations closest to zero on the other side of it (0.075 % of them at the default shape), and each flipped
# gate moves a whole term of these sums: 4 % relative error, cosine 0.9989-0.9993, in any bf16 formulation, while every
# other tensor stays above 0.9999 (measured: tools/grad_cosines.py; handing the oracle the same gates removes it:
# test_bf16_gradients_with_the_same_relu_gates).
RELU_GATED = ("linear1.weight", "linear1.bias", "norm2.weight", "norm2.bias")


def grad_bar(key, activation="relu", gated_bar=0.9985):
    return gated_bar if activation == "relu" and key.endswith(RELU_GATED) else 0.999


def check_grads(model, ref_grads, dtype, activation="relu", gated_bar=0.9985):
    for k, ref in ref_grads.items():
        g = dict(model.named_parameters())[k].grad.cpu()
        ref = torch.as_tensor(ref)
        if dtype == F32:
            assert (g - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-9, k
        else:
            assert cos(g, ref) >= grad_bar(k, activation, gated_bar), (k, cos(g, ref))


# ---------------------------------------------------------------------------- golden: tiny, full tensors
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("slab,keep", [(0, True), (0, False), (128, True)])   # kept numerators / recomputed / 4 ragged slabs
def test_tiny_golden_logits_loss_grads(dtype, slab, keep):
    g = golden("tiny.npz")
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=101)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=202)
    model = make_model(sh, sd, dtype, vocab_slab=slab, keep_softmax_numerators=keep)
    model.train()
    x = ref_cpu.concat_prefix(img, sos, cap).to(DEV)
    # (1) the reference's forward: logits for all rows
    logits = model(x, x)
    tol = 1e-4 if dtype == F32 else 5e-2
    assert (logits.detach().cpu() - torch.from_numpy(g["logits"])).abs().max().item() <= tol
    # (2) the reference's step body on those logits (decoder.py:172-190)
    capo = logits[-sh.L:].permute(1, 0, 2)
    loss = torch.nn.CrossEntropyLoss()(capo.reshape(-1, sh.vocab), labels.to(DEV).reshape(-1))
    assert abs(loss.item() - float(g["loss"])) <= (1e-5 if dtype == F32 else 1e-2) * float(g["loss"])
    loss.backward()
    ref_grads = {k: g["grad." + k] for k in ref_cpu.PARAM_KEYS}
    check_grads(model, ref_grads, dtype)
    # (3) the fused path: join + scored rows only, logits never materialised
    model.zero_grad()
    loss2, correct = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    assert abs(loss2.item() - float(g["loss"])) <= (1e-5 if dtype == F32 else 1e-2) * float(g["loss"])
    if dtype == F32:
        assert int(correct.item()) == int(g["correct"])
    loss2.backward()
    check_grads(model, ref_grads, dtype)


@pytest.mark.parametrize("dtype,slab,opt", [
    (F32, 0, {}), (BF16, 0, {}),
    (BF16, 8192, {}),                                   # the recomputing backward
    (BF16, 0, {"ce_row_reference": False}),             # numerators per 64-column slot, d(logits) made by a pass of its own
    (BF16, 0, {"ce_bias_in_wgrad": False}),             # vocabulary bias gradient by weighted column sums
    (BF16, 0, {"dgrad_transposed_weights": True}),      # data gradients on transposed weight copies
])
def test_default_b8_golden(dtype, slab, opt):
    g = golden("default_b8.npz")
    sh = DEFAULT_B8
    sd = helpers.make_state_dict(sh, seed=303)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=404)
    model = make_model(sh, sd, dtype, vocab_slab=slab, **opt)
    x = ref_cpu.concat_prefix(img, sos, cap).to(DEV)
    with torch.no_grad():
        logits = model(x, x)
    tol = 1e-4 if dtype == F32 else 5e-2
    assert (logits[-sh.L:, :, ::997].cpu() - torch.from_numpy(g["logits_tail_strided"])).abs().max().item() <= tol
    assert (logits[0, :, ::997].cpu() - torch.from_numpy(g["logits_row0_strided"])).abs().max().item() <= tol
    del logits
    loss, correct = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    assert abs(loss.item() - float(g["loss"])) <= (1e-5 if dtype == F32 else 1e-2) * float(g["loss"])
    if dtype == F32:
        assert int(correct.item()) == int(g["correct"])
    loss.backward()
    named = dict(model.named_parameters())
    for k in ref_cpu.PARAM_KEYS:
        gn = named[k].grad.norm().item()
        ref = float(g["gnorm." + k])
        assert abs(gn - ref) <= (2e-4 if dtype == F32 else 2e-2) * ref, (k, gn, ref)
    rows = named["output_projection.weight"].grad[::1009].cpu()
    ref_rows = torch.from_numpy(g["grad.output_projection.weight_rows"])
    if dtype == F32:
        assert (rows - ref_rows).abs().max().item() <= 1e-4 * ref_rows.abs().max().item()
        assert (named["norm2.weight"].grad.cpu() - torch.from_numpy(g["grad.norm2.weight"])).abs().max().item() <= \
            1e-4 * np.abs(g["grad.norm2.weight"]).max()
    else:
        assert cos(rows, ref_rows) >= 0.999
        assert cos(named["self_attn.in_proj_bias"].grad.cpu(), torch.from_numpy(g["grad.self_attn.in_proj_bias"])) >= 0.999
    # every gradient element-wise (vectors in full, matrices on a strided grid): a permutation or a sign flip inside a
    # tensor keeps its norm but not these
    gg = golden("default_b8_grads.npz")
    for k in ref_cpu.PARAM_KEYS:
        got = helpers.grad_slice(named[k].grad).cpu()
        ref = torch.from_numpy(gg["grad." + k])
        assert got.shape == ref.shape, k
        if dtype == F32:
            assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-12, k
        else:
            assert cos(got, ref) >= grad_bar(k), (k, cos(got, ref))


def test_bf16_gradients_with_the_same_relu_gates():
    """Default shape, batch 8, bf16: once the oracle is handed the on/off pattern of the ReLU that the bf16 forward
    produced, EVERY gradient reaches cosine 0.999 -- what separates the gate-downstream tensors from the f32 result is
    the few pre-activations whose sign bf16 changes, nothing else."""
    sh = DEFAULT_B8
    sd = helpers.make_state_dict(sh, seed=303)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=404)
    model = make_model(sh, sd, BF16, scored_rows_only=False)
    model.train()
    loss, _ = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    loss.backward()
    gate = (model.engine.saved[0]["a1"] != 0).reshape(sh.S, sh.B, sh.ff).cpu()
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    blocks, head = ref_cpu.split_params(leaf)
    x = ref_cpu.concat_prefix(img, sos, cap)
    h = ref_cpu.decoder_hidden(x, blocks, sh.nhead, keep=[{"relu_gate": gate}])
    tail = h[-sh.L:] @ head["output_projection.weight"].t() + head["output_projection.bias"]       # the scored rows
    lo, _, _ = ref_cpu.caption_loss(tail, labels, sh.L)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-2 * lo.item()
    named = dict(model.named_parameters())
    for k in leaf:
        assert cos(named[k].grad.cpu(), leaf[k].grad) >= 0.999, (k, cos(named[k].grad.cpu(), leaf[k].grad))


# ---------------------------------------------------------------------------- oracle: build-defined modes
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mask_mode,activation,n_blocks,use_pos,scored", [
    ("reference", "relu", 1, False, False),
    ("prefix_causal", "relu", 1, False, True),
    ("prefix_causal", "gelu", 2, True, True),
    ("none", "gelu", 3, False, True),
])
def test_modes_vs_oracle(dtype, mask_mode, activation, n_blocks, use_pos, scored):
    sh = TINY
    sds = [helpers.make_state_dict(sh, seed=900 + i) for i in range(n_blocks)]
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=606)
    pos = 0.3 * torch.randn(sh.S, sh.d, generator=torch.Generator().manual_seed(5)) if use_pos else None
    # product model: block 0 = reference keys, further blocks under blocks.{i}.
    full = dict(sds[0])
    for i in range(1, n_blocks):
        for k in ref_cpu.BLOCK_KEYS:
            full[block_key(i, k, n_blocks)] = sds[i][k]
    model = make_model(sh, full, dtype, n_blocks=n_blocks, mask_mode=mask_mode, activation=activation, scored_rows_only=scored)
    model.train()
    loss, correct = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV),
                                       pos=pos.to(DEV) if use_pos else None)
    loss.backward()
    # oracle
    leaf = [{k: sds[i][k].clone().requires_grad_(True) for k in ref_cpu.BLOCK_KEYS} for i in range(n_blocks)]
    head = {k: sds[0][k].clone().requires_grad_(True) for k in ref_cpu.PARAM_KEYS[12:]}
    x = ref_cpu.concat_prefix(img, sos, cap, pos)
    lg = ref_cpu.decoder_forward(x, leaf, head, sh.nhead, mask_mode=mask_mode, prefix_len=sh.P, activation=activation)
    lo, co, _ = ref_cpu.caption_loss(lg, labels, sh.L)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= (1e-5 if dtype == F32 else 1e-2) * lo.item()
    if dtype == F32:
        assert int(correct.item()) == co
    ref_grads = {k: head[k].grad for k in head}
    for i in range(n_blocks):
        for k in ref_cpu.BLOCK_KEYS:
            ref_grads[block_key(i, k, n_blocks)] = leaf[i][k].grad
    # bf16: 0.999 per tensor, except those read straight through the ReLU's gates (0.998 at this 64-wide shape)
    check_grads(model, ref_grads, dtype, activation=activation, gated_bar=0.998)


def test_composed_golden():
    """prefix-causal and stacked blocks against vectors produced by composing the reference class."""
    g = golden("composed.npz")
    sh = TINY
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=606)
    x = ref_cpu.concat_prefix(img, sos, cap).to(DEV)
    m = make_model(sh, helpers.make_state_dict(sh, seed=505), F32, mask_mode="prefix_causal")
    m.prefix_len = sh.P
    with torch.no_grad():
        lg = m(x, x)
    assert (lg.cpu() - torch.from_numpy(g["prefix_causal_logits"])).abs().max().item() <= 1e-4
    sa, sb = helpers.make_state_dict(sh, seed=707), helpers.make_state_dict(sh, seed=808)
    full = dict(sa)
    full.update({k: sb[k] for k in ref_cpu.PARAM_KEYS[12:]})
    for k in ref_cpu.BLOCK_KEYS:
        full[block_key(1, k, 2)] = sb[k]
    m2 = make_model(sh, full, F32, n_blocks=2)
    with torch.no_grad():
        lg2 = m2(x, x)
    assert (lg2.cpu() - torch.from_numpy(g["two_block_logits"])).abs().max().item() <= 1e-4
    m3 = make_model(sh, full, F32, n_blocks=2, mask_mode="prefix_causal")
    loss, _ = m3.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    assert abs(loss.item() - float(g["two_block_pc_loss"])) <= 1e-5 * float(g["two_block_pc_loss"])
    loss.backward()
    named = dict(m3.named_parameters())
    ga = torch.from_numpy(g["two_block_pc_grad_a_linear1_weight"])
    gb = torch.from_numpy(g["two_block_pc_grad_b_in_proj_weight"])
    assert (named["linear1.weight"].grad.cpu() - ga).abs().max().item() <= 1e-4 * ga.abs().max().item()
    assert (named["blocks.1.self_attn.in_proj_weight"].grad.cpu() - gb).abs().max().item() <= 1e-4 * gb.abs().max().item()


# ---------------------------------------------------------------------------- dropout: same masks handed to the oracle
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_training_dropout_vs_oracle_with_same_masks(dtype):
    sh = TINY
    p = 0.2
    sd = helpers.make_state_dict(sh, seed=111)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=222)
    model = make_model(sh, sd, dtype, dropout=p, scored_rows_only=False)
    model.train()
    loss, _ = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    loss.backward()
    step = model._step
    seed = model.options.dropout_seed
    S, B, d, ff, H = sh.S, sh.B, sh.d, sh.ff, sh.nhead
    keep = {
        "attn": ops.attn_dropout_mask(ops.dropout_keys(seed, 0, step, p), B, H, S, DEV).cpu(),
        "resid1": ops.dropout_mask(ops.dropout_keys(seed, 1, step, p), S * B * d, DEV).reshape(S, B, d).cpu(),
        "hidden": ops.dropout_mask(ops.dropout_keys(seed, 2, step, p), S * B * ff, DEV).reshape(S, B, ff).cpu(),
        "resid2": ops.dropout_mask(ops.dropout_keys(seed, 3, step, p), S * B * d, DEV).reshape(S, B, d).cpu(),
    }
    for v in keep.values():
        assert 0.7 < v.mean().item() < 0.9
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    blocks, head = ref_cpu.split_params(leaf)
    x = ref_cpu.concat_prefix(img, sos, cap)
    lg = ref_cpu.decoder_forward(x, blocks, head, sh.nhead, drop_p=p, keep=[keep])
    lo, _, _ = ref_cpu.caption_loss(lg, labels, sh.L)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= (2e-5 if dtype == F32 else 1e-2) * lo.item()
    check_grads(model, {k: leaf[k].grad for k in leaf}, dtype)
    # evaluation mode ignores dropout
    model.eval()
    with torch.no_grad():
        le, _ = model.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV))
    lg0 = ref_cpu.decoder_forward(x, *ref_cpu.split_params(sd), sh.nhead)
    l0, _, _ = ref_cpu.caption_loss(lg0, labels, sh.L)
    assert abs(le.item() - l0.item()) <= (2e-5 if dtype == F32 else 1e-2) * l0.item()


def test_scored_rows_only_is_exact():
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=121)
    img, sos, cap, labels = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=232)]
    outs = []
    for scored in (False, True):
        m = make_model(sh, sd, F32, scored_rows_only=scored)
        loss, correct = m.caption_loss((img, sos, cap), labels)
        loss.backward()
        outs.append((loss.item(), correct.item(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * abs(outs[0][0]) and outs[0][1] == outs[1][1]
    for k in outs[0][2]:
        a, b = outs[0][2][k], outs[1][2][k]
        assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item() + 1e-10, k


# ---------------------------------------------------------------------------- Adam / trainer
def test_trainer_two_adam_steps_match_reference():
    g = golden("tiny.npz")
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=101)
    img, sos, cap, labels = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=202)]
    model = make_model(sh, sd, F32)
    model.train()
    tr = CaptionTrainer(model, lr=1e-4, weight_decay=0.01)
    l1, _, rows = tr.step((img, sos, cap), labels)
    l2, _, _ = tr.step((img, sos, cap), labels)
    assert abs(l1.item() / rows - float(g["loss"])) <= 1e-5 * float(g["loss"])
    assert abs(l2.item() / rows - float(g["loss_step2"])) <= 1e-5 * float(g["loss_step2"])
    state = model.state_dict()
    for k in ref_cpu.PARAM_KEYS:
        assert (state[k].cpu() - torch.from_numpy(g["after2." + k])).abs().max().item() <= 3e-6, k


def test_stack_trainer_with_block_weight_images(gemm_mode):
    """A two-block stack trained through the parameter store: the images of the blocks' matrices it keeps (refreshed by a
    batched launch after every update) are what the persistent kernel reads, and leave every result as it is without them."""
    from image_caption_amd.engine import PANELS
    geo = dict(d_model=256, nhead=2, dim_feedforward=512, vocab=1003, P=20, L=8, B=6)
    finals = []
    for use in (True, False):
        torch.manual_seed(5)
        opts = EngineOptions(compute_dtype=BF16, mask_mode="prefix_causal", head_k_panels=use)
        model = TransformerDecoder(geo["d_model"], geo["nhead"], 6, geo["dim_feedforward"], 0.1, geo["vocab"], n_blocks=2, options=opts).to(DEV)
        model.train()
        tr = CaptionTrainer(model, lr=1e-3, weight_decay=0.01)
        assert all(("linear1.weight" + PANELS) in b for b in tr.W.blocks)
        g = torch.Generator().manual_seed(6)
        img = torch.randn(geo["B"], geo["P"], geo["d_model"], generator=g).to(DEV)
        sos = torch.randn(geo["B"], 1, geo["d_model"], generator=g).to(DEV)
        cap = torch.randn(geo["B"], geo["L"], geo["d_model"], generator=g).to(DEV)
        labels = torch.randint(0, geo["vocab"], (geo["B"], geo["L"]), generator=g).to(DEV)
        gemm_mode(10)                                     # every eligible product on the persistent kernel, whatever its size
        losses = [tr.step((img, sos, cap), labels)[0].item() for _ in range(3)]
        gemm_mode(0)
        # the images follow the updates: block z of the stored order is block n - 1 - z
        for name, panels in tr.store.block_panels.items():
            for i in range(2):
                wmat = tr.W.blocks[i][name]
                assert torch.equal(panels[1 - i], wmat.view(wmat.shape[0], -1, 32).permute(1, 0, 2)), (name, i)
        finals.append((losses, {k: v.clone() for k, v in model.state_dict().items()}))
    assert finals[0][0] == finals[1][0]
    for k in finals[0][1]:
        assert torch.equal(finals[0][1][k], finals[1][1][k]), k


def test_head_backward_past_8192_scored_rows():
    """More than 8192 scored rows (a per-GPU batch above 256): the weight gradient is cut along the rows so that the bias
    column still rides on it, and d(logits) stays a column-panel image.  Against the same step with a separate bias pass over
    a row-major d(logits), and against the f32 mode."""
    d, H, ff, V, P, L, B = 256, 2, 512, 1003, 3, 32, 288            # 9216 scored rows
    g = torch.Generator().manual_seed(11)
    img, sos, cap = (torch.randn(B, n, d, generator=g).to(DEV) for n in (P, 1, L))
    labels = torch.randint(0, V, (B, L), generator=g).to(DEV)
    grads = {}
    for name, opts in (("image", EngineOptions(compute_dtype=BF16)),
                       ("matrix", EngineOptions(compute_dtype=BF16, dl_panels=False, ce_bias_in_wgrad=False)),
                       ("f32", EngineOptions(compute_dtype=F32))):
        torch.manual_seed(12)
        model = TransformerDecoder(d, H, 6, ff, 0.0, V, options=opts).to(DEV)
        model.train()
        tr = CaptionTrainer(model, lr=0.0, weight_decay=0.0)            # the flat parameter store (zero rows behind the vocabulary matrix)
        loss_sum, _, rows = tr.step((img, sos, cap), labels)
        if name == "image":
            assert model.engine.last_dl_panels and model.engine.last_by_row
        grads[name] = (loss_sum.item() / rows, {k: v.clone() for k, v in tr.grads.items()})
    assert abs(grads["image"][0] - grads["matrix"][0]) <= 1e-6 * abs(grads["matrix"][0])
    assert abs(grads["image"][0] - grads["f32"][0]) <= 1e-2 * abs(grads["f32"][0])
    for k, gi in grads["image"][1].items():
        gm, gf = grads["matrix"][1][k], grads["f32"][1][k]
        assert (gi - gm).abs().max().item() <= 2e-3 * gm.abs().max().item() + 1e-12, k
        cos = torch.nn.functional.cosine_similarity(gi.flatten().double(), gf.flatten().double(), dim=0).item()
        assert cos >= 0.9985, (k, cos)


def test_ragged_last_batch_reuses_the_scratch_buffers():
    """A smaller batch after a larger one (the last batch of an epoch) allocates nothing new -- its scratch tensors are
    views of the existing ones -- and gives what a fresh model gives on that batch."""
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=101)
    img, sos, cap, labels = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=202)]
    nsmall = max(1, img.shape[0] // 2)
    small = tuple(t[:nsmall].contiguous() for t in (img, sos, cap))
    model = make_model(sh, sd, F32)
    model.train()
    model.caption_loss((img, sos, cap), labels)[0].backward()
    eng = model.engine
    held = {k: (v.data_ptr(), v.numel()) for k, v in eng._bufs.items() if torch.is_tensor(v)}
    model.zero_grad()
    loss = model.caption_loss(small, labels[:nsmall].contiguous())[0]
    loss.backward()
    after = {k: (v.data_ptr(), v.numel()) for k, v in eng._bufs.items() if torch.is_tensor(v)}
    assert after == held, "the smaller batch changed the engine's allocations"
    fresh = make_model(sh, sd, F32)
    fresh.train()
    want = fresh.caption_loss(small, labels[:nsmall].contiguous())[0]
    want.backward()
    assert abs(loss.item() - want.item()) <= 1e-6 * abs(want.item())
    for (k, p), (_, q) in zip(model.named_parameters(), fresh.named_parameters()):
        assert torch.equal(p.grad, q.grad), k


def test_trainer_bf16_loss_decreases_and_tracks_f32():
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=131)
    img, sos, cap, labels = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=242)]
    losses = {}
    for dtype in (F32, BF16):
        model = make_model(sh, sd, dtype)
        tr = CaptionTrainer(model, lr=1e-3, weight_decay=0.01)
        losses[dtype] = [tr.step((img, sos, cap), labels)[0].item() / (sh.B * sh.L) for _ in range(8)]
    assert losses[BF16][-1] < losses[BF16][0]
    for a, b in zip(losses[F32], losses[BF16]):
        assert abs(a - b) <= 1e-2 * a


# ---------------------------------------------------------------------------- greedy decoding
@pytest.mark.parametrize("name,sh,seeds,gain", [("tiny", TINY, (909, 1010, 1111), 4.0),
                                                ("default", DEFAULT_B8, (1212, 1313, 1414), 6.0)])
def test_greedy_ids_identical_in_f32_mode(name, sh, seeds, gain):
    g = golden("composed.npz")
    sd = helpers.make_state_dict(sh, seed=seeds[0], head_gain=gain)
    img, sos, _, _ = helpers.make_embeddings(sh, seed=seeds[1])
    emb = helpers.make_token_table(sh, seed=seeds[2])
    model = make_model(sh, sd, F32)
    model.eval()
    ids, margins = model.generate(img.to(DEV), sos.to(DEV), emb.to(DEV), max_len=sh.L, return_margins=True)
    ref_ids = torch.from_numpy(g[f"greedy_{name}_ids"])
    ref_margins = torch.from_numpy(g[f"greedy_{name}_margins"])
    assert float(ref_margins.min()) > 1e-3            # the fixture's decisions are far from ties
    assert torch.equal(ids.cpu(), ref_ids)
    live = ref_ids != sh.vocab - 1
    assert (margins.cpu() - ref_margins)[live].abs().max().item() <= 1e-3
    # bf16: identical wherever the reference's margin is well above bf16 noise, up to the first divergence
    mb = make_model(sh, sd, BF16)
    idb = mb.generate(img.to(DEV), sos.to(DEV), emb.to(DEV), max_len=sh.L).cpu()
    safe = (ref_margins > 0.25).cumprod(dim=1).bool()
    assert torch.equal(idb[safe], ref_ids[safe])


def test_greedy_eos_early_stop_changes_nothing():
    """Once every sequence has produced <eos> the loop may end early: the columns left hold <pad> either way."""
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=909, head_gain=4.0)
    img, sos, _, _ = helpers.make_embeddings(sh, seed=1010)
    emb = helpers.make_token_table(sh, seed=1111)
    model = make_model(sh, sd, F32)
    model.eval()
    args = (img[:1].to(DEV), sos[:1].to(DEV), emb.to(DEV))
    free = model.generate(*args, max_len=sh.L)
    eos = int(free[0, 2])                                   # the third token this sequence emits becomes <eos>
    pad = sh.vocab - 1
    full = model.generate(*args, max_len=sh.L, eos_id=eos, pad_id=pad)
    early = model.generate(*args, max_len=sh.L, eos_id=eos, pad_id=pad, stop_check_every=1)
    assert torch.equal(full, early)
    first = (full[0] == eos).nonzero()[0, 0].item()
    assert first <= 2 and bool((full[0, first + 1:] == pad).all())


def test_greedy_multi_block_matches_oracle():
    sh = TINY
    nb = 2
    sds = [helpers.make_state_dict(sh, seed=940 + i, head_gain=4.0) for i in range(nb)]
    full = dict(sds[0])
    for k in ref_cpu.BLOCK_KEYS:
        full[block_key(1, k, nb)] = sds[1][k]
    img, sos, _, _ = helpers.make_embeddings(sh, seed=1515)
    emb = helpers.make_token_table(sh, seed=1616)
    pos = 0.3 * torch.randn(sh.S, sh.d, generator=torch.Generator().manual_seed(6))
    model = make_model(sh, full, F32, n_blocks=nb, mask_mode="prefix_causal", activation="gelu")
    V = sh.vocab
    ids = model.generate(img.to(DEV), sos.to(DEV), emb.to(DEV), max_len=sh.L, pos=pos.to(DEV)).cpu()
    blocks = [{k: sds[i][k] for k in ref_cpu.BLOCK_KEYS} for i in range(nb)]
    head = {k: sds[0][k] for k in ref_cpu.PARAM_KEYS[12:]}
    with torch.no_grad():
        ref_ids, margins = ref_cpu.greedy_decode(img, sos, emb, blocks, head, sh.nhead, sh.L, V - 3, V - 2, V - 1,
                                                 mask_mode="prefix_causal", activation="gelu", pos=pos, return_margins=True)
    safe = (margins > 1e-3).cumprod(dim=1).bool()
    assert safe.float().mean().item() > 0.5
    assert torch.equal(ids[safe], ref_ids[safe])
    with pytest.raises(ValueError, match="prefix_causal"):
        make_model(sh, full, F32, n_blocks=nb).generate(img.to(DEV), sos.to(DEV), emb.to(DEV), max_len=4)


def test_decode_step_driver_equals_the_launch_by_launch_loop():
    """capdec_decode_step composes the same kernels the per-kernel loop issues: ids and margins agree bit for bit."""
    from image_caption_amd.engine import KernelTimer
    sh = TINY
    nb = 2
    sds = [helpers.make_state_dict(sh, seed=950 + i, head_gain=4.0) for i in range(nb)]
    full = dict(sds[0])
    for k in ref_cpu.BLOCK_KEYS:
        full[block_key(1, k, nb)] = sds[1][k]
    img, sos, _, _ = helpers.make_embeddings(sh, seed=1717)
    emb = helpers.make_token_table(sh, seed=1818)
    pos = 0.3 * torch.randn(sh.S, sh.d, generator=torch.Generator().manual_seed(8))
    for dtype in (F32, BF16):
        model = make_model(sh, full, dtype, n_blocks=nb, mask_mode="prefix_causal", activation="gelu")
        model.eval()
        args = (img.to(DEV), sos.to(DEV), emb.to(DEV))
        ids_a, mg_a = model.generate(*args, max_len=sh.L, pos=pos.to(DEV), return_margins=True)
        model.engine.timer = KernelTimer()                  # with a timer attached generate() launches kernel by kernel
        ids_b, mg_b = model.generate(*args, max_len=sh.L, pos=pos.to(DEV), return_margins=True)
        model.engine.timer = None
        assert torch.equal(ids_a, ids_b) and torch.equal(mg_a, mg_b)


def test_decode_step_rejects_bad_descriptors():
    import ctypes as C
    from image_caption_amd import _lib
    lib = _lib.load()
    dd = _lib.DecodeDesc()
    assert lib.capdec_decode_step(C.byref(dd), None) == -1          # CAPDEC_ERR_INVALID: empty geometry
    assert b"geometry" in lib.capdec_last_error()
    assert lib.capdec_decode_step_workspace(0, 768, 2048, 50260, _lib.BF16) == 0
    need = lib.capdec_decode_step_workspace(4, 64, 128, 100, _lib.F32)
    assert need > 0 and need % 256 == 0
    blocks = (_lib.BlockParams * 1)()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    dd.B, dd.d_model, dd.nhead, dd.dim_ff, dd.vocab, dd.n_blocks, dd.smax = 4, 64, 4, 128, 100, 1, 16
    dd.dtype, dd.blocks, dd.max_len, dd.t, dd.s_new = _lib.F32, blocks, 4, 0, 3
    for f in ("head_w", "emb", "x_last", "ids", "done", "out"):
        setattr(dd, f, buf.data_ptr())
    dd.workspace, dd.workspace_bytes = buf.data_ptr(), 16
    assert lib.capdec_decode_step(C.byref(dd), None) == -3          # CAPDEC_ERR_WORKSPACE
    dd.t = 4
    assert lib.capdec_decode_step(C.byref(dd), None) == -1 and b"token index" in lib.capdec_last_error()


# ---------------------------------------------------------------------------- drop-in surface
def test_prefix_join_and_process_batch():
    from image_caption_amd.img_text_concat import process_batch
    B, P, L, d = 3, 5, 4, 64
    img = torch.arange(B * P * d, dtype=torch.float32, device=DEV).reshape(B, P, d)
    sos = 1e4 + torch.arange(B * d, dtype=torch.float32, device=DEV).reshape(B, 1, d)
    cap = 2e4 + torch.arange(B * L * d, dtype=torch.float32, device=DEV).reshape(B, L, d)
    ids = torch.zeros(B, L, dtype=torch.int64, device=DEV)
    out = process_batch({"image_embeddings": img, "sos_embedding": sos, "caption_embeddings": cap,
                         "caption_input_ids": ids, "caption_labels": ids + 1})
    ref = ref_cpu.concat_prefix(img.cpu(), sos.cpu(), cap.cpu())
    assert out["decoder_inputs"].shape == (P + 1 + L, B, d)
    assert torch.equal(out["decoder_inputs"].cpu(), ref)
    assert out["caption_labels"].dtype == torch.int64 and out["caption_labels"].shape == (B, L)


def test_process_batch_reference_dict_form_with_injected_encoders():
    """process_batch({'image': [...], 'caption': [...]}) (img_text_concat.py:67-70): the strings each encoder callable
    receives are the reference's ("<sos> cap" to the text model and the input tokenizer, "cap <eos>" to the label
    tokenizer, img_text_concat.py:70-92), and the output is the join of what they return."""
    from image_caption_amd import img_text_concat as itc
    B, P, L, d = 3, 5, itc.max_caption_length, 64
    gen = torch.Generator().manual_seed(9)
    img = torch.randn(B, P, d, generator=gen)
    cap = torch.randn(B, L, d, generator=gen)
    sos = torch.randn(1, 1, d, generator=gen)
    seen = {}

    def patches(images):
        seen["images"] = list(images)
        return img.to(DEV)

    def tokens(strings):
        seen["text_model"] = list(strings)
        return cap.to(DEV)

    def tokenizer(strings, padding=None, max_length=None, truncation=None, return_tensors=None):
        assert (padding, max_length, truncation, return_tensors) == ("max_length", L, True, "pt")
        seen.setdefault("tokenizer", []).append(list(strings))
        ids = torch.full((len(strings), max_length), 50259, dtype=torch.int64)
        for r, s in enumerate(strings):
            words = s.split()[:max_length]
            ids[r, :len(words)] = torch.tensor([{"<sos>": 50257, "<eos>": 50258}.get(w, len(w)) for w in words])
        return {"input_ids": ids}

    enc = itc.Encoders(patches, tokens, tokenizer, lambda n: sos.expand(n, 1, d).contiguous().to(DEV))
    batch = {"image": ["im0", "im1", "im2"], "caption": ["a dog runs", "two people", "sky"]}
    with pytest.raises(ValueError, match="needs encoders"):
        itc.process_batch(batch)
    out = itc.process_batch(batch, encoders=enc)
    assert seen["images"] == batch["image"]
    assert seen["text_model"] == ["<sos> a dog runs", "<sos> two people", "<sos> sky"]
    assert seen["tokenizer"] == [seen["text_model"], ["a dog runs <eos>", "two people <eos>", "sky <eos>"]]
    ref = ref_cpu.concat_prefix(img, sos.expand(B, 1, d), cap)
    assert out["decoder_inputs"].shape == (P + 1 + L, B, d) and torch.equal(out["decoder_inputs"].cpu(), ref)
    assert out["caption_input_ids"][0, :4].tolist() == [50257, 1, 3, 4] and out["caption_labels"][2, :3].tolist() == [3, 50258, 50259]
    itc.set_encoders(enc)                      # the reference's one-argument call form
    try:
        again = itc.process_batch(batch)
    finally:
        itc.set_encoders(None)
    assert torch.equal(again["decoder_inputs"], out["decoder_inputs"])


def test_state_dict_and_init_match_reference_layout():
    torch.manual_seed(42)
    m = TransformerDecoder(64, 2, 6, 128, 0.1, 503)
    assert list(m.state_dict().keys()) == list(ref_cpu.PARAM_KEYS)
    # same constructors, same order => same initial weights as stock modules under the seed
    torch.manual_seed(42)
    attn = torch.nn.MultiheadAttention(64, 2, dropout=0.1)
    lin1 = torch.nn.Linear(64, 128)
    assert torch.equal(m.self_attn.in_proj_weight, attn.in_proj_weight) and torch.equal(m.linear1.weight, lin1.weight)
    mask = m.create_caption_mask(21)
    assert mask.shape == (21, 21) and float(mask.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="CPU tensor"):
        m(torch.zeros(21, 2, 64), None)


def test_create_caption_mask_past_12288_columns():
    """decoder.py:33-37 at a length where the mask is not all zero: columns from 16*768 on are causal."""
    g = golden("mask.npz")
    m = TransformerDecoder(16, 2, 6, 16, 0.0, 11).to(DEV)
    n = int(g["big_len"])
    mask = m.create_caption_mask(n)
    assert mask.shape == (n, n) and mask.dtype == torch.float32
    assert torch.equal(mask[-12:, -12:].cpu(), torch.from_numpy(g["big_corner"]))
    assert int(torch.isinf(mask).sum().item()) == int(g["big_neginf_count"])
    assert int((mask[:, :12288] != 0).sum().item()) == int(g["big_left_nonzero"])
    assert bool((mask[torch.isinf(mask)] < 0).all())
    assert torch.equal(torch.unique(m.create_caption_mask(229)).cpu(), torch.from_numpy(g["m229_unique"]))


def test_train_decoder_reproduces_the_reference_history():
    """train_decoder against what the reference's own loop logged on the same seeded data (decoder.py:96-216):
    90/10 split by position, both splits trained, ragged last batch, mean of per-batch mean losses, token accuracy."""
    from image_caption_amd.config import DecoderConfig
    from image_caption_amd.decoder import train_decoder
    g = golden("train_loop.npz")
    cfg_kw, data, seed = helpers.train_loop_case()
    cfg = DecoderConfig(**cfg_kw, device="cuda")
    torch.manual_seed(seed)                     # after the config: constructing it seeds torch (config.py:41)
    logs = []
    model = train_decoder(data, cfg, options=EngineOptions(compute_dtype=F32), log=lambda *a: None, logger=logs.append)
    hist = model.training_history
    assert len(hist) == len(g["loss"]) == 2 * cfg.num_epochs
    for rec, split, epoch, loss, acc in zip(hist, g["split"], g["epoch"], g["loss"], g["accuracy"]):
        assert rec["epoch"] == int(epoch)
        assert abs(rec[f"{split}_loss"] - float(loss)) <= 1e-4 * float(loss), (split, epoch, rec, float(loss))
        assert abs(rec[f"{split}_accuracy"] - float(acc)) < 1e-9, (split, epoch, rec, float(acc))
    assert float(g["accuracy"].max()) > 10.0    # the fixture's accuracy is not the trivial zero


def test_train_decoder_loop():
    from image_caption_amd.config import DecoderConfig
    from image_caption_amd.decoder import train_decoder
    from image_caption_amd.synthetic import make_captions
    cfg = DecoderConfig(d_model=768, nhead=8, dim_feedforward=2048, dropout=0.1, batch_size=8, num_epochs=2, device="cuda")
    gen = torch.Generator().manual_seed(0)
    N, S = 40, 229
    ids, labels = make_captions(N, 32, 50260, gen)
    data = {"decoder_inputs": torch.randn(S, N, 768, generator=gen), "caption_input_ids": ids, "caption_labels": labels}
    logs = []
    model = train_decoder(data, cfg, log=lambda *a: None, logger=logs.append)
    hist = model.training_history
    assert [list(h)[0] for h in hist] == ["train_loss", "test_loss", "train_loss", "test_loss"]
    assert hist[2]["train_loss"] < hist[0]["train_loss"]
    assert "total_runtime" in logs[-1]


def test_batch_source_resident_and_streamed_feed_identical_training():
    """Batches cut from the processed dataset (decoder.py:157-166): device-resident strided views and
    pinned-host streaming deliver the same samples, and training on either gives the same parameters
    as training on contiguous copies."""
    from image_caption_amd.dataset import BatchSource
    sh = TINY
    g = torch.Generator().manual_seed(77)
    N = 11
    data = {"decoder_inputs": torch.randn(sh.S, N, sh.d, generator=g),
            "caption_input_ids": torch.randint(0, sh.vocab, (N, sh.L), generator=g),
            "caption_labels": torch.randint(0, sh.vocab, (N, sh.L), generator=g)}
    ref = [(data["decoder_inputs"][:, i:i + 4, :].contiguous(), data["caption_labels"][i:i + 4]) for i in range(2, 10, 4)]
    results = []
    for mode in ("contiguous", True, False):
        sd = helpers.make_state_dict(sh, seed=606)
        model = make_model(sh, sd, F32)
        model.train()
        tr = CaptionTrainer(model, lr=1e-3, weight_decay=0.01)
        if mode == "contiguous":
            batches = [(x.to(DEV), y.to(DEV)) for x, y in ref]
        else:
            src = BatchSource(data, 4, DEV, lo=2, hi=10, resident=mode)
            assert src.resident == mode and len(src) == 2
            batches = src
        k = 0
        for x, y in batches:
            assert torch.equal(x.cpu(), ref[k][0]) and torch.equal(y.cpu(), ref[k][1])
            tr.step(x, y)
            k += 1
        assert k == 2
        results.append({n: p.detach().cpu().clone() for n, p in model.state_dict().items()})
    for other in results[1:]:
        for n in results[0]:
            assert torch.equal(results[0][n], other[n]), n


@pytest.mark.parametrize("scored,use_pos", [(True, False), (False, True)])
def test_norm1_fold_is_exact_in_f32(scored, use_pos):
    """Folding norm1's affine into in_proj (the default when no input gradient is asked for) is algebra, not an
    approximation: in the f32 mode loss and every gradient agree with the as-written path to rounding."""
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=321)
    g = torch.Generator().manual_seed(322)
    sd["norm1.weight"] = 1.0 + 0.3 * torch.randn(sh.d, generator=g)      # away from the (1, 0) a fresh LayerNorm has
    sd["norm1.bias"] = 0.3 * torch.randn(sh.d, generator=g)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=323)
    pos = 0.3 * torch.randn(sh.S, sh.d, generator=g).to(DEV) if use_pos else None
    out = {}
    for fold in (False, True):
        m = make_model(sh, sd, F32, scored_rows_only=scored, fold_norm1=fold)
        m.train()
        loss, correct = m.caption_loss((img.to(DEV), sos.to(DEV), cap.to(DEV)), labels.to(DEV), pos=pos)
        loss.backward()
        out[fold] = (loss.item(), int(correct.item()), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])
    assert out[True][1] == out[False][1]
    for k, ref in out[False][2].items():
        err = (out[True][2][k] - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-9, (k, err)


# ---------------------------------------------------------------------------- autograd state is per call
def test_two_forwards_before_one_backward():
    """The reference module may be called twice before backward (two losses summed, an evaluation in between).
    caption_loss owns its gradients per call; forward() detects that its saved activations were overwritten."""
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=141)
    a = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=252)]
    b = [t.to(DEV) for t in helpers.make_embeddings(sh, seed=253)]
    singles = []
    for img, sos, cap, labels in (a, b):
        m = make_model(sh, sd, F32)
        m.train()
        loss, _ = m.caption_loss((img, sos, cap), labels)
        loss.backward()
        singles.append({k: p.grad.clone() for k, p in m.named_parameters()})
    m = make_model(sh, sd, F32)
    m.train()
    la, _ = m.caption_loss((a[0], a[1], a[2]), a[3])
    lb, _ = m.caption_loss((b[0], b[1], b[2]), b[3])
    with torch.no_grad():
        m.caption_loss((a[0], a[1], a[2]), a[3])            # an evaluation between forward and backward
    (la + 2.0 * lb).backward()
    for k, p in m.named_parameters():
        want = singles[0][k] + 2.0 * singles[1][k]
        assert (p.grad - want).abs().max().item() <= 1e-6 * want.abs().max().item() + 1e-10, k
    # forward(): the second call reuses the engine's activation buffers, so the first graph must refuse
    x1 = ref_cpu.concat_prefix(*[t.cpu() for t in a[:3]]).to(DEV)
    x2 = ref_cpu.concat_prefix(*[t.cpu() for t in b[:3]]).to(DEV)
    y1 = m(x1, x1)
    y2 = m(x2, x2)
    with pytest.raises(RuntimeError, match="before the backward"):
        y1.sum().backward()
    m.zero_grad()
    y2.sum().backward()                                    # the latest forward's graph is intact
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


@pytest.mark.parametrize("layout", ["fp16", "inner_stride", "odd_offset"])
def test_positional_rows_survive_every_input_layout(layout):
    """A joined input the strided LayerNorm kernel cannot address as it lies (fp16, a non-unit inner stride, rows
    that are not 16-byte aligned) takes a conversion first -- and must still get its positional rows."""
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=151)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=262)
    pos = (0.3 * torch.randn(sh.S, sh.d, generator=torch.Generator().manual_seed(9))).to(DEV)
    x = ref_cpu.concat_prefix(img, sos, cap).half().float().to(DEV)        # exactly representable in fp16
    if layout == "fp16":
        odd = x.half()
    elif layout == "inner_stride":
        odd = torch.empty(sh.S, sh.B, sh.d * 2, device=DEV)[..., ::2]
        odd.copy_(x)
        assert odd.stride(2) == 2
    else:
        flat = torch.empty(x.numel() + 1, device=DEV)
        odd = flat[1:].view(sh.S, sh.B, sh.d)
        odd.copy_(x)
        assert odd.data_ptr() % 16 != 0
    m = make_model(sh, sd, F32)
    m.train()
    want, _ = m.caption_loss(x, labels.to(DEV), pos=pos)
    got, _ = m.caption_loss(odd, labels.to(DEV), pos=pos)
    none, _ = m.caption_loss(odd, labels.to(DEV))
    assert got.item() == want.item()
    assert abs(none.item() - want.item()) > 1e-4           # the positional term is not a no-op on this input
    lg = ref_cpu.decoder_forward(x.cpu() + pos.cpu().unsqueeze(1), *ref_cpu.split_params(sd), sh.nhead)
    lo, _, _ = ref_cpu.caption_loss(lg, labels, sh.L)
    assert abs(got.item() - lo.item()) <= 1e-5 * lo.item()
