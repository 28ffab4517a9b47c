"""Generate the golden vectors from the reference's own TransformerDecoder.

Run in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

The reference module is imported as it lies on disk (its one missing import,
``wandb``, is only used by functions that are never called here, so an empty
stand-in module object is registered for it).  Inputs and weights come from
``tests/helpers.py`` and can be regenerated from a seed, so only OUTPUTS of the
reference are stored.  Nothing of the reference's source is written anywhere.

Files written next to this script:
  tiny.npz        full tensors at a tiny shape: logits, loss, correct, all 14
                  grads, parameters after one torch.optim.Adam step
  default_b8.npz  config.py defaults, B=8: loss, correct, strided logit slice,
                  per-parameter gradient norms
  mask.npz        create_caption_mask known answers
  composed.npz    build-defined modes obtained by COMPOSING the reference class:
                  prefix-causal mask, two stacked blocks, greedy decoding
  default_b8_grads.npz  the same default-shape step: every small gradient in full and a
                  strided slice of every matrix gradient (slicing: helpers.grad_slice)
  train_loop.npz  the per-split loss / accuracy history logged by the reference's own
                  train_decoder (decoder.py:96-216) on a small seeded dataset
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

import helpers
from helpers import DEFAULT_B8, TINY

REF = "/root/reference"


WANDB_LOG = []      # what the reference hands to wandb.log


def import_reference():
    # wandb is not installed here; the reference only uses init / log / finish, none of which carries arithmetic:
    # log records its argument, the other two do nothing
    w = sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    w.init = lambda *a, **k: None
    w.finish = lambda *a, **k: None
    w.log = lambda rec, *a, **k: WANDB_LOG.append(dict(rec))
    sys.path.insert(0, REF)
    import decoder as ref_decoder            # noqa: E402  (the reference's file)
    sys.path.remove(REF)
    return ref_decoder


def ref_model(ref, sh, sd, dropout=0.0):
    m = ref.TransformerDecoder(d_model=sh.d, nhead=sh.nhead, num_layers=6, dim_feedforward=sh.ff,
                               dropout=dropout, vocab_size=sh.vocab)
    m.load_state_dict(sd)
    return m


def step_like_reference(m, x, labels, L):
    """decoder.py:169-183 replayed around the imported class."""
    out = m(x, x)
    cap = out[-L:, :, :].permute(1, 0, 2)
    loss = torch.nn.CrossEntropyLoss()(cap.reshape(-1, cap.size(-1)), labels.reshape(-1))
    pred = torch.argmax(cap, dim=-1)
    correct = (pred == labels).sum().item()
    return out, loss, correct, pred


def concat_like_reference(img, sos, cap):
    """img_text_concat.py:115-123 (that module cannot be imported offline: it
    downloads models at import time), restated from its text."""
    return torch.cat([img, sos, cap], dim=1).transpose(0, 1)


def gen_tiny(ref):
    sh = TINY
    sd = helpers.make_state_dict(sh, seed=101)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=202)
    x = concat_like_reference(img, sos, cap).contiguous()
    m = ref_model(ref, sh, sd)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0.01)
    out, loss, correct, pred = step_like_reference(m, x, labels, sh.L)
    opt.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    out2, loss2, _, _ = step_like_reference(m, x, labels, sh.L)
    opt.zero_grad()
    loss2.backward()
    opt.step()
    d = {"logits": out.detach().numpy(), "loss": np.float32(loss.item()), "correct": np.int64(correct),
         "pred": pred.numpy(), "loss_step2": np.float32(loss2.item())}
    for k, g in grads.items():
        d["grad." + k] = g.numpy()
    for k, p in m.state_dict().items():
        d["after2." + k] = p.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **d)
    print("tiny: loss", loss.item(), "correct", correct)


def gen_default(ref):
    sh = DEFAULT_B8
    sd = helpers.make_state_dict(sh, seed=303)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=404)
    x = concat_like_reference(img, sos, cap).contiguous()
    m = ref_model(ref, sh, sd)
    m.train()
    out, loss, correct, pred = step_like_reference(m, x, labels, sh.L)
    loss.backward()
    d = {"loss": np.float32(loss.item()), "correct": np.int64(correct),
         "logits_tail_strided": out.detach()[-sh.L:, :, ::997].numpy(),
         "logits_row0_strided": out.detach()[0, :, ::997].numpy(),
         "logit_absmax": np.float32(out.detach().abs().max().item())}
    for k, p in m.named_parameters():
        d["gnorm." + k] = np.float32(p.grad.norm().item())
        d["gabsmax." + k] = np.float32(p.grad.abs().max().item())
    d["grad.norm2.weight"] = m.norm2.weight.grad.numpy()
    d["grad.self_attn.in_proj_bias"] = m.self_attn.in_proj_bias.grad.numpy()
    d["grad.output_projection.weight_rows"] = m.output_projection.weight.grad[::1009].numpy()
    np.savez_compressed(os.path.join(HERE, "default_b8.npz"), **d)
    print("default_b8: loss", loss.item(), "correct", correct, "absmax", d["logit_absmax"])
    dg = {"grad." + k: helpers.grad_slice(p.grad.detach()).numpy() for k, p in m.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "default_b8_grads.npz"), **dg)


def gen_train_loop(ref):
    """The reference's own train_decoder on CPU: what it logs per split and epoch."""
    cfg, data, seed = helpers.train_loop_case()
    rcfg = sys.modules["config"].DecoderConfig(**cfg, device="cpu")       # the reference's config class
    del WANDB_LOG[:]
    torch.manual_seed(seed)                  # the model is initialised from the global generator (decoder.py:125-132)
    ref.train_decoder(data, rcfg)
    hist = [r for r in WANDB_LOG if "epoch" in r]
    d = {"loss": np.array([[v for k, v in r.items() if k.endswith("_loss")][0] for r in hist], dtype=np.float64),
         "accuracy": np.array([[v for k, v in r.items() if k.endswith("_accuracy")][0] for r in hist], dtype=np.float64),
         "split": np.array([[k[:-5] for k in r if k.endswith("_loss")][0] for r in hist]),
         "epoch": np.array([r["epoch"] for r in hist], dtype=np.int64)}
    np.savez_compressed(os.path.join(HERE, "train_loop.npz"), **d)
    print("train_loop:", list(zip(d["split"], d["loss"].round(4), d["accuracy"].round(2))))


def gen_mask(ref):
    m = ref.TransformerDecoder(d_model=16, nhead=2, dim_feedforward=16, vocab_size=11)
    m229 = m.create_caption_mask(229)
    big = 12288 + 8
    mb = m.create_caption_mask(big)
    d = {"m229_unique": torch.unique(m229).numpy(),
         "big_len": np.int64(big),
         "big_corner": mb[-12:, -12:].numpy(),
         "big_neginf_count": np.int64(torch.isinf(mb).sum().item()),
         "big_left_nonzero": np.int64((mb[:, :12288] != 0).sum().item())}
    np.savez_compressed(os.path.join(HERE, "mask.npz"), **d)
    print("mask: unique229", d["m229_unique"], "neginf", d["big_neginf_count"])


def gen_composed(ref):
    from oracle import ref_cpu       # only for the prefix-causal mask VALUES and greedy loop shape
    sh = TINY
    d = {}
    # (1) prefix-causal: the reference instance with its mask method replaced
    sd = helpers.make_state_dict(sh, seed=505)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=606)
    x = concat_like_reference(img, sos, cap).contiguous()
    m = ref_model(ref, sh, sd)
    m.eval()
    pc = ref_cpu.caption_mask(sh.S, "prefix_causal", sh.P)
    m.create_caption_mask = lambda n: pc[:n, :n]
    with torch.no_grad():
        d["prefix_causal_logits"] = m(x, x).numpy()

    # (2) two stacked blocks: instance A with identity head feeding instance B
    sd_a = helpers.make_state_dict(sh, seed=707)
    sd_b = helpers.make_state_dict(sh, seed=808)
    ma, mb = ref_model(ref, sh, sd_a), ref_model(ref, sh, sd_b)
    ma.output_projection = torch.nn.Identity()
    ma.eval(); mb.eval()
    with torch.no_grad():
        h = ma(x, x)
        d["two_block_logits"] = mb(h, h).numpy()
    # two blocks + prefix causal + grads wrt block A's linear1 (checks chaining of backward)
    ma.create_caption_mask = lambda n: pc[:n, :n]
    mb.create_caption_mask = lambda n: pc[:n, :n]
    h = ma(x, x)
    out = mb(h, h)
    cap_o = out[-sh.L:].permute(1, 0, 2)
    loss = torch.nn.CrossEntropyLoss()(cap_o.reshape(-1, sh.vocab), labels.reshape(-1))
    loss.backward()
    d["two_block_pc_loss"] = np.float32(loss.item())
    d["two_block_pc_grad_a_linear1_weight"] = ma.linear1.weight.grad.numpy()
    d["two_block_pc_grad_b_in_proj_weight"] = mb.self_attn.in_proj_weight.grad.numpy()

    # (3) greedy decoding, full recompute around the reference forward
    def greedy(m, img, sos, emb, L, sos_id, eos_id, pad_id):
        B = img.size(0)
        cur = torch.cat([img, sos, emb[sos_id].expand(B, 1, -1)], dim=1)
        ids = torch.full((B, L), pad_id, dtype=torch.long)
        margins = torch.zeros(B, L)
        done = torch.zeros(B, dtype=torch.bool)
        with torch.no_grad():
            for t in range(L):
                xx = cur.transpose(0, 1)
                logits = m(xx, xx)[-1]
                top = logits.topk(2, dim=-1)
                tok = torch.where(done, torch.full_like(top.indices[:, 0], pad_id), top.indices[:, 0])
                margins[:, t] = top.values[:, 0] - top.values[:, 1]
                ids[:, t] = tok
                done |= tok == eos_id
                if t + 1 < L:
                    cur = torch.cat([cur, emb[tok][:, None, :]], dim=1)
        return ids, margins

    for name, shp, seeds, gain in (("tiny", TINY, (909, 1010, 1111), 4.0),
                                   ("default", DEFAULT_B8, (1212, 1313, 1414), 6.0)):
        sdg = helpers.make_state_dict(shp, seed=seeds[0], head_gain=gain)
        img, sos, _, _ = helpers.make_embeddings(shp, seed=seeds[1])
        emb = helpers.make_token_table(shp, seed=seeds[2])
        mg = ref_model(ref, shp, sdg)
        mg.eval()
        V = shp.vocab
        ids, margins = greedy(mg, img, sos, emb, shp.L, V - 3, V - 2, V - 1)
        d[f"greedy_{name}_ids"] = ids.numpy()
        d[f"greedy_{name}_margins"] = margins.numpy()
        print(f"greedy {name}: min margin {margins.min().item():.3e}, distinct tokens {ids.unique().numel()}")
    np.savez_compressed(os.path.join(HERE, "composed.npz"), **d)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = import_reference()
    gen_tiny(ref)
    gen_default(ref)
    gen_mask(ref)
    gen_composed(ref)
    gen_train_loop(ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
