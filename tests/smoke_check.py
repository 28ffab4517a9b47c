"""One small invocation of the hot path on cuda:0 checked against the CPU oracle:
a fused training step (bf16 MFMA path) and a greedy decode (f32 validation path)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run():
    import helpers
    from helpers import TINY as sh
    from image_caption_amd.decoder import TransformerDecoder
    from image_caption_amd.engine import EngineOptions
    from oracle import ref_cpu

    dev = "cuda:0"
    sd = helpers.make_state_dict(sh, seed=7, head_gain=4.0)
    img, sos, cap, labels = helpers.make_embeddings(sh, seed=8)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    blocks, head = ref_cpu.split_params(leaf)
    lg = ref_cpu.decoder_forward(ref_cpu.concat_prefix(img, sos, cap), blocks, head, sh.nhead)
    ref_loss, _, _ = ref_cpu.caption_loss(lg, labels, sh.L)
    ref_loss.backward()

    m = TransformerDecoder(sh.d, sh.nhead, 6, sh.ff, 0.0, sh.vocab, options=EngineOptions(compute_dtype=torch.bfloat16))
    m.load_state_dict(sd)
    m.to(dev).train()
    loss, _ = m.caption_loss((img.to(dev), sos.to(dev), cap.to(dev)), labels.to(dev))
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-2 * ref_loss.item(), (loss.item(), ref_loss.item())
    for k, p in m.named_parameters():
        a, b = p.grad.cpu().flatten().double(), leaf[k].grad.flatten().double()
        c = float(a @ b / (a.norm() * b.norm() + 1e-30))
        # 0.999, except for the tensors read straight through the ReLU's on/off pattern: bf16 flips the gates of the
        # pre-activations nearest zero, which alone puts those four tensors at 0.9976-0.9993 at this 64-wide shape
        gated = k.endswith(("linear1.weight", "linear1.bias", "norm2.weight", "norm2.bias"))
        assert c >= (0.9975 if gated else 0.999), (k, c)

    m32 = TransformerDecoder(sh.d, sh.nhead, 6, sh.ff, 0.0, sh.vocab, options=EngineOptions(compute_dtype=torch.float32))
    m32.load_state_dict(sd)
    m32.to(dev).eval()
    emb = helpers.make_token_table(sh, seed=9)
    ids = m32.generate(img.to(dev), sos.to(dev), emb.to(dev), max_len=sh.L).cpu()
    V = sh.vocab
    with torch.no_grad():
        ref_ids, margins = ref_cpu.greedy_decode(img, sos, emb, *ref_cpu.split_params(sd), sh.nhead, sh.L, V - 3, V - 2, V - 1,
                                                 return_margins=True)
    safe = (margins > 1e-3).cumprod(dim=1).bool()
    assert torch.equal(ids[safe], ref_ids[safe]), "greedy ids differ from the oracle"
    print(f"smoke ok: loss {loss.item():.4f} (oracle {ref_loss.item():.4f}); greedy ids match on {int(safe.sum())}/{safe.numel()} decided steps")


if __name__ == "__main__":
    run()
