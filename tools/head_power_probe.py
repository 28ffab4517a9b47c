#!/usr/bin/env python3
"""How much of the distance between the vocabulary product and the 2.5 PFLOP/s MFMA peak is the chip's power envelope:
the fused logits + cross-entropy forward (8192 x 50260 x 768, numerators kept) and the vendor library's bare product of
the same shape (torch.matmul -> hipBLASLt, bf16 output, no loss), each on training-like random operands and on all-zero
operands (same instruction stream, almost no switching activity in the multipliers), plus the library's 8192^3.
Diagnostic only: nothing in the package calls torch.matmul.   Usage (on the GPU box): python tools/head_power_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    B, L, d, V = 256, 32, 768, 50260
    M = B * L
    Vp = (V + 127) // 128 * 128
    g = torch.Generator(device=DEV).manual_seed(0)
    labels = torch.randint(0, V, (B, L), device=DEV, generator=g)
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    rowref = torch.empty(M, device=DEV)
    probs = torch.empty(Vp // 32, M, 32, device=DEV, dtype=BF)
    out = torch.empty(M, V, device=DEV, dtype=BF)
    a8 = torch.empty(8192, 8192, device=DEV, dtype=BF)
    b8 = torch.empty(8192, 8192, device=DEV, dtype=BF)
    c8 = torch.empty(8192, 8192, device=DEV, dtype=BF)
    fl = 2.0 * M * V * d
    print("operands | fused logits+CE forward (whole entry point) us (TFLOP/s) | library bare product us (TFLOP/s) | library 8192^3 us (TFLOP/s)")
    for rnd in range(2):
        for kind in ("random", "zeros"):
            if kind == "random":
                h = torch.randn(M, d, device=DEV, generator=g).to(BF)
                w = (torch.randn(V, d, device=DEV, generator=g) * 0.03).to(BF)
                bias = torch.randn(V, device=DEV, generator=g) * 0.03
                a8.copy_(torch.randn(8192, 8192, device=DEV, generator=g))
                b8.copy_(torch.randn(8192, 8192, device=DEV, generator=g))
            else:
                h, w, bias = torch.zeros(M, d, device=DEV, dtype=BF), torch.zeros(V, d, device=DEV, dtype=BF), torch.zeros(V, device=DEV)
                a8.zero_(); b8.zero_()
            hp = ops.pack_kpanels(h, torch.empty(ops.kpanel_shape(M, d), device=DEV, dtype=BF))
            wp = ops.pack_kpanels(w, torch.empty(ops.kpanel_shape(V, d), device=DEV, dtype=BF))
            t_f = timeit(lambda: ops.logits_ce_fwd_ref(h, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs, rowref, h_panels=hp, w_panels=wp))
            t_l = timeit(lambda: torch.matmul(h, w.t(), out=out))
            t_s = timeit(lambda: torch.matmul(a8, b8.t(), out=c8))
            print(f"{kind:8s} | {t_f * 1e3:8.1f} ({fl / t_f / 1e9:6.1f}) | {t_l * 1e3:8.1f} ({fl / t_l / 1e9:6.1f}) | {t_s * 1e3:8.1f} ({2.0 * 8192 ** 3 / t_s / 1e9:6.1f})")


if __name__ == "__main__":
    main()
