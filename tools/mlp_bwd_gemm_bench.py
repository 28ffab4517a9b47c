#!/usr/bin/env python3
"""The backward products of the block on the 8192 scored rows (batch 256), each alone, under forced kernel choices
(capdecdbg_gemm_mode: 0 default, 1 register-staged, 2 ring 128x128, 4 ring 256x128, 5 ring 256x256, 6 pipelined 256x256,
7 pipelined 256x128).  Usage (on the GPU box): python tools/mlp_bwd_gemm_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    lib = _lib.load()
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    M, d, ff = 8192, 768, 2048
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)      # noqa: E731
    g768, g2048 = r(M, d).to(BF), r(M, ff).to(BF)
    a768, a2048 = r(M, d).to(BF), r(M, ff).to(BF)
    w1, w2, wo = (r(ff, d) * 0.03).to(BF), (r(d, ff) * 0.03).to(BF), (r(d, d) * 0.03).to(BF)
    o768b, o2048b, o768f = torch.empty(M, d, device=DEV, dtype=BF), torch.empty(M, ff, device=DEV, dtype=BF), torch.empty(M, d, device=DEV)
    gw1, gw2, gwo = torch.empty(ff, d, device=DEV), torch.empty(d, ff, device=DEV), torch.empty(d, d, device=DEV)
    ws = torch.empty(32 * max(3 * d * d, d * ff) * 4, dtype=torch.uint8, device=DEV)
    cases = [
        ("linear2 dgrad 8192x2048x768 (ReLU gate -> bf16)", 2.0 * M * ff * d,
         lambda: ops.gemm(g768, w2, o2048b, M=M, N=ff, K=d, b_kmajor=True, gate=a2048, gate_mode=_lib.GATE_RELU_OUT)),
        ("linear1 dgrad 8192x768x2048 (-> bf16)", 2.0 * M * ff * d, lambda: ops.gemm(g2048, w1, o768b, M=M, N=d, K=ff, b_kmajor=True)),
        ("out_proj dgrad 8192x768x768 (-> bf16)", 2.0 * M * d * d, lambda: ops.gemm(g768, wo, o768b, M=M, N=d, K=d, b_kmajor=True)),
        ("linear2 wgrad 768x2048x8192", 2.0 * M * ff * d, lambda: ops.gemm(g768, a2048, gw2, M=d, N=ff, K=M, a_kmajor=True, b_kmajor=True, workspace=ws)),
        ("linear1 wgrad 2048x768x8192", 2.0 * M * ff * d, lambda: ops.gemm(g2048, a768, gw1, M=ff, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws)),
        ("out_proj wgrad 768x768x8192", 2.0 * M * d * d, lambda: ops.gemm(g768, a768, gwo, M=d, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws)),
    ]
    modes = (0, 1, 2, 4, 5, 6, 7)
    print("product | us (TFLOP/s) under mode " + " | ".join(str(m) for m in modes))
    for name, fl, fn in cases:
        out = []
        for mode in modes:
            lib.capdecdbg_gemm_mode(mode)
            try:
                t = timeit(fn)
                out.append(f"{t:6.1f} ({fl / t / 1e6:5.0f})")
            except RuntimeError:
                out.append("     n/a     ")
        lib.capdecdbg_gemm_mode(0)
        print(f"{name:50s} | " + " | ".join(out))


if __name__ == "__main__":
    main()
