#!/usr/bin/env python3
"""The block products of the one-block step that run on the 8192 scored rows (batch 256), each alone, under the default
kernel choice and with the persistent kernel forced (capdecdbg_gemm_mode 10 / 11: 256x256 / 256x128 tiles).
Usage (on the GPU box): python tools/mlp_gemm_bench.py"""
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
    x768, x2048 = r(M, d).to(BF), r(M, ff).to(BF)
    w1, w2, wo, wq = (r(ff, d) * 0.03).to(BF), (r(d, ff) * 0.03).to(BF), (r(d, d) * 0.03).to(BF), (r(d, d) * 0.03).to(BF)
    b768, b2048 = r(d), r(ff)
    res = r(M, d)
    y768b, y2048b, y768f = torch.empty(M, d, device=DEV, dtype=BF), torch.empty(M, ff, device=DEV, dtype=BF), torch.empty(M, d, device=DEV)
    drop = ops.dropout_keys(1, 2, 3, 0.1)
    big = torch.empty(1 << 28, device=DEV)
    tfill = timeit(lambda: big.fill_(1.0), 5)
    cases = [
        ("Q projection 8192x768x768 (bias, scale -> bf16)", 2.0 * M * d * d, lambda: ops.gemm(x768, wq, y768b, bias=b768, col_scale=0.1, col_scale_n=d)),
        ("out_proj 8192x768x768 (bias, dropout, residual -> f32)", 2.0 * M * d * d, lambda: ops.gemm(x768, wo, y768f, bias=b768, drop=drop, residual=res)),
        ("linear1 8192x2048x768 (bias, ReLU, dropout -> bf16)", 2.0 * M * ff * d, lambda: ops.gemm(x768, w1, y2048b, bias=b2048, act=_lib.ACT_RELU, drop=drop)),
        ("linear2 8192x768x2048 (bias, dropout, residual -> f32)", 2.0 * M * ff * d, lambda: ops.gemm(x2048, w2, y768f, bias=b768, drop=drop, residual=res)),
    ]
    print("product | default: warm us (TFLOP/s), from cold caches | persistent 256x256 | persistent 256x128")
    for name, fl, fn in cases:
        out = []
        for mode in (0, 10, 11):
            lib.capdecdbg_gemm_mode(mode)
            tw = timeit(fn)

            def cold():
                big.fill_(1.0)
                fn()
            tc = timeit(cold, 5) - tfill
            out.append(f"{tw:6.1f} ({fl / tw / 1e6:6.1f}), {tc:6.1f}")
        lib.capdecdbg_gemm_mode(0)
        print(f"{name:56s} | " + " | ".join(out))


if __name__ == "__main__":
    main()
