#!/usr/bin/env python3
"""Timing of the vocabulary stage's two backward products at batch 256 under forced kernel choices
(capdecdbg_gemm_mode): data gradient 8192 x 768 over 50304 (split-K through a workspace) and weight gradient
50260 x 768 over 8192.  Usage (on the GPU box): python tools/head_gemm_modes.py [modes...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=10):
    for _ in range(3):
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
    modes = [int(a) for a in sys.argv[1:]] or [0, 6, 7]
    lib = _lib.load()
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    M, d, V, Vp = 8192, 768, 50260, 50304
    g = torch.Generator(device=DEV).manual_seed(0)
    dl = (torch.randn(M, Vp, device=DEV, generator=g) * 0.01).to(torch.bfloat16)
    dl[:, V:] = 0
    W = (torch.randn(Vp, d, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    h = torch.randn(M, d, device=DEV, generator=g).to(torch.bfloat16)
    dH = torch.empty(M, d, device=DEV)
    gW = torch.empty(V, d, device=DEV)
    ws = torch.empty(8 * M * d * 4, dtype=torch.uint8, device=DEV)
    fl = 2.0 * M * d * V
    for m in modes:
        lib.capdecdbg_gemm_mode(m)
        td = timeit(lambda: ops.gemm(dl, W, dH, M=M, N=d, K=Vp, b_kmajor=True, workspace=ws))
        tw = timeit(lambda: ops.gemm(dl[:, :V], h, gW, M=V, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws))
        lib.capdecdbg_gemm_mode(0)
        print(f"mode {m}: dgrad {td * 1e3:7.1f} us ({fl / td / 1e9:6.1f} TFLOP/s)   wgrad {tw * 1e3:7.1f} us ({fl / tw / 1e9:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
