#!/usr/bin/env python3
"""Timing of the vocabulary stage's two backward products at batch 256 with d(logits) read as a row-major matrix
against the same launches reading it as a column-panel image [V/32][M][32] (the values are not rearranged here: timing
only).  Usage (on the GPU box): python tools/head_gemm_panel_ab.py"""
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
    lib = _lib.load()
    lib.capdecdbg_gemm_a_panel.argtypes = [ctypes.c_long]
    lib.capdecdbg_gemm_a_panel.restype = None
    M, d, V, Vp = 8192, 768, 50260, 50304
    g = torch.Generator(device=DEV).manual_seed(0)
    dl = (torch.randn(M, Vp, device=DEV, generator=g) * 0.01).to(torch.bfloat16)
    dl[:, V:] = 0
    W = (torch.randn(Vp, d, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    h = torch.randn(M, d, device=DEV, generator=g).to(torch.bfloat16)
    dH = torch.empty(M, d, device=DEV)
    gW = torch.empty(V, d, device=DEV)
    ws = torch.empty(8 * M * d * 4, dtype=torch.uint8, device=DEV)
    big = torch.empty(1 << 28, device=DEV)
    tfill = timeit(lambda: big.fill_(1.0), 5)
    fl = 2.0 * M * d * V

    def dgrad(cold):
        if cold:
            big.fill_(1.0)
        ops.gemm(dl, W, dH, M=M, N=d, K=Vp, b_kmajor=True, workspace=ws)

    def wgrad(cold):
        if cold:
            big.fill_(1.0)
        ops.gemm(dl[:, :V], h, gW, M=V, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws)

    for _ in range(3):
        for rows in (0, M):
            lib.capdecdbg_gemm_a_panel(rows)
            td, tw = timeit(lambda: dgrad(False)), timeit(lambda: wgrad(False))
            tdc, twc = timeit(lambda: dgrad(True), 5) - tfill, timeit(lambda: wgrad(True), 5) - tfill
            lib.capdecdbg_gemm_a_panel(0)
            print(f"{'panels' if rows else 'matrix'}: dgrad {td * 1e3:7.1f} us ({fl / td / 1e9:6.1f} TFLOP/s), cold {tdc * 1e3:7.1f}   "
                  f"wgrad {tw * 1e3:7.1f} us ({fl / tw / 1e9:6.1f} TFLOP/s), cold {twc * 1e3:7.1f}")


if __name__ == "__main__":
    main()
