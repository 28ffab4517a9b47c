#!/usr/bin/env python3
"""Data-gradient products dX = dY W with W [out, in] as stored (reduction-major second operand, read back through the
transposing LDS read) against the same product on a transposed copy W^T [in, out] (row-major second operand).
Usage (GPU box): python tools/dgrad_layout_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    cases = [("head dgrad", 8192, 768, 50304, torch.float32, True),
             ("12-block linear2 dgrad", 74240, 2048, 768, torch.bfloat16, False),
             ("12-block linear1 dgrad", 74240, 768, 2048, torch.bfloat16, False),
             ("12-block in_proj dgrad", 74240, 768, 2304, torch.bfloat16, False),
             ("12-block out_proj dgrad", 74240, 768, 768, torch.bfloat16, False),
             ("1-block linear2 dgrad", 8192, 2048, 768, torch.bfloat16, False),
             ("1-block linear1 dgrad", 8192, 768, 2048, torch.bfloat16, False),
             ("1-block out_proj dgrad", 8192, 768, 768, torch.bfloat16, False)]
    print(f"{'product':28s} {'M x N x K':>22s} | W as stored: us (TFLOP/s) | W^T copy: us (TFLOP/s)")
    for name, M, N, K, odt, use_ws in cases:
        dy = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
        W = (0.05 * torch.randn(K, N, device=DEV, generator=g)).to(torch.bfloat16)          # [out = K, in = N]
        Wt = W.t().contiguous()
        out = torch.empty(M, N, device=DEV, dtype=odt)
        ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device=DEV) if use_ws else None
        a = timeit(lambda: ops.gemm(dy, W, out, M=M, N=N, K=K, b_kmajor=True, workspace=ws))
        ref = out.float().clone()
        b = timeit(lambda: ops.gemm(dy, Wt, out, M=M, N=N, K=K, b_kmajor=False, workspace=ws))
        err = (out.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
        fl = 2.0 * M * N * K
        print(f"{name:28s} {f'{M} x {N} x {K}':>22s} | {a:8.1f} ({fl / a / 1e6:7.1f}) | {b:8.1f} ({fl / b / 1e6:7.1f})   rel diff {err:.1e}")


if __name__ == "__main__":
    main()
