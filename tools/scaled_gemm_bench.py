#!/usr/bin/env python3
"""Micro-benchmark of the GEMM shapes of one block of the 12-block / 257-patch configuration at batch 256
(74240 rows): forward, data-gradient and weight-gradient products, each as the engine launches it.
Usage (on the GPU box): python tools/scaled_gemm_bench.py [reps] [gemm mode] [0: without the persistent pipelined kernel]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, reps):
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = _lib.load()
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    lib.capdecdbg_gemm_ppipe.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_ppipe.restype = None
    ppipe = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lib.capdecdbg_gemm_ppipe(ppipe)
    B, S, d, ff = 256, 290, 768, 2048
    M = S * B
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).to(BF)     # noqa: E731
    rows = []

    def add(name, flops, fn):
        lib.capdecdbg_gemm_mode(mode)
        ms = timeit(fn, reps)
        lib.capdecdbg_gemm_mode(0)
        rows.append((name, ms, flops / ms / 1e9))

    x = r(M, d)
    w_in, w_o, w1, w2 = r(3 * d, d) * 0.03, r(d, d) * 0.03, r(ff, d) * 0.03, r(d, ff) * 0.03
    b_in, b_o, b1, b2 = [torch.zeros(n, device=DEV) for n in (3 * d, d, ff, d)]
    qkv = torch.empty(M, 3 * d, device=DEV, dtype=BF)
    a1 = torch.empty(M, ff, device=DEV, dtype=BF)
    res = torch.randn(M, d, device=DEV, generator=g)
    xo = torch.empty(M, d, device=DEV)
    drop = ops.dropout_keys(1, 2, 3, 0.1)
    skws = torch.empty(32 * max(3 * d * d, d * ff) * 4, dtype=torch.uint8, device=DEV)
    add("fwd  in_proj          N=2304 K=768 ", 2.0 * M * 3 * d * d, lambda: ops.gemm(x, w_in, qkv, bias=b_in))
    add("fwd  out_proj+drop+res N=768 K=768 ", 2.0 * M * d * d, lambda: ops.gemm(x, w_o, xo, bias=b_o, drop=drop, residual=res))
    add("fwd  linear1+relu+drop N=2048 K=768", 2.0 * M * ff * d, lambda: ops.gemm(x, w1, a1, bias=b1, act=_lib.ACT_RELU, drop=drop))
    add("fwd  linear2+drop+res N=768 K=2048 ", 2.0 * M * ff * d, lambda: ops.gemm(a1, w2, xo, bias=b2, drop=drop, residual=res))
    # data gradients: dX = dY . W  (W [N_out, K_in] read reduction-major)
    dx = torch.empty(M, d, device=DEV, dtype=BF)
    dxf = torch.empty(M, d, device=DEV)
    da1 = torch.empty(M, ff, device=DEV, dtype=BF)
    add("dgrad in_proj   N=768 K=2304       ", 2.0 * M * 3 * d * d, lambda: ops.gemm(qkv, w_in, dx, M=M, N=d, K=3 * d, b_kmajor=True))
    add("dgrad out_proj  N=768 K=768        ", 2.0 * M * d * d, lambda: ops.gemm(x, w_o, dx, M=M, N=d, K=d, b_kmajor=True))
    add("dgrad linear2 + relu gate N=2048 K=768", 2.0 * M * ff * d,
        lambda: ops.gemm(x, w2, da1, M=M, N=ff, K=d, b_kmajor=True, gate=a1, gate_mode=_lib.GATE_RELU_OUT, gate_drop=drop))
    add("dgrad linear1   N=768 K=2048 (f32 out)", 2.0 * M * ff * d, lambda: ops.gemm(a1, w1, dxf, M=M, N=d, K=ff, b_kmajor=True))
    add("dgrad linear1   N=768 K=2048 (bf16 out)", 2.0 * M * ff * d, lambda: ops.gemm(a1, w1, dx, M=M, N=d, K=ff, b_kmajor=True))
    # weight gradients: dW = dY^T . X, both operands reduction-major, split-K
    gw_in, gw_o, gw1, gw2 = [torch.empty_like(w, dtype=torch.float32) for w in (w_in, w_o, w1, w2)]
    add("wgrad in_proj   3d x d  K=74240    ", 2.0 * M * 3 * d * d, lambda: ops.gemm(qkv, x, gw_in, M=3 * d, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=skws))
    add("wgrad out_proj  d x d   K=74240    ", 2.0 * M * d * d, lambda: ops.gemm(x, x, gw_o, M=d, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=skws))
    add("wgrad linear1   ff x d  K=74240    ", 2.0 * M * ff * d, lambda: ops.gemm(a1, x, gw1, M=ff, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=skws))
    add("wgrad linear2   d x ff  K=74240    ", 2.0 * M * ff * d, lambda: ops.gemm(x, a1, gw2, M=d, N=ff, K=M, a_kmajor=True, b_kmajor=True, workspace=skws))
    print(f"gemm mode {mode}, persistent pipelined kernel {'on' if ppipe else 'off'}")
    tot = 0.0
    for name, ms, tf in rows:
        tot += ms
        print(f"{name:44s} {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s")
    print(f"sum {tot:.3f} ms per block")


if __name__ == "__main__":
    main()
