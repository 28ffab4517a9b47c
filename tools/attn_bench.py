#!/usr/bin/env python3
"""Attention forward / backward timings at the two training shapes: scored rows of the one-block model
(S = 229, last 32 queries) and all rows of the stacked model (S = 290).  Usage (GPU box): python tools/attn_bench.py"""
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
    B, H, hd = 256, 8, 96
    d = H * hd
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, S, q_start, mode, prefix in (("one block, 32 scored queries, S=229", 229, 197, _lib.MASK_REFERENCE, 0),
                                            ("stacked, all 290 queries, prefix-causal", 290, 0, _lib.MASK_PREFIX_CAUSAL, 258)):
        nq = S - q_start
        qkv = (torch.randn(S * B, 3 * d, device=DEV, generator=g) * 0.5).to(BF)
        o = torch.zeros(S * B, d, device=DEV, dtype=BF)
        dout = torch.randn(S * B, d, device=DEV, generator=g).to(BF)
        dqkv = torch.zeros_like(qkv)
        lse, delta = torch.zeros(B * H * S, device=DEV), torch.zeros(B * H * S, device=DEV)
        for p in (0.1, 0.0):
            drop = ops.dropout_keys(1, 2, 3, p)
            kw = dict(q_start=q_start, nq=nq, mask_mode=mode, prefix_len=prefix, drop=drop)
            tf = timeit(lambda: ops.attn_fwd_packed(qkv, o, lse, B, H, S, hd, **kw))
            if nq <= 32 and hasattr(lib, "capdecdbg_attn_block_staging"):
                lib.capdecdbg_attn_block_staging(0)
                tfb = timeit(lambda: ops.attn_fwd_packed(qkv, o, lse, B, H, S, hd, **kw))
                tbb = timeit(lambda: ops.attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, **kw))
                lib.capdecdbg_attn_block_staging(1)
                print(f"   with the LDS-resident operand staged whole instead of in 64-key blocks: fwd {tfb:7.1f} us, bwd {tbb:7.1f} us")
            tb = timeit(lambda: ops.attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, **kw))
            fl = 4.0 * B * H * nq * S * hd
            kvb = 2.0 * B * S * d * 2
            print(f"{name:44s} p={p:.1f}  fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TFLOP/s, K+V at {kvb / tf / 1e6:5.2f} TB/s)   bwd (dq + dkv) {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
