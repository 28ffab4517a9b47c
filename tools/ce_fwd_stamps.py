#!/usr/bin/env python3
"""Clock stamps of the logits + cross-entropy forward under a forced kernel geometry.
Usage (on the GPU box): python tools/ce_fwd_stamps.py [modes...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402
import gemm_stamps  # noqa: E402

lib = _lib.load()
lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
lib.capdecdbg_gemm_mode.restype = None
DEV = "cuda"


def main():
    modes = [int(a) for a in sys.argv[1:]] or [0, 5]
    B, L, d, V = 256, 32, 768, 50260
    Mt = B * L
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(Mt, d, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(V, d, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.zeros(V, device=DEV)
    labels = torch.randint(0, V, (B, L), device=DEV, generator=g)
    ws = torch.empty(ops.logits_ce_workspace(Mt, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(Mt, device=DEV), torch.empty(Mt, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    probs = torch.empty(Mt, (V + 127) // 128 * 128, device=DEV, dtype=torch.bfloat16)
    nt = (Mt // 128) * ((V + 127) // 128)
    for m in modes:
        lib.capdecdbg_gemm_mode(m)
        gemm_stamps.stamp_call(f"mode {m}: logits+CE fwd", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws), nt, d // 32)
        gemm_stamps.stamp_call(f"mode {m}: logits+CE fwd keeping numerators",
                               lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs), nt, d // 32)
    lib.capdecdbg_gemm_mode(0)


if __name__ == "__main__":
    main()
