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
    lib.capdecdbg_gemm_stamps.argtypes = [ctypes.c_void_p]

    def persistent(m, keep):
        """The persistent kernel (modes 8 / 9) stamps per WORKGROUP: start, cycles summed over its tiles' k-loops and
        epilogues, end."""
        st = torch.zeros(512 * 8, dtype=torch.int64, device=DEV)
        fn = lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs if keep else None)  # noqa: E731
        fn(); fn()
        torch.cuda.synchronize()
        lib.capdecdbg_gemm_stamps(st.data_ptr())
        fn()
        torch.cuda.synchronize()
        lib.capdecdbg_gemm_stamps(None)
        s = st.view(512, 8).double().cpu()
        s = s[s[:, 7] > 0]
        tot = s[:, 7] - s[:, 0]
        tiles = (Mt // 256) * ((V + (127 if m == 8 else 255)) // (128 if m == 8 else 256)) / s.shape[0]
        print(f"mode {m}{' keeping numerators' if keep else ''}: {s.shape[0]} workgroups, {tiles:.1f} tiles each; per workgroup: total {tot.median():.0f} "
              f"(min {tot.min():.0f}, max {tot.max():.0f}), k-loops {s[:, 1].median():.0f} ({s[:, 1].median() / tiles / (d // 32):.0f} per step), "
              f"epilogues {s[:, 2].median():.0f} ({s[:, 2].median() / tiles:.0f} per tile); span {s[:, 7].max() - s[:, 0].min():.0f} cycles")

    for m in modes:
        lib.capdecdbg_gemm_mode(m)
        if m in (8, 9):
            persistent(m, False)
            persistent(m, True)
            continue
        gemm_stamps.stamp_call(f"mode {m}: logits+CE fwd", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws), nt, d // 32)
        gemm_stamps.stamp_call(f"mode {m}: logits+CE fwd keeping numerators",
                               lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs), nt, d // 32)
    lib.capdecdbg_gemm_mode(0)


if __name__ == "__main__":
    main()
