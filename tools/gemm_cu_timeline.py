#!/usr/bin/env python3
"""Per-CU phase overlap of the vocabulary-stage forward kernel, from the in-kernel clock stamps:
how much of a compute unit's time has its two resident workgroups both in the main loop, one in the main
loop and one in the epilogue, both in the epilogue, or fewer than two workgroups resident.
Usage (on the GPU box): python tools/gemm_cu_timeline.py"""
import ctypes
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"
lib = _lib.load()
lib.capdecdbg_gemm_stamps.argtypes = [ctypes.c_void_p]


def analyse(name, fn, ntiles):
    st = torch.zeros(ntiles * 8, dtype=torch.int64, device=DEV)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(st.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(None)
    s = st.view(ntiles, 8).cpu()
    s = s[s[:, 0] > 0]
    cus = defaultdict(list)
    for row in s.tolist():
        hw = row[6]
        xcc, low = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
        cu, sh, se = (low >> 8) & 0xF, (low >> 12) & 1, (low >> 13) & 7
        cus[(xcc, se, sh, cu)].append((row[0], row[2], row[7]))          # start, main loop end, end
    tot = defaultdict(float)
    for key, wgs in cus.items():
        ev = []
        for a, b, c in wgs:
            ev += [(a, "m", 1), (b, "m", -1), (b, "e", 1), (c, "e", -1)]
        ev.sort()
        nm = ne = 0
        prev = ev[0][0]
        for t, kind, dlt in ev:
            tot[(nm, ne)] += t - prev
            prev = t
            if kind == "m":
                nm += dlt
            else:
                ne += dlt
    total = sum(tot.values())
    print(f"{name}: {len(cus)} compute units, {len(s)} workgroups, {len(s) / len(cus):.1f} per CU")
    for (nm, ne), v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"    {nm} in prologue/main loop, {ne} in epilogue: {100 * v / total:5.1f} % of CU time")
    # example: the first CU's first workgroups
    key = sorted(cus)[0]
    t0 = min(w[0] for w in cus[key])
    print("    first CU:", " ".join(f"[{a - t0}..{b - t0}..{c - t0}]" for a, b, c in sorted(cus[key])[:8]))


def main():
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
    analyse("logits+CE fwd keeping numerators", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs), nt)
    lib.capdecdbg_gemm_extra_lds.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_extra_lds(24 * 1024)
    analyse("logits+CE fwd keeping numerators, ONE workgroup per CU", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs), nt)
    lib.capdecdbg_gemm_extra_lds(0)
    analyse("logits+CE fwd", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws), nt)
    xa = torch.randn(58624, d, device=DEV, generator=g).to(torch.bfloat16)
    wkv = (torch.randn(2 * d, d, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    kv = torch.empty(58624, 2 * d, device=DEV, dtype=torch.bfloat16)
    analyse("KV proj 58624x1536x768", lambda: ops.gemm(xa, wkv, kv, bias=bias[:2 * d]), (58624 // 128) * 12)


if __name__ == "__main__":
    main()
