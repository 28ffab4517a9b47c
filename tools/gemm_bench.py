#!/usr/bin/env python3
"""Micro-benchmark of the GEMM shapes one training step at batch 256 launches.
Usage (on the GPU box): python tools/gemm_bench.py [reps]"""
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B, S, d, ff, V, L = 256, 229, 768, 2048, 50260, 32
    M, Mt = S * B, L * B
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).to(BF)     # noqa: E731
    rows = []
    skws = torch.empty(32 * 2304 * 768 * 4, dtype=torch.uint8, device=DEV)

    def add(name, flops, fn):
        ms = timeit(fn, reps)
        rows.append((name, ms, flops / ms / 1e9))

    # activations / weights
    x_all, x_t = r(M, d), r(Mt, d)
    w_kv, w_q, w1, w2, wv = r(2 * d, d) * 0.03, r(d, d) * 0.03, r(ff, d) * 0.03, r(d, ff) * 0.03, r(V, d) * 0.03
    bias_kv, bias_ff, bias_d, bias_v = [torch.zeros(n, device=DEV) for n in (2 * d, ff, d, V)]
    kv = torch.empty(M, 2 * d, device=DEV, dtype=BF)
    add("fwd  KV proj      M=58624 N=1536 K=768 ", 2.0 * M * 2 * d * d, lambda: ops.gemm(x_all, w_kv, kv, bias=bias_kv))
    a1 = torch.empty(Mt, ff, device=DEV, dtype=BF)
    drop = ops.dropout_keys(1, 2, 3, 0.1)
    add("fwd  linear1+relu+drop M=8192 N=2048 K=768", 2.0 * Mt * ff * d,
        lambda: ops.gemm(x_t, w1, a1, bias=bias_ff, act=_lib.ACT_RELU, drop=drop))
    res = torch.randn(Mt, d, device=DEV, generator=g)
    xo = torch.empty(Mt, d, device=DEV)
    add("fwd  linear2+drop+res  M=8192 N=768 K=2048", 2.0 * Mt * ff * d, lambda: ops.gemm(a1, w2, xo, bias=bias_d, drop=drop, residual=res))
    # vocabulary stage
    labels = torch.randint(0, V, (B, L), device=DEV, generator=g)
    ws = torch.empty(ops.logits_ce_workspace(Mt, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(Mt, device=DEV), torch.empty(Mt, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    add("head logits+CE fwd     M=8192 N=50260 K=768", 2.0 * Mt * V * d, lambda: ops.logits_ce_fwd(x_t, wv, bias_v, labels, B, L, lse, amax, ls, cr, ws))
    slab = 8192
    dl = torch.empty(Mt, slab, device=DEV, dtype=BF)
    dbias = torch.zeros(V, device=DEV)
    bws = torch.empty(ops.logits_ce_bwd_workspace(Mt, 8192), dtype=torch.uint8, device=DEV)
    add("head CE bwd slab       M=8192 N=8192 K=768", 2.0 * Mt * slab * d,
        lambda: ops.logits_ce_bwd_slab(x_t, wv, bias_v, labels, B, L, lse, 1.0 / Mt, 0, slab, dl, dbias, bws))
    dH = torch.empty(Mt, d, device=DEV)
    add("head dgrad slab        M=8192 N=768 K=8192 (B kmajor)", 2.0 * Mt * slab * d, lambda: ops.gemm(dl, wv[:slab], dH, b_kmajor=True))
    dW = torch.empty(slab, d, device=DEV)
    add("head wgrad slab        M=8192 N=768 K=8192 (A,B kmajor)", 2.0 * Mt * slab * d, lambda: ops.gemm(dl, x_t, dW, a_kmajor=True, b_kmajor=True))
    # block backward on all rows
    dkv = r(M, 2 * d)
    gkv = torch.empty(2 * d, d, device=DEV)
    add("bwd  wgrad KV          M=1536 N=768 K=58624 (A,B kmajor)", 2.0 * M * 2 * d * d, lambda: ops.gemm(dkv, x_all, gkv, a_kmajor=True, b_kmajor=True, workspace=skws))
    dh1 = torch.empty(M, d, device=DEV)
    add("bwd  dgrad KV          M=58624 N=768 K=1536 (B kmajor)", 2.0 * M * 2 * d * d, lambda: ops.gemm(dkv, w_kv, dh1, b_kmajor=True))
    g2 = r(Mt, d)
    gw2 = torch.empty(d, ff, device=DEV)
    add("bwd  wgrad linear2     M=768 N=2048 K=8192 (A,B kmajor)", 2.0 * Mt * ff * d, lambda: ops.gemm(g2, a1, gw2, a_kmajor=True, b_kmajor=True, workspace=skws))
    # square reference points
    for n in (4096, 8192):
        a, b = r(n, n), r(n, n)
        c = torch.empty(n, n, device=DEV, dtype=BF)
        add(f"ref  square NT         {n}^3", 2.0 * n ** 3, lambda: ops.gemm(a, b, c))
    for name, ms, tf in rows:
        print(f"{name:60s} {ms * 1e3:9.1f} us  {tf:7.1f} TFLOP/s")
    # A/B of the row-major kernels in one process, interleaved: register-staged / 128x128 ring / 256x128 ring
    import ctypes
    lib = _lib.load()
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    a8, b8 = r(8192, 8192), r(8192, 8192)
    c8 = torch.empty(8192, 8192, device=DEV, dtype=BF)
    cases = {
        "KV proj 58624x1536x768": (2.0 * M * 2 * d * d, lambda: ops.gemm(x_all, w_kv, kv, bias=bias_kv)),
        "logits+CE fwd 8192x50260x768": (2.0 * Mt * V * d, lambda: ops.logits_ce_fwd(x_t, wv, bias_v, labels, B, L, lse, amax, ls, cr, ws)),
        "CE bwd slab 8192x8192x768": (2.0 * Mt * slab * d, lambda: ops.logits_ce_bwd_slab(x_t, wv, bias_v, labels, B, L, lse, 1.0 / Mt, 0, slab, dl, dbias, bws)),
        "linear1 8192x2048x768": (2.0 * Mt * ff * d, lambda: ops.gemm(x_t, w1, a1, bias=bias_ff, act=_lib.ACT_RELU, drop=drop)),
        "square 8192^3": (2.0 * 8192 ** 3, lambda: ops.gemm(a8, b8, c8)),
    }
    probs = torch.empty(Mt, (V + 127) // 128 * 128, device=DEV, dtype=BF)
    cases["logits+CE fwd keeping numerators"] = (2.0 * Mt * V * d, lambda: ops.logits_ce_fwd(x_t, wv, bias_v, labels, B, L, lse, amax, ls, cr, ws, probs=probs))
    t_with = timeit(lambda: ops.logits_ce_bwd_slab(x_t, wv, bias_v, labels, B, L, lse, 1.0 / Mt, 0, slab, dl, dbias, bws), 20)
    t_without = timeit(lambda: ops.logits_ce_bwd_slab(x_t, wv, bias_v, labels, B, L, lse, 1.0 / Mt, 0, slab, dl, None), 20)
    print(f"CE bwd slab: with bias gradient {t_with * 1e3:.1f} us, without {t_without * 1e3:.1f} us")
    print("A/B (TFLOP/s, median of 5 interleaved rounds): register | ring128 | ring256 (8 waves) | ring256 (4 waves, 2 WG/CU) | ring 256x256 | pipelined 256x256 (both plain products only) | pipelined 256x128")
    Vp = (V + 127) // 128 * 128
    dlw = torch.zeros(Mt, Vp, device=DEV, dtype=BF)
    dlw[:, :V] = (torch.randn(Mt, V, device=DEV, generator=g) * 1e-3).to(BF)
    wvp = torch.zeros(Vp, d, device=DEV, dtype=BF)
    wvp[:V] = wv
    dWw = torch.empty(V, d, device=DEV)
    hws = torch.empty(8 * Mt * d * 4, dtype=torch.uint8, device=DEV)
    M12 = 290 * 256
    dqkv12, x12 = r(M12, 3 * d), r(M12, d)
    gin12, gl2 = torch.empty(3 * d, d, device=DEV), torch.empty(d, ff, device=DEV)
    kcases = {
        "head dgrad slab 8192x768x8192 (B kmajor)": (2.0 * Mt * slab * d, lambda: ops.gemm(dl, wv[:slab], dH, b_kmajor=True)),
        "head dgrad 8192x768x50304 (B kmajor, split-K)": (2.0 * Mt * Vp * d, lambda: ops.gemm(dlw, wvp, dH, M=Mt, N=d, K=Vp, b_kmajor=True, workspace=hws)),
        "head wgrad slab 8192x768x8192 (A,B kmajor)": (2.0 * Mt * slab * d, lambda: ops.gemm(dl, x_t, dW, a_kmajor=True, b_kmajor=True)),
        "head wgrad 50260x768x8192 (A,B kmajor)": (2.0 * Mt * V * d, lambda: ops.gemm(dlw, x_t, dWw, M=V, N=d, K=Mt, a_kmajor=True, b_kmajor=True)),
        "dgrad KV 58624x768x1536 (B kmajor)": (2.0 * M * 2 * d * d, lambda: ops.gemm(dkv, w_kv, dh1, b_kmajor=True)),
        "wgrad KV 1536x768x58624 (A,B kmajor, split-K)": (2.0 * M * 2 * d * d, lambda: ops.gemm(dkv, x_all, gkv, a_kmajor=True, b_kmajor=True, workspace=skws)),
        # all rows through a block (stacked blocks): S = 290, M = 74240
        "wgrad in_proj 2304x768x74240 (split-K)": (2.0 * M12 * 3 * d * d, lambda: ops.gemm(dqkv12, x12, gin12, a_kmajor=True, b_kmajor=True, workspace=skws)),
        "wgrad out_proj 768x768x74240 (split-K)": (2.0 * M12 * d * d, lambda: ops.gemm(dqkv12[:, :d], x12, gin12[:d], a_kmajor=True, b_kmajor=True, workspace=skws)),
        "wgrad linear1 2048x768x74240 (split-K)": (2.0 * M12 * ff * d, lambda: ops.gemm(dqkv12[:, :ff], x12, gin12[:ff], a_kmajor=True, b_kmajor=True, workspace=skws)),
        "wgrad linear2 768x2048x74240 (split-K)": (2.0 * M12 * ff * d, lambda: ops.gemm(x12, dqkv12[:, :ff], gl2, a_kmajor=True, b_kmajor=True, workspace=skws)),
    }
    print("reduction-major operands: register | ring128 | ring 256x128 | ring 256x256 | quad 256x256 | default")
    for name, (fl, fn) in kcases.items():
        res = {1: [], 2: [], 4: [], 5: [], 6: [], 0: []}
        for _ in range(5):
            for mode in (1, 2, 4, 5, 6, 0):
                lib.capdecdbg_gemm_mode(mode)
                res[mode].append(fl / timeit(fn, 5) / 1e9)
        lib.capdecdbg_gemm_mode(0)
        print(f"{name:48s} " + " | ".join(f"{sorted(res[m])[2]:7.1f}" for m in (1, 2, 4, 5, 6, 0)))
    for name, (fl, fn) in cases.items():
        res = {1: [], 2: [], 3: [], 4: [], 5: [], 6: [], 7: []}
        for _ in range(5):
            for mode in (1, 2, 3, 4, 5, 6, 7):
                lib.capdecdbg_gemm_mode(mode)
                res[mode].append(fl / timeit(fn, 5) / 1e9)
        lib.capdecdbg_gemm_mode(0)
        med = {m: sorted(v)[2] for m, v in res.items()}
        print(f"{name:34s} {med[1]:7.1f} | {med[2]:7.1f} | {med[3]:7.1f} | {med[4]:7.1f} | {med[5]:7.1f} | {med[6]:7.1f} | {med[7]:7.1f}")


if __name__ == "__main__":
    main()
