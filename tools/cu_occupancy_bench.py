#!/usr/bin/env python3
"""How the vocabulary-stage and block products slow down while another kernel (a stand-in for an RCCL
all-reduce overlapped with the backward) holds some compute units.
Usage (on the GPU box): python tools/cu_occupancy_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def main():
    lib = _lib.load()
    lib.capdecdbg_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
    lib.capdecdbg_occupy.restype = ctypes.c_int
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    B, S, d, ff, V, L = 256, 229, 768, 2048, 50260, 32
    M, Mt = S * B, L * B
    Vp = (V + 127) // 128 * 128
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).to(BF)     # noqa: E731
    x_t, wv = r(Mt, d), torch.zeros(Vp, d, device=DEV, dtype=BF)
    wv[:V] = r(V, d) * 0.03
    dl = torch.zeros(Mt, Vp, device=DEV, dtype=BF)
    dl[:, :V] = (torch.randn(Mt, V, device=DEV, generator=g) * 1e-3).to(BF)
    dH, dW = torch.empty(Mt, d, device=DEV), torch.empty(V, d, device=DEV)
    hws = torch.empty(8 * Mt * d * 4, dtype=torch.uint8, device=DEV)
    dkv, w_kv, x_all = r(M, 2 * d), r(2 * d, d) * 0.03, r(M, d)
    dh1, gkv = torch.empty(M, d, device=DEV), torch.empty(2 * d, d, device=DEV)
    skws = torch.empty(8 * 2304 * 768 * 4, dtype=torch.uint8, device=DEV)
    cases = {
        "head dgrad 8192x768x50304 (split-K)": lambda: ops.gemm(dl, wv, dH, M=Mt, N=d, K=Vp, b_kmajor=True, workspace=hws),
        "head wgrad 50260x768x8192": lambda: ops.gemm(dl, x_t, dW, M=V, N=d, K=Mt, a_kmajor=True, b_kmajor=True),
        "dgrad KV 58624x768x1536": lambda: ops.gemm(dkv, w_kv, dh1, b_kmajor=True),
        "wgrad KV 1536x768x58624 (split-K)": lambda: ops.gemm(dkv, x_all, gkv, a_kmajor=True, b_kmajor=True, workspace=skws),
    }
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def timed(fn, wgs, lds):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            if wgs:
                rc = lib.capdecdbg_occupy(wgs, lds, 3000, ctypes.c_void_p(side.cuda_stream))
                assert rc == 0, rc
                torch.cuda._sleep(200000)            # the occupying workgroups are resident before the product starts
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(main_s)
            fn()
            e.record(main_s)
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        return sorted(ts)[3] * 1e3

    hogs = [(0, 0), (16, 8192), (32, 8192), (64, 8192), (16, 160 * 1024), (32, 160 * 1024), (64, 160 * 1024)]
    print("microseconds per launch; columns: alone | 16/32/64 co-resident workgroups (8 KiB LDS) | 16/32/64 exclusive CUs (160 KiB LDS)")
    for mode, label in ((0, "default geometry"), (4, "256x128 tiles forced")):
        lib.capdecdbg_gemm_mode(mode)
        print(label)
        for name, fn in cases.items():
            for _ in range(3):
                fn()
            row = [timed(fn, w, l) for w, l in hogs]
            print(f"  {name:40s} " + " | ".join(f"{t:7.1f}" for t in row))
    lib.capdecdbg_gemm_mode(0)


if __name__ == "__main__":
    main()
