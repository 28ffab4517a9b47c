#!/usr/bin/env python3
"""Where a GEMM workgroup spends its cycles: prologue / main loop / epilogue, from
in-kernel clock stamps (diagnostic build hook capdecdbg_gemm_stamps)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"
lib = _lib.load()
lib.capdecdbg_gemm_stamps.argtypes = [ctypes.c_void_p]
lib.capdecdbg_gemm_stamps.restype = None


def run(name, M, N, K, **kw):
    g = torch.Generator(device=DEV).manual_seed(0)
    akm, bkm = kw.pop("a_kmajor", False), kw.pop("b_kmajor", False)
    A = torch.randn((K, M) if akm else (M, K), device=DEV, generator=g).to(torch.bfloat16)
    B = (torch.randn((K, N) if bkm else (N, K), device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=kw.pop("out", torch.bfloat16))
    ntiles = ((M + 127) // 128) * ((N + 127) // 128)
    st = torch.zeros(ntiles * 4, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, **kw)
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(st.data_ptr())
    ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, **kw)
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(None)
    s = st.view(ntiles, 4).double().cpu()
    d = s[:, 1:] - s[:, :-1]
    span = (s[:, 3].max() - s[:, 0].min()).item()
    print(f"{name:42s} tiles {ntiles:6d}  prologue {d[:,0].median():8.0f}  mainloop {d[:,1].median():8.0f} ({d[:,1].median()/((K+63)//64):6.0f}/iter)"
          f"  epilogue {d[:,2].median():8.0f}  wg total {(s[:,3]-s[:,0]).median():8.0f}  kernel span {span:10.0f} ticks")


if __name__ == "__main__":
    bias = torch.zeros(4096, device=DEV)
    run("KV proj 58624x1536x768 (bias)", 58624, 1536, 768, bias=bias[:1536])
    run("square 4096^3", 4096, 4096, 4096)
    run("slab dgrad 8192x768x8192 (B kmajor)", 8192, 768, 8192, b_kmajor=True, out=torch.float32)
    run("one wave of tiles 2048x4096x768", 2048, 4096, 768)
    run("tiny grid 1024x1024x768", 1024, 1024, 768)
