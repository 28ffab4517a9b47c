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
    st = torch.zeros(ntiles * 8, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, **kw)
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(st.data_ptr())
    ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, **kw)
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(None)
    s = st.view(ntiles, 8).double().cpu()
    s = s[s[:, 0] > 0]                      # taller tiles launch fewer workgroups than the 128-row count
    ntiles = s.shape[0]
    for c in range(3, 7):                   # epilogue sub-stamps a kernel does not write: carry the previous one
        s[:, c] = torch.where(s[:, c] > 0, s[:, c], s[:, c - 1])
    sub = (s[:, 3:8] - s[:, 2:7]).median(0).values.tolist()
    s = torch.cat([s[:, :3], s[:, 7:8]], 1)
    d = s[:, 1:] - s[:, :-1]
    span = (s[:, 3].max() - s[:, 0].min()).item()
    name = f"{name} [epilogue: stage {sub[0]:.0f} rows {sub[1]:.0f} stage {sub[2]:.0f} rows {sub[3]:.0f}+{sub[4]:.0f}]"
    print(f"{name}\n    tiles {ntiles:6d}  prologue {d[:,0].median():8.0f}  mainloop {d[:,1].median():8.0f} ({d[:,1].median()/((K+63)//64):6.0f}/iter)"
          f"  epilogue {d[:,2].median():8.0f}  wg total {(s[:,3]-s[:,0]).median():8.0f}  kernel span {span:10.0f} ticks")


def stamp_call(name, fn, ntiles, iters):
    """Stamps of any library call that launches exactly one GEMM-class kernel of `ntiles` workgroups."""
    st = torch.zeros(ntiles * 8, dtype=torch.int64, device=DEV)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(st.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.capdecdbg_gemm_stamps(None)
    s = st.view(ntiles, 8).double().cpu()
    s = s[s[:, 0] > 0]                      # taller tiles launch fewer workgroups than the 128-row count
    ntiles = s.shape[0]
    for c in range(3, 7):                   # epilogue sub-stamps a kernel does not write: carry the previous one
        s[:, c] = torch.where(s[:, c] > 0, s[:, c], s[:, c - 1])
    sub = (s[:, 3:8] - s[:, 2:7]).median(0).values.tolist()
    s = torch.cat([s[:, :3], s[:, 7:8]], 1)
    d = s[:, 1:] - s[:, :-1]
    span = (s[:, 3].max() - s[:, 0].min()).item()
    name = f"{name} [epilogue: stage {sub[0]:.0f} rows {sub[1]:.0f} stage {sub[2]:.0f} rows {sub[3]:.0f}+{sub[4]:.0f}]"
    print(f"{name}\n    tiles {ntiles:6d}  prologue {d[:,0].median():8.0f}  mainloop {d[:,1].median():8.0f} ({d[:,1].median()/iters:6.0f}/iter)"
          f"  epilogue {d[:,2].median():8.0f}  wg total {(s[:,3]-s[:,0]).median():8.0f}  kernel span {span:10.0f} ticks")


def vocabulary_cases():
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
    stamp_call("logits+CE fwd 8192x50260x768", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws),
               (Mt // 128) * ((V + 127) // 128), d // 32)
    probs = torch.empty(Mt, (V + 127) // 128 * 128, device=DEV, dtype=torch.bfloat16)
    stamp_call("logits+CE fwd keeping the softmax numerators", lambda: ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs),
               (Mt // 128) * ((V + 127) // 128), d // 32)
    slab = 8192
    dl = torch.empty(Mt, slab, device=DEV, dtype=torch.bfloat16)
    dbias = torch.zeros(V, device=DEV)
    bws = torch.empty(ops.logits_ce_bwd_workspace(Mt, slab), dtype=torch.uint8, device=DEV)
    stamp_call("CE bwd slab 8192x8192x768", lambda: ops.logits_ce_bwd_slab(x, w, bias, labels, B, L, lse, 1.0 / Mt, 0, slab, dl, dbias, bws),
               (Mt // 128) * (slab // 128), d // 32)
    Vp = (V + 127) // 128 * 128
    dlw = torch.zeros(Mt, Vp, device=DEV, dtype=torch.bfloat16)
    dlw[:, :V] = (torch.randn(Mt, V, device=DEV, generator=g) * 1e-3).to(torch.bfloat16)
    dWw = torch.empty(V, d, device=DEV)
    stamp_call("head wgrad 50260x768x8192 (A,B kmajor)", lambda: ops.gemm(dlw, x, dWw, M=V, N=d, K=Mt, a_kmajor=True, b_kmajor=True),
               ((V + 127) // 128) * 6, Mt // 32)
    wp = torch.zeros(Vp, d, device=DEV, dtype=torch.bfloat16)
    wp[:V] = w
    dHh = torch.empty(Mt, d, device=DEV)
    stamp_call("head dgrad 8192x768x50304 (B kmajor, no split)", lambda: ops.gemm(dlw, wp, dHh, M=Mt, N=d, K=Vp, b_kmajor=True),
               (Mt // 128) * 6, Vp // 32)
    dW = torch.empty(slab, d, device=DEV)
    stamp_call("slab wgrad 8192x768x8192 (A,B kmajor)", lambda: ops.gemm(dl, x, dW, a_kmajor=True, b_kmajor=True), (slab // 128) * 6, slab // 64)


def deep_k_cases():
    """The two 256x256 kernels (forced through the diagnostic kernel switch) on main-loop-dominated products."""
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    for mode, label in ((5, "8 waves of 128x64"), (6, "4 waves of 128x128")):
        lib.capdecdbg_gemm_mode(mode)
        run(f"square 8192^3, {label}", 8192, 8192, 8192)
        run(f"wgrad-shaped 50304x768x8192 (A,B kmajor), {label}", 50304, 768, 8192, a_kmajor=True, b_kmajor=True, out=torch.float32)
    lib.capdecdbg_gemm_mode(0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deep":
        deep_k_cases()
        sys.exit(0)
    vocabulary_cases()
    bias = torch.zeros(4096, device=DEV)
    run("KV proj 58624x1536x768 (bias)", 58624, 1536, 768, bias=bias[:1536])
    run("square 4096^3", 4096, 4096, 4096)
    run("slab dgrad 8192x768x8192 (B kmajor)", 8192, 768, 8192, b_kmajor=True, out=torch.float32)
    run("one wave of tiles 2048x4096x768", 2048, 4096, 768)
    run("tiny grid 1024x1024x768", 1024, 1024, 768)
