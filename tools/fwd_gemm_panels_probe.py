#!/usr/bin/env python3
"""What k-panel images of the operands would give the large row-major forward products of the blocks (the persistent
kernel's standard epilogue): the images are handed in through the diagnostic override of head_fwd.hip, results compared
bit for bit.  Usage (on the GPU box): python tools/fwd_gemm_panels_probe.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops
DEV, BF = "cuda", torch.bfloat16
lib = _lib.load()
lib.capdecdbg_head_fwd_packed.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
lib.capdecdbg_head_fwd_packed.restype = None
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
def pack(x):
    r, k = x.shape
    return x.view(r, k // 32, 32).permute(1, 0, 2).contiguous()
g = torch.Generator(device=DEV).manual_seed(0)
big = torch.empty(1 << 28, device=DEV)
tfill = timeit(lambda: big.fill_(1.0), 5)
for name, M, N, K in (("KV projection", 58624, 1536, 768), ("in_proj, 12-block shape", 74240, 2304, 768), ("linear1, 12-block shape", 74240, 2048, 768)):
    x = torch.randn(M, K, device=DEV, generator=g).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.03).to(BF)
    b = torch.randn(N, device=DEV, generator=g)
    y = torch.empty(M, N, device=DEV, dtype=BF)
    xp, wp = pack(x), pack(w)
    ref = None
    for vn, v in (("row-major", (0, 0, 0, 0)), ("B image", (0, 0, wp.data_ptr(), N)), ("A and B images", (xp.data_ptr(), M, wp.data_ptr(), N))):
        lib.capdecdbg_head_fwd_packed(*v)
        fn = lambda: ops.gemm(x, w, y, bias=b)
        fn(); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        same = torch.equal(ref, y)
        def cold():
            big.fill_(1.0); fn()
        tw, tc = timeit(fn), timeit(cold, 5) - tfill
        print(f"{name:26s} {vn:16s}: warm {tw:7.1f} us ({2.0*M*N*K/tw/1e6:6.1f} TFLOP/s), cold {tc:7.1f} us, same bits {same}")
    lib.capdecdbg_head_fwd_packed(0, 0, 0, 0)
