#!/usr/bin/env python3
"""A/B of the persistent logits + cross-entropy forward with its operands read row-major against the same launch
reading k-panel images ([K/32][rows][32]) of them, at the benchmark's size (8192 x 50260 x 768); results compared.
Usage (on the GPU box): python tools/ce_fwd_packed_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"
lib = _lib.load()
lib.capdecdbg_head_fwd_packed.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
lib.capdecdbg_head_fwd_packed.restype = None


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def pack(x):
    r, k = x.shape
    return x.view(r, k // 32, 32).permute(1, 0, 2).contiguous()


def main():
    B, L, d, V = 256, 32, 768, 50260
    Mt = B * L
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(Mt, d, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(V, d, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn(V, device=DEV, generator=g) * 0.03
    labels = torch.randint(0, V, (B, L), device=DEV, generator=g)
    ws = torch.empty(ops.logits_ce_workspace(Mt, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(Mt, device=DEV), torch.empty(Mt, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    probs = torch.empty(Mt, (V + 127) // 128 * 128, device=DEV, dtype=torch.bfloat16)
    rowref = torch.empty(Mt, device=DEV)
    big = torch.empty(1 << 28, device=DEV)
    xp, wp = pack(x), pack(w)
    variants = {"row-major": (0, 0, 0, 0), "A packed": (xp.data_ptr(), Mt, 0, 0), "B packed": (0, 0, wp.data_ptr(), V),
                "both packed": (xp.data_ptr(), Mt, wp.data_ptr(), V)}

    def run(kind, cold):
        if cold:
            big.fill_(1.0)
        if kind == 0:
            ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=None)
        elif kind == 1:
            ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs)
        else:
            ops.logits_ce_fwd_ref(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs, rowref)

    ref = None
    for name, v in variants.items():
        lib.capdecdbg_head_fwd_packed(*v)
        probs.fill_(float("nan"))
        run(2, False)
        torch.cuda.synchronize()
        out = (lse.clone(), amax.clone(), ls.item(), probs[:, :V].clone())
        if ref is None:
            ref = out
        else:
            print(f"{name}: lse equal {torch.equal(out[0], ref[0])}, argmax equal {torch.equal(out[1], ref[1])}, "
                  f"loss equal {out[2] == ref[2]}, numerators equal {torch.equal(out[3], ref[3])}")
    tfill = timeit(lambda: big.fill_(1.0), 5)
    fl = 2.0 * Mt * V * d
    print("variant | argmax only warm us | numerators warm | row-reference entry warm | row-reference entry, cold caches")
    res = {n: [[], [], [], []] for n in variants}
    for _ in range(5):
        for name, v in variants.items():
            lib.capdecdbg_head_fwd_packed(*v)
            res[name][0].append(timeit(lambda: run(0, False)))
            res[name][1].append(timeit(lambda: run(1, False)))
            res[name][2].append(timeit(lambda: run(2, False)))
            res[name][3].append(timeit(lambda: run(2, True), 5) - tfill)
    lib.capdecdbg_head_fwd_packed(0, 0, 0, 0)
    for name in variants:
        a, b, c, d2 = (sorted(v)[2] for v in res[name])
        print(f"{name:12s} | {a * 1e3:8.1f} ({fl / a / 1e9:6.1f}) | {b * 1e3:8.1f} ({fl / b / 1e9:6.1f}) | {c * 1e3:8.1f} | {d2 * 1e3:8.1f}")


if __name__ == "__main__":
    main()
