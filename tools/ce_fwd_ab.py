#!/usr/bin/env python3
"""A/B of the kernels that can run the fused logits + cross-entropy forward at the benchmark's size
(8192 x 50260 x 768), interleaved in one process; each variant is also checked against the default one.
Usage (on the GPU box): python tools/ce_fwd_ab.py [modes...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"
lib = _lib.load()
lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
lib.capdecdbg_gemm_mode.restype = None


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


def main():
    modes = [int(a) for a in sys.argv[1:]] or [0, 4, 5, 7]
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
    big = torch.empty(1 << 28, device=DEV)          # 1 GiB written between launches: every launch starts from cold caches

    def run(keep, cold):
        if cold == 1:
            big.fill_(1.0)
        elif cold == 2:                             # caches turned over by reads: nothing dirty left behind
            torch.sum(big[: 1 << 27], dtype=torch.float32)
        ops.logits_ce_fwd(x, w, bias, labels, B, L, lse, amax, ls, cr, ws, probs=probs if keep else None)

    ref = {}
    for m in modes:
        lib.capdecdbg_gemm_mode(m)
        probs.fill_(float("nan"))
        run(True, False)
        torch.cuda.synchronize()
        out = (lse.clone(), amax.clone(), ls.item(), probs[:, :V].float().sum().item())
        if not ref:
            ref = out
        else:
            ok = torch.allclose(out[0], ref[0], rtol=0, atol=2e-5) and abs(out[2] - ref[2]) < 1e-5 * abs(ref[2]) and \
                abs(out[3] - ref[3]) < 1e-5 * abs(ref[3]) and (out[1] != ref[1]).float().mean().item() < 1e-3
            print(f"mode {m}: agrees with mode {modes[0]}: {ok}  (lse max diff {(out[0] - ref[0]).abs().max().item():.2e})")
    lib.capdecdbg_gemm_mode(0)
    tfill = timeit(lambda: big.fill_(1.0), 5)
    fl = 2.0 * Mt * V * d
    tsum = timeit(lambda: torch.sum(big[: 1 << 27], dtype=torch.float32), 5)
    print("mode | no numerators: warm us (TFLOP/s) | numerators: warm us (TFLOP/s) | numerators, cold caches us | "
          "no numerators, cold | numerators, caches turned over by reads | no numerators, same")
    res = {m: [[], [], [], [], [], []] for m in modes}
    for _ in range(5):
        for m in modes:
            lib.capdecdbg_gemm_mode(m)
            res[m][0].append(timeit(lambda: run(False, False)))
            res[m][1].append(timeit(lambda: run(True, False)))
            res[m][2].append(timeit(lambda: run(True, 1), 5) - tfill)
            res[m][3].append(timeit(lambda: run(False, 1), 5) - tfill)
            res[m][4].append(timeit(lambda: run(True, 2), 5) - tsum)
            res[m][5].append(timeit(lambda: run(False, 2), 5) - tsum)
    lib.capdecdbg_gemm_mode(0)
    for m in modes:
        a, b, c, d2, e2, f2 = (sorted(v)[2] for v in res[m])
        print(f"{m:4d} | {a * 1e3:8.1f} ({fl / a / 1e9:6.1f}) | {b * 1e3:8.1f} ({fl / b / 1e9:6.1f}) | {c * 1e3:8.1f} ({fl / c / 1e9:6.1f})"
              f" | {d2 * 1e3:8.1f} | {e2 * 1e3:8.1f} | {f2 * 1e3:8.1f}")


if __name__ == "__main__":
    main()
