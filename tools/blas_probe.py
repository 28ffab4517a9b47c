#!/usr/bin/env python3
"""What the vendor library (torch.matmul -> hipBLASLt) reaches on the vocabulary-stage products, as a yardstick for
the hand-written kernels.  Diagnostic only: nothing in the package calls torch.matmul.
Usage (on the GPU box): python tools/blas_probe.py"""
import torch

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, reps=20):
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
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).to(BF)     # noqa: E731
    Mt, d, V = 8192, 768, 50304
    h, w, dl = r(Mt, d), r(V, d) * 0.03, r(Mt, V) * 1e-3
    out_l = torch.empty(Mt, V, device=DEV, dtype=BF)
    out_w = torch.empty(V, d, device=DEV, dtype=BF)
    out_h = torch.empty(Mt, d, device=DEV, dtype=BF)
    a8, b8 = r(8192, 8192), r(8192, 8192)
    c8 = torch.empty(8192, 8192, device=DEV, dtype=BF)
    cases = [
        ("logits   h @ W^T     8192x50304x768 (bf16 out)", 2.0 * Mt * V * d, lambda: torch.matmul(h, w.t(), out=out_l)),
        ("wgrad    dl^T @ h    50304x768x8192", 2.0 * Mt * V * d, lambda: torch.matmul(dl.t(), h, out=out_w)),
        ("dgrad    dl @ W      8192x768x50304", 2.0 * Mt * V * d, lambda: torch.matmul(dl, w, out=out_h)),
        ("square NT            8192^3", 2.0 * 8192 ** 3, lambda: torch.matmul(a8, b8.t(), out=c8)),
    ]
    for name, fl, fn in cases:
        ms = timeit(fn)
        print(f"{name:50s} {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
