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
    # the 12-block configuration's plain products (M = 74240 rows): data gradients dY W and weight gradients dY^T X
    Mb, ff = 74240, 2048
    dy_d, dy_ff, dy_3d = r(Mb, d), r(Mb, ff), r(Mb, 3 * d)
    x_d, x_ff = r(Mb, d), r(Mb, ff)
    w_in, w1, w2, wo = r(3 * d, d) * 0.03, r(ff, d) * 0.03, r(d, ff) * 0.03, r(d, d) * 0.03
    o_d = torch.empty(Mb, d, device=DEV, dtype=BF)
    o_ff = torch.empty(Mb, ff, device=DEV, dtype=BF)
    g_in, g1, g2, go = (torch.empty(*t.shape, device=DEV, dtype=torch.float32) for t in (w_in, w1, w2, wo))
    cases += [
        ("dgrad in_proj   dQKV @ W_in   74240x768x2304", 2.0 * Mb * d * 3 * d, lambda: torch.matmul(dy_3d, w_in, out=o_d)),
        ("dgrad linear1   dZ1 @ W1      74240x768x2048", 2.0 * Mb * d * ff, lambda: torch.matmul(dy_ff, w1, out=o_d)),
        ("dgrad linear2   G2 @ W2       74240x2048x768 (no gate)", 2.0 * Mb * d * ff, lambda: torch.matmul(dy_d, w2, out=o_ff)),
        ("dgrad out_proj  G1 @ Wo       74240x768x768", 2.0 * Mb * d * d, lambda: torch.matmul(dy_d, wo, out=o_d)),
        ("fwd   in_proj   X @ W_in^T    74240x2304x768 (no bias)", 2.0 * Mb * d * 3 * d, lambda: torch.matmul(x_d, w_in.t())),
        ("wgrad in_proj   dQKV^T @ X    2304x768x74240 (bf16 out)", 2.0 * Mb * d * 3 * d, lambda: torch.matmul(dy_3d.t(), x_d)),
        ("wgrad linear1   dZ1^T @ X     2048x768x74240 (bf16 out)", 2.0 * Mb * d * ff, lambda: torch.matmul(dy_ff.t(), x_d)),
        ("wgrad linear2   G2^T @ A1     768x2048x74240 (bf16 out)", 2.0 * Mb * d * ff, lambda: torch.matmul(dy_d.t(), x_ff)),
        ("wgrad out_proj  G1^T @ O      768x768x74240 (bf16 out)", 2.0 * Mb * d * d, lambda: torch.matmul(dy_d.t(), x_d)),
    ]
    for name, fl, fn in cases:
        ms = timeit(fn)
        print(f"{name:50s} {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
