#!/usr/bin/env python3
"""Greedy decoding time per token at serving batch sizes: one composed capdec_decode_step call per token
against the same kernels launched one by one from Python.
Usage (on the GPU box): python tools/greedy_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd.decoder import TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions  # noqa: E402
from image_caption_amd.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    emb = torch.randn(50260, 768, device=dev)
    print("images | composed: ms/caption-batch, us/token | launch by launch: ms/caption-batch, us/token")
    for n in (1, 8, 64, 512, 4096):
        row = []
        for composed in (True, False):
            opts = EngineOptions(compute_dtype=torch.bfloat16, composed_decode_step=composed)
            model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, options=opts).to(dev)
            model.eval()
            b = make_batch(n, P=196, L=1, d=768, vocab_size=50260, seed=7, device=dev)
            img, sos = b["image_embeddings"], b["sos_embedding"]
            model.generate(img, sos, emb, max_len=32)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                model.generate(img, sos, emb, max_len=32)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            row.append(f"{dt * 1e3:8.2f} {dt / 32 * 1e6:8.1f}")
            del model
        print(f"{n:6d} | {row[0]} | {row[1]}")


if __name__ == "__main__":
    main()
