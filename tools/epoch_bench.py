#!/usr/bin/env python3
"""Throughput of the training loop fed from a device-resident processed dataset (strided batch
views, decoder.py:157-166) next to the same steps on a fixed joined batch.
Usage (GPU box): python tools/epoch_bench.py [samples] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd.dataset import BatchSource  # noqa: E402
from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = "cuda"
    S, d, L, V = 229, 768, 32, 50260
    g = torch.Generator(device=dev).manual_seed(42)
    data = {"decoder_inputs": torch.randn(S, n, d, device=dev, generator=g),
            "caption_input_ids": torch.randint(0, V, (n, L), device=dev, generator=g),
            "caption_labels": torch.randint(0, V, (n, L), device=dev, generator=g)}
    torch.manual_seed(42)
    model = TransformerDecoder(768, 8, 6, 2048, 0.1, V, options=EngineOptions(compute_dtype=torch.bfloat16)).to(dev)
    model.train()
    trainer = CaptionTrainer(model, lr=1e-4, weight_decay=0.01)
    src = BatchSource(data, bsz, dev, resident=True)
    for x, y in src:                       # warm-up epoch (buffers, first-use costs)
        trainer.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x, y in src:
        trainer.step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"epoch over {n} resident samples, batch {bsz}: {n / dt:9.1f} samples/s ({dt / len(src) * 1e3:.3f} ms/step)")
    x, y = next(iter(src))
    xc = x.contiguous()
    for _ in range(3):
        trainer.step(xc, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(len(src)):
        trainer.step(xc, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"same steps on one contiguous joined batch:      {n / dt:9.1f} samples/s ({dt / len(src) * 1e3:.3f} ms/step)")


if __name__ == "__main__":
    main()
