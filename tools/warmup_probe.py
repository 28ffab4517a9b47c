#!/usr/bin/env python3
"""How long a fresh process takes to reach its steady step time: the step time of consecutive groups of 5 training
steps at the bench configuration, from the first step on.  Usage (on the GPU box): python tools/warmup_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions  # noqa: E402
from image_caption_amd.synthetic import make_batch  # noqa: E402

dev = "cuda:0"
torch.manual_seed(42)
model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, options=EngineOptions()).to(dev)
model.train()
trainer = CaptionTrainer(model, lr=1e-4, weight_decay=0.01)
batch = make_batch(256, P=196, L=32, d=768, vocab_size=50260, seed=42, device=dev)
inputs = (batch["image_embeddings"], batch["sos_embedding"], batch["caption_embeddings"])
labels = batch["caption_labels"]
t_start = time.perf_counter()
out = []
for grp in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        trainer.step(inputs, labels)
    torch.cuda.synchronize()
    out.append(((time.perf_counter() - t0) / 5 * 1e3, time.perf_counter() - t_start))
print(" ".join(f"{ms:.2f}" for ms, _ in out))
print("seconds since the first step, per group:", " ".join(f"{t:.2f}" for _, t in out))
