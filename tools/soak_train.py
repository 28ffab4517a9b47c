#!/usr/bin/env python3
"""Soak test: many training steps at the benchmark configuration, looking for a non-finite loss (a rare race
would show as one).  Usage (on the GPU box): python tools/soak_train.py [steps] [gemm mode]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd import _lib  # noqa: E402
from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions, KernelTimer  # noqa: E402
from image_caption_amd.synthetic import make_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
dev = torch.device("cuda", 0)
lib.capdecdbg_gemm_mode(mode)
restarts = 0
bad_total = 0
done = 0
while done < steps:
    torch.manual_seed(42 + restarts)
    model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, options=EngineOptions()).to(dev)
    model.train()
    tr = CaptionTrainer(model, lr=1e-4, weight_decay=0.01)
    b = make_batch(256, P=196, L=32, d=768, vocab_size=50260, seed=42 + restarts, device=dev)
    inputs = (b["image_embeddings"], b["sos_embedding"], b["caption_embeddings"])
    labels = b["caption_labels"]
    model.engine.timer = KernelTimer() if restarts % 2 else None      # with and without event brackets between launches
    flags = []
    n = min(250, steps - done)
    for i in range(n):
        out = tr.step(inputs, labels)
        flags.append(~torch.isfinite(out[0]))
    model.engine.timer = None
    f = torch.cat(flags)
    nb = int(f.sum().item())
    if nb:
        print(f"run {restarts}: first non-finite loss at step {int(f.nonzero()[0])} of {n}; parameters finite: {bool(torch.isfinite(tr.store.w).all())}", flush=True)
    bad_total += 1 if nb else 0
    done += n
    restarts += 1
    del model, tr
print(f"mode {mode} geometry {os.environ.get('CAPDEC_HEAD_FWD_GEOMETRY', '0')}: {done} steps in {restarts} runs, {bad_total} runs went non-finite")
