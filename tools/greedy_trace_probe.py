#!/usr/bin/env python3
"""One greedy caption batch of N images, for a kernel trace: python tools/greedy_trace_probe.py N"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_caption_amd.decoder import TransformerDecoder  # noqa: E402
from image_caption_amd.engine import EngineOptions  # noqa: E402
from image_caption_amd.synthetic import make_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
emb = torch.randn(50260, 768, device=dev)
model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, options=EngineOptions(compute_dtype=torch.bfloat16)).to(dev)
model.eval()
b = make_batch(n, P=196, L=1, d=768, vocab_size=50260, seed=7, device=dev)
for _ in range(3):
    model.generate(b["image_embeddings"], b["sos_embedding"], emb, max_len=32)
torch.cuda.synchronize()
