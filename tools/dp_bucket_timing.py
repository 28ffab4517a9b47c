#!/usr/bin/env python3
"""Single-rank rehearsal of the data-parallel step (CAPDEC_DP_SINGLE_RANK=1: RCCL communicator of one rank, every
all-reduce an identity copy issued exactly as it would be with eight): for each gradient bucket, how long before the end
of the backward it was handed to RCCL -- the window its all-reduce has to hide in.  Batch 256, config.py defaults.

    CAPDEC_DP_SINGLE_RANK=1 python tools/dp_bucket_timing.py [--compress bf16] [--bucket-mb 64]
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compress", default=None)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    os.environ["CAPDEC_DP_SINGLE_RANK"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)                                   # RCCL's banner goes to descriptor 1
    from image_caption_amd.dp import init_process_group as dp_init
    dp_init("nccl", dev, rank=0, world_size=1)
    from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder
    from image_caption_amd.engine import EngineOptions
    from image_caption_amd.synthetic import make_batch
    torch.manual_seed(42)
    model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, options=EngineOptions()).to(dev)
    model.train()
    tr = CaptionTrainer(model, lr=1e-4, weight_decay=0.01, bucket_bytes=args.bucket_mb << 20, grad_compress=args.compress,
                        record_bucket_timing=True)
    b = make_batch(args.batch, P=196, L=32, d=768, vocab_size=50260, seed=42, device=dev)
    inputs, labels = (b["image_embeddings"], b["sos_embedding"], b["caption_embeddings"]), b["caption_labels"]
    for _ in range(60):
        tr.step(inputs, labels, global_batch=args.batch)
    torch.cuda.synchronize()
    leads = []
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        tr.step(inputs, labels, global_batch=args.batch)
        leads.append(tr.reducer.issue_lead_ms())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"# single-rank rehearsal, batch {args.batch}, bucket threshold {args.bucket_mb} MB, compress={args.compress}: "
          f"{ms:.3f} ms per step (with one host synchronisation per step to read the events), payload {tr.reducer.payload_bytes() / 1e6:.1f} MB per step",
          file=out)
    print("bucket,first_element,last_element,payload_MB,issued_ms_before_end_of_backward(mean),min,max", file=out)
    nb = len(leads[0])
    esz = 2 if args.compress == "bf16" else 4
    for i in range(nb):
        v = [l[i][2] for l in leads]
        a, e = leads[0][i][0], leads[0][i][1]
        print(f"{i},{a},{e},{(e - a) * esz / 1e6:.1f},{sum(v) / len(v):.3f},{min(v):.3f},{max(v):.3f}", file=out)
    out.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
