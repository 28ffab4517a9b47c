#!/usr/bin/env python3
"""Throughput of the captioning hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full training step of the reference's loop body (decoder.py:157-193)
on one synthetic Flickr-shaped batch that is already resident in HBM: prefix join
(img_text_concat.py:115-123) -> TransformerDecoder forward (dropout 0.1 active) ->
cross-entropy over the 32 caption rows + argmax count -> backward -> Adam with
coupled weight decay -> (N > 1) RCCL gradient all-reduce.  Workload at N=1:
BASELINE.json configs[1] (batch 256, bf16, config.py defaults); each extra GPU adds
another 256 samples (weak scaling; N=8 is configs[2]'s global batch 2048).
For N > 1 launch with torch.distributed.run, one rank per GPU.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel, timed with events on the launch stream inside the
                timed region, against the dense bf16 MFMA peak
  cpu_baseline  the reference's step as written (stock torch.nn, f32, CPU; oracle/),
                timed on this host's cores at the reference's batch size 8
  greedy        KV-cached greedy captioning of 4096 images (configs[3]), N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (datasheet; no 2:1 sparsity)
HBM_PEAK_GBS = 8000.0

# algorithmic forward FLOPs per sample at config.py defaults (2 FLOP per MAC): QKV 810.4M,
# attention 161.1M, out-proj 270.1M, MLP 1440.7M, scored-row vocabulary projection 2470.4M
FWD_FLOPS_PER_SAMPLE = 5.153e9
TRAIN_FLOPS_PER_SAMPLE = 3 * FWD_FLOPS_PER_SAMPLE - 0.8104e9   # no dX for the frozen input of block 0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=256)
    ap.add_argument("--global-batch", type=int, default=0, help="fixed global batch (strong scaling) instead of per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--n-blocks", type=int, default=1)
    ap.add_argument("--patches", type=int, default=196)
    ap.add_argument("--mask-mode", default="reference")
    ap.add_argument("--activation", default="relu")
    ap.add_argument("--all-rows", action="store_true", help="run every row through the last block (no scored-rows-only skip)")
    ap.add_argument("--spinup-steps", type=int, default=150, help="untimed steps before the warm-up (device clocks / runtime pools settle)")
    ap.add_argument("--no-fold-norm1", action="store_true", help="block 0's norm1 applied as written (A/B of folding its affine into in_proj)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-greedy", action="store_true")
    ap.add_argument("--greedy-only", action="store_true", help="skip the training benchmark (profiling of the decoding path)")
    ap.add_argument("--greedy-images", type=int, default=4096)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--no-wgrad-stream", action="store_true", help="weight-gradient products on the compute stream (A/B of the second stream)")
    ap.add_argument("--vocab-slab", type=int, default=0, help="vocabulary columns per backward slab (0: engine default)")
    return ap.parse_args()


def cpu_baseline(max_seconds=25.0):
    """decoder.py:169-190 as written, on the host CPU, batch 8 (config.py:14)."""
    from oracle.ref_cpu import AsWrittenStep
    from image_caption_amd.synthetic import make_captions
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    stepper = AsWrittenStep()
    gen = torch.Generator().manual_seed(42)
    B = 8
    x = torch.randn(229, B, 768, generator=gen)
    _, labels = make_captions(B, 32, 50260, gen)
    t0 = time.perf_counter()
    stepper.step(x, labels)                       # warm-up
    times = []
    while len(times) < 5 and time.perf_counter() - t0 < max_seconds:
        t = time.perf_counter()
        stepper.step(x, labels)
        times.append(time.perf_counter() - t)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} steps of batch {B} (S=229, d=768, V=50260, all-rows vocab projection, dropout 0.1, "
                      f"Adam wd 0.01), median {med * 1e3:.0f} ms/step, f32, torch {torch.__version__}"}


def pmc_traffic(span, args, B, whole_vocab):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot be collected inside this process); None unless this run has the configuration the
    counters were collected on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None
    on = doc.get("collected_on", {})
    same = (on.get("per_gpu_batch") == B and on.get("dtype") == args.dtype and on.get("n_blocks") == args.n_blocks
            and on.get("patches") == args.patches and whole_vocab and not args.all_rows)
    rec = doc.get("launches", {}).get(span)
    return rec["hbm_bytes_per_launch"] if same and rec else None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    rehearse = world == 1 and os.environ.get("CAPDEC_DP_SINGLE_RANK", "0") == "1"
    json_out = sys.stdout
    if world > 1 or rehearse:
        # RCCL prints a version banner on file descriptor 1 at communicator set-up: keep stdout for the one
        # JSON line, and send everything else that is written to descriptor 1 to stderr
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder
    from image_caption_amd.engine import EngineOptions, KernelTimer
    from image_caption_amd.synthetic import make_batch

    B = args.global_batch // world if args.global_batch else args.batch_per_gpu
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    opts = EngineOptions(compute_dtype=cdt, mask_mode=args.mask_mode, activation=args.activation,
                         scored_rows_only=not args.all_rows, fold_norm1=not args.no_fold_norm1)
    if args.vocab_slab:
        opts.vocab_slab = args.vocab_slab
    if args.no_wgrad_stream:
        opts.wgrad_stream = False
    torch.manual_seed(42)                                           # config.py:27,41
    model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, n_blocks=args.n_blocks, options=opts).to(dev)
    model.train()                                                   # decoder.py:144: dropout active
    trainer = CaptionTrainer(model, lr=1e-4, weight_decay=0.01, bucket_bytes=args.bucket_mb << 20)
    batch = make_batch(B, P=args.patches, L=32, d=768, vocab_size=50260, seed=42 + rank, device=dev)
    inputs = (batch["image_embeddings"], batch["sos_embedding"], batch["caption_embeddings"])
    labels = batch["caption_labels"]

    def sync():
        torch.cuda.synchronize()
        if world > 1 or rehearse:
            dist.barrier()
            torch.cuda.synchronize()

    if args.greedy_only:
        print(json.dumps({"greedy": greedy_bench(model, dev, args)}), flush=True)
        return
    # Device spin-up, untimed and before the W warm-up steps: a process's first ~0.2 s of sustained load contains one or
    # two 20-35 ms stalls (tools/warmup_probe.py shows them: clock / power-state changes or runtime housekeeping, at a
    # different step from run to run), which a short warm-up leaves inside the timed steps.  A fixed number of steps, so
    # that every rank issues the same collectives.
    for _ in range(args.spinup_steps):
        trainer.step(inputs, labels)
    sync()
    for _ in range(args.warmup):
        trainer.step(inputs, labels)
    timer = KernelTimer()
    model.engine.timer = timer
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(inputs, labels)
    sync()
    elapsed = time.perf_counter() - t0
    model.engine.timer = None
    if world > 1 or rehearse:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = out[0].item() / out[2]
    spans = timer.summary()

    if rank == 0:
        samples = B * world * args.steps
        value = samples / elapsed
        # ---- roofline of the dominant kernel: the vocabulary-sized GEMMs of the fused logits+CE stage
        Mh, d, V = 32 * B, 768, 50260
        slab = model.engine.last_slab
        flops = {"logits_ce_fwd": 2.0 * Mh * d * V,
                 "logits_ce_bwd_slab": 2.0 * Mh * d * min(slab, V),
                 "head_dgrad_slab": 2.0 * Mh * d * min(slab, V),
                 "head_wgrad_slab": 2.0 * Mh * d * min(slab, V)}
        kernel_names = {"logits_ce_fwd": "gemm_bf16_dma_kernel<false,false,1,2,2,8,3> (EPI_CE_FWD, 256x128 tiles) + ce_finalize_kernel",
                        "logits_ce_bwd_slab": "gemm_bf16_dma_kernel<false,false,2,2,2,8,3> (EPI_CE_BWD, 256x128 tiles) + colpart_reduce_kernel",
                        "head_dgrad_slab": "gemm_bf16_pipe_kernel<false,true,EPI_STD,8> (256x256 tiles, 4 waves of 128x128, split-K) + splitk_reduce_kernel",
                        "head_wgrad_slab": "gemm_bf16_pipe_kernel<true,true,EPI_STD,8> (256x256 tiles, 4 waves of 128x128)"}
        per_kernel = {}
        esz = 2 if args.dtype == "bf16" else 4
        Vp = (V + 127) // 128 * 128
        hbm_kernels = {}
        for name, (count, ms) in spans.items():
            if name == "ce_grad_from_probs" and ms > 0:
                # HBM-bound pass: numerators read and d(logits) written in place, plus the forward's per-slot partials
                nbytes = 2.0 * Mh * Vp * esz + 16.0 * Mh * 2 * ((V + 127) // 128)
                hbm_kernels[name] = {"launches_per_step": count // max(1, args.steps), "ms_per_launch": round(ms, 4),
                                     "algorithmic_gb": round(nbytes / 1e9, 3), "gb_per_s": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                     "frac_of_8tb_s": round(nbytes / (ms * 1e-3) / 8e12, 3)}
            if name in flops and ms > 0:
                f = flops[name]
                if name != "logits_ce_fwd":        # the last slab is narrower: use the mean slab width
                    nslab = (V + slab - 1) // slab
                    f = 2.0 * Mh * d * V / nslab
                per_kernel[name] = {"launches_per_step": count // max(1, args.steps), "ms_per_launch": round(ms, 4),
                                    "tflops": round(f / (ms * 1e-3) / 1e12, 1)}
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_launch"] * per_kernel[k]["launches_per_step"]) if per_kernel else None
        roofline = None
        if dom:
            a = per_kernel[dom]["tflops"]
            roofline = {"kernel": kernel_names[dom], "bound": "mfma", "achieved": a, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(a / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                        "traffic": pmc_traffic(dom, args, B, slab >= V),
                        "ms_per_launch": per_kernel[dom]["ms_per_launch"], "per_kernel": per_kernel, "hbm_bound_kernels": hbm_kernels,
                        "whole_step_algorithmic_tflops": round(value * TRAIN_FLOPS_PER_SAMPLE / 1e12, 1) if args.n_blocks == 1 and args.patches == 196 else None}
        result = {
            "metric": "train samples/sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "spinup_steps": args.spinup_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"single training step (prefix join + fwd/bwd + CE + Adam) at batch {B} per GPU, "
                                   f"S={args.patches + 33}, d=768, H=8, ff=2048, V=50260, {args.n_blocks} block(s), dropout 0.1",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "mask_mode": args.mask_mode, "activation": args.activation,
                       "last_block_scored_rows_only": not args.all_rows, "norm1_folded_into_in_proj": not args.no_fold_norm1},
            "final_loss": round(loss, 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_greedy:
            result["greedy"] = greedy_bench(model, dev, args)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), file=json_out, flush=True)
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


def greedy_bench(model, dev, args):
    """configs[3]: KV-cached greedy captioning of 4096 synthetic images, L=32, one GPU."""
    from image_caption_amd.synthetic import make_batch
    n = args.greedy_images
    model.eval()
    b = make_batch(n, P=args.patches, L=1, d=768, vocab_size=50260, seed=7, device=dev)
    emb = torch.randn(50260, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    img, sos = b["image_embeddings"], b["sos_embedding"]
    model.generate(img, sos, emb, max_len=32)                     # warm-up
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ids = model.generate(img, sos, emb, max_len=32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # roofline of the two kernels that dominate decoding, from one more (untimed) run with event brackets
    from image_caption_amd.engine import KernelTimer
    timer = KernelTimer()
    model.engine.timer = timer
    model.generate(img, sos, emb, max_len=32)
    model.engine.timer = None
    spans = timer.summary()
    esz = 2 if args.dtype == "bf16" else 4
    S0 = args.patches + 1
    roof = {}
    if "attn_decode" in spans:
        cnt, ms = spans["attn_decode"]
        steps = cnt // max(1, args.n_blocks)
        # every step reads the K and V rows written so far: lengths S0+1 .. S0+steps
        nbytes = sum(n * 768 * 2 * (S0 + 1 + t) * esz for t in range(steps)) * args.n_blocks
        roof["attn_decode_kernel"] = {"bound": "hbm", "launches": cnt, "ms_per_launch": round(ms, 4),
                                      "achieved": round(nbytes / (ms * cnt * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(nbytes / (ms * cnt * 1e-3) / 8e12, 3)}
    if "logits_argmax" in spans:
        cnt, ms = spans["logits_argmax"]
        tf = 2.0 * n * 768 * 50260 / (ms * 1e-3) / 1e12
        roof["logits_argmax (gemm_bf16_dma_kernel<false,false,1,2,2,8,3>)"] = {
            "bound": "mfma", "launches": cnt, "ms_per_launch": round(ms, 4), "achieved": round(tf, 1),
            "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    model.train()
    return {"metric": "greedy captions/sec", "value": round(n / dt, 1), "unit": "captions/s", "images": n, "max_len": 32,
            "ms_per_batch": round(dt * 1e3, 2), "kv_cache": True, "dtype": args.dtype,
            "distinct_tokens": int(ids.unique().numel()), "roofline": roof}


if __name__ == "__main__":
    main()
