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
                timed on this host's cores at the reference's batch size 8 (and once at 64)
  global_2048   configs[2] at THIS N: the same step at a fixed global batch of 2048
                (2048 / N samples per GPU), so a 1/2/4/8 sweep also yields the strong-scaling curve
  greedy        KV-cached greedy captioning of 4096 images (configs[3]), N=1 only, with the
                full-recompute loop around the CPU oracle timed beside it
  scaled        configs[4] (12 blocks, 257 patches, prefix-causal mask), N=1 only
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
# the scaled configuration (12 blocks, S=290, prefix-causal attention counted as full S x S): forward 43.886 GF
SCALED_TRAIN_FLOPS_PER_SAMPLE = 3 * 43.886e9 - 1.0263e9


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
    ap.add_argument("--no-ce-row-reference", action="store_true",
                    help="softmax numerators kept per 64-column slot and turned into d(logits) by a pass of its own (A/B)")
    ap.add_argument("--no-ce-bias-in-wgrad", action="store_true",
                    help="vocabulary bias gradient by a pass over the numerators instead of as a column of the weight-gradient product (A/B)")
    ap.add_argument("--dgrad-transposed-weights", action="store_true",
                    help="data gradients of the vocabulary matrix, linear2 and out_proj on transposed weight copies (row-major second operand; A/B)")
    ap.add_argument("--no-head-k-panels", action="store_true",
                    help="the vocabulary product reads its operands row-major (A/B of the k-panel images)")
    ap.add_argument("--no-dl-panels", action="store_true",
                    help="the kept softmax numerators / d(logits) stay a row-major matrix (A/B of the column-panel image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spans", action="store_true", help="no event brackets inside the timed steps (A/B of their cost)")
    ap.add_argument("--stage-steps", type=int, default=5,
                    help="extra untimed steps with an event bracket around every stage, after the timed ones (0: none, so that a "
                         "kernel trace of the run ends with the timed steps)")
    ap.add_argument("--no-yardstick", action="store_true", help="skip the vendor-library product timed beside the dominant kernel")
    ap.add_argument("--no-greedy", action="store_true")
    ap.add_argument("--no-scaled", action="store_true", help="skip the 12-block / 257-patch leg")
    ap.add_argument("--no-global-2048", action="store_true", help="skip the fixed-global-batch-2048 leg")
    ap.add_argument("--greedy-only", action="store_true", help="skip the training benchmark (profiling of the decoding path)")
    ap.add_argument("--scaled-only", action="store_true", help="only the 12-block / 257-patch leg (profiling)")
    ap.add_argument("--greedy-images", type=int, default=4096)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--no-ppipe", action="store_true", help="large data-gradient products on the tile kernels (A/B of the persistent pipelined kernel)")
    ap.add_argument("--no-fused-attn-bwd", action="store_true",
                    help="attention backward of the scored rows as a dQ kernel and a dK/dV kernel (A/B of the one-kernel form)")
    ap.add_argument("--no-half-tiles", action="store_true",
                    help="the persistent forward kernel walks whole 256x256 tiles only (A/B of the half tiles of its last round)")
    ap.add_argument("--no-wgrad-stream", action="store_true", help="weight-gradient products on the compute stream (A/B of the second stream)")
    ap.add_argument("--vocab-slab", type=int, default=0, help="vocabulary columns per backward slab (0: engine default)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ the host the CPU legs run on
def host_cpu():
    """(model name, physical cores of the machine, threads this job may use).  The job's share of the machine can
    be much smaller than the machine: the affinity mask and OMP_NUM_THREADS bound it."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and int(omp) > 0:
        share = min(share, int(omp))
    return model, (len(cores) or (os.cpu_count() or 1)), share


def cpu_baseline(max_seconds=25.0):
    """decoder.py:169-190 as written, on the host CPU, batch 8 (config.py:14); one more sample at batch 64."""
    from oracle.ref_cpu import AsWrittenStep
    from image_caption_amd.synthetic import make_captions
    model, phys, threads = host_cpu()
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    stepper = AsWrittenStep()
    gen = torch.Generator().manual_seed(42)

    def run(B, nsteps, budget):
        x = torch.randn(229, B, 768, generator=gen)
        _, labels = make_captions(B, 32, 50260, gen)
        t0 = time.perf_counter()
        stepper.step(x, labels)                       # warm-up
        times = []
        while len(times) < nsteps and time.perf_counter() - t0 < budget:
            t = time.perf_counter()
            stepper.step(x, labels)
            times.append(time.perf_counter() - t)
        times.sort()
        return times[len(times) // 2], len(times)

    med, n = run(8, 5, max_seconds)
    out = {"value": round(8 / med, 3), "unit": "samples/s", "cores": threads, "kind": "port",
           "cpu_model": model, "physical_cores_of_host": phys,
           "sample": f"{n} steps of batch 8 (S=229, d=768, V=50260, all-rows vocab projection, dropout 0.1, "
                     f"Adam wd 0.01), median {med * 1e3:.0f} ms/step, f32, torch {torch.__version__}, {threads} threads "
                     f"(this job's CPU share of a {phys}-core host)"}
    try:
        med64, n64 = run(64, 2, 20.0)                 # 3 GB of logits + as much gradient: the as-written step at a larger batch
        out["batch_64"] = {"value": round(64 / med64, 3), "unit": "samples/s", "sample": f"{n64} step(s), median {med64 * 1e3:.0f} ms/step"}
    except (RuntimeError, MemoryError) as e:          # host memory
        out["batch_64"] = {"value": None, "error": str(e)[:120]}
    return out


def cpu_baseline_greedy():
    """Greedy captioning with the reference's own means: it has no generation code, so each new token is one more call
    of its forward as written (decoder.py:40-68: every row through the block AND through the vocabulary projection),
    of which the last row is read.  Batch 8, 32 tokens, f32, host CPU."""
    from oracle.ref_cpu import AsWrittenStep
    model, phys, threads = host_cpu()
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(7)
    B, P, d, V, L = 8, 196, 768, 50260, 32
    torch.manual_seed(42)
    net = AsWrittenStep().model                       # stock initialisation
    net.eval()
    with torch.no_grad():
        net.output_projection.weight.mul_(6.0)        # as in greedy_bench
    img, sos = torch.randn(B, P, d, generator=gen), torch.randn(B, 1, d, generator=gen)
    emb = torch.randn(V, d, generator=gen)
    cur = torch.cat([img, sos, emb[V - 3].expand(B, 1, d)], dim=1)
    ids = torch.full((B, L), V - 1, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in range(L):
            tok = net(cur.transpose(0, 1))[-1].argmax(-1)
            tok = torch.where(done, torch.full_like(tok, V - 1), tok)
            ids[:, t] = tok
            done |= tok == V - 2
            if t + 1 < L:
                cur = torch.cat([cur, emb[tok][:, None, :]], dim=1)
    dt = time.perf_counter() - t0
    return {"value": round(B / dt, 3), "unit": "captions/s", "cores": threads, "kind": "port", "cpu_model": model,
            "sample": f"1 batch of {B} images, {L} tokens each, one as-written forward (all rows, all-rows vocabulary projection) "
                      f"per token, {dt:.1f} s, f32, {threads} threads; {int(ids.unique().numel())} distinct tokens"}


def pmc_traffic(span, args, B, whole_vocab):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot be collected inside this process); None unless this run has the configuration the
    counters were collected on."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        on = doc.get("collected_on", {})
        same = (on.get("per_gpu_batch") == B and on.get("dtype") == args.dtype and on.get("n_blocks") == args.n_blocks
                and on.get("patches") == args.patches and whole_vocab and not args.all_rows)
        rec = doc.get("launches", {}).get("logits_ce_fwd" if span == "logits_ce_fwd_kernel" else span)     # the counters are per kernel
        return rec["hbm_bytes_per_launch"] if same and rec else None
    return None


HEAD_SPANS = ("logits_ce_fwd", "ce_grad_from_probs", "ce_bwd_prepare", "ce_bias_grad", "logits_ce_bwd_slab", "head_wgrad_slab", "head_dgrad_slab")
KERNEL_NAMES = {"logits_ce_fwd_kernel": "head_fwd_kernel<2,4,8,4,true,false,3> (persistent fused logits+CE forward, 256x256 tiles, numerators against the label's logit; bracketed alone)",
                "logits_ce_fwd": "capdec_logits_ce_fwd_ref as a whole: label_logit_kernel + head_fwd_kernel + ce_finalize_kernel + ce_ref_fixup_kernel + ce_reduce_kernel",
                "logits_ce_bwd_slab": "gemm_bf16_dma_kernel<false,false,2,2,2,8,3> (EPI_CE_BWD, 256x128 tiles) + colpart_reduce_kernel",
                "head_dgrad_slab": "gemm_bf16_pipe_kernel<false,true,EPI_STD,8> (256x256 tiles, 4 waves of 128x128, split-K) + splitk_reduce_kernel",
                "head_wgrad_slab": "gemm_bf16_pipe_kernel<true,true,EPI_STD,8,KVEC> (256x256 tiles, 4 waves of 128x128; the bias gradient rides along as one more column)"}


def head_kernel_timing(on):
    """HIP event brackets the library puts around the logits+CE forward kernel alone (the engine's bracket of that
    stage also covers the finalize and loss-sum launches that follow it).  on=False: (launches, mean ms) collected."""
    import ctypes
    from image_caption_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, "capdecdbg_head_fwd_timing"):
        return None
    lib.capdecdbg_head_fwd_timing.argtypes = [ctypes.c_int]
    lib.capdecdbg_head_fwd_timing.restype = None
    lib.capdecdbg_head_fwd_timing_read.argtypes = [ctypes.POINTER(ctypes.c_double)]
    lib.capdecdbg_head_fwd_timing_read.restype = ctypes.c_int
    if on:
        lib.capdecdbg_head_fwd_timing(1)
        return None
    total = ctypes.c_double(0.0)
    n = lib.capdecdbg_head_fwd_timing_read(ctypes.byref(total))
    lib.capdecdbg_head_fwd_timing(0)
    return (n, total.value / n) if n > 0 else None


def head_roofline(spans, steps, B, slab, dtype):
    """Per-launch rates of the vocabulary-sized launches from their event brackets."""
    Mh, d, V = 32 * B, 768, 50260
    per_kernel, hbm_kernels = {}, {}
    esz = 2 if dtype == "bf16" else 4
    Vp = (V + 127) // 128 * 128
    nslab = (V + slab - 1) // slab
    for name, (count, ms) in spans.items():
        if ms <= 0:
            continue
        if name == "ce_bias_grad":
            nbytes = 1.0 * Mh * V * esz
            hbm_kernels[name] = {"launches_per_step": count // max(1, steps), "ms_per_launch": round(ms, 4),
                                 "algorithmic_gb": round(nbytes / 1e9, 3), "gb_per_s": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8tb_s": round(nbytes / (ms * 1e-3) / 8e12, 3)}
        if name == "ce_grad_from_probs":
            # HBM-bound pass: numerators read and d(logits) written in place, plus the forward's per-slot partials
            nbytes = 2.0 * Mh * Vp * esz + 16.0 * Mh * 2 * ((V + 127) // 128)
            hbm_kernels[name] = {"launches_per_step": count // max(1, steps), "ms_per_launch": round(ms, 4),
                                 "algorithmic_gb": round(nbytes / 1e9, 3), "gb_per_s": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8tb_s": round(nbytes / (ms * 1e-3) / 8e12, 3)}
        if name in KERNEL_NAMES:
            f = 2.0 * Mh * d * V / (1 if name.startswith("logits_ce_fwd") else nslab)      # the last slab is narrower: mean slab width
            per_kernel[name] = {"launches_per_step": count // max(1, steps), "ms_per_launch": round(ms, 4),
                                "tflops": round(f / (ms * 1e-3) / 1e12, 1)}
    return per_kernel, hbm_kernels


def library_yardstick(dev, B):
    """What the vendor library reaches on the dominant kernel's product alone (torch.matmul -> hipBLASLt, bf16 output, no
    bias, no loss, same operands' shapes and value distribution): a yardstick beside `roofline`, not a product path --
    nothing in image_caption_amd calls torch.matmul."""
    try:
        M, d, V = 32 * B, 768, 50260
        g = torch.Generator(device=dev).manual_seed(0)
        h = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(V, d, device=dev, generator=g) * 0.03).to(torch.bfloat16)
        out = torch.empty(M, V, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            torch.matmul(h, w.t(), out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            torch.matmul(h, w.t(), out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        return {"what": f"torch.matmul (hipBLASLt) {M}x{V}x{d}, bf16 output, no bias / loss / argmax, run alone back to back",
                "ms_per_launch": round(ms, 4), "tflops": round(2.0 * M * d * V / (ms * 1e-3) / 1e12, 1)}
    except Exception as ex:  # noqa: BLE001
        return {"error": str(ex)[:120]}


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
    from image_caption_amd.dp import init_process_group as dp_init
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
            dp_init("nccl", dev, rank=0, world_size=1)
        else:
            dp_init("nccl", dev)

    from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder
    from image_caption_amd.engine import EngineOptions, KernelTimer
    from image_caption_amd.synthetic import make_batch

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.no_ppipe:
        import ctypes
        from image_caption_amd import _lib
        _lib.load().capdecdbg_gemm_ppipe(ctypes.c_int(0))
    if args.no_fused_attn_bwd:
        import ctypes
        from image_caption_amd import _lib
        _lib.load().capdecdbg_attn_fused_bwd(ctypes.c_int(0))
    if args.no_half_tiles:
        import ctypes
        from image_caption_amd import _lib
        _lib.load().capdecdbg_head_fwd_half_tiles(ctypes.c_int(0))

    def sync():
        torch.cuda.synchronize()
        if world > 1 or rehearse:
            dist.barrier()
            torch.cuda.synchronize()

    def build(n_blocks, mask_mode, activation):
        opts = EngineOptions(compute_dtype=cdt, mask_mode=mask_mode, activation=activation,
                             scored_rows_only=not args.all_rows, fold_norm1=not args.no_fold_norm1)
        if args.vocab_slab:
            opts.vocab_slab = args.vocab_slab
        if args.no_ce_row_reference:
            opts.ce_row_reference = False
        if args.no_ce_bias_in_wgrad:
            opts.ce_bias_in_wgrad = False
        if args.dgrad_transposed_weights:
            opts.dgrad_transposed_weights = True
        if args.no_wgrad_stream:
            opts.wgrad_stream = False
        if args.no_head_k_panels:
            opts.head_k_panels = False
        if args.no_dl_panels:
            opts.dl_panels = False
        torch.manual_seed(42)                                           # config.py:27,41
        model = TransformerDecoder(768, 8, 6, 2048, 0.1, 50260, n_blocks=n_blocks, options=opts).to(dev)
        model.train()                                                   # decoder.py:144: dropout active
        return model, CaptionTrainer(model, lr=1e-4, weight_decay=0.01, bucket_bytes=args.bucket_mb << 20)

    def timed(model, trainer, B, patches, steps, warmup, spinup, spans_wanted=True, stage_steps=5):
        """spinup + warmup untimed steps, then `steps` steps between two barriers; the MAX over ranks of the elapsed
        time.  Returns (seconds, last step's outputs, event-bracket summary)."""
        batch = make_batch(B, P=patches, L=32, d=768, vocab_size=50260, seed=42 + rank, device=dev)
        inputs = (batch["image_embeddings"], batch["sos_embedding"], batch["caption_embeddings"])
        labels = batch["caption_labels"]
        for _ in range(spinup):
            trainer.step(inputs, labels, global_batch=B * world)
        sync()
        for _ in range(warmup):
            trainer.step(inputs, labels, global_batch=B * world)
        # inside the timed steps ONE kernel is bracketed: the dominant one, by the library's own event pair around that
        # launch.  An event record drains the stream (a 5-12 us bubble in the kernel trace): bracketing the five stages of
        # the vocabulary step cost 0.03-0.05 ms of a 3.2 ms step, bracketing every stage 0.13 ms.  The other launches'
        # durations come from separate, untimed steps bracketed stage by stage.
        timer = KernelTimer(only=())
        if spans_wanted:
            model.engine.timer = timer
            head_kernel_timing(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = trainer.step(inputs, labels, global_batch=B * world)
        sync()
        elapsed = time.perf_counter() - t0
        model.engine.timer = None
        hk = head_kernel_timing(False) if spans_wanted else None
        if world > 1 or rehearse:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        spans = timer.summary()
        if hk is not None:
            spans["logits_ce_fwd_kernel"] = hk
        if spans_wanted and stage_steps > 0:
            full = KernelTimer()
            model.engine.timer = full
            for _ in range(stage_steps):
                trainer.step(inputs, labels, global_batch=B * world)
            sync()
            model.engine.timer = None
            for name, rec in full.summary().items():
                if name not in spans:
                    spans[name] = (rec[0] * steps // stage_steps, rec[1])        # scaled to the timed steps' launch count
        return elapsed, out, spans

    if args.scaled_only:
        print(json.dumps({"scaled": scaled_bench(build, timed, args)}), flush=True)
        return
    if args.greedy_only:
        print(json.dumps({"greedy": greedy_bench(build(args.n_blocks, args.mask_mode, args.activation)[0], dev, args)}), flush=True)
        return
    model, trainer = build(args.n_blocks, args.mask_mode, args.activation)
    B = args.global_batch // world if args.global_batch else args.batch_per_gpu
    # Device spin-up, untimed and before the W warm-up steps: a process's first ~0.2 s of sustained load contains one or
    # two 20-35 ms stalls (tools/warmup_probe.py shows them: clock / power-state changes or runtime housekeeping, at a
    # different step from run to run), which a short warm-up leaves inside the timed steps.  A fixed number of steps, so
    # that every rank issues the same collectives.
    elapsed, out, spans = timed(model, trainer, B, args.patches, args.steps, args.warmup, args.spinup_steps, spans_wanted=not args.no_spans,
                                stage_steps=args.stage_steps)
    loss = out[0].item() / out[2]
    slab = model.engine.last_slab

    # configs[2] at this N: the same step at a fixed global batch of 2048 (every rank takes part: same collectives)
    g2048 = None
    if not args.no_global_2048 and not args.global_batch and args.n_blocks == 1 and 2048 % world == 0:
        Bg = 2048 // world
        if Bg == B:
            g2048 = {"value": round(B * world * args.steps / elapsed, 1), "per_gpu_batch": Bg, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                     "note": "identical to the main measurement at this N"}
        else:
            e2, _, _ = timed(model, trainer, Bg, args.patches, 10, 3, 10, spans_wanted=False)
            g2048 = {"value": round(2048 * 10 / e2, 1), "per_gpu_batch": Bg, "ms_per_step": round(e2 / 10 * 1e3, 3), "steps": 10}
        g2048.update({"metric": "train samples/sec at global batch 2048", "unit": "samples/s", "n_gpus": world, "scaling": "strong"})

    if rank == 0:
        samples = B * world * args.steps
        value = samples / elapsed
        # ---- roofline of the dominant kernel: the vocabulary-sized GEMMs of the fused logits+CE stage
        per_kernel, hbm_kernels = head_roofline(spans, args.steps, B, slab, args.dtype)
        blocks_ms = {k: round(ms * c / args.steps, 4) for k, (c, ms) in spans.items() if k.startswith(("block_", "join_"))}
        # the forward stage's bracket is the kernel's plus two small launches: the kernel's own entry stands for it
        cand = [k for k in per_kernel if not (k == "logits_ce_fwd" and "logits_ce_fwd_kernel" in per_kernel)]
        dom = max(cand, key=lambda k: per_kernel[k]["ms_per_launch"] * per_kernel[k]["launches_per_step"]) if cand else None
        roofline = None
        if dom:
            a = per_kernel[dom]["tflops"]
            roofline = {"kernel": KERNEL_NAMES[dom], "bound": "mfma", "achieved": a, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(a / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                        "traffic": pmc_traffic(dom, args, B, slab >= 50260),
                        "ms_per_launch": per_kernel[dom]["ms_per_launch"], "per_kernel": per_kernel, "hbm_bound_kernels": hbm_kernels,
                        "block_stage_ms_per_step": blocks_ms,
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
        if g2048 is not None:
            result["global_2048"] = g2048
        if world == 1 and not args.no_greedy:
            result["greedy"] = greedy_bench(build(args.n_blocks, args.mask_mode, args.activation)[0], dev, args)
            if not args.no_cpu_baseline:
                result["greedy"]["cpu_baseline"] = cpu_baseline_greedy()
        del model, trainer
        torch.cuda.empty_cache()
        if world == 1 and not args.no_scaled and args.n_blocks == 1:
            result["scaled"] = scaled_bench(build, timed, args)
        if world == 1 and roofline is not None and not args.no_yardstick and args.dtype == "bf16":
            roofline["library_yardstick"] = library_yardstick(dev, B)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), file=json_out, flush=True)
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


def scaled_bench(build, timed, args):
    """configs[4]: 12 blocks, 257 patches (S = 290), prefix-causal mask, 50260-token vocabulary, batch 256, one GPU."""
    B, P, nb = 256, 257, 12
    model, trainer = build(nb, "prefix_causal", args.activation)
    elapsed, out, _ = timed(model, trainer, B, P, 10, 3, 20, spans_wanted=False)
    value = B * 10 / elapsed
    # where the step goes: five more steps with event brackets around the stages (outside the timed region: a
    # bracket is two event records, and this configuration has some 300 of them per step)
    stage_ms = {}
    if args.stage_steps > 0:
        _, _, spans = timed(model, trainer, B, P, 5, 0, 0)
        stage_ms = {k: round(ms * c / 5, 3) for k, (c, ms) in spans.items()}
    tf = value * SCALED_TRAIN_FLOPS_PER_SAMPLE / 1e12
    top = max(stage_ms, key=stage_ms.get) if stage_ms else None
    return {"metric": "train samples/sec", "value": round(value, 1), "unit": "samples/s", "ms_per_step": round(elapsed / 10 * 1e3, 3),
            "steps": 10, "dtype": args.dtype,
            "config": {"workload": f"single training step at batch {B}, {nb} blocks, P={P} (S={P + 33}), d=768, H=8, ff=2048, "
                                   f"V=50260, prefix-causal mask, {args.activation}, dropout 0.1"},
            "final_loss": round(out[0].item() / out[2], 4),
            "algorithmic_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
            "algorithmic_gflop_per_sample": round(SCALED_TRAIN_FLOPS_PER_SAMPLE / 1e9, 1),
            "stage_ms_per_step": stage_ms, "dominant_stage": top}


def greedy_bench(model, dev, args):
    """configs[3]: KV-cached greedy captioning of 4096 synthetic images, L=32, one GPU."""
    from image_caption_amd.synthetic import make_batch
    n = args.greedy_images
    model.eval()
    b = make_batch(n, P=args.patches, L=1, d=768, vocab_size=50260, seed=7, device=dev)
    emb = torch.randn(50260, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    img, sos = b["image_embeddings"], b["sos_embedding"]
    # A freshly initialised model (not the one the training leg has just fitted to one batch of mostly-<pad> labels, which
    # emits <pad> for every image).  Its head scores every token within a hair of the others; six times the initial scale
    # (what the reference-composed fixtures use) gives decisions that depend on the image, which makes the token count
    # below a sanity signal.  The arithmetic per token is the same.
    with torch.no_grad():
        model.output_projection.weight.mul_(6.0)
    model._shadow.clear()
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
        roof["logits_argmax (head_fwd_kernel<2,4,8,4,false,false>: logits+argmax, nothing stored but the per-slot partials)"] = {
            "bound": "mfma", "launches": cnt, "ms_per_launch": round(ms, 4), "achieved": round(tf, 1),
            "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    model.train()
    return {"metric": "greedy captions/sec", "value": round(n / dt, 1), "unit": "captions/s", "images": n, "max_len": 32,
            "ms_per_batch": round(dt * 1e3, 2), "kv_cache": True, "dtype": args.dtype,
            "distinct_tokens": int(ids.unique().numel()), "roofline": roof}


if __name__ == "__main__":
    main()
