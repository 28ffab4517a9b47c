"""Data-parallel training on the GPU: two ranks (both on cuda:0, gloo carrying the CUDA gradient
buffer) each train on half of a batch; their parameters after two optimiser steps must equal those
of one process training on the whole batch.  Exercises CaptionTrainer + GradReducer end to end:
bucket launches from the engine's gradient-ready hooks, stream ordering against the side-stream
weight-gradient products, and the 1/world scaling inside the fused Adam."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from helpers import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, dtype_name, q, batch=6, dropout=0.0, backend="gloo", compress=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":
        torch.cuda.set_device(rank)              # RCCL: one GPU per rank
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import helpers as H
        from image_caption_amd.decoder import CaptionTrainer, TransformerDecoder
        from image_caption_amd.dp import shard_range
        from image_caption_amd.engine import EngineOptions
        dtype = getattr(torch, dtype_name)
        sh = H.Shape(d=64, nhead=2, ff=128, vocab=503, P=12, L=8, B=batch)
        sd = H.make_state_dict(sh, seed=501)
        img, sos, cap, labels = H.make_embeddings(sh, seed=502)

        def train(lo, hi, distributed):
            m = TransformerDecoder(sh.d, sh.nhead, 6, sh.ff, dropout, sh.vocab, options=EngineOptions(compute_dtype=dtype))
            m.load_state_dict(sd)
            m = m.to("cuda")
            m.train()
            tr = CaptionTrainer(m, lr=1e-3, weight_decay=0.01, bucket_bytes=64 << 10, grad_compress=compress if distributed else None)
            if not distributed:
                tr.reducer.enabled = False
                tr.reducer.world_size = 1
            assert tr.reducer.enabled == distributed
            losses = []
            for _ in range(2):
                out = tr.step((img[lo:hi].cuda(), sos[lo:hi].cuda(), cap[lo:hi].cuda()), labels[lo:hi].cuda(),
                              global_batch=sh.B if distributed else None)
                losses.append(out[0].item() / out[2])
            nb = len(tr.reducer.bucket_log)
            return {k: v.detach().float().cpu() for k, v in m.state_dict().items()}, losses, nb, m.options.dropout_seed

        lo, hi = shard_range(sh.B, rank, world)
        got, losses, nbuckets, seed = train(lo, hi, True)
        dist.barrier()
        msg = "ok"
        if dropout > 0.0:
            # the ranks draw different masks (so no single-process run reproduces the step), yet every rank must hold
            # the same parameters afterwards and they must be finite
            flat = torch.cat([got[k].reshape(-1) for k in sorted(got)])
            seeds = [None] * world
            dist.all_gather_object(seeds, seed)
            if backend == "nccl":
                flat = flat.cuda()
            both = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            if len(set(seeds)) != world:
                msg = f"ranks share a dropout seed: {seeds}"
            elif not all(torch.equal(both[0], b) for b in both[1:]):
                msg = "ranks ended with different parameters"
            elif not bool(torch.isfinite(flat).all()):
                msg = "non-finite parameters"
        elif rank == 0:
            want, ref_losses, _, _ = train(0, sh.B, False)
            tol = 2e-6 if dtype == torch.float32 else 2e-3
            if compress == "bf16":
                # Adam normalises: where the two ranks' gradients nearly cancel, rounding them to bfloat16 (2^-9 relative)
                # before the sum can flip the sign of the sum, and the parameter then moves by lr the other way -- no
                # element-wise bound below 2 * lr * steps holds.  What is bounded is the direction of each tensor's update
                # (cosine against the f32 exchange's update) and its size.
                for k in want:
                    du, dw = (got[k] - sd[k]).double().flatten(), (want[k] - sd[k]).double().flatten()
                    if float(dw.norm()) == 0.0:
                        continue
                    c = float(du @ dw / (du.norm() * dw.norm() + 1e-30))
                    if c < 0.98 or float((du - dw).abs().max()) > 2 * 2 * 1e-3 * 1.05:
                        msg = f"update of {k}: cosine {c:.4f} against the f32 exchange, max difference {float((du - dw).abs().max()):.2e}"
                        break
                want = {}
            for k in want:
                err = (got[k] - want[k]).abs().max().item()
                if err > tol * (want[k].abs().max().item() + 1.0):
                    msg = f"parameter {k} differs by {err:.3e}"
                    break
            if nbuckets < 2:
                msg = f"expected several buckets, saw {nbuckets}"
        q.put((rank, msg))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"error: {e!r} {traceback.format_exc()[-600:]}"))
    finally:
        dist.destroy_process_group()


# batch 6: equal shards; batch 5: shards of 3 and 2 samples, whose gradients must be weighted by their row counts
# (the loss is a mean over the global batch); dropout 0.1: per-rank masks
@pytest.mark.parametrize("dtype_name,batch,dropout", [("float32", 6, 0.0), ("bfloat16", 6, 0.0), ("float32", 5, 0.0),
                                                       ("float32", 6, 0.1)])
def test_two_rank_training_matches_single_process(dtype_name, batch, dropout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40) + (0 if dtype_name == "float32" else 1) + 2 * batch + (50 if dropout else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype_name, q, batch, dropout)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_two_rank_training_with_bf16_gradient_exchange():
    """GradReducer(compress="bf16"): after two steps every tensor's UPDATE points the way single-process training's
    does (cosine >= 0.98; its gradients never leave f32) and no element is further off than two Adam steps can
    carry it; several buckets were exchanged."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29810 + (os.getpid() % 40)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "float32", q, 6, 0.0, "gloo", "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two GPUs")
@pytest.mark.parametrize("dtype_name,batch,dropout", [("float32", 6, 0.0), ("bfloat16", 5, 0.0), ("float32", 6, 0.1)])
def test_two_rank_training_over_rccl(dtype_name, batch, dropout):
    """The same comparison with one GPU per rank and the gradient buckets reduced by RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29890 + (os.getpid() % 40) + (0 if dtype_name == "float32" else 1) + 2 * batch + (50 if dropout else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype_name, q, batch, dropout, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
