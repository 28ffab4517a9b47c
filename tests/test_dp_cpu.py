"""The data-parallel gradient exchange on 2 CPU processes (gloo): bucket boundaries
follow the order in which the backward finishes gradients, every element is summed
exactly once, and the per-rank batch shards tile the global batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, overlap, q, compress=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_caption_amd.dp import GradReducer
        from image_caption_amd.engine import HEAD_B, HEAD_W, ModelGeometry, ParamStore
        geo = ModelGeometry(64, 2, 128, 503, n_blocks=2)
        st = ParamStore(geo, "cpu", torch.float32, with_optimizer_state=False)
        red = GradReducer(st, None, bucket_bytes=40_000, overlap=overlap, compress=compress)
        # (bf16 exchange: values and sums below 256 are exact in bfloat16, so the same equality holds)
        big = 1000 if compress is None else 64
        assert red.payload_bytes() == st.numel * (4 if compress is None else 2)
        assert red.enabled and red.world_size == world
        for step in range(2):
            st.g.copy_(torch.arange(st.numel, dtype=torch.float32) % 61 + (rank + 1) * big + step)
            red.begin()
            V, slab = 503, 128
            for v0 in range(0, V, slab):                      # the order loss_and_grads announces gradients in
                red.on_grad_ready(HEAD_W, v0, min(slab, V - v0))
            red.on_grad_ready(HEAD_B, 0, V)
            for i in (1, 0):
                red.on_grad_ready(("block", i), 0, 0)
            if step == 0:
                red.finish()
            else:                                             # bucket-by-bucket completion: every element handed over once, in order
                seen = []
                red.drain(lambda a, b: seen.append((a, b)))
                assert seen == red.bucket_log and seen[0][0] == 0 and seen[-1][1] == st.numel
            want = (torch.arange(st.numel, dtype=torch.float32) % 61) * world + sum((r + 1) * big + step for r in range(world))
            assert torch.equal(st.g, want), "all-reduce result wrong"
            log = red.bucket_log
            assert log[0][0] == 0 and log[-1][1] == st.numel
            assert all(a[1] == b[0] for a, b in zip(log, log[1:])), "buckets must tile the flat buffer in order"
            if overlap:
                assert len(log) >= 3
            else:
                assert len(log) == 1
        q.put((rank, "ok", len(red.bucket_log)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"error: {e!r}", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,compress", [(True, None), (False, None), (True, "bf16")])
def test_bucketed_allreduce_two_ranks(overlap, compress):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + (1 if overlap else 0) + (2 if compress else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q, compress)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_range_tiles_the_batch():
    from image_caption_amd.dp import shard_range
    for n, w in ((2048, 8), (2048, 3), (5, 8), (256, 1)):
        cuts = [shard_range(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_reducer_is_a_no_op():
    from image_caption_amd.dp import GradReducer
    from image_caption_amd.engine import ModelGeometry, ParamStore
    st = ParamStore(ModelGeometry(64, 2, 128, 503), "cpu", torch.float32, with_optimizer_state=False)
    red = GradReducer(st)
    assert not red.enabled and red.world_size == 1
    red.begin(); red.on_grad_ready("output_projection.weight", 0, 503); red.finish()
