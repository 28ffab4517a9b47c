"""Data-parallel gradient exchange: bucketed all-reduce overlapped with backward.

The reference is single-process (there is nothing to port); the batch dimension of
``decoder_inputs`` / ``caption_labels`` (decoder.py:163,166) is independent and the
loss is a mean (decoder.py:140), so each rank takes a slice of the batch and the
ranks sum their gradients.

The flat gradient buffer of ``ParamStore`` is laid out in the order the backward
FINISHES gradients: vocabulary-weight slabs, vocabulary bias, then the blocks from
last to first.  The finished region is therefore always a growing prefix of the
buffer, and a bucket is simply the next contiguous slice of it.  Each bucket's
all-reduce (RCCL over xGMI when the backend is "nccl") is issued asynchronously
as soon as the slice is complete; the remaining backward kernels keep running on
the compute stream.  ``finish`` makes the compute stream wait for all buckets
before the optimiser kernel reads the gradients.

``compress="bf16"`` halves the bytes on the links: a bucket is rounded to bfloat16
into a staging buffer on the compute stream, the staging slice is what the ranks
sum, and after the bucket's wait it is widened back into the f32 gradient buffer,
which the optimiser reads as before (f32 moments, f32 update).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None

HEAD_W, HEAD_B = "output_projection.weight", "output_projection.bias"


class GradReducer:
    def __init__(self, store, process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True,
                 compress: Optional[str] = None, record_timing: bool = False):
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.store = store
        self.group = process_group
        self.compress = compress
        self._stage: Optional[torch.Tensor] = None    # bf16 image of the gradient buffer (compress="bf16"), made on first use
        # record_timing: an event on the compute stream at every bucket's issue and at the start of drain()/finish()
        # (= the end of the backward), so that a run can report how long before the end each bucket left
        self.record_timing = record_timing
        self.issue_events: List = []
        self.backward_end_event = None
        # CAPDEC_DP_SINGLE_RANK=1 keeps the exchange on with one rank (every all-reduce is then an identity):
        # a rehearsal of the RCCL call pattern on a one-GPU machine
        single_ok = os.environ.get("CAPDEC_DP_SINGLE_RANK", "0") == "1"
        self.enabled = bool(dist is not None and dist.is_available() and dist.is_initialized()
                            and (dist.get_world_size(process_group) > 1 or single_ok))
        self.world_size = dist.get_world_size(process_group) if self.enabled else 1
        self.rank = dist.get_rank(process_group) if self.enabled else 0
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self.ready_end = 0       # gradients in g[:ready_end] are final
        self.sent_end = 0        # g[:sent_end] has been handed to all-reduce
        self.works: List = []
        self.bucket_log: List = []   # (start, end) of every bucket of the last step
        # end offset (exclusive, padded) of every layout entry, in layout order
        self._ends = {}
        layout = store.layout
        for idx, (key, shape, off) in enumerate(layout):
            nxt = layout[idx + 1][2] if idx + 1 < len(layout) else store.numel
            self._ends[key] = nxt
        self._block_end = {}
        nb = store.geo.n_blocks
        for key, shape, off in layout:
            if key in (HEAD_W, HEAD_B):
                continue
            i = int(key.split(".")[1]) if key.startswith("blocks.") else 0
            self._block_end[i] = max(self._block_end.get(i, 0), self._ends[key])

    def begin(self):
        self.ready_end = 0
        self.sent_end = 0
        self.works = []
        self.bucket_log = []
        self.issue_events = []
        self.backward_end_event = None

    @staticmethod
    def _convert(src: torch.Tensor, dst: torch.Tensor):
        if src.is_cuda:
            from . import ops
            ops.cast(src, dst)
        else:                      # host tensors (the gloo tests of the bucket logic)
            dst.copy_(src)

    def _mark(self):
        if self.record_timing and self.store.g.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return None

    def _launch(self, end: int):
        if end <= self.sent_end:
            return
        a = self.sent_end
        sl = self.store.g[a:end]
        if self.compress == "bf16":
            if self._stage is None or self._stage.device != self.store.g.device:
                self._stage = torch.empty(self.store.numel, dtype=torch.bfloat16, device=self.store.g.device)
            # on the compute stream, after the kernels that produced the slice; the collective is ordered after it
            self._convert(sl, self._stage[a:end])
            sl = self._stage[a:end]
        self.bucket_log.append((a, end))
        self.issue_events.append(self._mark())
        self.works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.sent_end = end

    def _complete(self, work, a: int, b: int):
        work.wait()
        if self.compress == "bf16":
            self._convert(self._stage[a:b], self.store.g[a:b])

    def payload_bytes(self) -> int:
        """Bytes one rank hands to the collective per step."""
        return self.store.numel * (2 if self.compress == "bf16" else 4)

    def issue_lead_ms(self):
        """[(start, end, ms between the bucket's issue and the end of the backward)] of the last step (record_timing)."""
        if self.backward_end_event is None:
            return []
        torch.cuda.synchronize()
        return [(a, b, ev.elapsed_time(self.backward_end_event)) for (a, b), ev in zip(self.bucket_log, self.issue_events) if ev is not None]

    def on_grad_ready(self, key, v0: int, nv: int):
        """Called by the engine when a piece of the gradient is final:
        (HEAD_W, v0, nv) vocabulary rows [v0, v0+nv); (HEAD_B, ..) the bias;
        (("block", i), ..) every tensor of block i."""
        if not self.enabled:
            return
        if key == HEAD_W:
            off, shape = self.store.offsets[HEAD_W]
            end = off + (v0 + nv) * shape[1]
            if v0 + nv >= shape[0]:
                end = self._ends[HEAD_W]
        elif key == HEAD_B:
            end = self._ends[HEAD_B]
        else:
            end = self._block_end[key[1]]
        self.ready_end = max(self.ready_end, end)
        if self.overlap and self.ready_end - self.sent_end >= self.bucket_elems:
            self._launch(self.ready_end)

    def _launch_rest(self):
        self._launch(self.store.numel if self.ready_end >= self.store.numel else self.ready_end)
        if self.sent_end < self.store.numel:          # anything never announced (defensive)
            self._launch(self.store.numel)

    def finish(self):
        if not self.enabled:
            return
        self.backward_end_event = self._mark()
        self._launch_rest()
        for w, (a, b) in zip(self.works, self.bucket_log):
            self._complete(w, a, b)
        self.works = []

    def drain(self, apply):
        """finish(), bucket by bucket: ``apply(start, end)`` is called for each bucket, in buffer order, as
        soon as the current stream has been ordered after THAT bucket's all-reduce.  The optimiser update of
        the early buckets (the vocabulary matrix) thus runs while the last bucket is still being reduced."""
        if not self.enabled:
            apply(0, self.store.numel)
            return
        self.backward_end_event = self._mark()
        self._launch_rest()
        for w, (a, b) in zip(self.works, self.bucket_log):
            self._complete(w, a, b)
            apply(a, b)
        self.works = []


def init_process_group(backend: str = "nccl", device: Optional[torch.device] = None, high_priority: bool = True, **kw):
    """``torch.distributed.init_process_group`` for one process per GPU, with the communicator's stream created at
    the highest stream priority when the backend is RCCL ("nccl"): the large products of the backward launch one
    workgroup per CU, and a collective's workgroups only start on a CU between two of those, so they should be first
    in line when one retires.  ``CAPDEC_RCCL_HIGH_PRIORITY=0`` (or ``high_priority=False``) asks for the default
    priority; a torch build without the option falls back to it."""
    if device is not None and device.type == "cuda":
        kw.setdefault("device_id", device)
    want = high_priority and backend == "nccl" and os.environ.get("CAPDEC_RCCL_HIGH_PRIORITY", "1") != "0"
    if want:
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except (AttributeError, TypeError):
            opts = None
        if opts is not None:
            kw.setdefault("pg_options", opts)
    dist.init_process_group(backend, **kw)
    return dist.group.WORLD


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of n samples owned by `rank` (remainder spread over
    the first ranks).  Samples are independent along dim 1 of decoder_inputs /
    dim 0 of the labels (decoder.py:163,166)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
