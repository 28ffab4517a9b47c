"""The processed-dataset file and the batches cut from it.

File format (upload_to_wandb.py:17-31, read back at decoder.py:85): one ``torch.save`` of
``{'decoder_inputs': [S, N, d] float, 'caption_input_ids': [N, L] int64, 'caption_labels':
[N, L] int64}`` -- the per-batch outputs of process_batch concatenated along the sample axis.
A training batch is the strided slice ``decoder_inputs[:, i:j, :]`` (decoder.py:163): S
separate runs of (j-i)*d elements, 703 KB per sample at the default shape.

On an MI355X the whole 10 %-of-Flickr30k dataset (14,500 samples, 10.2 GB f32) is a small
fraction of the 288 GB of HBM, so the default is to keep it RESIDENT on the device and hand
the kernels the strided slice itself (capdec_seqfirst_ln_fwd normalises it where it lies; no
contiguous copy is made).  Datasets that do not fit stay in pinned host memory and are
streamed: each batch's S runs are copied asynchronously on a copy stream, two batches ahead
of the compute stream."""
from __future__ import annotations

import os
from typing import Dict, Iterable, Iterator, Optional, Tuple

import torch

KEYS = ("decoder_inputs", "caption_input_ids", "caption_labels")


def _check(data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    missing = [k for k in KEYS if k not in data]
    if missing:
        raise ValueError(f"processed dataset lacks {missing}; expected keys {KEYS}")
    x, ids, labels = (data[k] for k in KEYS)
    if x.dim() != 3 or ids.dim() != 2 or labels.dim() != 2:
        raise ValueError("decoder_inputs must be [S, N, d]; caption_input_ids / caption_labels [N, L]")
    n = x.size(1)
    if ids.size(0) != n or labels.size(0) != n or ids.shape != labels.shape:
        raise ValueError(f"sample counts disagree: decoder_inputs {n}, ids {tuple(ids.shape)}, labels {tuple(labels.shape)}")
    if not x.is_floating_point() or ids.dtype != torch.int64 or labels.dtype != torch.int64:
        raise ValueError("decoder_inputs must be floating point; caption ids / labels int64")
    return {k: data[k] for k in KEYS}


def concat_batches(batches: Iterable[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """The per-batch dicts of process_batch -> one dataset dict (upload_to_wandb.py:17-30):
    decoder_inputs along dim 1, ids and labels along dim 0."""
    xs, ids, labels = [], [], []
    for b in batches:
        xs.append(b["decoder_inputs"])
        ids.append(b["caption_input_ids"])
        labels.append(b["caption_labels"])
    if not xs:
        raise ValueError("no batches")
    return _check({"decoder_inputs": torch.cat(xs, dim=1), "caption_input_ids": torch.cat(ids, dim=0),
                   "caption_labels": torch.cat(labels, dim=0)})


def save_processed_dataset(data, path: str) -> str:
    """Writes ``processed_dataset.pt`` as the reference does (a list of batch dicts is concatenated
    first).  ``path`` may be a directory."""
    if not isinstance(data, dict):
        data = concat_batches(data)
    data = {k: v.detach().cpu() for k, v in _check(data).items()}
    if os.path.isdir(path):
        path = os.path.join(path, "processed_dataset.pt")
    torch.save(data, path)
    return path


def load_processed_dataset(path: str, *, mmap: bool = True) -> Dict[str, torch.Tensor]:
    """Reads a ``processed_dataset.pt`` (decoder.py:85) with a loader that executes nothing from
    the file (``weights_only=True``); ``mmap`` maps the tensor data instead of reading it, so a
    dataset larger than host memory can still be streamed."""
    if os.path.isdir(path):
        path = os.path.join(path, "processed_dataset.pt")
    try:
        data = torch.load(path, map_location="cpu", weights_only=True, mmap=mmap)
    except (RuntimeError, ValueError):
        if not mmap:
            raise
        data = torch.load(path, map_location="cpu", weights_only=True)      # files written by old torch versions cannot be mapped
    if not isinstance(data, dict):
        raise ValueError("processed dataset file does not hold a dict of tensors")
    return _check(data)


class BatchSource:
    """Yields ``(decoder_inputs [S, b, d], caption_labels [b, L])`` device tensors for the samples
    ``[lo, hi)`` in order, ``batch_size`` at a time (decoder.py:157-166).

    resident=True (default when the split fits ``resident_fraction`` of the free device memory):
    the split is copied to the device once and every batch is a strided VIEW of it.
    resident=False: pinned host staging; batch k+2 is in flight on a copy stream while batch k
    is consumed.  A yielded batch stays valid until two more have been drawn."""

    def __init__(self, data: Dict[str, torch.Tensor], batch_size: int, device, *, lo: int = 0, hi: Optional[int] = None,
                 resident: Optional[bool] = None, resident_fraction: float = 0.5, depth: int = 2):
        data = _check(data)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchSource feeds the GPU path only (device must be a CUDA device)")
        n = data["decoder_inputs"].size(1)
        self.lo, self.hi = lo, n if hi is None else hi
        if not (0 <= self.lo <= self.hi <= n) or batch_size <= 0:
            raise ValueError(f"bad sample range [{lo}, {hi}) of {n} or batch size {batch_size}")
        self.batch_size = batch_size
        self.depth = max(1, depth)
        x = data["decoder_inputs"][:, self.lo:self.hi, :]
        labels = data["caption_labels"][self.lo:self.hi]
        nbytes = x.numel() * x.element_size()
        if resident is None:
            free, _ = torch.cuda.mem_get_info(self.device)
            resident = nbytes <= resident_fraction * free
        self.resident = bool(resident)
        if self.resident:
            self.x = x.to(self.device)              # [S, n, d]; a strided source keeps its sample-major gaps out
            self.labels = labels.to(self.device)
        else:
            self.x = x if x.is_pinned() else self._pin(x)
            self.labels = labels.to(self.device)    # a few MB: always resident
            S, _, d = x.shape
            self._stage = [torch.empty(S, batch_size, d, dtype=x.dtype, device=self.device) for _ in range(self.depth + 1)]
            self._copy_stream = torch.cuda.Stream(device=self.device)

    @staticmethod
    def _pin(x: torch.Tensor, host_fraction: float = 0.5) -> torch.Tensor:
        """A pinned copy of the split when it fits comfortably in the host memory available now (pinning materialises
        the whole slice); otherwise the view itself -- a mapped file larger than memory streams through pageable copies."""
        nbytes = x.numel() * x.element_size()
        available = None                       # None: no figure to go by -> try to pin and fall back on failure
        try:
            import psutil
            available = psutil.virtual_memory().available
        except Exception:
            try:
                with open("/proc/meminfo") as f:
                    for line in f:
                        if line.startswith("MemAvailable:"):
                            available = int(line.split()[1]) * 1024
                            break
            except (OSError, ValueError, IndexError):
                available = None
        if available is not None and nbytes > host_fraction * available:
            return x
        try:
            return x.contiguous().pin_memory()
        except RuntimeError:
            return x

    def __len__(self) -> int:
        n = self.hi - self.lo
        return (n + self.batch_size - 1) // self.batch_size

    def _issue(self, k: int):
        """Starts the copy of batch k into its staging buffer; returns (view, event)."""
        i = k * self.batch_size
        j = min(i + self.batch_size, self.hi - self.lo)
        buf = self._stage[k % len(self._stage)][:, :j - i]
        # the consumer of the batch that used this buffer last must be done with it
        self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy_stream):
            for s in range(self.x.size(0)):          # S contiguous runs: one async copy each, no host-side gather
                buf[s].copy_(self.x[s, i:j], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return buf, ev, (i, j)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        nb = len(self)
        if self.resident:
            for k in range(nb):
                i = k * self.batch_size
                j = min(i + self.batch_size, self.hi - self.lo)
                yield self.x[:, i:j, :], self.labels[i:j]
            return
        pending = [self._issue(k) for k in range(min(self.depth, nb))]
        for k in range(nb):
            buf, ev, (i, j) = pending.pop(0)
            if k + self.depth < nb:
                pending.append(self._issue(k + self.depth))
            torch.cuda.current_stream(self.device).wait_event(ev)
            yield buf, self.labels[i:j]
