"""Host-side orchestration of the captioning hot path on libcapdec kernels.

``DecoderEngine`` sequences the kernels of one training step (prefix join ->
blocks forward -> fused vocabulary projection + cross-entropy -> backward) and of
KV-cached greedy decoding.  It owns only bookkeeping and reusable device buffers;
every floating-point operation on activations happens in libcapdec.so.

What each stage replaces in the reference:
  prefix join                 img_text_concat.py:115-123
  block forward               decoder.py:53-63
  scored-row vocab + CE       decoder.py:66,172-185
  backward                    decoder.py:189 (autograd of the above)
  greedy decoding             no counterpart (the reference has no generation code)
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib, ops
from ._lib import ACT_GELU, ACT_RELU, GATE_GELU_PRE, GATE_RELU_OUT, MASK_NONE, MASK_PREFIX_CAUSAL, MASK_REFERENCE

BLOCK_PARAM_SHAPES = (
    ("self_attn.in_proj_weight", lambda d, ff: (3 * d, d)),
    ("self_attn.in_proj_bias", lambda d, ff: (3 * d,)),
    ("self_attn.out_proj.weight", lambda d, ff: (d, d)),
    ("self_attn.out_proj.bias", lambda d, ff: (d,)),
    ("linear1.weight", lambda d, ff: (ff, d)),
    ("linear1.bias", lambda d, ff: (ff,)),
    ("linear2.weight", lambda d, ff: (d, ff)),
    ("linear2.bias", lambda d, ff: (d,)),
    ("norm1.weight", lambda d, ff: (d,)),
    ("norm1.bias", lambda d, ff: (d,)),
    ("norm2.weight", lambda d, ff: (d,)),
    ("norm2.bias", lambda d, ff: (d,)),
)
MATRIX_KEYS = ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight")
HEAD_W, HEAD_B = "output_projection.weight", "output_projection.bias"

_MASKS = {"none": MASK_NONE, "reference": MASK_REFERENCE, "prefix_causal": MASK_PREFIX_CAUSAL}
_ACTS = {"relu": ACT_RELU, "gelu": ACT_GELU}


@dataclass
class EngineOptions:
    """Knobs that have no counterpart in the reference's config.py."""
    compute_dtype: torch.dtype = torch.bfloat16   # bfloat16 = MFMA path, float32 = exact validation path
    mask_mode: str = "reference"                  # decoder.py:31-38 literally; "prefix_causal" / "none"
    activation: str = "relu"                      # decoder.py:60; "gelu" is an extension
    vocab_slab: int = 0                           # vocabulary columns per backward slab; 0: as wide as dl_budget_bytes allows
    dl_budget_bytes: int = 16 << 30               # cap on the d(logits) slab buffer (the whole vocabulary fits up to batch 5000 per GPU)
    wgrad_stream: bool = True                     # the block's weight-gradient products run on a second HIP stream
    keep_softmax_numerators: bool = True          # forward keeps exp(z - slot max) when d(logits) fits: no recomputation in backward
    dgrad_transposed_weights: bool = False        # data gradients whose weight is read along `out` (vocabulary, linear2, out_proj) take a
                                                  # transposed bf16 copy made once per step: a row-major second operand (measured slower in the step)
    ce_bias_in_wgrad: bool = True                 # ... and the vocabulary bias gradient taken as one more column of the weight-gradient product
    ce_row_reference: bool = True                 # ... kept against one reference per row (the label's logit), so that d(logits) is a
                                                  # row factor times the numerators and the backward products read them as they are
    composed_decode_step: bool = True             # greedy decoding: one capdec_decode_step call per token (False: a dozen calls)
    scored_rows_only: bool = True                 # last block: Q / MLP only for the rows the loss reads (exact)
    fold_norm1: bool = True                       # training without input gradient: block 0's norm1 affine is folded into in_proj
    dl_panels: bool = True                        # the kept numerators / d(logits) are stored as a column-panel image [V/32][rows][32]
    head_k_panels: bool = True                    # the vocabulary product reads k-panel images ([d/32][rows][32]) of h and of the vocabulary
                                                  # matrix: whole 128-byte lines per load instead of half lines (bit-identical results)
    dropout_seed: int = 42


@dataclass
class ModelGeometry:
    d_model: int
    nhead: int
    dim_feedforward: int
    vocab_size: int
    n_blocks: int = 1
    dropout: float = 0.0

    @property
    def hd(self) -> int:
        return self.d_model // self.nhead


def block_key(i: int, name: str, n_blocks: int) -> str:
    """state_dict key of a block parameter: the reference's own names for a
    one-block model, ``blocks.{i}.<name>`` for stacked blocks (i >= 1)."""
    return name if i == 0 else f"blocks.{i}.{name}"


class ParamStore:
    """All trainable tensors in ONE flat f32 buffer (+ gradient, Adam moments and a
    compute-dtype shadow of the matrices), with named views.  One flat buffer means
    one fused Adam launch and contiguous all-reduce buckets."""

    def __init__(self, geo: ModelGeometry, device, compute_dtype, with_optimizer_state: bool):
        self.geo = geo
        self.compute_dtype = compute_dtype
        d, ff, V = geo.d_model, geo.dim_feedforward, geo.vocab_size
        self.layout: List = []            # (key, shape, offset)
        off = 0
        # head first: its gradient is produced first in backward, so its all-reduce can start first
        # the vocabulary matrix is followed by zero rows up to a multiple of 128 rows (they stay zero: their
        # gradient is never written), so that products over the vocabulary can run whole 128-row tiles
        self.head_pad_rows = (V + 127) // 128 * 128 - V
        for key, shape, n in ((HEAD_W, (V, d), (V + self.head_pad_rows) * d), (HEAD_B, (V,), V)):
            self.layout.append((key, shape, off))
            off += (n + 63) // 64 * 64
        for i in reversed(range(geo.n_blocks)):
            for name, fn in BLOCK_PARAM_SHAPES:
                shape = fn(d, ff)
                self.layout.append((block_key(i, name, geo.n_blocks), shape, off))
                off += (math.prod(shape) + 63) // 64 * 64
        self.numel = off
        self.w = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device) if with_optimizer_state else None
        self.v = torch.zeros(off, dtype=torch.float32, device=device) if with_optimizer_state else None
        self.shadow = torch.zeros(off, dtype=compute_dtype, device=device) if compute_dtype != torch.float32 else self.w
        self.offsets = {k: (o, s) for k, s, o in self.layout}
        # k-panel image of the vocabulary matrix's shadow (what the persistent logits kernel reads), kept current by
        # refresh_head_panels() after every change of the parameters
        self.head_panels = (torch.zeros(ops.kpanel_shape(V, d), dtype=torch.bfloat16, device=device)
                            if compute_dtype == torch.bfloat16 and d % 32 == 0 else None)
        # ... and of the blocks' four matrices, for a stack (its forward products run over every row: large row-major
        # products that read the image 5 % faster; for a single block's short products the refresh would cost more than
        # it returns).  block_panels[name][z] is the image of block n_blocks - 1 - z's matrix: the blocks lie in that
        # order, a constant number of elements apart, so one batched launch per matrix kind refreshes them all.
        self.block_panels = {}
        if compute_dtype == torch.bfloat16 and geo.n_blocks > 1 and d % 32 == 0 and ff % 32 == 0:
            for name, fn in BLOCK_PARAM_SHAPES:
                if name in MATRIX_KEYS:
                    rows, K = fn(d, ff)
                    self.block_panels[name] = torch.zeros((geo.n_blocks,) + ops.kpanel_shape(rows, K), dtype=torch.bfloat16, device=device)

    def view(self, buf: torch.Tensor, key: str) -> torch.Tensor:
        o, shape = self.offsets[key]
        return buf[o:o + math.prod(shape)].view(shape)

    def param(self, key): return self.view(self.w, key)
    def grad(self, key): return self.view(self.g, key)
    def compute(self, key): return self.view(self.shadow, key)

    def refresh_shadow(self):
        if self.shadow is not self.w:
            ops.cast(self.w, self.shadow)
        self.refresh_head_panels()
        self.refresh_block_panels()

    def refresh_head_panels(self):
        if self.head_panels is not None:
            ops.pack_kpanels(self.compute(HEAD_W), self.head_panels)

    def refresh_block_panels(self):
        """The images of the blocks' matrices from the current shadow: four launches, whatever the number of blocks."""
        nb = self.geo.n_blocks
        for name, panels in self.block_panels.items():
            first, shape = self.offsets[block_key(nb - 1, name, nb)]              # the block stored first
            stride = self.offsets[block_key(nb - 2, name, nb)][0] - first
            ops.pack_kpanels_batched(self.shadow, first, shape[0], shape[1], nb, stride, panels)

    def load(self, state: Dict[str, torch.Tensor]):
        for key, shape, _ in self.layout:
            self.param(key).copy_(state[key].to(self.w.device, torch.float32).reshape(shape))
        self.refresh_shadow()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {key: self.param(key).clone() for key, _, _ in self.layout}


class KernelTimer:
    """Brackets chosen kernel launches with events recorded on the stream the kernels
    are launched on (torch's current stream), so per-launch device time can be read
    after the fact without synchronising inside the step."""

    def __init__(self, only=None):
        """only: names to bracket (None: all).  An event pair costs the stream a few microseconds, so a timed run
        brackets only what it reports from that run."""
        self.spans: Dict[str, list] = {}
        self.only = None if only is None else frozenset(only)

    @contextlib.contextmanager
    def span(self, name: str):
        if self.only is not None and name not in self.only:
            yield
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        yield
        e.record()
        self.spans.setdefault(name, []).append((s, e))

    def summary(self) -> Dict[str, tuple]:
        """name -> (launch count, mean ms per launch)."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = (len(ms), sum(ms) / max(1, len(ms)))
        return out


IN_PROJ_MASTER = "self_attn.in_proj_weight.f32"       # block 0 only: the f32 parameter behind the compute-dtype matrix
PANELS = ".k_panels"                                  # suffix of the optional k-panel image of a block matrix ([K/32, rows, 32] bf16)


class _Weights:
    """Per-call view of the tensors the kernels read: matrices in compute dtype,
    biases / LayerNorm affine in f32."""

    def __init__(self, blocks, head_w, head_b, head_pad_rows: int = 0, head_w_panels=None):
        self.blocks = blocks
        self.head_w = head_w
        self.head_b = head_b
        self.head_pad_rows = head_pad_rows      # zero rows that physically follow head_w's last row
        self.head_w_panels = head_w_panels      # k-panel image of head_w ([d/32, V, 32] bf16, ops.pack_kpanels) or None


def weights_from_store(store: ParamStore) -> _Weights:
    geo = store.geo
    blocks = []
    for i in range(geo.n_blocks):
        b = {}
        for name, _ in BLOCK_PARAM_SHAPES:
            key = block_key(i, name, geo.n_blocks)
            b[name] = store.compute(key) if name in MATRIX_KEYS else store.param(key)
        if i == 0:
            b[IN_PROJ_MASTER] = store.param(block_key(0, "self_attn.in_proj_weight", geo.n_blocks))
        for name, panels in store.block_panels.items():
            b[name + PANELS] = panels[geo.n_blocks - 1 - i]
        blocks.append(b)
    return _Weights(blocks, store.compute(HEAD_W), store.param(HEAD_B), store.head_pad_rows, store.head_panels)


def _seqfirst_addressable(t: torch.Tensor) -> bool:
    """Whether capdec_seqfirst_ln_fwd can read the [S, B, d] view as it lies: f32 / bf16, unit stride along d,
    every row 16-byte aligned."""
    if t.dtype not in (torch.float32, torch.bfloat16) or t.stride(2) != 1 or t.data_ptr() % 16:
        return False
    per16 = 4 if t.dtype == torch.float32 else 8
    return all(st % per16 == 0 for st in t.stride()[:2])


class DecoderEngine:
    def __init__(self, geo: ModelGeometry, options: Optional[EngineOptions] = None):
        self.geo = geo
        self.opt = options or EngineOptions()
        if geo.d_model % geo.nhead:
            raise ValueError("d_model must be divisible by nhead")
        if self.opt.mask_mode not in _MASKS:
            raise ValueError(f"mask_mode must be one of {sorted(_MASKS)}")
        if self.opt.activation not in _ACTS:
            raise ValueError(f"activation must be one of {sorted(_ACTS)}")
        _lib.load()
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._fresh = set()
        self.cdt = self.opt.compute_dtype
        self.timer: Optional[KernelTimer] = None
        self._side: Optional[torch.cuda.Stream] = None
        self._side_used = False
        self.generation = 0          # bumped by every forward: the saved activations belong to the latest one only

    def _span(self, name: str):
        return self.timer.span(name) if self.timer is not None else contextlib.nullcontext()

    # ------------------------------------------------------------------ buffers
    def buf(self, name: str, shape, dtype, device, zero: bool = False) -> torch.Tensor:
        """A named scratch tensor.  One allocation per (name, dtype), as large as the largest shape asked for so far: a
        smaller request (the ragged last batch of an epoch) is a view of its beginning, not a second allocation."""
        shape = tuple(int(s) for s in shape)
        n = math.prod(shape)
        key = (name, dtype)
        flat = self._bufs.get(key)
        if flat is None or flat.numel() < n or flat.device != torch.device(device):
            flat = torch.empty(max(n, 1), dtype=dtype, device=device)
            self._bufs[key] = flat
            self._fresh.add(key)
        t = flat[:n].view(shape)
        if zero:
            t.zero_()
        return t

    def _was_fresh(self, name: str, dtype) -> bool:
        """Whether the buffer was (re)allocated since this was last asked (contents that are written once need rewriting)."""
        key = (name, dtype)
        fresh = key in self._fresh
        self._fresh.discard(key)
        return fresh

    def _unit_affine(self, d, dev):
        key = ("unit_affine", d, str(dev))
        t = self._bufs.get(key)
        if t is None:
            t = (torch.ones(d, dtype=torch.float32, device=dev), torch.zeros(d, dtype=torch.float32, device=dev))
            self._bufs[key] = t
        return t

    def _on_side_stream(self, fn):
        """Runs fn (launches of kernels nothing downstream in this backward waits for: the block's
        weight-gradient products) on a second stream, ordered after everything issued so far on the
        current one.  Those products have few tiles and very deep K; next to them the dgrad chain and
        the attention backward leave CUs idle.  _join_side_stream() orders the current stream after them."""
        if not self.opt.wgrad_stream:
            fn()
            return
        main = torch.cuda.current_stream()
        if self._side is None or self._side.device != main.device:
            self._side = torch.cuda.Stream(device=main.device)
        ev = torch.cuda.Event()
        ev.record(main)
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            fn()
        self._side_used = True

    def _join_side_stream(self):
        if self._side_used:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_used = False

    def _keys(self, site: int, step: int, train: bool):
        p = self.geo.dropout if train else 0.0
        return ops.dropout_keys(self.opt.dropout_seed, site, step, p)

    # ------------------------------------------------------------------ one block, forward
    def _block_fwd(self, i, W, x_in, B, S, q_start, train, step, dev, h1_ready=None, in_proj=None):
        """x_in [S*B,d] f32 -> rows q_start*B.. of the block output [Mt,d] f32.  Saves
        what the backward needs in self.saved[i]."""
        geo, cdt = self.geo, self.cdt
        d, ff, H, hd = geo.d_model, geo.dim_feedforward, geo.nhead, geo.hd
        M, r0 = S * B, q_start * B
        Mt = M - r0
        w = W.blocks[i]
        sv = {}
        if h1_ready is None:
            h1 = self.buf(f"h1.{i}", (M, d), cdt, dev)
            mean1 = self.buf(f"mean1.{i}", (M,), torch.float32, dev)
            rstd1 = self.buf(f"rstd1.{i}", (M,), torch.float32, dev)
            with self._span("block_ln_fwd"):
                ops.ln_fwd(x_in, w["norm1.weight"], w["norm1.bias"], h1, mean1, rstd1)
        else:
            h1, mean1, rstd1 = h1_ready
        qkv = self.buf(f"qkv.{i}", (M, 3 * d), cdt, dev)
        scale = hd ** -0.5
        # in_proj: the block's own, or the pair with norm1's affine folded in (h1 then holds the normalised rows)
        in_w, in_b = in_proj if in_proj is not None else (w["self_attn.in_proj_weight"], w["self_attn.in_proj_bias"])
        # images of the matrices (a stack's parameter store keeps them): read by the products large enough for the persistent kernel
        img = (lambda name: w.get(name + PANELS)) if self.opt.head_k_panels else (lambda name: None)
        with self._span("block_qkv_fwd"):
            if q_start == 0:
                ops.gemm(h1, in_w, qkv, bias=in_b, col_scale=scale, col_scale_n=d,
                         b_panels=img("self_attn.in_proj_weight") if in_proj is None else None)
            else:
                ops.gemm(h1, in_w[d:], qkv[:, d:], bias=in_b[d:])
                ops.gemm(h1[r0:], in_w[:d], qkv[r0:, :d], bias=in_b[:d], col_scale=scale, col_scale_n=d)
        attn_o = self.buf(f"attn_o.{i}", (M, d), cdt, dev)
        lse = self.buf(f"lse.{i}", (B * H * S,), torch.float32, dev)
        k_attn = self._keys(4 * i + 0, step, train)
        with self._span("block_attn_fwd"):
            ops.attn_fwd_packed(qkv, attn_o, lse, B, H, S, hd, q_start=q_start, nq=S - q_start,
                                mask_mode=_MASKS[self.opt.mask_mode], prefix_len=self.prefix_len, drop=k_attn)
        x_mid = self.buf(f"x_mid.{i}", (Mt, d), torch.float32, dev)
        k_r1 = self._keys(4 * i + 1, step, train)
        with self._span("block_out_proj_fwd"):
            ops.gemm(attn_o[r0:], w["self_attn.out_proj.weight"], x_mid, bias=w["self_attn.out_proj.bias"], drop=k_r1, residual=x_in[r0:],
                     b_panels=img("self_attn.out_proj.weight"))
        h2 = self.buf(f"h2.{i}", (Mt, d), cdt, dev)
        mean2 = self.buf(f"mean2.{i}", (Mt,), torch.float32, dev)
        rstd2 = self.buf(f"rstd2.{i}", (Mt,), torch.float32, dev)
        with self._span("block_ln_fwd"):
            ops.ln_fwd(x_mid, w["norm2.weight"], w["norm2.bias"], h2, mean2, rstd2)
        a1 = self.buf(f"a1.{i}", (Mt, ff), cdt, dev)
        k_h = self._keys(4 * i + 2, step, train)
        z1 = None
        if self.opt.activation == "gelu" and train:
            z1 = self.buf(f"z1.{i}", (Mt, ff), cdt, dev)
        x_out = self.buf(f"x_out.{i}", (Mt, d), torch.float32, dev)
        k_r2 = self._keys(4 * i + 3, step, train)
        with self._span("block_mlp_fwd"):
            ops.gemm(h2, w["linear1.weight"], a1, bias=w["linear1.bias"], act=_ACTS[self.opt.activation], drop=k_h, aux_out=z1,
                     b_panels=img("linear1.weight"))
            ops.gemm(a1, w["linear2.weight"], x_out, bias=w["linear2.bias"], drop=k_r2, residual=x_mid, b_panels=img("linear2.weight"))
        sv.update(x_in=x_in, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, attn_o=attn_o, lse=lse, x_mid=x_mid, h2=h2,
                  mean2=mean2, rstd2=rstd2, a1=a1, z1=z1, q_start=q_start, keys=(k_attn, k_r1, k_h, k_r2),
                  folded=in_proj is not None)
        self.saved[i] = sv
        return x_out

    # ------------------------------------------------------------------ forward to the scored rows
    def forward_hidden(self, W: _Weights, source, *, L_scored: Optional[int], train: bool, step: int = 0, pos=None,
                       keep_joined: bool = True, fold_norm1: bool = False):
        """source = ("concat", img, sos, cap) or ("tensor", tgt[S,B,d]).  Returns
        (x [rows, d] f32 of the last block, first row index r0, B, S)."""
        geo = self.geo
        d = geo.d_model
        if source[0] == "concat":
            _, img, sos, cap = source
            _lib.require_cuda(img, sos, cap)
            B, P, _ = img.shape
            S = P + 1 + cap.shape[1]
            self.prefix_len = P
        else:
            tgt = source[1]
            _lib.require_cuda(tgt)
            S, B, _ = tgt.shape
            self.prefix_len = source[2] if len(source) > 2 else 0
        if self.opt.mask_mode == "prefix_causal" and self.prefix_len <= 0 and source[0] != "concat":
            raise ValueError("prefix_causal masking needs the prefix length")
        dev = source[1].device
        M = S * B
        self.generation += 1
        self.saved = [None] * geo.n_blocks
        nb = geo.n_blocks
        last_q0 = (S - L_scored) if (L_scored is not None and self.opt.scored_rows_only) else 0
        x = None
        for i in range(nb):
            q0 = last_q0 if i == nb - 1 else 0
            if i == 0:
                w0 = W.blocks[0]
                g1, b1 = w0["norm1.weight"], w0["norm1.bias"]
                in_proj = None
                if fold_norm1:
                    # self_attn(norm1(x)) = xhat (W*gamma)^T + (b + W beta): the LayerNorm kernels write the normalised
                    # rows (unit weight, zero bias) and the projection takes the folded pair.  What that buys is in the
                    # backward: no product back to the rows and no pass over them for norm1's affine gradients.
                    # (from the f32 parameter where the caller supplies it: one rounding to the compute dtype, not two)
                    in_w0 = w0.get(IN_PROJ_MASTER, w0["self_attn.in_proj_weight"])
                    wf = self.buf("in_proj_w_folded", tuple(in_w0.shape), self.cdt, dev)
                    bf = self.buf("in_proj_b_folded", (in_w0.shape[0],), torch.float32, dev)
                    ops.ln_fold_fwd(in_w0, w0["self_attn.in_proj_bias"], g1, b1, wf, bf)
                    in_proj = (wf, bf)
                    g1, b1 = self._unit_affine(d, dev)
                h1 = self.buf("h1.0", (M, d), self.cdt, dev)
                mean1 = self.buf("mean1.0", (M,), torch.float32, dev)
                rstd1 = self.buf("rstd1.0", (M,), torch.float32, dev)
                concat_src = None
                if source[0] == "concat":
                    # the joined rows are only needed as the residual of the rows block 0 carries on (and by a
                    # LayerNorm backward that must return an input gradient): otherwise they are not kept, and
                    # the backward reads img / sos / cap again
                    x0 = self.buf("x0", (M, d), torch.float32, dev)
                    with self._span("join_ln_fwd"):
                        ops.concat_ln_fwd(img, sos, cap, g1, b1, pos=pos, x_out=x0,
                                          x_row0=0 if keep_joined else q0 * B, y_out=h1, mean=mean1, rstd=rstd1)
                    concat_src = ("concat", img, sos, cap, pos)
                else:
                    # a strided seq-first slice (dataset[:, i:j, :]) is normalised where it lies, without a contiguous
                    # copy; anything the strided kernel cannot address (other dtypes, a non-unit inner stride, rows that
                    # are not 16-byte aligned) is first brought to a contiguous f32 tensor, which it can
                    if not _seqfirst_addressable(tgt):
                        tgt = tgt.float().contiguous()
                        if not _seqfirst_addressable(tgt):       # an f32 contiguous view at an odd offset
                            tgt = tgt.clone()
                    if pos is not None or not keep_joined or tgt.dtype != torch.float32 or not tgt.is_contiguous():
                        x0 = self.buf("x0", (M, d), torch.float32, dev)
                        ops.seqfirst_ln_fwd(tgt, g1, b1, pos=pos, x_out=x0,
                                            x_row0=0 if keep_joined else q0 * B, y_out=h1, mean=mean1, rstd=rstd1)
                        concat_src = ("seqfirst", tgt, pos)
                    else:
                        x0 = tgt.reshape(M, d)                   # already the joined f32 rows: normalise them in place
                        ops.ln_fwd(x0, g1, b1, h1, mean1, rstd1)
                x = self._block_fwd(0, W, x0, B, S, q0, train, step, dev, h1_ready=(h1, mean1, rstd1), in_proj=in_proj)
                self.saved[0]["concat_src"] = concat_src
            else:
                x = self._block_fwd(i, W, x, B, S, q0, train, step, dev)
        self._shape = (B, S, last_q0)
        return x, last_q0 * B, B, S

    # ------------------------------------------------------------------ loss (+ optional backward into grads)
    def loss_and_grads(self, W: _Weights, grads, source, labels, *, train: bool = True, step: int = 0, pos=None,
                       want_grads: bool = True, grad_scale: float = 1.0, on_grad_ready=None, need_input_grad: bool = False):
        """Mean cross-entropy over the B*L scored rows and the argmax-hit count
        (decoder.py:172-185); with want_grads also d(loss*grad_scale)/d(params) written
        into ``grads`` (name -> f32 tensor; block keys via block_key()).
        Returns (loss_sum [1] f32, correct [1] int32, M) as DEVICE tensors: no host sync."""
        geo, cdt = self.geo, self.cdt
        d, V = geo.d_model, geo.vocab_size
        _lib.require_cuda(labels)
        Bl, L = labels.shape
        x, r0, B, S = self.forward_hidden(W, source, L_scored=L, train=train, step=step, pos=pos,
                                          keep_joined=need_input_grad or not want_grads,
                                          fold_norm1=want_grads and not need_input_grad and self.opt.fold_norm1)
        assert Bl == B, "labels batch does not match inputs"
        dev = labels.device
        Mh = L * B
        x_tail = x[x.shape[0] - Mh:]                           # scored rows (l*B + b)
        h = self.buf("head_in", (Mh, d), cdt, dev)
        w_panels = W.head_w_panels if (self.opt.head_k_panels and cdt == torch.bfloat16) else None
        h_panels = None
        if w_panels is not None:
            # the cast of the scored rows also leaves their k-panel image
            h_panels = self.buf("head_in_panels", ops.kpanel_shape(Mh, d), cdt, dev)
            ops.pack_kpanels(x_tail, h_panels, rowmajor=h)
        else:
            ops.cast(x_tail, h) if cdt != torch.float32 else h.copy_(x_tail)
        ws = self.buf("ce_ws", (ops.logits_ce_workspace(Mh, V),), torch.uint8, dev)
        lse = self.buf("ce_lse", (Mh,), torch.float32, dev)
        amax = self.buf("ce_argmax", (Mh,), torch.int32, dev)
        loss_sum = torch.empty(1, dtype=torch.float32, device=dev)
        correct = torch.empty(1, dtype=torch.int32, device=dev)
        # ---- head backward works on d(logits) slabs [Mh, slab] in the compute dtype.  When the whole
        # vocabulary fits the budget there is one slab, and the forward can leave the softmax numerators in
        # it (f32 logits are never stored); otherwise the logits are recomputed slab by slab.
        gscale = grad_scale / Mh
        Vpad = (V + 127) // 128 * 128
        slab = self.opt.vocab_slab
        if slab <= 0:
            # the widest slab the budget allows: whole-vocabulary products fill the chip's tile rounds best
            slab = max(1024, self.opt.dl_budget_bytes // (Mh * (2 if cdt == torch.bfloat16 else 4)) // 128 * 128)
        slab = min((slab + 127) // 128 * 128, Vpad)
        self.last_slab = slab
        from_probs = want_grads and slab >= Vpad and self.opt.keep_softmax_numerators
        self.last_from_probs = from_probs
        dl = self.buf("ce_dl", (Mh, (slab + 127) // 128 * 128), cdt, dev) if want_grads else None
        by_row = from_probs and self.opt.ce_row_reference and cdt == torch.bfloat16 and d % 32 == 0 and d >= 256
        self.last_by_row = by_row
        # d(logits) as a column-panel image [Vp/32][Mh][32] (same bytes, same buffer): the data gradient then reads its
        # 32-deep steps as whole lines.  Only where both backward products run on the pipelined 256x256 kernel anyway and
        # the bias gradient rides on the weight-gradient product (nothing else reads the numerators).
        dl_img = (by_row and self.opt.dl_panels and self.opt.ce_bias_in_wgrad and not self.opt.dgrad_transposed_weights and
                  Mh >= 2048 and Mh % 256 == 0 and dl.shape[1] <= V + W.head_pad_rows)
        self.last_dl_panels = dl_img
        if dl_img:
            dl = dl.view(dl.shape[1] // 32, Mh, 32)
        head_wt = head_wt_ready = None
        if by_row and self.opt.dgrad_transposed_weights:
            # W^T for the data gradient, made beside the forward product (the second stream is idle until the backward)
            kk_t = dl.shape[1] if dl.shape[1] <= V + W.head_pad_rows else V
            wt_shape = (d, (kk_t + 7) // 8 * 8)
            head_wt = self.buf("head_w_t", wt_shape, cdt, dev)
            if self._was_fresh("head_w_t", cdt):
                head_wt[:, V:].zero_()                    # the columns of the padding rows: never written again
            self._on_side_stream(lambda: ops.transpose(W.head_w, head_wt[:, :V]))
            head_wt_ready = torch.cuda.Event()
            head_wt_ready.record(self._side if (self.opt.wgrad_stream and self._side is not None) else torch.cuda.current_stream())
        with self._span("logits_ce_fwd"):
            if by_row:
                rowref = self.buf("ce_rowref", (Mh,), torch.float32, dev)
                ops.logits_ce_fwd_ref(h, W.head_w, W.head_b, labels, B, L, lse, amax, loss_sum, correct, ws, dl, rowref,
                                      h_panels=h_panels, w_panels=w_panels)
            else:
                ops.logits_ce_fwd(h, W.head_w, W.head_b, labels, B, L, lse, amax, loss_sum, correct, ws, probs=dl if from_probs else None,
                                  h_panels=h_panels, w_panels=w_panels)
        if not want_grads:
            return loss_sum, correct, Mh
        dH = self.buf("ce_dH", (Mh, d), torch.float32, dev)
        gW, gb = grads[HEAD_W], grads[HEAD_B]
        bws = self.buf("ce_bwd_ws", (max(ops.logits_ce_bwd_workspace(Mh, slab), ops.colsum_workspace(Mh, V)),), torch.uint8, dev)
        # split-K slabs of the data gradient (8 of them) or, past 8192 scored rows, of the weight gradient with its bias column
        # (one per 8192 rows: the kernel keeps a slab's row factors in LDS) + that column's slabs
        wg_slabs = (Mh + 8191) // 8192
        hws = self.buf("head_splitk_ws", (max(8 * Mh * d * 4, (wg_slabs * V * d * 4 + 32 * V * 4 + 64) if wg_slabs > 1 else 0),), torch.uint8, dev)
        first = True
        if by_row:
            # d(logits) = rowscale[m] * numerators (label column patched): no pass over the [Mh, V] array but the products'
            # own reads and the weighted column sums for the bias
            hs = self.buf("head_in_scaled", (Mh, d), cdt, dev)
            rowscale = self.buf("ce_rowscale", (Mh,), torch.float32, dev)
            rowscale_lp = self.buf("ce_rowscale_lp", (Mh,), cdt, dev)
            with self._span("ce_bwd_prepare"):
                ops.logits_ce_bwd_ref_prepare(dl, V, h, labels, B, L, lse, rowref, gscale, hs, rowscale, rowscale_lp)
            if self.opt.ce_bias_in_wgrad:        # (the kernel keeps the row factors of a workgroup's k range in LDS: past 8192 rows the product is cut along them)
                # the bias gradient is one more column of the weight-gradient product (numerators^T . rowscale): formed by that
                # product's kernel on the way instead of by a pass over the numerators
                with self._span("head_wgrad_slab"):
                    ops.gemm(dl, hs, gW, M=V, N=d, K=Mh, a_kmajor=True, b_kmajor=True, workspace=hws, kvec=rowscale_lp, kvec_out=gb, a_panels=dl_img)
            else:
                # on this stream: the products' workgroups take a CU's whole register file, so a pass running beside them
                # only takes CUs away from them (on a second stream it cost the weight gradient twice its own duration)
                with self._span("ce_bias_grad"):
                    ops.colsum_weighted(dl[:, :V], rowscale, gb, bws)
                with self._span("head_wgrad_slab"):
                    ops.gemm(dl, hs, gW, M=V, N=d, K=Mh, a_kmajor=True, b_kmajor=True, workspace=hws)
            if on_grad_ready is not None:
                on_grad_ready(HEAD_W, 0, V)
            kk = dl.shape[0] * 32 if dl_img else (dl.shape[1] if dl.shape[1] <= V + W.head_pad_rows else V)
            with self._span("head_dgrad_slab"):
                if dl_img:
                    ops.gemm(dl, W.head_w, dH, M=Mh, N=d, K=kk, b_kmajor=True, workspace=hws, row_scale=rowscale, a_panels=True)
                elif head_wt is not None:
                    torch.cuda.current_stream().wait_event(head_wt_ready)
                    ops.gemm(dl, head_wt, dH, M=Mh, N=d, K=kk, workspace=hws, row_scale=rowscale)
                else:
                    ops.gemm(dl, W.head_w, dH, M=Mh, N=d, K=kk, b_kmajor=True, workspace=hws, row_scale=rowscale)
        for v0 in (() if by_row else range(0, V, slab)):
            nv = min(slab, V - v0)
            ld = (nv + 127) // 128 * 128
            dls = dl if ld == dl.shape[1] else dl.view(-1)[:Mh * ld].view(Mh, ld)
            if from_probs:
                with self._span("ce_grad_from_probs"):
                    ops.logits_ce_bwd_from_probs(dls, V, labels, B, L, lse, gscale, ws, gb, bws)
            else:
                with self._span("logits_ce_bwd_slab"):
                    ops.logits_ce_bwd_slab(h, W.head_w, W.head_b, labels, B, L, lse, gscale, v0, nv, dls, gb, bws)
            # the weight gradient first: nothing downstream waits for it but its all-reduce
            with self._span("head_wgrad_slab"):
                ops.gemm(dls, h, gW[v0:v0 + nv], M=nv, N=d, K=Mh, a_kmajor=True, b_kmajor=True, workspace=hws)
            if on_grad_ready is not None:
                on_grad_ready(HEAD_W, v0, nv)
            # the slab's padding columns of dl are zero, and so are the rows that follow the vocabulary matrix
            # in a flat parameter store: the reduction may then run over whole tiles
            kk = ld if v0 + ld <= V + W.head_pad_rows else nv
            with self._span("head_dgrad_slab"):
                ops.gemm(dls, W.head_w[v0:], dH, M=Mh, N=d, K=kk, b_kmajor=True, accumulate=not first, workspace=hws)
            first = False
        if on_grad_ready is not None:
            on_grad_ready(HEAD_B, 0, V)
        # ---- blocks backward
        nb = geo.n_blocks
        rows = x.shape[0]
        if rows == Mh:
            dx = dH
        else:
            dx = self.buf("dx_last", (rows, d), torch.float32, dev, zero=True)
            dx[rows - Mh:].copy_(dH)
        dx_input = None
        g2_done = False
        for i in reversed(range(nb)):
            need_dx_in = i > 0 or need_input_grad
            dx, g2_done = self._block_bwd(i, W, grads, dx, B, S, dev, need_dx_in, g2_done)
            if on_grad_ready is not None:
                on_grad_ready(("block", i), 0, 0)
        if need_input_grad:
            dx_input = dx
        self.input_grad = dx_input
        return loss_sum, correct, Mh

    # ------------------------------------------------------------------ one block, backward
    def _block_bwd(self, i, W, grads, dx_out, B, S, dev, need_dx_in, g2_done=False):
        """Returns (dx of the block's input or None, whether the block below already has its first gradient:
        the LayerNorm backward that ends this block then also wrote g_res.{(i-1) % 2} and its linear2.bias gradient)."""
        geo, cdt = self.geo, self.cdt
        d, ff, H, hd = geo.d_model, geo.dim_feedforward, geo.nhead, geo.hd
        nbk = geo.n_blocks
        w, sv = W.blocks[i], self.saved[i]
        q_start = sv["q_start"]
        M, r0 = S * B, q_start * B
        Mt = M - r0
        k_attn, k_r1, k_h, k_r2 = sv["keys"]
        G = lambda name: grads[block_key(i, name, nbk)]           # noqa: E731
        skws = self.buf("splitk_ws", (32 * max(3 * d * d, d * ff) * 4,), torch.uint8, dev)     # up to 32 f32 slabs of the largest weight
        csws = self.buf("colsum_ws", (max(ops.colsum_workspace(M, 3 * d), ops.colsum_workspace(Mt, 3 * d)),), torch.uint8, dev)
        G("linear1.bias").zero_()            # summed by atomics in a GEMM epilogue; the other bias gradients are overwritten
        # ---- MLP
        # two buffers in turn: the block above fills the next one while this block's side-stream products may still read theirs
        g2 = self.buf(f"g_res.{i % 2}", (Mt, d), cdt, dev)
        if not g2_done:
            with self._span("block_rowops_bwd"):
                ops.dropout_bwd_cast(dx_out, k_r2, g2, G("linear2.bias"), csws)
        self._on_side_stream(lambda: ops.gemm(g2, sv["a1"], G("linear2.weight"), M=d, N=ff, K=Mt, a_kmajor=True, b_kmajor=True, workspace=skws))
        dz1 = self.buf("dz1", (Mt, ff), cdt, dev)
        gcws = self.buf("gate_colsum_ws", ((Mt + 63) // 64 * ff * 4,), torch.uint8, dev)     # one partial row of bias sums per 64 rows
        tw = self.opt.dgrad_transposed_weights and cdt == torch.bfloat16
        w2, w2_km = w["linear2.weight"], True
        if tw:
            w2 = ops.transpose(w["linear2.weight"], self.buf("linear2_w_t", (ff, d), cdt, dev))
            w2_km = False
        with self._span("block_mlp_dgrad"):
            if self.opt.activation == "relu":
                ops.gemm(g2, w2, dz1, M=Mt, N=ff, K=d, b_kmajor=w2_km, gate=sv["a1"], gate_mode=GATE_RELU_OUT,
                         gate_drop=k_h, colsum=G("linear1.bias"), workspace=gcws)
            else:
                ops.gemm(g2, w2, dz1, M=Mt, N=ff, K=d, b_kmajor=w2_km, gate=sv["z1"], gate_mode=GATE_GELU_PRE,
                         gate_drop=k_h, colsum=G("linear1.bias"), workspace=gcws)
        self._on_side_stream(lambda: ops.gemm(dz1, sv["h2"], G("linear1.weight"), M=ff, N=d, K=Mt, a_kmajor=True, b_kmajor=True, workspace=skws))
        dh2 = self.buf("dh", (Mt, d), cdt, dev)
        with self._span("block_mlp_dgrad"):
            ops.gemm(dz1, w["linear1.weight"], dh2, M=Mt, N=d, K=ff, b_kmajor=True)
        dx_mid = self.buf(f"dx_mid", (Mt, d), torch.float32, dev)
        lnws = self.buf("ln_ws", (max(ops.ln_bwd_workspace(M, d), ops.ln_bwd_dropout_workspace(M, d)),), torch.uint8, dev)
        # ---- attention: the gradient entering out_proj (dropout backward of dx_mid, with out_proj's bias gradient)
        # leaves the LayerNorm backward's pass over the rows
        g1 = self.buf("g_res1", (Mt, d), cdt, dev)        # not g2's buffer: a side-stream product may still read that
        with self._span("block_ln_bwd"):
            ops.ln_bwd_dropout(dh2, sv["x_mid"], w["norm2.weight"], sv["mean2"], sv["rstd2"], dx_out, dx_mid, G("norm2.weight"), G("norm2.bias"),
                               k_r1, g1, G("self_attn.out_proj.bias"), lnws)
        self._on_side_stream(lambda: ops.gemm(g1, sv["attn_o"][r0:], G("self_attn.out_proj.weight"), M=d, N=d, K=Mt, a_kmajor=True,
                                              b_kmajor=True, workspace=skws))
        dattn = self.buf("dattn", (M, d), cdt, dev)
        with self._span("block_out_proj_dgrad"):
            if tw:
                wot = ops.transpose(w["self_attn.out_proj.weight"], self.buf("out_proj_w_t", (d, d), cdt, dev))
                ops.gemm(g1, wot, dattn[r0:], M=Mt, N=d, K=d)
            else:
                ops.gemm(g1, w["self_attn.out_proj.weight"], dattn[r0:], M=Mt, N=d, K=d, b_kmajor=True)
        dqkv = self.buf("dqkv", (M, 3 * d), cdt, dev)
        delta = self.buf("delta", (B * H * S,), torch.float32, dev)
        with self._span("block_attn_bwd"):
            ops.attn_bwd_packed(sv["qkv"], sv["attn_o"], sv["lse"], dattn, dqkv, delta, B, H, S, hd, q_start=q_start, nq=S - q_start,
                                mask_mode=_MASKS[self.opt.mask_mode], prefix_len=self.prefix_len, drop=k_attn, dq_scale=hd ** -0.5)
        gin_w, gin_b = G("self_attn.in_proj_weight"), G("self_attn.in_proj_bias")
        h1 = sv["h1"]
        # in_proj's bias gradient = column sums of dQKV = dQKV^T . 1: one more column of its weight-gradient product, formed by
        # that product's kernel (bf16; in f32 a pass of its own)
        lp = cdt == torch.bfloat16
        csws_side = self.buf("colsum_ws_side", (csws.numel(),), torch.uint8, dev)

        def wgrad_q(ws):
            ops.gemm(dqkv[r0:, :d], h1[r0:], gin_w[:d], M=d, N=d, K=Mt, a_kmajor=True, b_kmajor=True, workspace=ws,
                     kvec_out=gin_b[:d] if lp else None)

        def wgrad_kv(ws):
            ops.gemm(dqkv[:, d:], h1, gin_w[d:], M=2 * d, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws,
                     kvec_out=gin_b[d:] if lp else None)

        def wgrad_all(ws):
            ops.gemm(dqkv, h1, gin_w, M=3 * d, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=ws, kvec_out=gin_b if lp else None)

        def bias_grad_f32(ws):
            if lp:
                return
            if q_start == 0:
                ops.colsum(dqkv, gin_b, ws)
            else:
                ops.colsum(dqkv[r0:, :d], gin_b[:d], ws)
                ops.colsum(dqkv[:, d:], gin_b[d:], ws)

        split_streams = bool(sv.get("folded"))
        if split_streams:
            # nothing is left for the main stream but to wait for these gradients: it takes the query rows' product (with a
            # split-K workspace of its own) while the side stream runs the large one
            if q_start == 0:
                self._on_side_stream(lambda: wgrad_all(skws))
                bias_grad_f32(csws)
            else:
                self._on_side_stream(lambda: wgrad_kv(skws))
                skws_q = self.buf("splitk_ws_q", (33 * d * d * 4,), torch.uint8, dev)
                bias_grad_f32(csws)
                wgrad_q(skws_q)
        elif q_start == 0:
            def in_proj_grads():
                bias_grad_f32(csws_side)
                wgrad_all(skws)
        else:
            def in_proj_grads():
                bias_grad_f32(csws_side)
                wgrad_q(skws)
                wgrad_kv(skws)
        if not split_streams:
            self._on_side_stream(in_proj_grads)
        # dh1 is needed even when the block input is data: LayerNorm-1's affine gradients come from it
        in_w = w["self_attn.in_proj_weight"]
        if sv.get("folded"):
            # norm1's affine was folded into in_proj: gin_w holds G = dQKV^T xhat and gin_b the column sums of dQKV,
            # from which the weight gradient proper and norm1's affine gradients follow without touching the rows again
            assert not need_dx_in
            self._join_side_stream()
            fws = self.buf("ln_fold_ws", (ops.ln_fold_bwd_workspace(3 * d, d),), torch.uint8, dev)
            ops.ln_fold_bwd(gin_w, gin_b, in_w, w["norm1.weight"], w["norm1.bias"], G("norm1.weight"), G("norm1.bias"), fws)
            return None, False
        if not need_dx_in:
            # no gradient flows below this block: dh1 only feeds norm1's affine gradients, so it is kept in
            # the compute dtype, rows below / from r0 come from two products over disjoint rows, and the
            # LayerNorm backward skips dx (reading the joined rows from their sources when they were not kept)
            dh1 = self.buf("dh1c", (M, d), cdt, dev)
            if q_start == 0:
                ops.gemm(dqkv, in_w, dh1, M=M, N=d, K=3 * d, b_kmajor=True)
            else:
                ops.gemm(dqkv[:r0, d:], in_w[d:], dh1[:r0], M=r0, N=d, K=2 * d, b_kmajor=True)
                ops.gemm(dqkv[r0:], in_w, dh1[r0:], M=Mt, N=d, K=3 * d, b_kmajor=True)
            src = sv.get("concat_src")
            if src is not None and src[0] == "concat":
                ops.concat_ln_bwd_affine(dh1, src[1], src[2], src[3], w["norm1.weight"], sv["mean1"], sv["rstd1"],
                                         G("norm1.weight"), G("norm1.bias"), lnws, pos=src[4])
            elif src is not None:
                ops.seqfirst_ln_bwd_affine(dh1, src[1], w["norm1.weight"], sv["mean1"], sv["rstd1"],
                                           G("norm1.weight"), G("norm1.bias"), lnws, pos=src[2])
            else:
                ops.ln_bwd(dh1, sv["x_in"], w["norm1.weight"], sv["mean1"], sv["rstd1"], None, None,
                           G("norm1.weight"), G("norm1.bias"), lnws)
            self._join_side_stream()
            return None, False
        with self._span("block_qkv_dgrad"):
            if q_start == 0:
                # one product: like dh2 it reaches the LayerNorm backward in the compute dtype
                dh1 = self.buf("dh1c", (M, d), cdt, dev)
                ops.gemm(dqkv, in_w, dh1, M=M, N=d, K=3 * d, b_kmajor=True)
            else:
                dh1 = self.buf("dh1", (M, d), torch.float32, dev)     # summed from two products over different row ranges
                ops.gemm(dqkv[:, d:], in_w[d:], dh1, M=M, N=d, K=2 * d, b_kmajor=True)
                ops.gemm(dqkv[r0:, :d], in_w[:d], dh1[r0:], M=Mt, N=d, K=d, b_kmajor=True, accumulate=True)
        if q_start == 0:
            dres = dx_mid
        else:
            dres = self.buf("dres", (M, d), torch.float32, dev, zero=True)
            dres[r0:].copy_(dx_mid)
        dx_in = self.buf(f"dx_in.{i % 2}", (M, d), torch.float32, dev)
        # the block below starts from the dropout backward of dx_in over the same rows: take it from this pass
        below = i > 0 and self.saved[i - 1]["q_start"] == 0
        with self._span("block_ln_bwd"):
            if below:
                g_below = self.buf(f"g_res.{(i - 1) % 2}", (M, d), cdt, dev)
                ops.ln_bwd_dropout(dh1, sv["x_in"], w["norm1.weight"], sv["mean1"], sv["rstd1"], dres, dx_in, G("norm1.weight"), G("norm1.bias"),
                                   self.saved[i - 1]["keys"][3], g_below, grads[block_key(i - 1, "linear2.bias", nbk)], lnws)
            else:
                ops.ln_bwd(dh1, sv["x_in"], w["norm1.weight"], sv["mean1"], sv["rstd1"], dres, dx_in, G("norm1.weight"), G("norm1.bias"), lnws)
        self._join_side_stream()
        return dx_in, below

    # ------------------------------------------------------------------ all-rows logits (the reference's forward)
    def forward_logits(self, W: _Weights, source, *, train: bool, step: int = 0, pos=None) -> torch.Tensor:
        """decoder.py:40-68: logits [S,B,V] f32 for ALL rows (compatibility path)."""
        x, r0, B, S = self.forward_hidden(W, source, L_scored=None, train=train, step=step, pos=pos)
        dev = x.device
        d, V = self.geo.d_model, self.geo.vocab_size
        h = self.buf("all_in", (S * B, d), self.cdt, dev)
        ops.cast(x, h) if self.cdt != torch.float32 else h.copy_(x)
        out = torch.empty(S * B, V, dtype=torch.float32, device=dev)
        ops.gemm(h, W.head_w, out, bias=W.head_b)
        self._all_in = h
        return out.view(S, B, V)

    def backward_logits(self, W: _Weights, grads, dlogits: torch.Tensor, need_input_grad: bool = False):
        """Backward of forward_logits given d(loss)/d(logits) [S,B,V]."""
        geo, cdt = self.geo, self.cdt
        d, V = geo.d_model, geo.vocab_size
        B, S, _ = self._shape
        dev = dlogits.device
        M = S * B
        Vp = (V + 7) // 8 * 8
        dl = self.buf("dl_all", (M, Vp), cdt, dev, zero=True)
        dl[:, :V].copy_(dlogits.reshape(M, V))
        h = self._all_in
        ops.colsum(dl[:, :V], grads[HEAD_B])
        ops.gemm(dl, h, grads[HEAD_W], M=V, N=d, K=M, a_kmajor=True, b_kmajor=True)
        dx = self.buf("dx_all", (M, d), torch.float32, dev)
        ops.gemm(dl, W.head_w, dx, M=M, N=d, K=V, b_kmajor=True)
        for i in reversed(range(geo.n_blocks)):
            dx = self._block_bwd(i, W, grads, dx, B, S, dev, i > 0 or need_input_grad)
        self.input_grad = dx if need_input_grad else None

    # ------------------------------------------------------------------ KV-cached greedy decoding
    @torch.no_grad()
    def generate(self, W: _Weights, img, sos_slot, emb, *, max_len: int, sos_id: int, eos_id: int, pad_id: int,
                 pos=None, return_margins: bool = False, stop_check_every: int = 0):
        """Greedy captions for a batch of image prefixes.  seq0 = [img | sos_slot | E[sos]];
        step t scores the last row, appends E[argmax].  K/V of every row are computed
        once and cached ([B,H,2,S,hd] per block).  Returns ids [B,max_len] int64."""
        geo, cdt = self.geo, self.cdt
        d, ff, H, hd, V, nb = geo.d_model, geo.dim_feedforward, geo.nhead, geo.hd, geo.vocab_size, geo.n_blocks
        _lib.require_cuda(img, sos_slot, emb)
        if nb > 1 and self.opt.mask_mode != "prefix_causal":
            raise ValueError("KV-cached decoding with stacked blocks is only exact under mask_mode='prefix_causal'")
        B, P, _ = img.shape
        dev = img.device
        S0, smax = P + 2, P + 1 + max_len
        mask = _MASKS[self.opt.mask_mode]
        scale = hd ** -0.5
        caches = [self.buf(f"kv.{i}", (B, H, 2, smax, hd), cdt, dev) for i in range(nb)]
        ids = torch.full((B,), sos_id, dtype=torch.int64, device=dev)
        first = self.buf("gen_first", (B, 1, d), torch.float32, dev)
        ops.embed_rows(emb, ids, first.view(B, d))
        # ---- prefill: all P+2 rows through every block (last block: only the last row's query)
        M0 = S0 * B
        x = self.buf("gen_x0", (M0, d), torch.float32, dev)
        h1 = self.buf("gen_h1", (M0, d), cdt, dev)
        w0 = W.blocks[0]
        ops.concat_ln_fwd(img, sos_slot, first.to(img.dtype), w0["norm1.weight"], w0["norm1.bias"],
                          pos=pos[:S0].contiguous() if pos is not None else None, x_out=x, y_out=h1)
        kvs = (H * 2 * smax * hd, 2 * smax * hd, hd)
        for i in range(nb):
            w = W.blocks[i]
            last = i == nb - 1
            q0 = S0 - 1 if last else 0
            r0 = q0 * B
            if i > 0:
                ops.ln_fwd(x, w["norm1.weight"], w["norm1.bias"], h1)
            q = self.buf("gen_q", (M0, d), cdt, dev)
            # K/V of all rows straight into the cache; q rows only where needed
            in_w, in_b = w["self_attn.in_proj_weight"], w["self_attn.in_proj_bias"]
            if q0 == 0:
                ops.gemm(h1, in_w, q, bias=in_b, col_scale=scale, col_scale_n=d, kv_cache=caches[i], kv_geom=(d, d, B, H, hd, smax, 0))
            else:
                ops.gemm(h1, in_w[d:], q, bias=in_b[d:], kv_cache=caches[i], kv_geom=(0, d, B, H, hd, smax, 0))
                ops.gemm(h1[r0:], in_w[:d], q[r0:], bias=in_b[:d], col_scale=scale, col_scale_n=d)
            o = self.buf("gen_o", (M0, d), cdt, dev)
            desc_strides = ((d, hd, B * d), kvs, kvs, (d, hd, B * d))
            self._attn_cache_fwd(q, caches[i], o, desc_strides, B, H, S0, hd, smax, q0, S0 - q0, mask, P)
            Mt = M0 - r0
            x_mid = self.buf("gen_xmid", (Mt, d), torch.float32, dev)
            ops.gemm(o[r0:], w["self_attn.out_proj.weight"], x_mid, bias=w["self_attn.out_proj.bias"], residual=x[r0:])
            h2 = self.buf("gen_h2", (Mt, d), cdt, dev)
            ops.ln_fwd(x_mid, w["norm2.weight"], w["norm2.bias"], h2)
            a1 = self.buf("gen_a1", (Mt, ff), cdt, dev)
            ops.gemm(h2, w["linear1.weight"], a1, bias=w["linear1.bias"], act=_ACTS[self.opt.activation])
            x_next = self.buf(f"gen_xout.{i % 2}", (Mt, d), torch.float32, dev)
            ops.gemm(a1, w["linear2.weight"], x_next, bias=w["linear2.bias"], residual=x_mid)
            x = x_next
        x_last = x[x.shape[0] - B:]                                 # [B,d] hidden of the newest row
        # ---- decode loop
        out = torch.full((B, max_len), pad_id, dtype=torch.int64, device=dev)
        margins = torch.zeros(B, max_len, dtype=torch.float32, device=dev) if return_margins else None
        done = torch.zeros(B, dtype=torch.uint8, device=dev)
        if emb.dtype != torch.float32 or not emb.is_contiguous():
            raise ValueError("the token-embedding table must be a contiguous float32 [vocab, d_model] tensor")
        w_panels = W.head_w_panels if (self.opt.head_k_panels and cdt == torch.bfloat16) else None
        if w_panels is None and self.opt.head_k_panels and cdt == torch.bfloat16 and d % 32 == 0 and W.head_w.stride(1) == 1:
            # the weights stay as they are for the whole loop: one image serves every token
            w_panels = ops.pack_kpanels(W.head_w, self.buf("gen_head_panels", ops.kpanel_shape(V, d), cdt, dev))
        if self.timer is not None or not self.opt.composed_decode_step:
            # per-kernel event brackets are wanted: issue the step's launches one by one (the same kernels,
            # in the same order, as capdec_decode_step)
            self._decode_loop_by_op(W, caches, x_last, emb, pos, ids, done, out, margins, max_len, S0, smax, eos_id, pad_id,
                                    stop_check_every, w_panels)
            return (out, margins) if return_margins else out
        mg = self.buf("gen_margin", (B,), torch.float32, dev)
        # the residual stream of the newest row lives in one buffer that every step reads and rewrites
        xl = self.buf("gen_xlast", (B, d), torch.float32, dev)
        xl.copy_(x_last)
        if pos is not None:
            pos = pos.contiguous()
            if pos.shape[0] < smax:
                raise ValueError(f"pos has {pos.shape[0]} rows, decoding needs {smax}")
        import ctypes as C
        lib = _lib.load()
        wsb = lib.capdec_decode_step_workspace(B, d, ff, V, _lib.dtype_code(caches[0]))
        ws = self.buf("gen_step_ws", (wsb,), torch.uint8, dev)
        blocks = (_lib.BlockParams * nb)()
        for i in range(nb):
            w = W.blocks[i]
            for field, key in (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
                               ("in_proj_w", "self_attn.in_proj_weight"), ("in_proj_b", "self_attn.in_proj_bias"),
                               ("out_proj_w", "self_attn.out_proj.weight"), ("out_proj_b", "self_attn.out_proj.bias"),
                               ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                               ("linear1_w", "linear1.weight"), ("linear1_b", "linear1.bias"),
                               ("linear2_w", "linear2.weight"), ("linear2_b", "linear2.bias")):
                setattr(blocks[i], field, w[key].data_ptr())
            blocks[i].kv_cache = caches[i].data_ptr()
        dd = _lib.DecodeDesc()
        dd.B, dd.d_model, dd.nhead, dd.dim_ff, dd.vocab, dd.n_blocks, dd.smax = B, d, H, ff, V, nb, smax
        dd.dtype, dd.activation, dd.ln_eps = _lib.dtype_code(caches[0]), _ACTS[self.opt.activation], 1e-5
        dd.blocks = blocks
        dd.head_w, dd.head_b, dd.ld_head_w = W.head_w.data_ptr(), W.head_b.data_ptr(), W.head_w.stride(0)
        dd.emb, dd.pos = emb.data_ptr(), (pos.data_ptr() if pos is not None else None)
        dd.x_last, dd.ids, dd.done, dd.out = xl.data_ptr(), ids.data_ptr(), done.data_ptr(), out.data_ptr()
        dd.margin = mg.data_ptr() if return_margins else None       # unasked for, the head runs without the second maximum
        dd.max_len, dd.eos_id, dd.pad_id = max_len, eos_id, pad_id
        dd.workspace, dd.workspace_bytes = ws.data_ptr(), wsb
        if w_panels is not None:
            dd.head_w_panels, dd.head_w_panel_rows = w_panels.data_ptr(), w_panels.shape[1]
        stream = _lib.stream_ptr()
        for t in range(max_len):
            dd.t, dd.s_new = t, S0 + t                               # s_new: index of the row appended after token t
            with self._span("decode_step"):
                _lib.check(lib.capdec_decode_step(C.byref(dd), stream), "capdec_decode_step")
            if return_margins:
                margins[:, t].copy_(mg)
            # <eos> early stop: the remaining columns already hold <pad>.  Looking at `done` is a host
            # synchronisation, so it is only done every stop_check_every steps (0: never).
            if stop_check_every > 0 and t + 1 < max_len and (t + 1) % stop_check_every == 0 and bool(done.all().item()):
                break
        return (out, margins) if return_margins else out

    def _decode_loop_by_op(self, W, caches, x_last, emb, pos, ids, done, out, margins, max_len, S0, smax, eos_id, pad_id,
                           stop_check_every, w_panels=None):
        geo, cdt = self.geo, self.cdt
        d, ff, H, hd, V, nb = geo.d_model, geo.dim_feedforward, geo.nhead, geo.hd, geo.vocab_size, geo.n_blocks
        B, dev, scale = ids.shape[0], ids.device, geo.hd ** -0.5
        return_margins = margins is not None
        ws = self.buf("gen_ws", (ops.logits_ce_workspace(B, V),), torch.uint8, dev)
        hh = self.buf("gen_hh", (B, d), cdt, dev)
        mg = self.buf("gen_margin", (B,), torch.float32, dev)
        xs = self.buf("gen_xs", (B, d), torch.float32, dev)
        hs = self.buf("gen_hs", (B, d), cdt, dev)
        qs = self.buf("gen_qs", (B, d), cdt, dev)
        os_ = self.buf("gen_os", (B, d), cdt, dev)
        xm = self.buf("gen_xm", (B, d), torch.float32, dev)
        h2s = self.buf("gen_h2s", (B, d), cdt, dev)
        a1s = self.buf("gen_a1s", (B, ff), cdt, dev)
        xo = [self.buf(f"gen_xo{j}", (B, d), torch.float32, dev) for j in range(2)]
        for t in range(max_len):
            hp = None
            if w_panels is not None:
                hp = ops.pack_kpanels(x_last, self.buf("gen_hh_panels", ops.kpanel_shape(B, d), cdt, dev), rowmajor=hh)
            else:
                ops.cast(x_last, hh) if cdt != torch.float32 else hh.copy_(x_last)
            with self._span("logits_argmax"):
                ops.logits_argmax(hh, W.head_w, W.head_b, ids, mg if return_margins else None, ws, h_panels=hp, w_panels=w_panels)
            if return_margins:
                margins[:, t].copy_(mg)
            ops.greedy_update(ids, done, out, t, eos_id, pad_id)
            if t + 1 == max_len:
                break
            # <eos> early stop: the remaining columns already hold <pad>.  Looking at `done` is a host
            # synchronisation, so it is only done every stop_check_every steps (0: never).
            if stop_check_every > 0 and (t + 1) % stop_check_every == 0 and bool(done.all().item()):
                break
            s_new = S0 + t                                           # index of the appended row
            ops.embed_rows(emb, ids, xs, pos_row=pos[s_new].contiguous() if pos is not None else None)
            xcur = xs
            for i in range(nb):
                w = W.blocks[i]
                ops.ln_fwd(xcur, w["norm1.weight"], w["norm1.bias"], hs)
                ops.gemm(hs, w["self_attn.in_proj_weight"], qs, bias=w["self_attn.in_proj_bias"], col_scale=scale, col_scale_n=d,
                         kv_cache=caches[i], kv_geom=(d, d, B, H, hd, smax, s_new))
                with self._span("attn_decode"):
                    ops.attn_decode(qs, caches[i], os_, B, H, hd, smax, s_new + 1)
                ops.gemm(os_, w["self_attn.out_proj.weight"], xm, bias=w["self_attn.out_proj.bias"], residual=xcur)
                ops.ln_fwd(xm, w["norm2.weight"], w["norm2.bias"], h2s)
                ops.gemm(h2s, w["linear1.weight"], a1s, bias=w["linear1.bias"], act=_ACTS[self.opt.activation])
                ops.gemm(a1s, w["linear2.weight"], xo[i % 2], bias=w["linear2.bias"], residual=xm)
                xcur = xo[i % 2]
            x_last = xcur

    def _attn_cache_fwd(self, q, cache, o, strides, B, H, S, hd, smax, q_start, nq, mask, prefix):
        import ctypes as C
        from ._lib import AttnDesc, check, ptr, stream_ptr
        desc = AttnDesc()
        kbase = cache
        vbase = cache.view(-1)[smax * hd:]
        ops._fill_attn(desc, q, kbase, vbase, o, strides, B, H, S, hd, q_start, nq, mask, prefix, _lib.NO_DROPOUT, None)
        check(_lib.load().capdec_attn_fwd(C.byref(desc), stream_ptr()), "capdec_attn_fwd")
