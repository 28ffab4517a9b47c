"""Drop-in ``TransformerDecoder`` and ``train_decoder`` running on libcapdec.

Mirrors the reference's decoder.py interface:
  * ``TransformerDecoder(d_model, nhead, num_layers, dim_feedforward, dropout,
    vocab_size)`` with the same sub-module attributes and the same 14 state_dict
    keys (decoder.py:11-29), so ``load_state_dict`` round-trips with the reference
    and, under the same seed, the initial weights are identical;
  * ``forward(tgt, memory) -> [S,B,vocab]`` (decoder.py:40-68; ``memory`` ignored);
  * ``create_caption_mask(seq_len)`` (decoder.py:31-38);
  * ``train_decoder(processed_data, config)`` (decoder.py:96-216).
On top of that: ``caption_loss`` (scored rows only, logits never materialised),
``generate`` (KV-cached greedy decoding) and ``CaptionTrainer`` (fused
forward/backward/Adam, optional RCCL data parallelism).

Tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .config import config
from .engine import (BLOCK_PARAM_SHAPES, HEAD_B, HEAD_W, IN_PROJ_MASTER, MATRIX_KEYS, DecoderEngine, EngineOptions, ModelGeometry,
                     ParamStore, _Weights, block_key, weights_from_store)


class _BlockParams(nn.Module):
    """Parameter container of one extra block (same sub-module names as block 0)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class TransformerDecoder(nn.Module):
    def __init__(self, d_model=768, nhead=8, num_layers=6, dim_feedforward=2048, dropout=0.1, vocab_size=50260,
                 *, n_blocks: int = 1, options: Optional[EngineOptions] = None):
        super().__init__()
        # `num_layers` is accepted and unused, exactly like the reference (decoder.py:12-29
        # builds one block whatever it is given); stacking is requested with n_blocks.
        self.d_model = d_model
        self.nhead = nhead
        self.num_layers = num_layers
        self.dim_feedforward = dim_feedforward
        self.vocab_size = vocab_size
        self.n_blocks = n_blocks
        # same constructors in the same order as decoder.py:17-29 => same RNG stream => same init
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.output_projection = nn.Linear(d_model, vocab_size)
        if n_blocks > 1:
            self.blocks = nn.ModuleDict({str(i): _BlockParams(d_model, nhead, dim_feedforward, dropout) for i in range(1, n_blocks)})
        self.options = options or EngineOptions()
        self._engine: Optional[DecoderEngine] = None
        self._shadow: Dict[str, tuple] = {}
        self._step = 0
        self.prefix_len = 0          # needed by mask_mode="prefix_causal" when forward() gets a joined tensor

    # ------------------------------------------------------------------ plumbing
    @property
    def geometry(self) -> ModelGeometry:
        return ModelGeometry(self.d_model, self.nhead, self.dim_feedforward, self.vocab_size, self.n_blocks, float(self.dropout.p))

    @property
    def engine(self) -> DecoderEngine:
        if self._engine is None:
            self._engine = DecoderEngine(self.geometry, self.options)
        self._engine.geo.dropout = float(self.dropout.p)
        return self._engine

    def _named(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())

    def _weights(self) -> _Weights:
        """Kernel-ready views: matrices cast to the compute dtype (re-cast only when the
        parameter changed), vectors used as they are (f32)."""
        named = self._named()
        cdt = self.options.compute_dtype
        from ._lib import require_cuda
        require_cuda(*named.values())

        def mat(key):
            p = named[key]
            if p.dtype != torch.float32:
                raise TypeError("parameters must be float32 (the compute dtype is an engine option)")
            if cdt == torch.float32:
                return p.detach()
            ver, sh = self._shadow.get(key, (None, None))
            if sh is None or ver != (p._version, p.data_ptr()) or sh.device != p.device:
                sh = torch.empty(p.shape, dtype=cdt, device=p.device) if sh is None or sh.device != p.device else sh
                ops.cast(p.detach().contiguous(), sh)
                self._shadow[key] = ((p._version, p.data_ptr()), sh)
            return sh

        blocks = []
        for i in range(self.n_blocks):
            b = {}
            for name, _ in BLOCK_PARAM_SHAPES:
                key = block_key(i, name, self.n_blocks)
                b[name] = mat(key) if name in MATRIX_KEYS else named[key].detach()
            if i == 0:
                b[IN_PROJ_MASTER] = named[block_key(0, "self_attn.in_proj_weight", self.n_blocks)].detach()
            blocks.append(b)
        head_w = mat(HEAD_W)
        panels = None
        if cdt == torch.bfloat16 and self.options.head_k_panels and head_w.shape[1] % 32 == 0:
            # the k-panel image of the vocabulary matrix is remade when its shadow is
            p = named[HEAD_W]
            ver, panels = self._shadow.get("head.k_panels", (None, None))
            if panels is None or ver != (p._version, p.data_ptr()) or panels.device != p.device:
                panels = ops.pack_kpanels(head_w, torch.empty(ops.kpanel_shape(*head_w.shape), dtype=cdt, device=p.device))
                self._shadow["head.k_panels"] = ((p._version, p.data_ptr()), panels)
        return _Weights(blocks, head_w, named[HEAD_B].detach(), head_w_panels=panels)

    def _param_keys(self):
        keys = [HEAD_W, HEAD_B]
        for i in range(self.n_blocks):
            keys += [block_key(i, name, self.n_blocks) for name, _ in BLOCK_PARAM_SHAPES]
        return keys

    # ------------------------------------------------------------------ reference interface
    def create_caption_mask(self, seq_len):
        """Additive float mask [S,S] (decoder.py:31-38).  In the default "reference"
        mode: -inf everywhere, columns below 16*768 opened, strict upper triangle kept
        -- i.e. all zeros for any seq_len <= 12288."""
        dev = self.norm1.weight.device
        rows = torch.arange(seq_len, device=dev).unsqueeze(1)
        cols = torch.arange(seq_len, device=dev).unsqueeze(0)
        mode = self.options.mask_mode
        if mode == "reference":
            hidden = (cols >= 16 * 768) & (cols > rows)
        elif mode == "prefix_causal":
            hidden = (cols >= self.prefix_len) & (cols > rows)
        else:
            hidden = torch.zeros(seq_len, seq_len, dtype=torch.bool, device=dev)
        return torch.zeros(seq_len, seq_len, device=dev).masked_fill(hidden, float("-inf"))

    def forward(self, tgt, memory=None):
        """tgt [S,B,d_model] seq-first -> logits [S,B,vocab] for every row
        (decoder.py:40-68).  ``memory`` is accepted and unused, as in the reference."""
        keys = self._param_keys()
        named = self._named()
        return _LogitsFn.apply(self, tgt, *[named[k] for k in keys])

    # ------------------------------------------------------------------ fused paths
    def caption_loss(self, inputs, labels, pos: Optional[torch.Tensor] = None):
        """Mean cross-entropy over the last L rows and the argmax-hit count
        (decoder.py:169-185) without materialising logits.  ``inputs`` is either the
        joined tensor [S,B,d] or a tuple (img [B,P,d], sos [B,1,d], cap [B,L,d]) which is
        joined on the fly (img_text_concat.py:115-123).  Returns (loss, correct) as
        0-dim device tensors; ``loss`` is differentiable w.r.t. the parameters."""
        keys = self._param_keys()
        named = self._named()
        params = [named[k] for k in keys]
        # grad mode is always off inside Function.forward, so decide here
        want = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _CaptionLossFn.apply(self, inputs, labels, (pos, want), *params)

    @torch.no_grad()
    def generate(self, img, sos_slot, token_embedding, *, max_len=None, sos_id=None, eos_id=None, pad_id=None, pos=None,
                 return_margins=False, stop_check_every=0):
        """Greedy captions [B,max_len] for image prefixes img [B,P,d].  The special ids
        default to the reference tokenizer's add order (img_text_concat.py:22-26):
        <sos>=V-3, <eos>=V-2, <pad>=V-1.  A sequence that has produced <eos> emits <pad> from
        then on; with stop_check_every=k the loop also ends once every sequence has (checked
        every k tokens, one host synchronisation each)."""
        V = self.vocab_size
        max_len = config.max_caption_length if max_len is None else max_len
        sos_id = V - 3 if sos_id is None else sos_id
        eos_id = V - 2 if eos_id is None else eos_id
        pad_id = V - 1 if pad_id is None else pad_id
        return self.engine.generate(self._weights(), img, sos_slot, token_embedding, max_len=max_len, sos_id=sos_id,
                                    eos_id=eos_id, pad_id=pad_id, pos=pos, return_margins=return_margins,
                                    stop_check_every=stop_check_every)

    def _source(self, inputs):
        if isinstance(inputs, (tuple, list)):
            img, sos, cap = inputs
            return ("concat", img, sos, cap)
        return ("tensor", inputs, self.prefix_len)


def _grad_views(model: TransformerDecoder, keys, device):
    """Fresh gradient tensors for ONE autograd call.  They are owned by that call's graph node, not by the engine:
    a second forward before the first backward (two losses summed, an evaluation in between) cannot overwrite them."""
    named = model._named()
    return {k: torch.empty(named[k].shape, dtype=torch.float32, device=device) for k in keys}


class _LogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: TransformerDecoder, tgt, *params):
        from ._lib import require_cuda
        require_cuda(tgt)
        eng = model.engine
        train = model.training
        model._step += 1
        W = model._weights()
        ctx.model, ctx.W = model, W
        ctx.need_input = tgt.requires_grad
        src = ("tensor", tgt.detach(), model.prefix_len)
        out = eng.forward_logits(W, src, train=train, step=model._step)
        ctx.generation = eng.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if model.engine.generation != ctx.generation:
            # the activations this graph needs live in the engine's reusable buffers, and a later forward of the same
            # model has overwritten them: refuse rather than return the other call's gradients
            raise RuntimeError("TransformerDecoder.forward was called again (or caption_loss / generate ran) before the backward of "
                               "an earlier forward: the saved activations of that earlier call are gone. Run backward() before "
                               "the next forward of the same model, or use caption_loss(), whose gradients are owned per call.")
        keys = model._param_keys()
        grads = _grad_views(model, keys, dlogits.device)
        model.engine.backward_logits(ctx.W, grads, dlogits.contiguous(), need_input_grad=ctx.need_input)
        gin = None
        if ctx.need_input:
            B, S, _ = model.engine._shape
            gin = model.engine.input_grad.view(S, B, -1).clone()
        return (None, gin) + tuple(grads[k] for k in keys)


class _CaptionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: TransformerDecoder, inputs, labels, pos_want, *params):
        eng = model.engine
        model._step += 1
        W = model._weights()
        keys = model._param_keys()
        pos, want = pos_want
        grads = _grad_views(model, keys, labels.device) if want else None
        src = model._source(inputs)
        loss_sum, correct, M = eng.loss_and_grads(W, grads, src, labels, train=model.training, step=model._step, pos=pos,
                                                  want_grads=want)
        ctx.grads, ctx.keys = grads, keys
        ctx.mark_non_differentiable(correct)
        return loss_sum[0] / M, correct[0]

    @staticmethod
    def backward(ctx, dloss, _dcorrect):
        if ctx.grads is None:
            return (None, None, None, None) + tuple(None for _ in ctx.keys)
        return (None, None, None, None) + tuple(ctx.grads[k] * dloss for k in ctx.keys)


# ---------------------------------------------------------------------------- fused trainer
class CaptionTrainer:
    """Forward + backward + torch.optim.Adam-equivalent update (coupled weight decay,
    decoder.py:135-139) entirely on libcapdec kernels, over one flat parameter buffer.
    With ``torch.distributed`` initialised, gradients are summed across ranks by bucketed
    all-reduce (RCCL over xGMI) launched as soon as each bucket's gradient is complete
    -- the vocabulary slabs first -- and overlapped with the rest of the backward."""

    def __init__(self, model: TransformerDecoder, lr=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=None,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True, grad_compress: Optional[str] = None,
                 record_bucket_timing: bool = False):
        self.model = model
        self.lr = config.learning_rate if lr is None else lr
        self.weight_decay = config.weight_decay if weight_decay is None else weight_decay
        self.betas, self.eps = betas, eps
        dev = model.norm1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("CaptionTrainer needs the model on the GPU (model.to('cuda')); there is no CPU path")
        self.store = ParamStore(model.geometry, dev, model.options.compute_dtype, with_optimizer_state=True)
        self.store.load({k: v for k, v in model.state_dict().items()})
        # parameters become views of the flat buffer: model.state_dict() stays current
        named = model._named()
        for key, _, _ in self.store.layout:
            named[key].data = self.store.param(key)
        self.W = weights_from_store(self.store)
        self.grads = {key: self.store.grad(key) for key, _, _ in self.store.layout}
        self.step_count = 0
        from .dp import GradReducer
        # grad_compress="bf16": the ranks exchange bfloat16 gradients (half the bytes on the links); moments and update stay f32
        self.reducer = GradReducer(self.store, process_group, bucket_bytes=bucket_bytes, overlap=overlap, compress=grad_compress,
                                   record_timing=record_bucket_timing)
        if self.reducer.enabled and self.reducer.rank > 0:
            # every rank draws its own dropout masks: with one shared seed the shards of a global batch would all be
            # thinned by the same pattern (the engine is built lazily from the options, so this precedes its first use)
            model.options.dropout_seed = (model.options.dropout_seed + 0x9E3779B1 * self.reducer.rank) & 0x7FFFFFFF
            if model._engine is not None:
                model._engine.opt.dropout_seed = model.options.dropout_seed

    def step(self, inputs, labels, pos=None, global_batch: Optional[int] = None):
        """One optimisation step on one batch.  Returns (loss_sum [1], correct [1], n_rows)
        as device tensors (no host synchronisation).

        Data parallel: each rank passes ITS shard.  The loss is a mean over all rows of the global batch
        (decoder.py:140), so a rank's gradient is weighted by its share of the rows: pass ``global_batch`` (the
        number of samples over all ranks) whenever the shards are not equally large; equal shards need nothing."""
        model, eng = self.model, self.model.engine
        self.step_count += 1
        self.reducer.begin()
        grad_scale = 1.0
        if global_batch is not None and self.reducer.enabled:
            # ranks sum their gradients and the optimiser divides by world: local mean * (local rows * world / global rows)
            grad_scale = labels.shape[0] * self.reducer.world_size / float(global_batch)
        out = eng.loss_and_grads(self.W, self.grads, model._source(inputs), labels, train=model.training,
                                 step=self.step_count, pos=pos, want_grads=True, grad_scale=grad_scale,
                                 on_grad_ready=self.reducer.on_grad_ready)
        model._shadow.clear()      # module-mode shadows of the matrices are stale after this update
        st = self.store
        head_first, head_shape = st.offsets[HEAD_W]
        head_rows = head_shape[0]

        def adam(lo, hi):          # bucket boundaries are 64-element aligned: every slice stays 16-byte aligned
            ops.adam_step(st.w[lo:hi], st.g[lo:hi], st.m[lo:hi], st.v[lo:hi], self.step_count, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, grad_scale=1.0 / self.reducer.world_size,
                          w_bf16=st.shadow[lo:hi] if st.shadow is not st.w else None,
                          # the vocabulary matrix's k-panel image is refreshed by the same pass
                          panels=st.head_panels, panel_first=head_first - lo, panel_rows=head_rows)
        # one launch over the flat buffer on a single GPU; with several, one launch per all-reduce bucket, so
        # that the update of the vocabulary matrix overlaps the reduction of the last bucket
        self.reducer.drain(adam)
        st.refresh_block_panels()      # (a stack only: the images of the blocks' matrices follow the update)
        return out

    @torch.no_grad()
    def evaluate(self, inputs, labels, pos=None):
        eng = self.model.engine
        return eng.loss_and_grads(self.W, None, self.model._source(inputs), labels, train=False, want_grads=False, pos=pos)


# ---------------------------------------------------------------------------- the reference's training loop
def train_decoder(processed_data, config=config, *, options: Optional[EngineOptions] = None, log=print, logger=None):
    """Same contract as the reference's train_decoder (decoder.py:96-216):
    ``processed_data`` = {'decoder_inputs': [S,N,d], 'caption_input_ids': [N,L],
    'caption_labels': [N,L]}; 90/10 split by position; every epoch runs the train split
    and then the "test" split, taking optimiser steps on BOTH (model.train() is never left,
    decoder.py:144-147,188-190); per split the mean of the per-batch losses and the token
    accuracy are reported.  Differences: metrics are accumulated on the device and read once
    per split instead of twice per step (decoder.py:183,192); ``logger`` (a callable taking
    a dict) replaces wandb.log.  Returns the trained model."""
    start = time.time()
    total = processed_data["decoder_inputs"].size(1)
    train_size = int(config.train_fraction * total)
    bounds = (("train", 0, train_size), ("test", train_size, total))
    log(f"\nSplit data into:\n- Training: {train_size} examples\n- Testing: {total - train_size} examples")
    device = torch.device(config.device)
    if device.type != "cuda":
        raise RuntimeError("train_decoder runs on the GPU only (config.device='cuda'); there is no CPU path")
    model = TransformerDecoder(d_model=config.d_model, nhead=config.nhead, num_layers=config.num_layers,
                               dim_feedforward=config.dim_feedforward, dropout=config.dropout, vocab_size=50260,
                               options=options).to(device)
    trainer = CaptionTrainer(model, lr=config.learning_rate, weight_decay=config.weight_decay)
    L = config.max_caption_length
    if processed_data["caption_labels"].size(1) != L:
        raise ValueError(f"labels have {processed_data['caption_labels'].size(1)} columns, config.max_caption_length is {L}")
    # each split lives on the device when it fits (its batches are then strided views, decoder.py:163),
    # otherwise it is streamed from pinned host memory ahead of the compute stream
    from .dataset import BatchSource
    splits = [(name, BatchSource(processed_data, config.batch_size, device, lo=lo, hi=hi)) for name, lo, hi in bounds]
    history = []
    for epoch in range(config.num_epochs):
        model.train()
        for split_name, source in splits:
            loss_acc = torch.zeros(1, device=device)
            correct_acc = torch.zeros(1, dtype=torch.int64, device=device)
            batches = tokens = 0
            for x, y in source:
                loss_sum, correct, rows = trainer.step(x, y)
                loss_acc += loss_sum / rows
                correct_acc += correct
                batches += 1
                tokens += rows
            avg_loss = (loss_acc.item() / batches) if batches else 0
            accuracy = (correct_acc.item() / tokens) * 100 if tokens else 0
            rec = {f"{split_name}_loss": avg_loss, f"{split_name}_accuracy": accuracy, "epoch": epoch}
            history.append(rec)
            if logger is not None:
                logger(rec)
            log(f"Epoch {epoch + 1}/{config.num_epochs}, {split_name} Loss: {avg_loss:.4f}, Accuracy: {accuracy:.2f}%")
    runtime = time.time() - start
    if logger is not None:
        logger({"total_runtime": runtime})
    log(f"\nTotal runtime: {runtime:.2f} seconds")
    model.training_history = history
    return model
