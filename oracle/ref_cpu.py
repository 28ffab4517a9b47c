"""CPU oracle for the captioning hot path.  TEST INFRASTRUCTURE ONLY.

This file is a float32 restatement, in plain torch tensor ops, of the algorithm
the reference runs on its hot path.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``image_caption_amd/`` imports it, and the product
path never falls back to it.

Pinning: ``tests/golden/make_golden.py`` imports the reference's own
``decoder.TransformerDecoder`` in the build container and records its outputs;
``tests/test_oracle_golden.py`` checks every function here against those
recorded vectors.  Modes the reference does not have (``prefix_causal`` mask,
GELU, positional add, several blocks, greedy decoding) are marked
"build-defined" below; the golden vectors for them were produced by composing
the reference class (see make_golden.py), or, where even that is impossible
(GELU / positional add), they are pinned only through the ReLU / no-position
mode of the same code path.

Reference citations are ``file:line`` into the upstream repository.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

PARAM_KEYS = (
    "self_attn.in_proj_weight", "self_attn.in_proj_bias",
    "self_attn.out_proj.weight", "self_attn.out_proj.bias",
    "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
    "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
    "output_projection.weight", "output_projection.bias",
)
BLOCK_KEYS = PARAM_KEYS[:12]

# The reference zeroes mask columns below 16*768 (decoder.py:35).
REFERENCE_MASK_OPEN_COLUMNS = 16 * 768


# --------------------------------------------------------------------------
# img_text_concat.py:115-123 -- the prefix join
# --------------------------------------------------------------------------
def concat_prefix(img: torch.Tensor, sos: torch.Tensor, cap: torch.Tensor,
                  pos: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,P,d] + [B,1,d] + [B,L,d] -> [P+1+L, B, d] (rows: patches, sos, caption).

    Follows img_text_concat.py:115-123: cat along dim 1 then transpose(0, 1).
    ``pos`` ([S,d], build-defined, no reference counterpart) is added per row.
    """
    x = torch.cat([img, sos, cap], dim=1).transpose(0, 1).contiguous()
    # the three shape checks of img_text_concat.py:126-128
    assert x.size(0) == img.size(1) + 1 + cap.size(1)
    assert x.size(1) == img.size(0)
    assert x.size(2) == img.size(2)
    if pos is not None:
        x = x + pos[:, None, :]
    return x


# --------------------------------------------------------------------------
# decoder.py:31-38 -- the attention mask
# --------------------------------------------------------------------------
def caption_mask(seq_len: int, mode: str = "reference", prefix_len: int = 0) -> torch.Tensor:
    """Additive float mask [S,S].

    mode "reference": decoder.py:33-37 literally -- all -inf, columns below
    16*768 set to 0, then triu(diagonal=1).  For S <= 12288 this is all zeros.
    mode "prefix_causal" (build-defined): key j is visible to query i when
    j < prefix_len or j <= i.
    mode "none": zeros.
    """
    if mode == "reference":
        m = torch.full((seq_len, seq_len), float("-inf"))
        m[:, :REFERENCE_MASK_OPEN_COLUMNS] = 0.0
        return torch.triu(m, diagonal=1)
    if mode == "prefix_causal":
        i = torch.arange(seq_len)[:, None]
        j = torch.arange(seq_len)[None, :]
        vis = (j < prefix_len) | (j <= i)
        m = torch.zeros(seq_len, seq_len)
        m[~vis] = float("-inf")
        return m
    if mode == "none":
        return torch.zeros(seq_len, seq_len)
    raise ValueError(f"unknown mask mode {mode!r}")


# --------------------------------------------------------------------------
# decoder.py:40-68 -- one pre-LN block and the vocabulary projection
# --------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (decoder.py:25-26)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def _drop(x: torch.Tensor, mask: Optional[torch.Tensor], p: float) -> torch.Tensor:
    """Inverted dropout with an explicit keep-mask (1 = keep).  mask None = identity."""
    if mask is None:
        return x
    return x * mask.to(x.dtype) * (1.0 / (1.0 - p))


def _activation(x: torch.Tensor, activation: str) -> torch.Tensor:
    if activation == "relu":                 # decoder.py:60
        return torch.relu(x)
    if activation == "gelu":                 # build-defined (erf form)
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    raise ValueError(activation)


def attention(h: torch.Tensor, p: Dict[str, torch.Tensor], nhead: int, mask: torch.Tensor,
              attn_keep: Optional[torch.Tensor] = None, drop_p: float = 0.0) -> torch.Tensor:
    """nn.MultiheadAttention(h,h,h, attn_mask=mask)[0], seq-first (decoder.py:54).

    Packed in_proj, chunk into q,k,v, heads split as view(S, B*H, hd), q scaled by
    hd**-0.5, scores + mask, softmax, dropout on the probabilities, @v, merge
    heads, out_proj.  ``attn_keep`` is [B,H,S,S] (1 = keep).
    """
    S, B, d = h.shape
    hd = d // nhead
    qkv = h @ p["self_attn.in_proj_weight"].t() + p["self_attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    # [S,B,H,hd] -> [B,H,S,hd]
    q = q.reshape(S, B, nhead, hd).permute(1, 2, 0, 3) * (hd ** -0.5)
    k = k.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    scores = q @ k.transpose(-1, -2) + mask
    probs = torch.softmax(scores, dim=-1)
    probs = _drop(probs, attn_keep, drop_p)
    o = probs @ v                                            # [B,H,S,hd]
    o = o.permute(2, 0, 1, 3).reshape(S, B, d)
    return o @ p["self_attn.out_proj.weight"].t() + p["self_attn.out_proj.bias"]


def block_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], nhead: int, mask: torch.Tensor,
                  activation: str = "relu", drop_p: float = 0.0,
                  keep: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """decoder.py:53-63.  ``keep`` holds explicit dropout keep-masks:
    "attn" [B,H,S,S], "resid1" [S,B,d], "hidden" [S,B,ff], "resid2" [S,B,d].
    "relu_gate" [S,B,ff] (1 = open), a checker's device: the ReLU's on/off pattern is taken from the caller
    instead of the sign of the pre-activation, to compare with a reduced-precision forward whose pre-activations
    near zero fall on the other side (the comparison then measures everything but those flips)."""
    keep = keep or {}
    h = layer_norm(x, p["norm1.weight"], p["norm1.bias"])
    a = attention(h, p, nhead, mask, keep.get("attn"), drop_p)
    x = x + _drop(a, keep.get("resid1"), drop_p)
    h = layer_norm(x, p["norm2.weight"], p["norm2.bias"])
    if keep.get("relu_gate") is not None and activation == "relu":
        h = (h @ p["linear1.weight"].t() + p["linear1.bias"]) * keep["relu_gate"].to(h.dtype)
    else:
        h = _activation(h @ p["linear1.weight"].t() + p["linear1.bias"], activation)
    h = _drop(h, keep.get("hidden"), drop_p)
    h = h @ p["linear2.weight"].t() + p["linear2.bias"]
    return x + _drop(h, keep.get("resid2"), drop_p)


def decoder_hidden(tgt: torch.Tensor, blocks: Sequence[Dict[str, torch.Tensor]], nhead: int,
                   mask_mode: str = "reference", prefix_len: int = 0, activation: str = "relu",
                   drop_p: float = 0.0, keep: Optional[List[Dict[str, torch.Tensor]]] = None) -> torch.Tensor:
    """Residual stream after the last block.  One block = the reference
    (decoder.py ignores num_layers, decoder.py:12-29); more are build-defined."""
    mask = caption_mask(tgt.size(0), mask_mode, prefix_len)
    x = tgt
    for i, p in enumerate(blocks):
        x = block_forward(x, p, nhead, mask, activation, drop_p, keep[i] if keep else None)
    return x


def decoder_forward(tgt: torch.Tensor, blocks: Sequence[Dict[str, torch.Tensor]],
                    head: Dict[str, torch.Tensor], nhead: int, **kw) -> torch.Tensor:
    """TransformerDecoder.forward: logits [S,B,V] for ALL rows (decoder.py:66)."""
    x = decoder_hidden(tgt, blocks, nhead, **kw)
    return x @ head["output_projection.weight"].t() + head["output_projection.bias"]


# --------------------------------------------------------------------------
# decoder.py:169-185 -- loss and accuracy of one batch
# --------------------------------------------------------------------------
def caption_loss(logits: torch.Tensor, labels: torch.Tensor, max_caption_length: int):
    """Keep the last L rows (decoder.py:172), permute to [B,L,V] (:175),
    CrossEntropyLoss mean over B*L with pads included (:140,:178-179), and the
    argmax accuracy count (:182-183).  Returns (loss, correct, predictions)."""
    cap = logits[-max_caption_length:].permute(1, 0, 2)
    flat = cap.reshape(-1, cap.size(-1))
    tgt = labels.reshape(-1)
    lse = torch.logsumexp(flat, dim=-1)
    picked = flat.gather(1, tgt[:, None]).squeeze(1)
    loss = (lse - picked).mean()
    pred = cap.argmax(dim=-1)
    correct = int((pred == labels).sum())
    return loss, correct, pred


def split_params(sd: Dict[str, torch.Tensor]):
    """state_dict of a 1-block reference model -> ([block], head)."""
    block = {k: sd[k] for k in BLOCK_KEYS}
    head = {k: sd[k] for k in PARAM_KEYS[12:]}
    return [block], head


# --------------------------------------------------------------------------
# decoder.py:135-139,190 -- torch.optim.Adam with coupled L2
# --------------------------------------------------------------------------
def adam_step(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              weight_decay: float = 0.01):
    """One Adam update as torch.optim.Adam does it (not AdamW): g += wd*w;
    m,v EMA; bias-corrected; w -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
    Returns new (w, m, v); ``step`` counts from 1."""
    g = g + weight_decay * w
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    w = w - (lr / bc1) * (m / denom)
    return w, m, v


# --------------------------------------------------------------------------
# build-defined: greedy decoding by full recompute around forward
# --------------------------------------------------------------------------
def greedy_decode(img: torch.Tensor, sos_slot: torch.Tensor, emb: torch.Tensor,
                  blocks, head, nhead: int, max_len: int, sos_id: int, eos_id: int,
                  pad_id: int, mask_mode: str = "reference", activation: str = "relu",
                  pos: Optional[torch.Tensor] = None, return_margins: bool = False):
    """Full-recompute greedy loop.  The reference has no generation code; this is
    the procedure consistent with its training layout (decoder.py:172 scores row
    P+1+i against label i, whose input is the embedding of caption_input_ids[:, i],
    img_text_concat.py:74-76,130-134):

        seq0 = [img (P rows) | sos_slot | E[sos]];   for t in 0..L-1:
        logits = forward(seq_t)[-1]; tok_t = argmax; seq_{t+1} = seq_t ++ E[tok_t]

    Rows that have emitted eos emit pad from then on (their state still advances
    with E[pad]).  Returns ids [B,max_len] (and top1-top2 margins [B,max_len]).
    """
    B, P, d = img.shape
    cur = torch.cat([img, sos_slot, emb[sos_id].expand(B, 1, d)], dim=1)      # [B,P+2,d]
    ids = torch.full((B, max_len), pad_id, dtype=torch.long)
    margins = torch.zeros(B, max_len)
    done = torch.zeros(B, dtype=torch.bool)
    for t in range(max_len):
        x = cur.transpose(0, 1)
        if pos is not None:
            x = x + pos[: x.size(0), None, :]
        h = decoder_hidden(x, blocks, nhead, mask_mode=mask_mode, prefix_len=P, activation=activation)
        logits = h[-1] @ head["output_projection.weight"].t() + head["output_projection.bias"]
        top2 = logits.topk(2, dim=-1)
        tok = top2.indices[:, 0]
        margins[:, t] = top2.values[:, 0] - top2.values[:, 1]
        tok = torch.where(done, torch.full_like(tok, pad_id), tok)
        ids[:, t] = tok
        done = done | (tok == eos_id)
        if t + 1 < max_len:
            cur = torch.cat([cur, emb[tok][:, None, :]], dim=1)
    return (ids, margins) if return_margins else ids


# --------------------------------------------------------------------------
# the reference's step, as written, for the CPU baseline timing
# --------------------------------------------------------------------------
class AsWrittenStep:
    """The body of decoder.py:169-190 with stock torch.nn modules (all-rows vocab
    projection, dropout active, Adam with coupled weight decay) -- what the
    reference executes per batch, used as the timed CPU baseline."""

    def __init__(self, d_model=768, nhead=8, dim_feedforward=2048, dropout=0.1, vocab_size=50260,
                 lr=1e-4, weight_decay=0.01, max_caption_length=32):
        nn = torch.nn
        self.L = max_caption_length

        class Net(nn.Module):
            def __init__(s):
                super().__init__()
                s.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
                s.linear1 = nn.Linear(d_model, dim_feedforward)
                s.dropout = nn.Dropout(dropout)
                s.linear2 = nn.Linear(dim_feedforward, d_model)
                s.norm1 = nn.LayerNorm(d_model)
                s.norm2 = nn.LayerNorm(d_model)
                s.output_projection = nn.Linear(d_model, vocab_size)

            def forward(s, tgt):
                mask = caption_mask(tgt.size(0), "reference")
                t2 = s.norm1(tgt)
                t2 = s.self_attn(t2, t2, t2, attn_mask=mask)[0]
                tgt = tgt + s.dropout(t2)
                t2 = s.linear2(s.dropout(torch.relu(s.linear1(s.norm2(tgt)))))
                tgt = tgt + s.dropout(t2)
                return s.output_projection(tgt)

        self.model = Net()
        self.model.train()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self.crit = nn.CrossEntropyLoss()

    def step(self, decoder_inputs: torch.Tensor, labels: torch.Tensor):
        out = self.model(decoder_inputs)
        cap = out[-self.L:].permute(1, 0, 2)
        loss = self.crit(cap.reshape(-1, cap.size(-1)), labels.reshape(-1))
        correct = (cap.argmax(-1) == labels).sum().item()
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.item(), correct
