"""Seeded synthetic inputs and weights shared by the golden generator and the tests.

Everything is drawn from a CPU ``torch.Generator`` with explicit bounds, so the
same tensors can be regenerated on any machine from (config, seed) alone and
only outputs need to be stored in the golden files.
"""
from __future__ import annotations

import math
import os
import sys
from dataclasses import dataclass

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@dataclass(frozen=True)
class Shape:
    d: int = 768
    nhead: int = 8
    ff: int = 2048
    vocab: int = 50260
    P: int = 196
    L: int = 32
    B: int = 8

    @property
    def S(self) -> int:
        return self.P + 1 + self.L


TINY = Shape(d=64, nhead=2, ff=128, vocab=503, P=12, L=8, B=3)
DEFAULT_B8 = Shape()


def _uniform(gen, shape, bound):
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound


def make_block_params(sh: Shape, gen, bias_scale: float = 1.0):
    """One block's 12 tensors with the bounds of the default torch initialisers
    (xavier-uniform in_proj, kaiming-uniform(a=sqrt 5) Linear = +-1/sqrt(fan_in)).
    Biases and LayerNorm affine are randomised too (unlike the stock init, which
    zeroes some of them) so that every term of the arithmetic is exercised."""
    d, ff = sh.d, sh.ff
    xav = math.sqrt(6.0 / (d + 3 * d))
    return {
        "self_attn.in_proj_weight": _uniform(gen, (3 * d, d), xav),
        "self_attn.in_proj_bias": _uniform(gen, (3 * d,), 0.05 * bias_scale),
        "self_attn.out_proj.weight": _uniform(gen, (d, d), 1 / math.sqrt(d)),
        "self_attn.out_proj.bias": _uniform(gen, (d,), 0.05 * bias_scale),
        "linear1.weight": _uniform(gen, (ff, d), 1 / math.sqrt(d)),
        "linear1.bias": _uniform(gen, (ff,), 1 / math.sqrt(d)),
        "linear2.weight": _uniform(gen, (d, ff), 1 / math.sqrt(ff)),
        "linear2.bias": _uniform(gen, (d,), 1 / math.sqrt(ff)),
        "norm1.weight": 1.0 + _uniform(gen, (d,), 0.1),
        "norm1.bias": _uniform(gen, (d,), 0.1),
        "norm2.weight": 1.0 + _uniform(gen, (d,), 0.1),
        "norm2.bias": _uniform(gen, (d,), 0.1),
    }


def make_head_params(sh: Shape, gen, gain: float = 1.0):
    return {
        "output_projection.weight": _uniform(gen, (sh.vocab, sh.d), gain / math.sqrt(sh.d)),
        "output_projection.bias": _uniform(gen, (sh.vocab,), 1 / math.sqrt(sh.d)),
    }


def make_state_dict(sh: Shape, seed: int, head_gain: float = 1.0):
    gen = torch.Generator().manual_seed(seed)
    sd = make_block_params(sh, gen)
    sd.update(make_head_params(sh, gen, head_gain))
    return sd


def make_embeddings(sh: Shape, seed: int):
    """img [B,P,d], sos [B,1,d], cap [B,L,d] ~ N(0,1); labels [B,L] int64."""
    gen = torch.Generator().manual_seed(seed)
    img = torch.randn(sh.B, sh.P, sh.d, generator=gen)
    sos = torch.randn(sh.B, 1, sh.d, generator=gen)
    cap = torch.randn(sh.B, sh.L, sh.d, generator=gen)
    labels = torch.randint(0, sh.vocab, (sh.B, sh.L), generator=gen)
    return img, sos, cap, labels


def make_token_table(sh: Shape, seed: int):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(sh.vocab, sh.d, generator=gen)


def grad_slice(g: torch.Tensor) -> torch.Tensor:
    """What the default-shape fixture keeps of a gradient: vectors in full, matrices every 13th row and 7th
    column (the vocabulary matrix every 211th row and 3rd column)."""
    if g.dim() == 1:
        return g.clone()
    if g.shape[0] > 10000:
        return g[::211, ::3].clone()
    return g[::13, ::7].clone()


def train_loop_case():
    """Configuration, processed dataset and model seed of the train_decoder fixture: 40 samples -> 36 train
    (five batches, the last one ragged) + 4 test; labels drawn from six token ids so that token accuracy
    leaves zero within a few steps.  The vocabulary is the 50260 the reference's loop hard-codes (decoder.py:131)."""
    cfg = dict(d_model=64, nhead=2, num_layers=6, dim_feedforward=128, dropout=0.0, batch_size=8, learning_rate=2e-3,
               num_epochs=3, weight_decay=0.01, max_caption_length=8)
    gen = torch.Generator().manual_seed(4321)
    N, S, L = 40, 21, 8
    tokens = torch.tensor([17, 4242, 9001, 31337, 50257, 50259])
    labels = tokens[torch.randint(0, len(tokens), (N, L), generator=gen)]
    data = {"decoder_inputs": torch.randn(S, N, 64, generator=gen),
            "caption_input_ids": torch.roll(labels, 1, dims=1), "caption_labels": labels}
    return cfg, data, 1234
