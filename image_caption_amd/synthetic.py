"""Flickr-shaped synthetic batches (there is no network for the real dataset).

Embeddings ~ N(0,1) (ViT's last_hidden_state is post-LayerNorm; GPT-2 hidden states are
treated the same way); captions have clip(round(N(13,5)),3,31) tokens drawn uniformly
from the GPT-2 range, then <eos>, then <pad> up to max_caption_length, with the special
ids in the reference tokenizer's add order (img_text_concat.py:22-26):
<sos>=V-3, <eos>=V-2, <pad>=V-1.
"""
from __future__ import annotations

import torch


def special_ids(vocab_size: int):
    return {"sos": vocab_size - 3, "eos": vocab_size - 2, "pad": vocab_size - 1}


def make_captions(B: int, L: int, vocab_size: int, gen: torch.Generator):
    """(caption_input_ids, caption_labels), both [B,L] int64 on the CPU:
    input = <sos> w1 .. wn <pad>.., labels = w1 .. wn <eos> <pad>.. (img_text_concat.py:74-92)."""
    ids = special_ids(vocab_size)
    n = torch.clamp(torch.round(torch.randn(B, generator=gen) * 5 + 13), 3, L - 1).long()
    words = torch.randint(0, vocab_size - 3, (B, L), generator=gen)
    ar = torch.arange(L)[None, :]
    labels = torch.where(ar < n[:, None], words, torch.full_like(words, ids["pad"]))
    labels[torch.arange(B), n] = ids["eos"]
    inputs = torch.full_like(words, ids["pad"])
    inputs[:, 0] = ids["sos"]
    inputs[:, 1:] = torch.where(ar[:, 1:] <= n[:, None], words[:, :-1], inputs[:, 1:])
    return inputs, labels


def make_batch(B: int, P: int = 196, L: int = 32, d: int = 768, vocab_size: int = 50260, seed: int = 42,
               device="cuda", dtype=torch.float32):
    """One batch resident on `device`: img [B,P,d], sos [B,1,d], cap [B,L,d], ids, labels."""
    gen = torch.Generator().manual_seed(seed)
    inputs, labels = make_captions(B, L, vocab_size, gen)
    dgen = torch.Generator(device=device).manual_seed(seed)
    img = torch.randn(B, P, d, generator=dgen, device=device, dtype=torch.float32).to(dtype)
    sos = torch.randn(B, 1, d, generator=dgen, device=device, dtype=torch.float32).to(dtype)
    cap = torch.randn(B, L, d, generator=dgen, device=device, dtype=torch.float32).to(dtype)
    return {"image_embeddings": img, "sos_embedding": sos, "caption_embeddings": cap,
            "caption_input_ids": inputs.to(device), "caption_labels": labels.to(device)}
