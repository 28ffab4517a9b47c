"""The prefix-join step of the reference's img_text_concat.py, on the GPU.

``process_batch`` keeps the reference's output contract (img_text_concat.py:130-134):

    {'decoder_inputs':   [S, B, d] float, rows = P patches | 1 <sos> slot | L caption rows,
     'caption_input_ids': [B, L] int64,
     'caption_labels':    [B, L] int64}

with S = P + 1 + L (img_text_concat.py:98-99) and the three shape checks of
img_text_concat.py:126-128.  The reference computes the embeddings from PIL images
and strings with frozen ViT / GPT-2 encoders built at import time
(img_text_concat.py:8-65); those are outside this package, so a batch either
carries the embeddings themselves or is the reference's {'image', 'caption'} dict
together with injected encoder callables (``Encoders``).
"""
from __future__ import annotations

import torch

from . import ops
from .config import config

# the reference repeats config.max_caption_length as a module constant (img_text_concat.py:30)
max_caption_length = 32


def join_prefix(image_embeddings: torch.Tensor, sos_embedding: torch.Tensor, caption_embeddings: torch.Tensor,
                pos: torch.Tensor = None) -> torch.Tensor:
    """cat([img, sos, cap], dim=1).transpose(0, 1) (img_text_concat.py:115-123) as one
    gather kernel: [B,P,d] + [B,1,d] + [B,L,d] -> [P+1+L, B, d] float32, contiguous."""
    B, P, d = image_embeddings.shape
    L = caption_embeddings.shape[1]
    S = P + 1 + L
    out = torch.empty(S * B, d, dtype=torch.float32, device=image_embeddings.device)
    ops.concat_ln_fwd(image_embeddings.contiguous(), sos_embedding.contiguous(), caption_embeddings.contiguous(),
                      None, None, pos=pos, x_out=out)
    return out.view(S, B, d)


class Encoders:
    """The frozen models the reference builds at import time (img_text_concat.py:8-27), as injectable callables:
    they need downloaded assets, so this package constructs none of them.

      get_patch_embeddings(images)  -> [B,P,d] float   (img_text_concat.py:32-50: ViT patches without the CLS row)
      get_token_embeddings(strings) -> [B,L,d] float   (img_text_concat.py:52-65: text-model hidden states, padded /
                                                        truncated to max_caption_length; the callable adds its own
                                                        "<sos> ... <eos>" wrapping exactly as the reference's does)
      tokenizer(strings, padding=, max_length=, truncation=, return_tensors=) -> {'input_ids': [B,L] int64}
      sos_embedding(batch_size)     -> [B,1,d] float   (img_text_concat.py:102-104: hidden state of a lone "<sos>")
    """

    def __init__(self, get_patch_embeddings, get_token_embeddings, tokenizer, sos_embedding):
        self.get_patch_embeddings = get_patch_embeddings
        self.get_token_embeddings = get_token_embeddings
        self.tokenizer = tokenizer
        self.sos_embedding = sos_embedding


# what process_batch() uses for the reference's {'image', 'caption'} batches when no `encoders` argument is given
default_encoders = None


def set_encoders(encoders):
    """Installs the module-wide encoders (the counterpart of the reference's module-level vit_model / text_model /
    tokenizer objects), so that ``process_batch(batch)`` can be called with the reference's one-argument form."""
    global default_encoders
    default_encoders = encoders


def _embed_reference_batch(batch, enc):
    """The reference's own batch form {'image': [PIL images], 'caption': [str]} -> the five tensors of the embeddings
    form, through injected encoders (img_text_concat.py:69-104, same strings in the same order)."""
    captions = list(batch["caption"])
    with_sos = ["<sos> " + c for c in captions]
    tok = dict(padding="max_length", max_length=max_caption_length, truncation=True, return_tensors="pt")
    return {
        "image_embeddings": enc.get_patch_embeddings(batch["image"]),
        "caption_embeddings": enc.get_token_embeddings(with_sos),
        "sos_embedding": enc.sos_embedding(len(captions)),
        "caption_input_ids": enc.tokenizer(with_sos, **tok)["input_ids"],
        "caption_labels": enc.tokenizer([c + " <eos>" for c in captions], **tok)["input_ids"],
    }


def process_batch(batch, *, encoders=None):
    """Two batch forms, one output contract (img_text_concat.py:130-134):

      * the reference's own: {'image': [PIL images], 'caption': [str]} (img_text_concat.py:67-70) -- needs encoders
        (the ``encoders`` argument or ``set_encoders``), since the frozen ViT / GPT-2 models are not part of this package;
      * embeddings already computed: {'image_embeddings': [B,P,d], 'sos_embedding': [B,1,d],
        'caption_embeddings': [B,L,d], 'caption_input_ids': [B,L], 'caption_labels': [B,L]}.
    """
    if "image_embeddings" not in batch:
        enc = encoders if encoders is not None else default_encoders
        if enc is None:
            raise ValueError("a {'image', 'caption'} batch needs encoders: pass process_batch(batch, encoders=Encoders(...)) "
                             "or call set_encoders() once (the frozen ViT / GPT-2 models are not constructed by this package)")
        batch = _embed_reference_batch(batch, enc)
    image_embeddings = batch["image_embeddings"]
    sos_embedding = batch["sos_embedding"]
    caption_embeddings = batch["caption_embeddings"]
    batch_size = image_embeddings.size(0)
    seq_len = image_embeddings.size(1) + 1 + caption_embeddings.size(1)
    decoder_inputs = join_prefix(image_embeddings, sos_embedding, caption_embeddings)
    assert decoder_inputs.size(0) == seq_len, f"Sequence length mismatch: {decoder_inputs.size(0)} != {seq_len}"
    assert decoder_inputs.size(1) == batch_size, f"Batch size mismatch: {decoder_inputs.size(1)} != {batch_size}"
    assert decoder_inputs.size(2) == image_embeddings.size(2), "Embedding dimension mismatch"
    return {
        "decoder_inputs": decoder_inputs,
        "caption_input_ids": batch["caption_input_ids"],
        "caption_labels": batch["caption_labels"],
    }


def captions_from_ids(ids, tokenizer=None, *, eos_id=None, pad_id=None, sos_id=None, vocab_size=50260):
    """Generated id rows -> per-sample token lists (and text when a tokenizer is given): the step
    after greedy decoding.  A caption ends before the first <eos>; <pad> and <sos> never belong to
    it.  Special ids default to the reference tokenizer's add order (img_text_concat.py:22-26).
    ``tokenizer`` is any object with ``decode(list_of_ids) -> str`` (the reference's GPT-2
    tokenizer; its assets are not available offline, so none is constructed here)."""
    sos_id = vocab_size - 3 if sos_id is None else sos_id
    eos_id = vocab_size - 2 if eos_id is None else eos_id
    pad_id = vocab_size - 1 if pad_id is None else pad_id
    rows = ids.tolist() if hasattr(ids, "tolist") else [list(r) for r in ids]
    out = []
    for row in rows:
        toks = []
        for t in row:
            if t == eos_id:
                break
            if t != pad_id and t != sos_id:
                toks.append(int(t))
        out.append(toks)
    if tokenizer is None:
        return out
    return [tokenizer.decode(t) for t in out]
