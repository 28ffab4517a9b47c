"""Host-side logic that needs no GPU: config contract, synthetic data, parameter
layout, dropout key derivation, the product package never importing the oracle."""
import ast
import os

import pytest
import torch

from helpers import ROOT
from image_caption_amd import ops
from image_caption_amd.config import DecoderConfig, config
from image_caption_amd.decoder import TransformerDecoder
from image_caption_amd.engine import HEAD_B, HEAD_W, ModelGeometry, ParamStore
from image_caption_amd.synthetic import make_captions, special_ids


def test_config_contract():
    # field names and defaults of the reference's config.py:7-27
    want = dict(d_model=768, nhead=8, num_layers=6, dim_feedforward=2048, dropout=0.1, batch_size=8, learning_rate=1e-4,
                num_epochs=10, weight_decay=0.01, initial_data_fraction=0.1, train_fraction=0.9, validation_fraction=0.1,
                max_caption_length=32, seed=42)
    for k, v in want.items():
        assert getattr(config, k) == v
    assert config.device in ("cuda", "cpu")
    with pytest.raises(ValueError, match="train_fraction must be between 0.0 and 1.0"):
        DecoderConfig(train_fraction=0.0)
    with pytest.raises(ValueError, match="validation_fraction must be between 0.0 and 1.0"):
        DecoderConfig(validation_fraction=1.5)
    DecoderConfig(seed=7)
    a = torch.rand(3)
    DecoderConfig(seed=7)
    assert torch.equal(a, torch.rand(3))          # __post_init__ seeds torch (config.py:41)


def test_synthetic_captions_shape_and_specials():
    gen = torch.Generator().manual_seed(0)
    V, L = 50260, 32
    ids = special_ids(V)
    assert (ids["sos"], ids["eos"], ids["pad"]) == (50257, 50258, 50259)
    inp, lab = make_captions(64, L, V, gen)
    assert inp.shape == lab.shape == (64, L) and inp.dtype == torch.int64
    assert (inp[:, 0] == ids["sos"]).all()
    for b in range(64):
        n = int((lab[b] == ids["eos"]).nonzero()[0])
        assert 3 <= n <= L - 1
        assert (lab[b, n + 1:] == ids["pad"]).all() and (lab[b, :n] < V - 3).all()
        assert torch.equal(inp[b, 1:n + 1], lab[b, :n])       # input is the label shifted right behind <sos>


def test_param_store_layout_is_backward_order():
    geo = ModelGeometry(64, 2, 128, 503, n_blocks=3)
    st = ParamStore(geo, "cpu", torch.float32, with_optimizer_state=True)
    keys = [k for k, _, _ in st.layout]
    assert keys[0] == HEAD_W and keys[1] == HEAD_B
    assert keys[2].startswith("blocks.2.") and keys[14].startswith("blocks.1.") and keys[26] == "self_attn.in_proj_weight"
    offs = [o for _, _, o in st.layout]
    assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
    m = TransformerDecoder(64, 2, 6, 128, 0.0, 503, n_blocks=3)
    st.load(m.state_dict())
    sd = st.state_dict()
    assert set(sd) == set(m.state_dict())
    for k, v in m.state_dict().items():
        assert torch.equal(sd[k], v)
    st.param("linear1.bias").fill_(3.0)
    assert float(st.w[st.offsets["linear1.bias"][0]]) == 3.0    # views alias the flat buffer


def test_dropout_keys_are_deterministic_and_distinct():
    a = ops.dropout_keys(42, 1, 5, 0.1)
    b = ops.dropout_keys(42, 1, 5, 0.1)
    c = ops.dropout_keys(42, 2, 5, 0.1)
    d = ops.dropout_keys(42, 1, 6, 0.1)
    assert (a.key0, a.key1) == (b.key0, b.key1)
    assert len({(k.key0, k.key1) for k in (a, c, d)}) == 3
    off = ops.dropout_keys(42, 1, 5, 0.0)
    assert off.p == 0.0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image_caption_amd")
    for fn in os.listdir(pkg):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] == "oracle" for n in names), f"{fn} imports the oracle"


def test_captions_from_ids_cuts_at_eos_and_drops_specials():
    import torch
    from image_caption_amd.img_text_concat import captions_from_ids
    V = 50260
    ids = torch.tensor([[50257, 11, 12, 50258, 13, 50259], [5, 6, 7, 8, 9, 10], [50258, 1, 2, 3, 4, 5]])
    assert captions_from_ids(ids) == [[11, 12], [5, 6, 7, 8, 9, 10], []]

    class Tok:
        def decode(self, t):
            return " ".join(f"w{i}" for i in t)
    assert captions_from_ids(ids, Tok(), vocab_size=V) == ["w11 w12", "w5 w6 w7 w8 w9 w10", ""]
    assert captions_from_ids([[1, 2, 9, 3]], eos_id=9, pad_id=0, sos_id=7) == [[1, 2]]


def test_batch_source_pins_only_what_fits_in_host_memory():
    """A split larger than the allowed share of the available host memory stays the (possibly memory-mapped) view it
    is: pinning would first materialise the whole slice."""
    import torch
    from image_caption_amd.dataset import BatchSource
    x = torch.zeros(4, 10, 8)[:, 2:7, :]
    assert BatchSource._pin(x, host_fraction=0.0) is x
    y = BatchSource._pin(x, host_fraction=0.5)
    assert y.shape == x.shape and torch.equal(y, x)
