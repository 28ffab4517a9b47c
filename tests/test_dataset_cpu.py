"""The processed-dataset file format (upload_to_wandb.py:17-31 / decoder.py:85) on the CPU:
concatenation axes, round trip through the file, safe loading, validation."""
import os

import pytest
import torch

from image_caption_amd import dataset


def _batches(n_batches=3, S=7, B=4, d=8, L=5):
    g = torch.Generator().manual_seed(5)
    out = []
    for k in range(n_batches):
        b = B - (k == n_batches - 1)                       # a ragged last batch, as the reference's loader produces
        out.append({"decoder_inputs": torch.randn(S, b, d, generator=g),
                    "caption_input_ids": torch.randint(0, 50, (b, L), generator=g),
                    "caption_labels": torch.randint(0, 50, (b, L), generator=g)})
    return out


def test_concat_axes_and_file_round_trip(tmp_path):
    bs = _batches()
    data = dataset.concat_batches(bs)
    n = sum(b["decoder_inputs"].size(1) for b in bs)
    assert data["decoder_inputs"].shape == (7, n, 8) and data["caption_labels"].shape == (n, 5)
    # sample axis is dim 1 of decoder_inputs and dim 0 of ids / labels
    assert torch.equal(data["decoder_inputs"][:, 4:8], bs[1]["decoder_inputs"])
    assert torch.equal(data["caption_input_ids"][4:8], bs[1]["caption_input_ids"])
    path = dataset.save_processed_dataset(bs, str(tmp_path))
    assert os.path.basename(path) == "processed_dataset.pt"
    for mmap in (True, False):
        back = dataset.load_processed_dataset(str(tmp_path), mmap=mmap)
        for k in dataset.KEYS:
            assert torch.equal(back[k], data[k])
    # the file is what the reference's own torch.load(...) call expects: a plain dict of the three tensors
    raw = torch.load(path, weights_only=True)
    assert sorted(raw.keys()) == sorted(dataset.KEYS)


def test_validation_errors(tmp_path):
    data = dataset.concat_batches(_batches())
    bad = dict(data)
    del bad["caption_labels"]
    with pytest.raises(ValueError, match="lacks"):
        dataset.save_processed_dataset(bad, str(tmp_path / "a.pt"))
    bad = dict(data, caption_labels=data["caption_labels"][:-1])
    with pytest.raises(ValueError, match="disagree"):
        dataset.save_processed_dataset(bad, str(tmp_path / "b.pt"))
    bad = dict(data, caption_labels=data["caption_labels"].int())
    with pytest.raises(ValueError, match="int64"):
        dataset.save_processed_dataset(bad, str(tmp_path / "c.pt"))
    torch.save([1, 2, 3], str(tmp_path / "d.pt"))
    with pytest.raises(ValueError, match="dict"):
        dataset.load_processed_dataset(str(tmp_path / "d.pt"))
    with pytest.raises(RuntimeError, match="GPU path only"):
        dataset.BatchSource(data, 4, "cpu")
