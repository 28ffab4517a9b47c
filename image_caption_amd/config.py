"""Hyper-parameters of the captioning decoder.

Field names, defaults, validation and the seeding side effect are those of the
reference's ``config.py`` (config.py:4-68) so that code written against
``from config import config`` keeps working when pointed at this module.
Build-only knobs (mask mode, activation, compute dtype, block count) are NOT
added here; they live in ``engine.EngineOptions``.
"""
from dataclasses import dataclass
from typing import Optional

import torch


def _check_fraction(name: str, value: float) -> None:
    if not 0.0 < value <= 1.0:
        raise ValueError(f"{name} must be between 0.0 and 1.0")


@dataclass
class DecoderConfig:
    # architecture (config.py:7-11)
    d_model: int = 768
    nhead: int = 8
    num_layers: int = 6
    dim_feedforward: int = 2048
    dropout: float = 0.1
    # optimisation (config.py:14-17)
    batch_size: int = 8
    learning_rate: float = 1e-4
    num_epochs: int = 10
    weight_decay: float = 0.01
    # data (config.py:20-23)
    initial_data_fraction: float = 0.1
    train_fraction: float = 0.9
    validation_fraction: float = 0.1
    max_caption_length: int = 32
    # misc (config.py:26-27)
    device: Optional[str] = None
    seed: int = 42

    def __post_init__(self):
        _check_fraction("train_fraction", self.train_fraction)
        _check_fraction("validation_fraction", self.validation_fraction)
        if self.device is None:
            self.device = "cuda" if torch.cuda.is_available() else "cpu"
        torch.manual_seed(self.seed)
        if self.device == "cuda":
            torch.cuda.manual_seed(self.seed)


# the singleton the reference's modules import by name (config.py:46-68)
config = DecoderConfig()
