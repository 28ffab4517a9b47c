import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def gemm_mode():
    """Forces one bf16 kernel geometry (diagnostic hook of the library); restored afterwards."""
    import ctypes

    from image_caption_amd import _lib
    lib = _lib.load()
    lib.capdecdbg_gemm_mode.argtypes = [ctypes.c_int]
    lib.capdecdbg_gemm_mode.restype = None
    yield lib.capdecdbg_gemm_mode
    lib.capdecdbg_gemm_mode(0)
