"""The C-ABI library loads without a GPU and exports exactly what include/capdec.h
declares; the ctypes mirrors of its structs have the C compiler's layout."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from helpers import ROOT
from image_caption_amd import _lib

HEADER = os.path.join(ROOT, "include", "capdec.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(capdec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in capdec.h but not exported by libcapdec.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree on the symbol list"
    assert lib.capdec_abi_version() == _lib.ABI_VERSION
    assert lib.capdec_last_error() is not None


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "capdec.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(capdec_dropout), sizeof(capdec_gemm_desc), sizeof(capdec_attn_desc), sizeof(capdec_attn_bwd_desc),'
                   'offsetof(capdec_gemm_desc, kv_cache), offsetof(capdec_attn_desc, lse), offsetof(capdec_attn_bwd_desc, delta),'
                   'sizeof(capdec_kpanels), sizeof(capdec_decode_desc), offsetof(capdec_decode_desc, head_w_panels), sizeof(capdec_adam_panels));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.Dropout), ctypes.sizeof(_lib.GemmDesc), ctypes.sizeof(_lib.AttnDesc), ctypes.sizeof(_lib.AttnBwdDesc),
            _lib.GemmDesc.kv_cache.offset, _lib.AttnDesc.lse.offset, _lib.AttnBwdDesc.delta.offset,
            ctypes.sizeof(_lib.KPanels), ctypes.sizeof(_lib.DecodeDesc), _lib.DecodeDesc.head_w_panels.offset, ctypes.sizeof(_lib.AdamPanels)]
    assert got == want


def test_cpu_tensors_are_refused_not_computed():
    import torch
    from image_caption_amd import ops
    from image_caption_amd.decoder import TransformerDecoder
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cast(torch.zeros(4), torch.zeros(4, dtype=torch.bfloat16))
    m = TransformerDecoder(64, 2, 6, 128, 0.0, 503)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(21, 2, 64), None)
    with pytest.raises(RuntimeError, match="no CPU"):
        m.caption_loss(torch.zeros(21, 2, 64), torch.zeros(2, 8, dtype=torch.int64))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcapdec.so")
    with pytest.raises(RuntimeError, match="missing"):
        _lib.load()
