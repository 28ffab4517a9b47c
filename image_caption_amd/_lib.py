"""ctypes binding of libcapdec.so (the C ABI declared in include/capdec.h).

There is no fallback: if the shared library is missing, or a call returns an
error code, a RuntimeError is raised.  Tensors are passed as raw device
pointers (``tensor.data_ptr()``); torch only owns the memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcapdec.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
MASK_NONE, MASK_REFERENCE, MASK_PREFIX_CAUSAL = 0, 1, 2
GATE_NONE, GATE_RELU_OUT, GATE_GELU_PRE = 0, 1, 2
ABI_VERSION = 3

c_long, c_int, c_float, c_void_p = C.c_long, C.c_int, C.c_float, C.c_void_p


class Dropout(C.Structure):
    _fields_ = [("p", c_float), ("key0", C.c_uint32), ("key1", C.c_uint32)]


NO_DROPOUT = Dropout(0.0, 0, 0)


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_long), ("ldb", c_long), ("ldc", c_long),
        ("a_kmajor", c_int), ("b_kmajor", c_int),
        ("in_dtype", c_int), ("out_dtype", c_int),
        ("bias", c_void_p),
        ("col_scale", c_float), ("col_scale_n", c_int),
        ("act", c_int),
        ("drop", Dropout),
        ("residual", c_void_p), ("ldr", c_long),
        ("accumulate", c_int),
        ("gate", c_void_p), ("ldg", c_long), ("gate_mode", c_int), ("gate_dtype", c_int), ("gate_drop", Dropout),
        ("aux_out", c_void_p), ("ldaux", c_long),
        ("colsum", c_void_p),
        ("kv_cache", c_void_p), ("kv_q", c_int), ("kv_d", c_int), ("kv_B", c_int), ("kv_H", c_int), ("kv_hd", c_int),
        ("kv_smax", c_int), ("kv_s0", c_int),
        ("workspace", c_void_p), ("workspace_bytes", c_long),
        ("row_scale", c_void_p),
        ("kvec", c_void_p), ("kvec_out", c_void_p),
        ("a_panel_rows", c_long),
        ("b_panels", c_void_p), ("b_panel_rows", c_long),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p),
        ("q_bs", c_long), ("q_hs", c_long), ("q_ss", c_long),
        ("k_bs", c_long), ("k_hs", c_long), ("k_ss", c_long),
        ("v_bs", c_long), ("v_hs", c_long), ("v_ss", c_long),
        ("o_bs", c_long), ("o_hs", c_long), ("o_ss", c_long),
        ("dtype", c_int),
        ("B", c_int), ("H", c_int), ("S", c_int), ("hd", c_int),
        ("q_start", c_int), ("nq", c_int),
        ("mask_mode", c_int), ("prefix_len", c_int),
        ("drop", Dropout),
        ("lse", c_void_p),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("fwd", AttnDesc),
        ("dout", c_void_p), ("do_bs", c_long), ("do_hs", c_long), ("do_ss", c_long),
        ("dq", c_void_p), ("dq_bs", c_long), ("dq_hs", c_long), ("dq_ss", c_long),
        ("dk", c_void_p), ("dk_bs", c_long), ("dk_hs", c_long), ("dk_ss", c_long),
        ("dv", c_void_p), ("dv_bs", c_long), ("dv_hs", c_long), ("dv_ss", c_long),
        ("dq_scale", c_float),
        ("delta", c_void_p),
        ("colsum_q", c_void_p), ("colsum_kv", c_void_p),
    ]


class BlockParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "norm1_w", "norm1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "norm2_w", "norm2_b",
        "linear1_w", "linear1_b", "linear2_w", "linear2_b", "kv_cache")]


class DecodeDesc(C.Structure):
    _fields_ = [
        ("B", c_int), ("d_model", c_int), ("nhead", c_int), ("dim_ff", c_int), ("vocab", c_int), ("n_blocks", c_int), ("smax", c_int),
        ("dtype", c_int), ("activation", c_int), ("ln_eps", c_float),
        ("blocks", C.POINTER(BlockParams)),
        ("head_w", c_void_p), ("head_b", c_void_p), ("ld_head_w", c_long),
        ("emb", c_void_p), ("pos", c_void_p),
        ("x_last", c_void_p), ("ids", c_void_p), ("done", c_void_p), ("out", c_void_p), ("margin", c_void_p),
        ("max_len", c_int), ("t", c_int), ("s_new", c_int),
        ("eos_id", C.c_int64), ("pad_id", C.c_int64),
        ("workspace", c_void_p), ("workspace_bytes", c_long),
        ("head_w_panels", c_void_p), ("head_w_panel_rows", c_long),
    ]


class AdamPanels(C.Structure):
    _fields_ = [("panels", c_void_p), ("first", c_long), ("rows", c_long), ("K", c_int), ("panel_rows", c_long)]


class KPanels(C.Structure):
    _fields_ = [("a", c_void_p), ("a_rows", c_long), ("b", c_void_p), ("b_rows", c_long), ("probs_rows", c_long)]


# name -> (restype, argtypes); every symbol include/capdec.h declares
SIGNATURES = {
    "capdec_abi_version": (c_int, []),
    "capdec_last_error": (C.c_char_p, []),
    "capdec_dropout_mask": (c_int, [Dropout, c_long, c_void_p, c_void_p]),
    "capdec_concat_ln_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                     c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "capdec_seqfirst_ln_fwd": (c_int, [c_void_p, c_int, c_long, c_long, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int,
                                       c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "capdec_seqfirst_ln_bwd_affine": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "capdec_ln_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_long, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "capdec_ln_fold_fwd": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "capdec_ln_fold_bwd_workspace": (c_long, [c_int, c_int]),
    "capdec_ln_fold_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "capdec_ln_bwd_workspace": (c_long, [c_long, c_int]),
    "capdec_ln_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "capdec_ln_bwd_dropout_workspace": (c_long, [c_long, c_int]),
    "capdec_ln_bwd_dropout": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, Dropout, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "capdec_concat_ln_bwd_affine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "capdec_gemm_bias_act": (c_int, [C.POINTER(GemmDesc), c_void_p]),
    "capdec_attn_fwd": (c_int, [C.POINTER(AttnDesc), c_void_p]),
    "capdec_attn_bwd": (c_int, [C.POINTER(AttnBwdDesc), c_void_p]),
    "capdec_attn_bwd_colsum_rows": (None, [c_int, c_int, c_int, c_int, C.POINTER(c_long), C.POINTER(c_long)]),
    "capdec_attn_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "capdec_logits_ce_workspace": (c_long, [c_long, c_int]),
    "capdec_logits_ce_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_long, c_long,
                                     c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_long, C.POINTER(KPanels), c_void_p, c_long, c_void_p]),
    "capdec_pack_kpanels": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "capdec_pack_kpanels_batched": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "capdec_logits_ce_bwd_from_probs": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_int, c_int, c_void_p, c_float,
                                                c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_logits_ce_bwd_slab": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_long, c_long,
                                          c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_int,
                                          c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_logits_ce_bwd_workspace": (c_long, [c_long, c_int]),
    "capdec_logits_ce_fwd_ref": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_long, c_long,
                                         c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_long, c_void_p, C.POINTER(KPanels), c_void_p, c_long, c_void_p]),
    "capdec_logits_ce_bwd_ref_prepare": (c_int, [c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_int,
                                                 c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_colsum_weighted": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_logits_argmax": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_long, c_long,
                                     c_void_p, c_void_p, C.POINTER(KPanels), c_void_p, c_long, c_void_p]),
    "capdec_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_void_p]),
    "capdec_transpose_bf16": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p]),
    "capdec_colsum_workspace": (c_long, [c_long, c_int]),
    "capdec_dropout_bwd_cast": (c_int, [c_void_p, c_long, c_int, Dropout, c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_colsum": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "capdec_colsum_static": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_void_p]),
    "capdec_embed_rows": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "capdec_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                                 c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, C.POINTER(AdamPanels), c_void_p]),
    "capdec_greedy_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, C.c_int64, C.c_int64, c_void_p]),
    "capdec_decode_step_workspace": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "capdec_decode_step": (c_int, [C.POINTER(DecodeDesc), c_void_p]),
}

_lib = None


def load():
    """dlopen libcapdec.so and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.capdec_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libcapdec ABI {got} != binding ABI {ABI_VERSION}: rebuild the library")
    if os.environ.get("CAPDEC_ATTN_LEAN_DQ") and hasattr(lib, "capdecdbg_attn_lean_dq"):                # diagnostic
        lib.capdecdbg_attn_lean_dq.argtypes = [c_int]
        lib.capdecdbg_attn_lean_dq.restype = None
        lib.capdecdbg_attn_lean_dq(int(os.environ["CAPDEC_ATTN_LEAN_DQ"]))
    if os.environ.get("CAPDEC_HEAD_FWD_STORE_NT") and hasattr(lib, "capdecdbg_head_fwd_store_nt"):      # diagnostic
        lib.capdecdbg_head_fwd_store_nt.argtypes = [c_int]
        lib.capdecdbg_head_fwd_store_nt.restype = None
        lib.capdecdbg_head_fwd_store_nt(int(os.environ["CAPDEC_HEAD_FWD_STORE_NT"]))
    if os.environ.get("CAPDEC_ROUTE_SMALL") and hasattr(lib, "capdecdbg_gemm_route_small"):      # diagnostic: 0 = without the short-product routing rules
        lib.capdecdbg_gemm_route_small.argtypes = [c_int]
        lib.capdecdbg_gemm_route_small.restype = None
        lib.capdecdbg_gemm_route_small(int(os.environ["CAPDEC_ROUTE_SMALL"]))
    geo = os.environ.get("CAPDEC_HEAD_FWD_GEOMETRY")      # diagnostic: tile geometry of the persistent logits+CE forward (0 / 1)
    if geo is not None and hasattr(lib, "capdecdbg_head_fwd_geometry"):
        lib.capdecdbg_head_fwd_geometry.argtypes = [c_int]
        lib.capdecdbg_head_fwd_geometry.restype = None
        lib.capdecdbg_head_fwd_geometry(int(geo))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().capdec_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("image_caption_amd runs on an MI355X through libcapdec; got a CPU tensor "
                               "(there is no CPU fallback)")
