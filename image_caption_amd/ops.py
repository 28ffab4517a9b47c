"""Tensor-level wrappers over the C ABI (one function per entry point).

These only translate torch tensors into pointers / leading dimensions and check
return codes; all arithmetic happens in libcapdec.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_NONE, BF16, F32, GATE_NONE, MASK_NONE, NO_DROPOUT, AttnBwdDesc, AttnDesc, Dropout, GemmDesc,
                   check, dtype_code, ptr, require_cuda, stream_ptr)


def _st():
    return stream_ptr()


def dropout_keys(seed: int, site: int, step: int = 0, p: float = 0.0) -> Dropout:
    """Derive the two 32-bit keys of one dropout site of one step from a seed."""
    if p <= 0.0:
        return Dropout(0.0, 0, 0)
    x = (seed * 0x9E3779B97F4A7C15 + site * 0xBF58476D1CE4E5B9 + step * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & (2**64 - 1)
    x ^= x >> 31
    x = (x * 0xD6E8FEB86659FD93) & (2**64 - 1)
    x ^= x >> 29
    return Dropout(float(p), x & 0xFFFFFFFF, (x >> 32) & 0xFFFFFFFF)


def dropout_mask(d: Dropout, n: int, device) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=device)
    require_cuda(out)
    check(_lib.load().capdec_dropout_mask(d, n, ptr(out), _st()), "capdec_dropout_mask")
    return out


def attn_dropout_mask(d: Dropout, B: int, H: int, S: int, device) -> torch.Tensor:
    """Keep-mask [B,H,S,S] of the attention probabilities: the kernels index a row of S keys with a stride of
    S rounded up to a multiple of 4 (each lane's four keys then share one hash)."""
    S4 = (S + 3) // 4 * 4
    return dropout_mask(d, B * H * S * S4, device).reshape(B, H, S, S4)[..., :S]


def concat_ln_fwd(img, sos, cap, gamma, beta, *, pos=None, eps=1e-5, x_out=None, x_row0=0, y_out=None, mean=None, rstd=None):
    require_cuda(img, sos, cap, pos, gamma, beta, x_out, y_out, mean, rstd)
    B, P, d = img.shape
    L = cap.shape[1]
    assert sos.shape == (B, 1, d) and cap.shape == (B, L, d)
    assert img.is_contiguous() and sos.is_contiguous() and cap.is_contiguous()
    assert img.dtype == sos.dtype == cap.dtype
    check(_lib.load().capdec_concat_ln_fwd(ptr(img), ptr(sos), ptr(cap), dtype_code(img), ptr(pos), ptr(gamma), ptr(beta),
                                           eps, B, P, L, d, ptr(x_out), x_row0, ptr(y_out),
                                           dtype_code(y_out) if y_out is not None else F32, ptr(mean), ptr(rstd), _st()),
          "capdec_concat_ln_fwd")


def _seqfirst_strides(tgt):
    assert tgt.dim() == 3 and tgt.stride(2) == 1, "seq-first batch must be [S, B, d] with unit stride along d"
    return tgt.stride(0), tgt.stride(1)


def seqfirst_ln_fwd(tgt, gamma, beta, *, pos=None, eps=1e-5, x_out=None, x_row0=0, y_out=None, mean=None, rstd=None):
    """tgt: [S, B, d] view (any seq / batch strides, e.g. dataset[:, i:j, :] on the device)."""
    require_cuda(tgt, pos, gamma, beta, x_out, y_out, mean, rstd)
    S, B, d = tgt.shape
    ss, bs = _seqfirst_strides(tgt)
    check(_lib.load().capdec_seqfirst_ln_fwd(ptr(tgt), dtype_code(tgt), ss, bs, ptr(pos), ptr(gamma), ptr(beta), eps, S, B, d,
                                             ptr(x_out), x_row0, ptr(y_out), dtype_code(y_out) if y_out is not None else F32,
                                             ptr(mean), ptr(rstd), _st()), "capdec_seqfirst_ln_fwd")


def seqfirst_ln_bwd_affine(dy, tgt, gamma, mean, rstd, dgamma, dbeta, partial, pos=None):
    require_cuda(dy, tgt, pos, gamma, mean, rstd, dgamma, dbeta, partial)
    S, B, d = tgt.shape
    ss, bs = _seqfirst_strides(tgt)
    assert dy.is_contiguous() and dy.shape == (S * B, d)
    check(_lib.load().capdec_seqfirst_ln_bwd_affine(ptr(dy), dtype_code(dy), ptr(tgt), dtype_code(tgt), ss, bs, ptr(pos), ptr(gamma),
                                                    ptr(mean), ptr(rstd), S, B, d, ptr(dgamma), ptr(dbeta), ptr(partial), _st()),
          "capdec_seqfirst_ln_bwd_affine")


def ln_fwd(x, gamma, beta, y_out, mean=None, rstd=None, eps=1e-5):
    require_cuda(x, gamma, beta, y_out)
    M, d = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and y_out.is_contiguous()
    check(_lib.load().capdec_ln_fwd(ptr(x), ptr(gamma), ptr(beta), eps, M, d, ptr(y_out), dtype_code(y_out), ptr(mean), ptr(rstd), _st()),
          "capdec_ln_fwd")


def ln_fold_fwd(W, bias, gamma, beta, Wf, bf):
    """Wf = W * gamma (per input column), bf = bias + W @ beta: LayerNorm's affine folded into the Linear after it."""
    require_cuda(W, bias, gamma, beta, Wf, bf)
    N, K = W.shape
    assert Wf.shape == W.shape and bf.shape == (N,) and bf.dtype == torch.float32
    check(_lib.load().capdec_ln_fold_fwd(ptr(W), dtype_code(W), _ld(W), ptr(bias), ptr(gamma), ptr(beta), N, K, ptr(Wf), dtype_code(Wf),
                                         _ld(Wf), ptr(bf), _st()), "capdec_ln_fold_fwd")


def ln_fold_bwd_workspace(N, K) -> int:
    return _lib.load().capdec_ln_fold_bwd_workspace(N, K)


def ln_fold_bwd(G, s, W, gamma, beta, dgamma, dbeta, partial):
    """G [N,K] f32 = dY^T xhat becomes the Linear's weight gradient in place; dgamma / dbeta are overwritten."""
    require_cuda(G, s, W, gamma, beta, dgamma, dbeta, partial)
    N, K = W.shape
    assert G.shape == (N, K) and G.dtype == torch.float32 and s.shape == (N,) and s.dtype == torch.float32
    assert partial.numel() * partial.element_size() >= ln_fold_bwd_workspace(N, K)
    check(_lib.load().capdec_ln_fold_bwd(ptr(G), _ld(G), ptr(s), ptr(W), dtype_code(W), _ld(W), ptr(gamma), ptr(beta), N, K,
                                         ptr(dgamma), ptr(dbeta), ptr(partial), _st()), "capdec_ln_fold_bwd")


def ln_bwd_workspace(M, d) -> int:
    return _lib.load().capdec_ln_bwd_workspace(M, d)


def ln_bwd(dy, x, gamma, mean, rstd, dx_in, dx_out, dgamma, dbeta, partial):
    """dx_out None: only the affine gradients."""
    require_cuda(dy, x, gamma, mean, rstd, dx_out, dgamma, dbeta, partial)
    M, d = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and (dx_out is None or dx_out.is_contiguous())
    check(_lib.load().capdec_ln_bwd(ptr(dy), dtype_code(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), M, d,
                                    ptr(dx_in), ptr(dx_out), ptr(dgamma), ptr(dbeta), ptr(partial), _st()), "capdec_ln_bwd")


def ln_bwd_dropout_workspace(M, d) -> int:
    return _lib.load().capdec_ln_bwd_dropout_workspace(M, d)


def ln_bwd_dropout(dy, x, gamma, mean, rstd, dx_in, dx_out, dgamma, dbeta, drop: Dropout, g, gbias, partial):
    """ln_bwd plus, from the same pass, g = dropout-backward of dx_out in g's dtype and gbias = column sums of g."""
    require_cuda(dy, x, gamma, mean, rstd, dx_out, dgamma, dbeta, g, gbias, partial)
    M, d = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dx_out.is_contiguous() and g.is_contiguous()
    assert g.shape == (M, d) and dx_out.shape == (M, d) and gbias.numel() == d and gbias.dtype == torch.float32
    assert partial.numel() * partial.element_size() >= ln_bwd_dropout_workspace(M, d)
    check(_lib.load().capdec_ln_bwd_dropout(ptr(dy), dtype_code(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), M, d,
                                            ptr(dx_in), ptr(dx_out), ptr(dgamma), ptr(dbeta), drop, ptr(g), dtype_code(g), ptr(gbias),
                                            ptr(partial), _st()), "capdec_ln_bwd_dropout")


def concat_ln_bwd_affine(dy, img, sos, cap, gamma, mean, rstd, dgamma, dbeta, partial, pos=None):
    require_cuda(dy, img, sos, cap, pos, gamma, mean, rstd, dgamma, dbeta, partial)
    B, P, d = img.shape
    L = cap.shape[1]
    assert dy.is_contiguous() and dy.shape == ((P + 1 + L) * B, d)
    assert img.is_contiguous() and sos.is_contiguous() and cap.is_contiguous() and img.dtype == sos.dtype == cap.dtype
    check(_lib.load().capdec_concat_ln_bwd_affine(ptr(dy), dtype_code(dy), ptr(img), ptr(sos), ptr(cap), dtype_code(img), ptr(pos),
                                                  ptr(gamma), ptr(mean), ptr(rstd), B, P, L, d, ptr(dgamma), ptr(dbeta),
                                                  ptr(partial), _st()), "capdec_concat_ln_bwd_affine")


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "matrix operands must be 2-D with unit inner stride"
    return t.stride(0)


def gemm(A, B, Cout, *, M=None, N=None, K=None, a_kmajor=False, b_kmajor=False, bias=None, col_scale=1.0, col_scale_n=0,
         act=ACT_NONE, drop: Dropout = NO_DROPOUT, residual=None, accumulate=False, gate=None, gate_mode=GATE_NONE,
         gate_drop: Dropout = NO_DROPOUT, aux_out=None, colsum=None, kv_cache=None, kv_geom=None, workspace=None, row_scale=None, kvec=None, kvec_out=None,
         a_panels=False, b_panels=None):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  A is [M,K] (or [K,M] if a_kmajor),
    B is [N,K] (or [K,N] if b_kmajor).  a_panels: A is the column-panel image [cols/32, R, 32] of that stored matrix
    (M and K must then be given)."""
    require_cuda(A, B, Cout, bias, residual, gate, aux_out, colsum, kv_cache, row_scale, kvec, kvec_out)
    assert A.dtype == B.dtype
    if a_panels:
        assert A.dim() == 3 and A.shape[2] == 32 and A.is_contiguous() and M is not None and K is not None
        assert A.shape[1] >= (K if a_kmajor else M) and A.shape[0] * 32 >= (M if a_kmajor else K)
    if M is None:
        M = A.shape[1] if a_kmajor else A.shape[0]
    if K is None:
        K = A.shape[0] if a_kmajor else A.shape[1]
    if N is None:
        N = B.shape[1] if b_kmajor else B.shape[0]
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = (32 if a_panels else _ld(A)), _ld(B), _ld(Cout)
    if a_panels:
        d.a_panel_rows = A.shape[1]
    if b_panels is not None:               # k-panel image of the row-major B, for the kernels that read it
        require_cuda(b_panels)
        assert not b_kmajor and b_panels.dtype == torch.bfloat16 and b_panels.is_contiguous() and b_panels.dim() == 3
        assert b_panels.shape[0] * 32 == K and b_panels.shape[1] >= N and b_panels.shape[2] == 32
        d.b_panels, d.b_panel_rows = ptr(b_panels), b_panels.shape[1]
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.in_dtype, d.out_dtype = dtype_code(A), dtype_code(Cout)
    d.bias = ptr(bias)
    d.col_scale, d.col_scale_n = col_scale, col_scale_n
    d.act = act
    d.drop = drop
    if residual is not None:
        d.residual, d.ldr = ptr(residual), _ld(residual)
    d.accumulate = int(accumulate)
    if gate is not None:
        d.gate, d.ldg, d.gate_mode, d.gate_dtype = ptr(gate), _ld(gate), gate_mode, dtype_code(gate)
        d.gate_drop = gate_drop
    if aux_out is not None:
        assert aux_out.dtype == Cout.dtype
        d.aux_out, d.ldaux = ptr(aux_out), _ld(aux_out)
    d.colsum = ptr(colsum)
    if kv_cache is not None:
        qcols, dm, Bk, H, hd, smax, s0 = kv_geom
        assert kv_cache.dtype == Cout.dtype
        d.kv_cache, d.kv_q, d.kv_d, d.kv_B, d.kv_H, d.kv_hd, d.kv_smax, d.kv_s0 = ptr(kv_cache), qcols, dm, Bk, H, hd, smax, s0
    if workspace is not None:
        require_cuda(workspace)
        d.workspace, d.workspace_bytes = ptr(workspace), workspace.numel() * workspace.element_size()
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.numel() == M and row_scale.is_contiguous()
        d.row_scale = ptr(row_scale)
    if kvec_out is not None:               # kvec None: a vector of ones
        assert kvec is None or (kvec.dtype == torch.bfloat16 and kvec.numel() == K and kvec.is_contiguous())
        assert A.dtype == torch.bfloat16 and kvec_out.dtype == torch.float32 and kvec_out.numel() >= M and kvec_out.is_contiguous()
        d.kvec, d.kvec_out = ptr(kvec), ptr(kvec_out)
    check(_lib.load().capdec_gemm_bias_act(C.byref(d), _st()), "capdec_gemm_bias_act")
    return Cout


def _fill_attn(desc: AttnDesc, q, k, v, o, strides, B, H, S, hd, q_start, nq, mask_mode, prefix_len, drop, lse):
    require_cuda(q, k, v, o, lse)
    desc.q, desc.k, desc.v, desc.o = ptr(q), ptr(k), ptr(v), ptr(o)
    (desc.q_bs, desc.q_hs, desc.q_ss), (desc.k_bs, desc.k_hs, desc.k_ss), (desc.v_bs, desc.v_hs, desc.v_ss), \
        (desc.o_bs, desc.o_hs, desc.o_ss) = strides
    desc.dtype = dtype_code(q)
    desc.B, desc.H, desc.S, desc.hd = B, H, S, hd
    desc.q_start, desc.nq = q_start, nq
    desc.mask_mode, desc.prefix_len = mask_mode, prefix_len
    desc.drop = drop
    desc.lse = ptr(lse)


def packed_strides(B, d, hd, width):
    """(batch, head, seq) strides of a head slice of a packed [S*B, width] row matrix."""
    return (width, hd, B * width)


def attn_fwd_packed(qkv, o, lse, B, H, S, hd, *, q_start=0, nq=None, mask_mode=MASK_NONE, prefix_len=0,
                    drop: Dropout = NO_DROPOUT):
    """qkv [S*B, 3d] (q pre-scaled), o [S*B, d], lse [B*H*S]."""
    d = H * hd
    nq = S - q_start if nq is None else nq
    desc = AttnDesc()
    s3, s1 = packed_strides(B, d, hd, 3 * d), packed_strides(B, d, hd, d)
    _fill_attn(desc, qkv, qkv[:, d:], qkv[:, 2 * d:], o, (s3, s3, s3, s1), B, H, S, hd, q_start, nq, mask_mode, prefix_len, drop, lse)
    check(_lib.load().capdec_attn_fwd(C.byref(desc), _st()), "capdec_attn_fwd")


def attn_bwd_colsum_rows(B, S, nq, hd):
    """(rows of colsum_q, rows of colsum_kv) for attn_bwd_packed with these shapes."""
    qr, kr = C.c_long(0), C.c_long(0)
    _lib.load().capdec_attn_bwd_colsum_rows(B, S, nq, hd, C.byref(qr), C.byref(kr))
    return qr.value, kr.value


def attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, *, q_start=0, nq=None, mask_mode=MASK_NONE, prefix_len=0,
                    drop: Dropout = NO_DROPOUT, dq_scale=1.0, colsum_q=None, colsum_kv=None):
    """colsum_q [rows_q, H*hd] / colsum_kv [rows_kv, 2*H*hd] (f32, optional; rows from attn_bwd_colsum_rows): per-workgroup
    column sums of dQ and dK | dV, whose sums over the rows are the packed projection's bias gradient."""
    d = H * hd
    nq = S - q_start if nq is None else nq
    if colsum_q is not None or colsum_kv is not None:
        require_cuda(colsum_q, colsum_kv)
        qr, kr = attn_bwd_colsum_rows(B, S, nq, hd)
        assert colsum_q is None or (colsum_q.dtype == torch.float32 and colsum_q.shape == (qr, d) and colsum_q.is_contiguous())
        assert colsum_kv is None or (colsum_kv.dtype == torch.float32 and colsum_kv.shape == (kr, 2 * d) and colsum_kv.is_contiguous())
    desc = AttnBwdDesc()
    s3, s1 = packed_strides(B, d, hd, 3 * d), packed_strides(B, d, hd, d)
    _fill_attn(desc.fwd, qkv, qkv[:, d:], qkv[:, 2 * d:], o, (s3, s3, s3, s1), B, H, S, hd, q_start, nq, mask_mode, prefix_len, drop, lse)
    desc.dout, (desc.do_bs, desc.do_hs, desc.do_ss) = ptr(dout), s1
    desc.dq, (desc.dq_bs, desc.dq_hs, desc.dq_ss) = ptr(dqkv), s3
    desc.dk, (desc.dk_bs, desc.dk_hs, desc.dk_ss) = ptr(dqkv[:, d:]), s3
    desc.dv, (desc.dv_bs, desc.dv_hs, desc.dv_ss) = ptr(dqkv[:, 2 * d:]), s3
    desc.dq_scale = dq_scale
    desc.delta = ptr(delta)
    desc.colsum_q, desc.colsum_kv = ptr(colsum_q), ptr(colsum_kv)
    check(_lib.load().capdec_attn_bwd(C.byref(desc), _st()), "capdec_attn_bwd")


def attn_decode(q, kv_cache, o, B, H, hd, smax, length):
    require_cuda(q, kv_cache, o)
    check(_lib.load().capdec_attn_decode(ptr(q), ptr(kv_cache), ptr(o), dtype_code(q), B, H, hd, smax, length, _st()),
          "capdec_attn_decode")


def logits_ce_workspace(M, V) -> int:
    return _lib.load().capdec_logits_ce_workspace(M, V)


def kpanel_shape(rows: int, K: int):
    """Shape of the k-panel image of a [rows, K] matrix: [K / 32 panels, rows, 32]."""
    assert K % 32 == 0
    return (K // 32, rows, 32)


def pack_kpanels(src, panels, rowmajor=None):
    """panels[k // 32, r, k % 32] = bf16(src[r, k]); rowmajor (optional, bf16 [rows, K]) receives the plain cast as well."""
    require_cuda(src, panels, rowmajor)
    rows, K = src.shape
    assert panels.dtype == torch.bfloat16 and panels.is_contiguous() and panels.shape[0] == K // 32 and panels.shape[2] == 32 and panels.shape[1] >= rows
    assert src.stride(1) == 1 and (rowmajor is None or (rowmajor.dtype == torch.bfloat16 and rowmajor.shape == src.shape and rowmajor.stride(1) == 1))
    check(_lib.load().capdec_pack_kpanels(ptr(src), dtype_code(src), rows, K, _ld(src), ptr(panels), panels.shape[1],
                                          ptr(rowmajor), _ld(rowmajor) if rowmajor is not None else 0, _st()), "capdec_pack_kpanels")
    return panels


def pack_kpanels_batched(src_flat, first, rows, K, batch, src_batch_stride, panels):
    """Images of `batch` row-major [rows, K] matrices that lie src_batch_stride elements apart in the 1-D tensor src_flat,
    the first at element `first`: panels[z] ([batch, K/32, rows, 32] bf16) = image of matrix z."""
    require_cuda(src_flat, panels)
    assert src_flat.dim() == 1 and src_flat.is_contiguous() and panels.is_contiguous() and panels.dtype == torch.bfloat16
    assert tuple(panels.shape) == (batch, K // 32, rows, 32) and first + (batch - 1) * src_batch_stride + rows * K <= src_flat.numel()
    check(_lib.load().capdec_pack_kpanels_batched(src_flat.data_ptr() + first * src_flat.element_size(), dtype_code(src_flat), rows, K, K,
                                                  src_batch_stride, ptr(panels), rows, panels[0].numel(), batch, _st()),
          "capdec_pack_kpanels_batched")
    return panels


def _kpanels(h_panels, w_panels, M, V, d):
    """The capdec_kpanels argument (None when no image is given); the images are [d / 32, rows, 32] bf16 tensors."""
    if h_panels is None and w_panels is None:
        return None
    for t, n in ((h_panels, M), (w_panels, V)):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous() and t.shape[0] == d // 32 and t.shape[1] >= n and t.shape[2] == 32)
    require_cuda(h_panels, w_panels)
    return _lib.KPanels(ptr(h_panels), h_panels.shape[1] if h_panels is not None else 0,
                        ptr(w_panels), w_panels.shape[1] if w_panels is not None else 0, 0)


def logits_ce_fwd(h, W, bias, labels, B, L, lse, argmax, loss_sum, correct, workspace, probs=None, h_panels=None, w_panels=None):
    """probs (optional, [M, ld] of h's dtype): receives the softmax numerators for logits_ce_bwd_from_probs.
    h_panels / w_panels (optional): k-panel images of h / W (pack_kpanels)."""
    require_cuda(h, W, bias, labels, lse, argmax, loss_sum, correct, workspace, probs)
    M, d = h.shape
    V = W.shape[0]
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    assert probs is None or (probs.dtype == h.dtype and probs.shape[0] == M)
    check(_lib.load().capdec_logits_ce_fwd(ptr(h), ptr(W), ptr(bias), dtype_code(h), M, V, d, _ld(h), _ld(W), ptr(labels), B, L,
                                           ptr(lse), ptr(argmax), ptr(loss_sum), ptr(correct),
                                           ptr(probs), _ld(probs) if probs is not None else 0, _kpanels(h_panels, w_panels, M, V, d),
                                           ptr(workspace), workspace.numel() * workspace.element_size(), _st()), "capdec_logits_ce_fwd")


def probs_panel_shape(M: int, V: int):
    """Shape of the column-panel image of the numerators / d(logits) [M, roundup128(V)]: [Vp / 32, M, 32]."""
    return ((V + 127) // 128 * 4, M, 32)


def logits_ce_fwd_ref(h, W, bias, labels, B, L, lse, argmax, loss_sum, correct, workspace, probs, rowref, h_panels=None, w_panels=None):
    """logits_ce_fwd with the numerators kept against one reference per row (rowref [M] f32, written here): probs [M, ld] bf16
    = exp(z - rowref).  The backward starts from logits_ce_bwd_ref_prepare."""
    require_cuda(h, W, bias, labels, lse, argmax, loss_sum, correct, workspace, probs, rowref)
    M, d = h.shape
    V = W.shape[0]
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    assert probs.dtype == h.dtype == torch.bfloat16
    assert rowref.dtype == torch.float32 and rowref.numel() == M and rowref.is_contiguous()
    kp = _kpanels(h_panels, w_panels, M, V, d)
    ld_probs, prows = _probs_layout(probs, M)
    if prows:
        kp = kp if kp is not None else _lib.KPanels(None, 0, None, 0, 0)
        kp.probs_rows = prows
    check(_lib.load().capdec_logits_ce_fwd_ref(ptr(h), ptr(W), ptr(bias), dtype_code(h), M, V, d, _ld(h), _ld(W), ptr(labels), B, L,
                                               ptr(lse), ptr(argmax), ptr(loss_sum), ptr(correct),
                                               ptr(probs), ld_probs, ptr(rowref), kp, ptr(workspace),
                                               workspace.numel() * workspace.element_size(), _st()), "capdec_logits_ce_fwd_ref")


def _probs_layout(probs, M):
    """(leading dimension, rows per panel) of a numerator buffer: a matrix [M, ld] (rows per panel 0) or its column-panel
    image [ld / 32, R >= M, 32] (probs_panel_shape)."""
    if probs.dim() == 3:
        assert probs.is_contiguous() and probs.shape[2] == 32 and probs.shape[1] >= M
        return probs.shape[0] * 32, probs.shape[1]
    assert probs.shape[0] == M
    return _ld(probs), 0


def logits_ce_bwd_ref_prepare(probs, V, h, labels, B, L, lse, rowref, gscale, h_scaled, rowscale, rowscale_bf16=None):
    """In place: the label's numerator of every row is patched so that d(logits) = rowscale[m] * probs[m, :];
    h_scaled = rowscale[:, None] * h (bf16), rowscale [M] f32."""
    require_cuda(probs, h, labels, lse, rowref, h_scaled, rowscale, rowscale_bf16)
    assert rowscale_bf16 is None or (rowscale_bf16.dtype == torch.bfloat16 and rowscale_bf16.numel() == h.shape[0])
    M, d = h.shape
    assert h_scaled.shape == (M, d) and h_scaled.dtype == h.dtype == torch.bfloat16
    assert rowscale.dtype == torch.float32 and rowscale.numel() == M
    ld_probs, prows = _probs_layout(probs, M)
    check(_lib.load().capdec_logits_ce_bwd_ref_prepare(ptr(probs), ld_probs, M, V, ptr(h), _ld(h), d, ptr(labels), B, L,
                                                       ptr(lse), ptr(rowref), gscale, ptr(h_scaled), _ld(h_scaled), ptr(rowscale), ptr(rowscale_bf16),
                                                       prows, _st()),
          "capdec_logits_ce_bwd_ref_prepare")


def logits_ce_bwd_from_probs(probs_dl, V, labels, B, L, lse, gscale, fwd_workspace, dbias=None, workspace=None):
    """In place: softmax numerators -> (softmax - onehot) * gscale; dbias (optional) overwritten."""
    require_cuda(probs_dl, labels, lse, fwd_workspace, dbias, workspace)
    M = probs_dl.shape[0]
    wsb = workspace.numel() * workspace.element_size() if workspace is not None else 0
    check(_lib.load().capdec_logits_ce_bwd_from_probs(ptr(probs_dl), dtype_code(probs_dl), M, V, _ld(probs_dl), ptr(labels), B, L,
                                                      ptr(lse), gscale, ptr(fwd_workspace),
                                                      fwd_workspace.numel() * fwd_workspace.element_size(),
                                                      ptr(dbias), ptr(workspace), wsb, _st()), "capdec_logits_ce_bwd_from_probs")


def logits_ce_bwd_workspace(M, nv) -> int:
    return _lib.load().capdec_logits_ce_bwd_workspace(M, nv)


def logits_ce_bwd_slab(h, W, bias, labels, B, L, lse, gscale, v0, nv, dl, dbias, workspace=None):
    require_cuda(h, W, bias, labels, lse, dl, dbias, workspace)
    M, d = h.shape
    V = W.shape[0]
    wsb = workspace.numel() * workspace.element_size() if workspace is not None else 0
    check(_lib.load().capdec_logits_ce_bwd_slab(ptr(h), ptr(W), ptr(bias), dtype_code(h), M, V, d, _ld(h), _ld(W), ptr(labels), B, L,
                                                ptr(lse), gscale, v0, nv, ptr(dl), _ld(dl), ptr(dbias), ptr(workspace), wsb, _st()),
          "capdec_logits_ce_bwd_slab")


def logits_argmax(h, W, bias, out_ids, margin, workspace, h_panels=None, w_panels=None):
    require_cuda(h, W, bias, out_ids, margin, workspace)
    M, d = h.shape
    V = W.shape[0]
    check(_lib.load().capdec_logits_argmax(ptr(h), ptr(W), ptr(bias), dtype_code(h), M, V, d, _ld(h), _ld(W), ptr(out_ids), ptr(margin),
                                           _kpanels(h_panels, w_panels, M, V, d), ptr(workspace), workspace.numel() * workspace.element_size(), _st()), "capdec_logits_argmax")


def cast(src, dst):
    require_cuda(src, dst)
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    check(_lib.load().capdec_cast(ptr(src), dtype_code(src), ptr(dst), dtype_code(dst), src.numel(), _st()), "capdec_cast")
    return dst


def transpose(src, dst):
    """dst[c, r] = src[r, c] (bf16, 2-D, unit inner stride on both sides)."""
    require_cuda(src, dst)
    rows, cols = src.shape
    assert src.dtype == dst.dtype == torch.bfloat16 and dst.shape == (cols, rows) and src.stride(1) == 1 and dst.stride(1) == 1
    check(_lib.load().capdec_transpose_bf16(ptr(src), src.stride(0), rows, cols, ptr(dst), dst.stride(0), _st()), "capdec_transpose_bf16")
    return dst


def colsum_workspace(M, N) -> int:
    return _lib.load().capdec_colsum_workspace(M, N)


def _colsum_ws(workspace, M, N, device):
    if workspace is None:
        workspace = torch.empty(colsum_workspace(M, N), dtype=torch.uint8, device=device)
    return workspace, workspace.numel() * workspace.element_size()


def dropout_bwd_cast(dx, drop: Dropout, g, colsum=None, workspace=None):
    """colsum (optional) is overwritten; workspace: colsum_workspace(M, N) bytes (allocated here if None)."""
    require_cuda(dx, g, colsum, workspace)
    M, N = dx.shape
    assert dx.dtype == torch.float32 and dx.is_contiguous() and g.is_contiguous()
    wsb = 0
    if colsum is not None:
        workspace, wsb = _colsum_ws(workspace, M, N, dx.device)
    check(_lib.load().capdec_dropout_bwd_cast(ptr(dx), M, N, drop, ptr(g), dtype_code(g), ptr(colsum), ptr(workspace), wsb, _st()),
          "capdec_dropout_bwd_cast")


def colsum(x, out, workspace=None):
    """out[n] = sum_m x[m, n] (overwritten, deterministic)."""
    require_cuda(x, out, workspace)
    M, N = x.shape
    workspace, wsb = _colsum_ws(workspace, M, N, x.device)
    check(_lib.load().capdec_colsum(ptr(x), dtype_code(x), M, N, _ld(x), ptr(out), ptr(workspace), wsb, _st()), "capdec_colsum")


def colsum_weighted(x, row_weight, out, workspace=None):
    """out[n] = sum_m row_weight[m] * x[m, n] (overwritten, deterministic)."""
    require_cuda(x, row_weight, out, workspace)
    M, N = x.shape
    assert row_weight.dtype == torch.float32 and row_weight.numel() == M
    workspace, wsb = _colsum_ws(workspace, M, N, x.device)
    check(_lib.load().capdec_colsum_weighted(ptr(x), dtype_code(x), M, N, _ld(x), ptr(row_weight), ptr(out), ptr(workspace), wsb, _st()),
          "capdec_colsum_weighted")


def embed_rows(table, ids, out, pos_row=None):
    require_cuda(table, ids, out, pos_row)
    n, d = out.shape
    check(_lib.load().capdec_embed_rows(ptr(table), ptr(ids), n, d, ptr(pos_row), ptr(out), _st()), "capdec_embed_rows")


def adam_step(w, g, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, w_bf16=None, panels=None, panel_first=0, panel_rows=0):
    """panels (optional, [K/32, >= panel_rows, 32] bf16): k-panel image of the row-major [panel_rows, K] matrix that starts
    panel_first elements after w[0] (may be negative when w is a slice of the flat buffer), refreshed by the same pass."""
    require_cuda(w, g, m, v, w_bf16, panels)
    pd = None
    if panels is not None:
        assert panels.dtype == torch.bfloat16 and panels.is_contiguous() and panels.shape[2] == 32 and panels.shape[1] >= panel_rows
        pd = _lib.AdamPanels(ptr(panels), panel_first, panel_rows, panels.shape[0] * 32, panels.shape[1])
    check(_lib.load().capdec_adam_step(ptr(w), ptr(g), ptr(m), ptr(v), w.numel(), step, lr, beta1, beta2, eps, weight_decay,
                                       grad_scale, ptr(w_bf16), pd, _st()), "capdec_adam_step")


def greedy_update(ids, done, out, t, eos_id, pad_id):
    require_cuda(ids, done, out)
    B, L = out.shape
    check(_lib.load().capdec_greedy_update(ptr(ids), ptr(done), ptr(out), B, L, t, eos_id, pad_id, _st()), "capdec_greedy_update")
