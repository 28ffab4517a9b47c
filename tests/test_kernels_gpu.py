"""Per-entry-point checks of libcapdec on a real MI355X, each against a plain torch
float32 computation of the same op on the same inputs (bf16 inputs are rounded
first, then the reference is computed in f32).  Tolerances: f32 paths 1e-4
relative to the output scale; bf16 paths 2e-2 (bf16 has 8 significant bits;
accumulation is f32)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from image_caption_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def close(a, b, tol, what=""):
    a = a.float()
    b = b.float()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


# ---------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.5e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 328, 136), (77, 503, 64), (1000, 72, 520), (33, 40, 8), (64, 136, 44)])
def test_gemm_layouts(dtype, tol, akm, bkm, M, N, K):
    A = rnd(M, K, dtype=dtype, seed=1)
    B = rnd(N, K, dtype=dtype, seed=2)
    if dtype == torch.bfloat16 and ((akm and M % 8) or (bkm and N % 8)):
        pytest.skip("k-major bf16 operands need a leading dimension that is a multiple of 8")
    def padded(t):      # same values, row pitch rounded up to 8 elements, padding poisoned
        buf = torch.full((t.shape[0], (t.shape[1] + 7) // 8 * 8), float("nan"), device=DEV, dtype=t.dtype)
        buf[:, :t.shape[1]] = t
        return buf[:, :t.shape[1]]
    Aop = A.t().contiguous() if akm else padded(A)
    Bop = B.t().contiguous() if bkm else padded(B)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Aop, Bop, out, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm)
    ref = A.float() @ B.float().t()
    close(out, ref, tol, "gemm")


# 1 register-staged, 2 ring 128x128, 3 ring 256x128 (8 waves), 4 ring 256x128 (4 waves of 128x64),
# 5 ring 256x256 (8 waves of 128x64; plain products only), 6 pipelined 256x256 with 4 waves of 128x128 (plain products
# only), 7 pipelined 256x128 with 4 waves of 128x64 (plain products in any layout, row-major with every epilogue)
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (300, 264, 192), (1000, 72, 96), (515, 640, 768), (64, 136, 64)])
def test_gemm_every_geometry(gemm_mode, mode, M, N, K):
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=31), rnd(N, K, dtype=torch.bfloat16, scale=0.1, seed=32)
    bias, res = rnd(N, seed=33), rnd(M, N, seed=34)
    ref = A.float() @ B.float().t() + bias
    gemm_mode(mode)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.gemm(A, B, out, bias=bias)
    out2 = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A, B, out2, bias=bias, residual=res)
    gemm_mode(0)
    close(out, ref, 1.5e-2, f"mode {mode}")
    close(out2, ref + res, 1.5e-2, f"mode {mode} + residual")


@pytest.mark.parametrize("mode", [1, 2, 4, 5, 6, 7])
@pytest.mark.parametrize("akm,bkm", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(264, 136, 96), (520, 72, 2080), (136, 264, 100)])
def test_gemm_reduction_major_every_geometry(gemm_mode, mode, akm, bkm, M, N, K):
    # K = 100 is not a whole number of 32-deep tiles: only the all-reduction-major case may take the ring
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=35), rnd(N, K, dtype=torch.bfloat16, scale=0.1, seed=36)
    if K % 8 and not (akm and bkm):
        pytest.skip("row-major bf16 operands need a leading dimension that is a multiple of 8")
    Aop = A.t().contiguous() if akm else A
    Bop = B.t().contiguous() if bkm else B
    ref = A.float() @ B.float().t()
    gemm_mode(mode)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Aop, Bop, out, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm)
    gemm_mode(0)
    close(out, ref, 1.5e-2, f"mode {mode}")


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 7])
def test_logits_ce_every_geometry(gemm_mode, mode):
    B_, L, V, d = 5, 8, 1111, 64                       # 40 rows: ragged in M; 1111 columns: ragged in N
    M = B_ * L
    h = rnd(M, d, dtype=torch.bfloat16, seed=41)
    w = rnd(V, d, dtype=torch.bfloat16, scale=0.3, seed=42)
    b = rnd(V, seed=43)
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(44)).to(DEV)
    z = h.float() @ w.float().t() + b
    lab_rows = labels.t().reshape(-1)                  # row m = l * B + b
    ref_lse = torch.logsumexp(z, 1)
    ref_loss = (ref_lse - z.gather(1, lab_rows[:, None])[:, 0]).sum()
    ref_dl = (torch.softmax(z, 1) - torch.nn.functional.one_hot(lab_rows, V)) / M
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    Vp = (V + 127) // 128 * 128
    dl = torch.full((M, Vp), float("nan"), device=DEV, dtype=torch.bfloat16)
    dbias = torch.full((V,), float("nan"), device=DEV)
    bws = torch.empty(ops.logits_ce_bwd_workspace(M, V), dtype=torch.uint8, device=DEV)
    gemm_mode(mode)
    ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws)
    ops.logits_ce_bwd_slab(h, w, b, labels, B_, L, lse, 1.0 / M, 0, V, dl, dbias, bws)
    gemm_mode(0)
    close(lse, ref_lse, 2e-3, "lse")
    assert abs(ls.item() - ref_loss.item()) <= 2e-3 * abs(ref_loss.item())
    assert (amax.long() == z.argmax(1)).float().mean().item() >= 0.95     # bf16 near-ties may flip
    close(dl[:, :V], ref_dl, 1e-2, "dl")
    close(dbias, ref_dl.sum(0), 1e-2, "dbias")


@pytest.mark.parametrize("M,N,K,relu,p,scale_n", [
    (515, 640, 768, True, 0.1, 0),            # ragged rows, several tiles
    (1000, 2056, 256, False, 0.0, 0),         # N a multiple of 8 only: the last tile's columns end inside a 64-column wave tile
    (300, 2304, 288, False, 0.2, 768),        # column scale on the first 768 columns (the query part of in_proj), 9 k-steps
    (8192, 2304, 768, True, 0.1, 768),        # more tiles than workgroups: 288, the last 32 walked as 64 halves of 256x128
    (25500, 760, 256, True, 0.1, 0),          # 300 tiles, 44 as halves; ragged rows, and right halves that end inside a 32-column wave tile
    (25600, 640, 256, False, 0.0, 0),         # ... right halves of the last tile column wholly past N
])
@pytest.mark.parametrize("pmode", [10, 11, 12])       # 256x256 and 256x128 tiles of the ring kernel; the pipelined persistent kernel
def test_persistent_standard_epilogue(gemm_mode, pmode, M, N, K, relu, p, scale_n):
    """The persistent kernels' bias / column-scale / ReLU / dropout epilogue (modes 10 / 11 / 12 force them; 12 leaves a
    reduction that is not a multiple of 64 to the others) against the tile kernels (mode 4) and torch: same dropout mask
    element for element, values within bf16 rounding."""
    A = rnd(M, K, dtype=torch.bfloat16, seed=95)
    B = rnd(N, K, dtype=torch.bfloat16, scale=2.0 / math.sqrt(K), seed=96)
    bias = rnd(N, seed=97)
    drop = ops.dropout_keys(5, 6, 7, p) if p > 0 else _lib.NO_DROPOUT
    kw = dict(bias=bias, act=_lib.ACT_RELU if relu else _lib.ACT_NONE, drop=drop, col_scale=0.25, col_scale_n=scale_n)
    outs = []
    for mode in (pmode, 4):
        out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        gemm_mode(mode)
        ops.gemm(A, B, out, **kw)
        gemm_mode(0)
        outs.append(out.float())
    z = A.float() @ B.float().t() + bias
    if scale_n:
        z[:, :scale_n] *= 0.25
    if relu:
        z = torch.relu(z)
    kept = outs[1] != 0 if p > 0 else torch.ones_like(z, dtype=torch.bool)
    ref = torch.where(kept, z / (1.0 - p), torch.zeros_like(z))
    assert bool(torch.isfinite(outs[0]).all())
    if p > 0:
        # the two kernels regenerate the same mask: zeros in the same places wherever the value itself is not zero
        nz = ref != 0
        assert torch.equal((outs[0] != 0) & nz, (outs[1] != 0) & nz)
        assert abs(float(kept.float().mean()) - (1.0 - p) * (float((z != 0).float().mean()) if relu else 1.0)) < 0.02
    close(outs[0], ref, 1.5e-2, "persistent vs torch")
    close(outs[0], outs[1], 1e-2, "persistent vs tile kernel")


@pytest.mark.parametrize("mode", [10, 11, 12])        # 256x256 and 256x128 tiles of the persistent ring kernel; the pipelined one
@pytest.mark.parametrize("M,N,K,p", [
    (515, 768, 768, 0.1),            # out_proj shape: three column tiles, ragged rows
    (1000, 260, 2048, 0.0),          # N a multiple of 4 only, deep reduction (linear2)
    (8192, 768, 2048, 0.1),          # more tiles than workgroups
    (25500, 760, 256, 0.1),          # 300 tiles of 256x256, the last 44 walked as halves; ragged rows and columns
])
def test_persistent_residual_epilogue(gemm_mode, mode, M, N, K, p):
    """y = residual + dropout(x W^T + b) in f32 from the persistent kernel against the tile kernels and torch."""
    A = rnd(M, K, dtype=torch.bfloat16, seed=101)
    B = rnd(N, K, dtype=torch.bfloat16, scale=2.0 / math.sqrt(K), seed=102)
    bias, res = rnd(N, seed=103), rnd(M, N, seed=104)
    drop = ops.dropout_keys(8, 9, 10, p) if p > 0 else _lib.NO_DROPOUT
    outs = []
    for md in (mode, 4):
        out = torch.full((M, N), float("nan"), device=DEV)
        gemm_mode(md)
        ops.gemm(A, B, out, bias=bias, drop=drop, residual=res)
        gemm_mode(0)
        outs.append(out)
    z = A.float() @ B.float().t() + bias
    kept = (outs[1] - res).abs() > 1e-7 if p > 0 else torch.ones_like(z, dtype=torch.bool)
    ref = res + torch.where(kept, z / (1.0 - p), torch.zeros_like(z))
    assert bool(torch.isfinite(outs[0]).all())
    if p > 0:
        assert torch.equal((outs[0] - res).abs() > 1e-7, kept)       # the same mask, element for element
    close(outs[0], ref, 1.5e-2, "persistent vs torch")
    close(outs[0], outs[1], 1e-3, "persistent vs tile kernel")


@pytest.mark.parametrize("M,N,K,what", [
    (20000, 2304, 768, "fwd"),        # 79 x 9 = 711 tiles: every workgroup walks two or three, the last row tile ragged
    (20000, 2304, 768, "fwd_image"),  # ... with the weight read as its k-panel image
    (20000, 768, 2304, "dgrad"),      # data gradient dY W: reduction-major second operand, 72 k-steps
    (20000, 2048, 768, "dgrad"),
    (300, 256, 256, "dgrad"),         # one tile column, the minimum of 8 k-steps, fewer tiles than workgroups
])
def test_pipelined_persistent_kernel(gemm_mode, M, N, K, what):
    """The persistent pipelined kernel (mode 12 forces it) on the shapes it is routed: forward epilogues on a row-major
    weight or its k-panel image, and plain data gradients on a reduction-major weight, against torch f32 products of the
    same bf16 operands and against the tile kernels (mode 4); twice in a row (the ring state of a launch must not leak)."""
    A = rnd(M, K, dtype=torch.bfloat16, seed=111)
    drop = ops.dropout_keys(3, 4, 5, 0.1)
    make = lambda: torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)   # noqa: E731
    if what == "dgrad":
        W = rnd(K, N, dtype=torch.bfloat16, scale=2.0 / math.sqrt(K), seed=112)       # [out = K of the product, in = N]
        ref = A.float() @ W.float()
        run = lambda out: ops.gemm(A, W, out, M=M, N=N, K=K, b_kmajor=True)             # noqa: E731
    else:
        W = rnd(N, K, dtype=torch.bfloat16, scale=2.0 / math.sqrt(K), seed=112)
        bias = rnd(N, seed=113)
        z = A.float() @ W.float().t() + bias
        z[:, :768] *= 0.5
        z = torch.relu(z)
        panels = ops.pack_kpanels(W, torch.empty(ops.kpanel_shape(N, K), device=DEV, dtype=torch.bfloat16)) if what == "fwd_image" else None
        run = lambda out: ops.gemm(A, W, out, bias=bias, act=_lib.ACT_RELU, drop=drop, col_scale=0.5, col_scale_n=768, b_panels=panels)   # noqa: E731
    outs = []
    for mode in (12, 12, 4):
        out = make()
        gemm_mode(mode)
        run(out)
        gemm_mode(0)
        outs.append(out.float())
    assert bool(torch.isfinite(outs[0]).all())
    assert torch.equal(outs[0], outs[1])
    if what == "dgrad":
        close(outs[0], ref, 1e-2, "pipelined persistent vs torch")
    else:
        kept = outs[2].abs() > 1e-7
        ref = torch.where(kept, z / 0.9, torch.zeros_like(z))
        nz = z != 0
        assert torch.equal((outs[0].abs() > 1e-7) & nz, kept & nz)     # the same dropout mask, element for element
        close(outs[0], ref, 1.5e-2, "pipelined persistent vs torch")
    close(outs[0], outs[2], 1e-2, "pipelined persistent vs tile kernel")


@pytest.mark.parametrize("mode", [8, 9])              # the two tile geometries of the persistent forward kernel
@pytest.mark.parametrize("B_,L,V,d,with_bias", [
    (5, 8, 1111, 256, True),            # one ragged tile in M, ragged in N
    (37, 9, 2000, 288, False),          # 333 rows: two M-tiles, the second ragged; no bias; 9 k-steps
    (1, 3, 130, 256, True),             # fewer rows than one MFMA block, fewer columns than one tile
    (64, 9, 50260, 768, True),          # more tiles than workgroups: every workgroup walks several
])
def test_persistent_logits_ce_forward(gemm_mode, mode, B_, L, V, d, with_bias):
    """The persistent bf16 forward (packed max/argmax, numerators stored from registers, the label's logit formed by
    the finalize pass) against torch, with and without kept numerators, and as the argmax entry point."""
    M = B_ * L
    h = rnd(M, d, dtype=torch.bfloat16, seed=71)
    w = rnd(V, d, dtype=torch.bfloat16, scale=2.0 / math.sqrt(d), seed=72)
    b = rnd(V, seed=73) if with_bias else None
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(74)).to(DEV)
    labels[0, 0], labels[-1, -1] = V - 1, 0
    z = h.float() @ w.float().t() + (b if with_bias else 0.0)
    lab_rows = labels.t().reshape(-1)
    ref_lse = torch.logsumexp(z, 1)
    ref_loss = (ref_lse - z.gather(1, lab_rows[:, None])[:, 0]).sum()
    ref_dl = (torch.softmax(z, 1) - torch.nn.functional.one_hot(lab_rows, V)) / M
    top2 = z.topk(2, -1)
    decided = (top2.values[:, 0] - top2.values[:, 1]) > 0.05
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    Vp = (V + 127) // 128 * 128
    bws = torch.empty(ops.logits_ce_bwd_workspace(M, V), dtype=torch.uint8, device=DEV)
    for keep in (True, False):
        lse, amax = torch.full((M,), float("nan"), device=DEV), torch.full((M,), -1, dtype=torch.int32, device=DEV)
        ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
        dl = torch.full((M, Vp), float("nan"), device=DEV, dtype=torch.bfloat16) if keep else None
        gemm_mode(mode)
        ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, probs=dl)
        gemm_mode(0)
        close(lse, ref_lse, 2e-3, "lse")
        assert abs(ls.item() - ref_loss.item()) <= 2e-3 * abs(ref_loss.item())
        assert torch.equal(amax[decided].long(), top2.indices[decided, 0])
        assert int(amax.min()) >= 0 and int(amax.max()) < V
        if keep:
            assert float(dl.float().min()) >= 0.0 and float(dl.float().max()) <= 1.0 + 1e-5     # also: no NaN left
            assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
            dbias = torch.full((V,), float("nan"), device=DEV)
            ops.logits_ce_bwd_from_probs(dl, V, labels, B_, L, lse, 1.0 / M, ws, dbias, bws)
            close(dl[:, :V], ref_dl, 1e-2, "dl from kept numerators")
            close(dbias, ref_dl.sum(0), 1e-2, "dbias")
    ids = torch.full((M,), -1, dtype=torch.int64, device=DEV)
    gemm_mode(mode)
    ops.logits_argmax(h, w, b, ids, None, ws)
    gemm_mode(0)
    assert torch.equal(ids[decided], top2.indices[decided, 0])
    assert int(ids.min()) >= 0 and int(ids.max()) < V


@pytest.mark.parametrize("mode", [8, 9])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_persistent_argmax_tie_rule(gemm_mode, mode, sign):
    """The packed argmax of the persistent kernel: exactly equal logits resolve to the lowest column for maxima of either
    sign (torch.argmax's rule) -- inside a lane, across the lanes of a row, across 64-column slots --, and two logits
    closer than 64 ulp resolve to one of the two."""
    M, V, d = 48, 300, 256
    h = rnd(M, d, dtype=torch.bfloat16, seed=81)
    w = rnd(V, d, dtype=torch.bfloat16, scale=0.5 / math.sqrt(d), seed=82)
    pairs = [(5, 6), (5, 37), (17, 201), (70, 127), (130, 299), (0, 63)]
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    for ulps in (0, 20):
        for c1, c2 in pairs:
            w2 = w.clone()
            w2[c2] = w2[c1]                                   # same operands, same order of summation: equal bit for bit
            # all other logits far below the pair; the pair's level positive or negative
            b = torch.full((V,), -100.0 if sign > 0 else -300.0, device=DEV)
            b[c1] = b[c2] = 50.0 * sign
            if ulps:
                b[c1] = torch.nextafter(b[c1], torch.tensor(-1e9, device=DEV))   # the LOWER column is smaller by a few ulp of the logit
            ids = torch.full((M,), -1, dtype=torch.int64, device=DEV)
            gemm_mode(mode)
            ops.logits_argmax(h, w2, b, ids, None, ws)
            gemm_mode(0)
            if ulps == 0:
                assert bool((ids == c1).all()), (sign, c1, c2, ids[:8].tolist())
            else:
                assert bool(((ids == c1) | (ids == c2)).all()), (sign, c1, c2, ids[:8].tolist())


@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,K,extra", [(1, 32, 0), (37, 288, 3), (8192, 768, 0), (50260, 768, 44), (300, 2048, 0)])
def test_pack_kpanels(src_dtype, rows, K, extra):
    """k-panel image [K/32][rows][32] of a matrix (with rows to spare per panel, which must stay untouched) and the
    row-major cast the same pass can leave; source with a leading dimension larger than K."""
    ld = K + (8 if rows < 1000 else 0)
    src = rnd(rows, ld, dtype=src_dtype, seed=301)[:, :K]
    panels = torch.full(ops.kpanel_shape(rows + extra, K), 7.0, device=DEV, dtype=torch.bfloat16)
    rm = torch.full((rows, K), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.pack_kpanels(src, panels, rowmajor=rm)
    want = src.to(torch.bfloat16)
    assert torch.equal(rm, want)
    assert torch.equal(panels[:, :rows], want.view(rows, K // 32, 32).permute(1, 0, 2))
    assert extra == 0 or bool((panels[:, rows:] == 7.0).all())
    panels2 = torch.empty_like(panels)
    ops.pack_kpanels(src, panels2)
    assert torch.equal(panels2[:, :rows], panels[:, :rows])


@pytest.mark.parametrize("mode", [8, 9])
@pytest.mark.parametrize("B_,L,V,d,extra", [(5, 8, 1111, 256, 0), (37, 9, 2000, 288, 5), (1, 3, 130, 256, 1), (64, 9, 50260, 768, 0)])
def test_logits_ce_forward_from_k_panels(gemm_mode, mode, B_, L, V, d, extra):
    """The persistent forward reading k-panel images of h, of W or of both gives bit for bit what it gives on the row-major
    operands: every entry point that takes them (training forward in both forms, greedy argmax)."""
    M = B_ * L
    h = rnd(M, d, dtype=torch.bfloat16, seed=271)
    w = rnd(V, d, dtype=torch.bfloat16, scale=2.0 / math.sqrt(d), seed=272)
    b = rnd(V, seed=273)
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(274)).to(DEV)
    # images with rows to spare per panel, filled with NaN: nothing past the matrix may be read as data
    hp = torch.full(ops.kpanel_shape(M + extra, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    wp = torch.full(ops.kpanel_shape(V + extra, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.pack_kpanels(h, hp)
    ops.pack_kpanels(w, wp)
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    Vp = (V + 127) // 128 * 128

    def run(kind, hpan, wpan):
        lse, amax = torch.full((M,), float("nan"), device=DEV), torch.full((M,), -1, dtype=torch.int32, device=DEV)
        ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
        dl = torch.full((M, Vp), float("nan"), device=DEV, dtype=torch.bfloat16)
        rowref = torch.full((M,), float("nan"), device=DEV)
        ids = torch.full((M,), -1, dtype=torch.int64, device=DEV)
        gemm_mode(mode)
        if kind == 0:
            ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, probs=dl, h_panels=hpan, w_panels=wpan)
        elif kind == 1:
            ops.logits_ce_fwd_ref(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, dl, rowref, h_panels=hpan, w_panels=wpan)
        else:
            ops.logits_argmax(h, w, b, ids, None, ws, h_panels=hpan, w_panels=wpan)
        gemm_mode(0)
        return [ids] if kind == 2 else [lse, amax, ls, cr, dl] + ([rowref] if kind == 1 else [])

    for kind in (0, 1, 2):
        base = run(kind, None, None)
        assert not bool(torch.isnan(base[0].float()).any()) and int(base[0 if kind == 2 else 1].min()) >= 0
        for hpan, wpan in ((hp, None), (None, wp), (hp, wp)):
            got = run(kind, hpan, wpan)
            for x, y in zip(got, base):
                assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else torch.nan_to_num(x),
                                   y.view(torch.int16) if y.dtype == torch.bfloat16 else torch.nan_to_num(y))


@pytest.mark.parametrize("B_,L,V,d,extra,hot_rows", [
    (8, 32, 2000, 256, 0, 0),          # one M-tile, ragged vocabulary
    (37, 32, 1111, 288, 3, 0),         # 1184 rows: ragged last M-tile, rows to spare per panel
    (64, 32, 50260, 768, 0, 0),        # the benchmark's vocabulary: tail split of the weight gradient, split-K data gradient
    (16, 32, 5000, 256, 0, 5),         # rows redone against their maximum write into the image too
])
def test_numerators_as_column_panel_image(B_, L, V, d, extra, hot_rows):
    """capdec_logits_ce_fwd_ref writing the numerators as a column-panel image [Vp/32][R][32], the patch of
    capdec_logits_ce_bwd_ref_prepare in that layout, and both backward products reading the image (A row-major for the data
    gradient, reduction-major for the weight gradient with the bias column) -- against the same calls on the matrix."""
    M = B_ * L
    if M % 32:
        pytest.skip("the weight gradient reads the image along its rows: whole 32-row steps only")
    h = rnd(M, d, dtype=torch.bfloat16, seed=371)
    if hot_rows:
        h[:hot_rows] *= 40.0
    w = rnd(V, d, dtype=torch.bfloat16, scale=2.0 / math.sqrt(d), seed=372)
    b = rnd(V, seed=373)
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(374)).to(DEV)
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    Vp = (V + 127) // 128 * 128
    R = M + extra
    outs = []
    for img in (False, True):
        lse, amax = torch.full((M,), float("nan"), device=DEV), torch.full((M,), -1, dtype=torch.int32, device=DEV)
        ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
        dl = torch.full((Vp // 32, R, 32) if img else (M, Vp), float("nan"), device=DEV, dtype=torch.bfloat16)
        rowref = torch.full((M,), float("nan"), device=DEV)
        ops.logits_ce_fwd_ref(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, dl, rowref)
        as_matrix = (lambda t: t[:, :M].permute(1, 0, 2).reshape(M, Vp)) if img else (lambda t: t)
        num = as_matrix(dl).clone()
        hs = torch.full((M, d), float("nan"), device=DEV, dtype=torch.bfloat16)
        rowscale = torch.full((M,), float("nan"), device=DEV)
        rowscale_lp = torch.full((M,), float("nan"), device=DEV, dtype=torch.bfloat16)
        ops.logits_ce_bwd_ref_prepare(dl, V, h, labels, B_, L, lse, rowref, 1.0 / M, hs, rowscale, rowscale_lp)
        patched = as_matrix(dl).clone()
        if img and extra:
            assert bool(torch.isnan(dl[:, M:].float()).all())                     # the spare rows of every panel are untouched
        hws = torch.empty(8 * M * d * 4, dtype=torch.uint8, device=DEV)
        gW, gb = torch.full((V, d), float("nan"), device=DEV), torch.full((V,), float("nan"), device=DEV)
        dH = torch.full((M, d), float("nan"), device=DEV)
        wpad = torch.zeros(Vp, d, device=DEV, dtype=torch.bfloat16)
        wpad[:V] = w
        ops.gemm(dl, hs, gW, M=V, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=hws, kvec=rowscale_lp, kvec_out=gb, a_panels=img)
        if img:
            ops.gemm(dl, wpad, dH, M=M, N=d, K=Vp, b_kmajor=True, workspace=hws, row_scale=rowscale, a_panels=True)
        else:
            ops.gemm(dl, wpad, dH, M=M, N=d, K=Vp, b_kmajor=True, workspace=hws, row_scale=rowscale)
        outs.append((lse, amax, ls, cr, rowref, num, patched, hs, rowscale, gW, gb, dH))
    a, c = outs
    for i in range(9):
        assert torch.equal(a[i].view(torch.int16) if a[i].dtype == torch.bfloat16 else a[i], c[i].view(torch.int16) if c[i].dtype == torch.bfloat16 else c[i]), i
    assert not bool(torch.isnan(a[5].float()).any())
    close(c[9], a[9], 1e-3, "weight gradient from the image")
    close(c[10], a[10], 1e-3, "bias column from the image")
    close(c[11], a[11], 1e-3, "data gradient from the image")


@pytest.mark.parametrize("akm", [False, True])
@pytest.mark.parametrize("bkm", [False, True])
@pytest.mark.parametrize("M,N,K,extra,with_ws", [
    (264, 136, 96, 0, False),        # ragged last tile in both directions
    (520, 72, 2080, 5, True),        # deep K through the workspace (split-K), rows to spare per panel
    (33, 40, 64, 0, False),          # less than one tile
    (1000, 768, 4096, 0, True),
])
def test_gemm_reads_a_column_panel_image(akm, bkm, M, N, K, extra, with_ws):
    """capdec_gemm_desc.a_panel_rows: A given as the column-panel image [cols/32][R][32] of its stored matrix, in both
    orientations and with either layout of B, against the same product on the matrix (same kernel, forced)."""
    A = rnd(K, M, dtype=torch.bfloat16, seed=401) if akm else rnd(M, K, dtype=torch.bfloat16, seed=401)
    B = rnd(K, N, dtype=torch.bfloat16, seed=402) if bkm else rnd(N, K, dtype=torch.bfloat16, seed=402)
    rows, cols = A.shape
    img = torch.full(((cols + 31) // 32, rows + extra, 32), float("nan"), device=DEV, dtype=torch.bfloat16)
    padded = torch.zeros(rows, img.shape[0] * 32, device=DEV, dtype=torch.bfloat16)
    padded[:, :cols] = A
    img[:, :rows] = padded.view(rows, -1, 32).permute(1, 0, 2)
    ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device=DEV) if with_ws else None
    scale = rnd(M, seed=403)
    want = scale[:, None] * ((A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t()))
    got = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(img, B, got, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, workspace=ws, row_scale=scale, a_panels=True)
    close(got, want, 1.5e-2, "product from the image")


def test_pack_kpanels_batched_and_gemm_b_image(gemm_mode):
    """Images of several equally shaped matrices of a flat buffer in one launch; a large row-major product given the image
    of its B returns the same bits, on both geometries of the persistent kernel, with either epilogue, and by default."""
    rows, K, batch, stride, first = 200, 96, 3, 200 * 96 + 64, 128
    flat = rnd(first + batch * stride, dtype=torch.bfloat16, seed=501)
    panels = torch.full((batch,) + ops.kpanel_shape(rows, K), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.pack_kpanels_batched(flat, first, rows, K, batch, stride, panels)
    for z in range(batch):
        mat = flat[first + z * stride: first + z * stride + rows * K].view(rows, K)
        assert torch.equal(panels[z], mat.view(rows, K // 32, 32).permute(1, 0, 2))
    for M, N, K2, mode in ((515, 640, 768, 10), (515, 640, 768, 11), (8192, 2048, 768, 0)):
        x = rnd(M, K2, dtype=torch.bfloat16, seed=502)
        w = rnd(N, K2, dtype=torch.bfloat16, scale=0.05, seed=503)
        b = rnd(N, seed=504)
        res = rnd(M, N, seed=505)
        wp = ops.pack_kpanels(w, torch.empty(ops.kpanel_shape(N + 3, K2), device=DEV, dtype=torch.bfloat16).fill_(float("nan")))
        drop = ops.dropout_keys(7, 1, 2, 0.1)
        outs = []
        for img in (None, wp):
            y1 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
            y2 = torch.full((M, N), float("nan"), device=DEV)
            gemm_mode(mode)
            ops.gemm(x, w, y1, bias=b, act=_lib.ACT_RELU, drop=drop, b_panels=img)
            ops.gemm(x, w, y2, bias=b, drop=drop, residual=res, b_panels=img)
            gemm_mode(0)
            outs.append((y1, y2))
        assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1])
        assert not bool(torch.isnan(outs[0][1]).any())


@pytest.mark.parametrize("mode", [8, 9])
@pytest.mark.parametrize("B_,L,V,d,with_bias,hot_rows", [
    (5, 8, 1111, 256, True, 0),
    (37, 9, 2000, 288, False, 0),
    (1, 3, 130, 256, True, 0),
    (64, 9, 50260, 768, True, 0),
    (37, 9, 2000, 288, True, 7),         # rows whose label is (nearly) impossible: redone against their maximum
])
def test_logits_ce_row_reference(gemm_mode, mode, B_, L, V, d, with_bias, hot_rows):
    """Forward with the numerators kept against one reference per row (the label's logit), and the backward built on it:
    d(logits) = rowscale * numerators, the bias gradient as weighted column sums, both products -- against torch."""
    M = B_ * L
    h = rnd(M, d, dtype=torch.bfloat16, seed=171)
    if hot_rows:
        h[:hot_rows] *= 40.0                                                       # logits spread over +-100: label far below the maximum
    w = rnd(V, d, dtype=torch.bfloat16, scale=2.0 / math.sqrt(d), seed=172)
    b = rnd(V, seed=173) if with_bias else None
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(174)).to(DEV)
    labels[0, 0], labels[-1, -1] = V - 1, 0
    z = h.float() @ w.float().t() + (b if with_bias else 0.0)
    lab_rows = labels.t().reshape(-1)
    zl = z.gather(1, lab_rows[:, None])[:, 0]
    ref_lse = torch.logsumexp(z, 1)
    ref_loss = (ref_lse - zl).sum()
    ref_dl = (torch.softmax(z, 1) - torch.nn.functional.one_hot(lab_rows, V)) / M
    top2 = z.topk(2, -1)
    decided = (top2.values[:, 0] - top2.values[:, 1]) > 0.05 * top2.values[:, 0].abs().clamp(min=1.0)
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    Vp = (V + 127) // 128 * 128
    lse, amax = torch.full((M,), float("nan"), device=DEV), torch.full((M,), -1, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    dl = torch.full((M, Vp), float("nan"), device=DEV, dtype=torch.bfloat16)
    rowref = torch.full((M,), float("nan"), device=DEV)
    gemm_mode(mode)
    ops.logits_ce_fwd_ref(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, dl, rowref)
    gemm_mode(0)
    close(lse, ref_lse, 2e-3, "lse")
    assert abs(ls.item() - ref_loss.item()) <= 2e-3 * abs(ref_loss.item())
    assert torch.equal(amax[decided].long(), top2.indices[decided, 0])
    assert int(amax.min()) >= 0 and int(amax.max()) < V
    assert cr.item() == int((amax.long() == lab_rows).sum())
    far = (z.max(1).values - zl) > 50.0
    assert int(far.sum()) >= hot_rows // 2 if hot_rows else int(far.sum()) == 0       # the hot rows (most of them) are out of range
    want_ref = torch.where(far, z.max(1).values, zl)
    ok_rows = ((z.max(1).values - zl) - 50.0).abs() > 1.0                         # not within rounding of the switch point
    close(rowref[ok_rows], want_ref[ok_rows], 2e-3, "row reference")
    assert not bool(torch.isnan(dl.float()).any()) and float(dl.float().min()) >= 0.0
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
    num = torch.exp(z - rowref[:, None])
    rel = ((dl[:, :V].float() - num).abs() / (num + 1e-30)).max().item()
    assert rel <= 2e-2, f"numerators: max relative error {rel:.3e}"
    # backward pieces
    hs = torch.full((M, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    rowscale = torch.full((M,), float("nan"), device=DEV)
    rowscale_lp = torch.full((M,), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.logits_ce_bwd_ref_prepare(dl, V, h, labels, B_, L, lse, rowref, 1.0 / M, hs, rowscale, rowscale_lp)
    assert torch.equal(rowscale_lp, rowscale.to(torch.bfloat16))
    close(rowscale[:, None] * dl[:, :V].float(), ref_dl, 1e-2, "d(logits) = rowscale * numerators")
    close(hs, rowscale[:, None] * h.float(), 1e-2, "scaled head input")
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
    dbias = torch.full((V,), float("nan"), device=DEV)
    ops.colsum_weighted(dl[:, :V], rowscale, dbias)
    close(dbias, ref_dl.sum(0), 1e-2, "dbias")
    gW = torch.full((V, d), float("nan"), device=DEV)
    hws = torch.empty(8 * M * d * 4, dtype=torch.uint8, device=DEV)
    ops.gemm(dl, hs, gW, M=V, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=hws)
    close(gW, ref_dl.t() @ h.float(), 1e-2, "weight gradient")
    # the bias gradient as one more column of that product
    gW2, dbias2 = torch.full((V, d), float("nan"), device=DEV), torch.full((V,), float("nan"), device=DEV)
    ops.gemm(dl, hs, gW2, M=V, N=d, K=M, a_kmajor=True, b_kmajor=True, workspace=hws, kvec=rowscale_lp, kvec_out=dbias2)
    assert torch.equal(gW2, gW)
    close(dbias2, ref_dl.sum(0), 1e-2, "dbias from the weight-gradient product")
    wp = torch.zeros(Vp, d, device=DEV, dtype=torch.bfloat16)
    wp[:V] = w
    for use_ws in (True, False):                                                   # row factor in the split-K fold / as a pass of its own
        dH = torch.full((M, d), float("nan"), device=DEV)
        ops.gemm(dl, wp, dH, M=M, N=d, K=Vp, b_kmajor=True, workspace=hws if use_ws else None, row_scale=rowscale)
        close(dH, ref_dl @ w.float(), 1e-2, "data gradient")


@pytest.mark.parametrize("M,N,K,with_ws", [
    (8192, 2048, 2048, False),          # 256 tiles of 256x256, one launch: the pipelined kernel forms the vector product itself
    (1024, 1024, 16384, True),          # 16 tiles, split-K through the workspace: slabs of the vector product, summed afterwards
    (50260, 768, 8192, True),           # the vocabulary weight gradient: whole rounds + a split-K tail, ragged last tile
    (8192, 2048, 16384, False),         # k range too long for the LDS behind the ring: separate pass
    (200, 328, 520, False),             # small: separate pass
])
def test_gemm_vector_column(M, N, K, with_ws):
    """kvec_out = A . kvec next to a plain reduction-major product: same C as without it, the vector product to bf16 operand
    precision."""
    A = rnd(K, (M + 127) // 128 * 128, dtype=torch.bfloat16, seed=90)[:, :M]          # rows 16-byte aligned, like the numerators
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.1, seed=91)
    v = rnd(K, dtype=torch.bfloat16, seed=92)
    ws = torch.empty(8 * max(M, 4096) * N * 4, dtype=torch.uint8, device=DEV) if with_ws else None
    C0 = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A, B, C0, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, workspace=ws)
    C1 = torch.full((M, N), float("nan"), device=DEV)
    out = torch.full((M,), float("nan"), device=DEV)
    ops.gemm(A, B, C1, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, workspace=ws, kvec=v, kvec_out=out)
    assert torch.equal(C1, C0)
    close(out, A.float().t() @ v.float(), 2e-3, "vector column")
    out2 = torch.full((M,), float("nan"), device=DEV)
    ops.gemm(A, B, C1, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, workspace=ws, kvec=v, kvec_out=out2)
    assert torch.equal(out2, out)                                       # fixed summation order
    # no vector: ones, i.e. the column sums of the reduction-major operand
    out3 = torch.full((M,), float("nan"), device=DEV)
    ops.gemm(A, B, C1, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, workspace=ws, kvec_out=out3)
    assert torch.equal(C1, C0)
    close(out3, A.float().sum(0), 2e-3, "column sums as a vector column")


def test_gemm_vector_column_cuts_deep_products():
    """A product too deep for the vector's k range to fit the kernel (K > 8192) and with too many tiles for split-K to be
    chosen for its own sake: given a workspace it is cut along K so that the vector column still rides along."""
    M, N, K = 8192, 2048, 16384
    A = rnd(K, M, dtype=torch.bfloat16, seed=590)
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.1, seed=591)
    v = rnd(K, dtype=torch.bfloat16, seed=592)
    ws = torch.empty(2 * M * N * 4 + 32 * M * 4 + 64, dtype=torch.uint8, device=DEV)
    C0 = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A, B, C0, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True)
    C1 = torch.full((M, N), float("nan"), device=DEV)
    out = torch.full((M,), float("nan"), device=DEV)
    ops.gemm(A, B, C1, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, workspace=ws, kvec=v, kvec_out=out)
    close(C1, C0, 1e-5, "product in two slabs")
    close(out, A.float().t() @ v.float(), 2e-3, "vector column")


def test_persistent_logits_ce_forward_repeats_exactly(gemm_mode):
    """200 launches of each geometry on one input give bit-identical numerators and partial sums: the operand ring's
    waits are sufficient on every tile of every launch (a stage read before it had landed shows up here)."""
    B_, L, V, d = 64, 32, 50260, 768
    M = B_ * L
    h = rnd(M, d, dtype=torch.bfloat16, seed=81)
    w = rnd(V, d, dtype=torch.bfloat16, scale=0.05, seed=82)
    b = rnd(V, seed=83)
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(84)).to(DEV)
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    Vp = (V + 127) // 128 * 128
    lse, amax = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    dl = torch.empty(M, Vp, device=DEV, dtype=torch.bfloat16)
    for mode in (8, 9):
        gemm_mode(mode)
        ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, probs=dl)
        first = (dl.clone(), lse.clone(), amax.clone())
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for _ in range(200):
            ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, probs=dl)
            bad += (dl.view(torch.int16) != first[0].view(torch.int16)).sum() + (lse != first[1]).sum() + (amax != first[2]).sum()
        gemm_mode(0)
        assert int(bad) == 0, f"mode {mode}: {int(bad)} elements differed between launches"
    # the same with the numerators taken against the label's logit (the training default)
    rowref = torch.empty(M, device=DEV)
    for mode in (8, 9):
        gemm_mode(mode)
        ops.logits_ce_fwd_ref(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, dl, rowref)
        first = (dl.clone(), lse.clone(), amax.clone(), rowref.clone())
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for _ in range(200):
            ops.logits_ce_fwd_ref(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, dl, rowref)
            bad += ((dl.view(torch.int16) != first[0].view(torch.int16)).sum() + (lse != first[1]).sum() + (amax != first[2]).sum()
                    + (rowref != first[3]).sum())
        gemm_mode(0)
        assert int(bad) == 0, f"row reference, mode {mode}: {int(bad)} elements differed between launches"


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B_,L,V,d", [(5, 8, 1111, 64), (16, 32, 50260, 768), (3, 8, 503, 64)])
def test_logits_ce_backward_from_kept_numerators(dtype, tol, B_, L, V, d):
    """Forward keeps exp(z - slot max); the backward turns it into (softmax - onehot)/M in place,
    and must agree with the recomputing backward and with torch."""
    M = B_ * L
    h = rnd(M, d, dtype=dtype, seed=61)
    w = rnd(V, d, dtype=dtype, scale=0.3 if d == 64 else 0.05, seed=62)
    b = rnd(V, seed=63)
    labels = torch.randint(0, V, (B_, L), generator=torch.Generator().manual_seed(64)).to(DEV)
    z = h.float() @ w.float().t() + b
    lab_rows = labels.t().reshape(-1)
    ref_dl = (torch.softmax(z, 1) - torch.nn.functional.one_hot(lab_rows, V)) / M
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    lse, amax = torch.empty(M, device=DEV), torch.empty(M, dtype=torch.int32, device=DEV)
    ls, cr = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    Vp = (V + 127) // 128 * 128
    dl = torch.full((M, Vp), float("nan"), device=DEV, dtype=dtype)
    ops.logits_ce_fwd(h, w, b, labels, B_, L, lse, amax, ls, cr, ws, probs=dl)
    close(lse, torch.logsumexp(z, 1), 2e-3 if dtype == torch.bfloat16 else 1e-5, "lse")
    # numerators in [0, 1] (fma rounding may exceed 1 by an ulp or two), every element written (no NaN left)
    assert float(dl.float().min()) >= 0.0 and float(dl.float().max()) <= 1.0 + 1e-5
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
    dbias = torch.full((V,), float("nan"), device=DEV)
    bws = torch.empty(ops.logits_ce_bwd_workspace(M, V), dtype=torch.uint8, device=DEV)
    ops.logits_ce_bwd_from_probs(dl, V, labels, B_, L, lse, 1.0 / M, ws, dbias, bws)
    close(dl[:, :V], ref_dl, tol, "dl from kept numerators")
    close(dbias, ref_dl.sum(0), tol, "dbias")
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
    dl2 = torch.empty(M, Vp, device=DEV, dtype=dtype)
    ops.logits_ce_bwd_slab(h, w, b, labels, B_, L, lse, 1.0 / M, 0, V, dl2, None, None)
    close(dl[:, :V], dl2[:, :V], tol, "kept numerators vs recomputation")


@pytest.mark.parametrize("M,N,K,out", [(768, 256, 4096, torch.float32), (200, 328, 8200, torch.bfloat16), (128, 128, 2048, torch.float32)])
@pytest.mark.parametrize("mode", [0, 1, 5, 6, 7])
def test_gemm_split_k(gemm_mode, mode, M, N, K, out):
    gemm_mode(mode)
    # weight-gradient shape: few output tiles, deep K; both operands reduction-major
    A = rnd(K, M, dtype=torch.bfloat16, seed=70)
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.1, seed=71)
    ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device=DEV)
    ref = A.float().t() @ B.float()
    got = torch.full((M, N), float("nan"), device=DEV, dtype=out)
    ops.gemm(A, B, got, a_kmajor=True, b_kmajor=True, workspace=ws)
    close(got, ref, 1e-2 if out == torch.bfloat16 else 2e-3, "split-k")
    plain = torch.empty(M, N, device=DEV, dtype=out)
    ops.gemm(A, B, plain, a_kmajor=True, b_kmajor=True)
    close(got, plain, 1e-2 if out == torch.bfloat16 else 1e-5, "split-k vs single pass")
    if out == torch.float32:
        acc = ref.clone()
        ops.gemm(A, B, acc, a_kmajor=True, b_kmajor=True, workspace=ws, accumulate=True)
        close(acc, 2 * ref, 2e-3, "split-k accumulate")


@pytest.mark.parametrize("akm", [True, False])
def test_gemm_tail_split(akm):
    """A plain deep-K product whose 256x256 tiles fill 2.3 rounds of the 256 CUs is cut into the whole rounds and a
    split-K tail when a workspace is given (the vocabulary weight gradient: 50260 x 768 over 8192 rows): same result as
    the single launch, rows on both sides of the cut compared with torch."""
    M, N, K = 50260, 768, 4096
    A = rnd(K, 50304, dtype=torch.bfloat16, seed=121)[:, :M] if akm else rnd(M, K, dtype=torch.bfloat16, seed=121)      # leading dimension 50304, as d(logits)
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.05, seed=122)
    ws = torch.empty(8 * 8192 * N * 4, dtype=torch.uint8, device=DEV)
    one = torch.full((M, N), float("nan"), device=DEV)
    two = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(A, B, one, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=True)
    ops.gemm(A, B, two, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=True, workspace=ws)
    close(two, one, 1e-3, "tail split vs single launch")
    rows = torch.cat([torch.arange(0, 512), torch.arange(43000, 44032), torch.arange(M - 300, M)]).to(DEV)
    Ar = (A.t() if akm else A)[rows].float()
    close(two[rows], Ar @ B.float(), 1.5e-2, "tail split vs torch")


@pytest.mark.parametrize("mode", [0, 4, 6, 7])
def test_head_dgrad_shape_split_k(gemm_mode, mode):
    """The data gradient of the vocabulary projection as the engine launches it: 8192 x 768 over a reduction of 50304
    (the padded vocabulary), W reduction-major, split into 8 slabs.  The refills of every workgroup's last steps reach
    past the end of both operands and rely on their buffer descriptors' byte counts."""
    M, N, K = 8192, 768, 50304
    A = rnd(M, K, dtype=torch.bfloat16, scale=0.05, seed=91)
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.05, seed=92)
    ws = torch.empty(8 * M * N * 4, dtype=torch.uint8, device=DEV)
    ref = A.float() @ B.float()
    out = torch.full((M, N), float("nan"), device=DEV)
    gemm_mode(mode)
    ops.gemm(A, B, out, M=M, N=N, K=K, b_kmajor=True, workspace=ws)
    base = out.clone()
    ops.gemm(A, B, out, M=M, N=N, K=K, b_kmajor=True, accumulate=True, workspace=ws)
    gemm_mode(0)
    close(base, ref, 1.5e-2, f"mode {mode}")
    close(out, 2 * ref, 1.5e-2, f"mode {mode}, accumulated")


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.5e-2), (torch.float32, 3e-5)])
def test_gemm_epilogue_forward(dtype, tol):
    M, N, K = 300, 264, 192
    A, B = rnd(M, K, dtype=dtype, seed=3), rnd(N, K, dtype=dtype, scale=0.1, seed=4)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    drop = ops.dropout_keys(7, 3, 1, p=0.25)
    keep = ops.dropout_mask(drop, M * N, DEV).reshape(M, N)
    assert 0.70 < keep.mean().item() < 0.80
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    aux = torch.empty(M, N, device=DEV, dtype=dtype)
    cs = torch.zeros(N, device=DEV)
    ops.gemm(A, B, out, bias=bias, col_scale=0.5, col_scale_n=100, act=_lib.ACT_RELU, drop=drop, residual=res,
             aux_out=aux, colsum=cs)
    z = A.float() @ B.float().t() + bias
    z[:, :100] *= 0.5
    ref = torch.relu(z) * keep / 0.75 + res
    close(aux, z, tol, "aux")
    close(out, ref, tol, "epilogue")
    close(cs, out.float().sum(0), 2e-3, "colsum")
    # gelu + accumulate into f32
    out2 = res.clone()
    ops.gemm(A, B, out2, bias=bias, act=_lib.ACT_GELU, accumulate=True)
    z2 = A.float() @ B.float().t() + bias
    close(out2, torch.nn.functional.gelu(z2) + res, tol, "gelu+accumulate")


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.5e-2), (torch.float32, 3e-5)])
def test_gemm_gates(dtype, tol):
    M, N, K = 260, 136, 72
    A, B = rnd(M, K, dtype=dtype, seed=8), rnd(K, N, dtype=dtype, scale=0.2, seed=9)     # B stored [K,N]
    saved = torch.relu(rnd(M, N, seed=10)).to(dtype)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm(A, B, out, b_kmajor=True, gate=saved, gate_mode=_lib.GATE_RELU_OUT, gate_drop=_lib.Dropout(0.2, 0, 0))
    ref = (A.float() @ B.float()) * (saved.float() != 0) / 0.8
    close(out, ref, tol, "relu gate")
    z = rnd(M, N, seed=11).to(dtype)
    gd = ops.dropout_keys(1, 2, 3, p=0.1)
    keep = ops.dropout_mask(gd, M * N, DEV).reshape(M, N)
    ops.gemm(A, B, out, b_kmajor=True, gate=z, gate_mode=_lib.GATE_GELU_PRE, gate_drop=gd)
    zf = z.float().requires_grad_(True)
    torch.nn.functional.gelu(zf).sum().backward()
    ref = (A.float() @ B.float()) * zf.grad * keep / 0.9
    close(out, ref, tol, "gelu gate")
    # column sums of the gated product (the bias gradient): atomic adds, or -- with a workspace of one partial row
    # per 64 rows -- a fixed-order two-stage sum; both accumulate into what the buffer holds
    for mode, gate, gdrop, want in ((_lib.GATE_RELU_OUT, saved, _lib.Dropout(0.2, 0, 0), (A.float() @ B.float()) * (saved.float() != 0) / 0.8),
                                    (_lib.GATE_GELU_PRE, z, gd, ref)):
        cs_a = torch.full((N,), 0.5, device=DEV)
        ops.gemm(A, B, out, b_kmajor=True, gate=gate, gate_mode=mode, gate_drop=gdrop, colsum=cs_a)
        ws = torch.empty((M + 63) // 64 * N * 4, dtype=torch.uint8, device=DEV)
        runs = []
        for _ in range(2):
            cs_w = torch.full((N,), 0.5, device=DEV)
            ops.gemm(A, B, out, b_kmajor=True, gate=gate, gate_mode=mode, gate_drop=gdrop, colsum=cs_w, workspace=ws)
            runs.append(cs_w)
        close(out, want, tol, "gated product with column sums")
        close(cs_a, want.sum(0) + 0.5, tol * 4, "column sums (atomic)")
        close(runs[0], want.sum(0) + 0.5, tol * 4, "column sums (two-stage)")
        if dtype == torch.bfloat16:                                   # the two-stage path exists for bf16 output and gate
            assert torch.equal(runs[0], runs[1])


@pytest.mark.parametrize("mode", [0, 4, 7, 12])
def test_gated_dgrad_every_geometry(gemm_mode, mode):
    """The data gradient through linear2 and the ReLU gate as the engine launches it (W reduction-major, gate and dropout
    applied in the epilogue, bias-gradient column sums through the workspace), on the ring and the pipelined kernels."""
    M, N, K = 1000, 2048, 768
    A = rnd(M, K, dtype=torch.bfloat16, seed=111)
    B = rnd(K, N, dtype=torch.bfloat16, scale=0.05, seed=112)
    saved = torch.relu(rnd(M, N, seed=113)).to(torch.bfloat16)
    gd = ops.dropout_keys(4, 5, 6, p=0.1)
    # the saved activation is what came out of ReLU AND dropout: its zeros carry both masks, the keys only the scale
    want = (A.float() @ B.float()) * (saved.float() != 0) / 0.9
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=DEV)
    ws = torch.empty((M + 63) // 64 * N * 4, dtype=torch.uint8, device=DEV)
    gemm_mode(mode)
    ops.gemm(A, B, out, b_kmajor=True, gate=saved, gate_mode=_lib.GATE_RELU_OUT, gate_drop=gd, colsum=cs, workspace=ws)
    gemm_mode(0)
    close(out, want, 1.5e-2, f"mode {mode}")
    close(cs, want.sum(0), 6e-2, f"mode {mode}: column sums")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_kv_scatter(dtype):
    Bb, H, hd, smax, s0, ns = 3, 2, 32, 10, 4, 2
    d = H * hd
    M = ns * Bb
    x = rnd(M, 64, dtype=dtype, seed=12)
    W = rnd(3 * d, 64, dtype=dtype, scale=0.2, seed=13)
    qout = torch.zeros(M, d, device=DEV, dtype=dtype)
    cache = torch.zeros(Bb, H, 2, smax, hd, device=DEV, dtype=dtype)
    ops.gemm(x, W, qout, col_scale=0.25, col_scale_n=d, kv_cache=cache, kv_geom=(d, d, Bb, H, hd, smax, s0))
    y = x.float() @ W.float().t()
    close(qout, y[:, :d] * 0.25, 1.5e-2, "q")
    kv = y[:, d:].reshape(ns, Bb, 2, H, hd).permute(1, 3, 2, 0, 4)          # [B,H,2,ns,hd]
    close(cache[:, :, :, s0:s0 + ns], kv, 1.5e-2, "kv")
    assert cache[:, :, :, :s0].abs().max().item() == 0 and cache[:, :, :, s0 + ns:].abs().max().item() == 0
    # K/V columns only (no query columns)
    cache2 = torch.zeros_like(cache)
    dummy = torch.zeros(M, 2 * d, device=DEV, dtype=dtype)
    ops.gemm(x, W[d:], dummy, kv_cache=cache2, kv_geom=(0, d, Bb, H, hd, smax, s0))
    assert torch.equal(cache2, cache)


# ---------------------------------------------------------------------------- LayerNorm / concat
@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("d", [64, 768])
def test_concat_ln_fwd_bwd(ydt, d):
    B, P, L = 5, 12, 8
    S = P + 1 + L
    img, sos, cap = rnd(B, P, d, seed=20), rnd(B, 1, d, seed=21), rnd(B, L, d, seed=22)
    pos = rnd(S, d, seed=23)
    gamma, beta = 1 + 0.1 * rnd(d, seed=24), 0.1 * rnd(d, seed=25)
    x = torch.empty(S * B, d, device=DEV)
    y = torch.empty(S * B, d, device=DEV, dtype=ydt)
    mean, rstd = torch.empty(S * B, device=DEV), torch.empty(S * B, device=DEV)
    ops.concat_ln_fwd(img, sos, cap, gamma, beta, pos=pos, x_out=x, y_out=y, mean=mean, rstd=rstd)
    xr = (torch.cat([img, sos, cap], 1).transpose(0, 1) + pos[:, None]).reshape(S * B, d)
    assert torch.equal(x, xr)
    yr = torch.nn.functional.layer_norm(xr, (d,), gamma, beta, 1e-5)
    close(y, yr, 1e-2 if ydt == torch.bfloat16 else 2e-6, "ln")
    y2 = torch.empty_like(y)
    ops.ln_fwd(x, gamma, beta, y2, mean, rstd)
    assert torch.equal(y2, y)
    # backward
    dy = rnd(S * B, d, seed=26).to(ydt)
    dx_in = rnd(S * B, d, seed=27)
    dx = torch.empty_like(x)
    dg, db = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    ws = torch.empty(ops.ln_bwd_workspace(S * B, d), dtype=torch.uint8, device=DEV)
    ops.ln_bwd(dy, x, gamma, mean, rstd, dx_in, dx, dg, db, ws)
    xg = xr.clone().requires_grad_(True)
    gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xg, (d,), gg, bg, 1e-5).backward(dy.float())
    close(dx, xg.grad + dx_in, 1e-5, "dx")
    close(dg, gg.grad, 1e-5, "dgamma")
    close(db, bg.grad, 1e-5, "dbeta")
    # affine gradients only: from the kept rows, and from the sources of the join
    dg2, db2 = torch.full((d,), float("nan"), device=DEV), torch.full((d,), float("nan"), device=DEV)
    ops.ln_bwd(dy, x, gamma, mean, rstd, None, None, dg2, db2, ws)
    close(dg2, gg.grad, 1e-5, "dgamma, affine only")
    close(db2, bg.grad, 1e-5, "dbeta, affine only")
    dg3, db3 = torch.full((d,), float("nan"), device=DEV), torch.full((d,), float("nan"), device=DEV)
    ops.concat_ln_bwd_affine(dy, img, sos, cap, gamma, mean, rstd, dg3, db3, ws, pos=pos)
    assert torch.equal(dg3, dg2) and torch.equal(db3, db2)          # same grid, same rows, same order: bit-identical
    # the joined rows below x_row0 are not written
    x2 = torch.full_like(x, float("nan"))
    y3 = torch.empty_like(y)
    ops.concat_ln_fwd(img, sos, cap, gamma, beta, pos=pos, x_out=x2, x_row0=P * B + 2, y_out=y3, mean=mean, rstd=rstd)
    assert torch.equal(y3, y) and torch.equal(x2[P * B + 2:], x[P * B + 2:]) and bool(torch.isnan(x2[:P * B + 2]).all())


# ---------------------------------------------------------------------------- attention
def test_dropout_mask_statistics():
    """The counter-based mask: drop rate p, the four 16-bit fields of one hash equally likely and uncorrelated,
    no correlation along rows or columns of the shapes in use, different keys independent."""
    n = 1 << 22
    for p in (0.1, 0.5):
        keep = ops.dropout_mask(ops.dropout_keys(42, 3, 7, p), n, DEV)
        assert abs((1 - keep.mean().item()) - p) < 2e-3
        for f in range(4):
            assert abs((1 - keep[f::4].mean().item()) - p) < 3e-3
        k = keep - keep.mean()
        for lag in (1, 2, 3, 4, 229, 232, 768, 2048):
            assert abs((k[:-lag] * k[lag:]).mean().item() / k.var().item()) < 5e-3, lag
    a = ops.dropout_mask(ops.dropout_keys(42, 3, 7, 0.1), n, DEV)
    for other in (ops.dropout_keys(42, 3, 8, 0.1), ops.dropout_keys(42, 4, 7, 0.1), ops.dropout_keys(43, 3, 7, 0.1)):
        b = ops.dropout_mask(other, n, DEV)
        assert abs(torch.corrcoef(torch.stack([a, b]))[0, 1].item()) < 5e-3
    assert ops.dropout_mask(ops.dropout_keys(1, 1, 1, 0.0), 1024, DEV).min().item() == 1.0


def torch_attention(qkv, B, H, S, hd, mask, keep, p):
    d = H * hd
    q, k, v = qkv.float().reshape(S, B, 3, H, hd).permute(2, 1, 3, 0, 4)       # [B,H,S,hd] each
    s = q @ k.transpose(-1, -2) + mask
    pr = torch.softmax(s, -1)
    lse = torch.logsumexp(s, -1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    o = (pr @ v).permute(2, 0, 1, 3).reshape(S * B, d)
    return o, lse


def mask_for(S, mode, prefix):
    i = torch.arange(S, device=DEV)[:, None]
    j = torch.arange(S, device=DEV)[None, :]
    if mode == _lib.MASK_NONE:
        vis = torch.ones(S, S, dtype=torch.bool, device=DEV)
    elif mode == _lib.MASK_REFERENCE:
        vis = (j < 12288) | (j <= i)
    else:
        vis = (j < prefix) | (j <= i)
    return torch.zeros(S, S, device=DEV).masked_fill(~vis, float("-inf"))


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B,H,S,hd,mode,prefix,p", [
    (2, 2, 21, 32, _lib.MASK_NONE, 0, 0.0),
    (2, 8, 229, 96, _lib.MASK_REFERENCE, 0, 0.0),
    (3, 2, 70, 96, _lib.MASK_PREFIX_CAUSAL, 37, 0.0),
    (2, 3, 130, 64, _lib.MASK_PREFIX_CAUSAL, 100, 0.1),
    (1, 2, 290, 96, _lib.MASK_NONE, 0, 0.1),
    (1, 1, 200, 128, _lib.MASK_REFERENCE, 0, 0.0),
    (67, 4, 150, 96, _lib.MASK_PREFIX_CAUSAL, 120, 0.1),    # 268 heads: some persistent workgroups walk two, one ragged last round
    (130, 4, 90, 64, _lib.MASK_NONE, 0, 0.0),               # 520 heads: every persistent workgroup walks two or three
])
def test_attention_fwd_bwd(dtype, tol, B, H, S, hd, mode, prefix, p):
    d = H * hd
    qkv = rnd(S * B, 3 * d, dtype=dtype, scale=0.7, seed=30)
    drop = ops.dropout_keys(3, 1, 5, p=p)
    keep = ops.attn_dropout_mask(drop, B, H, S, DEV) if p > 0 else None
    o = torch.full((S * B, d), float("nan"), device=DEV, dtype=dtype)
    lse = torch.empty(B * H * S, device=DEV)
    ops.attn_fwd_packed(qkv, o, lse, B, H, S, hd, mask_mode=mode, prefix_len=prefix, drop=drop)
    qr = qkv.float().requires_grad_(True)
    oref, lref = torch_attention(qr, B, H, S, hd, mask_for(S, mode, prefix), keep, p)
    close(o, oref, tol, "o")
    close(lse.reshape(B, H, S), lref, 1e-2 if dtype == torch.bfloat16 else 1e-5, "lse")
    dout = rnd(S * B, d, dtype=dtype, seed=31)
    oref.backward(dout.float())
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(B * H * S, device=DEV)
    ops.attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, mask_mode=mode, prefix_len=prefix, drop=drop, dq_scale=1.0)
    close(dqkv[:, :d], qr.grad[:, :d], tol * 1.5, "dq")
    close(dqkv[:, d:2 * d], qr.grad[:, d:2 * d], tol * 1.5, "dk")
    close(dqkv[:, 2 * d:], qr.grad[:, 2 * d:], tol * 1.5, "dv")
    if dtype == torch.bfloat16:
        # the same call leaving per-tile column sums of dQ and dK | dV: outputs unchanged, sums = those of the gradients
        rq, rk = ops.attn_bwd_colsum_rows(B, S, S, hd)
        csq, cskv = torch.full((rq, d), float("nan"), device=DEV), torch.full((rk, 2 * d), float("nan"), device=DEV)
        dqkv2 = torch.full_like(qkv, float("nan"))
        ops.attn_bwd_packed(qkv, o, lse, dout, dqkv2, delta, B, H, S, hd, mask_mode=mode, prefix_len=prefix, drop=drop, dq_scale=0.5,
                            colsum_q=csq, colsum_kv=cskv)
        assert torch.equal(dqkv2[:, d:], dqkv[:, d:])
        close(dqkv2[:, :d], 0.5 * qr.grad[:, :d], tol * 1.5, "dq, scaled")
        close(csq.sum(0), 0.5 * qr.grad[:, :d].sum(0), tol, "column sums of dq")
        close(cskv.sum(0), qr.grad[:, d:].sum(0), tol, "column sums of dk | dv")


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("S,hd,nq", [(50, 32, 9), (229, 96, 32), (130, 64, 20), (150, 96, 100), (300, 96, 70)])      # <= 32 queries: V staged in 64-key blocks; the last two: the whole-sequence kernels
def test_attention_query_range(dtype, tol, S, hd, nq):
    # only the last nq queries are computed (scored rows of the last block)
    B, H = 2, 2
    d = H * hd
    qkv = rnd(S * B, 3 * d, dtype=dtype, scale=0.7, seed=32)
    o = torch.zeros(S * B, d, device=DEV, dtype=dtype)
    lse = torch.zeros(B * H * S, device=DEV)
    ops.attn_fwd_packed(qkv, o, lse, B, H, S, hd, q_start=S - nq, nq=nq)
    qr = qkv.float().requires_grad_(True)
    oref, _ = torch_attention(qr, B, H, S, hd, mask_for(S, 0, 0), None, 0.0)
    close(o[(S - nq) * B:], oref[(S - nq) * B:], tol, "o tail")
    assert o[:(S - nq) * B].abs().max().item() == 0
    dout = rnd(S * B, d, dtype=dtype, seed=33)
    dout[:(S - nq) * B] = 0
    oref.backward(dout.float())
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B * H * S, device=DEV)
    ops.attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, q_start=S - nq, nq=nq)
    close(dqkv, qr.grad, tol * 1.5, "dqkv")
    if dtype == torch.bfloat16:
        rq, rk = ops.attn_bwd_colsum_rows(B, S, nq, hd)
        csq, cskv = torch.full((rq, d), float("nan"), device=DEV), torch.full((rk, 2 * d), float("nan"), device=DEV)
        dqkv2 = torch.zeros_like(qkv)
        ops.attn_bwd_packed(qkv, o, lse, dout, dqkv2, delta, B, H, S, hd, q_start=S - nq, nq=nq, colsum_q=csq, colsum_kv=cskv)
        if nq > 32:
            assert torch.equal(dqkv2, dqkv)
        else:
            # up to 32 queries the plain call runs the one-kernel backward and the call with column sums the two kernels:
            # the same sums in another order (dQ over the waves' key ranges, delta over eight threads per row)
            close(dqkv2, dqkv, 1e-2, "one-kernel backward vs the dQ and dK/dV kernels")
            close(dqkv2, qr.grad, tol * 1.5, "dqkv (two kernels)")
        close(csq.sum(0), qr.grad[:, :d].sum(0), tol, "column sums of dq")
        close(cskv.sum(0), qr.grad[:, d:].sum(0), tol, "column sums of dk | dv")


@pytest.mark.parametrize("B,H,S,hd,nq,mode,prefix,p", [
    (3, 2, 229, 96, 32, _lib.MASK_PREFIX_CAUSAL, 198, 0.1),     # the one-block model's scored rows under the prefix-causal mask
    (2, 3, 130, 64, 17, _lib.MASK_PREFIX_CAUSAL, 100, 0.2),     # a ragged second query tile, a ragged last key tile
    (5, 1, 40, 128, 32, _lib.MASK_NONE, 0, 0.1),                # fewer keys than one pass of the four waves; S < q_start + 32 never happens: nq <= S
    (2, 2, 290, 96, 32, _lib.MASK_REFERENCE, 0, 0.0),           # the last block of the stacked model
    (1, 4, 20, 32, 20, _lib.MASK_PREFIX_CAUSAL, 0, 0.0),        # every query scored, purely causal
])
def test_attention_backward_one_kernel(B, H, S, hd, nq, mode, prefix, p):
    """At most 32 scored queries: dQ, dK and dV from one kernel (the default) against torch and against the dQ + dK/dV
    pair (capdecdbg_attn_fused_bwd(0)), under masks and dropout; twice in a row (identical bits: the waves' partial dQ are
    summed in a fixed order)."""
    import ctypes
    lib = _lib.load()
    lib.capdecdbg_attn_fused_bwd.argtypes = [ctypes.c_int]
    lib.capdecdbg_attn_fused_bwd.restype = None
    d = H * hd
    q0 = S - nq
    qkv = rnd(S * B, 3 * d, dtype=torch.bfloat16, scale=0.7, seed=36)
    drop = ops.dropout_keys(3, 1, 5, p=p)
    keep = ops.attn_dropout_mask(drop, B, H, S, DEV) if p > 0 else None
    o = torch.zeros(S * B, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B * H * S, device=DEV)
    ops.attn_fwd_packed(qkv, o, lse, B, H, S, hd, q_start=q0, nq=nq, mask_mode=mode, prefix_len=prefix, drop=drop)
    qr = qkv.float().requires_grad_(True)
    oref, _ = torch_attention(qr, B, H, S, hd, mask_for(S, mode, prefix), keep, p)
    dout = rnd(S * B, d, dtype=torch.bfloat16, seed=37)
    dout[:q0 * B] = 0
    oref.backward(dout.float())
    outs, deltas = [], []
    for fused in (1, 1, 0):
        dqkv = torch.zeros_like(qkv)
        delta = torch.zeros(B * H * S, device=DEV)
        lib.capdecdbg_attn_fused_bwd(fused)
        try:
            ops.attn_bwd_packed(qkv, o, lse, dout, dqkv, delta, B, H, S, hd, q_start=q0, nq=nq, mask_mode=mode, prefix_len=prefix, drop=drop,
                                dq_scale=0.5)
        finally:
            lib.capdecdbg_attn_fused_bwd(1)
        outs.append(dqkv.float())
        deltas.append(delta)
    want = qr.grad.clone()
    want[:, :d] *= 0.5
    assert torch.equal(outs[0], outs[1])
    close(outs[0], want, 3e-2, "one kernel vs torch")
    close(outs[0], outs[2], 1e-2, "one kernel vs the dQ and dK/dV kernels")
    close(deltas[0], deltas[2], 1e-3, "delta")
    if q0 > 0:
        assert outs[0][:q0 * B, :d].abs().max().item() == 0        # rows that are not scored receive no dQ


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("H,hd,smax,length", [(8, 96, 229, 229), (2, 32, 21, 14), (3, 64, 290, 257), (2, 128, 40, 1)])
def test_attention_decode(dtype, tol, H, hd, smax, length):
    B = 5
    q = rnd(B, H * hd, dtype=dtype, scale=0.5, seed=34)
    cache = rnd(B, H, 2, smax, hd, dtype=dtype, seed=35)
    o = torch.empty(B, H * hd, device=DEV, dtype=dtype)
    ops.attn_decode(q, cache, o, B, H, hd, smax, length)
    qf = q.float().reshape(B, H, 1, hd)
    k, v = cache[:, :, 0, :length].float(), cache[:, :, 1, :length].float()
    ref = (torch.softmax(qf @ k.transpose(-1, -2), -1) @ v).reshape(B, H * hd)
    close(o, ref, tol, "decode")


# ---------------------------------------------------------------------------- vocabulary projection + CE
@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B,L,V,d", [(3, 8, 503, 64), (16, 32, 50260, 768)])
def test_logits_ce(dtype, tol, B, L, V, d):
    M = B * L
    h = rnd(M, d, dtype=dtype, seed=40)
    W = rnd(V, d, dtype=dtype, scale=1 / math.sqrt(d), seed=41)
    bias = rnd(V, scale=0.05, seed=42)
    labels = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(43)).to(DEV)
    labels[0, 0], labels[-1, -1] = V - 1, 0
    ws = torch.empty(ops.logits_ce_workspace(M, V), dtype=torch.uint8, device=DEV)
    lse = torch.empty(M, device=DEV)
    amax = torch.empty(M, dtype=torch.int32, device=DEV)
    loss_sum, correct = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int32, device=DEV)
    ops.logits_ce_fwd(h, W, bias, labels, B, L, lse, amax, loss_sum, correct, ws)
    hr = h.float().requires_grad_(True)
    Wr = W.float().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    z = hr @ Wr.t() + br                                     # row m = l*B + b
    tgt = labels.t().reshape(-1)
    loss_ref = torch.nn.functional.cross_entropy(z, tgt)
    close(lse, torch.logsumexp(z, -1), 1e-3 if dtype == torch.bfloat16 else 1e-6, "lse")
    assert abs(loss_sum.item() / M - loss_ref.item()) < (2e-3 if dtype == torch.bfloat16 else 2e-5) * loss_ref.item()
    top2 = z.topk(2, -1)
    decided = (top2.values[:, 0] - top2.values[:, 1]) > (0.05 if dtype == torch.bfloat16 else 1e-4)
    assert torch.equal(amax[decided].long(), top2.indices[decided, 0])
    if dtype == torch.float32:
        assert correct.item() == int((z.argmax(-1) == tgt).sum())
    # argmax entry point
    ids = torch.empty(M, dtype=torch.int64, device=DEV)
    margin = torch.empty(M, device=DEV)
    ops.logits_argmax(h, W, bias, ids, margin, ws)
    assert torch.equal(ids[decided], top2.indices[decided, 0])
    close(margin, top2.values[:, 0] - top2.values[:, 1], 0.1 if dtype == torch.bfloat16 else 1e-3, "margin")
    # backward, slab by slab
    loss_ref.backward()
    dbias = torch.full((V,), float("nan"), device=DEV)
    dH = torch.zeros(M, d, device=DEV)
    dW = torch.empty(V, d, device=DEV)
    slab = 256 if V < 1000 else 8192
    for v0 in range(0, V, slab):
        nv = min(slab, V - v0)
        ld = (nv + 127) // 128 * 128
        dl = torch.full((M, ld), float("nan"), device=DEV, dtype=dtype)
        bws = torch.empty(ops.logits_ce_bwd_workspace(M, nv), dtype=torch.uint8, device=DEV)
        ops.logits_ce_bwd_slab(h, W, bias, labels, B, L, lse, 1.0 / M, v0, nv, dl, dbias, bws)
        dl_ref = (torch.softmax(z.detach(), -1) - torch.nn.functional.one_hot(tgt, V))[:, v0:v0 + nv] / M
        close(dl[:, :nv], dl_ref, 2e-2 if dtype == torch.bfloat16 else 1e-5, "dl")
        assert dl[:, nv:].abs().max().item() == 0 if ld > nv else True
        ops.gemm(dl, W[v0:], dH, M=M, N=d, K=nv, b_kmajor=True, accumulate=True)
        ops.gemm(dl, h, dW[v0:v0 + nv], M=nv, N=d, K=M, a_kmajor=True, b_kmajor=True)
    close(dbias, br.grad, 2e-2 if dtype == torch.bfloat16 else 1e-4, "dbias")
    close(dH, hr.grad, 2e-2 if dtype == torch.bfloat16 else 1e-4, "dH")
    close(dW, Wr.grad, 2e-2 if dtype == torch.bfloat16 else 1e-4, "dW")


@pytest.mark.parametrize("M,d", [(37, 64), (5000, 768), (40000, 768), (300, 2048)])
@pytest.mark.parametrize("dyt,gt,p", [(torch.bfloat16, torch.bfloat16, 0.1), (torch.float32, torch.float32, 0.1),
                                      (torch.bfloat16, torch.bfloat16, 0.0)])
def test_ln_bwd_with_dropout_output(M, d, dyt, gt, p):
    """The LayerNorm backward that also emits the dropout backward of its dx (decoder.py:58 under decoder.py:55) against
    the two separate passes: rows bit-identical, column sums to rounding of a different summation order."""
    x = rnd(M, d, seed=80)
    gamma, beta = 1 + 0.1 * rnd(d, seed=81), 0.1 * rnd(d, seed=82)
    y = torch.empty(M, d, device=DEV, dtype=dyt)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.ln_fwd(x, gamma, beta, y, mean, rstd)
    dy = rnd(M, d, seed=83).to(dyt)
    dx_in = rnd(M, d, seed=84)
    drop = ops.dropout_keys(5, 1, 3, p=p)
    # separate passes
    dx0 = torch.empty_like(x)
    dg0, db0 = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    ws = torch.empty(max(ops.ln_bwd_workspace(M, d), ops.ln_bwd_dropout_workspace(M, d)), dtype=torch.uint8, device=DEV)
    ops.ln_bwd(dy, x, gamma, mean, rstd, dx_in, dx0, dg0, db0, ws)
    g0 = torch.empty(M, d, device=DEV, dtype=gt)
    gb0 = torch.empty(d, device=DEV)
    ops.dropout_bwd_cast(dx0, drop, g0, gb0)
    # one pass; outputs poisoned first
    dx1 = torch.full_like(x, float("nan"))
    dg1, db1, gb1 = (torch.full((d,), float("nan"), device=DEV) for _ in range(3))
    g1 = torch.full((M, d), float("nan"), device=DEV).to(gt)
    ops.ln_bwd_dropout(dy, x, gamma, mean, rstd, dx_in, dx1, dg1, db1, drop, g1, gb1, ws)
    assert torch.equal(dx1, dx0) and torch.equal(g1, g0)
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)          # same grid and fold order as the plain kernel
    close(gb1, gb0, 1e-5, "bias gradient from the fused pass")
    if p > 0:
        assert int((g1 == 0).sum()) > 0.05 * M * d                  # the mask was applied
    # no incoming dx, in place over a dx_in alias, and twice the same bits
    dx2 = dx_in.clone()
    gb2 = torch.empty(d, device=DEV)
    g2 = torch.empty_like(g1)
    ops.ln_bwd_dropout(dy, x, gamma, mean, rstd, dx2, dx2, dg1, db1, drop, g2, gb2, ws)
    assert torch.equal(dx2, dx0) and torch.equal(g2, g0) and torch.equal(gb2, gb1)
    ops.ln_bwd(dy, x, gamma, mean, rstd, None, dx0, dg0, db0, ws)
    ops.ln_bwd_dropout(dy, x, gamma, mean, rstd, None, dx1, dg1, db1, drop, g1, gb1, ws)
    assert torch.equal(dx1, dx0)


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16])
def test_seqfirst_strided_ln_fwd_bwd(in_dtype):
    """LayerNorm-1 straight off a strided dataset slice [:, i:j, :] (decoder.py:163)."""
    S, N, d, i, j = 9, 11, 64, 3, 8
    B = j - i
    full = rnd(S, N, d, seed=70).to(in_dtype)
    tgt = full[:, i:j, :]                                   # seq stride N*d, batch stride d, not contiguous
    assert not tgt.is_contiguous()
    pos = rnd(S, d, seed=71)
    gamma, beta = 1 + 0.1 * rnd(d, seed=72), 0.1 * rnd(d, seed=73)
    xr = (tgt.float() + pos[:, None]).reshape(S * B, d)
    x = torch.full((S * B, d), float("nan"), device=DEV)
    y = torch.empty(S * B, d, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(S * B, device=DEV), torch.empty(S * B, device=DEV)
    ops.seqfirst_ln_fwd(tgt, gamma, beta, pos=pos, x_out=x, x_row0=2 * B, y_out=y, mean=mean, rstd=rstd)
    assert torch.equal(x[2 * B:], xr[2 * B:]) and bool(torch.isnan(x[:2 * B]).all())
    close(y, torch.nn.functional.layer_norm(xr, (d,), gamma, beta, 1e-5), 1e-2, "ln of a strided slice")
    dy = rnd(S * B, d, seed=74).to(torch.bfloat16)
    ws = torch.empty(ops.ln_bwd_workspace(S * B, d), dtype=torch.uint8, device=DEV)
    dg, db = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    ops.seqfirst_ln_bwd_affine(dy, tgt, gamma, mean, rstd, dg, db, ws, pos=pos)
    dg2, db2 = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    ops.ln_bwd(dy, xr.contiguous(), gamma, mean, rstd, None, None, dg2, db2, ws)
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
    bad = rnd(S, N, d, seed=75).as_strided((S, B, d), (N * d + 2, d, 1))           # rows no longer 16-byte aligned
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        ops.seqfirst_ln_fwd(bad, gamma, beta, y_out=torch.empty(S * B, d, device=DEV))


# ---------------------------------------------------------------------------- element-wise
def test_elementwise_and_adam():
    M, N = 130, 72
    dx = rnd(M, N, seed=50)
    drop = ops.dropout_keys(9, 4, 2, p=0.1)
    keep = ops.dropout_mask(drop, M * N, DEV).reshape(M, N)
    g = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    cs = torch.full((N,), float("nan"), device=DEV)                 # column sums are overwritten, not accumulated
    ops.dropout_bwd_cast(dx, drop, g, cs)
    ref = dx * keep / 0.9
    close(g, ref, 1e-2, "dropout_bwd_cast")
    close(cs, ref.sum(0), 1e-5, "colsum of dropout_bwd_cast")
    cs2 = torch.full((N,), float("nan"), device=DEV)
    ops.colsum(g, cs2)
    close(cs2, g.float().sum(0), 1e-5, "colsum")
    # many stripes, a strided view, ragged width; twice the same bits (fixed summation order)
    big = rnd(5000, 350, seed=52).to(torch.bfloat16)
    view = big[7:, 3:340]
    c3, c4 = torch.empty(337, device=DEV), torch.empty(337, device=DEV)
    ops.colsum(view, c3)
    ops.colsum(view, c4)
    close(c3, view.float().sum(0), 1e-5, "colsum of a strided view")
    assert torch.equal(c3, c4)
    with pytest.raises(RuntimeError, match="workspace"):
        ops.colsum(big, torch.empty(350, device=DEV), workspace=torch.empty(64, dtype=torch.uint8, device=DEV))
    b = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.cast(dx, b)
    assert torch.equal(b, dx.to(torch.bfloat16))
    # embedding rows
    table = rnd(40, 64, seed=51)
    ids = torch.tensor([3, 39, 0, 3], device=DEV)
    out = torch.empty(4, 64, device=DEV)
    ops.embed_rows(table, ids, out, pos_row=table[1])
    assert torch.equal(out, table[ids] + table[1])
    # Adam vs torch.optim.Adam (coupled weight decay), 3 steps, odd length
    n = 1003
    w0 = rnd(n, seed=52)
    w = w0.clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    wb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    pw = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([pw], lr=1e-3, weight_decay=0.01)
    for step in range(1, 4):
        gr = rnd(n, seed=60 + step)
        ops.adam_step(w, gr, m, v, step, 1e-3, 0.9, 0.999, 1e-8, 0.01, w_bf16=wb)
        pw.grad = gr.clone()
        opt.step()
    close(w, pw.data, 2e-6, "adam")
    assert torch.equal(wb, w.to(torch.bfloat16))
    # the same update keeps the k-panel image of a matrix inside the buffer current: whole buffer, and slices that cut
    # through the matrix (the data-parallel trainer updates bucket by bucket)
    first, rows, K = 64, 37, 96
    n2 = first + rows * K + 131
    w2, g2 = rnd(n2, seed=70), rnd(n2, seed=71)
    for cuts in ((0, n2), (0, 1000, 2048, n2)):
        ww, m2, v2 = w2.clone(), torch.zeros(n2, device=DEV), torch.zeros(n2, device=DEV)
        pan = torch.full(ops.kpanel_shape(rows + 2, K), 7.0, device=DEV, dtype=torch.bfloat16)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ops.adam_step(ww[lo:hi], g2[lo:hi], m2[lo:hi], v2[lo:hi], 1, 1e-3, 0.9, 0.999, 1e-8, 0.01,
                          panels=pan, panel_first=first - lo, panel_rows=rows)
        mat = ww[first:first + rows * K].view(rows, K).to(torch.bfloat16)
        assert torch.equal(pan[:, :rows], mat.view(rows, K // 32, 32).permute(1, 0, 2))
        assert bool((pan[:, rows:] == 7.0).all())
        assert not torch.equal(ww, w2)
    # greedy bookkeeping
    ids = torch.tensor([5, 7, 9], device=DEV)
    done = torch.tensor([0, 1, 0], dtype=torch.uint8, device=DEV)
    outb = torch.zeros(3, 4, dtype=torch.int64, device=DEV)
    ops.greedy_update(ids, done, outb, 2, 9, 11)
    assert ids.tolist() == [5, 11, 9] and done.tolist() == [0, 1, 1] and outb[:, 2].tolist() == [5, 11, 9]


def test_errors_are_loud():
    A = rnd(16, 12, dtype=torch.bfloat16)      # leading dimension 12 is not a multiple of 8
    B = rnd(16, 12, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.gemm(A, B, torch.empty(16, 16, device=DEV))
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.gemm(A.cpu(), B.cpu(), torch.empty(16, 16))


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("N,K,M", [(2304, 768, 300), (96, 64, 40), (70, 40, 17)])
def test_ln_fold_pair_matches_autograd(dtype, tol, N, K, M):
    """LayerNorm affine folded into the Linear after it: forward identity and, from G = dY^T xhat and s = colsum(dY),
    the gradients autograd gives for the Linear weight and the LayerNorm weight / bias."""
    W32, bias = rnd(N, K, scale=0.1, seed=81), rnd(N, seed=82)
    gamma, beta = 1.0 + 0.3 * rnd(K, seed=83), 0.2 * rnd(K, seed=84)
    W = W32.to(dtype)
    Wf = torch.full((N, K), float("nan"), device=DEV, dtype=dtype)
    bf = torch.full((N,), float("nan"), device=DEV)
    ops.ln_fold_fwd(W, bias, gamma, beta, Wf, bf)
    close(Wf, W.float() * gamma, tol, "folded weight")
    close(bf, bias + W.float() @ beta, 2e-5, "folded bias")
    if dtype == torch.bfloat16:                     # f32 parameter folded straight into a bf16 matrix: one rounding
        Wf2, bf2 = torch.full((N, K), float("nan"), device=DEV, dtype=dtype), torch.full((N,), float("nan"), device=DEV)
        ops.ln_fold_fwd(W32, bias, gamma, beta, Wf2, bf2)
        assert torch.equal(Wf2, (W32 * gamma).to(dtype))
        close(bf2, bias + W32 @ beta, 2e-5, "folded bias from the f32 parameter")
    # backward against autograd on y = (xhat*gamma + beta) W^T + b
    xhat, dY = rnd(M, K, seed=85), rnd(M, N, scale=0.1, seed=86)
    Wp, g, b = W.float().clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ((xhat * g + b) @ Wp.t() + bias).backward(dY)
    G = (dY.t() @ xhat).contiguous()
    s = dY.sum(0).contiguous()
    dgamma, dbeta = torch.full((K,), float("nan"), device=DEV), torch.full((K,), float("nan"), device=DEV)
    ws = torch.empty(ops.ln_fold_bwd_workspace(N, K), dtype=torch.uint8, device=DEV)
    ops.ln_fold_bwd(G, s, W, gamma, beta, dgamma, dbeta, ws)
    close(G, Wp.grad, 2e-5, "weight gradient")
    close(dgamma, g.grad, 2e-5, "dgamma")
    close(dbeta, b.grad, 2e-5, "dbeta")


@pytest.mark.parametrize("rows,cols", [(768, 768), (50260, 768), (2048, 768), (67, 130), (1, 5)])
def test_transpose_bf16(rows, cols):
    src = rnd(rows, cols, dtype=torch.bfloat16, seed=95)
    dst = torch.full((cols, (rows + 7) // 8 * 8), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.transpose(src, dst[:, :rows])
    assert torch.equal(dst[:, :rows], src.t())
    if dst.shape[1] > rows:
        assert bool(torch.isnan(dst[:, rows:].float()).all())          # nothing written past the matrix
    odd = torch.full((cols, rows + 1), float("nan"), device=DEV, dtype=torch.bfloat16)      # odd leading dimension: element-wise path
    ops.transpose(src, odd[:, :rows])
    assert torch.equal(odd[:, :rows], src.t())
