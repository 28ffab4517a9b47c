// GEMM with fused epilogues for the captioning decoder (gfx950).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// bf16 operands, f32 accumulators, MFMA 16x16x32.  Two main loops share the epilogues:
//   * an LDS-DMA ring (operand tiles go global -> LDS without staging registers, counted
//     vmcnt waits): 256x128 tiles with 4 waves of 128x64 where the grid allows it, else
//     128x128 tiles with 4 waves of 64x64; two workgroups per CU either way;
//   * a register-staged 128x128x64 loop for ragged K and the small split-K products.
// Out-of-range reads return 0 through the buffer descriptor, which implements every
// M/N/K tail.  An operand stored reduction-major ([K][rows], the dgrad / wgrad cases) is
// read back with the hardware transposing LDS read (ds_read_b64_tr_b16), so no
// transposed copies of weights or activations are ever made.
// f32 operands: exact-f32 FMA main loop (validation mode), same epilogues.
//
// Every epilogue works on the accumulator tile after it has been staged through a
// wave-private LDS tile, row-major, so global stores are 16 B per lane and contiguous
// across 8-16 lanes.
#include "common.h"
#include "head_fwd.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int CS_LD = 68;                        // floats per row of a wave's 64x64 staging tile
constexpr int CS_WAVE_BYTES = 64 * CS_LD * 4;    // 17408
constexpr int LDS_BYTES = 4 * CS_WAVE_BYTES;     // 69632 >= 2 stages * (16 KiB A + 16 KiB B)
constexpr int STAGE_BYTES = 32768;
constexpr int TILE_BYTES = 16384;

enum { EPI_STD = 0, EPI_CE_FWD = 1, EPI_CE_BWD = 2, EPI_STD_SIDE = 3 };

struct GemmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    int out_dtype;
    const float* bias;
    float col_scale; int col_scale_n;
    int act;
    float drop_scale; uint32_t drop_thr, drop_k0, drop_k1;      // thr 0 = off
    const float* residual; long ldr;
    int accumulate;
    const void* gate; long ldg; int gate_mode; int gate_dtype;
    float gate_scale; uint32_t gate_thr, gate_k0, gate_k1;
    void* aux_out; long ldaux;
    float* colsum;
    void* kv_cache; int kv_q, kv_d, kv_B, kv_H, kv_hd, kv_smax, kv_s0;
    // cross-entropy epilogues
    const int64_t* labels; int ce_B, ce_L; int ce_col0;
    f32x4* part4;                    // CE forward partials [np slots][M rows] of {max, sum exp(z-max), second max, argmax bits}
    float* zlabel; int np; int want_sec;
    int fast_epi;                    // STD: bias/scale/relu/dropout only, vector-aligned output (see epilogue_std_fast)
    const float* lse; float gscale; int store_cols;
    float* colpart; long ld_colpart;  // CE backward: per-(64-row group) column sums, [ceil(M/64)][ld_colpart]
    int split_k; long split_stride;    // split-K: blockIdx.y owns a K range and writes slab C + y*split_stride
    unsigned long long* stamps;      // diagnostic: 8 clock stamps per workgroup, NULL in normal use
    // one more column of a plain product: kvec_out[y * kvec_out_stride + m] = sum over the workgroup's k range of A(m,k) * kvec[k]
    const bf16_t* kvec; float* kvec_out; long kvec_out_stride;
    long a_panel;                    // > 0: A's stored matrix is a column-panel image with this many rows per panel (pipelined 256x256 kernel only)
};

// ------------------------------------------------------------------ tile order
// Blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
// run of logical tile ids (bijective for any grid size).  About 64 consecutive ids
// are resident on an XCD at a time and should share operand panels in its L2:
// with many N-tiles the ids walk groups of 8 M-tiles x all N-tiles (an 8 x 8 block
// is resident); with at most 16 N-tiles they walk N fastest, so that every resident
// M-tile has ALL its N-tiles resident and its A panel -- the large streamed operand
// of the dgrad / wgrad products, d(logits) -- leaves HBM once instead of once per
// residency window.
__device__ __forceinline__ void map_tile(int tm, int tn, int& tile_m, int& tile_n) {
    const int nwg = tm * tn;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GROUP = tn <= 16 ? 1 : 8;
    const int per_group = GROUP * tn;
    const int grp = logical / per_group;
    const int first_m = grp * GROUP;
    const int gsz = min(tm - first_m, GROUP);
    const int in = logical - grp * per_group;
    tile_m = first_m + in % gsz;
    tile_n = in / gsz;
}

// ------------------------------------------------------------------ LDS images
// Row-major operand tile [128 rows][64 k] bf16, 128-B rows; the 16-B chunk index is
// XORed with (row & 7) so that the 16 rows one ds_read_b128 group touches fall on
// 16 different slots of the 256-B bank row.
__device__ __forceinline__ int lds_off_rowmajor(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }
// Reduction-major operand tile [64 k][128 cols] bf16, 256-B rows; 32-B column blocks
// are XORed with a 3-bit function of k that is distinct over the 8 k-rows one
// transposing read touches per half-wave.
__device__ __forceinline__ int kswz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int lds_off_kmajor(int krow, int chunk16) { return krow * 256 + ((chunk16 ^ (kswz(krow) << 1)) << 4); }

template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rb, int ks, int lane) {
    if constexpr (!TR) {
        const int row = rb + (lane & 15);
        const int kc = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(tile + lds_off_rowmajor(row, kc));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int krow = ks * 32 + 8 * g + q;
        const int col = rb + 4 * pp;
        const int off = lds_off_kmajor(krow, col >> 3) + ((col & 7) << 1);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + off));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + off + 4 * 256));
        bf16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// ------------------------------------------------------------------ operand fetch
// A thread moves 4 chunks of 16 B per operand tile: chunk c = tid + 256*i.  The row (or
// k-row) of chunk i is that of chunk 0 plus a constant, so one 32-bit offset per thread
// serves all four loads; the per-i row advance goes into the (scalar) buffer descriptor,
// and the four LDS destinations differ by a compile-time constant.
struct OperandView {
    const bf16_t* base;   // first element of this block's panel
    long remaining;       // elements addressable from base
    long ld;
    int rows_left;        // valid rows (non-reduction index) from the panel start
    long panel = 0;       // 0: a plain matrix; R > 0: the stored matrix [R][cols] lies as a column-panel image [cols / 32][R][32]
};
struct ThreadPlan {
    unsigned voff;        // byte offset of chunk 0 inside a k-tile (0xFFFFFFFF: never in range)
    int lds0;             // LDS byte offset of chunk 0 inside a tile
    int kc8;              // row-major: first k of this thread's chunk inside the k-tile
};

template <bool TR>
__device__ __forceinline__ ThreadPlan make_plan(const OperandView& v, int tid) {
    ThreadPlan t;
    if constexpr (!TR) {
        const int row = tid >> 3, kc = tid & 7;
        t.voff = (unsigned)(((long)row * v.ld + kc * 8) * 2);
        t.lds0 = lds_off_rowmajor(row, kc);               // rows of chunk i: row + 32*i, same (row & 7)
        t.kc8 = kc * 8;
    } else {
        const int krow = tid >> 4, rc = tid & 15;
        t.voff = rc * 8 < v.rows_left ? (unsigned)(((long)krow * v.ld + rc * 8) * 2) : 0xFFFFFFFFu;
        t.lds0 = lds_off_kmajor(krow, rc);                 // k-rows of chunk i: krow + 16*i, same swizzle bits
        t.kc8 = 0;
    }
    return t;
}

template <bool TR>
__device__ __forceinline__ void fetch_tile(const OperandView& v, const ThreadPlan& t, int k0, int K, u32x4 (&r)[4]) {
    const long step = (TR ? 16L : 32L) * v.ld;             // elements between chunk i and chunk i+1
    const long adv = TR ? (long)k0 * v.ld : (long)k0;
    unsigned off = t.voff;
    if constexpr (!TR) {
        if (k0 + t.kc8 >= K) off = 0xFFFFFFFFu;            // whole chunk beyond K
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // bytes addressable from this chunk row's base; reads past the operand's end return 0.
        // Clamped to [0, 0xFFFFFFF0] with 32-bit scalar compares only (a 64-bit compare would
        // go to the VALU and take VGPRs that in-flight loads own).
        const long rem = (v.remaining - adv - i * step) * 2;
        const int hi = (int)(rem >> 32);
        const unsigned lo = (unsigned)rem;
        const unsigned rec = hi < 0 ? 0u : (hi > 0 ? 0xFFFFFFF0u : (lo > 0xFFFFFFF0u ? 0xFFFFFFF0u : lo));
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(v.base + adv + i * step), 0, (int)rec, 0x00020000);
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
}

// Registers -> LDS.  Only here (when the data is needed anyway, never right after the
// load was issued) a row-major chunk that straddles a ragged K gets its elements at
// k >= K zeroed; the test is wave-uniform and false for every tile but a ragged last one.
template <bool TR>
__device__ __forceinline__ void stash_tile(char* tile, const ThreadPlan& t, u32x4 (&r)[4], int k0, int K) {
    if constexpr (!TR) {
        if ((K & 7) != 0 && k0 + BK > K) {
            const int kk = K - (k0 + t.kc8);
            if (kk > 0 && kk < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int dw = 0; dw < 4; ++dw) {
                        if (2 * dw >= kk) r[i][dw] = 0u;
                        else if (2 * dw + 1 >= kk) r[i][dw] &= 0xFFFFu;
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tile + t.lds0 + i * 4096) = r[i];
}

// ------------------------------------------------------------------ epilogues
// All take the wave's 64x64 f32 tile in LDS (row stride CS_LD) and the global
// coordinates (mw, nw) of its corner.  A lane owns 4 fixed columns (n .. n+3) and
// the rows it*4 + (lane>>4), it = 0..15.  Everything that depends only on the
// column (bias) or only on the row (label, lse) is fetched once, ahead of the row
// loop, and per-element side inputs (residual, gate, C for accumulate) are fetched
// four rows ahead of their use, so no global-load latency sits inside the loop.

__device__ __forceinline__ f32x4 load4_any(const void* base, long idx, int dtype, bool vec, int nvalid) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        if (dtype == CAPDEC_BF16) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>((const bf16_t*)base + idx);
            r = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        } else {
            r = *reinterpret_cast<const f32x4*>((const float*)base + idx);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) r[e] = load_as_f32(base, idx + e, dtype);
    }
    return r;
}
__device__ __forceinline__ void store4_any(void* base, long idx, int dtype, bool vec, int nvalid, const float (&v)[4]) {
    if (vec) {
        if (dtype == CAPDEC_BF16) {
            bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
            *reinterpret_cast<bf16x4*>((bf16_t*)base + idx) = o;
        } else {
            *reinterpret_cast<f32x4*>((float*)base + idx) = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) store_from_f32(base, idx + e, dtype, v[e]);
    }
}

// SIDE = false: no per-element side input (residual / gate / accumulate): the row loop
// contains no global load, so its stores are never waited for.
// SIDE = true: side inputs of 4 rows are fetched together, then the 4 rows are finished.
// keep flags of the 4 consecutive elements idx .. idx+3: one hash when idx is a multiple of 4 (every lane's
// column group is, when N is a multiple of 4), else one per element
__device__ __forceinline__ void keep_flags4(uint32_t k0, uint32_t k1, uint32_t idx, uint32_t thr, bool (&keep)[4]) {
    if ((idx & 3) == 0) dropout_keep4(k0, k1, idx, thr, keep);
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[e] = dropout_keep(k0, k1, idx + e, thr);
    }
}

template <bool SIDE>
__device__ __forceinline__ void epilogue_std(const GemmArgs& p, const float* cs, int mw, int nw, int lane) {
    const int c4 = (lane & 15) * 4, g = lane >> 4;
    const int n = nw + c4;
    const int nvalid = min(4, p.N - n);                    // <= 0: this lane owns no column
    const bool full = nvalid == 4;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    f32x4 scale4 = {1.f, 1.f, 1.f, 1.f};                   // the tile already holds acc + bias
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (n + e < p.col_scale_n) scale4[e] = p.col_scale;
    const bool vec_c = full && (p.ldc & 3) == 0;
    const bool vec_r = full && (p.ldr & 3) == 0;
    const bool vec_g = full && (p.ldg & 3) == 0;
    const bool vec_x = full && (p.ldaux & 3) == 0;
    const float relu_floor = p.act == CAPDEC_ACT_RELU ? 0.f : -INFINITY;

    auto finish_row = [&](int it, f32x4 a, f32x4 res, f32x4 gat, f32x4 cin) {
        const long m = (long)mw + it * 4 + g;
        if (m >= p.M || nvalid <= 0) return;
        f32x4 x = a * scale4;
        const f32x4 pre = x;
        if (p.act == CAPDEC_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = gelu_f(x[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], relu_floor);
        }
        if constexpr (SIDE) {
            if (p.gate_mode == CAPDEC_GATE_RELU_OUT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (gat[e] != 0.f) ? x[e] * p.gate_scale : 0.f;
            } else if (p.gate_mode == CAPDEC_GATE_GELU_PRE) {
                bool gkeep[4] = {true, true, true, true};
                if (p.gate_thr) keep_flags4(p.gate_k0, p.gate_k1, (uint32_t)(m * p.N + n), p.gate_thr, gkeep);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = gelu_grad_f(gat[e]);
                    if (p.gate_thr) f = gkeep[e] ? f * p.gate_scale : 0.f;
                    x[e] *= f;
                }
            }
        }
        if (p.drop_thr) {
            bool keep[4];
            keep_flags4(p.drop_k0, p.drop_k1, (uint32_t)(m * p.N + n), p.drop_thr, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = keep[e] ? x[e] * p.drop_scale : 0.f;
        }
        if constexpr (SIDE) x += res + cin;
        float v[4], pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = e < nvalid ? x[e] : 0.f; pv[e] = pre[e]; csum[e] += v[e]; }
        if (p.aux_out) store4_any(p.aux_out, m * p.ldaux + n, p.out_dtype, vec_x, nvalid, pv);
        if (p.kv_cache && n >= p.kv_q) {
            // K/V columns: cache[b][h][k|v][kv_s0 + s][j]; 4 consecutive columns stay inside one head (hd % 4 == 0)
            const int nn = n - p.kv_q;
            const int which = nn / p.kv_d, rem = nn - which * p.kv_d;
            const int hh = rem / p.kv_hd, jj = rem - hh * p.kv_hd;
            const long sq = m / p.kv_B, b = m - sq * p.kv_B;
            const long dst = ((((b * p.kv_H + hh) * 2 + which) * p.kv_smax) + p.kv_s0 + sq) * p.kv_hd + jj;
            store4_any(p.kv_cache, dst, p.out_dtype, full, nvalid, v);
        } else {
            store4_any(p.C, m * p.ldc + n, p.out_dtype, vec_c, nvalid, v);
        }
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!SIDE) {
#pragma unroll 2
        for (int it = 0; it < 16; ++it)
            finish_row(it, *reinterpret_cast<const f32x4*>(cs + (it * 4 + g) * CS_LD + c4), zero4, zero4, zero4);
    } else {
#pragma unroll 1
        for (int ch = 0; ch < 4; ++ch) {
            f32x4 res[4], gat[4], cin[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long m = (long)mw + (ch * 4 + j) * 4 + g;
                const bool live = m < p.M && nvalid > 0;
                res[j] = gat[j] = cin[j] = zero4;
                if (p.residual && live) res[j] = load4_any(p.residual, m * p.ldr + n, CAPDEC_F32, vec_r, nvalid);
                if (p.gate_mode != CAPDEC_GATE_NONE && live) gat[j] = load4_any(p.gate, m * p.ldg + n, p.gate_dtype, vec_g, nvalid);
                if (p.accumulate && live) cin[j] = load4_any(p.C, m * p.ldc + n, CAPDEC_F32, vec_c, nvalid);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int it = ch * 4 + j;
                finish_row(it, *reinterpret_cast<const f32x4*>(cs + (it * 4 + g) * CS_LD + c4), res[j], gat[j], cin[j]);
            }
        }
    }
    if (p.colsum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = csum[e];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16 && e < nvalid) atomicAdd(p.colsum + n + e, s);
        }
    }
}

__device__ __forceinline__ int row_label(const GemmArgs& p, long m) {
    const long l = m / p.ce_B, b = m - l * p.ce_B;
    return (int)p.labels[b * p.ce_L + l];
}

constexpr float LOG2E = 1.4426950408889634f;

// per (row, 64-column slot): max, argmax, sum exp(z - max) (and the second max when asked for); the
// label's logit.  One ROW per lane: the wave's 64x64 tile (bias already added, columns past N = -inf)
// is read back row-wise (row stride 68 floats puts 16 consecutive rows on 16 different 16-B bank slots,
// so the ds_read_b128 are conflict-free), and the whole reduction over the 64 columns stays inside the
// lane -- no cross-lane traffic, no global loads, no bounds checks.  The maximum is tracked per 4-column
// chunk (5 VALU ops per 4 columns) and the winning column is located inside the winning chunk afterwards.
// The four results of a row go out as one 16-byte store, rows contiguous across lanes.
template <int W, typename TO>
__device__ __forceinline__ void epilogue_ce_fwd(const GemmArgs& p, float* cs, int mw, int nw, int slot, int lane) {
    const long m = (long)mw + lane;
    const bool live = m < p.M;
    float* row = cs + lane * CS_LD;
    float m1 = -INFINITY;
    int ci = 0;
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c4 * 4);
        const float cm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        const bool gt = cm > m1;                             // strict: the lowest chunk wins a tie
        ci = gt ? c4 : ci;
        m1 = gt ? cm : m1;
    }
    const f32x4 w = *reinterpret_cast<const f32x4*>(row + ci * 4);
    const int a1 = nw + ci * 4 + (w[0] == m1 ? 0 : (w[1] == m1 ? 1 : (w[2] == m1 ? 2 : 3)));
    float m2 = -INFINITY;
    if (p.want_sec) {
#pragma unroll
        for (int c4 = 0; c4 < 16; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) m2 = fmaxf(m2, nw + c4 * 4 + e == a1 ? -INFINITY : v[e]);
        }
    }
    if (live && p.labels) {
        const int lab = row_label(p, m) - nw;                // column of the label inside this slot, if 0..63
        if (lab >= 0 && lab < 64) p.zlabel[m] = row[lab];
    }
    float sum = 0.f;
    const float nref = m1 > -INFINITY ? -m1 * LOG2E : 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c4 * 4);
        f32x4 ev;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ev[e] = __builtin_amdgcn_exp2f(fmaf(v[e], LOG2E, nref)); sum += ev[e]; }
        if (p.C) *reinterpret_cast<f32x4*>(row + c4 * 4) = ev;       // kept for the store pass below
    }
    if (live && slot < p.np) p.part4[(long)slot * p.M + m] = f32x4{m1, sum, m2, __int_as_float(a1)};
    if (!p.C) return;
    // exp(z - slot max), the numerators of the softmax, go out for the backward (which then needs no second
    // product over the vocabulary): W consecutive columns per lane, one 16-byte store per row
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    constexpr int LPR = 64 / W, RPI = 64 / LPR;
    const int c = (lane % LPR) * W, g = lane / LPR;
    const int n = nw + c;
    if (n >= p.store_cols) return;
    TO* C = (TO*)p.C;
#pragma unroll 2
    for (int it = 0; it < 64 / RPI; ++it) {
        const int r = it * RPI + g;
        const long mm = (long)mw + r;
        if (mm >= p.M) continue;
        const float* src = cs + r * CS_LD + c;
        if constexpr (sizeof(TO) == 2) {
            if constexpr (W == 8) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)a[e]; o[4 + e] = (bf16_t)b[e]; }
                // 0.8 GB that is next read after everything in the caches has turned over: stream it past them,
                // so that it does not evict the W / x panels the main loops of the other tiles are re-reading
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(C + mm * p.ldc + n));
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)a[e];
                *reinterpret_cast<bf16x4*>(C + mm * p.ldc + n) = o;
            }
        } else {
            *reinterpret_cast<f32x4*>(C + mm * p.ldc + n) = *reinterpret_cast<const f32x4*>(src);
        }
    }
}

// dl = (softmax - onehot) * gscale for one vocabulary slab (tile holds z = logits incl. bias; -inf past N).
// A lane owns W consecutive columns (W = 8 bf16 or 4 f32: one 16-byte store per row) of the rows
// it*RPI + lane/(64/W).  exp(z - lse) is one fma + one v_exp; the one-hot is applied by the single lane
// that owns the label's column.
template <int W, typename TO>
__device__ __forceinline__ void epilogue_ce_bwd(const GemmArgs& p, float* cs, int mw, int nw, int lane) {
    constexpr int LPR = 64 / W, RPI = 64 / LPR;              // lanes per row, rows per iteration
    const int c = (lane % LPR) * W, g = lane / LPR;
    const int n = nw + c;
    float csum[W];
#pragma unroll
    for (int e = 0; e < W; ++e) csum[e] = 0.f;
    {   // row constants, one row per lane, parked in the pad columns: label column (slab-relative), -lse*log2(e)
        const long m = (long)mw + lane;
        int lab = -1;
        float nl = 0.f;
        if (m < p.M) { lab = row_label(p, m) - p.ce_col0; nl = -p.lse[m] * LOG2E; }
        cs[lane * CS_LD + 64] = __int_as_float(lab);
        cs[lane * CS_LD + 65] = nl;
    }
    TO* C = (TO*)p.C;
#pragma unroll 2
    for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + g;
        const long m = (long)mw + row;
        if (m >= p.M || n >= p.store_cols) continue;
        const float* src = cs + row * CS_LD;
        const int lab = __float_as_int(src[64]) - n;
        const float nl = src[65];
        float v[W];
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = __builtin_amdgcn_exp2f(fmaf(a[e], LOG2E, nl));      // exp2(-inf) = 0 past N
        }
        if ((unsigned)lab < (unsigned)W) {
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] -= e == lab ? 1.f : 0.f;
        }
#pragma unroll
        for (int e = 0; e < W; ++e) { v[e] *= p.gscale; csum[e] += v[e]; }
        if constexpr (sizeof(TO) == 2) {
            if constexpr (W == 8) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(C + m * p.ldc + n));      // streamed: read back only by the dgrad / wgrad
            } else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                *reinterpret_cast<bf16x4*>(C + m * p.ldc + n) = o;
            }
        } else {
            static_assert(W == 4, "f32 rows go out 4 columns per lane");
            *reinterpret_cast<f32x4*>(C + m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    if (p.colpart) {
        // column sums of this wave's 64 rows -> one row of the partial buffer (plain 16-B stores;
        // atomics on the bias gradient would collide 128-fold per column)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = csum[e];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            csum[e] = t;
        }
        // a wave whose 64 rows all lie past M owns no row of the [ceil(M/64)][ld] partial buffer
        if (lane < LPR && n < p.store_cols && mw < p.M) {
            float* dst = p.colpart + (long)(mw >> 6) * p.ld_colpart + n;
#pragma unroll
            for (int q = 0; q < W / 4; ++q)
                *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{csum[4 * q], csum[4 * q + 1], csum[4 * q + 2], csum[4 * q + 3]};
        }
    }
}

// The common forward epilogue, straight-line: (acc + bias) * column scale, [pre-activation copy], ReLU / GELU, dropout,
// one 16-byte store per row and lane (W = 8 bf16 or 4 f32 columns).  The host selects it when nothing else is asked
// for and N, ldc and the base of C keep every lane's W columns aligned and either all inside N or all outside.
template <int W, typename TO>
__device__ __forceinline__ void epilogue_std_fast(const GemmArgs& p, const float* cs, int mw, int nw, int lane) {
    constexpr int LPR = 64 / W, RPI = 64 / LPR;
    const int c = (lane % LPR) * W, g = lane / LPR;
    const int n = nw + c;
    if (n >= p.N) return;
    float sc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) sc[e] = n + e < p.col_scale_n ? p.col_scale : 1.f;
    const float relu_floor = p.act == CAPDEC_ACT_RELU ? 0.f : -INFINITY;
    TO* C = (TO*)p.C;
#pragma unroll 2
    for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + g;
        const long m = (long)mw + row;
        if (m >= p.M) continue;
        float v[W];
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(cs + row * CS_LD + c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = a[e] * sc[4 * q + e];
        }
        if (p.aux_out) {                     // the pre-activation values, kept for the backward of the activation
            TO* X = (TO*)p.aux_out;
            if constexpr (sizeof(TO) == 2) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                *reinterpret_cast<bf16x8*>(X + m * p.ldaux + n) = o;
            } else {
                *reinterpret_cast<f32x4*>(X + m * p.ldaux + n) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        if (p.act == CAPDEC_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = gelu_f(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = fmaxf(v[e], relu_floor);
        }
        if (p.drop_thr) {
            const uint32_t base = (uint32_t)(m * p.N + n);          // a multiple of W: N and n are
#pragma unroll
            for (int q = 0; q < W / 4; ++q) {
                bool keep[4];
                dropout_keep4(p.drop_k0, p.drop_k1, base + 4 * q, p.drop_thr, keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = keep[e] ? v[4 * q + e] * p.drop_scale : 0.f;
            }
        }
        if constexpr (sizeof(TO) == 2) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x8*>(C + m * p.ldc + n) = o;
        } else {
            *reinterpret_cast<f32x4*>(C + m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// The backward product through activation + dropout (dz = (dy W) * act'(z) * keep / (1-p)) when nothing else is
// asked for: 8 bf16 columns per lane, the gate tensor read and the result written 16 bytes at a time, all eight
// gate loads of the lane in flight before the first use.  The host selects it (fast_epi == 16) when out and gate
// are bf16 and N, ldc, ldg and both bases keep every lane's 8 columns aligned and inside N.
__device__ __forceinline__ void epilogue_gate_fast(const GemmArgs& p, const float* cs, int mw, int nw, int lane) {
    constexpr int W = 8, LPR = 64 / W, RPI = 64 / LPR;
    const int c = (lane % LPR) * W, g = lane / LPR;
    const int n = nw + c;
    if (n >= p.N) return;
    const bf16_t* gate = (const bf16_t*)p.gate;
    bf16_t* C = (bf16_t*)p.C;
    float csum[W];
#pragma unroll
    for (int e = 0; e < W; ++e) csum[e] = 0.f;
    bf16x8 gv[64 / RPI];
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
        const long m = (long)mw + it * RPI + g;
        gv[it] = m < p.M ? *reinterpret_cast<const bf16x8*>(gate + m * p.ldg + n) : bf16x8{};
    }
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + g;
        const long m = (long)mw + row;
        if (m >= p.M) continue;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cs + row * CS_LD + c), a1 = *reinterpret_cast<const f32x4*>(cs + row * CS_LD + c + 4);
        float v[W] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (p.gate_mode == CAPDEC_GATE_RELU_OUT) {
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = (float)gv[it][e] != 0.f ? v[e] * p.gate_scale : 0.f;
        } else {
            bool keep[W] = {true, true, true, true, true, true, true, true};
            if (p.gate_thr) {
                const uint32_t base = (uint32_t)(m * p.N + n);          // a multiple of 8
                bool k0[4], k1[4];
                dropout_keep4(p.gate_k0, p.gate_k1, base, p.gate_thr, k0);
                dropout_keep4(p.gate_k0, p.gate_k1, base + 4, p.gate_thr, k1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { keep[e] = k0[e]; keep[4 + e] = k1[e]; }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = keep[e] ? v[e] * (gelu_grad_f((float)gv[it][e]) * p.gate_scale) : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < W; ++e) { o[e] = (bf16_t)v[e]; csum[e] += v[e]; }
        *reinterpret_cast<bf16x8*>(C + m * p.ldc + n) = o;
    }
    if (p.colpart) {
        // column sums of this wave's 64 rows -> one row of the partial buffer, folded later in a fixed order
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = csum[e];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            csum[e] = t;
        }
        if (lane < LPR && mw < p.M) {
            float* dst = p.colpart + (long)(mw >> 6) * p.ld_colpart + n;
            *reinterpret_cast<f32x4*>(dst) = f32x4{csum[0], csum[1], csum[2], csum[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
        }
    }
}

// The MFMAs are issued with the operands swapped (B fragment first), so that a lane holds
//   C[row = lane & 15][cols 4*(lane>>4) .. +3]   of each 16x16 tile:
// four CONSECUTIVE columns of one row, which go to the row-major LDS tile as one 16-byte write.
//
// Column constants of a wave's tile in that layout (a lane holds 16 distinct columns): the bias, and for
// the cross-entropy epilogues -inf past N (such columns never win a max and add exp(-inf) = 0), which
// leaves the row loops free of bias loads and bounds checks.
template <int EPI>
__device__ __forceinline__ void load_col_consts(const GemmArgs& p, int n0, int wc, int lane, f32x4 (&cb)[4]) {
    const float past = (EPI == EPI_CE_FWD || EPI == EPI_CE_BWD) ? -INFINITY : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
        if (p.bias && n + 3 < p.N && (reinterpret_cast<uintptr_t>(p.bias + n) & 15) == 0) {
            cb[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) cb[j][r] = n + r < p.N ? (p.bias ? p.bias[n + r] : 0.f) : past;
        }
    }
}

// Accumulators (+ column constants) -> the wave's LDS tile: 16 writes of 16 bytes per 64 rows.
__device__ __forceinline__ void stage_acc(char* smem, const f32x4 (*acc)[4], const f32x4 (&cb)[4], int wave, int lane) {
    float* cs = reinterpret_cast<float*>(smem + wave * CS_WAVE_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(cs + (i * 16 + (lane & 15)) * CS_LD + j * 16 + 4 * (lane >> 4)) = acc[i][j] + cb[j];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): own wave's LDS writes landed
    __builtin_amdgcn_wave_barrier();
}

// the wave's staged 64x64 tile has its corner at (mw, nw); slot = index of its 64-column group
template <int EPI>
__device__ __forceinline__ void run_epilogue_at(const GemmArgs& p, float* cs, int mw, int nw, int slot, int lane) {
    if constexpr (EPI == EPI_STD) {
        if (p.fast_epi == 8) epilogue_std_fast<8, bf16_t>(p, cs, mw, nw, lane);
        else if (p.fast_epi == 4) epilogue_std_fast<4, float>(p, cs, mw, nw, lane);
        else epilogue_std<false>(p, cs, mw, nw, lane);
    } else if constexpr (EPI == EPI_STD_SIDE) {
        if (p.fast_epi == 16) epilogue_gate_fast(p, cs, mw, nw, lane);
        else epilogue_std<true>(p, cs, mw, nw, lane);
    }
    else if constexpr (EPI == EPI_CE_FWD) {
        if (p.fast_epi == 8) epilogue_ce_fwd<8, bf16_t>(p, cs, mw, nw, slot, lane);
        else if (p.out_dtype == CAPDEC_BF16) epilogue_ce_fwd<4, bf16_t>(p, cs, mw, nw, slot, lane);
        else epilogue_ce_fwd<4, float>(p, cs, mw, nw, slot, lane);
    }
    else {
        if (p.fast_epi == 8) epilogue_ce_bwd<8, bf16_t>(p, cs, mw, nw, lane);
        else if (p.out_dtype == CAPDEC_BF16) epilogue_ce_bwd<4, bf16_t>(p, cs, mw, nw, lane);
        else epilogue_ce_bwd<4, float>(p, cs, mw, nw, lane);
    }
}
template <int EPI>
__device__ __forceinline__ void run_epilogue(const GemmArgs& p, char* smem, int m0, int n0, int tile_n, int wave, int lane) {
    const int wr = wave >> 1, wc = wave & 1;
    run_epilogue_at<EPI>(p, reinterpret_cast<float*>(smem + wave * CS_WAVE_BYTES), m0 + wr * 64, n0 + wc * 64, tile_n * 2 + wc, lane);
}

// ------------------------------------------------------------------ bf16 MFMA kernel
template <bool A_TR, bool B_TR, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    map_tile(tm, tn, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    OperandView va, vb;
    if (A_TR) { va.base = (const bf16_t*)p.A + m0; va.remaining = (long)p.K * p.lda - m0; }
    else      { va.base = (const bf16_t*)p.A + (long)m0 * p.lda; va.remaining = (long)(p.M - m0) * p.lda; }
    va.ld = p.lda; va.rows_left = p.M - m0;
    if (B_TR) { vb.base = (const bf16_t*)p.B + n0; vb.remaining = (long)p.K * p.ldb - n0; }
    else      { vb.base = (const bf16_t*)p.B + (long)n0 * p.ldb; vb.remaining = (long)(p.N - n0) * p.ldb; }
    vb.ld = p.ldb; vb.rows_left = p.N - n0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: this block covers k-tiles [kt0, kt0 + nk) and writes its own f32 slab
    const int nk_total = (p.K + BK - 1) / BK;
    int kt0 = 0, nk = nk_total;
    if (p.split_k > 1) {
        const int kper = (nk_total + p.split_k - 1) / p.split_k;
        kt0 = blockIdx.y * kper;
        nk = min(kper, nk_total - kt0);
        p.C = (float*)p.C + (long)blockIdx.y * p.split_stride;
    }
    const int kb = kt0 * BK;
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
    // One k-tile of register prefetch: the loads of tile t+1 are issued before tile t is
    // multiplied out of LDS and written to the other LDS stage after it.  (A second register
    // set was tried: hipcc's wait-count insertion then drains every load at each LDS write,
    // so it only cost registers.  Deeper prefetch lives in the LDS-DMA kernel below.)
    u32x4 ra[4], rb[4];
    const ThreadPlan ta = make_plan<A_TR>(va, tid), tb = make_plan<B_TR>(vb, tid);
    fetch_tile<A_TR>(va, ta, kb, p.K, ra);
    fetch_tile<B_TR>(vb, tb, kb, p.K, rb);
    stash_tile<A_TR>(smem, ta, ra, kb, p.K);
    stash_tile<B_TR>(smem + TILE_BYTES, tb, rb, kb, p.K);
    __syncthreads();
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();

    for (int kt = 0; kt < nk; ++kt) {
        const char* sA = smem + (kt & 1) * STAGE_BYTES;
        const char* sB = sA + TILE_BYTES;
        const bool more = kt + 1 < nk;
        const int knext = kb + (kt + 1) * BK;
        if (more) {
            fetch_tile<A_TR>(va, ta, knext, p.K, ra);
            fetch_tile<B_TR>(vb, tb, knext, p.K, rb);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<A_TR>(sA, wr * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = read_frag<B_TR>(sB, wc * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);     // swapped: see stage_acc
        }
        if (more) {
            char* dA = smem + ((kt + 1) & 1) * STAGE_BYTES;
            stash_tile<A_TR>(dA, ta, ra, knext, p.K);
            stash_tile<B_TR>(dA + TILE_BYTES, tb, rb, knext, p.K);
        }
        __syncthreads();
    }

    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
    f32x4 cb[4];
    load_col_consts<EPI>(p, n0, wave & 1, lane, cb);
    stage_acc(smem, acc, cb, wave, lane);
    run_epilogue<EPI>(p, smem, m0, n0, tile_n, wave, lane);
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------ bf16 MFMA kernel, LDS-DMA ring
// Same tile and epilogues, but operand tiles go global -> LDS directly
// (buffer_load ... lds, 16 B per lane, no staging registers) into a ring of NST
// 32-deep stages.  NST-1 stages are in flight while one is multiplied; the wait is a
// COUNTED vmcnt (the NST-2 youngest stages stay in flight across the barrier), so a
// tile's HBM/L2 latency is covered by NST-1 MFMA phases.  An LDS-DMA writes
// wave-base + 16*lane, i.e. the LDS image is lane-linear per wave instruction;
// the bank-conflict swizzle is therefore applied to the per-lane SOURCE address and
// undone by the same XOR on the fragment read.  Needs K % 32 == 0 unless both operands are
// reduction-major (other K: the register-staged kernel above); M/N tails are out-of-range
// rows whose results are never stored.
constexpr int DBK = 32, DTILE_BYTES = 8192;      // k-depth of a ring stage; bytes of a 128-row (or 128-column) 32-deep operand image

// row-major [128 rows][32 k] bf16, 64-B rows; chunk (0..3) XOR 3*((row>>3)&1): conflict-free ds_read_b128
__device__ __forceinline__ int dlds_rowmajor(int row, int kc) { return row * 64 + ((kc ^ (((row >> 3) & 1) * 3)) << 4); }

template <bool TR>
__device__ __forceinline__ bf16x8 dread_frag(const char* tile, int rb, int lane) {
    if constexpr (!TR) return *reinterpret_cast<const bf16x8*>(tile + dlds_rowmajor(rb + (lane & 15), lane >> 4));
    else return read_frag<true>(tile + (rb >> 7) * DTILE_BYTES, rb & 127, 0, lane);
}

// byte offset (from the k-tile's base) a lane fetches for 1-KiB piece `piece` of a tile image
template <bool TR>
__device__ __forceinline__ unsigned dma_src_offset(const OperandView& v, int piece, int lane) {
    if constexpr (!TR) {
        const int row = 16 * piece + (lane >> 2), pc = lane & 3;
        const int kc = pc ^ (((row >> 3) & 1) * 3);
        return (unsigned)(((long)row * v.ld + kc * 8) * 2);
    } else {
        // a reduction-major tile wider than 128 columns is a row of 8-KiB sub-images of 128 columns each
        const int sub = piece >> 3, pp = piece & 7;
        const int krow = 4 * pp + (lane >> 4), pc = lane & 15;
        const int col = sub * 128 + (pc ^ (kswz(krow) << 1)) * 8;
        if (v.panel) return col < v.rows_left ? (unsigned)((((long)(col >> 5) * v.panel + krow) * 32 + (col & 31)) * 2) : 0xFFFFFFFFu;
        return col < v.rows_left ? (unsigned)(((long)krow * v.ld + col) * 2) : 0xFFFFFFFFu;
    }
}

// elements between the images of k and k + 1 (k a multiple of 32 for a row-major panel image, whose panels are `panel` rows of 32)
template <bool TR>
__device__ __forceinline__ long k_unit(const OperandView& v) { return TR ? (v.panel ? 32L : v.ld) : (v.panel ? v.panel : 1L); }

// buffer descriptor of an operand's k-tile starting at k0 (reads past the operand's end return 0)
template <bool TR>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const OperandView& v, int k0) {
    const long adv = (long)k0 * k_unit<TR>(v);
    const long rem = (v.remaining - adv) * 2;
    const int hi = (int)(rem >> 32);
    const unsigned lo = (unsigned)rem;
    const unsigned rec = hi < 0 ? 0u : (hi > 0 ? 0xFFFFFFF0u : (lo > 0xFFFFFFF0u ? 0xFFFFFFF0u : lo));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(v.base + adv), 0, (int)rec, 0x00020000);
}
// one 1-KiB piece: lane l's 16 bytes land at dst + 16*l
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned off, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, off, 0, 0, 0);
}
// NJ pieces per wave; piece j of wave w is image piece NW*j + w
template <bool TR, int NJ, int NW>
__device__ __forceinline__ void dma_tile(const OperandView& v, const unsigned* off, int k0, char* tile, int wave) {
    __amdgpu_buffer_rsrc_t rs = dma_rsrc<TR>(v, k0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma_piece(rs, off[j], tile + (NW * j + wave) * 1024);
}

// Geometry <WM, WN, TI, NST>: WM x WN waves, each wave TI x 4 MFMA tiles (16*TI rows x 64 columns),
// NST ring stages.
//   <2,2,4,4>  128x128, 4 waves, 2 workgroups/CU: small grids
//   <2,2,8,3>  256x128, 4 waves of 128x64, 2 workgroups/CU: per MFMA a quarter fewer DMA pieces and LDS
//              reads than 128x128 (whose 16 KiB per 32-deep step needs the CU's whole 64 B/clk fill path),
//              and the co-resident workgroup's main loop covers this one's epilogue (K = 768 products)
//   <2,4,8,4>  256x256, 8 waves of 128x64, 1 workgroup/CU, 4 stages: half the operand bytes per MFMA of
//              256x128 and a deeper ring, for the deep-K products (head dgrad / wgrad) whose waves
//              otherwise spend half their time parked on the stage wait
//   <4,2,4,4>  256x128, 8 waves of 64x64, 1 workgroup/CU (diagnostic)
// A reduction-major image wider than 128 columns is stored as 128-column sub-images.
template <int WM, int WN, int TI, int NST> struct DmaGeo {
    static constexpr int NW = WM * WN;                      // waves
    static constexpr int WROWS = 16 * TI;                   // rows per wave
    static constexpr int BMV = WM * WROWS;                  // tile rows
    static constexpr int BNV = WN * 64;                     // tile columns
    static constexpr int A_BYTES = BMV * 64;                // one 32-deep A image
    static constexpr int B_BYTES = BNV * 64;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NJA = A_BYTES / 1024 / NW;         // A pieces per wave
    static constexpr int NJB = B_BYTES / 1024 / NW;         // B pieces per wave
    static constexpr int INFLIGHT = (NST - 2) * (NJA + NJB);// DMA instructions that may stay outstanding at the wait
    static constexpr int RING = NST * STAGE;
    static constexpr int CSTAGE = NW * CS_WAVE_BYTES;
    static constexpr int LDS = RING > CSTAGE ? RING : CSTAGE;
    static constexpr int WG_PER_CU = 2 * LDS <= 160 * 1024 && NW == 4 ? 2 : 1;
};

template <bool A_TR, bool B_TR, int EPI, int WM, int WN, int TI, int NST>
__global__ __launch_bounds__(64 * WM * WN, (DmaGeo<WM, WN, TI, NST>::WG_PER_CU)) void gemm_bf16_dma_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = DmaGeo<WM, WN, TI, NST>;
    const int tm = (p.M + G::BMV - 1) / G::BMV, tn = (p.N + G::BNV - 1) / G::BNV;
    int tile_m, tile_n;
    map_tile(tm, tn, tile_m, tile_n);
    const int m0 = tile_m * G::BMV, n0 = tile_n * G::BNV;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;

    OperandView va, vb;
    if (A_TR) { va.base = (const bf16_t*)p.A + m0; va.remaining = (long)p.K * p.lda - m0; }
    else      { va.base = (const bf16_t*)p.A + (long)m0 * p.lda; va.remaining = (long)(p.M - m0) * p.lda; }
    va.ld = p.lda; va.rows_left = p.M - m0;
    if (B_TR) { vb.base = (const bf16_t*)p.B + n0; vb.remaining = (long)p.K * p.ldb - n0; }
    else      { vb.base = (const bf16_t*)p.B + (long)n0 * p.ldb; vb.remaining = (long)(p.N - n0) * p.ldb; }
    vb.ld = p.ldb; vb.rows_left = p.N - n0;
    unsigned offa[G::NJA], offb[G::NJB];
#pragma unroll
    for (int j = 0; j < G::NJA; ++j) offa[j] = dma_src_offset<A_TR>(va, G::NW * j + wave, lane);
#pragma unroll
    for (int j = 0; j < G::NJB; ++j) offb[j] = dma_src_offset<B_TR>(vb, G::NW * j + wave, lane);

    f32x4 cb[4];                             // column constants of the epilogue: fetched now, used after the main loop
    load_col_consts<EPI>(p, n0, wc, lane, cb);
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: this block covers k-tiles [kt0, kt0 + nk) and writes its own f32 slab
    const int nk_total = (p.K + DBK - 1) / DBK;
    int kt0 = 0, nk = nk_total;
    if (p.split_k > 1) {
        const int kper = (nk_total + p.split_k - 1) / p.split_k;
        kt0 = blockIdx.y * kper;
        nk = max(0, min(kper, nk_total - kt0));
        p.C = (float*)p.C + (long)blockIdx.y * p.split_stride;
    }
    if (p.stamps && tid == 0) {
        p.stamps[(long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
        // where the workgroup runs: XCC_ID (register 20) in the high word, HW_ID (register 4: CU, SH, SE) in the low word
        p.stamps[(long)blockIdx.x * 8 + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                             __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    // a stage index past the last k-tile still issues its (harmless) loads, so that the
    // number of DMA instructions in flight -- what vmcnt counts -- is the same in every iteration
    auto dma_stage = [&](int kt, int slot) {
        char* st = smem + slot * G::STAGE;
        dma_tile<A_TR, G::NJA, G::NW>(va, offa, (kt0 + kt) * DBK, st, wave);
        dma_tile<B_TR, G::NJB, G::NW>(vb, offb, (kt0 + kt) * DBK, st + G::A_BYTES, wave);
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) dma_stage(s, s);
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();

    int rslot = 0, wslot = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // NST-1 stages of this wave are outstanding; all but the NST-2 youngest = stage kt
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();        // every wave's share of stage kt has landed; stage kt-1 is no longer read
        const char* sA = smem + rslot * G::STAGE;
        const char* sB = sA + G::A_BYTES;
        // all fragment reads of the step go out first, then the refill DMAs (whose issue covers the LDS
        // latency), then the MFMAs, each waiting only for the fragments it consumes
        bf16x8 af[TI], bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = dread_frag<B_TR>(sB, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = dread_frag<A_TR>(sA, wr * G::WROWS + i * 16, lane);
        __builtin_amdgcn_sched_barrier(0);
        // refill of the stage that was read in iteration kt-1: one DMA piece after each row of MFMAs, so that
        // its issue time is covered by matrix work instead of delaying the first MFMA of the step
        char* wst = smem + wslot * G::STAGE;
        const __amdgpu_buffer_rsrc_t rsa = dma_rsrc<A_TR>(va, (kt0 + kt + NST - 1) * DBK);
        const __amdgpu_buffer_rsrc_t rsb = dma_rsrc<B_TR>(vb, (kt0 + kt + NST - 1) * DBK);
        rslot = rslot + 1 == NST ? 0 : rslot + 1;
        wslot = wslot + 1 == NST ? 0 : wslot + 1;
        constexpr int NP = G::NJA + G::NJB;
        static_assert(NP <= TI, "one DMA piece per MFMA row at most");
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);     // swapped: see stage_acc
            if (i < NP) {
                __builtin_amdgcn_sched_barrier(0);
                if (i < G::NJA) dma_piece(rsa, offa[i], wst + (G::NW * i + wave) * 1024);
                else dma_piece(rsb, offb[i - G::NJA], wst + G::A_BYTES + (G::NW * (i - G::NJA) + wave) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing refills must land before LDS is reused
    __syncthreads();

    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
    // the wave's rows go through its private 64x64 LDS tile, 64 rows at a time
#pragma unroll
    for (int h = 0; h < TI / 4; ++h) {
        stage_acc(smem, acc + 4 * h, cb, wave, lane);
        if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 3 + 2 * h] = __builtin_amdgcn_s_memtime();
        run_epilogue_at<EPI>(p, reinterpret_cast<float*>(smem + wave * CS_WAVE_BYTES), m0 + wr * G::WROWS + h * 64, n0 + wc * 64,
                             (n0 >> 6) + wc, lane);
        if (p.stamps && tid == 0 && h == 0) p.stamps[(long)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime();
    }
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------ bf16 MFMA kernel, 4 waves of 128x128
// The ring kernels above read 12 KiB of fragments per wave and 32-deep step for 32 MFMAs; with the stage refills
// (which also go through the LDS ports) two co-resident 256x128 workgroups or one 8-wave 256x256 workgroup move
// ~144 KiB through the LDS per 1024 MFMA cycles of the CU -- more than its 128 B/clk carry.  Here a wave owns a
// 128x128 block of a 256x256 tile: 16 KiB of fragments for 64 MFMAs, 96 KiB per 1024 MFMA cycles with the refills.
// The 256 accumulator registers leave room for ONE wave per SIMD, so nothing else covers the LDS latency: the
// fragments are double-buffered in registers -- those of step k+1 are read (and the ring refilled) between the
// MFMA rows of step k -- and the ring holds NST = 4 stages, all in flight or being read.  Plain products only
// (EPI_STD, incl. split-K slabs); K as for the ring kernels.
// Transposing fragment read that the compiler cannot see into.  It makes every LDS read it cannot prove independent
// of the LDS-DMA writes wait for ALL of them (s_waitcnt vmcnt(0)), and the transposing-read builtin carries nothing to
// prove it with -- which empties the ring at every step.  As inline assembly the two reads are just instructions:
// no wait is inserted, and none for their results either, so the caller waits (lgkmcnt) before it uses them.
// `addr` = LDS byte address of the lane's first 8 bytes (image base + tr_lane_offset).
__device__ __forceinline__ int tr_lane_offset(int rb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = rb + 4 * pp;
    return lds_off_kmajor(8 * g + q, col >> 3) + ((col & 7) << 1);
}
__device__ __forceinline__ bf16x8 read_frag_tr_opaque(unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(addr));
    return __builtin_bit_cast(bf16x8, (u32x4{lo[0], lo[1], hi[0], hi[1]}));
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(const char, p); }

// acc += a x b with the accumulator tied in place in the accumulation registers.  Left to the register allocator, a
// kernel whose accumulators fill all 256 of them gets three-address MFMAs fed through copies.  Being opaque to the
// compiler, it gets no hazard padding: the caller keeps dependent uses (the same accumulator, reads of the result) far away.
__device__ __forceinline__ void mfma_inplace(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(__builtin_bit_cast(u32x4, a)), "v"(__builtin_bit_cast(u32x4, b)));
}

// Geometry of the pipelined kernel: 4 waves (2 x 2), each 128 rows x 16*TJ columns.
//   TJ = 8: 256x256 tile, 4 ring stages of 32 KiB, one workgroup per CU        (main-loop-dominated plain products)
//   TJ = 4: 256x128 tile, 3 ring stages of 24 KiB, two workgroups per CU, whose epilogues and main loops overlap
//           (K = 768 products with heavy epilogues); each wave keeps the MFMA pipe fed on its own, so a workgroup
//           whose neighbour is in its epilogue does not drop to half rate
template <int TJ> struct PipeGeo {
    static constexpr int BN = 32 * TJ;
    static constexpr int A_BYTES = 16384, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    static constexpr int NST = TJ == 8 ? 4 : 3;
    static constexpr int PA = 4, PB = B_BYTES / 1024 / 4, NP = PA + PB;     // DMA pieces per wave and stage
    static constexpr int RING = NST * STAGE;
    static constexpr int LDS = RING > 4 * CS_WAVE_BYTES ? RING : 4 * CS_WAVE_BYTES;
    static constexpr int WG_PER_CU = TJ == 8 ? 1 : 2;
};
// s_waitcnt immediate (gfx9 layout: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]); expcnt is left alone
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (lgkm << 8); }

// KVEC: the workgroups of the first tile column also form A . kvec (one more column of the product, a [K] vector that is
// kept in LDS behind the ring): per k-step one extra MFMA per 16 rows, whose B-side fragment holds kvec's 32 values in
// the lanes of column 0 and zeros elsewhere (those lanes read a zero slot), accumulated in arch VGPRs.  The two waves
// that share a row block (and its A fragments) take alternate 16-row blocks of it.
constexpr int KVEC_BYTES = 16384 + 64;      // up to 8192 k per workgroup, + the zero slot
__device__ __forceinline__ u32x4 lds_read_b128_opaque(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void mfma_inplace_v(f32x4& c, const u32x4& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(__builtin_bit_cast(u32x4, b)));
}

// KVEC = 2: the vector is all ones (row sums of A, i.e. column sums of a reduction-major operand): a constant fragment,
// nothing in LDS, any k range.
template <bool A_TR, bool B_TR, int EPI, int TJ, int KVEC = 0>
__global__ __launch_bounds__(256, (PipeGeo<TJ>::WG_PER_CU)) void gemm_bf16_pipe_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = PipeGeo<TJ>;
    constexpr int NST = G::NST, STAGE = G::STAGE, PA = G::PA, PB = G::PB, NP = G::NP, WCOLS = 16 * TJ;
    const int tm = (p.M + 255) / 256, tn = (p.N + G::BN - 1) / G::BN;
    int tile_m, tile_n;
    map_tile(tm, tn, tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * G::BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    OperandView va, vb;
    if (A_TR) { va.base = (const bf16_t*)p.A + m0; va.remaining = (long)p.K * p.lda - m0; }
    else      { va.base = (const bf16_t*)p.A + (long)m0 * p.lda; va.remaining = (long)(p.M - m0) * p.lda; }
    va.ld = p.lda; va.rows_left = p.M - m0;
    if (B_TR) { vb.base = (const bf16_t*)p.B + n0; vb.remaining = (long)p.K * p.ldb - n0; }
    else      { vb.base = (const bf16_t*)p.B + (long)n0 * p.ldb; vb.remaining = (long)(p.N - n0) * p.ldb; }
    vb.ld = p.ldb; vb.rows_left = p.N - n0;
    if (p.a_panel) {
        // A's stored matrix ([M][K], or [K][M] when reduction-major) as a column-panel image: [cols / 32][a_panel rows][32]
        va.panel = p.a_panel;
        const long total = (long)(((A_TR ? p.M : p.K) + 31) / 32) * p.a_panel * 32;
        const long first = A_TR ? (long)(m0 >> 5) * p.a_panel * 32 : (long)m0 * 32;
        va.base = (const bf16_t*)p.A + first; va.remaining = total - first;
        if (!A_TR) va.ld = 32;
    }
    unsigned offa[PA], offb[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) offa[j] = dma_src_offset<A_TR>(va, 4 * j + wave, lane);
#pragma unroll
    for (int j = 0; j < PB; ++j) offb[j] = dma_src_offset<B_TR>(vb, 4 * j + wave, lane);
    // column constants of the epilogue: with two workgroups per CU fetched now and used after the main loop; the
    // 256-wide tile has no registers to park them in and fetches them after it
    f32x4 cb[4];
    if constexpr (TJ == 4) load_col_consts<EPI>(p, n0 + wc * WCOLS, 0, lane, cb);
    f32x4 acc[8][TJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_total = (p.K + DBK - 1) / DBK;
    int kt0 = 0, nk = nk_total;
    if (p.split_k > 1) {
        const int kper = (nk_total + p.split_k - 1) / p.split_k;
        kt0 = blockIdx.y * kper;
        nk = max(0, min(kper, nk_total - kt0));
        p.C = (float*)p.C + (long)blockIdx.y * p.split_stride;
    }
    // KVEC: this workgroup's k range of the vector -> LDS (zeros past K), before the ring starts: ordinary loads, complete
    // before the first DMA is issued, so the counted waits below see DMAs only
    const bool do_kvec = KVEC != 0 && tile_n == 0;
    unsigned gaddr = 0, gstride = 0;
    u32x4 gf[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    f32x4 bacc[KVEC ? 4 : 1];
    if constexpr (KVEC != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (KVEC == 2) {
        const unsigned one2 = (lane & 15) == 0 ? 0x3F803F80u : 0u;      // two bf16 ones in the lanes of column 0
        gf[0] = gf[1] = u32x4{one2, one2, one2, one2};
    }
    if constexpr (KVEC == 1) {
        char* kv = smem + G::LDS;
        if (do_kvec) {
            const long k0 = (long)kt0 * DBK;
            for (int c = tid * 8; c < nk * DBK; c += 256 * 8) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = k0 + c + e < p.K ? p.kvec[k0 + c + e] : (bf16_t)0.f;
                *reinterpret_cast<bf16x8*>(kv + c * 2) = v;
            }
            if (tid < 4) *reinterpret_cast<u32x4*>(kv + 16384 + tid * 16) = u32x4{0, 0, 0, 0};
        }
        __syncthreads();
        // lanes of column 0 walk the vector, 32 values (64 bytes) per k-step, 8 per lane group; the others stay on the zero slot
        const bool col0 = (lane & 15) == 0;
        gaddr = lds_addr(kv) + (col0 ? (unsigned)((lane >> 4) * 16) : 16384u);
        gstride = col0 ? 64u : 0u;
        if (do_kvec) { gf[0] = lds_read_b128_opaque(gaddr); gaddr += gstride; }
    }
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
    // all NST stages go out at once; a stage index past the last k-tile still issues its (harmless) loads, so that the
    // number of DMA instructions in flight -- what vmcnt counts -- is the same in every iteration
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        dma_tile<A_TR, PA, 4>(va, offa, (kt0 + s) * DBK, smem + s * STAGE, wave);
        dma_tile<B_TR, PB, 4>(vb, offb, (kt0 + s) * DBK, smem + s * STAGE + G::A_BYTES, wave);
    }
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();

    // B fragments are double-buffered (every MFMA row of a step uses all of them); an A fragment is dead once its row of
    // MFMAs has been issued, so the next step's takes its place while the following row runs (the last one goes
    // through a spare register), which puts every LDS read at least one MFMA row before the end of the step.
    bf16x8 fa[8], fb[2][TJ], fa7;
    int troffA[A_TR ? 8 : 1], troffB[B_TR ? TJ : 1];     // lane offsets of the 16-row fragments inside a reduction-major sub-image
    if constexpr (A_TR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) troffA[i] = tr_lane_offset(i * 16, lane);
    }
    if constexpr (B_TR) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) troffB[j] = tr_lane_offset(((wc * WCOLS) & 127) + j * 16, lane);
    }
    auto read_a = [&](const char* sA, int i) {
        if constexpr (A_TR) return read_frag_tr_opaque(lds_addr(sA + wr * DTILE_BYTES) + troffA[i]);
        else return dread_frag<false>(sA, wr * 128 + i * 16, lane);
    };
    auto read_b = [&](const char* sB, int j) {
        if constexpr (B_TR) return read_frag_tr_opaque(lds_addr(sB + ((wc * WCOLS) >> 7) * DTILE_BYTES) + troffB[j]);
        else return dread_frag<false>(sB, wc * WCOLS + j * 16, lane);
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NP) : "memory");
    __builtin_amdgcn_s_barrier();            // stage 0 has landed
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = read_a(smem, i);
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[0][j] = read_b(smem + G::A_BYTES, j);
    int slot = 0;                            // ring slot of stage kt
    // the refill of step kt reads k-tile kt0 + kt + NST: its start and the bytes addressable from there advance by a
    // constant per step, which keeps the descriptor arithmetic inside the main loop to a few scalar instructions
    const long stepA = DBK * k_unit<A_TR>(va) * 2, stepB = DBK * k_unit<B_TR>(vb) * 2;      // bytes
    const char* refA = (const char*)va.base + (long)(kt0 + NST) * stepA;
    const char* refB = (const char*)vb.base + (long)(kt0 + NST) * stepB;
    long remA = va.remaining * 2 - (long)(kt0 + NST) * stepA, remB = vb.remaining * 2 - (long)(kt0 + NST) * stepB;
    auto rsrc_at = [](const char* base, long rem) {
        const unsigned rec = rem <= 0 ? 0u : (rem > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)rem);
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)rec, 0x00020000);
    };
    // One step: MFMAs on the fragments of stage kt (B: cbf; last A: ca7, the others fa[0..6]) while those of stage kt+1
    // are read (B: nb; last A: na7) and the slot stage kt occupied is refilled with stage kt+NST.  Two steps with the
    // buffer roles exchanged make one pass of the loop, so no fragment is ever moved between registers.
    auto step = [&](bf16x8 (&cbf)[TJ], bf16x8 (&nb)[TJ], bf16x8& ca7, bf16x8& na7, u32x4& cg, u32x4& ng) {
        // This wave's share of stage kt+1 has landed (NST-2 younger stages may stay in flight) and its fragments of stage
        // kt are in registers; after the barrier that holds for every wave, i.e. stage kt+1 is readable and stage kt's
        // slot free.  (The builtin, so that the compiler knows the fragments are in.)
        __builtin_amdgcn_s_waitcnt(waitcnt_imm((NST - 2) * NP, 0));
        __builtin_amdgcn_s_barrier();
        const int nslot = slot + 1 == NST ? 0 : slot + 1;
        const char* sA = smem + nslot * STAGE;
        const char* sB = sA + G::A_BYTES;
        char* wst = smem + slot * STAGE;
        slot = nslot;
        __amdgpu_buffer_rsrc_t rsa, rsb;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the row's LDS reads and its refill piece are spread between its MFMAs (a wave issues in order: whatever
            // waits for a queue slot holds back the MFMAs behind it)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_inplace(acc[i][j], cbf[j], i < 7 ? fa[i] : ca7);     // operands swapped: see stage_acc
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0 && i < TJ) nb[i] = read_b(sB, i);
                if (j == TJ / 2 - 1) {
                    if (i > 0) fa[i - 1] = read_a(sA, i - 1);              // (the last A fragment lives in ca7 / na7)
                    else { rsa = rsrc_at(refA, remA); refA += stepA; remA -= stepA; }
                }
                if (j == TJ / 2 && i == 0) { rsb = rsrc_at(refB, remB); refB += stepB; remB -= stepB; }
                if (j == TJ - 3 && i == 7) na7 = read_a(sA, 7);
                if (j == TJ - 2 && i < NP) {
                    if (i < PA) dma_piece(rsa, offa[i], wst + (4 * i + wave) * 1024);
                    else dma_piece(rsb, offb[i - PA], wst + G::A_BYTES + (4 * (i - PA) + wave) * 1024);
                }
            }
            if constexpr (KVEC != 0) {
                // the vector's product for this row block, before the block's A fragment is replaced (next row, j = TJ/2 - 1);
                // the next step's 32 values are requested once per step (a read past the range fetches ring bytes nobody uses)
                if (do_kvec) {
                    if ((i & 1) == wc) mfma_inplace_v(bacc[i >> 1], cg, i < 7 ? fa[i] : ca7);
                    if constexpr (KVEC == 1) {
                        if (i == 0) { ng = lds_read_b128_opaque(gaddr); gaddr += gstride; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(fb[0], fb[1], fa[7], fa7, gf[0], gf[1]);
        step(fb[1], fb[0], fa7, fa[7], gf[1], gf[0]);
    }
    if (kt < nk) step(fb[0], fb[1], fa[7], fa7, gf[0], gf[1]);
    __builtin_amdgcn_s_waitcnt(0xC07F);          // the fragment reads of the stage past the end are in
    // Those last reads feed nothing, and to the compiler an inline-assembly read is complete when issued: without the
    // uses below it hands their destination registers to the next value at once -- in the last step that was the
    // 64-bit constant the refill descriptors' byte count is clamped against, which the landing LDS data then
    // overwrote: an unbounded descriptor on exactly the refills that reach past the end of the operand.
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa[i])));
    asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa7)));
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb[0][j])));
        asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb[1][j])));
    }
    if constexpr (KVEC != 0) { asm volatile("" ::"v"(gf[0])); asm volatile("" ::"v"(gf[1])); }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");             // the last MFMAs (opaque to the compiler) have written back
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // trailing refills and fragment reads are done before LDS is reused
    __syncthreads();
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();

    if constexpr (KVEC != 0) {
        // row 0 of each extra tile: lanes 0..15 hold element 0 for the block's 16 rows
        if (do_kvec && lane < 16) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int m = m0 + wr * 128 + (2 * ii + wc) * 16 + lane;
                if (m < p.M) p.kvec_out[(long)blockIdx.y * p.kvec_out_stride + m] = bacc[ii][0];
            }
        }
    }
    // the wave's block goes through its private 64x64 LDS tile one 64x64 piece at a time
    float* cs = reinterpret_cast<float*>(smem + wave * CS_WAVE_BYTES);
#pragma unroll
    for (int g = 0; g < TJ / 4; ++g) {
        const int nw = n0 + wc * WCOLS + g * 64;
        if constexpr (TJ != 4) load_col_consts<EPI>(p, nw, 0, lane, cb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(cs + (i * 16 + (lane & 15)) * CS_LD + j * 16 + 4 * (lane >> 4)) = acc[4 * h + i][4 * g + j] + cb[j];
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): own wave's LDS writes landed
            __builtin_amdgcn_wave_barrier();
            if (p.stamps && tid == 0 && g == 0) p.stamps[(long)blockIdx.x * 8 + 3 + 2 * h] = __builtin_amdgcn_s_memtime();
            run_epilogue_at<EPI>(p, cs, m0 + wr * 128 + h * 64, nw, nw >> 6, lane);
            if (p.stamps && tid == 0 && g == 0 && h == 0) p.stamps[(long)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime();
        }
    }
    if (p.stamps && tid == 0) p.stamps[(long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------ exact-f32 kernel (validation mode)
template <typename T, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FK = 16, FLD = 132;
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + FK * FLD;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    map_tile(tm, tn, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tid >> 4, tx = tid & 15;
    const T* A = (const T*)p.A;
    const T* B = (const T*)p.B;
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < p.K; k0 += FK) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + NTHREADS * i;
            int mm, kk;
            if (p.a_kmajor) { kk = e >> 7; mm = e & 127; } else { mm = e >> 4; kk = e & 15; }
            float v = 0.f;
            if (m0 + mm < p.M && k0 + kk < p.K)
                v = to_f32<T>(p.a_kmajor ? A[(long)(k0 + kk) * p.lda + m0 + mm] : A[(long)(m0 + mm) * p.lda + k0 + kk]);
            As[kk * FLD + mm] = v;
            int nn;
            if (p.b_kmajor) { kk = e >> 7; nn = e & 127; } else { nn = e >> 4; kk = e & 15; }
            v = 0.f;
            if (n0 + nn < p.N && k0 + kk < p.K)
                v = to_f32<T>(p.b_kmajor ? B[(long)(k0 + kk) * p.ldb + n0 + nn] : B[(long)(n0 + nn) * p.ldb + k0 + kk]);
            Bs[kk * FLD + nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FK; ++kk) {
            float a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = As[kk * FLD + ty * 8 + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = Bs[kk * FLD + tx * 8 + j];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int mm = ty * 8 + i, nn = tx * 8 + j;
            const int w = (mm >> 6) * 2 + (nn >> 6);
            const float past = (EPI == EPI_CE_FWD || EPI == EPI_CE_BWD) ? -INFINITY : 0.f;
            const float cbias = n0 + nn < p.N ? (p.bias ? p.bias[n0 + nn] : 0.f) : past;
            reinterpret_cast<float*>(smem + w * CS_WAVE_BYTES)[(mm & 63) * CS_LD + (nn & 63)] = acc[i][j] + cbias;
        }
    __syncthreads();
    run_epilogue<EPI>(p, smem, m0, n0, tile_n, wave, lane);
}

// ------------------------------------------------------------------ launch
unsigned long long* g_stamps = nullptr;
thread_local bool t_kvec_done = false;      // the last launch_gemm formed A . kvec next to the product
constexpr long WIDE_MIN_WGS = 218;   // 256x256 tiles (one workgroup per CU) pay off from 85 % of the 256 CUs busy
int g_extra_lds = 0;                 // diagnostic: LDS bytes added to every ring launch (lowers the workgroups per CU)
long g_dbg_a_panel = 0;  // diagnostic: the next 256x256 pipelined launches read A as a column-panel image with this many rows per panel
int g_route_small = 1;   // the two routing rules for short products below (diagnostic hook capdecdbg_gemm_route_small: 0 = without)
int g_gemm_mode = 0;     // diagnostic: force a kernel (1 register-staged, 2 ring 128x128, 3 ring 256x128 8 waves, 4 ring 256x128 4 waves, 5 ring 256x256, 6 pipelined 256x256, 7 pipelined 256x128)
int g_ppipe = 1;         // the persistent pipelined kernel takes the products it pays for (diagnostic hook capdecdbg_gemm_ppipe: 0 = never)
// where the persistent pipelined kernel is the faster choice: at least two full rounds of 256x256 tiles on the 256 CUs
// (a workgroup's ring then runs across several tiles) and a reduction of at least 512
static bool ppipe_pays(const GemmArgs& a) {
    const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    return tiles >= 512 && a.K >= 512;
}

template <int EPI>
int launch_gemm(const GemmArgs& a_in, int in_dtype, hipStream_t st) {
    GemmArgs a = a_in;
    a.stamps = g_stamps;
    const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
    const long nwg = (long)tm * tn;
    if (nwg <= 0) return CAPDEC_OK;
    CAPDEC_CHECK(nwg < (1L << 31), "gemm: too many tiles");
    dim3 grid((unsigned)nwg, a.split_k > 1 ? a.split_k : 1), block(NTHREADS);
    if (in_dtype == CAPDEC_BF16) {
        CAPDEC_CHECK((a.lda % 8) == 0 && (a.ldb % 8) == 0, "gemm(bf16): lda/ldb must be multiples of 8 (got %ld, %ld)", a.lda, a.ldb);
        CAPDEC_CHECK(aligned16(a.A) && aligned16(a.B), "gemm(bf16): A/B must be 16-byte aligned");
        // Kernel choice.  The LDS-DMA ring needs whole 32-deep k-tiles, except when both operands are
        // reduction-major (their k-tail lies past the end of the buffer descriptor and reads as zero).
        // Row-major products always take it; with reduction-major operands (two transposing LDS reads per
        // fragment) it wins where the grid is large and K deep (the vocabulary-sized dgrad / wgrad), while
        // the small split-K weight gradients of the block measured faster on the register-staged kernel.
        // The 256-row geometry needs at least one workgroup per CU.
        enum { KERN_REG, KERN_RING128, KERN_TALL, KERN_BIG, KERN_WIDE, KERN_QUAD, KERN_TALLP };
        const bool any_km = a.a_kmajor || a.b_kmajor;
        const bool ring_ok = a.K >= DBK && ((a.K % DBK) == 0 || (a.a_kmajor && a.b_kmajor));
        const int split = a.split_k > 1 ? a.split_k : 1;
        const long big_tiles = (long)((a.M + 255) / 256) * tn;
        int kern = KERN_REG;
        if (ring_ok && !a.a_kmajor) {
            if (big_tiles * split >= 256 && (!a.b_kmajor || a.K >= 512)) kern = KERN_TALL;
            else if (!a.b_kmajor || (a.K >= 2048 && split == 1)) kern = KERN_RING128;
        } else if (ring_ok) {
            if (big_tiles >= 512) kern = KERN_TALL;
            else if (nwg >= 256 && a.K >= 2048 && split == 1) kern = KERN_RING128;
        }
        const long wide_tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
        // deep-K plain products: 256x256 tiles, one workgroup per CU, as soon as every CU gets one -- four waves of
        // 128x128 (measured 4-9 % faster than eight of 128x64 from K = 1536 on), except for outputs of a handful of
        // tiles, whose split-K slabs are mostly epilogue
        if (EPI == EPI_STD && ring_ok && a.K >= 1536 && wide_tiles * split >= WIDE_MIN_WGS) {
            if (wide_tiles >= 16) kern = KERN_QUAD;
            else if (a.K >= 2048) kern = KERN_WIDE;
            // slabs of fewer than 40 k-steps (a block's weight gradients over the 8192 scored rows of a batch of 256, cut into
            // ten slabs): the 8-wave ring, whose tile has less to fill and drain (53 -> 47 us alone)
            if (g_route_small && kern == KERN_QUAD && a.K >= 2048 && a.K / split < 1280) kern = KERN_WIDE;
        }
        // a single round of tall tiles over a reduction of 48 to 128 k-steps with a reduction-major second operand (linear1's
        // data gradient on those rows): the pipelined 256x128 kernel (43 -> 39 us alone; both together -17 us in the step)
        if (g_route_small && EPI == EPI_STD && ring_ok && !a.a_kmajor && a.b_kmajor && split == 1 && a.K >= 1536 && a.K <= 4096 &&
            big_tiles >= 128 && big_tiles < 512) kern = KERN_TALLP;
        if (g_gemm_mode == 1) kern = KERN_REG;
        else if (g_gemm_mode == 5 && ring_ok && (EPI == EPI_STD || !any_km)) kern = KERN_WIDE;
        else if (g_gemm_mode == 6 && ring_ok && EPI == EPI_STD) kern = KERN_QUAD;
        else if (g_gemm_mode == 7 && ring_ok && (EPI == EPI_STD || !any_km || (EPI == EPI_STD_SIDE && !a.a_kmajor))) kern = KERN_TALLP;
        else if (g_gemm_mode == 2 && ring_ok) kern = KERN_RING128;
        else if (g_gemm_mode == 3 && ring_ok && !any_km) kern = KERN_BIG;
        else if (g_gemm_mode == 4 && ring_ok) kern = KERN_TALL;
#define CAPDEC_RING_LAUNCH(AT, BT, WM, WN, TI, NST, NWG)                                               \
    do {                                                                                               \
        using G = DmaGeo<WM, WN, TI, NST>;                                                             \
        auto kfn = gemm_bf16_dma_kernel<AT, BT, EPI, WM, WN, TI, NST>;                                 \
        ensure_dynamic_lds((const void*)kfn, 160 * 1024); \
        const int lds_bytes = G::LDS + g_extra_lds <= 160 * 1024 ? G::LDS + g_extra_lds : G::LDS;      \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(NWG), split), dim3(64 * G::NW), lds_bytes, st, a);    \
    } while (0)
#define CAPDEC_GEMM_CASE(AT, BT)                                                                       \
    do {                                                                                               \
        if (kern == KERN_RING128) CAPDEC_RING_LAUNCH(AT, BT, 2, 2, 4, 4, nwg);                            \
        else {                                                                                         \
            ensure_dynamic_lds((const void*)gemm_bf16_kernel<AT, BT, EPI>, LDS_BYTES); \
            hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, EPI>), grid, block, LDS_BYTES, st, a);        \
        }                                                                                              \
    } while (0)
#define CAPDEC_PIPE_LAUNCH(AT, BT, E, TJ, NWG)                                                         \
    do {                                                                                               \
        auto kfn = gemm_bf16_pipe_kernel<AT, BT, E, TJ>;                                               \
        ensure_dynamic_lds((const void*)kfn, 160 * 1024); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(NWG), split), dim3(256), PipeGeo<TJ>::LDS, st, a);     \
    } while (0)
#define CAPDEC_QUAD_LAUNCH(AT, BT) CAPDEC_PIPE_LAUNCH(AT, BT, EPI_STD, 8, wide_tiles)
#define CAPDEC_QUAD_KVEC_LAUNCH(MODE)                                                                  \
    do {                                                                                               \
        auto kfn = gemm_bf16_pipe_kernel<true, true, EPI_STD, 8, MODE>;                                \
        ensure_dynamic_lds((const void*)kfn, 160 * 1024); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(wide_tiles), split), dim3(256), PipeGeo<8>::LDS + (MODE == 1 ? KVEC_BYTES : 0), st, a); \
        t_kvec_done = true;                                                                            \
    } while (0)
        if (g_dbg_a_panel && kern == KERN_QUAD) a.a_panel = g_dbg_a_panel;
        if (a.a_panel) {      // only the pipelined 256x256 kernel reads such an image
            CAPDEC_CHECK(EPI == EPI_STD && (a.K % DBK) == 0 && a.K >= DBK, "gemm: a column-panel image of A goes with plain products, K %% 32 == 0");
            kern = KERN_QUAD;
        }
        if (kern == KERN_TALLP) {
            if constexpr (EPI == EPI_STD) {
                if (a.a_kmajor && a.b_kmajor) CAPDEC_PIPE_LAUNCH(true, true, EPI_STD, 4, big_tiles);
                else if (a.a_kmajor) CAPDEC_PIPE_LAUNCH(true, false, EPI_STD, 4, big_tiles);
                else if (a.b_kmajor) CAPDEC_PIPE_LAUNCH(false, true, EPI_STD, 4, big_tiles);
                else CAPDEC_PIPE_LAUNCH(false, false, EPI_STD, 4, big_tiles);
            } else if constexpr (EPI == EPI_STD_SIDE) {
                if (a.b_kmajor) CAPDEC_PIPE_LAUNCH(false, true, EPI_STD_SIDE, 4, big_tiles);      // the dgrad through an activation gate
                else CAPDEC_PIPE_LAUNCH(false, false, EPI_STD_SIDE, 4, big_tiles);
            } else {
                CAPDEC_PIPE_LAUNCH(false, false, EPI, 4, big_tiles);
            }
        } else
        if (kern == KERN_QUAD) {
            if constexpr (EPI == EPI_STD) {
                // the vector product rides along when a workgroup's k range fits the LDS behind the ring
                const bool kvec_fits = a.kvec && a.kvec_out && ((a.K + DBK - 1) / DBK + split - 1) / split * DBK <= 8192;
                if (a.a_kmajor && a.b_kmajor && kvec_fits) CAPDEC_QUAD_KVEC_LAUNCH(1);
                else if (a.a_kmajor && a.b_kmajor && !a.kvec && a.kvec_out) CAPDEC_QUAD_KVEC_LAUNCH(2);       // a vector of ones
                else if (a.a_kmajor && a.b_kmajor) CAPDEC_QUAD_LAUNCH(true, true);
                else if (a.a_kmajor) CAPDEC_QUAD_LAUNCH(true, false);
                else if (a.b_kmajor) CAPDEC_QUAD_LAUNCH(false, true);
                else CAPDEC_QUAD_LAUNCH(false, false);
            }
        } else if (kern == KERN_WIDE) {
            if constexpr (EPI == EPI_STD) {
                if (a.a_kmajor && a.b_kmajor) CAPDEC_RING_LAUNCH(true, true, 2, 4, 8, 4, wide_tiles);
                else if (a.a_kmajor) CAPDEC_RING_LAUNCH(true, false, 2, 4, 8, 4, wide_tiles);
                else if (a.b_kmajor) CAPDEC_RING_LAUNCH(false, true, 2, 4, 8, 4, wide_tiles);
                else CAPDEC_RING_LAUNCH(false, false, 2, 4, 8, 4, wide_tiles);
            } else {
                CAPDEC_RING_LAUNCH(false, false, 2, 4, 8, 4, wide_tiles);
            }
        } else if (kern == KERN_TALL) {
            if (a.a_kmajor && a.b_kmajor) CAPDEC_RING_LAUNCH(true, true, 2, 2, 8, 3, big_tiles);
            else if (a.a_kmajor) CAPDEC_RING_LAUNCH(true, false, 2, 2, 8, 3, big_tiles);
            else if (a.b_kmajor) CAPDEC_RING_LAUNCH(false, true, 2, 2, 8, 3, big_tiles);
            else CAPDEC_RING_LAUNCH(false, false, 2, 2, 8, 3, big_tiles);
        } else if (kern == KERN_BIG) CAPDEC_RING_LAUNCH(false, false, 4, 2, 4, 4, big_tiles);
        else if (!a.a_kmajor && !a.b_kmajor) CAPDEC_GEMM_CASE(false, false);
        else if (!a.a_kmajor && a.b_kmajor) CAPDEC_GEMM_CASE(false, true);
        else if (a.a_kmajor && !a.b_kmajor) CAPDEC_GEMM_CASE(true, false);
        else CAPDEC_GEMM_CASE(true, true);
#undef CAPDEC_GEMM_CASE
#undef CAPDEC_RING_LAUNCH
#undef CAPDEC_QUAD_LAUNCH
#undef CAPDEC_QUAD_KVEC_LAUNCH
#undef CAPDEC_PIPE_LAUNCH
    } else {
        ensure_dynamic_lds((const void*)gemm_f32_kernel<float, EPI>, LDS_BYTES);
        hipLaunchKernelGGL((gemm_f32_kernel<float, EPI>), grid, block, LDS_BYTES, st, a);
    }
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

// ------------------------------------------------------------------ CE finalize kernels
// Fold the per-slot partials into lse / argmax / margin / row loss.  A block owns 32 rows; thread
// (row = t & 31, group = t >> 5) folds every 32nd slot of its row with an online rescaling of the sum
// (the 32 rows of a slot are 512 contiguous bytes), and the first 32 threads merge the 32 groups in a
// fixed order (deterministic).  32 rows per block puts 256 blocks on the chip at 8192 rows.
struct RowFold { float m1; int a1; float m2; float s; };
// Which of two slot maxima leads.  PACKED: the records come from the persistent kernel, whose maxima carry their column
// inside the slot in the low 6 mantissa bits -- values are compared without those bits (so that equal logits in
// different slots are equal here, whatever their position in the slot), ties go to the lower column.
__device__ __forceinline__ bool leads(float pm, int pa, float m1, int a1, bool packed) {
    if (packed) {
        pm = __uint_as_float(__float_as_uint(pm) & 0xFFFFFFC0u);
        m1 = __uint_as_float(__float_as_uint(m1) & 0xFFFFFFC0u);
    }
    return pm > m1 || (pm == m1 && pa < a1);
}
__device__ __forceinline__ void fold_push(RowFold& t, float pm, int pa, float p2, float ps, bool packed) {
    if (leads(pm, pa, t.m1, t.a1, packed)) {
        t.s = t.s * (t.m1 > -INFINITY ? __expf(t.m1 - pm) : 0.f) + ps;
        t.m2 = fmaxf(t.m1, p2); t.m1 = pm; t.a1 = pa;
    } else {
        if (pm > -INFINITY) t.s += ps * __expf(pm - t.m1);
        t.m2 = fmaxf(t.m2, pm);
    }
}
__device__ __forceinline__ void fold_push_ref(RowFold& t, float pm, int pa, float p2, float ps) {
    if (leads(pm, pa, t.m1, t.a1, true)) { t.m2 = fmaxf(t.m1, p2); t.m1 = pm; t.a1 = pa; }
    else t.m2 = fmaxf(t.m2, pm);
    t.s += ps;
}
constexpr int FIN_ROWS = 32, FIN_GROUPS = 32;
// dot = NULL: the label's logit comes from zlabel[] (written by the forward epilogue).  Otherwise it is formed here,
// z[m, label] = h[m, :] . W[label, :] + bias[label] (bf16 operands, f32 sum in a fixed order): thread (row, group)
// takes every 32nd 16-byte chunk of the two rows.
struct LabelDot { const bf16_t* h; const bf16_t* W; const float* bias; long ldh, ldw; int d; };
// REF: the partials were taken against a reference value per row, zlabel[m] = the label's logit (a value every row's
// maximum is at least equal to): a slot holds {max (z - ref), sum exp(z - ref)}, the sums add up as they are, and
// lse = ref + log(sum), row loss = log(sum).  rowflag[m] = 1 where the row's maximum lies so far above the reference
// that numerators relative to it leave the range kept safe for the products of the backward (or the sum overflowed):
// those rows are redone against their maximum by ce_ref_fixup_kernel.
constexpr float CE_REF_LIMIT = 50.f;
template <bool REF>
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const f32x4* part4, const float* zlabel, const int64_t* labels,
                                                           int ce_B, int ce_L, long M, int np,
                                                           float* lse_out, int32_t* argmax_out, int64_t* ids_out, float* margin_out,
                                                           float* rowloss, int32_t* rowcorrect, LabelDot dot, int32_t* rowflag, int packed) {
    __shared__ RowFold sh[FIN_GROUPS][FIN_ROWS];
    __shared__ float zdot[FIN_GROUPS][FIN_ROWS + 1];
    const int r = threadIdx.x & (FIN_ROWS - 1), w = threadIdx.x / FIN_ROWS;
    const long m = (long)blockIdx.x * FIN_ROWS + r;
    if (dot.h && labels) {
        // lanes of a wave: 32 rows x 2 groups; a row's 32 groups read 32 consecutive chunks
        float a = 0.f;
        if (m < M) {
            const long l = m / ce_B, b = m - l * ce_B;
            const bf16_t* hr = dot.h + m * dot.ldh;
            const bf16_t* wr = dot.W + labels[b * ce_L + l] * dot.ldw;
            for (int c = w * 8; c < dot.d; c += FIN_GROUPS * 8) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(hr + c), y = *reinterpret_cast<const bf16x8*>(wr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf((float)x[e], (float)y[e], a);
            }
        }
        zdot[w][r] = a;
    }
    RowFold t = {-INFINITY, 0x7fffffff, -INFINITY, 0.f};
    if (m < M) {
        // FIN_UNROLL loads in flight before the first fold: the pass is a latency-bound read of np * 16 bytes per row
        constexpr int FIN_UNROLL = REF ? 13 : 6;      // (the 8-byte records: a thread's 25 slots of a 50k vocabulary in two batches)
        for (int sl = w; sl < np; sl += FIN_UNROLL * FIN_GROUPS) {
            f32x4 q[FIN_UNROLL];
#pragma unroll
            for (int u = 0; u < FIN_UNROLL; ++u)
                if (sl + u * FIN_GROUPS < np) {
                    if constexpr (REF) {
                        // 8-byte records {packed maximum, sum}: the argmax is the slot's first column + 63 - (the low 6 bits)
                        const f32x2 r2 = reinterpret_cast<const f32x2*>(part4)[(long)(sl + u * FIN_GROUPS) * M + m];
                        const int arg = (sl + u * FIN_GROUPS) * 64 + 63 - (int)(__float_as_uint(r2[0]) & 63u);
                        q[u] = f32x4{r2[0], r2[1], -INFINITY, __int_as_float(arg)};
                    } else {
                        q[u] = part4[(long)(sl + u * FIN_GROUPS) * M + m];
                    }
                }
#pragma unroll
            for (int u = 0; u < FIN_UNROLL; ++u)
                if (sl + u * FIN_GROUPS < np) {
                    if constexpr (REF) fold_push_ref(t, q[u][0], __float_as_int(q[u][3]), q[u][2], q[u][1]);
                    else fold_push(t, q[u][0], __float_as_int(q[u][3]), q[u][2], q[u][1], packed != 0);
                }
        }
    }
    sh[w][r] = t;
    __syncthreads();
    if (w == 0 && m < M) {
#pragma unroll 1
        for (int i = 1; i < FIN_GROUPS; ++i) {
            const RowFold o = sh[i][r];
            if constexpr (REF) fold_push_ref(t, o.m1, o.a1, o.m2, o.s);
            else fold_push(t, o.m1, o.a1, o.m2, o.s, packed != 0);
        }
        const float lse = REF ? zlabel[m] + logf(t.s) : t.m1 + logf(t.s);
        if constexpr (REF) rowflag[m] = (t.m1 > CE_REF_LIMIT || !(t.s < 3.0e38f)) ? 1 : 0;
        if (lse_out) lse_out[m] = lse;
        if (argmax_out) argmax_out[m] = t.a1;
        if (ids_out) ids_out[m] = t.a1;
        if (margin_out) margin_out[m] = t.m1 - t.m2;
        if (labels) {
            const long l = m / ce_B, b = m - l * ce_B;
            const long lab = labels[b * ce_L + l];
            float zl;
            if (dot.h) {
                zl = dot.bias ? dot.bias[lab] : 0.f;
#pragma unroll 1
                for (int i = 0; i < FIN_GROUPS; ++i) zl += zdot[i][r];
            } else {
                zl = zlabel[m];
            }
            rowloss[m] = REF ? logf(t.s) : lse - zl;
            rowcorrect[m] = ((long)t.a1 == lab) ? 1 : 0;
        }
    }
}

// fixed-order sum of the per-row losses (deterministic)
__global__ void ce_reduce_kernel(const float* rowloss, const int32_t* rowcorrect, long M, float* loss_sum, int32_t* correct) {
    __shared__ float sf[16];
    __shared__ int si[16];
    float a = 0.f;
    int c = 0;
    for (long i = threadIdx.x; i < M; i += blockDim.x) { a += rowloss[i]; c += rowcorrect[i]; }
    a = wave_sum(a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) { sf[threadIdx.x >> 6] = a; si[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f; int rc = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { r += sf[i]; rc += si[i]; }
        loss_sum[0] = r; correct[0] = rc;
    }
}

// ---- cross-entropy against a per-row reference (the label's logit): kernels around the persistent forward
// z[m, label(m)] = h[m, :] . W[label, :] + bias[label], one wave per row (bf16 operands, f32 sum in a fixed order)
__global__ __launch_bounds__(256) void label_logit_kernel(const bf16_t* h, long ldh, const bf16_t* W, long ldw, const float* bias, int d,
                                                           const int64_t* labels, int ce_B, int ce_L, long M, float* out) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const long l = m / ce_B, b = m - l * ce_B;
    const long lab = labels[b * ce_L + l];
    const bf16_t* hr = h + m * ldh;
    const bf16_t* wr = W + lab * ldw;
    float a = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(hr + c), y = *reinterpret_cast<const bf16x8*>(wr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf((float)x[e], (float)y[e], a);
    }
    a = wave_sum(a);
    if (lane == 0) out[m] = a + (bias ? bias[lab] : 0.f);
}

// Rows the finalize pass flagged (numerators against the label's logit out of the safe range: a label the model gives
// a probability below e^-50) are redone here against their own maximum, one workgroup per row and plain dot products:
// slow, and expected to run for no row at all.  Rewrites the row of numerators, its reference, lse and loss.
// element (m, v) of the numerators: a matrix [M][ld], or (prows > 0) a column-panel image [ld / 32][prows][32]
__device__ __forceinline__ long probs_index(long m, long v, long ld, long prows) {
    return prows ? ((v >> 5) * prows + m) * 32 + (v & 31) : m * ld + v;
}
constexpr int FIX_ROWS = 64;
__global__ __launch_bounds__(256) void ce_ref_fixup_kernel(const int32_t* rowflag, long M, const bf16_t* h, long ldh, const bf16_t* W, long ldw,
                                                            const float* bias, int d, int V, bf16_t* probs, long ld, long prows,
                                                            float* rowref, float* lse, float* rowloss) {
    __shared__ float sm[4], ss[4];
    // a workgroup looks after FIX_ROWS rows; normally none is flagged, which one look at all its flags settles (walking them
    // one dependent load after the other cost 10 us per launch)
    {
        const long mf = (long)blockIdx.x * FIX_ROWS + threadIdx.x;
        const int flagged = (threadIdx.x < FIX_ROWS && mf < M) ? rowflag[mf] : 0;
        if (!__syncthreads_or(flagged)) return;
    }
    for (long m = (long)blockIdx.x * FIX_ROWS; m < min(M, ((long)blockIdx.x + 1) * FIX_ROWS); ++m) {
    if (!rowflag[m]) continue;
    __syncthreads();                                                  // the previous flagged row's shared sums are consumed
    const bf16_t* hr = h + m * ldh;
    auto logit = [&](int v) {
        const bf16_t* wr = W + (long)v * ldw;
        float a = 0.f;
        for (int c = 0; c < d; c += 8) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(hr + c), y = *reinterpret_cast<const bf16x8*>(wr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf((float)x[e], (float)y[e], a);
        }
        return a + (bias ? bias[v] : 0.f);
    };
    const float zl = rowref[m];
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, logit(v));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float sum = 0.f;
    for (int v = threadIdx.x; v < ld; v += 256) {
        const float e = v < V ? __expf(logit(v) - mx) : 0.f;
        sum += e;
        probs[probs_index(m, v, ld, prows)] = (bf16_t)e;
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = mx + logf((ss[0] + ss[1]) + (ss[2] + ss[3]));
        lse[m] = l; rowloss[m] = l - zl; rowref[m] = mx;
    }
    }
}

// What the backward products need besides the numerators e[m, v] = exp(z - ref_m):  with g_m = gscale * exp(ref_m - lse_m),
//   d(logits)[m, v] = g_m * e'[m, v],   e' = e except e'[m, label] = e[m, label] - exp(lse_m - ref_m).
// One wave per row: patches the label's numerator, writes rowscale[m] = g_m and hs[m, :] = g_m * h[m, :] (the weight
// gradient's second operand; the data gradient applies g_m to its rows instead).
__global__ __launch_bounds__(256) void ce_ref_bwd_prepare_kernel(bf16_t* probs, long ld, long prows, const bf16_t* h, long ldh, int d,
                                                                  const int64_t* labels, int ce_B, int ce_L, long M,
                                                                  const float* lse, const float* rowref, float gscale,
                                                                  bf16_t* hs, long ldhs, float* rowscale, bf16_t* rowscale_lp) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float e = rowref[m] - lse[m];
    const float g = gscale * __expf(e);
    for (int c = lane * 8; c < d; c += 512) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(h + m * ldh + c);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16_t)(g * (float)x[k]);
        *reinterpret_cast<bf16x8*>(hs + m * ldhs + c) = o;
    }
    if (lane == 0) {
        rowscale[m] = g;
        if (rowscale_lp) rowscale_lp[m] = (bf16_t)g;
        const long l = m / ce_B, b = m - l * ce_B;
        bf16_t* pl = probs + probs_index(m, labels[b * ce_L + l], ld, prows);
        *pl = (bf16_t)((float)*pl - __expf(-e));
    }
}

// out[m] = sum over the split-K slabs of the vector product, in slab order
__global__ void kvec_reduce_kernel(const float* slabs, int split, long M, float* out) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float a = 0.f;
    for (int y = 0; y < split; ++y) a += slabs[(long)y * M + m];
    out[m] = a;
}
// out[y][m] = sum over k chunk y of A(m, k) * v[k] (v NULL: ones), for products whose kernel does not form it on the way.
// A block owns 64 values of m; its 4 waves take every 4th k of the chunk, 8 loads in flight each, and are added in wave order.
__global__ __launch_bounds__(256) void kvec_plain_kernel(const bf16_t* A, long lda, int a_kmajor, long M, int K, const bf16_t* v, float* out, int kchunk) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long m = (long)blockIdx.x * 64 + tx;
    const int k0 = blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
    float a = 0.f;
    if (m < M) {
        auto at = [&](int k) { return (float)(a_kmajor ? A[(long)k * lda + m] : A[m * lda + k]) * (v ? (float)v[k] : 1.f); };
        int k = k0 + ty;
        for (; k + 28 < k1; k += 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = at(k + 4 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        for (; k < k1; k += 4) a += at(k);
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && m < M) out[(long)blockIdx.y * M + m] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

// x[m, :] *= s[m]
template <typename T>
__global__ void scale_rows_kernel(T* x, long ldx, long M, int N, const float* s) {
    const long total = M * N, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long m = i / N, n = i - m * N;
        x[m * ldx + n] = from_f32<T>(to_f32<T>(x[m * ldx + n]) * s[m]);
    }
}

uint32_t keyed_thr(const capdec_dropout& d) { return d.p > 0.f ? dropout_threshold(d.p) : 0u; }


// bias gradient of one vocabulary slab: dbias[n] = sum over 64-row groups of the partial column sums.
// A block owns 64 columns; its 16 row groups each add every 16th partial row (4 rows in flight per
// thread), and the 16 partial results are added in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float* part, int nrows, long ld, int nv, float* out, int accumulate) {
    __shared__ float sh[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < nv) {
        int r = rg;
        for (; r + 48 < nrows; r += 64) {
            a0 += part[(long)r * ld + n];
            a1 += part[(long)(r + 16) * ld + n];
            a2 += part[(long)(r + 32) * ld + n];
            a3 += part[(long)(r + 48) * ld + n];
        }
        for (; r < nrows; r += 16) a0 += part[(long)r * ld + n];
    }
    sh[rg][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && n < nv) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += sh[i][c];
        out[n] = accumulate ? out[n] + a : a;
    }
}

// d(logits) from the stored softmax numerators e = exp(z - slot max), in place:
//   dl = e * exp(slot max - lse) * gscale - [column == label] * gscale.
// A block owns 64 rows x 256 columns (4 slots): first one thread per (row, slot) forms the factor from
// the forward's partials (rows contiguous across lanes), then a thread owns 8 columns of 8 rows.
// Column sums of the block's 64 rows go to colpart[blockIdx.y][ld] (bias gradient, folded afterwards).
template <typename T>
__global__ __launch_bounds__(256) void ce_grad_from_probs_kernel(T* pd, long ld, long M, const f32x4* part4, int np, const float* lse,
                                                                  const int64_t* labels, int ce_B, int ce_L, float gscale,
                                                                  float* colpart, long ld_colpart) {
    __shared__ float fs[4][64];
    __shared__ int labs[64];
    __shared__ float red[8][32][8];
    const int t = threadIdx.x;
    const long m0 = (long)blockIdx.y * 64;
    const int n0 = blockIdx.x * 256;
    {
        const int r = t & 63, sl = t >> 6;
        const long m = m0 + r;
        const int slot = (n0 >> 6) + sl;
        float f = 0.f;
        if (m < M && slot < np) {
            const float sm = part4[(long)slot * M + m][0];
            if (sm > -INFINITY) f = __builtin_amdgcn_exp2f((sm - lse[m]) * LOG2E) * gscale;
        }
        fs[sl][r] = f;
        if (sl == 0) {
            int lab = -1 - n0;
            if (m < M) { const long l = m / ce_B, b = m - l * ce_B; lab = (int)labels[b * ce_L + l] - n0; }
            labs[r] = lab;
        }
    }
    __syncthreads();
    const int cc = t & 31, rg = t >> 5;
    const long n = n0 + cc * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (n < ld) {
#pragma unroll 2
        for (int i = 0; i < 8; ++i) {
            const int r = rg + 8 * i;
            const long m = m0 + r;
            if (m >= M) continue;
            T* ptr = pd + m * ld + n;
            float v[8];
            if constexpr (sizeof(T) == 2) {
                const bf16x8 q = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(ptr));      // streamed once: keep it out of the caches
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)q[e];
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ptr), b = *reinterpret_cast<const f32x4*>(ptr + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            }
            const float f = fs[cc >> 3][r];
            const int lab = labs[r] - cc * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= f;
            if ((unsigned)lab < 8u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] -= e == lab ? gscale : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
            if constexpr (sizeof(T) == 2) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(ptr));
            } else {
                *reinterpret_cast<f32x4*>(ptr) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(ptr + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
    if (!colpart) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cc][e] = acc[e];
    __syncthreads();
    if (rg == 0 && n < ld) {
        float s8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = acc[e];
#pragma unroll
        for (int w = 1; w < 8; ++w)
#pragma unroll
            for (int e = 0; e < 8; ++e) s8[e] += red[w][cc][e];
        float* dst = colpart + (long)blockIdx.y * ld_colpart + n;
        *reinterpret_cast<f32x4*>(dst) = f32x4{s8[0], s8[1], s8[2], s8[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{s8[4], s8[5], s8[6], s8[7]};
    }
}

// out = (accumulate ? out : 0) + sum over slabs, fixed order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int split, long slab, int M, int N, T* out, long ldc, int accumulate,
                                                            const float* row_scale) {
    // out[m, n] (+)= sum over the slabs, always in slab order; four columns per thread and the loads of four
    // slabs in flight at a time when the shapes allow 16-byte accesses
    const long stride = (long)gridDim.x * blockDim.x;
    if ((N & 3) == 0 && (ldc & 3) == 0 && (slab & 3) == 0 && sizeof(T) == 4 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0) {
        const long total4 = slab >> 2;
        for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += stride) {
            const long i = i4 << 2;
            const long m = i / N, n = i - m * N;
            float* dst = reinterpret_cast<float*>(out) + m * ldc + n;
            f32x4 a = accumulate ? *reinterpret_cast<const f32x4*>(dst) : f32x4{0.f, 0.f, 0.f, 0.f};
            int s2 = 0;
            for (; s2 + 4 <= split; s2 += 4) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (long)s2 * slab + i);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(ws + (long)(s2 + 1) * slab + i);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(ws + (long)(s2 + 2) * slab + i);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(ws + (long)(s2 + 3) * slab + i);
                a = (((a + v0) + v1) + v2) + v3;
            }
            for (; s2 < split; ++s2) a += *reinterpret_cast<const f32x4*>(ws + (long)s2 * slab + i);
            if (row_scale) a *= row_scale[m];          // never together with accumulate (checked on the host)
            *reinterpret_cast<f32x4*>(dst) = a;
        }
        return;
    }
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long m = i / N, n = i - m * N;
        float a = accumulate ? to_f32<T>(out[m * ldc + n]) : 0.f;
        for (int s2 = 0; s2 < split; ++s2) a += ws[s2 * slab + i];
        if (row_scale) a *= row_scale[m];
        out[m * ldc + n] = from_f32<T>(a);
    }
}

// Diagnostic stand-in for a communication kernel that holds compute units while a product runs: every
// workgroup spins until the constant-rate clock (100 MHz) has advanced by `ticks`, then exits.
__global__ __launch_bounds__(256) void occupy_kernel(long ticks, int* sink) {
    extern __shared__ char hold[];
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while ((long)(wall_clock64() - t0) < ticks) { __builtin_amdgcn_s_sleep(8); ++spins; }
    if (sink && spins == 0xffffffffu) { hold[threadIdx.x] = 1; sink[0] = hold[0]; }     // never true: keeps the LDS allocation
}

}  // namespace

// diagnostic hook (not part of the declared ABI): `wgs` workgroups of 256 threads with `lds_bytes` of LDS each
// stay resident for `microseconds` on `stream`
extern "C" int capdecdbg_occupy(int wgs, int lds_bytes, long microseconds, void* stream) {
    if (wgs <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 100000) return 1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
    hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), lds_bytes, (hipStream_t)stream, microseconds * 100, (int*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// diagnostic hook (not part of the declared ABI): clock stamps of every workgroup of the next GEMM launches
extern "C" void capdecdbg_gemm_stamps(void* device_buffer) { g_stamps = (unsigned long long*)device_buffer; }
extern "C" void capdecdbg_gemm_mode(int mode) { g_gemm_mode = mode; }
extern "C" void capdecdbg_gemm_a_panel(long rows) { g_dbg_a_panel = rows; }
extern "C" void capdecdbg_gemm_route_small(int on) { g_route_small = on; }
extern "C" void capdecdbg_gemm_ppipe(int on) { g_ppipe = on; }
extern "C" void capdecdbg_gemm_extra_lds(int bytes) { g_extra_lds = bytes > 0 ? bytes : 0; }

// ==================================================================== C ABI
namespace { thread_local bool t_row_scaled = false; }      // the last product applied its row_scale on the way out
static int gemm_bias_act_impl(const capdec_gemm_desc* d, void* stream);

extern "C" int capdec_gemm_bias_act(const capdec_gemm_desc* d, void* stream) {
    if (!d || (!d->row_scale && !d->kvec_out)) return gemm_bias_act_impl(d, stream);
    CAPDEC_CHECK(!d->accumulate && !d->bias && d->col_scale_n == 0 && d->act == CAPDEC_ACT_NONE && d->drop.p <= 0.f && !d->residual &&
                 !d->gate && !d->aux_out && !d->colsum && !d->kv_cache, "gemm: row_scale / kvec go with plain products only");
    CAPDEC_CHECK((!d->kvec || d->kvec_out) && (!d->kvec_out || d->in_dtype == CAPDEC_BF16), "gemm: kvec needs kvec_out, kvec_out bf16 operands");
    t_row_scaled = false; t_kvec_done = false;
    const int rc = gemm_bias_act_impl(d, stream);
    if (rc || d->M == 0 || d->N == 0) { t_row_scaled = true; t_kvec_done = true; return rc; }
    hipStream_t st = (hipStream_t)stream;
    CAPDEC_CHECK(!(d->kvec_out && !t_kvec_done && d->a_panel_rows), "gemm: kvec with a column-panel image of A needs a k range of at most 8192 per workgroup");
    if (d->kvec_out && !t_kvec_done) {     // the kernel that ran does not form the vector product: a pass over A, in up to 32
                                           // k chunks through the end of the workspace when there is one
        const long kv_bytes = 32L * d->M * 4;
        int chunks = 1;
        float* slabs = d->kvec_out;
        if (d->workspace && d->workspace_bytes >= kv_bytes && d->K >= 4096) {
            chunks = 32;
            slabs = (float*)((char*)d->workspace + ((d->workspace_bytes - kv_bytes) & ~15L));
        }
        const int kchunk = (d->K + chunks - 1) / chunks;
        hipLaunchKernelGGL(kvec_plain_kernel, dim3((unsigned)((d->M + 63) / 64), chunks), dim3(256), 0, st, (const bf16_t*)d->A, d->lda, d->a_kmajor,
                           (long)d->M, d->K, (const bf16_t*)d->kvec, slabs, kchunk);
        if (chunks > 1)
            hipLaunchKernelGGL(kvec_reduce_kernel, dim3((unsigned)((d->M + 255) / 256)), dim3(256), 0, st, (const float*)slabs, chunks, (long)d->M, d->kvec_out);
    }
    if (d->row_scale && !t_row_scaled) {
        // the kernel that ran has no row factor in its epilogue: one more pass over C
        const long total = (long)d->M * d->N;
        const int nb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        if (d->out_dtype == CAPDEC_BF16)
            hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (bf16_t*)d->C, d->ldc, (long)d->M, d->N, d->row_scale);
        else
            hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(nb), dim3(256), 0, st, (float*)d->C, d->ldc, (long)d->M, d->N, d->row_scale);
    }
    CAPDEC_LAUNCH_CHECK();
    t_row_scaled = true; t_kvec_done = true;
    return CAPDEC_OK;
}

static int gemm_bias_act_impl(const capdec_gemm_desc* d, void* stream) {
    CAPDEC_CHECK(d != nullptr, "gemm: null descriptor");
    CAPDEC_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, "gemm: negative dimension");
    CAPDEC_CHECK(d->in_dtype == CAPDEC_BF16 || d->in_dtype == CAPDEC_F32, "gemm: bad in_dtype");
    CAPDEC_CHECK(d->out_dtype == CAPDEC_BF16 || d->out_dtype == CAPDEC_F32, "gemm: bad out_dtype");
    CAPDEC_CHECK(!d->accumulate || d->out_dtype == CAPDEC_F32, "gemm: accumulate needs f32 output");
    CAPDEC_CHECK(d->A && d->B && d->C, "gemm: null operand");
    CAPDEC_CHECK((long)d->M * d->N < (1L << 32) || (d->drop.p <= 0.f && d->gate_mode != CAPDEC_GATE_GELU_PRE),
                 "gemm: dropout index space exceeds 2^32 elements");
    if (d->M == 0 || d->N == 0) return CAPDEC_OK;
    // Plain deep-K products on 256x256 tiles whose last round of workgroups would leave most CUs idle (the vocabulary
    // weight gradient: 591 tiles on 256 CUs = 2.3 rounds billed as 3): the M-tiles of that last round are cut off and
    // run as a second, split-K product through the workspace -- 2 rounds + 1/3 instead of 3.
    // (not with a vector column whose k range has to be cut to fit the kernel's LDS: that product is split along K below)
    const bool kvec_deep = d->kvec && d->kvec_out && d->K > 8192;
    if (d->in_dtype == CAPDEC_BF16 && g_gemm_mode == 0 && d->workspace && d->K >= 2048 && !d->bias && d->col_scale_n == 0 && !kvec_deep &&
        d->act == CAPDEC_ACT_NONE && d->drop.p <= 0.f && !d->residual && !d->gate && !d->aux_out && !d->colsum && !d->kv_cache &&
        ((d->K % 32) == 0 || (d->a_kmajor && d->b_kmajor))) {
        const long tmw = (d->M + 255) / 256, tnw = (d->N + 255) / 256, wide = tmw * tnw, rem = wide % 256;
        const long tail_m = (rem + tnw - 1) / tnw, head_m = tmw - tail_m;
        const long tail_slab = tail_m * 256 * (long)d->N * 4;
        if (wide > 256 && rem > 0 && rem <= 128 && head_m > 0 && tail_m * tnw <= 128 && d->workspace_bytes >= 2 * tail_slab) {
            const long m0 = head_m * 256;
            const long esz = 2, osz = d->out_dtype == CAPDEC_BF16 ? 2 : 4;
            capdec_gemm_desc h1 = *d, t1 = *d;
            h1.M = (int)m0;
            h1.workspace = nullptr; h1.workspace_bytes = 0;
            t1.M = d->M - (int)m0;
            t1.A = (const char*)d->A + (d->a_panel_rows ? (d->a_kmajor ? (m0 >> 5) * d->a_panel_rows * 32 : m0 * 32)
                                                        : (d->a_kmajor ? m0 : m0 * d->lda)) * esz;
            t1.C = (char*)d->C + m0 * d->ldc * osz;
            if (d->row_scale) t1.row_scale = d->row_scale + m0;
            if (d->kvec_out) t1.kvec_out = d->kvec_out + m0;
            int rc = capdec_gemm_bias_act(&h1, stream);
            if (rc) return rc;
            return capdec_gemm_bias_act(&t1, stream);
        }
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.B = d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.a_kmajor = d->a_kmajor; a.b_kmajor = d->b_kmajor;
    a.out_dtype = d->out_dtype;
    a.bias = d->bias; a.col_scale = d->col_scale; a.col_scale_n = d->col_scale_n;
    a.act = d->act;
    a.drop_thr = keyed_thr(d->drop); a.drop_k0 = d->drop.key0; a.drop_k1 = d->drop.key1;
    a.drop_scale = d->drop.p > 0.f ? 1.0f / (1.0f - d->drop.p) : 1.0f;
    a.residual = d->residual; a.ldr = d->ldr;
    a.accumulate = d->accumulate;
    a.gate = d->gate; a.ldg = d->ldg; a.gate_mode = d->gate ? d->gate_mode : CAPDEC_GATE_NONE; a.gate_dtype = d->gate_dtype;
    a.gate_thr = keyed_thr(d->gate_drop); a.gate_k0 = d->gate_drop.key0; a.gate_k1 = d->gate_drop.key1;
    a.gate_scale = d->gate_drop.p > 0.f ? 1.0f / (1.0f - d->gate_drop.p) : 1.0f;
    a.aux_out = d->aux_out; a.ldaux = d->ldaux;
    a.colsum = d->colsum;
    a.kv_cache = d->kv_cache; a.kv_q = d->kv_q; a.kv_d = d->kv_d; a.kv_B = d->kv_B; a.kv_H = d->kv_H; a.kv_hd = d->kv_hd;
    a.kv_smax = d->kv_smax; a.kv_s0 = d->kv_s0;
    a.kvec = (const bf16_t*)d->kvec; a.kvec_out = d->kvec_out; a.kvec_out_stride = 0;
    CAPDEC_CHECK(!d->b_panels || (d->in_dtype == CAPDEC_BF16 && !d->b_kmajor && (d->K % 32) == 0 && d->b_panel_rows >= d->N && aligned16(d->b_panels)),
                 "gemm: a k-panel image of B goes with a row-major bf16 B, K %% 32 == 0, b_panel_rows >= N, 16-byte aligned");
    if (d->a_panel_rows) {
        CAPDEC_CHECK(d->in_dtype == CAPDEC_BF16 && (d->K % 32) == 0 && d->a_panel_rows >= (d->a_kmajor ? d->K : d->M) && !d->bias && d->col_scale_n == 0 &&
                     d->act == CAPDEC_ACT_NONE && d->drop.p <= 0.f && !d->residual && !d->gate && !d->aux_out && !d->colsum && !d->kv_cache,
                     "gemm: a column-panel image of A goes with plain bf16 products, K %% 32 == 0, a_panel_rows >= the stored matrix's rows");
        a.a_panel = d->a_panel_rows;
    }
    if (a.kv_cache) {
        CAPDEC_CHECK(a.kv_d > 0 && a.kv_B > 0 && a.kv_H > 0 && a.kv_hd > 0 && a.kv_H * a.kv_hd == a.kv_d && (a.kv_hd % 4) == 0,
                     "gemm: inconsistent kv scatter geometry");
        CAPDEC_CHECK((a.kv_q == 0 || a.kv_q == a.kv_d) && a.N == a.kv_q + 2 * a.kv_d && (a.M % a.kv_B) == 0 && a.kv_s0 >= 0 && a.kv_s0 + a.M / a.kv_B <= a.kv_smax,
                     "gemm: kv scatter out of range (N=%d, M=%d, s0=%d, smax=%d)", a.N, a.M, a.kv_s0, a.kv_smax);
    }
    // split-K for plain products whose tile grid cannot fill the chip (weight gradients: few tiles, very deep K)
    const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int nkt = (a.K + BK - 1) / BK;
    const bool plain = !a.bias && a.col_scale_n == 0 && a.act == CAPDEC_ACT_NONE && !a.drop_thr && !a.residual &&
                       a.gate_mode == CAPDEC_GATE_NONE && !a.aux_out && !a.colsum && !a.kv_cache;
    // A vector column (kvec) rides on the pipelined kernel only while a workgroup's k range of the vector fits its LDS (8192
    // values): a deeper product with both operands reduction-major is cut along K for that, whatever its tile count
    const bool kvec_split = d->in_dtype == CAPDEC_BF16 && plain && d->workspace && kvec_deep && a.a_kmajor && a.b_kmajor && g_gemm_mode == 0;
    if (d->in_dtype == CAPDEC_BF16 && plain && d->workspace && (tiles <= 512 || kvec_split) && nkt >= 32) {
        // the split that best fills whole rounds of the 512 resident workgroups (two per CU), counted in the
        // tiles of the kernel that will run; the smallest such split, since every slab is written and re-read
        const bool ring_ok = (a.K % 32) == 0 || (a.a_kmajor && a.b_kmajor);
        const long wide = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
        // at most 32 slabs, each at least 8 k-tiles deep, as many as the workspace holds
        const long slab_bytes = (long)a.M * a.N * 4;
        // with a vector product: up to 32 slabs of M floats at the end of the workspace
        const long kv_bytes = a.kvec_out ? 32L * a.M * 4 : 0;
        const long ws_bytes = d->workspace_bytes > kv_bytes ? (d->workspace_bytes - kv_bytes) & ~15L : 0;
        float* kv_slabs = a.kvec_out ? (float*)((char*)d->workspace + ws_bytes) : nullptr;
        long maxsp = nkt / 8 < 32 ? nkt / 8 : 32;
        if (ws_bytes / slab_bytes < maxsp) maxsp = ws_bytes / slab_bytes;
        // 256x256 tiles (one workgroup per CU) when some split gives (nearly) every CU one; else 256x128 / 128x128 tiles
        const bool wide_possible = ring_ok && a.K >= 2048 && wide * maxsp >= WIDE_MIN_WGS;
        const bool tall_possible = !a.a_kmajor && a.K >= 512 && (a.K % 32) == 0;
        const long unit = wide_possible ? wide : (tall_possible ? (long)((a.M + 255) / 256) * ((a.N + BN - 1) / BN) : tiles);
        const long resident = wide_possible ? 256 : 512;
        long split = 1;
        double best = 0.0;
        for (long sp = 1; sp <= maxsp; ++sp) {
            const long n = unit * sp;
            if (wide_possible && n < WIDE_MIN_WGS) continue;
            const double eff = (double)n / (double)((n + resident - 1) / resident * resident);
            if (eff > best + 0.03) { best = eff; split = sp; }
        }
        const long nk32 = (a.K + DBK - 1) / DBK;
        if (kvec_split && (nk32 + split - 1) / split * DBK > 8192) {
            // the fewest further slabs whose k ranges (whole 32-deep steps) hold at most 8192 values
            long fit = 0;
            for (long sp = split + 1; sp <= maxsp && !fit; ++sp)
                if ((nk32 + sp - 1) / sp * DBK <= 8192) fit = sp;
            if (fit) split = fit;                       // else: the workspace does not hold that many slabs, and the vector product takes its own pass
        }
        const long slab = (long)a.M * a.N;
        if (split > 1 && ws_bytes >= split * slab * 4) {
            GemmArgs b = a;
            b.C = d->workspace; b.ldc = a.N; b.out_dtype = CAPDEC_F32; b.accumulate = 0;
            b.split_k = (int)split; b.split_stride = slab;
            if (a.kvec_out) { b.kvec_out = kv_slabs; b.kvec_out_stride = a.M; }
            b.fast_epi = (a.N % 4) == 0 && ((slab * 4) % 16) == 0 && aligned16(d->workspace) ? 4 : 0;
            int rc = launch_gemm<EPI_STD>(b, d->in_dtype, (hipStream_t)stream);
            if (rc) return rc;
            const int nb = (int)((slab + 255) / 256 > 2048 ? 2048 : (slab + 255) / 256);
            if (a.out_dtype == CAPDEC_BF16)
                hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)d->workspace, (int)split, slab, a.M, a.N, (bf16_t*)a.C, a.ldc, a.accumulate, d->row_scale);
            else
                hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)d->workspace, (int)split, slab, a.M, a.N, (float*)a.C, a.ldc, a.accumulate, d->row_scale);
            if (a.kvec_out && t_kvec_done)
                hipLaunchKernelGGL(kvec_reduce_kernel, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                   (const float*)kv_slabs, (int)split, (long)a.M, d->kvec_out);
            CAPDEC_LAUNCH_CHECK();
            if (d->row_scale) t_row_scaled = true;
            return CAPDEC_OK;
        }
    }
    // The large data gradients dY W of a stack (row-major dY, the weight read along its output index: reduction-major second
    // operand, plain bf16 result, at least two rounds of 256x256 tiles): the persistent pipelined kernel of ppipe.hip -- the
    // 4-wave / 128x128 main loop of the pipelined kernel below, the persistent ring and register epilogue of head_fwd.hip.
    // Diagnostic mode 12 forces it wherever it is supported (also on a row-major second operand with a forward epilogue:
    // measured level with the eight-wave persistent kernel below, so not routed), capdecdbg_gemm_ppipe(0) switches it off.
    if (d->in_dtype == CAPDEC_BF16 && a.out_dtype == CAPDEC_BF16 && !a.a_kmajor && !a.residual && !a.accumulate && a.gate_mode == CAPDEC_GATE_NONE &&
        !a.aux_out && !a.colsum && !a.kv_cache && !a.kvec && !a.kvec_out && !d->row_scale && !a.a_panel &&
        (a.act == CAPDEC_ACT_NONE || a.act == CAPDEC_ACT_RELU) &&
        (g_gemm_mode == 12 || (g_gemm_mode == 0 && g_ppipe && a.b_kmajor && ppipe_pays(a)))) {
        HeadFwdArgs ha = {};
        ha.A = (const bf16_t*)a.A; ha.B = (const bf16_t*)a.B; ha.bias = a.bias; ha.C = (bf16_t*)a.C;
        ha.lda = a.lda; ha.ldb = a.ldb; ha.ldc = a.ldc; ha.M = a.M; ha.N = a.N; ha.K = a.K; ha.store_cols = a.N;
        ha.b_tr = a.b_kmajor; ha.mode = 1;
        ha.col_scale = a.col_scale_n > 0 ? a.col_scale : 1.f; ha.col_scale_n = a.col_scale_n;
        ha.relu = a.act == CAPDEC_ACT_RELU;
        ha.drop_thr = a.drop_thr; ha.drop_k0 = a.drop_k0; ha.drop_k1 = a.drop_k1; ha.drop_scale = a.drop_scale;
        if (d->b_panels && !a.b_kmajor) { ha.Bp = (const bf16_t*)d->b_panels; ha.b_rows = d->b_panel_rows; }
        if (ppipe_supported(ha)) return ppipe_launch(ha, (hipStream_t)stream);
    }
    // Large row-major bf16 products with nothing but bias / column scale / ReLU / dropout behind them (the in_proj and
    // linear1 forwards of a stack at full batch) go to the persistent kernel of head_fwd.hip: one operand ring for the
    // whole launch, epilogue on the accumulator registers.  Diagnostic modes 10 / 11 force it at any size (256x256 /
    // 256x128 tiles), any other non-zero mode keeps the tile kernels.
    if (d->in_dtype == CAPDEC_BF16 && a.out_dtype == CAPDEC_BF16 && !a.a_kmajor && !a.b_kmajor && !a.residual && !a.accumulate &&
        a.gate_mode == CAPDEC_GATE_NONE && !a.aux_out && !a.colsum && !a.kv_cache &&
        (a.act == CAPDEC_ACT_NONE || a.act == CAPDEC_ACT_RELU) &&
        (g_gemm_mode == 10 || g_gemm_mode == 11 || (g_gemm_mode == 0 && (long)a.M * a.N >= (1L << 24) && a.K <= 1024))) {
        // (from one full round of 256x256 tiles on: linear1 on the 8192 scored rows of a batch of 256 runs 39.6 -> 31.6 us)
        HeadFwdArgs ha = {};
        ha.A = (const bf16_t*)a.A; ha.B = (const bf16_t*)a.B; ha.bias = a.bias; ha.C = (bf16_t*)a.C;
        ha.lda = a.lda; ha.ldb = a.ldb; ha.ldc = a.ldc; ha.M = a.M; ha.N = a.N; ha.K = a.K; ha.store_cols = a.N;
        ha.mode = 1; ha.col_scale = a.col_scale_n > 0 ? a.col_scale : 1.f; ha.col_scale_n = a.col_scale_n;
        ha.relu = a.act == CAPDEC_ACT_RELU;
        ha.drop_thr = a.drop_thr; ha.drop_k0 = a.drop_k0; ha.drop_k1 = a.drop_k1; ha.drop_scale = a.drop_scale;
        if (d->b_panels) { ha.Bp = (const bf16_t*)d->b_panels; ha.b_rows = d->b_panel_rows; }
        if (head_fwd_supported(ha)) return head_fwd_launch(ha, g_gemm_mode == 11 ? 0 : 1, (hipStream_t)stream);
    }
    // ... and those that add a dropped-out product to an f32 residual stream (the out_proj and linear2 forwards); shallow ones
    // of 192 to 511 tall tiles (out_proj and linear2 on the 8192 scored rows of a batch of 256: 31.3 -> 24.4 and 48.9 -> 41.2 us
    // alone, -4.5 and -6.5 us in the step) take the 256x128 geometry
    const long tall_tiles = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    const bool small_residual = tall_tiles >= 192 && tall_tiles < 512 && a.K <= 2048;
    if (d->in_dtype == CAPDEC_BF16 && a.out_dtype == CAPDEC_F32 && !a.a_kmajor && !a.b_kmajor && a.residual && !a.accumulate &&
        a.gate_mode == CAPDEC_GATE_NONE && !a.aux_out && !a.colsum && !a.kv_cache && a.act == CAPDEC_ACT_NONE && a.col_scale_n == 0 &&
        (g_gemm_mode == 10 || g_gemm_mode == 11 || (g_gemm_mode == 0 && ((long)a.M * a.N >= (1L << 25) || small_residual)))) {
        HeadFwdArgs ha = {};
        ha.A = (const bf16_t*)a.A; ha.B = (const bf16_t*)a.B; ha.bias = a.bias; ha.Cf = (float*)a.C; ha.residual = a.residual;
        ha.lda = a.lda; ha.ldb = a.ldb; ha.ldc = a.ldc; ha.ldr = a.ldr; ha.M = a.M; ha.N = a.N; ha.K = a.K; ha.store_cols = a.N;
        ha.mode = 2;
        ha.drop_thr = a.drop_thr; ha.drop_k0 = a.drop_k0; ha.drop_k1 = a.drop_k1; ha.drop_scale = a.drop_scale;
        if (d->b_panels) { ha.Bp = (const bf16_t*)d->b_panels; ha.b_rows = d->b_panel_rows; }
        if (head_fwd_supported(ha)) return head_fwd_launch(ha, (g_gemm_mode == 11 || (g_gemm_mode == 0 && small_residual)) ? 0 : 1, (hipStream_t)stream);
    }
    // straight-line epilogue: bias / column scale / ReLU / dropout only, every lane's 16-byte group of columns
    // aligned and entirely inside or outside N
    if (!a.residual && !a.accumulate && a.gate_mode == CAPDEC_GATE_NONE && !a.colsum && !a.kv_cache && aligned16(a.C) &&
        (!a.aux_out || aligned16(a.aux_out))) {
        const int w = a.out_dtype == CAPDEC_BF16 ? 8 : 4;
        if ((a.N % w) == 0 && (a.ldc % w) == 0 && (!a.aux_out || (a.ldaux % w) == 0)) a.fast_epi = w;
    }
    // gate-only epilogue (the dgrad through activation + dropout) with 16-byte accesses
    // (column sums then go through the workspace: one partial row per 64 rows, folded in a fixed order)
    const long colpart_bytes = (long)((a.M + 63) / 64) * a.N * 4;
    if (a.gate_mode != CAPDEC_GATE_NONE && !a.residual && !a.accumulate && a.col_scale_n == 0 && a.act == CAPDEC_ACT_NONE && !a.drop_thr &&
        !a.aux_out && !a.kv_cache && a.out_dtype == CAPDEC_BF16 && a.gate_dtype == CAPDEC_BF16 &&
        (a.N % 8) == 0 && (a.ldc % 8) == 0 && (a.ldg % 8) == 0 && aligned16(a.C) && aligned16(a.gate) &&
        (!a.colsum || (d->workspace && aligned16(d->workspace) && d->workspace_bytes >= colpart_bytes))) {
        a.fast_epi = 16;
        if (a.colsum) {
            a.colpart = (float*)d->workspace; a.ld_colpart = a.N; a.colsum = nullptr;
            int rc = launch_gemm<EPI_STD_SIDE>(a, d->in_dtype, (hipStream_t)stream);
            if (rc) return rc;
            hipLaunchKernelGGL(colpart_reduce_kernel, dim3((a.N + 63) / 64), dim3(1024), 0, (hipStream_t)stream,
                               (const float*)d->workspace, (int)((a.M + 63) / 64), (long)a.N, a.N, d->colsum, 1);
            CAPDEC_LAUNCH_CHECK();
            return CAPDEC_OK;
        }
    }
    if (a.residual || a.accumulate || a.gate_mode != CAPDEC_GATE_NONE) return launch_gemm<EPI_STD_SIDE>(a, d->in_dtype, (hipStream_t)stream);
    return launch_gemm<EPI_STD>(a, d->in_dtype, (hipStream_t)stream);
}

extern "C" long capdec_logits_ce_workspace(long M, int V) {
    const long np = 2L * ((V + BN - 1) / BN);
    return (M * np * 4 + M * 4) * 4 + 256;
}

namespace {
struct CeWs { f32x4* part4; float* zlabel; float* rowloss; int32_t* rowcorrect; int32_t* rowflag; int np; };
bool carve_ce_ws(void* ws, long bytes, long M, int V, CeWs& o) {
    if (!ws || bytes < capdec_logits_ce_workspace(M, V) || ((uintptr_t)ws & 15)) return false;
    o.np = 2 * ((V + BN - 1) / BN);
    const long n = M * o.np;
    o.part4 = (f32x4*)ws;
    o.zlabel = (float*)ws + 4 * n; o.rowloss = o.zlabel + M; o.rowcorrect = (int32_t*)(o.rowloss + M); o.rowflag = o.rowcorrect + M;
    return true;
}
void fill_head_args(GemmArgs& a, const void* h, const void* W, const float* bias, long M, int V, int d, long ldh, long ldw) {
    memset(&a, 0, sizeof(a));
    a.A = h; a.B = W; a.M = (int)M; a.N = V; a.K = d; a.lda = ldh; a.ldb = ldw;
    a.bias = bias;
}
// Which geometry of the persistent forward kernel (head_fwd.hip) runs this forward, -1: the generic tile kernel.
// bf16 products large enough to occupy every CU for several tiles take it; diagnostic modes 8 / 9 force its two
// geometries at any size, any other non-zero mode forces the generic kernels.
int g_head_fwd_geometry = 1;
int head_fwd_choice(const GemmArgs& a, int dtype, HeadFwdArgs& ha) {
    if (dtype != CAPDEC_BF16 || a.want_sec) return -1;
    ha = HeadFwdArgs{(const bf16_t*)a.A, (const bf16_t*)a.B, a.bias, (bf16_t*)a.C, a.lda, a.ldb, a.ldc, a.M, a.N, a.K,
                     a.store_cols, a.part4, a.np, g_stamps};
    if (a.C && a.out_dtype != CAPDEC_BF16) return -1;
    if (!head_fwd_supported(ha)) return -1;
    if (g_gemm_mode == 8) return 0;
    if (g_gemm_mode == 9) return 1;
    if (g_gemm_mode != 0) return -1;
    return (long)a.M * a.N >= (1L << 24) ? g_head_fwd_geometry : -1;
}
// k-panel images of the operands: only the persistent kernel reads them (every other kernel takes the row-major operands)
bool panels_ok(const capdec_kpanels* k, long M, long V) {
    if (!k) return true;
    return (!k->a || (aligned16(k->a) && k->a_rows >= M)) && (!k->b || (aligned16(k->b) && k->b_rows >= V));
}
void set_panels(HeadFwdArgs& ha, const capdec_kpanels* k) {
    if (!k) return;
    ha.Ap = (const bf16_t*)k->a; ha.a_rows = k->a_rows;
    ha.Bp = (const bf16_t*)k->b; ha.b_rows = k->b_rows;
}
}  // namespace

// diagnostic hook (not part of the declared ABI): default geometry of the persistent forward kernel
extern "C" void capdecdbg_head_fwd_geometry(int g) { g_head_fwd_geometry = g < 0 ? -1 : (g ? 1 : 0); }     // -1: never

extern "C" int capdec_logits_ce_fwd(const void* h, const void* W, const float* bias, int dtype,
                                    long M, int V, int d, long ldh, long ldw,
                                    const int64_t* labels, int B, int L,
                                    float* lse, int32_t* argmax, float* loss_sum, int32_t* correct,
                                    void* probs, long ld_probs, const capdec_kpanels* panels,
                                    void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(h && W && labels && lse && loss_sum && correct, "logits_ce_fwd: null pointer");
    CAPDEC_CHECK(panels_ok(panels, M, V) && (!panels || panels->probs_rows == 0), "logits_ce_fwd: k-panel images must be 16-byte aligned and hold at least M / V rows per panel (probs_rows: capdec_logits_ce_fwd_ref only)");
    CAPDEC_CHECK(M == (long)B * L && M < (1L << 31), "logits_ce_fwd: M (%ld) must equal B*L", M);
    CAPDEC_CHECK(!probs || (ld_probs >= V && (ld_probs % 8) == 0 && ld_probs <= (long)((V + BN - 1) / BN) * BN && aligned16(probs)),
                 "logits_ce_fwd: ld_probs (%ld) must be a multiple of 8 in [V, roundup128(V)] and probs 16-byte aligned", ld_probs);
    CeWs ws;
    if (!carve_ce_ws(workspace, workspace_bytes, M, V, ws)) {
        capdec_set_error("logits_ce_fwd: workspace too small (%ld < %ld)", workspace_bytes, capdec_logits_ce_workspace(M, V));
        return CAPDEC_ERR_WORKSPACE;
    }
    GemmArgs a;
    fill_head_args(a, h, W, bias, M, V, d, ldh, ldw);
    a.C = probs; a.ldc = ld_probs; a.store_cols = (int)ld_probs; a.out_dtype = dtype;      // NULL: nothing but the partials is stored
    if (probs && dtype == CAPDEC_BF16) a.fast_epi = 8;
    a.labels = labels; a.ce_B = B; a.ce_L = L;
    a.part4 = ws.part4; a.zlabel = ws.zlabel; a.np = ws.np;
    hipStream_t st = (hipStream_t)stream;
    LabelDot dot = {};
    HeadFwdArgs ha;
    const int geo = head_fwd_choice(a, dtype, ha);
    int rc;
    if (geo >= 0) {
        set_panels(ha, panels);
        rc = head_fwd_launch(ha, geo, st);
        dot = LabelDot{(const bf16_t*)h, (const bf16_t*)W, bias, ldh, ldw, d};     // that kernel leaves the label's logit to the finalize pass
    } else {
        rc = launch_gemm<EPI_CE_FWD>(a, dtype, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finalize_kernel<false>, dim3((unsigned)((M + FIN_ROWS - 1) / FIN_ROWS)), dim3(1024), 0, st,
                       ws.part4, ws.zlabel, labels, B, L, M, ws.np,
                       lse, argmax, (int64_t*)nullptr, (float*)nullptr, ws.rowloss, ws.rowcorrect, dot, (int32_t*)nullptr, geo >= 0 ? 1 : 0);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, ws.rowloss, ws.rowcorrect, M, loss_sum, correct);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_logits_ce_fwd_ref(const void* h, const void* W, const float* bias, int dtype,
                                        long M, int V, int d, long ldh, long ldw,
                                        const int64_t* labels, int B, int L,
                                        float* lse, int32_t* argmax, float* loss_sum, int32_t* correct,
                                        void* probs, long ld_probs, float* rowref, const capdec_kpanels* panels,
                                        void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(h && W && labels && lse && loss_sum && correct && probs && rowref, "logits_ce_fwd_ref: null pointer");
    CAPDEC_CHECK(panels_ok(panels, M, V), "logits_ce_fwd_ref: k-panel images must be 16-byte aligned and hold at least M / V rows per panel");
    CAPDEC_CHECK(dtype == CAPDEC_BF16, "logits_ce_fwd_ref: bf16 operands only");
    CAPDEC_CHECK(M == (long)B * L && M < (1L << 31), "logits_ce_fwd_ref: M (%ld) must equal B*L", M);
    CAPDEC_CHECK(ld_probs >= V && (ld_probs % 8) == 0 && ld_probs <= (long)((V + BN - 1) / BN) * BN && aligned16(probs),
                 "logits_ce_fwd_ref: ld_probs (%ld) must be a multiple of 8 in [V, roundup128(V)] and probs 16-byte aligned", ld_probs);
    CAPDEC_CHECK(aligned16(rowref), "logits_ce_fwd_ref: rowref must be 16-byte aligned");
    CeWs ws;
    if (!carve_ce_ws(workspace, workspace_bytes, M, V, ws)) {
        capdec_set_error("logits_ce_fwd_ref: workspace too small (%ld < %ld)", workspace_bytes, capdec_logits_ce_workspace(M, V));
        return CAPDEC_ERR_WORKSPACE;
    }
    if (M == 0) return CAPDEC_OK;
    HeadFwdArgs ha = {};
    ha.A = (const bf16_t*)h; ha.B = (const bf16_t*)W; ha.bias = bias; ha.C = (bf16_t*)probs;
    ha.lda = ldh; ha.ldb = ldw; ha.ldc = ld_probs; ha.M = (int)M; ha.N = V; ha.K = d; ha.store_cols = (int)ld_probs;
    ha.part4 = ws.part4; ha.np = ws.np; ha.mode = 3; ha.rowref = rowref;
    set_panels(ha, panels);
    const long prows = panels ? panels->probs_rows : 0;
    CAPDEC_CHECK(prows == 0 || (prows >= M && (ld_probs % 32) == 0), "logits_ce_fwd_ref: a column-panel image of probs needs probs_rows >= M and ld_probs %% 32 == 0");
    ha.c_panel_rows = prows;
    CAPDEC_CHECK(head_fwd_supported(ha), "logits_ce_fwd_ref: shape not supported (d %% 32 == 0, d >= 256, 16-byte aligned rows)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(label_logit_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const bf16_t*)h, ldh, (const bf16_t*)W, ldw, bias, d,
                       labels, B, L, M, rowref);
    const int rc = head_fwd_launch(ha, g_gemm_mode == 8 ? 0 : g_head_fwd_geometry == 0 ? 0 : 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finalize_kernel<true>, dim3((unsigned)((M + FIN_ROWS - 1) / FIN_ROWS)), dim3(1024), 0, st,
                       ws.part4, (const float*)rowref, labels, B, L, M, ws.np,
                       lse, argmax, (int64_t*)nullptr, (float*)nullptr, ws.rowloss, ws.rowcorrect, LabelDot{}, ws.rowflag, 1);
    hipLaunchKernelGGL(ce_ref_fixup_kernel, dim3((unsigned)((M + FIX_ROWS - 1) / FIX_ROWS)), dim3(256), 0, st, (const int32_t*)ws.rowflag, M, (const bf16_t*)h, ldh,
                       (const bf16_t*)W, ldw, bias, d, V, (bf16_t*)probs, ld_probs, prows, rowref, lse, ws.rowloss);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, ws.rowloss, ws.rowcorrect, M, loss_sum, correct);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_logits_ce_bwd_ref_prepare(void* probs, long ld_probs, long M, int V,
                                                const void* h, long ldh, int d,
                                                const int64_t* labels, int B, int L,
                                                const float* lse, const float* rowref, float gscale,
                                                void* h_scaled, long ld_hs, float* rowscale, void* rowscale_bf16,
                                                long probs_panel_rows, void* stream) {
    CAPDEC_CHECK(probs && h && labels && lse && rowref && h_scaled && rowscale, "logits_ce_bwd_ref_prepare: null pointer");
    CAPDEC_CHECK(M == (long)B * L && M < (1L << 31), "logits_ce_bwd_ref_prepare: M (%ld) must equal B*L", M);
    CAPDEC_CHECK(ld_probs >= V && (d % 8) == 0 && (ldh % 8) == 0 && (ld_hs % 8) == 0 && ld_hs >= d && aligned16(h) && aligned16(h_scaled),
                 "logits_ce_bwd_ref_prepare: rows of h / h_scaled must be 16-byte aligned, d a multiple of 8");
    if (M == 0) return CAPDEC_OK;
    CAPDEC_CHECK(probs_panel_rows == 0 || (probs_panel_rows >= M && (ld_probs % 32) == 0), "logits_ce_bwd_ref_prepare: probs_panel_rows must be 0 or >= M, with ld_probs %% 32 == 0");
    hipLaunchKernelGGL(ce_ref_bwd_prepare_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)probs, ld_probs, probs_panel_rows,
                       (const bf16_t*)h, ldh, d, labels, B, L, M, lse, rowref, gscale, (bf16_t*)h_scaled, ld_hs, rowscale, (bf16_t*)rowscale_bf16);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_logits_argmax(const void* h, const void* W, const float* bias, int dtype,
                                    long M, int V, int d, long ldh, long ldw,
                                    int64_t* out_ids, float* margin, const capdec_kpanels* panels,
                                    void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(h && W && out_ids, "logits_argmax: null pointer");
    CAPDEC_CHECK(panels_ok(panels, M, V) && (!panels || panels->probs_rows == 0), "logits_argmax: k-panel images must be 16-byte aligned and hold at least M / V rows per panel (probs_rows: capdec_logits_ce_fwd_ref only)");
    CAPDEC_CHECK(M < (1L << 31), "logits_argmax: M too large");
    CeWs ws;
    if (!carve_ce_ws(workspace, workspace_bytes, M, V, ws)) {
        capdec_set_error("logits_argmax: workspace too small (%ld < %ld)", workspace_bytes, capdec_logits_ce_workspace(M, V));
        return CAPDEC_ERR_WORKSPACE;
    }
    GemmArgs a;
    fill_head_args(a, h, W, bias, M, V, d, ldh, ldw);
    a.C = nullptr;
    a.ce_B = 1; a.ce_L = 1;
    a.part4 = ws.part4; a.zlabel = ws.zlabel; a.np = ws.np; a.want_sec = margin != nullptr;
    hipStream_t st = (hipStream_t)stream;
    HeadFwdArgs ha;
    const int geo = margin ? -1 : head_fwd_choice(a, dtype, ha);      // the second maximum (margins) is the generic epilogue's
    if (geo >= 0) set_panels(ha, panels);
    const int rc = geo >= 0 ? head_fwd_launch(ha, geo, st) : launch_gemm<EPI_CE_FWD>(a, dtype, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finalize_kernel<false>, dim3((unsigned)((M + FIN_ROWS - 1) / FIN_ROWS)), dim3(1024), 0, st,
                       ws.part4, ws.zlabel, (const int64_t*)nullptr, 1, 1, M, ws.np,
                       (float*)nullptr, (int32_t*)nullptr, out_ids, margin, (float*)nullptr, (int32_t*)nullptr, LabelDot{}, (int32_t*)nullptr, geo >= 0 ? 1 : 0);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_logits_ce_bwd_from_probs(void* probs_dl, int dtype, long M, int V, long ld,
                                               const int64_t* labels, int B, int L, const float* lse, float gscale,
                                               const void* fwd_workspace, long fwd_workspace_bytes,
                                               float* dbias, void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(probs_dl && labels && lse, "logits_ce_bwd_from_probs: null pointer");
    CAPDEC_CHECK(dtype == CAPDEC_BF16 || dtype == CAPDEC_F32, "logits_ce_bwd_from_probs: bad dtype");
    CAPDEC_CHECK(M == (long)B * L && M < (1L << 31), "logits_ce_bwd_from_probs: M (%ld) must equal B*L", M);
    CAPDEC_CHECK(ld >= V && (ld % 8) == 0 && ld <= (long)((V + BN - 1) / BN) * BN && aligned16(probs_dl),
                 "logits_ce_bwd_from_probs: ld (%ld) must be a multiple of 8 in [V, roundup128(V)]", ld);
    CeWs ws;
    if (!carve_ce_ws((void*)fwd_workspace, fwd_workspace_bytes, M, V, ws)) {
        capdec_set_error("logits_ce_bwd_from_probs: forward workspace too small (%ld < %ld)", fwd_workspace_bytes, capdec_logits_ce_workspace(M, V));
        return CAPDEC_ERR_WORKSPACE;
    }
    if (dbias && (!workspace || workspace_bytes < capdec_logits_ce_bwd_workspace(M, V))) {
        capdec_set_error("logits_ce_bwd_from_probs: workspace too small (%ld < %ld)", workspace_bytes, capdec_logits_ce_bwd_workspace(M, V));
        return CAPDEC_ERR_WORKSPACE;
    }
    const long ldp = (long)((V + BN - 1) / BN) * BN;
    dim3 grid((unsigned)((ld + 255) / 256), (unsigned)((M + 63) / 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
    float* part = dbias ? (float*)workspace : nullptr;
    if (dtype == CAPDEC_BF16)
        hipLaunchKernelGGL(ce_grad_from_probs_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)probs_dl, ld, M, ws.part4, ws.np, lse, labels, B, L, gscale, part, ldp);
    else
        hipLaunchKernelGGL(ce_grad_from_probs_kernel<float>, grid, block, 0, st, (float*)probs_dl, ld, M, ws.part4, ws.np, lse, labels, B, L, gscale, part, ldp);
    if (dbias)
        hipLaunchKernelGGL(colpart_reduce_kernel, dim3((V + 63) / 64), dim3(1024), 0, st, (const float*)part, (int)((M + 63) / 64), ldp, V, dbias, 0);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" long capdec_logits_ce_bwd_workspace(long M, int nv) {
    return ((M + 63) / 64) * (long)((nv + BN - 1) / BN * BN) * 4;
}

extern "C" int capdec_logits_ce_bwd_slab(const void* h, const void* W, const float* bias, int dtype,
                                         long M, int V, int d, long ldh, long ldw,
                                         const int64_t* labels, int B, int L, const float* lse, float gscale,
                                         int v0, int nv, void* dl, long ld_dl, float* dbias,
                                         void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(h && W && labels && lse && dl, "logits_ce_bwd_slab: null pointer");
    CAPDEC_CHECK(M == (long)B * L && M < (1L << 31), "logits_ce_bwd_slab: M (%ld) must equal B*L", M);
    CAPDEC_CHECK(v0 >= 0 && nv > 0 && v0 + nv <= V, "logits_ce_bwd_slab: slab [%d,%d) outside vocabulary %d", v0, v0 + nv, V);
    CAPDEC_CHECK(ld_dl >= nv && (ld_dl % 4) == 0 && ld_dl <= (long)((nv + BN - 1) / BN) * BN,
                 "logits_ce_bwd_slab: ld_dl (%ld) must be a multiple of 4 in [nv, roundup128(nv)]", ld_dl);
    if (dbias && (!workspace || workspace_bytes < capdec_logits_ce_bwd_workspace(M, nv))) {
        capdec_set_error("logits_ce_bwd_slab: workspace too small (%ld < %ld)", workspace_bytes, capdec_logits_ce_bwd_workspace(M, nv));
        return CAPDEC_ERR_WORKSPACE;
    }
    GemmArgs a;
    const size_t esz = dtype == CAPDEC_BF16 ? 2 : 4;
    fill_head_args(a, h, (const char*)W + (size_t)v0 * ldw * esz, bias ? bias + v0 : nullptr, M, nv, d, ldh, ldw);
    a.C = dl; a.ldc = ld_dl; a.out_dtype = dtype; a.store_cols = (int)ld_dl;
    if (dtype == CAPDEC_BF16 && (ld_dl % 8) == 0 && aligned16(dl)) a.fast_epi = 8;      // 8 columns = one 16-byte store per lane
    a.labels = labels; a.ce_B = B; a.ce_L = L; a.ce_col0 = v0;
    a.lse = lse; a.gscale = gscale;
    const long ldp = (long)((nv + BN - 1) / BN) * BN;
    if (dbias) { a.colpart = (float*)workspace; a.ld_colpart = ldp; }
    int rc = launch_gemm<EPI_CE_BWD>(a, dtype, (hipStream_t)stream);
    if (rc || !dbias) return rc;
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3((nv + 63) / 64), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)workspace, (int)((M + 63) / 64), ldp, nv, dbias + v0, 0);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
