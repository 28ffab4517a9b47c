// Self-attention for the captioning decoder (gfx950).
//
// Training / prefill (bf16): one workgroup per (sequence, head, slab of up to 64
// queries); each wave owns 16 queries and walks the keys in blocks of 64 with an
// online softmax.  Scores are computed transposed (S^T = K Q^T) so that a lane
// holds 4 consecutive keys of ONE query: the probabilities are then already the
// B operand of the next MFMA (O^T = V^T P^T).  Only the operand that is consumed
// TRANSPOSED sits in LDS (the sequence's whole V, [S,hd], S <= 768 at hd 96, read
// back with the transposing LDS read); K is a row-major operand, whose fragments
// are 16 contiguous bytes per lane and come straight from global memory, one key
// block ahead of their use.  The backward uses the same trick twice: a
// query-major kernel for dQ (K in LDS, V direct; it also produces delta = rowsum(dO*O))
// and a key-major kernel for dK/dV, both recomputing P from the saved log-sum-exp; for at most 32 queries (the scored rows
// of the one-block model) one kernel forms all three, so that K and V of a head cross the memory system once.
// Validation (f32): simple exact-f32 kernels with the same interface.
// Decoding: one query row against a contiguous K/V cache, HBM-bound.
#include "common.h"

namespace {

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, v_bs, v_hs, v_ss, o_bs, o_hs, o_ss;
    int B, H, S, hd, q_start, nq, mask_mode, open_cols;
    int S4;                // S rounded up to a multiple of 4: row stride of the dropout index space
    uint32_t drop_thr, drop_k0, drop_k1; float drop_scale;
    float* lse;
    // backward
    const void* dout; long do_bs, do_hs, do_ss;
    void* dq; long dq_bs, dq_hs, dq_ss;
    void* dk; long dk_bs, dk_hs, dk_ss;
    void* dv; long dv_bs, dv_hs, dv_ss;
    float dq_scale; float* delta;
    // optional: column sums of dq and of dk | dv per 16-row tile t of a (sequence, head):
    // colsum_q[(b * ntq + t) * H*hd + h*hd + c], colsum_kv[(b * ntk + t) * 2*H*hd + {0, H*hd} + h*hd + c]
    float* colsum_q; float* colsum_kv; int ntq, ntk;
};

__device__ __forceinline__ bool key_visible(int q, int key, int S, int mask_mode, int open_cols) {
    return key < S && (mask_mode == CAPDEC_MASK_NONE || key <= q || key < open_cols);
}

template <int HD> struct Geo {
    static constexpr int ROWB = HD * 2 + 16;   // LDS bytes per row (16-B pad)
    static constexpr int KS = HD / 32;         // 32-deep k-steps over the head dim
    static constexpr int DT = HD / 16;         // 16-wide tiles over the head dim
    static constexpr int CPR = HD / 8;         // 16-B chunks per row
};

// rows [row0, row0+nvalid) of a strided [*, HD] bf16 matrix -> LDS, zero rows up to npad;
// four 16-byte loads per thread are in flight before the first LDS write
template <int HD, int U = 4>
__device__ __forceinline__ void rows_to_lds(char* lds, const bf16_t* base, long sstride, int row0, int nvalid, int npad,
                                            int tid, int nthreads) {
    constexpr int CPR = Geo<HD>::CPR, ROWB = Geo<HD>::ROWB;
    const int total = npad * CPR;
    for (int c0 = tid; c0 < total; c0 += U * nthreads) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int row = c / CPR, ch = c - row * CPR;
            v[u] = u32x4{0, 0, 0, 0};
            if (c < total && row < nvalid) v[u] = *reinterpret_cast<const u32x4*>(base + (long)(row0 + row) * sstride + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int row = c / CPR, ch = c - row * CPR;
            if (c < total) *reinterpret_cast<u32x4*>(lds + row * ROWB + ch * 16) = v[u];
        }
    }
}

// two matrices of the same shape at once (the whole-sequence kernels): all loads of both are in flight before the
// first LDS write, so a workgroup that has the compute unit to itself pays one memory latency per batch, not two
template <int HD, int U>
__device__ __forceinline__ void rows_to_lds2(char* lds0, const bf16_t* base0, long ss0, char* lds1, const bf16_t* base1, long ss1,
                                             int row0, int nvalid, int npad, int tid, int nthreads) {
    constexpr int CPR = Geo<HD>::CPR, ROWB = Geo<HD>::ROWB;
    const int total = npad * CPR;
    for (int c0 = tid; c0 < total; c0 += U * nthreads) {
        u32x4 v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int row = c / CPR, ch = c - row * CPR;
            v0[u] = v1[u] = u32x4{0, 0, 0, 0};
            if (c < total && row < nvalid) {
                v0[u] = *reinterpret_cast<const u32x4*>(base0 + (long)(row0 + row) * ss0 + ch * 8);
                v1[u] = *reinterpret_cast<const u32x4*>(base1 + (long)(row0 + row) * ss1 + ch * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            const int row = c / CPR, ch = c - row * CPR;
            if (c < total) {
                *reinterpret_cast<u32x4*>(lds0 + row * ROWB + ch * 16) = v0[u];
                *reinterpret_cast<u32x4*>(lds1 + row * ROWB + ch * 16) = v1[u];
            }
        }
    }
}

// The same two matrices held in registers between their fetch and their LDS image: the whole-sequence kernels are
// persistent, and fetch the next (sequence, head)'s operands while they compute on the current one's -- with one
// workgroup per CU (both operands of a 300-row sequence fill most of the LDS) nothing else would cover that fetch.
// A thread owns one 16-byte chunk position of the rows r, r + RP, r + 2 RP, ... (RP = rows that 512 threads cover at
// once), so its NU loads and stores differ by constant offsets: no address register per load.  Rows from NU * RP on
// (longer sequences) are fetched when they are stored.
template <int HD, int NU>
struct HeadPair {
    static constexpr int CPR = Geo<HD>::CPR, ROWB = Geo<HD>::ROWB;
    static constexpr int RP = 512 / CPR;
    u32x4 a[NU], b[NU];
    __device__ __forceinline__ void fetch(const bf16_t* base0, long ss0, const bf16_t* base1, long ss1, int row0, int nvalid, int tid) {
        const int r = tid / CPR, ch = tid - r * CPR;
        const bf16_t* p0 = base0 + (long)(row0 + r) * ss0 + ch * 8;
        const bf16_t* p1 = base1 + (long)(row0 + r) * ss1 + ch * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            a[u] = b[u] = u32x4{0, 0, 0, 0};
            if (r < RP && r + u * RP < nvalid) {
                a[u] = *reinterpret_cast<const u32x4*>(p0 + (long)(u * RP) * ss0);
                b[u] = *reinterpret_cast<const u32x4*>(p1 + (long)(u * RP) * ss1);
            }
        }
    }
    __device__ __forceinline__ void store(char* lds0, const bf16_t* base0, long ss0, char* lds1, const bf16_t* base1, long ss1,
                                          int row0, int nvalid, int npad, int tid, int nthreads) {
        const int r = tid / CPR, ch = tid - r * CPR;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (r < RP && r + u * RP < npad) {
                *reinterpret_cast<u32x4*>(lds0 + (r + u * RP) * ROWB + ch * 16) = a[u];
                *reinterpret_cast<u32x4*>(lds1 + (r + u * RP) * ROWB + ch * 16) = b[u];
            }
        }
        for (int c = NU * RP * CPR + tid; c < npad * CPR; c += nthreads) {       // rows past the prefetched ones
            const int row = c / CPR, cc = c - row * CPR;
            u32x4 x = u32x4{0, 0, 0, 0}, y = u32x4{0, 0, 0, 0};
            if (row < nvalid) {
                x = *reinterpret_cast<const u32x4*>(base0 + (long)(row0 + row) * ss0 + cc * 8);
                y = *reinterpret_cast<const u32x4*>(base1 + (long)(row0 + row) * ss1 + cc * 8);
            }
            *reinterpret_cast<u32x4*>(lds0 + row * ROWB + cc * 16) = x;
            *reinterpret_cast<u32x4*>(lds1 + row * ROWB + cc * 16) = y;
        }
    }
};
constexpr int SEQ_NU = 8;      // at 96 columns: 8 x 42 = 336 rows per matrix

// the 4 x KS row-major fragments of a 64-row block of a [S, HD] matrix, straight from global memory:
// lane (g,i) of tile t gets M[row0 + 16t + i][32ks + 8g .. +7] (zero past S)
template <int HD>
__device__ __forceinline__ void frags_from_global(const bf16_t* base, long sstride, int row0, int S, int lane,
                                                  bf16x8 (&f)[4][Geo<HD>::KS]) {
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = row0 + t * 16 + i;
#pragma unroll
        for (int ks = 0; ks < Geo<HD>::KS; ++ks) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
            f[t][ks] = row < S ? *reinterpret_cast<const bf16x8*>(base + (long)row * sstride + ks * 32 + 8 * g) : z;
        }
    }
}

// row-major MFMA operand: lane (g,i) gets M[row0+i][k0 + 8g .. +7]
// First statement of a block that only some key / query blocks need (masking): keeps the block a branch.  Left to its
// cost model the compiler turns such a block into selects that every pass of the loop executes -- a quarter of the vector
// instructions of the attention loops, for a mask that four blocks in five do not need.
__device__ __forceinline__ void masked_path() { asm volatile(""); }

template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * Geo<HD>::ROWB + (k0 + 8 * (lane >> 4)) * 2);
}
// transposed MFMA operand from a [row][col] LDS image: lane (g,i) gets
//   { M[rA + 4g + 0..3][col0 + i], M[rB + 4g + 0..3][col0 + i] }
template <int HD>
__device__ __forceinline__ bf16x8 frag_cols(const char* lds, int rA, int rB, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int off = (4 * g + (i >> 2)) * Geo<HD>::ROWB + (col0 + 4 * (i & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, lds + rA * Geo<HD>::ROWB + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, lds + rB * Geo<HD>::ROWB + off));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// one LDS-DMA instruction: lane l's 16 bytes at base + off land at dst + 16 l (dst wave-uniform); bytes from `rem` on read as zero
__device__ __forceinline__ void lds_dma_16(const bf16_t* base, long rem, unsigned off, char* dst) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rem > 0 ? (int)rem : 0, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, off, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 zero_frag() {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)0.f;
    return r;
}
// v combined with the values the lanes 16, 32 and 48 further on (mod 64) hold.  The whole-sequence forward, whose
// waves are bound by their own instruction stream, takes the register-swap form (no LDS round trip); the slab
// kernels and the backward, which wait on memory, keep the shuffles they were measured with (no gain from the swaps).
typedef __attribute__((ext_vector_type(2))) unsigned int u32pair;
template <bool SWAP>
__device__ __forceinline__ float quad_max(float v) {
    if constexpr (SWAP) {
        u32pair r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        return fmaxf(v, __shfl_xor(v, 32, 64));
    }
}
template <bool SWAP>
__device__ __forceinline__ float quad_sum(float v) {
    if constexpr (SWAP) {
        u32pair r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    }
}

// two floats -> one dword of two bf16 (a single v_cvt_pk_bf16_f32; element-wise conversions cost two and a permute)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16pair;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16pair t;
    t[0] = (bf16_t)lo; t[1] = (bf16_t)hi;
    return __builtin_bit_cast(uint32_t, t);
}
// Column sums over the 16 rows of a wave's accumulator tiles: lane (g, i) holds columns 16 dt + 4g .. + 3 of row i; after
// the fold every lane holds the sums over i.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int DT>
__device__ __forceinline__ void fold_rows16(f32x4 (&a)[DT]) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // data-parallel-primitive adds inside the row of 16 lanes: neighbours, quad halves, mirrored halves of 8, mirrored row
            float v = a[dt][r];
            v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
            a[dt][r] = v;
        }
}
// ... and leave as one row of partial sums: dst[16 dt + 4g .. + 3] = scale * sums (lanes with i == 0 write)
template <int DT>
__device__ __forceinline__ void write_colsums(const f32x4 (&a)[DT], float scale, int g, int i, float* dst) {
    if (i == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16 + 4 * g) = a[dt] * scale;
    }
}

// x, y: a lane's 4 columns (4g .. 4g+3) of two adjacent 16-column accumulator tiles, scaled and rounded to bf16; returns 8
// consecutive columns of one of the tiles: lanes with even g get columns 4g .. 4g+7 of x's tile, odd g columns
// 4(g-1) .. 4(g-1)+7 of y's tile.  Every lane of the wave must execute it.
__device__ __forceinline__ u32x4 swap_pair(const f32x4& x, const f32x4& y, float scale) {
    const uint32_t x0 = pack_bf16(x[0] * scale, x[1] * scale), x1 = pack_bf16(x[2] * scale, x[3] * scale);
    const uint32_t y0 = pack_bf16(y[0] * scale, y[1] * scale), y1 = pack_bf16(y[2] * scale, y[3] * scale);
    const u32pair s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    const u32pair s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

// the value quad lane R holds, in all four lanes of every aligned quad
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true); }

constexpr float ATT_LOG2E = 1.4426950408889634f;
// keys [0, key_limit) are the visible ones for query q: a prefix, in every mask mode
__device__ __forceinline__ int key_limit(int q, int S, int mask_mode, int open_cols) {
    return mask_mode == CAPDEC_MASK_NONE ? S : min(S, max(q + 1, open_cols));
}

// ================================================================== forward (bf16 MFMA)
// SEQ = false: a workgroup owns up to 64 queries of one (sequence, head); K fragments come from global memory.
// SEQ = true (long query ranges): ONE workgroup of 8 waves owns the whole (sequence, head); K and V are both
// staged in LDS once and every wave walks the query tiles wave, wave + 8, ...  K, V, Q and O then cross the
// memory system once per head instead of once per 64-query slab (V) and once per wave (K).
// Workgroup -> (sequence, head): consecutive workgroup ids go to the 8 XCDs in turn, each with an L2 of its own, and a
// head's rows are 2*hd-byte pieces of rows that hold all heads (the packed QKV layout) -- pieces that share 128-byte
// lines with the neighbouring heads'.  Each XCD therefore gets a contiguous run of (sequence, head) pairs: the heads
// of a sequence are fetched through one L2 at about the same time, and a line is read from memory once.
__device__ __forceinline__ int xcd_logical(int bid, int n) {
    const int qq = n >> 3, r = n & 7, x = bid & 7;
    return (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + (bid >> 3);
}

// BLK = 1 or 2 (waves per workgroup; at most 32 queries, SEQ = false): V is not staged whole but in 64-key blocks, double
// buffered by LDS-DMA: the next block is on its way while the current one is multiplied.  27 KB of LDS instead of 53
// (S = 229): six workgroups per CU instead of three, each with a K and a V block in flight -- the kernel is bound by
// the bytes it keeps in flight, not by its arithmetic.
template <int HD, bool SEQ, int BLK = 0>
__global__ __launch_bounds__(SEQ ? 512 : (BLK ? 64 * BLK : 256), SEQ ? 1 : 3) void attn_fwd_mfma(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<HD>;
    static_assert(!(SEQ && BLK), "block staging belongs to the slab form");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int SP = (p.S + 63) & ~63;
    const int nthreads = blockDim.x;
    char* Vl = smem;
    char* Kl = smem + SP * G::ROWB;                                   // SEQ only
    const int nheads = p.B * p.H;
    HeadPair<HD, SEQ ? SEQ_NU : 1> pre;
    auto fetch_head = [&](int hh) {
        const int fb = hh / p.H, fh = hh - fb * p.H;
        pre.fetch((const bf16_t*)p.v + fb * p.v_bs + fh * p.v_hs, p.v_ss, (const bf16_t*)p.k + fb * p.k_bs + fh * p.k_hs, p.k_ss, 0, p.S, tid);
    };
    const int bh0 = xcd_logical(blockIdx.x, gridDim.x);
    if constexpr (SEQ) fetch_head(bh0);
    // SEQ: persistent, the workgroup walks (sequence, head) pairs bh0, + gridDim.x, ...
    for (int bh = bh0;; bh += gridDim.x) {
    const int b = bh / p.H, h = bh - b * p.H;
    const bf16_t* kbase = (const bf16_t*)p.k + b * p.k_bs + h * p.k_hs;
    bf16x8 kf[4][G::KS];
    if constexpr (!SEQ) frags_from_global<HD>(kbase, p.k_ss, 0, p.S, lane, kf);       // first key block, in flight during the staging of V
    const bf16_t* vbase = (const bf16_t*)p.v + b * p.v_bs + h * p.v_hs;
    // BLK: a 64-key block of V goes global -> LDS by LDS-DMA (no staging registers): 1-KiB pieces, lane l's 16 bytes land at
    // piece base + 16 l, so a lane fetches the chunk that belongs at that position of the padded [64][ROWB] image (the
    // pad chunk of a row fetches nothing); wave w issues pieces w, w + BLK, ...
    constexpr int VPIECES = 64 * G::ROWB / 1024, VPW = BLK ? (VPIECES + BLK - 1) / BLK : 1;
    static_assert(!BLK || (64 * G::ROWB) % 1024 == 0, "a V block is whole 1-KiB pieces");
    unsigned voff[VPW];
    if constexpr (BLK != 0) {
#pragma unroll
        for (int u = 0; u < VPW; ++u) {
            const int pos = ((wave + u * BLK) * 64 + lane), row = pos / (G::CPR + 1), ch = pos - row * (G::CPR + 1);
            voff[u] = ch < G::CPR ? (unsigned)(((long)row * p.v_ss + ch * 8) * 2) : 0xFFFFFFFFu;
        }
    }
    auto dma_vblock = [&](int kb, char* buf) {
        // rows from S on lie past the descriptor's range and arrive as zeros
        const long rem = ((long)(p.S - kb - 1) * p.v_ss + HD) * 2;
#pragma unroll
        for (int u = 0; u < VPW; ++u) {
            const int piece = wave + u * BLK;
            if (piece < VPIECES) lds_dma_16(vbase + (long)kb * p.v_ss, rem, voff[u], buf + piece * 1024);
        }
    };
    if constexpr (SEQ) {
        __syncthreads();                                              // every wave is done with the previous head's images
        pre.store(Vl, vbase, p.v_ss, Kl, kbase, p.k_ss, 0, p.S, SP, tid, nthreads);
    } else if constexpr (BLK != 0) {
        dma_vblock(0, Vl);
    } else {
        rows_to_lds<HD>(Vl, vbase, p.v_ss, 0, p.S, SP, tid, nthreads);
    }
    if constexpr (BLK == 0) __syncthreads();
    bool fetched = !SEQ || bh + (int)gridDim.x >= nheads;            // the next head's operands are on their way (or there is none)
    const int qend = min(p.q_start + p.nq, p.S);
    const int qstep = SEQ ? 16 * (nthreads >> 6) : (1 << 30);         // SEQ: the wave's next query tile
    for (int qw = p.q_start + (SEQ ? 0 : blockIdx.y * (nthreads >> 2)) + wave * 16; qw < qend; qw += qstep) {
    const int q = qw + i;
    const bool qok = q < qend;
    const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + h * p.q_hs + (long)q * p.q_ss;
    bf16x8 qf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) qf[ks] = qok ? *reinterpret_cast<const bf16x8*>(qrow + ks * 32 + 8 * g) : zero_frag();
    f32x4 oacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t idx_row = (uint32_t)((bh * p.S + q) * (long)p.S4);       // a multiple of 4, as are the key bases below
    const int klim = key_limit(q, p.S, p.mask_mode, p.open_cols);           // this lane's query sees keys < klim
    const int klim_wave = key_limit(qw, p.S, p.mask_mode, p.open_cols);     // the smallest limit in the wave (its first query)

    if constexpr (SEQ) {
        // the query rows are in before the key loop starts: inside it nothing then waits on the memory pipeline, which
        // the next head's operands enter during this wave's last tile
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, qf[ks])));
        // SEQ keeps a key block's K fragments in registers too, read from the LDS image one block ahead
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) kf[t][ks] = frag_rows<HD>(Kl, t * 16, ks * 32, lane);
    }
    for (int kb = 0; kb < SP; kb += 64) {
        const char* Vb = BLK ? Vl + ((kb >> 6) & 1) * 64 * G::ROWB : Vl + kb * G::ROWB;
        if constexpr (BLK != 0) {
            // this block's V (and K fragments) have landed; after the barrier every wave's pieces have, and every wave is done
            // with the other buffer, which the next block's pieces may now overwrite
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kb + 64 < SP) dma_vblock(kb + 64, Vl + (((kb >> 6) + 1) & 1) * 64 * G::ROWB);
        }
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[ks], s[t], 0, 0, 0);
        }
        // SEQ: V's fragments for the block's first 32 keys are requested now and used after the softmax arithmetic
        bf16x8 vfr[SEQ ? G::DT : 1];
        if constexpr (SEQ) {
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) vfr[dt] = frag_cols<HD>(Vl, kb, kb + 16, dt * 16, lane);
        }
        if constexpr (!SEQ) {
            // the next key block's fragments replace the ones just consumed; the softmax and the P V products
            // below cover their latency
            if (kb + 64 < SP) frags_from_global<HD>(kbase, p.k_ss, kb + 64, p.S, lane, kf);
        } else {
            if (kb + 64 < SP) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) kf[t][ks] = frag_rows<HD>(Kl, kb + 64 + t * 16, ks * 32, lane);
            }
            // the wave's last tile of this head, its query rows consumed: the next head's operands follow them into
            // the memory pipeline (loads return in order: issued any earlier, the query rows would wait for these)
            if (!fetched && kb == 0 && qw + qstep >= qend) { fetch_head(bh + gridDim.x); fetched = true; }
        }
        // masking only in key blocks that some query of the wave does not see entirely (a wave-uniform test)
        if (kb + 64 > klim_wave) {
            masked_path();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = kb + t * 16 + 4 * g + r < klim ? s[t][r] : -INFINITY;
        }
        float bm = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int t = 1; t < 4; ++t) bm = fmaxf(bm, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        bm = quad_max<SEQ>(bm);
        const float m_new = fmaxf(m_run, bm);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * ATT_LOG2E);
        const float nm2 = -m_use * ATT_LOG2E;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) oacc[dt] *= alpha;
        m_run = m_new;
        u32x4 pfu[2];
        bf16x8 pfe[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bool keep[4] = {true, true, true, true};
            if (p.drop_thr) dropout_keep4(p.drop_k0, p.drop_k1, idx_row + kb + t * 16 + 4 * g, p.drop_thr, keep);     // the lane's 4 keys: one hash
            float pk[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[t][r], ATT_LOG2E, nm2));
                l_run += pv;
                pk[r] = keep[r] ? pv : 0.f;          // the 1/(1-p) of the dropout is applied once, to the finished row
            }
            if constexpr (SEQ) {
                pfu[t >> 1][(t & 1) * 2 + 0] = pack_bf16(pk[0], pk[1]);
                pfu[t >> 1][(t & 1) * 2 + 1] = pack_bf16(pk[2], pk[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) pfe[t >> 1][(t & 1) * 4 + r] = (bf16_t)pk[r];
            }
        }
        bf16x8 pf[2];
        if constexpr (SEQ) { pf[0] = __builtin_bit_cast(bf16x8, pfu[0]); pf[1] = __builtin_bit_cast(bf16x8, pfu[1]); }
        else { pf[0] = pfe[0]; pf[1] = pfe[1]; }
        if constexpr (SEQ) {
            // those of the other 32 keys are requested before the first half's products, which cover their latency
            bf16x8 vfr2[G::DT];
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) vfr2[dt] = frag_cols<HD>(Vl, kb + 32, kb + 48, dt * 16, lane);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf[0], oacc[dt], 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr2[dt], pf[1], oacc[dt], 0, 0, 0);
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        frag_cols<HD>(Vb, (2 * st) * 16, (2 * st + 1) * 16, dt * 16, lane), pf[st], oacc[dt], 0, 0, 0);
        }
    }
    const float l = quad_sum<SEQ>(l_run);
    const float inv = l > 0.f ? p.drop_scale / l : 0.f;
    if (qok) {
        bf16_t* orow = (bf16_t*)p.o + b * p.o_bs + h * p.o_hs + (long)q * p.o_ss;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(oacc[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + dt * 16 + 4 * g) = o;
        }
        if (g == 0 && p.lse) p.lse[(long)bh * p.S + q] = m_run + __logf(l);
    }
    }
    if constexpr (SEQ) {
        if (!fetched) fetch_head(bh + gridDim.x);                     // a wave without a query tile
    }
    if (!SEQ || bh + (int)gridDim.x >= nheads) break;
    }
}

// ================================================================== backward: dQ (query-major)
// SEQ as in the forward: one workgroup of 8 waves per (sequence, head), K and V both in LDS.
// BLK = 1 or 2 (waves per workgroup; at most 32 queries, SEQ = false): K staged in 64-key blocks by LDS-DMA, double
// buffered, as V is in the forward.  The next block is requested after the current block's last LDS read (the compiler
// drains outstanding LDS-DMA before any LDS read it cannot tell apart from the DMA's target).
// LEAN (SEQ only, head width up to 96): sixteen waves per workgroup and no register look-ahead -- the fragments are read from
// LDS where they are used, which leaves 128 registers per wave, so four waves per SIMD cover each other's LDS and exp
// latencies and the head's 19 query tiles take two passes instead of three (dQ at S = 290: 250 -> 222 us; twelve waves
// measured level with eight; the same squeeze on the dK/dV kernel spills inside its loop and loses 40 us).
template <int HD, bool SEQ, int BLK = 0, bool LEAN = false>
__global__ __launch_bounds__(SEQ ? (LEAN ? 1024 : 512) : (BLK ? 64 * BLK : 256), SEQ ? (LEAN ? 4 : 1) : (BLK ? 2 : 3)) void attn_bwd_dq_mfma(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<HD>;
    static_assert(!(SEQ && BLK), "block staging belongs to the slab form");
    const int bh = xcd_logical(blockIdx.x, gridDim.x), b = bh / p.H, h = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int SP = (p.S + 63) & ~63;
    const int nthreads = blockDim.x;
    char* Kl = smem;
    char* Vl = smem + SP * G::ROWB;                                   // SEQ only
    const bf16_t* vbase = (const bf16_t*)p.v + b * p.v_bs + h * p.v_hs;
    bf16x8 vf[4][G::KS];
    if constexpr (!SEQ) frags_from_global<HD>(vbase, p.v_ss, 0, p.S, lane, vf);       // V is only a row-major operand here: no LDS copy
    const bf16_t* kbase = (const bf16_t*)p.k + b * p.k_bs + h * p.k_hs;
    constexpr int KPIECES = 64 * G::ROWB / 1024, KPW = BLK ? (KPIECES + BLK - 1) / BLK : 1;
    unsigned koff[KPW];
    if constexpr (BLK != 0) {
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const int pos = ((wave + u * BLK) * 64 + lane), row = pos / (G::CPR + 1), ch = pos - row * (G::CPR + 1);
            koff[u] = ch < G::CPR ? (unsigned)(((long)row * p.k_ss + ch * 8) * 2) : 0xFFFFFFFFu;
        }
    }
    auto dma_kblock = [&](int kb, char* buf) {
        const long rem = ((long)(p.S - kb - 1) * p.k_ss + HD) * 2;      // rows from S on arrive as zeros
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            const int piece = wave + u * BLK;
            if (piece < KPIECES) lds_dma_16(kbase + (long)kb * p.k_ss, rem, koff[u], buf + piece * 1024);
        }
    };
    if constexpr (SEQ) rows_to_lds2<HD, 8>(Kl, kbase, p.k_ss, Vl, vbase, p.v_ss, 0, p.S, SP, tid, nthreads);
    else if constexpr (BLK != 0) dma_kblock(0, Kl);
    else rows_to_lds<HD>(Kl, kbase, p.k_ss, 0, p.S, SP, tid, nthreads);
    if constexpr (BLK == 0) __syncthreads();
    const int qend = min(p.q_start + p.nq, p.S);
    const int qstep = SEQ ? 16 * (nthreads >> 6) : (1 << 30);
    for (int qw = p.q_start + (SEQ ? 0 : blockIdx.y * (nthreads >> 2)) + wave * 16; qw < qend; qw += qstep) {
    const int q = qw + i;
    const bool qok = q < qend;
    const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + h * p.q_hs + (long)q * p.q_ss;
    const bf16_t* dorow = (const bf16_t*)p.dout + b * p.do_bs + h * p.do_hs + (long)q * p.do_ss;
    const bf16_t* orow = (const bf16_t*)p.o + b * p.o_bs + h * p.o_hs + (long)q * p.o_ss;
    bf16x8 qf[G::KS], dof[G::KS];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        qf[ks] = qok ? *reinterpret_cast<const bf16x8*>(qrow + ks * 32 + 8 * g) : zero_frag();
        dof[ks] = qok ? *reinterpret_cast<const bf16x8*>(dorow + ks * 32 + 8 * g) : zero_frag();
        const bf16x8 of = qok ? *reinterpret_cast<const bf16x8*>(orow + ks * 32 + 8 * g) : zero_frag();
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)dof[ks][e] * (float)of[e];
    }
    const float nlse2 = qok ? -p.lse[(long)bh * p.S + q] * ATT_LOG2E : 0.f;
    const int klim = qok ? key_limit(q, p.S, p.mask_mode, p.open_cols) : 0;
    const int klim_wave = qw + 16 <= qend ? key_limit(qw, p.S, p.mask_mode, p.open_cols) : 0;     // 0: a partial tile always masks
    // delta = rowsum(dO * O): the four lanes (g) of a query hold disjoint quarters of its row
    const float delta = quad_sum<false>(dsum);
    if (qok && g == 0) p.delta[(long)bh * p.S + q] = delta;       // the key-major kernel reads it
    const uint32_t idx_row = (uint32_t)((bh * p.S + q) * (long)p.S4);
    f32x4 acc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) acc[dt] = f32x4{0, 0, 0, 0};

    // SEQ: the K and V fragments of a 16-key tile are read from the LDS images one tile ahead of their products (two
    // register sets), K's transposed fragments for the dQ products during the block's second half: left to the compiler,
    // every LDS read sits directly in front of its MFMA and is waited for there
    bf16x8 kfr[2][G::KS], vfr[2][G::KS];
    auto read_rows = [&](int key0, int set) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kfr[set][ks] = frag_rows<HD>(Kl, key0, ks * 32, lane);
            vfr[set][ks] = frag_rows<HD>(Vl, key0, ks * 32, lane);
        }
    };
    if constexpr (SEQ && !LEAN) read_rows(0, 0);
    for (int kb = 0; kb < SP; kb += 64) {
        const char* Kb = BLK ? Kl + ((kb >> 6) & 1) * 64 * G::ROWB : Kl + kb * G::ROWB;
        if constexpr (BLK != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this block's K pieces (and V fragments) have landed
            __syncthreads();                                          // ... every wave's, and every wave is done with the other buffer
        }
        u32x4 dsu[2];
        bf16x8 kcf[SEQ ? 2 : 1][SEQ ? G::DT : 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 s = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
            if constexpr (SEQ && !LEAN) {
                if (t < 3) read_rows(kb + (t + 1) * 16, (t + 1) & 1);
                else if (kb + 64 < SP) read_rows(kb + 64, 0);
                if (t == 2) {
#pragma unroll
                    for (int dt = 0; dt < G::DT; ++dt) kcf[0][dt] = frag_cols<HD>(Kl, kb, kb + 16, dt * 16, lane);
                }
                if (t == 3) {
#pragma unroll
                    for (int dt = 0; dt < G::DT; ++dt) kcf[1][dt] = frag_cols<HD>(Kl, kb + 32, kb + 48, dt * 16, lane);
                }
            }
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                if constexpr (SEQ && !LEAN) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[t & 1][ks], qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[t & 1][ks], dof[ks], dp, 0, 0, 0);
                } else if constexpr (SEQ) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Kb, t * 16, ks * 32, lane), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Vl, kb + t * 16, ks * 32, lane), dof[ks], dp, 0, 0, 0);
                } else {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Kb, t * 16, ks * 32, lane), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][ks], dof[ks], dp, 0, 0, 0);
                }
            }
            bool keep[4] = {true, true, true, true};
            if (p.drop_thr) dropout_keep4(p.drop_k0, p.drop_k1, idx_row + kb + t * 16 + 4 * g, p.drop_thr, keep);
            const bool masked = kb + 64 > klim_wave;          // wave-uniform: some query of the wave does not see the whole block
            float dsv[4], pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], ATT_LOG2E, nlse2));
            if (masked) {
                masked_path();
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = kb + t * 16 + 4 * g + r < klim ? pv[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dsv[r] = pv[r] * (fmaf(keep[r] ? dp[r] : 0.f, p.drop_scale, -delta));
            dsu[t >> 1][(t & 1) * 2 + 0] = pack_bf16(dsv[0], dsv[1]);
            dsu[t >> 1][(t & 1) * 2 + 1] = pack_bf16(dsv[2], dsv[3]);
        }
        const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, dsu[0]), __builtin_bit_cast(bf16x8, dsu[1])};
        if constexpr (!SEQ) {
            // the next key block of V replaces the fragments just consumed; the dS K products below cover the latency
            if (kb + 64 < SP) frags_from_global<HD>(vbase, p.v_ss, kb + 64, p.S, lane, vf);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                if constexpr (SEQ && !LEAN) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcf[st][dt], dsf[st], acc[dt], 0, 0, 0);
                else acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        frag_cols<HD>(Kb, (2 * st) * 16, (2 * st + 1) * 16, dt * 16, lane), dsf[st], acc[dt], 0, 0, 0);
            }
        if constexpr (BLK != 0) {
            if (kb + 64 < SP) dma_kblock(kb + 64, Kl + (((kb >> 6) + 1) & 1) * 64 * G::ROWB);
        }
    }
    if (qok) {
        bf16_t* dqrow = (bf16_t*)p.dq + b * p.dq_bs + h * p.dq_hs + (long)q * p.dq_ss;
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[dt][r] * p.dq_scale);
            *reinterpret_cast<bf16x4*>(dqrow + dt * 16 + 4 * g) = o;
        }
    }
    if (p.colsum_q) {                                                // the tile's column sums (rows past the range hold zeros)
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) acc[dt] = qok ? acc[dt] : f32x4{0, 0, 0, 0};
        fold_rows16<G::DT>(acc);
        write_colsums<G::DT>(acc, p.dq_scale, g, i, p.colsum_q + ((long)b * p.ntq + ((qw - p.q_start) >> 4)) * ((long)p.H * HD) + h * HD);
    }
    }
}

// ================================================================== backward: dK, dV (key-major)
// SEQ = true: one workgroup per (sequence, head) walks ALL key tiles (wave, wave + nwaves, ...), so Q and dO are staged
// once per head instead of once per 64-key slab.  Twelve waves where the registers allow three per SIMD (head width up
// to 96: 162 registers), eight otherwise: the loop is bound by its vector arithmetic (55 % of SIMD cycles against 20 % on
// the matrix pipe, by the counters), and a third wave per SIMD lets more of the two overlap (317 -> 297 us at S = 290).
template <int HD> constexpr int dkv_seq_threads() { return HD <= 96 ? 768 : 512; }
template <int HD, bool SEQ>
__global__ __launch_bounds__(SEQ ? dkv_seq_threads<HD>() : 256, SEQ ? (HD <= 96 ? 3 : 2) : 3) void attn_bwd_dkv_mfma(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<HD>;
    const int bh = xcd_logical(blockIdx.x, gridDim.x), b = bh / p.H, h = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int nq = min(p.nq, p.S - p.q_start);
    const int NQP = (nq + 31) & ~31;
    char* Ql = smem;
    char* Dl = smem + NQP * G::ROWB;
    float* lse_l = reinterpret_cast<float*>(smem + 2 * NQP * G::ROWB);
    float* del_l = lse_l + NQP;
    const int nthreads = blockDim.x;
    rows_to_lds2<HD, SEQ ? 8 : 2>(Ql, (const bf16_t*)p.q + b * p.q_bs + h * p.q_hs, p.q_ss, Dl, (const bf16_t*)p.dout + b * p.do_bs + h * p.do_hs, p.do_ss,
                                  p.q_start, nq, NQP, tid, nthreads);
    for (int c = tid; c < NQP; c += nthreads) {
        lse_l[c] = c < nq ? -p.lse[(long)bh * p.S + p.q_start + c] * ATT_LOG2E : 0.f;      // kept as -lse * log2(e)
        del_l[c] = c < nq ? p.delta[(long)bh * p.S + p.q_start + c] : 0.f;
    }
    __syncthreads();
    const int kstep = SEQ ? 16 * (nthreads >> 6) : (1 << 30);
    for (int kw = (SEQ ? 0 : blockIdx.y * 64) + wave * 16; kw < p.S; kw += kstep) {
    const int key = kw + i;
    const bool kok = key < p.S;
    const bf16_t* krow = (const bf16_t*)p.k + b * p.k_bs + h * p.k_hs + (long)key * p.k_ss;
    const bf16_t* vrow = (const bf16_t*)p.v + b * p.v_bs + h * p.v_hs + (long)key * p.v_ss;
    bf16x8 kf[G::KS], vf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        kf[ks] = kok ? *reinterpret_cast<const bf16x8*>(krow + ks * 32 + 8 * g) : zero_frag();
        vf[ks] = kok ? *reinterpret_cast<const bf16x8*>(vrow + ks * 32 + 8 * g) : zero_frag();
    }
    f32x4 dkacc[G::DT], dvacc[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) { dkacc[dt] = f32x4{0, 0, 0, 0}; dvacc[dt] = f32x4{0, 0, 0, 0}; }
    // this lane's key is seen by the local queries [qlo, qhi); a query block inside every lane's range needs no masking
    const bool open_key = p.mask_mode == CAPDEC_MASK_NONE || key < p.open_cols;
    const int qlo = open_key ? 0 : max(0, key - p.q_start);
    const int qhi = kok ? nq : 0;
    const bool open_tile = p.mask_mode == CAPDEC_MASK_NONE || kw + 15 < p.open_cols;
    const int qlo_wave = open_tile ? 0 : max(0, kw + 15 - p.q_start);
    const int qhi_wave = kw + 16 <= p.S ? nq : 0;
    // dropout: the 4 lanes of an aligned quad hold 4 adjacent keys of the same 4 queries, i.e. the 4 fields of the
    // same 4 hashes: each lane computes the hash of ONE query and the quad exchanges them
    const uint32_t grp_key = (uint32_t)(key >> 2), grp_stride = (uint32_t)(p.S4 >> 2);
    const uint32_t fsel = i & 3, fshift = (i & 1) * 16;

    for (int q0 = 0; q0 < NQP; q0 += 32) {
        bf16x8 pf, dsf;
        const bool masked = q0 < qlo_wave || q0 + 32 > qhi_wave;        // wave-uniform
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Ql, q0 + t * 16, ks * 32, lane), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Dl, q0 + t * 16, ks * 32, lane), vf[ks], dp, 0, 0, 0);
            }
            const int qb = q0 + t * 16 + 4 * g;                // first of the lane's 4 local query rows
            const f32x4 nl2 = *reinterpret_cast<const f32x4*>(lse_l + qb), dl4 = *reinterpret_cast<const f32x4*>(del_l + qb);
            DropBits hq;
            hq.a = hq.b = 0xFFFFFFFFu;
            if (p.drop_thr) hq = dropout_hash4(p.drop_k0, p.drop_k1, (uint32_t)(bh * p.S + p.q_start + qb + (int)fsel) * grp_stride + grp_key);
            // the word holding field (key & 3) of the hash of query r, which quad lane r computed
            const uint32_t hwa[4] = {quad_bcast<0>(hq.a), quad_bcast<1>(hq.a), quad_bcast<2>(hq.a), quad_bcast<3>(hq.a)};
            const uint32_t hwb[4] = {quad_bcast<0>(hq.b), quad_bcast<1>(hq.b), quad_bcast<2>(hq.b), quad_bcast<3>(hq.b)};
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], ATT_LOG2E, nl2[r]));
            if (masked) {
                masked_path();
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = (qb + r >= qlo && qb + r < qhi) ? pv[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = ((((fsel & 2) ? hwb[r] : hwa[r]) >> fshift) & 0xFFFFu) >= p.drop_thr;
                pf[t * 4 + r] = (bf16_t)(keep ? pv[r] : 0.f);                 // 1/(1-p) is applied to the finished dV
                dsf[t * 4 + r] = (bf16_t)(pv[r] * fmaf(keep ? dp[r] : 0.f, p.drop_scale, -dl4[r]));
            }
        }
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(Dl, q0, q0 + 16, dt * 16, lane), pf, dvacc[dt], 0, 0, 0);
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(Ql, q0, q0 + 16, dt * 16, lane), dsf, dkacc[dt], 0, 0, 0);
        }
    }
    {
        bf16_t* dkrow = (bf16_t*)p.dk + b * p.dk_bs + h * p.dk_hs + (long)key * p.dk_ss;
        bf16_t* dvrow = (bf16_t*)p.dv + b * p.dv_bs + h * p.dv_hs + (long)key * p.dv_ss;
        const int colq = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;
#pragma unroll
        for (int dt = 0; dt < G::DT; dt += 2) {
            // a lane's 4 + 4 columns of two adjacent 16-column tiles become 8 consecutive columns of one of them
            // (v_permlane16_swap per dword, every lane takes part): 16-byte stores, 64 contiguous bytes per row and instruction
            const u32x4 dk8 = swap_pair(dkacc[dt], dkacc[dt + 1], 1.f), dv8 = swap_pair(dvacc[dt], dvacc[dt + 1], p.drop_scale);
            if (kok) {
                *reinterpret_cast<u32x4*>(dkrow + dt * 16 + colq) = dk8;
                *reinterpret_cast<u32x4*>(dvrow + dt * 16 + colq) = dv8;
            }
        }
        if (p.colsum_kv) {                                           // the tile's column sums (keys past the sequence hold zeros)
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                dkacc[dt] = kok ? dkacc[dt] : f32x4{0, 0, 0, 0};
                dvacc[dt] = kok ? dvacc[dt] : f32x4{0, 0, 0, 0};
            }
            float* dst = p.colsum_kv + ((long)b * p.ntk + (kw >> 4)) * (2L * p.H * HD) + h * HD;
            fold_rows16<G::DT>(dkacc);
            write_colsums<G::DT>(dkacc, 1.f, g, i, dst);
            fold_rows16<G::DT>(dvacc);
            write_colsums<G::DT>(dvacc, p.drop_scale, g, i, dst + (long)p.H * HD);
        }
    }
    }
}

// ================================================================== backward, at most 32 queries: dQ, dK and dV in one kernel
// The scored-rows case of the one-block model: 32 queries per (sequence, head) against all S keys.  The two kernels above
// each read K and V of every head (the query-major one only to form 32 rows of dQ); here one workgroup of four waves per
// (sequence, head) reads them once.  A wave takes 32 keys per pass (two 16-key tiles): per tile it forms S and dP for both
// query tiles from K / V fragments that come straight from global memory (Q and dO of the 32 queries sit in LDS), dV and dK
// of the tile as the key-major kernel does, and leaves the tile's K rows and dS in a wave-private LDS area; after the two
// tiles the transposing reads of that area give K^T and dS^T, and dQ^T += K^T dS^T over the pass's 32 keys.  The waves'
// partial dQ are summed at the end in wave order (through LDS: deterministic), delta = rowsum(dO * O) is formed on the way in.
template <int HD> struct FusedGeo {
    using G = Geo<HD>;
    static constexpr int DSROWB = 32 * 2 + 16;                                  // a key's 32 dS values (bf16) + pad
    static constexpr int QD_BYTES = 2 * 32 * G::ROWB + 2 * 32 * 4;              // Q, dO, -lse * log2(e), delta
    static constexpr int WAVE_STAGE = 32 * G::ROWB + 32 * DSROWB;               // K rows and dS of the pass's 32 keys
    static constexpr int EXCH_WAVE = 2 * G::DT * 64 * 16;                       // a wave's partial dQ: 2 query tiles x DT tiles x f32x4 per lane
    static constexpr int AREA = 4 * WAVE_STAGE > 3 * EXCH_WAVE ? 4 * WAVE_STAGE : 3 * EXCH_WAVE;
    static constexpr int LDS = QD_BYTES + AREA;
};
// transposed operand from a [row][col] bf16 image with row pitch PITCH: lane (g,i) gets { M[rA + 4g + 0..3][col0 + i], M[rB + 4g + 0..3][col0 + i] }
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_cols_pitch(const char* lds, int rA, int rB, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int off = (4 * g + (i >> 2)) * PITCH + (col0 + 4 * (i & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, lds + rA * PITCH + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, lds + rB * PITCH + off));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int HD>
__global__ __launch_bounds__(256, HD <= 96 ? 3 : 2) void attn_bwd_fused32(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<HD>;
    using F = FusedGeo<HD>;
    const int bh = xcd_logical(blockIdx.x, gridDim.x), b = bh / p.H, h = bh - b * p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int nq = min(p.nq, p.S - p.q_start);                 // 1 .. 32
    char* Ql = smem;
    char* Dl = smem + 32 * G::ROWB;
    float* lse_l = reinterpret_cast<float*>(smem + 2 * 32 * G::ROWB);
    float* del_l = lse_l + 32;
    char* area = smem + F::QD_BYTES;
    char* Kst = area + wave * F::WAVE_STAGE;
    char* dSst = Kst + 32 * G::ROWB;
    const bf16_t* qbase = (const bf16_t*)p.q + b * p.q_bs + h * p.q_hs;
    const bf16_t* dobase = (const bf16_t*)p.dout + b * p.do_bs + h * p.do_hs;
    const bf16_t* kbase = (const bf16_t*)p.k + b * p.k_bs + h * p.k_hs;
    const bf16_t* vbase = (const bf16_t*)p.v + b * p.v_bs + h * p.v_hs;
    // K and V fragments of a 16-key tile, straight from memory (zero past S).  The first tile's are requested before the
    // staging of Q and dO, every later tile's as soon as the S / dP products have consumed the previous ones.
    bf16x8 kf[G::KS], vf[G::KS];
    auto load_kv = [&](int kw) {
        const int key = kw + i;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(kbase + (long)key * p.k_ss + ks * 32 + 8 * g) : zero_frag();
            vf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(vbase + (long)key * p.v_ss + ks * 32 + 8 * g) : zero_frag();
        }
    };
    load_kv(wave * 32);
    rows_to_lds2<HD, 2>(Ql, qbase, p.q_ss, Dl, dobase, p.do_ss, p.q_start, nq, 32, tid, 256);
    __syncthreads();
    {
        // delta = rowsum(dO * O): eight threads per query, each a few 16-byte chunks of the row (dO from LDS, O from memory)
        const int ql = tid >> 3, part = tid & 7;
        float dsum = 0.f;
        if (ql < nq) {
            const bf16_t* orow = (const bf16_t*)p.o + b * p.o_bs + h * p.o_hs + (long)(p.q_start + ql) * p.o_ss;
            for (int c = part; c < G::CPR; c += 8) {
                const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + c * 8);
                const bf16x8 df = *reinterpret_cast<const bf16x8*>(Dl + ql * G::ROWB + c * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum += (float)df[e] * (float)of[e];
            }
        }
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64);
        if (part == 0) {
            del_l[ql] = ql < nq ? dsum : 0.f;
            lse_l[ql] = ql < nq ? -p.lse[(long)bh * p.S + p.q_start + ql] * ATT_LOG2E : 0.f;
            if (ql < nq) p.delta[(long)bh * p.S + p.q_start + ql] = dsum;
        }
    }
    __syncthreads();

    f32x4 dqacc[2][G::DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) dqacc[qt][dt] = f32x4{0, 0, 0, 0};
    const uint32_t grp_stride = (uint32_t)(p.S4 >> 2);
    const uint32_t fsel = i & 3, fshift = (i & 1) * 16;

    for (int kp = wave * 32; kp < p.S; kp += 128) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kw = kp + t * 16, key = kw + i;
            const bool kok = key < p.S;
            // the tile's K rows, for the transposing reads of the dQ products
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) *reinterpret_cast<bf16x8*>(Kst + (t * 16 + i) * G::ROWB + (ks * 32 + 8 * g) * 2) = kf[ks];
            const bool open_key = p.mask_mode == CAPDEC_MASK_NONE || key < p.open_cols;
            const int qlo = open_key ? 0 : max(0, key - p.q_start);
            const int qhi = kok ? nq : 0;
            const bool open_tile = p.mask_mode == CAPDEC_MASK_NONE || kw + 15 < p.open_cols;
            const bool masked = !(open_tile || kw + 15 <= p.q_start) || nq < 32 || kw + 16 > p.S;        // wave-uniform
            const uint32_t grp_key = (uint32_t)(key >> 2);
            bf16x8 pf, dsf;
            f32x4 s2[2], dp2[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                s2[tq] = f32x4{0, 0, 0, 0}; dp2[tq] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    s2[tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Ql, tq * 16, ks * 32, lane), kf[ks], s2[tq], 0, 0, 0);
                    dp2[tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<HD>(Dl, tq * 16, ks * 32, lane), vf[ks], dp2[tq], 0, 0, 0);
                }
            }
            // the next tile's fragments replace the ones just consumed: the arithmetic and the products below cover their latency
            {
                const int nkw = t == 0 ? kw + 16 : kp + 128;
                if (nkw < p.S) load_kv(nkw);
            }
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const f32x4 s = s2[tq], dp = dp2[tq];
                const int qb = tq * 16 + 4 * g;                // first of the lane's 4 local query rows
                const f32x4 nl2 = *reinterpret_cast<const f32x4*>(lse_l + qb), dl4 = *reinterpret_cast<const f32x4*>(del_l + qb);
                DropBits hq;
                hq.a = hq.b = 0xFFFFFFFFu;
                if (p.drop_thr) hq = dropout_hash4(p.drop_k0, p.drop_k1, (uint32_t)(bh * p.S + p.q_start + qb + (int)fsel) * grp_stride + grp_key);
                const uint32_t hwa[4] = {quad_bcast<0>(hq.a), quad_bcast<1>(hq.a), quad_bcast<2>(hq.a), quad_bcast<3>(hq.a)};
                const uint32_t hwb[4] = {quad_bcast<0>(hq.b), quad_bcast<1>(hq.b), quad_bcast<2>(hq.b), quad_bcast<3>(hq.b)};
                float pv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], ATT_LOG2E, nl2[r]));
                if (masked) {
                    masked_path();
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (qb + r >= qlo && qb + r < qhi) ? pv[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool keep = ((((fsel & 2) ? hwb[r] : hwa[r]) >> fshift) & 0xFFFFu) >= p.drop_thr;
                    pf[tq * 4 + r] = (bf16_t)(keep ? pv[r] : 0.f);
                    dsf[tq * 4 + r] = (bf16_t)(pv[r] * fmaf(keep ? dp[r] : 0.f, p.drop_scale, -dl4[r]));
                }
            }
            // dS of (key i, queries 4g .. 4g+3 of either query tile) -> the wave's [key][query] image
            {
                const u32x4 du = __builtin_bit_cast(u32x4, dsf);
                *reinterpret_cast<u32x2*>(dSst + (t * 16 + i) * F::DSROWB + (4 * g) * 2) = u32x2{du[0], du[1]};
                *reinterpret_cast<u32x2*>(dSst + (t * 16 + i) * F::DSROWB + (16 + 4 * g) * 2) = u32x2{du[2], du[3]};
            }
            f32x4 dkacc[G::DT], dvacc[G::DT];
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(Dl, 0, 16, dt * 16, lane), pf, f32x4{0, 0, 0, 0}, 0, 0, 0);
                dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<HD>(Ql, 0, 16, dt * 16, lane), dsf, f32x4{0, 0, 0, 0}, 0, 0, 0);
            }
            bf16_t* dkrow = (bf16_t*)p.dk + b * p.dk_bs + h * p.dk_hs + (long)key * p.dk_ss;
            bf16_t* dvrow = (bf16_t*)p.dv + b * p.dv_bs + h * p.dv_hs + (long)key * p.dv_ss;
            const int colq = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;
#pragma unroll
            for (int dt = 0; dt < G::DT; dt += 2) {
                const u32x4 dk8 = swap_pair(dkacc[dt], dkacc[dt + 1], 1.f), dv8 = swap_pair(dvacc[dt], dvacc[dt + 1], p.drop_scale);
                if (kok) {
                    *reinterpret_cast<u32x4*>(dkrow + dt * 16 + colq) = dk8;
                    *reinterpret_cast<u32x4*>(dvrow + dt * 16 + colq) = dv8;
                }
            }
        }
        // dQ^T += K^T dS^T over the pass's 32 keys (the wave's own LDS area: written above by this wave only)
        bf16x8 dst[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dst[qt] = frag_cols_pitch<F::DSROWB>(dSst, 0, 16, qt * 16, lane);
#pragma unroll
        for (int dt = 0; dt < G::DT; ++dt) {
            const bf16x8 kt = frag_cols<HD>(Kst, 0, 16, dt * 16, lane);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dqacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dst[qt], dqacc[qt][dt], 0, 0, 0);
        }
    }
    // ---- the waves' partial dQ, summed in wave order
    __syncthreads();                                           // every wave is done with its staging area, which the exchange overlays
    if (wave > 0) {
        char* ex = area + (wave - 1) * F::EXCH_WAVE;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) *reinterpret_cast<f32x4*>(ex + ((qt * G::DT + dt) * 64 + lane) * 16) = dqacc[qt][dt];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int ql = qt * 16 + i;
            bf16_t* dqrow = (bf16_t*)p.dq + b * p.dq_bs + h * p.dq_hs + (long)(p.q_start + ql) * p.dq_ss;
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                f32x4 a = dqacc[qt][dt];
#pragma unroll
                for (int w = 0; w < 3; ++w) a += *reinterpret_cast<const f32x4*>(area + w * F::EXCH_WAVE + ((qt * G::DT + dt) * 64 + lane) * 16);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(a[r] * p.dq_scale);
                if (ql < nq) *reinterpret_cast<bf16x4*>(dqrow + dt * 16 + 4 * g) = o;
            }
        }
    }
}

// delta[bh][q] = sum_j do[q][j] * o[q][j]
template <typename T>
__global__ void attn_delta_kernel(AttnArgs p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.B * p.H * p.nq;
    if (idx >= total) return;
    const int h = idx % p.H;
    const long sb = idx / p.H;
    const int b = sb % p.B;
    const int q = p.q_start + (int)(sb / p.B);
    if (q >= p.S) return;
    const T* o = (const T*)p.o + b * p.o_bs + h * p.o_hs + (long)q * p.o_ss;
    const T* d = (const T*)p.dout + b * p.do_bs + h * p.do_hs + (long)q * p.do_ss;
    float a = 0.f;
    for (int j = 0; j < p.hd; ++j) a += to_f32<T>(o[j]) * to_f32<T>(d[j]);
    p.delta[((long)b * p.H + h) * p.S + q] = a;
}

// ================================================================== exact-f32 validation kernels
// one 64-thread block per (b, h, query)
template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_simple(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    float* pr = qs + p.hd;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int q = p.q_start + blockIdx.y, lane = threadIdx.x;
    if (q >= p.S) return;
    const T* qrow = (const T*)p.q + b * p.q_bs + h * p.q_hs + (long)q * p.q_ss;
    const T* kb = (const T*)p.k + b * p.k_bs + h * p.k_hs;
    const T* vb = (const T*)p.v + b * p.v_bs + h * p.v_hs;
    for (int j = lane; j < p.hd; j += 64) qs[j] = to_f32<T>(qrow[j]);
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < p.S; key += 64) {
        float s = -INFINITY;
        if (key_visible(q, key, p.S, p.mask_mode, p.open_cols)) {
            s = 0.f;
            const T* kr = kb + (long)key * p.k_ss;
            for (int j = 0; j < p.hd; ++j) s = fmaf(qs[j], to_f32<T>(kr[j]), s);
        }
        pr[key] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int key = lane; key < p.S; key += 64) {
        const float e = pr[key] == -INFINITY ? 0.f : expf(pr[key] - mx);
        sum += e;
        pr[key] = e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const uint32_t idx_row = (uint32_t)((bh * p.S + q) * (long)p.S4);
    for (int key = lane; key < p.S; key += 64) {
        float pv = pr[key] * inv;
        if (p.drop_thr) pv = dropout_keep(p.drop_k0, p.drop_k1, idx_row + key, p.drop_thr) ? pv * p.drop_scale : 0.f;
        pr[key] = pv;
    }
    __syncthreads();
    T* orow = (T*)p.o + b * p.o_bs + h * p.o_hs + (long)q * p.o_ss;
    for (int j = lane; j < p.hd; j += 64) {
        float a = 0.f;
        for (int key = 0; key < p.S; ++key) a = fmaf(pr[key], to_f32<T>(vb[(long)key * p.v_ss + j]), a);
        orow[j] = from_f32<T>(a);
    }
    if (lane == 0 && p.lse) p.lse[(long)bh * p.S + q] = mx + logf(sum);
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dq_simple(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    float* ds_ = qs + p.hd;
    float* dos = ds_ + p.S;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int q = p.q_start + blockIdx.y, lane = threadIdx.x;
    if (q >= p.S) return;
    const T* qrow = (const T*)p.q + b * p.q_bs + h * p.q_hs + (long)q * p.q_ss;
    const T* dorow = (const T*)p.dout + b * p.do_bs + h * p.do_hs + (long)q * p.do_ss;
    const T* kb = (const T*)p.k + b * p.k_bs + h * p.k_hs;
    const T* vb = (const T*)p.v + b * p.v_bs + h * p.v_hs;
    for (int j = lane; j < p.hd; j += 64) { qs[j] = to_f32<T>(qrow[j]); dos[j] = to_f32<T>(dorow[j]); }
    __syncthreads();
    const float lse = p.lse[(long)bh * p.S + q], delta = p.delta[(long)bh * p.S + q];
    const uint32_t idx_row = (uint32_t)((bh * p.S + q) * (long)p.S4);
    for (int key = lane; key < p.S; key += 64) {
        float dsv = 0.f;
        if (key_visible(q, key, p.S, p.mask_mode, p.open_cols)) {
            float s = 0.f, dp = 0.f;
            const T* kr = kb + (long)key * p.k_ss;
            const T* vr = vb + (long)key * p.v_ss;
            for (int j = 0; j < p.hd; ++j) { s = fmaf(qs[j], to_f32<T>(kr[j]), s); dp = fmaf(dos[j], to_f32<T>(vr[j]), dp); }
            if (p.drop_thr) dp = dropout_keep(p.drop_k0, p.drop_k1, idx_row + key, p.drop_thr) ? dp * p.drop_scale : 0.f;
            dsv = expf(s - lse) * (dp - delta);
        }
        ds_[key] = dsv;
    }
    __syncthreads();
    T* dqrow = (T*)p.dq + b * p.dq_bs + h * p.dq_hs + (long)q * p.dq_ss;
    for (int j = lane; j < p.hd; j += 64) {
        float a = 0.f;
        for (int key = 0; key < p.S; ++key) a = fmaf(ds_[key], to_f32<T>(kb[(long)key * p.k_ss + j]), a);
        dqrow[j] = from_f32<T>(a * p.dq_scale);
    }
}

// one 64-thread block per (b, h, key)
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dkv_simple(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks_ = reinterpret_cast<float*>(smem);
    float* vs_ = ks_ + p.hd;
    float* pd_ = vs_ + p.hd;      // dropped probabilities per query
    float* ds_ = pd_ + p.S;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int key = blockIdx.y, lane = threadIdx.x;
    const T* krow = (const T*)p.k + b * p.k_bs + h * p.k_hs + (long)key * p.k_ss;
    const T* vrow = (const T*)p.v + b * p.v_bs + h * p.v_hs + (long)key * p.v_ss;
    const T* qb = (const T*)p.q + b * p.q_bs + h * p.q_hs;
    const T* dob = (const T*)p.dout + b * p.do_bs + h * p.do_hs;
    for (int j = lane; j < p.hd; j += 64) { ks_[j] = to_f32<T>(krow[j]); vs_[j] = to_f32<T>(vrow[j]); }
    __syncthreads();
    const int qend = min(p.q_start + p.nq, p.S);
    for (int q = p.q_start + lane; q < qend; q += 64) {
        float pd = 0.f, dsv = 0.f;
        if (key_visible(q, key, p.S, p.mask_mode, p.open_cols)) {
            float s = 0.f, dp = 0.f;
            const T* qr = qb + (long)q * p.q_ss;
            const T* dr = dob + (long)q * p.do_ss;
            for (int j = 0; j < p.hd; ++j) { s = fmaf(to_f32<T>(qr[j]), ks_[j], s); dp = fmaf(to_f32<T>(dr[j]), vs_[j], dp); }
            const float pv = expf(s - p.lse[(long)bh * p.S + q]);
            pd = pv;
            if (p.drop_thr) {
                const bool keep = dropout_keep(p.drop_k0, p.drop_k1, (uint32_t)((bh * p.S + q) * (long)p.S4) + key, p.drop_thr);
                pd = keep ? pv * p.drop_scale : 0.f;
                dp = keep ? dp * p.drop_scale : 0.f;
            }
            dsv = pv * (dp - p.delta[(long)bh * p.S + q]);
        }
        pd_[q] = pd; ds_[q] = dsv;
    }
    __syncthreads();
    T* dkrow = (T*)p.dk + b * p.dk_bs + h * p.dk_hs + (long)key * p.dk_ss;
    T* dvrow = (T*)p.dv + b * p.dv_bs + h * p.dv_hs + (long)key * p.dv_ss;
    for (int j = lane; j < p.hd; j += 64) {
        float a = 0.f, c = 0.f;
        for (int q = p.q_start; q < qend; ++q) {
            a = fmaf(ds_[q], to_f32<T>(qb[(long)q * p.q_ss + j]), a);
            c = fmaf(pd_[q], to_f32<T>(dob[(long)q * p.do_ss + j]), c);
        }
        dkrow[j] = from_f32<T>(a);
        dvrow[j] = from_f32<T>(c);
    }
}

// ================================================================== decode: one query row vs. the K/V cache
// cache [B][H][2][smax][HD]; thread (kk, part) streams 16-B chunks, fully coalesced.
template <typename T, int HD>
__global__ void attn_decode_kernel(const T* q, const T* cache, T* o, int H, int smax, int len) {
    constexpr int E = 16 / sizeof(T);            // elements per 16-B chunk
    constexpr int PARTS = HD / E;
    constexpr int NTH = (HD == 96) ? 192 : 256;
    constexpr int KPP = NTH / PARTS;             // keys per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);          // [HD]
    float* sc = qs + HD;                                 // [len]
    float* pd = sc + ((len + 3) & ~3);                   // [len*PARTS] then reused as [KPP][HD]
    __shared__ float red[16];
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int part = tid % PARTS, kk = tid / PARTS;
    const T* Kc = cache + ((long)bh * 2 + 0) * smax * HD;
    const T* Vc = cache + ((long)bh * 2 + 1) * smax * HD;
    for (int j = tid; j < HD; j += NTH) qs[j] = to_f32<T>(q[(long)bh * HD + j]);
    __syncthreads();
    // K and V are streamed once per generated token: non-temporal loads keep them from evicting the weights
    for (int key = kk; key < len; key += KPP) {
        const T* src = Kc + (long)key * HD + part * E;
        float a = 0.f;
        if constexpr (sizeof(T) == 2) {
            const bf16x8 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src));
#pragma unroll
            for (int e = 0; e < E; ++e) a = fmaf(qs[part * E + e], (float)v[e], a);
        } else {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
#pragma unroll
            for (int e = 0; e < E; ++e) a = fmaf(qs[part * E + e], v[e], a);
        }
        pd[key * PARTS + part] = a;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int key = tid; key < len; key += NTH) {
        float s = 0.f;
        for (int pp = 0; pp < PARTS; ++pp) s += pd[key * PARTS + pp];
        sc[key] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int key = tid; key < len; key += NTH) { const float e = expf(sc[key] - mx); sc[key] = e; sum += e; }
    sum = block_sum(sum, red);
    __syncthreads();
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int key = kk; key < len; key += KPP) {
        const T* src = Vc + (long)key * HD + part * E;
        const float w = sc[key];
        if constexpr (sizeof(T) == 2) {
            const bf16x8 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src));
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(w, (float)v[e], acc[e]);
        } else {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(w, v[e], acc[e]);
        }
    }
    __syncthreads();   // everyone is done reading pd as scores
#pragma unroll
    for (int e = 0; e < E; ++e) pd[kk * HD + part * E + e] = acc[e];
    __syncthreads();
    const float inv = 1.f / sum;
    for (int j = tid; j < HD; j += NTH) {
        float a = 0.f;
        for (int k2 = 0; k2 < KPP; ++k2) a += pd[k2 * HD + j];
        o[(long)bh * HD + j] = from_f32<T>(a * inv);
    }
}

// ------------------------------------------------------------------ host helpers
int fill_args(const capdec_attn_desc* d, AttnArgs& a) {
    CAPDEC_CHECK(d && d->q && d->k && d->v, "attn: null pointer");
    CAPDEC_CHECK(d->dtype == CAPDEC_BF16 || d->dtype == CAPDEC_F32, "attn: bad dtype");
    CAPDEC_CHECK(d->B > 0 && d->H > 0 && d->S > 0 && d->hd > 0, "attn: bad shape");
    CAPDEC_CHECK(d->q_start >= 0 && d->nq >= 0 && d->q_start + d->nq <= d->S, "attn: query range [%d,%d) outside sequence %d", d->q_start, d->q_start + d->nq, d->S);
    CAPDEC_CHECK((long)d->B * d->H * d->S * ((d->S + 3) & ~3) < (1L << 32) || d->drop.p <= 0.f, "attn: dropout index space exceeds 2^32");
    memset(&a, 0, sizeof(a));
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o;
    a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_ss = d->q_ss;
    a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_ss = d->k_ss;
    a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_ss = d->v_ss;
    a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_ss = d->o_ss;
    a.B = d->B; a.H = d->H; a.S = d->S; a.hd = d->hd; a.q_start = d->q_start; a.nq = d->nq;
    a.S4 = (d->S + 3) & ~3;
    a.mask_mode = d->mask_mode;
    a.open_cols = d->mask_mode == CAPDEC_MASK_REFERENCE ? 16 * 768 : (d->mask_mode == CAPDEC_MASK_PREFIX_CAUSAL ? d->prefix_len : 0);
    a.drop_thr = d->drop.p > 0.f ? dropout_threshold(d->drop.p) : 0u;
    a.drop_k0 = d->drop.key0; a.drop_k1 = d->drop.key1;
    a.drop_scale = d->drop.p > 0.f ? 1.f / (1.f - d->drop.p) : 1.f;
    a.lse = d->lse;
    return CAPDEC_OK;
}

// grid of the persistent whole-sequence kernels: one workgroup per CU of the MI355X, each walking (sequence, head) pairs
unsigned seq_grid(const AttnArgs& a) { const long n = (long)a.B * a.H; return (unsigned)(n < 256 ? n : 256); }

bool strides_ok8(long a, long b, long c) { return (a % 8) == 0 && (b % 8) == 0 && (c % 8) == 0; }

template <typename K>
void set_lds(K kernel, int bytes) {
    (void)bytes;                                      // the ceiling is raised once to the CU's 160 KiB, whatever this launch asks for
    ensure_dynamic_lds((const void*)kernel, 160 * 1024);
}

#define HD_DISPATCH(HDV, ...)                                      \
    switch (HDV) {                                                 \
        case 32: { constexpr int HD = 32; __VA_ARGS__; } break;    \
        case 64: { constexpr int HD = 64; __VA_ARGS__; } break;    \
        case 96: { constexpr int HD = 96; __VA_ARGS__; } break;    \
        case 128: { constexpr int HD = 128; __VA_ARGS__; } break;  \
        default: capdec_set_error("attn: head dim %d not supported (32/64/96/128)", HDV); return CAPDEC_ERR_INVALID; \
    }

}  // namespace

// ==================================================================== C ABI
namespace { int g_attn_blocks = 1; }
// diagnostic hook (not part of the declared ABI): 0 = forwards of at most 32 queries stage V whole, as longer ranges do
extern "C" void capdecdbg_attn_block_staging(int on) { g_attn_blocks = on; }
static int g_attn_lean_dq = 1;    // whole-sequence dQ kernel with 12 waves and no register look-ahead (diagnostic hook: 0 = 8 waves with look-ahead)
extern "C" void capdecdbg_attn_lean_dq(int on) { g_attn_lean_dq = on; }
static int g_attn_fused = 1;      // at most 32 queries: dQ, dK, dV from one kernel (diagnostic hook: 0 = the two kernels)
extern "C" void capdecdbg_attn_fused_bwd(int on) { g_attn_fused = on; }

extern "C" int capdec_attn_fwd(const capdec_attn_desc* d, void* stream) {
    AttnArgs a;
    int rc = fill_args(d, a);
    if (rc) return rc;
    CAPDEC_CHECK(d->o, "attn_fwd: null output");
    if (d->nq == 0) return CAPDEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == CAPDEC_BF16) {
        CAPDEC_CHECK(strides_ok8(a.q_bs, a.q_hs, a.q_ss) && strides_ok8(a.k_bs, a.k_hs, a.k_ss) && strides_ok8(a.v_bs, a.v_hs, a.v_ss) &&
                     (a.o_bs % 4) == 0 && (a.o_hs % 4) == 0 && (a.o_ss % 4) == 0 && aligned16(a.q) && aligned16(a.k) && aligned16(a.v) && aligned16(a.o),
                     "attn_fwd(bf16): strides must be multiples of 8 elements and bases 16-byte aligned");
        const int SP = (a.S + 63) & ~63;
        const int lds = SP * (a.hd * 2 + 16);
        CAPDEC_CHECK(lds <= 160 * 1024, "attn_fwd(bf16): sequence %d too long for the LDS-resident V design (hd=%d)", a.S, a.hd);
        if (a.nq > 64 && 2 * lds <= 160 * 1024) {
            // long query range: one workgroup of 8 waves per (sequence, head), K and V both resident in LDS
            HD_DISPATCH(a.hd, set_lds(attn_fwd_mfma<HD, true>, 2 * lds);
                        hipLaunchKernelGGL((attn_fwd_mfma<HD, true>), dim3(seq_grid(a)), dim3(512), 2 * lds, st, a));
        } else {
            const int nwaves = a.nq >= 64 ? 4 : (a.nq + 15) / 16;        // a wave per 16 queries, at most 4 per workgroup
            dim3 grid(a.B * a.H, (a.nq + 16 * nwaves - 1) / (16 * nwaves)), block(64 * nwaves);
            const int lds_blk = 2 * 64 * (a.hd * 2 + 16);                 // two 64-key blocks of V
            if (nwaves == 1 && g_attn_blocks) {
                HD_DISPATCH(a.hd, set_lds(attn_fwd_mfma<HD, false, 1>, lds_blk); hipLaunchKernelGGL((attn_fwd_mfma<HD, false, 1>), grid, block, lds_blk, st, a));
            } else if (nwaves == 2 && g_attn_blocks) {
                HD_DISPATCH(a.hd, set_lds(attn_fwd_mfma<HD, false, 2>, lds_blk); hipLaunchKernelGGL((attn_fwd_mfma<HD, false, 2>), grid, block, lds_blk, st, a));
            } else {
                HD_DISPATCH(a.hd, set_lds(attn_fwd_mfma<HD, false>, lds); hipLaunchKernelGGL((attn_fwd_mfma<HD, false>), grid, block, lds, st, a));
            }
        }
    } else {
        const int lds = (a.hd + a.S) * 4;
        dim3 grid(a.B * a.H, a.nq), block(64);
        hipLaunchKernelGGL(attn_fwd_simple<float>, grid, block, lds, st, a);
    }
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

namespace {
// which kernels capdec_attn_bwd runs and their grids along the queries (dq kernel) / keys (dk/dv kernel) of a
// (sequence, head); 16-row tiles per (sequence, head), each of which leaves one row of partial column sums
struct BwdPlan { bool seq_q, seq_kv; int nwaves_q, nyq, nyk, ntq, ntk; };
BwdPlan bwd_plan(int S, int nq, int hd) {
    BwdPlan pl;
    const int SP = (S + 63) & ~63;
    const int lds_q = SP * (hd * 2 + 16);
    pl.seq_q = nq > 64 && 2 * lds_q <= 160 * 1024;
    pl.seq_kv = nq > 64 && S > 128;
    pl.nwaves_q = nq >= 64 ? 4 : (nq + 15) / 16;
    pl.nyq = pl.seq_q ? 1 : (nq > 0 ? (nq + 16 * pl.nwaves_q - 1) / (16 * pl.nwaves_q) : 0);
    pl.nyk = pl.seq_kv ? 1 : (S + 63) / 64;
    pl.ntq = (nq + 15) / 16; pl.ntk = (S + 15) / 16;               // 16-row tiles: one row of partial column sums each
    return pl;
}
}  // namespace

extern "C" void capdec_attn_bwd_colsum_rows(int B, int S, int nq, int hd, long* q_rows, long* kv_rows) {
    const BwdPlan pl = bwd_plan(S, nq, hd);
    if (q_rows) *q_rows = (long)B * pl.ntq;
    if (kv_rows) *kv_rows = (long)B * pl.ntk;
}

extern "C" int capdec_attn_bwd(const capdec_attn_bwd_desc* d, void* stream) {
    CAPDEC_CHECK(d, "attn_bwd: null descriptor");
    AttnArgs a;
    int rc = fill_args(&d->fwd, a);
    if (rc) return rc;
    CAPDEC_CHECK(d->fwd.o && d->fwd.lse && d->dout && d->dq && d->dk && d->dv && d->delta, "attn_bwd: null pointer");
    a.dout = d->dout; a.do_bs = d->do_bs; a.do_hs = d->do_hs; a.do_ss = d->do_ss;
    a.dq = d->dq; a.dq_bs = d->dq_bs; a.dq_hs = d->dq_hs; a.dq_ss = d->dq_ss;
    a.dk = d->dk; a.dk_bs = d->dk_bs; a.dk_hs = d->dk_hs; a.dk_ss = d->dk_ss;
    a.dv = d->dv; a.dv_bs = d->dv_bs; a.dv_hs = d->dv_hs; a.dv_ss = d->dv_ss;
    a.dq_scale = d->dq_scale; a.delta = d->delta;
    const BwdPlan pl = bwd_plan(a.S, a.nq, a.hd);
    a.colsum_q = d->colsum_q; a.colsum_kv = d->colsum_kv; a.ntq = pl.ntq; a.ntk = pl.ntk;
    CAPDEC_CHECK((!d->colsum_q && !d->colsum_kv) || d->fwd.dtype == CAPDEC_BF16, "attn_bwd: column sums come with the bf16 kernels only");
    CAPDEC_CHECK((!d->colsum_q || aligned16(d->colsum_q)) && (!d->colsum_kv || aligned16(d->colsum_kv)), "attn_bwd: column-sum buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long nrow = (long)a.B * a.H * a.nq;
    // delta = rowsum(dO * O): the bf16 query-major kernel produces it itself
    if (a.nq > 0 && d->fwd.dtype != CAPDEC_BF16)
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, st, a);
    if (d->fwd.dtype == CAPDEC_BF16) {
        CAPDEC_CHECK(strides_ok8(a.q_bs, a.q_hs, a.q_ss) && strides_ok8(a.k_bs, a.k_hs, a.k_ss) && strides_ok8(a.v_bs, a.v_hs, a.v_ss) &&
                     strides_ok8(a.do_bs, a.do_hs, a.do_ss) && (a.dq_bs % 4) == 0 && (a.dq_hs % 4) == 0 && (a.dq_ss % 4) == 0 &&
                     strides_ok8(a.dk_bs, a.dk_hs, a.dk_ss) && strides_ok8(a.dv_bs, a.dv_hs, a.dv_ss) &&
                     aligned16(a.dout) && aligned16(a.dq) && aligned16(a.dk) && aligned16(a.dv),
                     "attn_bwd(bf16): strides must be multiples of 8 elements and bases 16-byte aligned");
        if (a.nq > 0 && a.nq <= 32 && !a.colsum_q && !a.colsum_kv && g_attn_fused) {
            // the scored-rows case: one kernel for dQ, dK and dV (K and V of a head are read once)
            HD_DISPATCH(a.hd, set_lds(attn_bwd_fused32<HD>, FusedGeo<HD>::LDS);
                        hipLaunchKernelGGL((attn_bwd_fused32<HD>), dim3(a.B * a.H), dim3(256), FusedGeo<HD>::LDS, st, a));
            CAPDEC_LAUNCH_CHECK();
            return CAPDEC_OK;
        }
        const int SP = (a.S + 63) & ~63;
        const int lds_q = SP * (a.hd * 2 + 16);
        const int NQP = (a.nq + 31) & ~31;
        const int lds_k = 2 * NQP * (a.hd * 2 + 16) + 2 * NQP * 4;
        CAPDEC_CHECK(lds_q <= 160 * 1024 && lds_k <= 160 * 1024, "attn_bwd(bf16): sequence %d too long for the LDS-resident design (hd=%d)", a.S, a.hd);
        dim3 block(256);
        if (pl.seq_q && g_attn_lean_dq && a.hd <= 96) {
            HD_DISPATCH(a.hd, set_lds(attn_bwd_dq_mfma<HD, true, 0, true>, 2 * lds_q);
                        hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, true, 0, true>), dim3(a.B * a.H), dim3(1024), 2 * lds_q, st, a));
        } else if (pl.seq_q) {
            HD_DISPATCH(a.hd, set_lds(attn_bwd_dq_mfma<HD, true>, 2 * lds_q);
                        hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, true>), dim3(a.B * a.H), dim3(512), 2 * lds_q, st, a));
        } else if (a.nq > 0) {
            const int nwaves = pl.nwaves_q;
            dim3 gq(a.B * a.H, pl.nyq), bq(64 * nwaves);
            const int lds_blk = 2 * 64 * (a.hd * 2 + 16);                 // two 64-key blocks of K
            if (nwaves == 1 && g_attn_blocks) {
                HD_DISPATCH(a.hd, set_lds(attn_bwd_dq_mfma<HD, false, 1>, lds_blk); hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, false, 1>), gq, bq, lds_blk, st, a));
            } else if (nwaves == 2 && g_attn_blocks) {
                HD_DISPATCH(a.hd, set_lds(attn_bwd_dq_mfma<HD, false, 2>, lds_blk); hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, false, 2>), gq, bq, lds_blk, st, a));
            } else {
                HD_DISPATCH(a.hd, set_lds(attn_bwd_dq_mfma<HD, false>, lds_q); hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, false>), gq, bq, lds_q, st, a));
            }
        }
        if (pl.seq_kv) {
            // many queries staged per workgroup: stage them once per (sequence, head) and walk every key tile
            HD_DISPATCH(a.hd, set_lds(attn_bwd_dkv_mfma<HD, true>, lds_k);
                        hipLaunchKernelGGL((attn_bwd_dkv_mfma<HD, true>), dim3(a.B * a.H), dim3(dkv_seq_threads<HD>()), lds_k, st, a));
        } else {
            dim3 gk(a.B * a.H, pl.nyk);
            HD_DISPATCH(a.hd, set_lds(attn_bwd_dkv_mfma<HD, false>, lds_k); hipLaunchKernelGGL((attn_bwd_dkv_mfma<HD, false>), gk, block, lds_k, st, a));
        }
    } else {
        if (a.nq > 0) {
            const int lds_q = (2 * a.hd + a.S) * 4;
            hipLaunchKernelGGL(attn_bwd_dq_simple<float>, dim3(a.B * a.H, a.nq), dim3(64), lds_q, st, a);
        }
        const int lds_k = (2 * a.hd + 2 * a.S) * 4;
        hipLaunchKernelGGL(attn_bwd_dkv_simple<float>, dim3(a.B * a.H, a.S), dim3(64), lds_k, st, a);
    }
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_attn_decode(const void* q, const void* kv_cache, void* o, int dtype,
                                  int B, int H, int hd, int smax, int len, void* stream) {
    CAPDEC_CHECK(q && kv_cache && o, "attn_decode: null pointer");
    CAPDEC_CHECK(B > 0 && H > 0 && len > 0 && len <= smax, "attn_decode: bad shape (len=%d, smax=%d)", len, smax);
    CAPDEC_CHECK(dtype == CAPDEC_BF16 || dtype == CAPDEC_F32, "attn_decode: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int nth = hd == 96 ? 192 : 256;
    const int E = dtype == CAPDEC_BF16 ? 8 : 4;
    const int parts = hd / E, kpp = nth / parts;
    const int pd_floats = len * parts > kpp * hd ? len * parts : kpp * hd;
    const int lds = (hd + ((len + 3) & ~3) + pd_floats) * 4;
    CAPDEC_CHECK(lds <= 64 * 1024, "attn_decode: cache length %d too long", len);
    dim3 grid(B * H), block(nth);
    if (dtype == CAPDEC_BF16) {
        HD_DISPATCH(hd, hipLaunchKernelGGL((attn_decode_kernel<bf16_t, HD>), grid, block, lds, st, (const bf16_t*)q, (const bf16_t*)kv_cache, (bf16_t*)o, H, smax, len));
    } else {
        HD_DISPATCH(hd, hipLaunchKernelGGL((attn_decode_kernel<float, HD>), grid, block, lds, st, (const float*)q, (const float*)kv_cache, (float*)o, H, smax, len));
    }
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
