// Fused vocabulary projection + cross-entropy forward for the training / greedy hot path (gfx950), replacing
// decoder.py:66 on the scored rows + decoder.py:172-183 without ever storing logits:
//
//   per (row, 64-column slot): a reference value r >= (nearly) max_n z, sum_n exp(z - r), argmax_n z
//   optionally e[m, n] = exp(z[m, n] - r(m, slot(n))) as bf16 (the softmax numerators the backward starts from)
//
// with z = h W^T + bias, bf16 operands, f32 accumulation (MFMA 16x16x32).  What differs from the generic GEMM
// kernels of gemm.hip, which this one takes its LDS-DMA ring from:
//   * PERSISTENT workgroups: each walks its share of the output tiles, and the operand ring never stops -- the
//     first stages of the next tile are fetched during the last k-steps of the current one, so a tile has no
//     prologue, and no workgroup is launched per tile;
//   * the epilogue works on the accumulator REGISTERS: a lane holds 16 of the 64 columns of one row per 16-row
//     block, the other 48 are in the lanes 16, 32 and 48 further on, and two v_permlane swaps finish a row
//     reduction.  No LDS staging, so the ring keeps filling underneath, and nothing waits for LDS round trips;
//   * maximum and argmax in one pass: the 6-bit column index replaces the low mantissa bits of each logit and an
//     ordinary float max carries it along.  The reference r is that packed value (within 2^-17 relative of the true
//     maximum), which is all a softmax reference needs: numerators, sums and the final log-sum-exp stay exact.
//     TIE RULE of the argmax: logits within 64 ulp (2^-17 relative) of each other are not told apart by value; among
//     them the lowest column wins (exactly equal logits therefore resolve as torch.argmax does).  Mode 0 holds to
//     that for either sign; mode 3 packs for positive values only -- its values are z - z_label, a row's maximum is
//     never below the label's ~0, and a negative slot maximum cannot win the row;
//   * the bias initialises the accumulators (first MFMA of a tile takes it as its C operand); a tile's slice of it
//     travels through the LDS like the operands do (one more DMA piece per tile), because an ordinary register load
//     next to in-flight LDS-DMA makes the compiler drain every outstanding memory operation before its first use;
//   * numerators leave as 16-byte stores: a v_permlane16_swap per dword turns each lane's 2 x 4 columns of two
//     adjacent 16-column tiles into 8 consecutive columns.
// The label's logit (needed for the loss) is not extracted here: the finalize kernel forms it as a 768-long dot
// product per row.
#include <type_traits>

#include "head_fwd.h"

namespace {

constexpr int DBK = 32;                          // k-depth of a ring stage
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PAST = -3.0e38f;                 // "logit" of a column past N: finite, so that index bits can be packed into it

// row-major [rows][32 k] bf16 stage image, 64-B rows; chunk (0..3) XOR 3*((row>>3)&1): conflict-free ds_read_b128
__device__ __forceinline__ int dlds(int row, int kc) { return row * 64 + ((kc ^ (((row >> 3) & 1) * 3)) << 4); }

template <int WM, int WN, int TI, int NST> struct Geo {
    static constexpr int NW = WM * WN;
    static constexpr int WROWS = 16 * TI;
    static constexpr int BMV = WM * WROWS, BNV = WN * 64;
    static constexpr int A_BYTES = BMV * 64, B_BYTES = BNV * 64, STAGE = A_BYTES + B_BYTES;
    static constexpr int NJA = A_BYTES / 1024 / NW, NJB = B_BYTES / 1024 / NW, NP = NJA + NJB;
    static constexpr int INFLIGHT = (NST - 2) * NP;      // DMA instructions that may stay outstanding at a step's wait
    static constexpr int RING = NST * STAGE;
    static constexpr int BIAS_BYTES = 1024;                // one DMA piece: up to 256 columns of f32 bias
    static constexpr int REF_BYTES = 1024;                 // one DMA piece: a tile's 256 rows of f32 row references
    static constexpr int LDS = RING + 2 * BIAS_BYTES + 2 * REF_BYTES;      // + the current and the next tile's bias and row-reference slices
    static constexpr int WG_PER_CU = (2 * LDS <= 160 * 1024 && NW == 4) ? 2 : 1;
    static_assert(BNV * 4 <= BIAS_BYTES, "a tile's bias slice is one DMA piece");
    static_assert(NP <= TI, "one DMA piece per MFMA row at most");
    static_assert(BMV * 4 <= REF_BYTES && NW >= 2, "a tile's row references are one DMA piece, fetched by wave 1");
};

struct View { const char* base; long rem; };     // an operand panel: first byte, bytes addressable from there

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const char* base, long rem) {
    const unsigned rec = rem <= 0 ? 0u : (rem > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)rem);
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)rec, 0x00020000);
}
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned off, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, off, 0, 0, 0);
}

// tile order: each XCD (blocks b, b + 8, ... share one) gets a contiguous run of logical ids, which walk groups of
// 8 M-tiles x all N-tiles, M fastest: the 64 tiles an XCD works on at a time share 8 row panels and 8 column panels
__device__ __forceinline__ void map_tile(int vb, int tm, int tn, int& tile_m, int& tile_n) {
    const int nwg = tm * tn;
    const int qq = nwg >> 3, r = nwg & 7, xcd = vb & 7;
    const int logical = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (vb >> 3);
    const int per_group = 8 * tn;
    const int grp = logical / per_group;
    const int first_m = grp * 8;
    const int gsz = min(tm - first_m, 8);
    const int in = logical - grp * per_group;
    tile_m = first_m + in % gsz;
    tile_n = in / gsz;
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// x combined with the values the lanes 16, 32 and 48 further on (mod 64) hold
__device__ __forceinline__ float quad_max(float x) {
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}

template <int WM, int WN, int TI, int NST, bool KEEP, bool STAMPS, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN, (Geo<WM, WN, TI, NST>::WG_PER_CU)) void head_fwd_kernel(HeadFwdArgs p) {
    using G = Geo<WM, WN, TI, NST>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int tm = (p.M + G::BMV - 1) / G::BMV, tn = (p.N + G::BNV - 1) / G::BNV;
    const int ntiles = tm * tn;
    const int nk = p.K / DBK;                               // whole 32-deep steps, at least NST of them (checked on the host)
    // Half tiles (the standard and the residual epilogue on 256x256 tiles): when the tile count leaves a last round that at
    // most half the workgroups would take, its tiles are cut into two 256x128 halves, one per workgroup (a wave then owns
    // 128x32), so that round costs about half a tile's time instead of a whole one.  nfull tiles are walked as they are.
    constexpr bool HALVES = (MODE == 1 || MODE == 2) && WM == 2 && WN == 4;
    const int nfull = HALVES ? ntiles - p.half_tiles : ntiles;

    // what a lane fetches for its pieces of a stage: tile-independent
    unsigned offa[G::NJA], offb[G::NJB];
#pragma unroll
    for (int j = 0; j < G::NJA; ++j) {
        const int row = 16 * (G::NW * j + wave) + (lane >> 2), kc = (lane & 3) ^ (((row >> 3) & 1) * 3);
        offa[j] = (unsigned)(((long)row * p.lda + kc * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < G::NJB; ++j) {
        const int row = 16 * (G::NW * j + wave) + (lane >> 2), kc = (lane & 3) ^ (((row >> 3) & 1) * 3);
        offb[j] = (unsigned)(((long)row * p.ldb + kc * 8) * 2);
    }
    const int fragA = dlds(wr * G::WROWS + r16, q), fragB = G::A_BYTES + dlds(wc * 64 + r16, q);   // + 1024 per 16 rows
    const int fragBh = G::A_BYTES + dlds(wc * 32 + r16, q);                                        // in a half tile

    int m0 = 0, n0 = 0;
    // returns 0: past the last tile, 1: a whole tile, 2: a half tile (columns nn .. nn + 127)
    auto views_of = [&](int vb, View& va, View& vbv, int& mm, int& nn) {
        int kind = vb < nfull ? 1 : 0, id = vb, hsel = 0;
        if (HALVES && kind == 0 && vb < nfull + (int)gridDim.x) {
            // the workgroups w and w + 8 (same XCD) share tile nfull + pair: the left and the right half
            const int w = vb - nfull, pair = (w >> 4) * 8 + (w & 7);
            if (pair < p.half_tiles) { kind = 2; id = nfull + pair; hsel = (w >> 3) & 1; }
        }
        if (kind) {
            int tile_m, tile_n;
            map_tile(id, tm, tn, tile_m, tile_n);
            mm = tile_m * G::BMV; nn = tile_n * G::BNV + hsel * 128;
            va.base = (const char*)(p.A + (long)mm * p.lda); va.rem = (long)(p.M - mm) * p.lda * 2;
            vbv.base = (const char*)(p.B + (long)nn * p.ldb); vbv.rem = (long)(p.N - nn) * p.ldb * 2;
        } else {                                          // past the last tile: loads that fetch nothing but still count
            mm = nn = 0;
            va.base = (const char*)p.A; va.rem = 0;
            vbv.base = (const char*)p.B; vbv.rem = 0;
        }
        return kind;
    };
    auto issue_stage = [&](const View& va, const View& vbv, int k, int slot) {
        char* st = smem + slot * G::STAGE;
        const __amdgpu_buffer_rsrc_t rsa = rsrc_at(va.base + k * p.kstepA, va.rem - (long)k * p.remdecA);
        const __amdgpu_buffer_rsrc_t rsb = rsrc_at(vbv.base + k * p.kstepB, vbv.rem - (long)k * p.remdecB);
#pragma unroll
        for (int j = 0; j < G::NJA; ++j) dma_piece(rsa, offa[j], st + (G::NW * j + wave) * 1024);
#pragma unroll
        for (int j = 0; j < G::NJB; ++j) dma_piece(rsb, offb[j], st + G::A_BYTES + (G::NW * j + wave) * 1024);
    };
    // a tile's bias slice -> LDS (wave 0 only; lanes past the slice read the following columns or nothing)
    auto issue_bias = [&](int nn, int sel) {
        if (p.bias && wave == 0) {
            const __amdgpu_buffer_rsrc_t rb = rsrc_at((const char*)(p.bias + nn), (long)(p.N - nn) * 4);
            dma_piece(rb, (unsigned)lane * 16u, smem + G::RING + sel * G::BIAS_BYTES);
        }
    };
    // mode 3: the tile's slice of the row references -> LDS the same way (wave 1)
    auto issue_ref = [&](int mm, int sel) {
        if (MODE == 3 && wave == 1) {
            const __amdgpu_buffer_rsrc_t rb = rsrc_at((const char*)(p.rowref + mm), (long)(p.M - mm) * 4);
            dma_piece(rb, (unsigned)lane * 16u, smem + G::RING + 2 * G::BIAS_BYTES + sel * G::REF_BYTES);
        }
    };
    // column constants of a tile from that slice: the bias, PAST beyond N
    auto read_cb = [&](int nn, int sel, f32x4 (&cb)[4], int wcols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wc * wcols + j * 16 + 4 * q;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b = *reinterpret_cast<const f32x4*>(smem + G::RING + sel * G::BIAS_BYTES + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) cb[j][e] = nn + c + e < p.N ? b[e] : ((MODE == 0 || MODE == 3) ? PAST : 0.f);
        }
    };

    int vb = blockIdx.x;
    View va, vbv, na, nbv;
    int kind = views_of(vb, va, vbv, m0, n0);              // (a grid never starts on a half tile: those come after a whole round)
    unsigned long long t_start = 0, t_main = 0, t_epi = 0;
    if (STAMPS && tid == 0) t_start = __builtin_amdgcn_s_memtime();
    int bsel = 0;                                          // which of the two bias slices belongs to the current tile
    issue_bias(n0, 0);
    issue_ref(m0, 0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue_stage(va, vbv, s, s);
    // LDS-DMA loads and global stores of one wave do NOT complete in issue order relative to each other (a counted
    // vmcnt wait that relied on the epilogue's stores being "younger" than the next tile's first stages let stages be
    // read before they had landed, about once in 10^5 tiles).  So no counted wait ever has a store between the loads it
    // counts: the stages in flight are drained (vmcnt(0)) right before a tile's first store is issued.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 acc[TI][4];
    int rslot = 0, wslot = NST - 1;

    // one 32-deep step: wait for the stage, read its fragments, multiply, refill the slot read one step ago with
    // stage (rv, rk) -- of this tile or, in the last NST-1 steps, of the next one
    auto step = [&](auto first, auto early, auto halfc, const View& rva, const View& rvb, int rk, int bias_nn, int ref_mm) {
        constexpr int NJ = decltype(halfc)::value ? 2 : 4;          // 16-column blocks per wave
        // A tile's first NST-1 stages landed before the previous tile's epilogue stored anything (see there): no wait.
        // Later steps: all but the NST-2 youngest stages.  Stores may still be outstanding then; they only make this
        // wait stricter (loads complete in order among themselves), never laxer.
        if constexpr (!decltype(early)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        const char* st = smem + rslot * G::STAGE;
        f32x4 cb[4];
        float rr[MODE == 3 ? TI : 1];
        if constexpr (decltype(first)::value) {
            read_cb(n0, bsel, cb, NJ * 16);
            if constexpr (MODE == 3) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    rr[i] = *reinterpret_cast<const float*>(smem + G::RING + 2 * G::BIAS_BYTES + bsel * G::REF_BYTES + (wr * G::WROWS + 16 * i + r16) * 4);
            }
        }
        bf16x8 af[TI], bfr[4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + (NJ == 2 ? fragBh : fragB) + j * 1024);
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + fragA + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        char* wst = smem + wslot * G::STAGE;
        const __amdgpu_buffer_rsrc_t rsa = rsrc_at(rva.base + rk * p.kstepA, rva.rem - (long)rk * p.remdecA);
        const __amdgpu_buffer_rsrc_t rsb = rsrc_at(rvb.base + rk * p.kstepB, rvb.rem - (long)rk * p.remdecB);
        rslot = rslot + 1 == NST ? 0 : rslot + 1;
        wslot = wslot + 1 == NST ? 0 : wslot + 1;
        if (bias_nn >= 0) { issue_bias(bias_nn, bsel ^ 1); issue_ref(ref_mm, bsel ^ 1); }       // ahead of this step's refill pieces: they land before those do
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // operands swapped: a lane then holds C[row = lane & 15][columns 4*(lane>>4) .. +3] of each 16x16 tile
                if constexpr (decltype(first)::value) {
                    f32x4 c0 = cb[j];
                    if constexpr (MODE == 3) c0 -= rr[i];          // the accumulators hold z - (the row's reference) from the start
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], c0, 0, 0, 0);
                } else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            if (i < G::NP) {
                __builtin_amdgcn_sched_barrier(0);
                if (i < G::NJA) dma_piece(rsa, offa[i], wst + (G::NW * i + wave) * 1024);
                else dma_piece(rsb, offb[i - G::NJA], wst + G::A_BYTES + (G::NW * (i - G::NJA) + wave) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;

    const unsigned qbits = (unsigned)(3 - q) << 2;
    // numerators: after the swap a lane holds 8 consecutive columns starting at colq (+ 32 for the second pair of tiles)
    const int colq = (q & 1) ? 16 + 4 * (q - 1) : 4 * q;
    // C as a matrix: row pitch ldc, a lane's second 8 columns 64 bytes on; as a column-panel image [ldc / 32][rows][32]: row
    // pitch 32 elements, the second 8 columns one panel (rows x 64 bytes) on
    const bool c_img = (MODE == 0 || MODE == 3) && p.c_panel_rows > 0;
    const unsigned c_lane_off = (unsigned)(((long)r16 * (c_img ? 32 : p.ldc) + colq) * 2);
    const unsigned c_pair_off = c_img ? (unsigned)(p.c_panel_rows * 64) : 64u;
    for (;;) {
        const int vnext = vb + gridDim.x;
        int mn, nn;
        const int nkind = views_of(vnext, na, nbv, mn, nn);
        unsigned long long t0 = 0;
        if (STAMPS && tid == 0) t0 = __builtin_amdgcn_s_memtime();
        // ---- the tile's k-steps: the first NST-1 find their stages landed, the last NST-1 refill from the next tile
        auto k_steps = [&](auto halfc) {
            step(T{}, T{}, halfc, va, vbv, NST - 1, -1, 0);
#pragma unroll
            for (int kt = 1; kt < NST - 1; ++kt) step(F{}, T{}, halfc, va, vbv, kt + NST - 1, -1, 0);
            for (int kt = NST - 1; kt < nk - (NST - 1); ++kt) step(F{}, F{}, halfc, va, vbv, kt + NST - 1, -1, 0);
#pragma unroll
            for (int s = 0; s < NST - 1; ++s) step(F{}, F{}, halfc, na, nbv, s, s == 0 ? nn : -1, mn);
        };
        const bool half = HALVES && kind == 2;               // workgroup-uniform
        if constexpr (HALVES) {
            if (half) k_steps(T{}); else k_steps(F{});
        } else k_steps(F{});
        const int nj = half ? 2 : 4;                         // 16-column blocks of the wave's part of the tile
        unsigned long long t1 = 0;
        if (STAMPS && tid == 0) { t1 = __builtin_amdgcn_s_memtime(); t_main += t1 - t0; }

        // ---- epilogue on the accumulator registers
        const int mw = m0 + wr * G::WROWS, nw = n0 + wc * (half ? 32 : 64);
        const int slot = nw >> 6;
        // stores: one descriptor per 16-row block (scalar arithmetic), one lane offset for all of them; a lane whose row or
        // columns do not exist gets an offset no descriptor holds, and the store is dropped -- but always issued
        const bool slot_ok = slot < p.np;
        f32x4 res_buf[MODE == 2 ? 2 : 1][4];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int mb = mw + i * 16;                      // first row of the block
            const bool row_ok = r16 < p.M - mb;
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = (!HALVES || j < nj) ? acc[i][j][e] : 0.f;
            if constexpr (MODE == 2) {
                // residual + dropout(z + bias) -> f32: the lane's 4 consecutive columns of each 16-column tile are one
                // 16-byte load and one 16-byte store
                const __amdgpu_buffer_rsrc_t ors = rsrc_at((const char*)(p.Cf + (long)mb * p.ldc + nw), 0xFFFFFFF0L);
                // the residual rows of a block are requested one block ahead of their use (deeper made no difference)
                auto load_res = [&](int ii, f32x4 (&dst)[4]) {
                    const int mbb = mw + ii * 16;
                    const __amdgpu_buffer_rsrc_t rr = rsrc_at((const char*)(p.residual + (long)mbb * p.ldr + nw), 0xFFFFFFF0L);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = r16 < p.M - mbb && 16 * j + 4 * q < p.N - nw && j < nj;
                        const unsigned off = ok ? (unsigned)(((long)r16 * p.ldr + 16 * j + 4 * q) * 4) : 0xFFFFFFFFu;
                        dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0));
                    }
                };
                if (i == 0) load_res(0, res_buf[0]);
                if (i + 1 < TI) load_res(i + 1, res_buf[(i + 1) & 1]);
                f32x4 (&rv)[4] = res_buf[i & 1];
                if (p.drop_thr) {
                    const unsigned base = (unsigned)((long)(mb + r16) * p.N + nw + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool keep[4];
                        dropout_keep4(p.drop_k0, p.drop_k1, base + 16 * j, p.drop_thr, keep);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * j + e] = keep[e] ? v[4 * j + e] * p.drop_scale : 0.f;
                    }
                }
                if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's first stages have landed: see the prologue
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = row_ok && 16 * j + 4 * q < p.N - nw && j < nj;
                    const unsigned off = ok ? (unsigned)(((long)r16 * p.ldc + 16 * j + 4 * q) * 4) : 0xFFFFFFFFu;
                    const f32x4 o = {v[4 * j] + rv[j][0], v[4 * j + 1] + rv[j][1], v[4 * j + 2] + rv[j][2], v[4 * j + 3] + rv[j][3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ors, off, 0, 0);
                }
            } else if constexpr (MODE == 1) {
                // standard epilogue: column scale (uniform per 64-column wave tile), ReLU, dropout on the element index m * N + n
                const float sc = nw < p.col_scale_n ? p.col_scale : 1.f;
                const float fl = p.relu ? 0.f : -INFINITY;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = fmaxf(v[t] * sc, fl);
                if (p.drop_thr) {
                    const unsigned base = (unsigned)((long)(mb + r16) * p.N + nw + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool keep[4];
                        dropout_keep4(p.drop_k0, p.drop_k1, base + 16 * j, p.drop_thr, keep);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * j + e] = keep[e] ? v[4 * j + e] * p.drop_scale : 0.f;
                    }
                }
                if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's first stages have landed: see the prologue
            } else {
                // maximum with the column riding in the low mantissa bits: column 16j + 4q + e, stored complemented so
                // that among (nearly) equal positive values the lowest column wins
                float pm = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const unsigned c = (unsigned)(((3 - (t >> 2)) << 4) | (3 - (t & 3)));
                    const float f = __uint_as_float((__float_as_uint(v[t]) & 0xFFFFFFC0u) | c);
                    pm = t == 0 ? f : fmaxf(pm, f);
                }
                bool neg_rule = false;
                if constexpr (MODE == 0) {
                    // A negative float grows more negative with its low bits, so under the packing above the HIGHEST
                    // column would win among (nearly) equal negative values.  Where a row's slot maximum is negative
                    // (every one of its 64 logits is: rare, so behind a wave-uniform branch) the lane's candidates are
                    // packed again with the index as it is, and the lowest column wins as it does for positive values.
                    neg_rule = quad_max(pm) < 0.f;
                    if (__builtin_amdgcn_ballot_w64(neg_rule) != 0) {
                        float pn = 0.f;
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const unsigned c = (unsigned)(((t >> 2) << 4) | (t & 3));
                            const float f = __uint_as_float((__float_as_uint(v[t]) & 0xFFFFFFC0u) | c);
                            pn = t == 0 ? f : fmaxf(pn, f);
                        }
                        if (neg_rule) pm = pn;
                    }
                }
                pm = quad_max(__uint_as_float(__float_as_uint(pm) | (neg_rule ? (unsigned)q << 2 : qbits)));
                // mode 3: the values are z - (row reference) already, and the numerators and their sum are taken against
                // that reference, not the slot's maximum; else a slot wholly past N: numerators exp2(-inf) = 0
                const float nref = MODE == 3 ? 0.f : (pm > -1.0e38f ? -pm * LOG2E : 0.f);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) { v[t] = __builtin_amdgcn_exp2f(fmaf(v[t], LOG2E, nref)); sum += v[t]; }
                sum = quad_sum(sum);
                if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's first stages have landed: see the prologue
                if constexpr (MODE == 3) {
                    // 8-byte records {packed maximum, sum}: the column inside the slot rides in the maximum's low bits
                    const u32x2_t rec = {__float_as_uint(pm), __float_as_uint(sum)};
                    const __amdgpu_buffer_rsrc_t prs = rsrc_at((const char*)p.part4 + ((long)slot * p.M + mb) * 8, 0xFFFFFFF0L);
                    const unsigned off = (q == 0 && row_ok && slot_ok) ? (unsigned)(r16 * 8) : 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b64(rec, prs, off, 0, 0);
                } else {
                    const int low6 = (int)(__float_as_uint(pm) & 63u);
                    const int arg = nw + (neg_rule ? low6 : 63 - low6);
                    const f32x4 rec = {pm, sum, -INFINITY, __int_as_float(arg)};
                    const __amdgpu_buffer_rsrc_t prs = rsrc_at((const char*)(p.part4 + ((long)slot * p.M + mb)), 0xFFFFFFF0L);
                    const unsigned off = (q == 0 && row_ok && slot_ok) ? (unsigned)(r16 * 16) : 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rec), prs, off, 0, 0);
                }
            }
            if constexpr (KEEP && MODE != 2) {
                const __amdgpu_buffer_rsrc_t crs = rsrc_at((const char*)(c_img ? p.C + ((long)(nw >> 5) * p.c_panel_rows + mb) * 32
                                                                                       : p.C + (long)mb * p.ldc + nw), 0xFFFFFFF0L);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    // tiles 2jp and 2jp+1: this lane's 4 + 4 columns become 8 consecutive ones of one of them
                    const unsigned x0 = pack_bf16(v[8 * jp + 0], v[8 * jp + 1]), x1 = pack_bf16(v[8 * jp + 2], v[8 * jp + 3]);
                    const unsigned y0 = pack_bf16(v[8 * jp + 4], v[8 * jp + 5]), y1 = pack_bf16(v[8 * jp + 6], v[8 * jp + 7]);
                    const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    const bool ok = row_ok && colq + 32 * jp < p.store_cols - nw && 2 * jp < nj;
                    const unsigned off = ok ? c_lane_off + c_pair_off * jp : 0xFFFFFFFFu;
                    // the numerators (0.8 GB that nothing reads before the backward products) leave as non-temporal
                    // stores: with the operands read as k-panel images that is 1.5 % faster inside the training step
                    // (with row-major operands, whose loads asked for twice the lines, it was 5-10 % slower there);
                    // the outputs of the standard epilogue, which the next kernel reads, stay write-back
                    if ((MODE == 0 || MODE == 3) && p.store_nt) __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, crs, off, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, crs, off, 0, 0);
                }
            }
        }
        if (STAMPS && tid == 0) t_epi += __builtin_amdgcn_s_memtime() - t1;
        if (nkind == 0) break;
        vb = vnext; kind = nkind;
        va = na; vbv = nbv; m0 = mn; n0 = nn;
        bsel ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the loads issued past the last tile must not outlive the workgroup's LDS
    if (STAMPS && tid == 0) {
        unsigned long long* s = p.stamps + (long)blockIdx.x * 8;
        s[0] = t_start; s[1] = t_main; s[2] = t_epi; s[7] = __builtin_amdgcn_s_memtime();
        s[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

// Diagnostic (not part of the declared ABI): event brackets around the kernel alone, for a benchmark that wants this
// launch's duration without the finalize pass that follows it in the same entry point.
constexpr int TIMING_PAIRS = 512;
struct LaunchTiming { bool on = false; int n = 0; hipEvent_t ev[2 * TIMING_PAIRS] = {}; bool made = false; } g_timing;

// diagnostic: k-panel images handed in beside the entry points' row-major operands (used when the sizes agree)
int g_store_nt = 1;      // numerator stores non-temporal (diagnostic hook: 0 = write-back)
int g_half_tiles = 1;    // the last round's tiles as halves (diagnostic hook: 0 = whole tiles only)
struct PackedOverride { const bf16_t* Ap = nullptr; long a_rows = 0; const bf16_t* Bp = nullptr; long b_rows = 0; } g_packed;

template <int WM, int WN, int TI, int NST>
int launch_geo(const HeadFwdArgs& a_in, hipStream_t st) {
    using G = Geo<WM, WN, TI, NST>;
    HeadFwdArgs a = a_in;
    if (!a.Ap && g_packed.Ap && g_packed.a_rows >= a.M) { a.Ap = g_packed.Ap; a.a_rows = g_packed.a_rows; }
    if (!a.Bp && g_packed.Bp && g_packed.b_rows >= a.N) { a.Bp = g_packed.Bp; a.b_rows = g_packed.b_rows; }
    a.store_nt = g_store_nt;
    a.kstepA = a.kstepB = DBK * 2; a.remdecA = a.remdecB = DBK * 2;
    if (a.Ap) { a.A = a.Ap; a.lda = DBK; a.kstepA = a.a_rows * (DBK * 2); a.remdecA = 0; }
    if (a.Bp) { a.B = a.Bp; a.ldb = DBK; a.kstepB = a.b_rows * (DBK * 2); a.remdecB = 0; }
    const long tiles = (long)((a.M + G::BMV - 1) / G::BMV) * ((a.N + G::BNV - 1) / G::BNV);
    if (tiles <= 0) return CAPDEC_OK;
    long grid = 256L * G::WG_PER_CU;                      // every workgroup resident at once; a multiple of the 8 XCDs
    if (tiles < grid) grid = tiles;
    // a last round of at most half the workgroups: its tiles go out as halves, one per workgroup (see the kernel)
    a.half_tiles = 0;
    if ((a.mode == 1 || a.mode == 2) && WM == 2 && WN == 4 && grid == 256 && g_half_tiles && tiles % 256 != 0 && tiles % 256 <= 128)
        a.half_tiles = (int)(tiles % 256);
    auto go = [&](auto kfn) {
        ensure_dynamic_lds((const void*)kfn, 160 * 1024);      // once per (kernel, device)
        const bool timing = g_timing.on && g_timing.made && g_timing.n < TIMING_PAIRS && (a.mode == 0 || a.mode == 3) && a.C;      // the training forward only
        if (timing) (void)hipEventRecord(g_timing.ev[2 * g_timing.n], st);
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(64 * G::NW), G::LDS, st, a);
        if (timing) { (void)hipEventRecord(g_timing.ev[2 * g_timing.n + 1], st); ++g_timing.n; }
    };
    if (a.mode == 3) go(head_fwd_kernel<WM, WN, TI, NST, true, false, 3>);
    else if (a.mode == 2) go(head_fwd_kernel<WM, WN, TI, NST, false, false, 2>);
    else if (a.mode == 1) go(head_fwd_kernel<WM, WN, TI, NST, true, false, 1>);
    else if (a.stamps) {                                  // diagnostic build of the same kernel: clock stamps per workgroup
        if (a.C) go(head_fwd_kernel<WM, WN, TI, NST, true, true>);
        else go(head_fwd_kernel<WM, WN, TI, NST, false, true>);
    } else if (a.C) go(head_fwd_kernel<WM, WN, TI, NST, true, false>);
    else go(head_fwd_kernel<WM, WN, TI, NST, false, false>);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

}  // namespace

extern "C" void capdecdbg_head_fwd_store_nt(int on) { g_store_nt = on; }
extern "C" void capdecdbg_head_fwd_half_tiles(int on) { g_half_tiles = on; }
extern "C" void capdecdbg_head_fwd_packed(const void* Ap, long a_rows, const void* Bp, long b_rows) {
    g_packed = PackedOverride{(const bf16_t*)Ap, a_rows, (const bf16_t*)Bp, b_rows};
}
// on != 0: start collecting (up to 512 launches on the current device); on == 0: stop
extern "C" void capdecdbg_head_fwd_timing(int on) {
    if (on && !g_timing.made) {
        bool ok = true;
        for (int i = 0; i < 2 * TIMING_PAIRS && ok; ++i) ok = hipEventCreate(&g_timing.ev[i]) == hipSuccess;
        g_timing.made = ok;
    }
    g_timing.on = on != 0;
    if (on) g_timing.n = 0;
}
// sum of the collected durations in ms (call after synchronising the stream); returns the number of launches
extern "C" int capdecdbg_head_fwd_timing_read(double* total_ms) {
    double t = 0.0;
    for (int i = 0; i < g_timing.n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_timing.ev[2 * i], g_timing.ev[2 * i + 1]) == hipSuccess) t += ms;
    }
    if (total_ms) *total_ms = t;
    return g_timing.n;
}

bool head_fwd_supported(const HeadFwdArgs& a) {
    if (a.mode == 3 && (!a.C || !a.rowref || !aligned16(a.rowref) || !a.part4)) return false;
    if (a.mode == 1 && (!a.C || (a.col_scale_n % 64) != 0 || (a.N % 8) != 0 || a.store_cols != a.N)) return false;
    if (a.mode == 2 && (a.C || !a.Cf || !a.residual || (a.N % 4) != 0 || (a.ldc % 4) != 0 || (a.ldr % 4) != 0 || !aligned16(a.Cf) ||
                        !aligned16(a.residual) || 256L * a.ldc * 4 >= 0x7FFFFFFFL || 256L * a.ldr * 4 >= 0x7FFFFFFFL)) return false;
    if (a.c_panel_rows && ((a.mode != 0 && a.mode != 3) || !a.C || a.c_panel_rows < a.M || (a.ldc % 32) != 0 || a.store_cols != a.ldc ||
                           a.c_panel_rows * 64 >= 0x7FFFFFFFL)) return false;
    return a.K >= 8 * DBK && (a.K % DBK) == 0 && (a.lda % 8) == 0 && (a.ldb % 8) == 0 && aligned16(a.A) && aligned16(a.B) &&
           (!a.C || ((a.ldc % 8) == 0 && (a.store_cols % 8) == 0 && aligned16(a.C))) &&
           (long)a.np * a.M * 16 < 0xFFFFFFF0L && 256L * a.lda * 2 < 0x7FFFFFFFL && 256L * a.ldb * 2 < 0x7FFFFFFFL &&
           (!a.C || 256L * a.ldc * 2 < 0x7FFFFFFFL);
}

int head_fwd_launch(const HeadFwdArgs& a, int geometry, hipStream_t st) {
    if (geometry == 1) return launch_geo<2, 4, 8, 4>(a, st);
    return launch_geo<2, 2, 8, 3>(a, st);
}
