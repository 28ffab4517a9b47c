// Persistent pipelined bf16 product with the epilogue on the accumulator registers (gfx950):
//
//   C = epilogue(A B^T + bias),  A [M, K] row-major (or its k-panel image),
//                                B [N, K] row-major (or its k-panel image), or reduction-major [K, N] (b_tr: the
//                                weight of a data gradient dX = dY W, read along its output index)
//
// for the large plain products of a stack of blocks: the data gradients of decoder.py:189 through in_proj, out_proj and
// linear1 (decoder.py:54,59) at full batch, which it runs 15-20 % faster than the tile kernels; the forward form (row-major
// weight, bias / scale / ReLU / dropout) is measured level with the eight-wave persistent kernel and is not routed to it.  It joins what two older kernels each have one half of:
//   * the MAIN LOOP of the pipelined 256x256 kernel of gemm.hip: 4 waves, each a 128x128 block of the tile with its 256
//     accumulators in the accumulation registers; fragments double-buffered in registers, the LDS reads and the ring's
//     refill instructions of step k+1 spread between the 64 MFMAs of step k.  A 128x128 wave tile reads 16 KiB of
//     fragments per 64 MFMAs -- 96 KiB per 1024 MFMA cycles of the CU with the refills, against the 128 KiB (the LDS's
//     whole 128 B/clk) that eight waves of 128x64 move;
//   * the PERSISTENT structure and register epilogue of head_fwd.hip: 256 workgroups walk the tiles, the 4-stage operand
//     ring never stops (the first stages of tile t+1 are fetched during the last steps of tile t), and the epilogue reads
//     the accumulators as they lie -- a lane holds 4 consecutive columns of one row per 16x16 block -- so no LDS staging
//     tile competes with the ring for the 160 KiB.
// Rules the loop keeps (each measured or found the hard way in those two kernels):
//   * LDS-DMA loads and global stores of one wave complete in no fixed order relative to each other: no counted vmcnt
//     wait has a store between the loads it counts.  The stages in flight are drained (vmcnt(0)) before a tile's first
//     store; the next tile's first NST-1 steps need no wait; later waits only get stricter by outstanding stores.
//   * inline-assembly loads and MFMAs are complete, to the compiler, when issued: every fragment register of an
//     assembly read is used or named only after an lgkmcnt wait, and the accumulators are read after the MFMA pipeline has
//     drained (s_nop).  tools/check_asm_loads.py walks the ISA for the first.
// Epilogue (mode 1 of HeadFwdArgs): (z + bias) * column scale, ReLU, dropout -> bf16; a plain data gradient uses none of it.
#include <type_traits>

#include "head_fwd.h"

namespace {

constexpr int PBK = 32;                          // k-depth of a ring stage
constexpr int PNST = 4, PSTAGE = 32768, PA_BYTES = 16384, PNP = 8;      // stages, bytes per stage, A's share, DMA pieces per wave and stage
constexpr int PRING = PNST * PSTAGE;
constexpr int PLDS = PRING + 2 * 1024;           // + the current and the next tile's bias slices (256 f32 each)

// row-major [rows][32 k] stage image, 64-B rows; chunk (0..3) XOR 3*((row>>3)&1): conflict-free ds_read_b128
__device__ __forceinline__ int plds_row(int row, int kc) { return row * 64 + ((kc ^ (((row >> 3) & 1) * 3)) << 4); }
// reduction-major [32 k][128 cols] sub-image, 256-B rows; 32-B column blocks XORed with a 3-bit function of k
__device__ __forceinline__ int pkswz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int plds_kmajor(int krow, int chunk16) { return krow * 256 + ((chunk16 ^ (pkswz(krow) << 1)) << 4); }
__device__ __forceinline__ int ptr_lane_offset(int rb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = rb + 4 * pp;
    return plds_kmajor(8 * g + q, col >> 3) + ((col & 7) << 1);
}
// Transposing fragment read as assembly: through the builtin the compiler drains every outstanding LDS-DMA first.
// Complete, to the compiler, when issued: the caller waits (lgkmcnt) before anything touches the result.
__device__ __forceinline__ bf16x8 pread_tr(unsigned addr) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(addr));
    return __builtin_bit_cast(bf16x8, (u32x4{lo[0], lo[1], hi[0], hi[1]}));
}
__device__ __forceinline__ unsigned plds_addr(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(const char, p); }

// acc (+)= a x b with the accumulator in the accumulation registers, tied in place (left to the register allocator, 256
// accumulators get three-address MFMAs fed through copies).  No hazard padding comes with assembly: dependent uses stay away.
__device__ __forceinline__ void pmfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(__builtin_bit_cast(u32x4, a)), "v"(__builtin_bit_cast(u32x4, b)));
}
// a tile's first step: C = 0, so the accumulators need no clearing pass
__device__ __forceinline__ void pmfma_first(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(__builtin_bit_cast(u32x4, a)), "v"(__builtin_bit_cast(u32x4, b)));
}
// s_waitcnt immediate (gfx9 layout: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]); expcnt is left alone
constexpr int pwaitcnt(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (lgkm << 8); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t prsrc(const char* base, long rem) {
    const unsigned rec = rem <= 0 ? 0u : (rem > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)rem);
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)rec, 0x00020000);
}
__device__ __forceinline__ void pdma(__amdgpu_buffer_rsrc_t rs, unsigned off, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, off, 0, 0, 0);
}

// tile order: each XCD (workgroups b, b + 8, ... share one) gets a contiguous run of logical ids; wide outputs walk groups
// of 8 M-tiles x all N-tiles, M fastest; narrow ones (at most 16 N-tiles) walk N fastest, so that every resident M-tile
// has all its N-tiles resident and the large streamed operand leaves HBM once
__device__ __forceinline__ void pmap_tile(int vb, int tm, int tn, int& tile_m, int& tile_n) {
    const int nwg = tm * tn;
    const int qq = nwg >> 3, r = nwg & 7, xcd = vb & 7;
    const int logical = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (vb >> 3);
    const int group = tn <= 16 ? 1 : 8;
    const int per_group = group * tn;
    const int grp = logical / per_group;
    const int first_m = grp * group;
    const int gsz = min(tm - first_m, group);
    const int in = logical - grp * per_group;
    tile_m = first_m + in % gsz;
    tile_n = in / gsz;
}

typedef __attribute__((ext_vector_type(2))) __bf16 pbf16x2;
__device__ __forceinline__ unsigned ppack(float a, float b) {
    pbf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}

struct PTile { const char* a; long arem; const char* b; long brem; int m0, n0; };

// f(integral_constant<0>) ... f(integral_constant<N-1>): a loop whose index is a compile-time constant in every pass, so that
// arrays indexed by it stay in registers whatever the unroller thinks of the body's size (an accumulator array indexed by a
// run-time value goes to scratch memory, and every MFMA with it)
template <int I, int N, typename F>
__device__ __forceinline__ void pstatic_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pstatic_for<I + 1, N>(f);
    }
}

template <bool B_TR>
__global__ __launch_bounds__(256, 1) void ppipe_kernel(HeadFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
    const int ntiles = tm * tn;
    const int nk = p.K / PBK;                     // even, >= 2 * PNST (checked on the host)

    // what a lane fetches for its pieces of a stage: tile-independent (b_tr: N is a multiple of 256, so no column of a tile lies past it)
    unsigned offa[4], offb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * (4 * j + wave) + (lane >> 2), kc = (lane & 3) ^ (((row >> 3) & 1) * 3);
        offa[j] = (unsigned)(((long)row * p.lda + kc * 8) * 2);
        if constexpr (!B_TR) {
            offb[j] = (unsigned)(((long)row * p.ldb + kc * 8) * 2);
        } else {
            const int piece = 4 * j + wave, sub = piece >> 3, pp = piece & 7;
            const int krow = 4 * pp + (lane >> 4), pc = lane & 15;
            const int col = sub * 128 + (pc ^ (pkswz(krow) << 1)) * 8;
            offb[j] = (unsigned)(((long)krow * p.ldb + col) * 2);
        }
    }
    // fragment addresses inside a stage: A rows wr*128 + 16 i + r16, B columns wc*128 + 16 j + r16 (+ 1024 per 16 rows / columns)
    const int fragA = plds_row(wr * 128 + r16, q);
    const int fragB = PA_BYTES + (B_TR ? wc * 8192 + ptr_lane_offset(0, lane) : plds_row(wc * 128 + r16, q));
    // (reduction-major: the 16 columns of fragment j lie 32 bytes further on per fragment inside a k-row, across an XOR of
    // 32-byte blocks that depends on the row only -- so the per-j offsets are computed once)
    int troffB[B_TR ? 8 : 1];
    if constexpr (B_TR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) troffB[j] = PA_BYTES + wc * 8192 + ptr_lane_offset(j * 16, lane);
    }

    auto tile_of = [&](int vb) {
        PTile t;
        if (vb < ntiles) {
            int tile_m, tile_n;
            pmap_tile(vb, tm, tn, tile_m, tile_n);
            t.m0 = tile_m * 256; t.n0 = tile_n * 256;
            t.a = (const char*)(p.A + (long)t.m0 * p.lda); t.arem = (long)(p.M - t.m0) * p.lda * 2;
            if constexpr (B_TR) { t.b = (const char*)(p.B + t.n0); t.brem = ((long)p.K * p.ldb - t.n0) * 2; }
            else { t.b = (const char*)(p.B + (long)t.n0 * p.ldb); t.brem = (long)(p.N - t.n0) * p.ldb * 2; }
        } else {                                  // past the last tile: loads that fetch nothing but still count
            t.m0 = t.n0 = 0;
            t.a = (const char*)p.A; t.arem = 0;
            t.b = (const char*)p.B; t.brem = 0;
        }
        return t;
    };
    auto issue_stage = [&](const char* ra, long rema, const char* rb, long remb, int slot) {
        char* st = smem + slot * PSTAGE;
        const __amdgpu_buffer_rsrc_t rsa = prsrc(ra, rema), rsb = prsrc(rb, remb);
#pragma unroll
        for (int j = 0; j < 4; ++j) pdma(rsa, offa[j], st + (4 * j + wave) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) pdma(rsb, offb[j], st + PA_BYTES + (4 * j + wave) * 1024);
    };
    // a tile's slice of the bias -> LDS, one DMA piece issued by wave 0 (columns past N arrive as zeros); read by every wave in
    // the tile's epilogue, many barriers later.  Issued ahead of a step's refill pieces, so the counted waits, which let the
    // youngest pieces stay in flight, only get stricter by it.
    auto issue_bias = [&](int n0, int sel) {
        if (p.bias && wave == 0)
            pdma(prsrc((const char*)(p.bias + n0), (long)(p.N - n0) * 4), (unsigned)lane * 16u, smem + PRING + sel * 1024);
    };
    auto read_a = [&](const char* st, int i) { return *reinterpret_cast<const bf16x8*>(st + fragA + i * 1024); };
    auto read_b = [&](const char* st, int j) {
        if constexpr (B_TR) return pread_tr(plds_addr(st) + (unsigned)troffB[j]);
        else return *reinterpret_cast<const bf16x8*>(st + fragB + j * 1024);
    };

    int vb = blockIdx.x;
    PTile cur = tile_of(vb);
    // refill stream: the stage that the next refill fetches starts at (refA, refB); its addressable bytes remA / remB
    const char* refA = cur.a; const char* refB = cur.b;
    long remA = cur.arem, remB = cur.brem;
    int bsel = 0;                                // which of the two bias slices belongs to the current tile
    issue_bias(cur.n0, 0);
#pragma unroll
    for (int s = 0; s < PNST; ++s) {
        issue_stage(refA, remA, refB, remB, s);
        refA += p.kstepA; remA -= p.remdecA; refB += p.kstepB; remB -= p.remdecB;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc[8][8];
    bf16x8 fa[8], fb[2][8], fa7;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = read_a(smem, i);
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = read_b(smem, j);
    int slot = 0;                                // ring slot of the stage whose fragments are in registers

    // One step: MFMAs on the fragments of stage k (B: cbf; last A: ca7, the others fa[0..6]) while those of stage k+1 are
    // read (B: nb; last A: na7) and the slot stage k occupied is refilled with the stage PNST further on.
    auto step = [&](auto first, auto early, bf16x8 (&cbf)[8], bf16x8 (&nb)[8], bf16x8& ca7, bf16x8& na7) {
        // the wave's fragment reads of this stage are in; its share of the next stage has landed (the first PNST-1 steps of a
        // tile: landed before the previous tile's epilogue stored anything); after the barrier that holds for every wave
        if constexpr (decltype(early)::value) __builtin_amdgcn_s_waitcnt(pwaitcnt(63, 0));
        else __builtin_amdgcn_s_waitcnt(pwaitcnt((PNST - 2) * PNP, 0));
        __builtin_amdgcn_s_barrier();
        const int nslot = slot + 1 == PNST ? 0 : slot + 1;
        const char* sn = smem + nslot * PSTAGE;
        char* wst = smem + slot * PSTAGE;
        slot = nslot;
        __amdgpu_buffer_rsrc_t rsa, rsb;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(first)::value) pmfma_first(acc[i][j], cbf[j], i < 7 ? fa[i] : ca7);
                else pmfma(acc[i][j], cbf[j], i < 7 ? fa[i] : ca7);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) nb[i] = read_b(sn, i);
                if (j == 3) {
                    if (i > 0) fa[i - 1] = read_a(sn, i - 1);
                    else { rsa = prsrc(refA, remA); refA += p.kstepA; remA -= p.remdecA; }
                }
                if (j == 4 && i == 0) { rsb = prsrc(refB, remB); refB += p.kstepB; remB -= p.remdecB; }
                if (j == 5 && i == 7) na7 = read_a(sn, 7);
                if (j == 6) {
                    if (i < 4) pdma(rsa, offa[i], wst + (4 * i + wave) * 1024);
                    else pdma(rsb, offb[i - 4], wst + PA_BYTES + (4 * (i - 4) + wave) * 1024);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using T = std::true_type;
    using F = std::false_type;

    const int colq = (q & 1) ? 16 + 4 * (q - 1) : 4 * q;      // bf16 output: after the swap a lane holds 8 consecutive columns from here
    for (;;) {
        const int vnext = vb + gridDim.x;
        const PTile nxt = tile_of(vnext);
        // ---- the tile's k-steps
        step(T{}, T{}, fb[0], fb[1], fa[7], fa7);
        step(F{}, T{}, fb[1], fb[0], fa7, fa[7]);
        step(F{}, T{}, fb[0], fb[1], fa[7], fa7);
        step(F{}, F{}, fb[1], fb[0], fa7, fa[7]);
        for (int kt = 4; kt < nk; kt += 2) {
            // from step nk - PNST on the refills fetch the next tile's first stages
            if (kt == nk - PNST) { refA = nxt.a; remA = nxt.arem; refB = nxt.b; remB = nxt.brem; issue_bias(nxt.n0, bsel ^ 1); }
            step(F{}, F{}, fb[0], fb[1], fa[7], fa7);
            step(F{}, F{}, fb[1], fb[0], fa7, fa[7]);
        }
        // ---- epilogue on the accumulator registers.  The last MFMAs have written back; the next tile's stages in flight have
        // landed before the first store is issued (see the rules at the top).  The wave walks its 16 blocks of 16 rows x 64
        // columns (b = 8 g + i: column group g, row block i).  It only stores: an epilogue that also READ from memory per block
        // (a residual stream, an activation gate) was measured on this structure and lost to the eight-wave kernels -- with
        // loads and stores both outstanding the compiler can only wait for everything (vmcnt(0)) before each use of a loaded
        // value, and with one wave per SIMD nothing covers that round trip.
        __builtin_amdgcn_s_waitcnt(pwaitcnt(63, 0));            // (the look-ahead reads have landed: their registers may be reused)
        const int mw = cur.m0 + wr * 128;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pstatic_for<0, 16>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int g = b >> 3, i = b & 7;
            const int nw = cur.n0 + wc * 128 + g * 64;
            const int mb = mw + i * 16;
            const bool row_ok = r16 < p.M - mb;
            // column constants of the 64-column group: the tile's bias slice in LDS (zeros past N)
            f32x4 cb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias) cb[j] = *reinterpret_cast<const f32x4*>(smem + PRING + bsel * 1024 + (wc * 128 + g * 64 + 16 * j + 4 * q) * 4);
            }
            float v[16];
            const float sc = nw < p.col_scale_n ? p.col_scale : 1.f;
            const float fl = p.relu ? 0.f : -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = fmaxf((acc[i][4 * g + j][e] + cb[j][e]) * sc, fl);
            if (p.drop_thr) {
                const unsigned base = (unsigned)((long)(mb + r16) * p.N + nw + 4 * q);      // the element index of the dropout mask
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool keep[4];
                    dropout_keep4(p.drop_k0, p.drop_k1, base + 16 * j, p.drop_thr, keep);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = keep[e] ? v[4 * j + e] * p.drop_scale : 0.f;
                }
            }
            // bf16: tiles 2jp and 2jp+1 -- this lane's 4 + 4 columns become 8 consecutive ones of one of them, one 16-byte store
            const __amdgpu_buffer_rsrc_t crs = prsrc((const char*)(p.C + (long)mb * p.ldc + nw), 0xFFFFFFF0L);
            const unsigned lane_off = (unsigned)(((long)r16 * p.ldc + colq) * 2);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const unsigned x0 = ppack(v[8 * jp + 0], v[8 * jp + 1]), x1 = ppack(v[8 * jp + 2], v[8 * jp + 3]);
                const unsigned y0 = ppack(v[8 * jp + 4], v[8 * jp + 5]), y1 = ppack(v[8 * jp + 6], v[8 * jp + 7]);
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const bool ok = row_ok && colq + 32 * jp < p.N - nw;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, crs, ok ? lane_off + 64u * jp : 0xFFFFFFFFu, 0, 0);
            }
        });
        if (vnext >= ntiles) break;
        vb = vnext;
        cur = nxt;
        bsel ^= 1;
        // the next tile's stage 0 again (its fragments were read during the last step, and given up for the epilogue's registers)
        {
            const char* st0 = smem + slot * PSTAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = read_a(st0, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[0][j] = read_b(st0, j);
        }
    }
    // the refills issued past the last tile must not outlive the workgroup's LDS; the last look-ahead reads feed nothing, but
    // their registers stay named until the data is in
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa[i])));
    asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa7)));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb[0][j])));
        asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb[1][j])));
    }
}

}  // namespace

bool ppipe_supported(const HeadFwdArgs& a) {
    if (a.mode != 1 || !a.C) return false;
    if (a.Ap || a.c_panel_rows || a.stamps) return false;
    if (a.K < 2 * PNST * PBK || (a.K % (2 * PBK)) != 0) return false;
    if ((a.lda % 8) != 0 || (a.ldb % 8) != 0 || !aligned16(a.A) || !aligned16(a.B)) return false;
    if (a.bias && (!aligned16(a.bias) || (a.N % 4) != 0)) return false;
    if (256L * a.lda * 2 >= 0x7FFFFFFFL || (!a.b_tr && 256L * a.ldb * 2 >= 0x7FFFFFFFL) || (a.b_tr && 32L * a.ldb * 2 >= 0x7FFFFFFFL)) return false;
    if (a.b_tr && ((a.N % 256) != 0 || a.Bp)) return false;
    return (a.col_scale_n % 64) == 0 && (a.N % 8) == 0 && (a.ldc % 8) == 0 && aligned16(a.C) && 256L * a.ldc * 2 < 0x7FFFFFFFL;
}

int ppipe_launch(const HeadFwdArgs& a_in, hipStream_t st) {
    HeadFwdArgs a = a_in;
    if (a.b_tr) {
        a.kstepA = PBK * 2; a.remdecA = PBK * 2;
        a.kstepB = (long)PBK * a.ldb * 2; a.remdecB = (int)(PBK * a.ldb * 2);
    } else {
        a.kstepA = a.kstepB = PBK * 2; a.remdecA = a.remdecB = PBK * 2;
        if (a.Bp) { a.B = a.Bp; a.ldb = PBK; a.kstepB = a.b_rows * (PBK * 2); a.remdecB = 0; }
    }
    const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    if (tiles <= 0) return CAPDEC_OK;
    const long grid = tiles < 256 ? tiles : 256;              // every workgroup resident at once; a multiple of the 8 XCDs when full
    auto go = [&](auto kfn) {
        ensure_dynamic_lds((const void*)kfn, 160 * 1024);
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(256), PLDS, st, a);
    };
    if (a.b_tr) go(ppipe_kernel<true>); else go(ppipe_kernel<false>);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
