// HBM-bound row kernels: prefix join + LayerNorm, LayerNorm backward, casts,
// dropout backward, bias-gradient column sums, token-embedding rows, Adam.
// One wave per row for the LayerNorm family (a 768-wide f32 row is 3 KiB =
// 3 x 16 B per lane), 16-B vector accesses everywhere.
#include <stdarg.h>
#include <type_traits>
#include "common.h"

// ------------------------------------------------------------------ error text
static thread_local char g_err[512] = "";
void capdec_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* capdec_last_error(void) { return g_err; }
extern "C" int capdec_abi_version(void) { return CAPDEC_ABI_VERSION; }

namespace {

constexpr int LN_MAX_D = 2048;        // 8 x float4 per lane

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// the same for data that is read once per step (the join's sources): non-temporal, so that it does not
// displace what the following kernels re-read from the caches
template <typename T>
__device__ __forceinline__ f32x4 load4_stream(const T* p);
template <>
__device__ __forceinline__ f32x4 load4_stream<float>(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
template <>
__device__ __forceinline__ f32x4 load4_stream<bf16_t>(const bf16_t* p) {
    const bf16x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ------------------------------------------------------------------ LayerNorm forward (+ prefix join)
// src row for output row r = s*B + b: patches, then the sos slot, then the caption
// NV: float4 groups per lane (4 covers rows of up to 1024 columns, 8 up to 2048)
template <typename TIN, typename TOUT, bool CONCAT, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIN* img, long img_bs, long img_ss, const TIN* sos, const TIN* cap, const float* xin,
                                                      const float* pos, const float* gamma, const float* beta, float eps,
                                                      long M, int B, int P, int L, int d,
                                                      float* xout, long x_row0, TOUT* y, float* mean, float* rstd) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    // a wave walks rows r, r + nwaves, ...: long-lived waves keep more bytes in flight than one wave per row, and the
    // next row's loads are issued before the current row's two reductions, so a wave always has a row on its way
    auto fetch = [&](long r, f32x4 (&v)[NV]) {
        long s = 0;
        const TIN* src = nullptr;
        if constexpr (CONCAT) {
            s = r / B;
            const long b = r - s * B;
            src = s < P ? img + b * img_bs + s * img_ss : (s == P ? sos + b * d : cap + (b * L + (s - P - 1)) * d);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < d) {
                if constexpr (CONCAT) {
                    v[i] = load4_stream<TIN>(src + c);
                    if (pos) v[i] += *reinterpret_cast<const f32x4*>(pos + s * d + c);
                } else {
                    v[i] = *reinterpret_cast<const f32x4*>(xin + r * d + c);
                }
            }
        }
    };
    long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[NV], vn[NV];
    if (r < M) fetch(r, v);
    for (; r < M; r += nwaves) {
        if (r + nwaves < M) fetch(r + nwaves, vn);
        float* xo = r < x_row0 ? nullptr : xout;          // the joined row itself is only kept from x_row0 on
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < d) {
                if constexpr (CONCAT) {
                    if (xo) *reinterpret_cast<f32x4*>(xo + r * d + c) = v[i];
                }
                sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        if (y) {
            const float mu = wave_sum(sum) / d;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + i * 256;
                if (c < d) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mu; sq += t * t; }
                }
            }
            const float rs = rsqrtf(wave_sum(sq) / d + eps);
            if (lane == 0) { if (mean) mean[r] = mu; if (rstd) rstd[r] = rs; }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + i * 256;
                if (c < d) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + bb[e];
                    store4<TOUT>(y + r * d + c, o);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vn[i];
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// x comes either from a [M,d] f32 buffer or (CONCAT) straight from the three tensors the forward
// joined, so the joined rows need not be kept.  dx_out == NULL: only the affine gradients are wanted
// (the first block's input is data), which drops the two row reductions and the row store.
// TG != void: the rows of dx_out also leave as g = dx_out * keep/(1-p) in TG (the gradient entering the residual
// branch below this LayerNorm, i.e. the backward of `x + dropout(y)` w.r.t. y), and the column sums of g (the bias
// gradient of the Linear that produced y) are accumulated next to dgamma / dbeta as a third partial row.
struct LnDropOut { uint32_t thr, k0, k1; float scale; void* g; };

template <typename TDY, typename TIN, bool CONCAT, int NV, typename TG = void>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* dy, const float* x, const TIN* img, long img_bs, long img_ss,
                                                      const TIN* sos, const TIN* cap,
                                                      const float* pos, int B, int P, int L, const float* gamma,
                                                      const float* mean, const float* rstd, long M, int d,
                                                      const float* dx_in, float* dx_out, float* partial, LnDropOut dr) {
    constexpr bool GOUT = !std::is_void<TG>::value;
    constexpr int NR = GOUT ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float red[4 * NR * 64 * 4];   // [wave][dgamma|dbeta|dbias][lane][4], one 1024-column slab per pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[NV], db[NV], dc[GOUT ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) { dg[i] = f32x4{0, 0, 0, 0}; db[i] = f32x4{0, 0, 0, 0}; }
    if constexpr (GOUT) {
#pragma unroll
        for (int i = 0; i < NV; ++i) dc[i] = f32x4{0, 0, 0, 0};
    }
    const long nwaves = (long)gridDim.x * 4;
    for (long r = (long)blockIdx.x * 4 + wave; r < M; r += nwaves) {
        const float mu = mean[r], rs = rstd[r];
        long sq = 0;
        const TIN* src = nullptr;
        if constexpr (CONCAT) {
            sq = r / B;
            const long b = r - sq * B;
            src = sq < P ? img + b * img_bs + sq * img_ss : (sq == P ? sos + b * d : cap + (b * L + (sq - P - 1)) * d);
        }
        f32x4 g[NV], xh[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            g[i] = f32x4{0, 0, 0, 0}; xh[i] = f32x4{0, 0, 0, 0};
            if (c < d) {
                const f32x4 dyv = load4<TDY>(dy + r * d + c);
                f32x4 xv;
                if constexpr (CONCAT) {
                    xv = load4_stream<TIN>(src + c);
                    if (pos) xv += *reinterpret_cast<const f32x4*>(pos + sq * d + c);
                } else {
                    xv = *reinterpret_cast<const f32x4*>(x + r * d + c);
                }
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = dyv[e] * gm[e];
                    c1 += g[i][e];
                    c2 += g[i][e] * xh[i][e];
                    dg[i][e] += dyv[e] * xh[i][e];
                    db[i][e] += dyv[e];
                }
            }
        }
        if (dx_out) {
            c1 = wave_sum(c1) / d;
            c2 = wave_sum(c2) / d;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + i * 256;
                if (c < d) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
                    if (dx_in) o += *reinterpret_cast<const f32x4*>(dx_in + r * d + c);
                    *reinterpret_cast<f32x4*>(dx_out + r * d + c) = o;
                    if constexpr (GOUT) {
                        if (dr.thr) {
                            bool keep[4];
                            dropout_keep4(dr.k0, dr.k1, (uint32_t)(r * d + c), dr.thr, keep);    // d % 4 == 0: one hash per 4 columns
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = keep[e] ? o[e] * dr.scale : 0.f;
                        }
                        store4<TG>((TG*)dr.g + r * d + c, o);
                        dc[i] += o;
                    }
                }
            }
        }
    }
    // fold the 4 waves of this block, one 256-lane x 4 slab per pass, into partial[block][2][d]
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * 256 >= d) break;
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[((wave * NR + 0) * 64 + lane) * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&red[((wave * NR + 1) * 64 + lane) * 4]) = db[i];
        if constexpr (GOUT) *reinterpret_cast<f32x4*>(&red[((wave * NR + 2) * 64 + lane) * 4]) = dc[i];
        __syncthreads();
        if (wave < NR) {
            const int c = lane * 4 + i * 256;
            if (c < d) {
                f32x4 s = f32x4{0, 0, 0, 0};
                for (int w = 0; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * NR + wave) * 64 + lane) * 4]);
                *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * NR + wave) * d + c) = s;
            }
        }
    }
}

// 64 columns per block; 16 row groups each sum a fixed interleaved subset of the
// per-block partials, then a fixed-order fold: deterministic for a given grid.  A partial holds NR rows of d
// (dgamma, dbeta and, with NR == 3, the bias sums).
// gridDim.y > 1: block row y folds partials [y * chunk, (y + 1) * chunk) into partial y of `o0`
// (a [gridDim.y][NR][d] array, o1 / o2 unused), which a second launch folds to the end.
template <int NR>
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* partial, int nblk, int d, float* o0, float* o1, float* o2, int chunk) {
    __shared__ float sa[NR][16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    int k0 = 0, k1 = nblk;
    float* out[3] = {o0, o1, o2};
    if (gridDim.y > 1) {
        k0 = blockIdx.y * chunk;
        k1 = min(nblk, k0 + chunk);
#pragma unroll
        for (int j = 0; j < NR; ++j) out[j] = o0 + ((long)blockIdx.y * NR + j) * d;
    }
    float a[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) a[j] = 0.f;
    if (c < d)
        for (int k = k0 + ty; k < k1; k += 16) {
#pragma unroll
            for (int j = 0; j < NR; ++j) a[j] += partial[((long)k * NR + j) * d + c];
        }
#pragma unroll
    for (int j = 0; j < NR; ++j) sa[j][ty][tx] = a[j];
    __syncthreads();
    if (ty == 0 && c < d) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float r = 0.f;
            for (int t = 0; t < 16; ++t) r += sa[j][t][tx];
            out[j][c] = r;
        }
    }
}

// Folds nb partials into dgamma / dbeta (and dbias when given: then a partial has three rows).  Many partials (the
// row-streaming backward writes one per workgroup) are folded in two launches, the first spread over LN_FOLD_GROUPS
// times as many workgroups; `scratch` holds LN_FOLD_GROUPS partials.
constexpr int LN_FOLD_GROUPS = 32;
template <int NR>
void launch_ln_finalize_rows(const float* partial, int nb, int d, float* dgamma, float* dbeta, float* dbias, float* scratch, hipStream_t st) {
    const unsigned gx = (unsigned)((d + 63) / 64);
    float* nf = nullptr;
    if (scratch && nb >= 8 * LN_FOLD_GROUPS) {
        const int chunk = (nb + LN_FOLD_GROUPS - 1) / LN_FOLD_GROUPS;
        hipLaunchKernelGGL(ln_bwd_finalize_kernel<NR>, dim3(gx, LN_FOLD_GROUPS), dim3(1024), 0, st, partial, nb, d, scratch, nf, nf, chunk);
        hipLaunchKernelGGL(ln_bwd_finalize_kernel<NR>, dim3(gx), dim3(1024), 0, st, (const float*)scratch, LN_FOLD_GROUPS, d, dgamma, dbeta, dbias, 0);
    } else {
        hipLaunchKernelGGL(ln_bwd_finalize_kernel<NR>, dim3(gx), dim3(1024), 0, st, partial, nb, d, dgamma, dbeta, dbias, 0);
    }
}
void launch_ln_finalize(const float* partial, int nb, int d, float* dgamma, float* dbeta, float* scratch, hipStream_t st) {
    launch_ln_finalize_rows<2>(partial, nb, d, dgamma, dbeta, nullptr, scratch, st);
}

// Workgroups of a LayerNorm backward over M rows.  With dx the kernel is a read-modify-write stream whose rows each
// need two reductions before their store: many short-lived waves keep the memory system busy.  Without dx it only
// accumulates column sums: fewer, longer-lived waves leave fewer partial rows to write and fold.
int ln_bwd_blocks(long M, bool with_dx) {
    long nb = with_dx ? (M + 15) / 16 : (M + 31) / 32;
    const long cap = with_dx ? 2048 : 512;
    if (nb < 1) nb = 1;
    if (nb > cap) nb = cap;
    return (int)nb;
}

// ------------------------------------------------------------------ element-wise
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* s, TD* d, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = from_f32<TD>(to_f32<TS>(s[i]));
}

// k-panel image of a matrix: panels[k / 32][r][k % 32] = bf16(src[r, k]).  A block takes 16 rows x 256 columns: a
// thread reads 8 consecutive elements of a row (the 32 lanes of a row 512 or 1024 consecutive bytes) and writes them as one
// 16-byte chunk; a wave's store covers 8 panels x 128 consecutive bytes (2 rows).  Optionally the plain bf16 cast as well.
template <typename TS>
__global__ __launch_bounds__(256) void pack_kpanels_kernel(const TS* src, long ld_src, long rows, int K, bf16_t* panels, long panel_rows,
                                                            bf16_t* rowmajor, long ld_rm, long src_batch_stride, long panels_batch_stride) {
    src += (long)blockIdx.z * src_batch_stride;            // matrix z of a batch (one matrix: z = 0)
    panels += (long)blockIdx.z * panels_batch_stride;
    const int c = threadIdx.x & 31;                        // 8-element chunk inside the block's 256 columns
    const int k = blockIdx.x * 256 + c * 8;
    if (k >= K) return;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const long r = (long)blockIdx.y * 16 + pass * 8 + (threadIdx.x >> 5);
        if (r >= rows) continue;
        bf16x8 v;
        if constexpr (sizeof(TS) == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + r * ld_src + k), b = *reinterpret_cast<const f32x4*>(src + r * ld_src + k + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (bf16_t)a[e]; v[4 + e] = (bf16_t)b[e]; }
        } else {
            v = *reinterpret_cast<const bf16x8*>(src + r * ld_src + k);
        }
        *reinterpret_cast<bf16x8*>(panels + ((long)(k >> 5) * panel_rows + r) * 32 + (k & 31)) = v;
        if (rowmajor) *reinterpret_cast<bf16x8*>(rowmajor + r * ld_rm + k) = v;
    }
}

__global__ void dropout_mask_kernel(uint32_t thr, uint32_t k0, uint32_t k1, long n, float* out) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (!thr || dropout_keep(k0, k1, (uint32_t)i, thr)) ? 1.f : 0.f;
}

// Column sums in two deterministic stages.  Stage 1: a block owns a stripe of rows x 256 columns; a
// thread owns 4 columns and every 4th row of the stripe (4 loads in flight), the 4 row lanes are folded
// through LDS, and the stripe's sums go to partial[stripe][N].  With DROP the same pass also writes
// g = dx * keep/(1-p) in g's type (backward of `x + dropout(y)` w.r.t. y).  Stage 2 folds the stripes
// in a fixed order.
template <typename TI, typename TG, bool DROP>
__global__ __launch_bounds__(256) void colsum_stripe_kernel(const TI* x, long M, int N, long ldx, int rows_per_stripe,
                                                             uint32_t thr, uint32_t k0, uint32_t k1, float scale,
                                                             TG* g, float* partial, const float* roww) {
    __shared__ __attribute__((aligned(16))) float red[4][64][4];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + cg * 4;
    const long r0 = (long)blockIdx.y * rows_per_stripe;
    const long r1 = r0 + rows_per_stripe < M ? r0 + rows_per_stripe : M;
    f32x4 acc = f32x4{0, 0, 0, 0};
    if (n < N) {
        const bool full = n + 3 < N && (ldx & 3) == 0 && (!DROP || (N & 3) == 0);
        auto one_row = [&](long r) {
            f32x4 v = f32x4{0, 0, 0, 0};
            if (full) v = load4<TI>(x + r * ldx + n);
            else for (int e = 0; e < 4 && n + e < N; ++e) v[e] = to_f32<TI>(x[r * ldx + n + e]);
            if (roww) v *= roww[r];                                  // weighted column sums: sum_m w[m] x[m, n]
            if constexpr (DROP) {
                if (thr) {
                    const uint32_t base = (uint32_t)(r * N + n);
                    bool keep[4];
                    if ((base & 3) == 0) dropout_keep4(k0, k1, base, thr, keep);      // one hash for the lane's 4 columns
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) keep[e] = dropout_keep(k0, k1, base + e, thr);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * scale : 0.f;
                }
                if (full) store4<TG>(g + r * N + n, v);
                else for (int e = 0; e < 4 && n + e < N; ++e) g[r * N + n + e] = from_f32<TG>(v[e]);
            }
            acc += v;
        };
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) { one_row(r); one_row(r + 4); one_row(r + 8); one_row(r + 12); }
        for (; r < r1; r += 4) one_row(r);
    }
    if (!partial) return;
    *reinterpret_cast<f32x4*>(&red[rl][cg][0]) = acc;
    __syncthreads();
    if (rl == 0 && n < N) {
        f32x4 t = acc;
#pragma unroll
        for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(&red[w][cg][0]);
        for (int e = 0; e < 4 && n + e < N; ++e) partial[(long)blockIdx.y * N + n + e] = t[e];
    }
}

// out[n] = sum over stripes of partial[stripe][n]: 64 columns per block, 16 row groups, fixed-order fold
__global__ __launch_bounds__(1024) void stripe_fold_kernel(const float* partial, int nstripes, int N, float* out) {
    __shared__ float sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a = 0.f;
    if (c < N)
        for (int k = ty; k < nstripes; k += 16) a += partial[(long)k * N + c];
    sh[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c < N) {
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) r += sh[j][tx];
        out[c] = r;
    }
}

// Weighted column sums of a wide bf16 array whose rows are 16-byte aligned (the vocabulary bias gradient from kept softmax
// numerators): stage 1 with 16-byte loads.  A block owns a stripe of rows x 512 columns; a thread owns 8 columns and
// every 4th row of the stripe, 4 loads in flight; the 4 row lanes are folded through LDS.  Columns from N on (up to the
// row stride) may be read but are never written.
__global__ __launch_bounds__(256) void wcolsum_bf16_kernel(const bf16_t* x, long M, int N, long ldx, int rows_per_stripe,
                                                            const float* roww, float* partial) {
    __shared__ __attribute__((aligned(16))) float red[4][64][8];
    const int lane = threadIdx.x & 63, rl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long r0 = (long)blockIdx.y * rows_per_stripe;
    const long r1 = r0 + rows_per_stripe < M ? r0 + rows_per_stripe : M;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long n = (long)blockIdx.x * 512 + lane * 8;
    const bool in = n + 8 <= ldx && n < N;            // the whole 16-byte group lies inside the row
    if (in) {
        auto add = [&](const bf16x8& q, float w) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf((float)q[e], w, acc[e]);
        };
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const bf16x8 q0 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + r * ldx + n));
            const bf16x8 q1 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + (r + 4) * ldx + n));
            const bf16x8 q2 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + (r + 8) * ldx + n));
            const bf16x8 q3 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + (r + 12) * ldx + n));
            add(q0, roww[r]); add(q1, roww[r + 4]); add(q2, roww[r + 8]); add(q3, roww[r + 12]);
        }
        for (; r < r1; r += 4) add(__builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + r * ldx + n)), roww[r]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][lane][e] = acc[e];
    __syncthreads();
    if (rl == 0 && in) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = ((acc[e] + red[1][lane][e]) + red[2][lane][e]) + red[3][lane][e];
            if (n + e < N) partial[(long)blockIdx.y * N + n + e] = t;
        }
    }
}

// rows per stripe: about 256 stripes, at least 16 rows each, a multiple of the 4 row lanes
int colsum_stripe_rows(long M) {
    long r = (M + 255) / 256;
    if (r < 16) r = 16;
    return (int)((r + 3) / 4 * 4);
}

__global__ void embed_rows_kernel(const float* table, const int64_t* ids, long n, int d, const float* pos_row, float* out) {
    const long i = blockIdx.x;
    if (i >= n) return;
    const float* src = table + ids[i] * d;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        if (pos_row) v += *reinterpret_cast<const f32x4*>(pos_row + c);
        *reinterpret_cast<f32x4*>(out + i * d + c) = v;
    }
}

// torch.optim.Adam (coupled L2): g += wd*w; m,v EMA; w -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, long n,
                                                    float lr_over_bc1, float inv_sqrt_bc2, float beta1, float beta2, float eps,
                                                    float wd, float gscale, bf16_t* wb,
                                                    bf16_t* panels, long pfirst, long pn, int pK, long prows) {
    // panels: k-panel image [pK / 32][prows][32] of the row-major matrix that occupies elements pfirst .. pfirst + pn - 1
    // (pfirst a multiple of 4, pK a multiple of 32: a thread's 4 elements never straddle a row or a panel)
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            // one pass over 1.3 GB that nothing re-reads soon: streaming (non-temporal) accesses keep it
            // from evicting the weights' bf16 shadow and the activations from L2 / the Infinity Cache
            f32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + i));
            const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
            f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
            f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * gscale + wd * wv[e];
                mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
                vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
                wv[e] -= lr_over_bc1 * (mv[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps));
            }
            __builtin_nontemporal_store(wv, reinterpret_cast<f32x4*>(w + i));
            __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + i));
            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
            if (wb) store4<bf16_t>(wb + i, wv);
            if (panels && i >= pfirst && i - pfirst < pn) {
                const unsigned idx = (unsigned)(i - pfirst), r = idx / (unsigned)pK, k = idx - r * (unsigned)pK;
                store4<bf16_t>(panels + ((long)(k >> 5) * prows + r) * 32 + (k & 31), wv);
            }
        } else {
            for (long j = i; j < n; ++j) {
                const float gg = g[j] * gscale + wd * w[j];
                m[j] = beta1 * m[j] + (1.f - beta1) * gg;
                v[j] = beta2 * v[j] + (1.f - beta2) * gg * gg;
                w[j] -= lr_over_bc1 * (m[j] / (sqrtf(v[j]) * inv_sqrt_bc2 + eps));
                if (wb) wb[j] = (bf16_t)w[j];
                if (panels && j >= pfirst && j - pfirst < pn) {
                    const unsigned idx = (unsigned)(j - pfirst), r = idx / (unsigned)pK, k = idx - r * (unsigned)pK;
                    panels[((long)(k >> 5) * prows + r) * 32 + (k & 31)] = (bf16_t)w[j];
                }
            }
        }
    }
}

__global__ void greedy_update_kernel(int64_t* ids, uint8_t* done, int64_t* out, int B, int L, int t, int64_t eos, int64_t pad) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t tok = ids[b];
    if (done[b]) tok = pad;
    ids[b] = tok;
    out[(long)b * L + t] = tok;
    if (tok == eos) done[b] = 1;
}

// ------------------------------------------------------------------ LayerNorm affine folded into the Linear that follows it
// y = (xhat*gamma + beta) W^T + b = xhat (W*gamma)^T + (b + W beta): the Linear can consume the normalised rows
// directly.  One wave per output row j: Wf[j,:] = W[j,:] * gamma, bf[j] = b[j] + sum_c W[j,c] beta[c].
template <typename TW, typename TO>
__global__ __launch_bounds__(256) void ln_fold_fwd_kernel(const TW* W, long ldw, const float* bias, const float* gamma, const float* beta,
                                                          int N, int K, TO* Wf, long ldwf, float* bf) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    float acc = 0.f;
    for (int c = lane * 4; c < K; c += 256) {
        const f32x4 w = load4<TW>(W + (long)j * ldw + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = w[e] * g[e]; acc += w[e] * b[e]; }
        store4<TO>(Wf + (long)j * ldwf + c, o);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) bf[j] = (bias ? bias[j] : 0.f) + acc;
}

// The backward of that folding.  G = dY^T xhat (the weight-gradient product taken with the normalised rows) and
// s = column sums of dY (the bias gradient) give everything the LayerNorm + Linear pair owes:
//   dW[j,c] = gamma[c] G[j,c] + beta[c] s[j]        (in place over G)
//   dgamma[c] = sum_j W[j,c] G[j,c],   dbeta[c] = sum_j W[j,c] s[j]
// -- no product back to the rows (dY W) and no pass over the rows.  A block owns FOLD_ROWS rows j and writes one partial
// row pair [2][K]; ln_bwd_finalize_kernel folds the blocks in a fixed order.
constexpr int FOLD_ROWS = 8;
template <typename TW>
__global__ __launch_bounds__(256) void ln_fold_bwd_kernel(float* G, long ldg, const float* s, const TW* W, long ldw, const float* gamma,
                                                          const float* beta, int N, int K, float* partial) {
    const int j0 = blockIdx.x * FOLD_ROWS, j1 = min(N, j0 + FOLD_ROWS);
    for (int c = threadIdx.x * 4; c < K; c += 1024) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
        for (int j = j0; j < j1; ++j) {
            const f32x4 w = load4<TW>(W + (long)j * ldw + c);
            f32x4 gv = *reinterpret_cast<const f32x4*>(G + (long)j * ldg + c);
            const float sj = s[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { dg[e] += w[e] * gv[e]; db[e] += w[e] * sj; gv[e] = g[e] * gv[e] + b[e] * sj; }
            *reinterpret_cast<f32x4*>(G + (long)j * ldg + c) = gv;
        }
        *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * 2) * K + c) = dg;
        *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * 2 + 1) * K + c) = db;
    }
}

int grid_for(long n, int per_block, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

// ==================================================================== C ABI
extern "C" int capdec_dropout_mask(capdec_dropout d, long n, float* out, void* stream) {
    CAPDEC_CHECK(out && n >= 0 && n < (1L << 32), "dropout_mask: bad arguments");
    if (n == 0) return CAPDEC_OK;
    const uint32_t thr = d.p > 0.f ? dropout_threshold(d.p) : 0u;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, thr, d.key0, d.key1, n, out);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

template <typename TIN, typename TOUT, bool CONCAT>
static void launch_ln_fwd(const void* img, long img_bs, long img_ss, const void* sos, const void* cap, const float* xin, const float* pos,
                          const float* gamma, const float* beta, float eps, long M, int B, int P, int L, int d,
                          float* xout, long x_row0, void* y, float* mean, float* rstd, hipStream_t st) {
    const long nblk = (M + 3) / 4 < 2048 ? (M + 3) / 4 : 2048;       // 8 workgroups per CU, each wave walks several rows
    if (d <= 1024)
        hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, CONCAT, 4>), dim3((unsigned)nblk), dim3(256), 0, st,
                           (const TIN*)img, img_bs, img_ss, (const TIN*)sos, (const TIN*)cap, xin, pos, gamma, beta, eps, M, B, P, L, d,
                           xout, x_row0, (TOUT*)y, mean, rstd);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, CONCAT, 8>), dim3((unsigned)nblk), dim3(256), 0, st,
                           (const TIN*)img, img_bs, img_ss, (const TIN*)sos, (const TIN*)cap, xin, pos, gamma, beta, eps, M, B, P, L, d,
                           xout, x_row0, (TOUT*)y, mean, rstd);
}

extern "C" int capdec_concat_ln_fwd(const void* img, const void* sos, const void* cap, int in_dtype,
                                    const float* pos, const float* gamma, const float* beta, float eps,
                                    int B, int P, int L, int d,
                                    float* x, long x_row0, void* y, int y_dtype, float* mean, float* rstd, void* stream) {
    CAPDEC_CHECK(img && sos && cap, "concat_ln_fwd: null input");
    CAPDEC_CHECK(x_row0 >= 0, "concat_ln_fwd: negative x_row0");
    CAPDEC_CHECK(B > 0 && P >= 0 && L >= 0 && d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "concat_ln_fwd: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    CAPDEC_CHECK(x || y, "concat_ln_fwd: nothing to write");
    CAPDEC_CHECK(!y || (gamma && beta), "concat_ln_fwd: LayerNorm output needs gamma/beta");
    const long M = (long)(P + 1 + L) * B;
    hipStream_t st = (hipStream_t)stream;
    const bool ib = in_dtype == CAPDEC_BF16, ob = y_dtype == CAPDEC_BF16;
    if (ib && ob) launch_ln_fwd<bf16_t, bf16_t, true>(img, (long)P * d, (long)d, sos, cap, nullptr, pos, gamma, beta, eps, M, B, P, L, d, x, x_row0, y, mean, rstd, st);
    else if (ib) launch_ln_fwd<bf16_t, float, true>(img, (long)P * d, (long)d, sos, cap, nullptr, pos, gamma, beta, eps, M, B, P, L, d, x, x_row0, y, mean, rstd, st);
    else if (ob) launch_ln_fwd<float, bf16_t, true>(img, (long)P * d, (long)d, sos, cap, nullptr, pos, gamma, beta, eps, M, B, P, L, d, x, x_row0, y, mean, rstd, st);
    else launch_ln_fwd<float, float, true>(img, (long)P * d, (long)d, sos, cap, nullptr, pos, gamma, beta, eps, M, B, P, L, d, x, x_row0, y, mean, rstd, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_seqfirst_ln_fwd(const void* tgt, int in_dtype, long seq_stride, long batch_stride,
                                      const float* pos, const float* gamma, const float* beta, float eps,
                                      int S, int B, int d,
                                      float* x, long x_row0, void* y, int y_dtype, float* mean, float* rstd, void* stream) {
    CAPDEC_CHECK(tgt && x_row0 >= 0, "seqfirst_ln_fwd: bad arguments");
    CAPDEC_CHECK(S > 0 && B > 0 && d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "seqfirst_ln_fwd: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    const int align = in_dtype == CAPDEC_BF16 ? 8 : 4;
    CAPDEC_CHECK((seq_stride % align) == 0 && (batch_stride % align) == 0 && aligned16(tgt),
                 "seqfirst_ln_fwd: strides must keep rows 16-byte aligned (seq %ld, batch %ld)", seq_stride, batch_stride);
    CAPDEC_CHECK(x || y, "seqfirst_ln_fwd: nothing to write");
    CAPDEC_CHECK(!y || (gamma && beta), "seqfirst_ln_fwd: LayerNorm output needs gamma/beta");
    const long M = (long)S * B;
    hipStream_t st = (hipStream_t)stream;
    const bool ib = in_dtype == CAPDEC_BF16, ob = y_dtype == CAPDEC_BF16;
    // one segment of S rows per sample: the "patch" segment of the join with its own strides
    if (ib && ob) launch_ln_fwd<bf16_t, bf16_t, true>(tgt, batch_stride, seq_stride, tgt, tgt, nullptr, pos, gamma, beta, eps, M, B, S, 0, d, x, x_row0, y, mean, rstd, st);
    else if (ib) launch_ln_fwd<bf16_t, float, true>(tgt, batch_stride, seq_stride, tgt, tgt, nullptr, pos, gamma, beta, eps, M, B, S, 0, d, x, x_row0, y, mean, rstd, st);
    else if (ob) launch_ln_fwd<float, bf16_t, true>(tgt, batch_stride, seq_stride, tgt, tgt, nullptr, pos, gamma, beta, eps, M, B, S, 0, d, x, x_row0, y, mean, rstd, st);
    else launch_ln_fwd<float, float, true>(tgt, batch_stride, seq_stride, tgt, tgt, nullptr, pos, gamma, beta, eps, M, B, S, 0, d, x, x_row0, y, mean, rstd, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_ln_fwd(const float* x, const float* gamma, const float* beta, float eps, long M, int d,
                             void* y, int y_dtype, float* mean, float* rstd, void* stream) {
    CAPDEC_CHECK(x && y && gamma && beta, "ln_fwd: null pointer");
    CAPDEC_CHECK(d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "ln_fwd: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    if (M <= 0) return CAPDEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == CAPDEC_BF16) launch_ln_fwd<float, bf16_t, false>(nullptr, 0, 0, nullptr, nullptr, x, nullptr, gamma, beta, eps, M, 1, 0, 0, d, nullptr, 0, y, mean, rstd, st);
    else launch_ln_fwd<float, float, false>(nullptr, 0, 0, nullptr, nullptr, x, nullptr, gamma, beta, eps, M, 1, 0, 0, d, nullptr, 0, y, mean, rstd, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" long capdec_ln_bwd_workspace(long M, int d) { return (long)(ln_bwd_blocks(M, true) + LN_FOLD_GROUPS) * 2 * d * 4; }

extern "C" int capdec_ln_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                             const float* mean, const float* rstd, long M, int d,
                             const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                             void* partial, void* stream) {
    CAPDEC_CHECK(dy && x && gamma && mean && rstd && dgamma && dbeta && partial, "ln_bwd: null pointer");
    CAPDEC_CHECK(d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "ln_bwd: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    if (M <= 0) return CAPDEC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ln_bwd_blocks(M, dx_out != nullptr);
    const float* nf = nullptr;
#define CAPDEC_LNB_PLAIN(TDY, NVV)                                                                                               \
    hipLaunchKernelGGL((ln_bwd_kernel<TDY, float, false, NVV>), dim3(nb), dim3(256), 0, st, (const TDY*)dy, x, nf, 0L, 0L, nf, nf, nf, 1, 0, 0, \
                       gamma, mean, rstd, M, d, dx_in, dx_out, (float*)partial, LnDropOut{})
    if (dy_dtype == CAPDEC_BF16) { if (d <= 1024) CAPDEC_LNB_PLAIN(bf16_t, 4); else CAPDEC_LNB_PLAIN(bf16_t, 8); }
    else { if (d <= 1024) CAPDEC_LNB_PLAIN(float, 4); else CAPDEC_LNB_PLAIN(float, 8); }
#undef CAPDEC_LNB_PLAIN
    // the workspace (capdec_ln_bwd_workspace) holds the fold's scratch behind the largest grid's partials
    launch_ln_finalize((const float*)partial, nb, d, dgamma, dbeta, (float*)partial + (long)ln_bwd_blocks(M, true) * 2 * d, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" long capdec_ln_bwd_dropout_workspace(long M, int d) { return (long)(ln_bwd_blocks(M, true) + LN_FOLD_GROUPS) * 3 * d * 4; }

extern "C" int capdec_ln_bwd_dropout(const void* dy, int dy_dtype, const float* x, const float* gamma,
                                     const float* mean, const float* rstd, long M, int d,
                                     const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                                     capdec_dropout drop, void* g, int g_dtype, float* gbias,
                                     void* partial, void* stream) {
    CAPDEC_CHECK(dy && x && gamma && mean && rstd && dx_out && dgamma && dbeta && g && gbias && partial, "ln_bwd_dropout: null pointer");
    CAPDEC_CHECK(d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "ln_bwd_dropout: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    CAPDEC_CHECK(g_dtype == CAPDEC_BF16 || g_dtype == CAPDEC_F32, "ln_bwd_dropout: g must be bf16 or f32");
    CAPDEC_CHECK(M * d < (1L << 32) || drop.p <= 0.f, "ln_bwd_dropout: dropout index space exceeds 2^32 elements");
    hipStream_t st = (hipStream_t)stream;
    if (M <= 0) { (void)hipMemsetAsync(gbias, 0, (size_t)d * 4, st); return CAPDEC_OK; }
    LnDropOut dr;
    dr.thr = drop.p > 0.f ? dropout_threshold(drop.p) : 0u;
    dr.k0 = drop.key0; dr.k1 = drop.key1;
    dr.scale = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    dr.g = g;
    const int nb = ln_bwd_blocks(M, true);
    const float* nf = nullptr;
#define CAPDEC_LNB_DROP(TDY, NVV, TGG)                                                                                                 \
    hipLaunchKernelGGL((ln_bwd_kernel<TDY, float, false, NVV, TGG>), dim3(nb), dim3(256), 0, st, (const TDY*)dy, x, nf, 0L, 0L, nf, nf, nf, 1, 0, 0, \
                       gamma, mean, rstd, M, d, dx_in, dx_out, (float*)partial, dr)
#define CAPDEC_LNB_DROP_G(TDY, NVV) do { if (g_dtype == CAPDEC_BF16) CAPDEC_LNB_DROP(TDY, NVV, bf16_t); else CAPDEC_LNB_DROP(TDY, NVV, float); } while (0)
    if (dy_dtype == CAPDEC_BF16) { if (d <= 1024) CAPDEC_LNB_DROP_G(bf16_t, 4); else CAPDEC_LNB_DROP_G(bf16_t, 8); }
    else { if (d <= 1024) CAPDEC_LNB_DROP_G(float, 4); else CAPDEC_LNB_DROP_G(float, 8); }
#undef CAPDEC_LNB_DROP_G
#undef CAPDEC_LNB_DROP
    launch_ln_finalize_rows<3>((const float*)partial, nb, d, dgamma, dbeta, gbias, (float*)partial + (long)nb * 3 * d, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

namespace {
// affine gradients of a LayerNorm whose input rows are re-read from their sources: a first segment of P rows
// per sample with strides (img_bs, img_ss), then the sos row, then L caption rows
int ln_bwd_affine_from_sources(const void* dy, int dy_dtype, const void* img, long img_bs, long img_ss, const void* sos, const void* cap,
                               int in_dtype, const float* pos, const float* gamma, const float* mean, const float* rstd,
                               int B, int P, int L, int d, float* dgamma, float* dbeta, void* partial, hipStream_t st) {
    const long M = (long)(P + (L >= 0 ? 1 + L : 0)) * B;
    const int nb = ln_bwd_blocks(M, false);
    const float* nf = nullptr;
    float* nd = nullptr;
    const int Lk = L >= 0 ? L : 0;
#define CAPDEC_LNB2(TDY, TIN, NVV)                                                                                       \
    hipLaunchKernelGGL((ln_bwd_kernel<TDY, TIN, true, NVV>), dim3(nb), dim3(256), 0, st, (const TDY*)dy, nf, (const TIN*)img, \
                       img_bs, img_ss, (const TIN*)sos, (const TIN*)cap, pos, B, P, Lk, gamma, mean, rstd, M, d, nf, nd, (float*)partial, LnDropOut{})
#define CAPDEC_LNB(TDY, TIN) do { if (d <= 1024) CAPDEC_LNB2(TDY, TIN, 4); else CAPDEC_LNB2(TDY, TIN, 8); } while (0)
    const bool db = dy_dtype == CAPDEC_BF16, ib = in_dtype == CAPDEC_BF16;
    if (db && ib) CAPDEC_LNB(bf16_t, bf16_t);
    else if (db) CAPDEC_LNB(bf16_t, float);
    else if (ib) CAPDEC_LNB(float, bf16_t);
    else CAPDEC_LNB(float, float);
#undef CAPDEC_LNB
#undef CAPDEC_LNB2
    launch_ln_finalize((const float*)partial, nb, d, dgamma, dbeta, nullptr, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
}  // namespace

extern "C" int capdec_concat_ln_bwd_affine(const void* dy, int dy_dtype, const void* img, const void* sos, const void* cap, int in_dtype,
                                           const float* pos, const float* gamma, const float* mean, const float* rstd,
                                           int B, int P, int L, int d, float* dgamma, float* dbeta, void* partial, void* stream) {
    CAPDEC_CHECK(dy && img && sos && cap && gamma && mean && rstd && dgamma && dbeta && partial, "concat_ln_bwd_affine: null pointer");
    CAPDEC_CHECK(B > 0 && P >= 0 && L >= 0 && d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "concat_ln_bwd_affine: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    return ln_bwd_affine_from_sources(dy, dy_dtype, img, (long)P * d, (long)d, sos, cap, in_dtype, pos, gamma, mean, rstd, B, P, L, d,
                                      dgamma, dbeta, partial, (hipStream_t)stream);
}

extern "C" int capdec_seqfirst_ln_bwd_affine(const void* dy, int dy_dtype, const void* tgt, int in_dtype, long seq_stride, long batch_stride,
                                             const float* pos, const float* gamma, const float* mean, const float* rstd,
                                             int S, int B, int d, float* dgamma, float* dbeta, void* partial, void* stream) {
    CAPDEC_CHECK(dy && tgt && gamma && mean && rstd && dgamma && dbeta && partial, "seqfirst_ln_bwd_affine: null pointer");
    CAPDEC_CHECK(S > 0 && B > 0 && d > 0 && (d % 4) == 0 && d <= LN_MAX_D, "seqfirst_ln_bwd_affine: need d %% 4 == 0 and d <= %d (d=%d)", LN_MAX_D, d);
    const int align = in_dtype == CAPDEC_BF16 ? 8 : 4;
    CAPDEC_CHECK((seq_stride % align) == 0 && (batch_stride % align) == 0 && aligned16(tgt),
                 "seqfirst_ln_bwd_affine: strides must keep rows 16-byte aligned (seq %ld, batch %ld)", seq_stride, batch_stride);
    return ln_bwd_affine_from_sources(dy, dy_dtype, tgt, batch_stride, seq_stride, tgt, tgt, in_dtype, pos, gamma, mean, rstd, B, S, -1, d,
                                      dgamma, dbeta, partial, (hipStream_t)stream);
}

extern "C" int capdec_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
    CAPDEC_CHECK(src && dst && n >= 0, "cast: bad arguments");
    if (n == 0) return CAPDEC_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(grid_for(n, 256, 8192)), b(256);
    if (src_dtype == CAPDEC_F32 && dst_dtype == CAPDEC_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == CAPDEC_BF16 && dst_dtype == CAPDEC_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == CAPDEC_F32 && dst_dtype == CAPDEC_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

static int pack_kpanels_impl(const void* src, int src_dtype, long rows, int K, long ld_src, long src_batch_stride,
                            void* panels, long panel_rows, long panels_batch_stride, int batch, void* rowmajor, long ld_rm, void* stream);

extern "C" int capdec_pack_kpanels(const void* src, int src_dtype, long rows, int K, long ld_src,
                                   void* panels, long panel_rows, void* rowmajor, long ld_rm, void* stream) {
    return pack_kpanels_impl(src, src_dtype, rows, K, ld_src, 0, panels, panel_rows, 0, 1, rowmajor, ld_rm, stream);
}

extern "C" int capdec_pack_kpanels_batched(const void* src, int src_dtype, long rows, int K, long ld_src, long src_batch_stride,
                                           void* panels, long panel_rows, long panels_batch_stride, int batch, void* stream) {
    const long per16 = src_dtype == CAPDEC_F32 ? 4 : 8;
    CAPDEC_CHECK(batch >= 0 && batch <= 65535 && (src_batch_stride % per16) == 0 && (panels_batch_stride % 8) == 0,
                 "pack_kpanels_batched: the matrices and their images must lie whole 16-byte units apart (batch <= 65535)");
    if (batch == 0) return CAPDEC_OK;
    return pack_kpanels_impl(src, src_dtype, rows, K, ld_src, src_batch_stride, panels, panel_rows, panels_batch_stride, batch, nullptr, 0, stream);
}

static int pack_kpanels_impl(const void* src, int src_dtype, long rows, int K, long ld_src, long src_batch_stride,
                            void* panels, long panel_rows, long panels_batch_stride, int batch, void* rowmajor, long ld_rm, void* stream) {
    CAPDEC_CHECK(src && panels, "pack_kpanels: null pointer");
    CAPDEC_CHECK(src_dtype == CAPDEC_F32 || src_dtype == CAPDEC_BF16, "pack_kpanels: bad dtype");
    const long per16 = src_dtype == CAPDEC_F32 ? 4 : 8;
    CAPDEC_CHECK(rows >= 0 && K > 0 && (K % 32) == 0 && ld_src >= K && (ld_src % per16) == 0 && panel_rows >= rows && aligned16(src) && aligned16(panels),
                 "pack_kpanels: K (%d) must be a multiple of 32, rows of src 16-byte aligned, panel_rows (%ld) >= rows (%ld)", K, panel_rows, rows);
    CAPDEC_CHECK(!rowmajor || (ld_rm >= K && (ld_rm % 8) == 0 && aligned16(rowmajor)), "pack_kpanels: rows of the row-major copy must be 16-byte aligned");
    if (rows == 0) return CAPDEC_OK;
    CAPDEC_CHECK((rows + 15) / 16 <= 65535L * 16, "pack_kpanels: too many rows");
    hipStream_t st = (hipStream_t)stream;
    // the grid's y extent is limited to 65535: rows beyond that go to further launches
    for (long r0 = 0; r0 < rows; r0 += 65535L * 16) {
        const long nr = rows - r0 < 65535L * 16 ? rows - r0 : 65535L * 16;
        dim3 g((unsigned)((K + 255) / 256), (unsigned)((nr + 15) / 16), (unsigned)batch), b(256);
        bf16_t* rm = rowmajor ? (bf16_t*)rowmajor + r0 * ld_rm : nullptr;
        if (src_dtype == CAPDEC_F32)
            hipLaunchKernelGGL(pack_kpanels_kernel<float>, g, b, 0, st, (const float*)src + r0 * ld_src, ld_src, nr, K, (bf16_t*)panels + r0 * 32, panel_rows, rm, ld_rm,
                               src_batch_stride, panels_batch_stride);
        else
            hipLaunchKernelGGL(pack_kpanels_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)src + r0 * ld_src, ld_src, nr, K, (bf16_t*)panels + r0 * 32, panel_rows, rm, ld_rm,
                               src_batch_stride, panels_batch_stride);
    }
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

// ------------------------------------------------------------------ transposed copy of a bf16 matrix
// dst[c, r] = src[r, c]: 64x64 tiles through LDS (row pitch 66 elements: neighbouring rows of a column fall on different
// banks).  PAIR: both leading dimensions even and both bases 4-byte aligned -- two elements per access on both sides.
namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <bool PAIR>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* src, long lds_, bf16_t* dst, long ldd, int rows, int cols) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 column pairs x 8 rows per pass
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = r0 + ty + 8 * p, c = c0 + 2 * tx;
        bf16_t a = (bf16_t)0.f, b = (bf16_t)0.f;
        if (r < rows) {
            if (PAIR && c + 1 < cols) {
                const bf16x2_t v = *reinterpret_cast<const bf16x2_t*>(src + (long)r * lds_ + c);
                a = v[0]; b = v[1];
            } else {
                if (c < cols) a = src[(long)r * lds_ + c];
                if (c + 1 < cols) b = src[(long)r * lds_ + c + 1];
            }
        }
        tile[ty + 8 * p][2 * tx] = a; tile[ty + 8 * p][2 * tx + 1] = b;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int c = c0 + ty + 8 * p, r = r0 + 2 * tx;              // output row c, output columns r, r + 1
        if (c < cols) {
            const bf16_t a = tile[2 * tx][ty + 8 * p], b = tile[2 * tx + 1][ty + 8 * p];
            if (PAIR && r + 1 < rows) {
                bf16x2_t v; v[0] = a; v[1] = b;
                *reinterpret_cast<bf16x2_t*>(dst + (long)c * ldd + r) = v;
            } else {
                if (r < rows) dst[(long)c * ldd + r] = a;
                if (r + 1 < rows) dst[(long)c * ldd + r + 1] = b;
            }
        }
    }
}
}  // namespace

extern "C" int capdec_transpose_bf16(const void* src, long ld_src, long rows, long cols, void* dst, long ld_dst, void* stream) {
    CAPDEC_CHECK(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows, "transpose: bad arguments");
    CAPDEC_CHECK(rows < (1L << 31) && cols < (1L << 31) && (rows + 63) / 64 < 65536, "transpose: too large");
    if (rows == 0 || cols == 0) return CAPDEC_OK;
    const bool pair = (ld_src & 1) == 0 && (ld_dst & 1) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0;
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    if (pair) hipLaunchKernelGGL(transpose_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, (int)rows, (int)cols);
    else hipLaunchKernelGGL(transpose_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, (int)rows, (int)cols);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" long capdec_colsum_workspace(long M, int N) {
    const int rows = colsum_stripe_rows(M);
    return ((M + rows - 1) / rows) * (long)N * 4 + 16;
}

extern "C" int capdec_dropout_bwd_cast(const float* dx, long M, int N, capdec_dropout d,
                                       void* g, int g_dtype, float* colsum,
                                       void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(dx && g && M >= 0 && N > 0, "dropout_bwd_cast: bad arguments");
    CAPDEC_CHECK(M * N < (1L << 32) || d.p <= 0.f, "dropout_bwd_cast: dropout index space exceeds 2^32 elements");
    if (colsum && (!workspace || workspace_bytes < capdec_colsum_workspace(M, N))) {
        capdec_set_error("dropout_bwd_cast: workspace too small (%ld < %ld)", workspace_bytes, capdec_colsum_workspace(M, N));
        return CAPDEC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (colsum) (void)hipMemsetAsync(colsum, 0, (size_t)N * 4, st);
        return CAPDEC_OK;
    }
    const uint32_t thr = d.p > 0.f ? dropout_threshold(d.p) : 0u;
    const float scale = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    const int rows = colsum_stripe_rows(M);
    const int nstripes = (int)((M + rows - 1) / rows);
    dim3 grid((N + 255) / 256, (unsigned)nstripes), block(256);
    float* part = colsum ? (float*)workspace : nullptr;
    if (g_dtype == CAPDEC_BF16)
        hipLaunchKernelGGL((colsum_stripe_kernel<float, bf16_t, true>), grid, block, 0, st, dx, M, N, (long)N, rows, thr, d.key0, d.key1, scale, (bf16_t*)g, part, (const float*)nullptr);
    else
        hipLaunchKernelGGL((colsum_stripe_kernel<float, float, true>), grid, block, 0, st, dx, M, N, (long)N, rows, thr, d.key0, d.key1, scale, (float*)g, part, (const float*)nullptr);
    if (colsum) hipLaunchKernelGGL(stripe_fold_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)part, nstripes, N, colsum);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

namespace {
int colsum_launch(const void* x, int dtype, long M, int N, long ldx, const float* roww, float* colsum, void* workspace, hipStream_t st) {
    if (M == 0) { (void)hipMemsetAsync(colsum, 0, (size_t)N * 4, st); return CAPDEC_OK; }
    const int rows = colsum_stripe_rows(M);
    const int nstripes = (int)((M + rows - 1) / rows);
    dim3 grid((N + 255) / 256, (unsigned)nstripes), block(256);
    float* nog = nullptr;
    if (dtype == CAPDEC_BF16 && roww && N >= 4096 && (ldx % 8) == 0 && ((uintptr_t)x & 15) == 0 && ((long)(N + 7) / 8 * 8 <= ldx)) {
        hipLaunchKernelGGL(wcolsum_bf16_kernel, dim3((N + 511) / 512, (unsigned)nstripes), block, 0, st, (const bf16_t*)x, M, N, ldx, rows, roww, (float*)workspace);
        hipLaunchKernelGGL(stripe_fold_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)workspace, nstripes, N, colsum);
        CAPDEC_LAUNCH_CHECK();
        return CAPDEC_OK;
    }
    else if (dtype == CAPDEC_BF16)
        hipLaunchKernelGGL((colsum_stripe_kernel<bf16_t, float, false>), grid, block, 0, st, (const bf16_t*)x, M, N, ldx, rows, 0u, 0u, 0u, 1.f, nog, (float*)workspace, roww);
    else
        hipLaunchKernelGGL((colsum_stripe_kernel<float, float, false>), grid, block, 0, st, (const float*)x, M, N, ldx, rows, 0u, 0u, 0u, 1.f, nog, (float*)workspace, roww);
    hipLaunchKernelGGL(stripe_fold_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)workspace, nstripes, N, colsum);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
}  // namespace

extern "C" int capdec_colsum(const void* x, int dtype, long M, int N, long ldx, float* colsum,
                             void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(x && colsum && M >= 0 && N > 0 && ldx >= N, "colsum: bad arguments");
    if (!workspace || workspace_bytes < capdec_colsum_workspace(M, N)) {
        capdec_set_error("colsum: workspace too small (%ld < %ld)", workspace_bytes, capdec_colsum_workspace(M, N));
        return CAPDEC_ERR_WORKSPACE;
    }
    return colsum_launch(x, dtype, M, N, ldx, nullptr, colsum, workspace, (hipStream_t)stream);
}

extern "C" int capdec_colsum_weighted(const void* x, int dtype, long M, int N, long ldx, const float* row_weight, float* colsum,
                                      void* workspace, long workspace_bytes, void* stream) {
    CAPDEC_CHECK(x && colsum && row_weight && M >= 0 && N > 0 && ldx >= N, "colsum_weighted: bad arguments");
    if (!workspace || workspace_bytes < capdec_colsum_workspace(M, N)) {
        capdec_set_error("colsum_weighted: workspace too small (%ld < %ld)", workspace_bytes, capdec_colsum_workspace(M, N));
        return CAPDEC_ERR_WORKSPACE;
    }
    return colsum_launch(x, dtype, M, N, ldx, row_weight, colsum, workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------ column sums without a caller workspace
// The stripe partials of capdec_colsum live in a fixed array inside the code object instead of a buffer the
// caller allocates, for callers that cannot allocate (HIP graph capture, serving loops).  Each GPU holds its own
// copy of the array at its own address; host code only ever uses the address hipGetSymbolAddress returned for
// the current device, never the array's name.
constexpr long COLSUM_STATIC_BYTES = CAPDEC_COLSUM_STATIC_BYTES;
static __device__ __align__(256) unsigned char scratch[COLSUM_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];            // per device: device address of `scratch`, 0 until first use there
// address of this device's copy of the array (resolved once per device, with that device current).  The kernels run on
// the device that owns `stream`: it has to be the current one, or the address would be another GPU's copy.
int static_scratch(void** out, hipStream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        capdec_set_error("colsum_static: cannot identify the current device (%s)", hipGetErrorString(e));
        return CAPDEC_ERR_INVALID;
    }
    if (stream) {
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e == hipSuccess && (int)sdev != dev) {
            capdec_set_error("colsum_static: the stream belongs to device %d, the current device is %d (make the stream's device current)", (int)sdev, dev);
            return CAPDEC_ERR_INVALID;
        }
        (void)hipGetLastError();       // a runtime that cannot tell (e != hipSuccess) is not an error of this call
    }
    void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            capdec_set_error("colsum_static: hipGetSymbolAddress failed on device %d (%s)", dev, hipGetErrorString(e));
            return CAPDEC_ERR_LAUNCH;
        }
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);      // every thread resolves the same address
    }
    *out = p;
    return CAPDEC_OK;
}
}  // namespace

extern "C" int capdec_colsum_static(const void* x, int dtype, long M, int N, long ldx, float* colsum, void* stream) {
    CAPDEC_CHECK(x && colsum && M >= 0 && N > 0 && ldx >= N, "colsum_static: bad arguments");
    const long need = capdec_colsum_workspace(M, N);
    if (need > COLSUM_STATIC_BYTES) {
        capdec_set_error("colsum_static: M=%ld, N=%d needs %ld bytes of stripe partials, the built-in buffer holds %ld "
                         "(use capdec_colsum with a caller workspace)", M, N, need, COLSUM_STATIC_BYTES);
        return CAPDEC_ERR_WORKSPACE;
    }
    void* ws = nullptr;
    const int rc = static_scratch(&ws, (hipStream_t)stream);
    if (rc) return rc;
    // every partial the fold reads is written by the stripe pass of the same call: nothing has to start at zero
    return capdec_colsum(x, dtype, M, N, ldx, colsum, ws, COLSUM_STATIC_BYTES, stream);
}

extern "C" int capdec_embed_rows(const float* table, const int64_t* ids, long n, int d, const float* pos_row,
                                 float* out, void* stream) {
    CAPDEC_CHECK(table && ids && out && n >= 0 && d > 0 && (d % 4) == 0, "embed_rows: bad arguments");
    if (n == 0) return CAPDEC_OK;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, table, ids, n, d, pos_row, out);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_adam_step(float* w, const float* g, float* m, float* v, long n, int step,
                                float lr, float beta1, float beta2, float eps, float weight_decay,
                                float grad_scale, void* w_bf16, const capdec_adam_panels* panels, void* stream) {
    CAPDEC_CHECK(w && g && m && v && n >= 0 && step >= 1, "adam_step: bad arguments");
    bf16_t* pp = nullptr;
    long pfirst = 0, pn = 0, prows = 0;
    int pK = 0;
    if (panels && panels->panels) {
        CAPDEC_CHECK(panels->rows > 0 && panels->K > 0 && (panels->K % 32) == 0 && panels->panel_rows >= panels->rows && (panels->first % 4) == 0 &&
                     panels->rows * (long)panels->K < (1L << 32) && aligned16(panels->panels),
                     "adam_step: the matrix behind a k-panel image must start at a multiple of 4 elements, K a multiple of 32, panel_rows >= rows");
        pp = (bf16_t*)panels->panels; pfirst = panels->first; pn = panels->rows * (long)panels->K; pK = panels->K; prows = panels->panel_rows;
    }
    CAPDEC_CHECK(aligned16(w) && aligned16(g) && aligned16(m) && aligned16(v), "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return CAPDEC_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, (hipStream_t)stream,
                       w, g, m, v, n, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale, (bf16_t*)w_bf16,
                       pp, pfirst, pn, pK, prows);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_greedy_update(int64_t* ids, uint8_t* done, int64_t* out, int B, int L, int t,
                                    int64_t eos_id, int64_t pad_id, void* stream) {
    CAPDEC_CHECK(ids && done && out && B > 0 && t >= 0 && t < L, "greedy_update: bad arguments");
    hipLaunchKernelGGL(greedy_update_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, done, out, B, L, t, eos_id, pad_id);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" int capdec_ln_fold_fwd(const void* W, int w_dtype, long ldw, const float* bias, const float* gamma, const float* beta,
                                  int N, int K, void* Wf, int wf_dtype, long ldwf, float* bf, void* stream) {
    CAPDEC_CHECK(W && gamma && beta && Wf && bf, "ln_fold_fwd: null pointer");
    const int al = w_dtype == CAPDEC_BF16 ? 8 : 4, alf = wf_dtype == CAPDEC_BF16 ? 8 : 4;
    CAPDEC_CHECK(N > 0 && K > 0 && (K % 4) == 0 && (ldw % al) == 0 && (ldwf % alf) == 0 && aligned16(W) && aligned16(Wf) && aligned16(gamma) && aligned16(beta),
                 "ln_fold_fwd: K must be a multiple of 4, rows 16-byte aligned (K=%d, ldw=%ld, ldwf=%ld)", K, ldw, ldwf);
    CAPDEC_CHECK(w_dtype == CAPDEC_F32 || wf_dtype == CAPDEC_BF16, "ln_fold_fwd: a bf16 matrix folds into a bf16 matrix");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + 3) / 4), block(256);
    if (w_dtype == CAPDEC_BF16)
        hipLaunchKernelGGL((ln_fold_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)W, ldw, bias, gamma, beta, N, K, (bf16_t*)Wf, ldwf, bf);
    else if (wf_dtype == CAPDEC_BF16)
        hipLaunchKernelGGL((ln_fold_fwd_kernel<float, bf16_t>), grid, block, 0, st, (const float*)W, ldw, bias, gamma, beta, N, K, (bf16_t*)Wf, ldwf, bf);
    else
        hipLaunchKernelGGL((ln_fold_fwd_kernel<float, float>), grid, block, 0, st, (const float*)W, ldw, bias, gamma, beta, N, K, (float*)Wf, ldwf, bf);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}

extern "C" long capdec_ln_fold_bwd_workspace(int N, int K) { return (long)((N + FOLD_ROWS - 1) / FOLD_ROWS) * 2 * K * 4; }

extern "C" int capdec_ln_fold_bwd(float* G, long ldg, const float* s, const void* W, int w_dtype, long ldw, const float* gamma, const float* beta,
                                  int N, int K, float* dgamma, float* dbeta, void* partial, void* stream) {
    CAPDEC_CHECK(G && s && W && gamma && beta && dgamma && dbeta && partial, "ln_fold_bwd: null pointer");
    const int al = w_dtype == CAPDEC_BF16 ? 8 : 4;
    CAPDEC_CHECK(N > 0 && K > 0 && (K % 4) == 0 && (ldg % 4) == 0 && (ldw % al) == 0 && aligned16(G) && aligned16(W) && aligned16(gamma) && aligned16(beta) && aligned16(partial),
                 "ln_fold_bwd: K must be a multiple of 4, rows 16-byte aligned (K=%d, ldg=%ld, ldw=%ld)", K, ldg, ldw);
    hipStream_t st = (hipStream_t)stream;
    const int nb = (N + FOLD_ROWS - 1) / FOLD_ROWS;
    if (w_dtype == CAPDEC_BF16)
        hipLaunchKernelGGL(ln_fold_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, G, ldg, s, (const bf16_t*)W, ldw, gamma, beta, N, K, (float*)partial);
    else
        hipLaunchKernelGGL(ln_fold_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, G, ldg, s, (const float*)W, ldw, gamma, beta, N, K, (float*)partial);
    launch_ln_finalize((const float*)partial, nb, K, dgamma, dbeta, nullptr, st);
    CAPDEC_LAUNCH_CHECK();
    return CAPDEC_OK;
}
