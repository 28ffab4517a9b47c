// Shared device/host helpers for libcapdec (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/capdec.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---------------------------------------------------------------- errors
void capdec_set_error(const char* fmt, ...);

#define CAPDEC_CHECK(cond, ...)                   \
    do {                                          \
        if (!(cond)) {                            \
            capdec_set_error(__VA_ARGS__);        \
            return CAPDEC_ERR_INVALID;            \
        }                                         \
    } while (0)

#define CAPDEC_LAUNCH_CHECK()                                             \
    do {                                                                  \
        hipError_t e__ = hipGetLastError();                               \
        if (e__ != hipSuccess) {                                          \
            capdec_set_error("launch failed: %s", hipGetErrorString(e__)); \
            return CAPDEC_ERR_LAUNCH;                                     \
        }                                                                 \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- dynamic LDS above 64 KiB
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON A DEVICE: it is set once per (kernel, device),
// whichever host thread launches first there (lock-free table; a full table or an unknown device sets it every time).
inline void ensure_dynamic_lds(const void* fn, int bytes) {
    struct Slot { const void* fn; unsigned long long devs; };
    static Slot table[512];
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        const size_t h = (size_t)(((uintptr_t)fn >> 4) * 2654435761u);
        for (int probe = 0; probe < 512; ++probe) {
            Slot& s = table[(h + probe) & 511];
            const void* cur = __atomic_load_n(&s.fn, __ATOMIC_ACQUIRE);
            if (!cur) {
                const void* expected = nullptr;
                cur = __atomic_compare_exchange_n(&s.fn, &expected, fn, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) ? fn : expected;
            }
            if (cur != fn) continue;
            if (__atomic_load_n(&s.devs, __ATOMIC_ACQUIRE) & (1ull << dev)) return;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            __atomic_fetch_or(&s.devs, 1ull << dev, __ATOMIC_RELEASE);
            return;
        }
    }
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---------------------------------------------------------------- dtype access
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float load_as_f32(const void* p, long i, int dtype) {
    return dtype == CAPDEC_BF16 ? (float)((const bf16_t*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void store_from_f32(void* p, long i, int dtype, float v) {
    if (dtype == CAPDEC_BF16) ((bf16_t*)p)[i] = (bf16_t)v; else ((float*)p)[i] = v;
}

// ---------------------------------------------------------------- dropout RNG
// Counter-based: keep(idx) depends only on (key0, key1, idx), so the backward pass regenerates the
// forward's mask from the same keys with no stored mask.  One hash serves FOUR consecutive elements:
// element idx takes the 16-bit field (idx & 3) of hash(key0, key1, idx >> 2) and is kept when that field
// is >= thr16 = floor(p * 65536).  32-bit integer multiplies are quarter-rate instructions, so the hash
// (one multiply-xorshift round, then two different multipliers for the two output words: four multiplies)
// is shared by the four elements wherever a lane holds an aligned group of them.
struct DropBits { uint32_t a, b; };    // fields 0,1 = low/high half of a; fields 2,3 = low/high half of b
__device__ __forceinline__ DropBits dropout_hash4(uint32_t key0, uint32_t key1, uint32_t group) {
    uint32_t x = (group ^ key1) * 0x9E3779B1U + key0;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    DropBits h;
    h.a = x * 0x846ca68bU; h.a ^= h.a >> 16;
    h.b = x * 0x9E485565U; h.b ^= h.b >> 16;
    return h;
}
__device__ __forceinline__ uint32_t dropout_field(DropBits h, uint32_t f) {
    const uint32_t w = (f & 2) ? h.b : h.a;
    return (f & 1) ? w >> 16 : w & 0xFFFFu;
}
// any single element
__device__ __forceinline__ bool dropout_keep(uint32_t key0, uint32_t key1, uint32_t idx, uint32_t thr) {
    return dropout_field(dropout_hash4(key0, key1, idx >> 2), idx & 3) >= thr;
}
// the four elements idx4 .. idx4+3, idx4 a multiple of 4
__device__ __forceinline__ void dropout_keep4(uint32_t key0, uint32_t key1, uint32_t idx4, uint32_t thr, bool (&keep)[4]) {
    const DropBits h = dropout_hash4(key0, key1, idx4 >> 2);
    keep[0] = (h.a & 0xFFFFu) >= thr; keep[1] = (h.a >> 16) >= thr;
    keep[2] = (h.b & 0xFFFFu) >= thr; keep[3] = (h.b >> 16) >= thr;
}
// thr16 of a probability p > 0 (at least 1, so that "dropout on" and thr != 0 stay equivalent)
static inline uint32_t dropout_threshold(float p) {
    double t = (double)p * 65536.0;
    if (t < 1.0) return 1;
    if (t >= 65535.0) return 65535U;
    return (uint32_t)t;
}

// ---------------------------------------------------------------- activations
// Exact (erf-based) GELU, with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at f32 rounding level
// for the CDF) so that the CDF and the density share ONE exponential: e^{-x^2/2} is both the density's
// and, with y = |x| / sqrt(2), the e^{-y^2} of the erf formula.  One v_exp, one v_rcp and a dozen FMAs per element,
// against a libm erff plus a second exponential.
__device__ __forceinline__ void normal_cdf_pdf(float x, float& cdf, float& pdf) {
    const float u = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);       // e^{-x^2/2}
    const float y = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, y, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float half_erf = fmaf(-0.5f * poly, u, 0.5f);                          // erf(y) / 2
    cdf = 0.5f + copysignf(half_erf, x);
    pdf = 0.39894228040143268f * u;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, pdf;
    normal_cdf_pdf(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, pdf;
    normal_cdf_pdf(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reduce for blockDim.x <= 1024 (multiple of 64); scratch holds >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}
