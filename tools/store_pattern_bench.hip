// How fast can a workgroup of 4 waves put a 256-row bf16 tile into a [8192][50304] matrix, as a function of how a
// wave's 16-byte-per-lane store instruction is shaped (rows x contiguous bytes per row)?  Diagnostic for the GEMM
// epilogues, which store 8 rows x 128 B per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/store_pattern_bench tools/store_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SEG = contiguous bytes per row covered by one wave instruction (128, 256, 512, 1024); the workgroup's tile is
// 256 rows x (2*SEG bytes if SEG == 128 else SEG bytes): waves split it like the GEMM does (2x2 waves of 128 rows x 128 B)
// or, for wider segments, as 4 row bands of 64 rows.
template <int SEG, bool NT>
__global__ __launch_bounds__(256, 2) void store_kernel(char* C, long ld_bytes, int tiles_m, int tiles_n, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;                 // lanes per row, rows per instruction
    constexpr int TILE_W = SEG == 128 ? 256 : SEG;                // tile width in bytes
    const int ntiles = tiles_m * tiles_n;
    u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int rep = 0; rep < reps; ++rep)
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int tm = t / tiles_n, tn = t % tiles_n;
            long row0, col0;
            int nrows;
            if (SEG == 128) { row0 = tm * 256L + (wave >> 1) * 128; col0 = (long)tn * TILE_W + (wave & 1) * 128; nrows = 128; }
            else { row0 = tm * 256L + wave * 64; col0 = (long)tn * TILE_W; nrows = 64; }
            char* base = C + row0 * ld_bytes + col0 + (lane % LPR) * 16;
            const int g = lane / LPR;
#pragma unroll 4
            for (int it = 0; it < nrows / RPI; ++it) {
                u32x4* dst = reinterpret_cast<u32x4*>(base + (long)(it * RPI + g) * ld_bytes);
                v[1] += it;
                if (NT) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
}

template <int SEG, bool NT>
static void run(char* C, long ld_bytes, int rows, long width_bytes, int wgs, const char* label) {
    constexpr int TILE_W = SEG == 128 ? 256 : SEG;
    const int tiles_m = rows / 256, tiles_n = (int)(width_bytes / TILE_W);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    store_kernel<SEG, NT><<<wgs, 256>>>(C, ld_bytes, tiles_m, tiles_n, 1);
    CHECK(hipDeviceSynchronize());
    const int reps = 4;
    CHECK(hipEventRecord(a));
    store_kernel<SEG, NT><<<wgs, 256>>>(C, ld_bytes, tiles_m, tiles_n, reps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)tiles_m * tiles_n * 256.0 * TILE_W * reps;
    printf("%-34s seg %4d B  %s  wgs %4d : %7.1f us per pass  %7.1f GB/s  %5.2f B/clk/CU at 2.4 GHz (%d CUs busy)\n", label, SEG,
           NT ? "nontemporal" : "plain      ", wgs, ms * 1e3 / reps, bytes / ms / 1e6, bytes / ms / 1e6 / 2.4 / (wgs < 256 ? wgs : 256),
           wgs < 256 ? wgs : 256);
}

int main() {
    const int rows = 8192;
    const long ld_bytes = 50304L * 2;
    char* C;
    CHECK(hipMalloc(&C, rows * ld_bytes));
    CHECK(hipMemset(C, 0, rows * ld_bytes));
    const long width = ld_bytes / 1024 * 1024;       // whole tiles for every variant
    for (int wgs : {64, 256, 512}) {
        run<128, true>(C, ld_bytes, rows, width, wgs, "8 rows x 128 B (GEMM epilogue)");
        run<128, false>(C, ld_bytes, rows, width, wgs, "8 rows x 128 B (GEMM epilogue)");
        run<256, true>(C, ld_bytes, rows, width, wgs, "4 rows x 256 B");
        run<256, false>(C, ld_bytes, rows, width, wgs, "4 rows x 256 B");
        run<512, true>(C, ld_bytes, rows, width, wgs, "2 rows x 512 B");
        run<512, false>(C, ld_bytes, rows, width, wgs, "2 rows x 512 B");
        run<1024, true>(C, ld_bytes, rows, width, wgs, "1 row x 1024 B");
        run<1024, false>(C, ld_bytes, rows, width, wgs, "1 row x 1024 B");
    }
    CHECK(hipFree(C));
    return 0;
}
