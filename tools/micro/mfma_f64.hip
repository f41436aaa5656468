// f64 MFMA issue rate on gfx950 and its overlap with f64 VALU work of other waves on the same SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64.hip -o tools/micro/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: every wave issues MFMAs (NACC independent accumulators); MODE 1: every wave f64 VALU (8 chains);
// MODE 2: even waves MFMA, odd waves VALU (same SIMD: a 512-thread block puts waves w and w+4 on SIMD w%4)
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k(double* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (MODE == 2 && wave < 4);
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    double s = 0.0;
    if (do_mfma) {
        d4 c[NACC];
        for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        double d[8];
        for (int i = 0; i < 8; ++i) d[i] = 1.0 + 1e-9 * i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(b));
        }
        for (int i = 0; i < 8; ++i) s += d[i];
    }
    if (s == 12345.678) out[0] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks_per_cu) {
    double* out; CK(hipMalloc(&out, 8));
    const int iters = 2048;
    dim3 grid(256 * blocks_per_cu), block(512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE, NACC><<<grid, block>>>(out, 16); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0)); k<MODE, NACC><<<grid, block>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double waves = double(grid.x) * 8;
    const double mfma_waves = MODE == 0 ? waves : (MODE == 2 ? waves / 2 : 0), valu_waves = MODE == 1 ? waves : (MODE == 2 ? waves / 2 : 0);
    const double n_mfma = mfma_waves * iters * 8.0 * NACC, n_valu = valu_waves * iters * 64.0;
    printf("%-34s blocks/CU %d: %8.3f ms", name, blocks_per_cu, best);
    if (n_mfma > 0) printf("  MFMA %.3e/s = %.1f cycles per MFMA per SIMD at 2.4 GHz (%.1f TFLOP/s)", n_mfma / (best * 1e-3), 1024.0 * 2.4e9 / (n_mfma / (best * 1e-3)),
                           n_mfma * 2048.0 / (best * 1e-3) / 1e12);
    if (n_valu > 0) printf("  VALU %.3e/s = %.2f cycles per v_fma_f64 per SIMD", n_valu / (best * 1e-3), 1024.0 * 2.4e9 / (n_valu / (best * 1e-3)));
    printf("\n");
    CK(hipFree(out));
}

int main() {
    run<0, 1>("MFMA f64 16x16x4, 1 accumulator", 1);
    run<0, 4>("MFMA f64 16x16x4, 4 accumulators", 1);
    run<0, 4>("MFMA f64 16x16x4, 4 accumulators", 2);
    run<1, 1>("VALU v_fma_f64 only", 1);
    run<2, 4>("MFMA (waves 0-3) + VALU (waves 4-7)", 1);
    run<2, 4>("MFMA (waves 0-3) + VALU (waves 4-7)", 2);
    return 0;
}
