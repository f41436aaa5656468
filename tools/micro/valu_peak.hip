// VALU issue-rate microbenchmark for gfx950: wave64 instructions per second for a few instruction kinds, every CU busy.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_peak.hip -o tools/micro/valu_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8]; double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; d[i] = 1.0 + 1e-9 * a[i]; f[i] = 1.0f + 1e-6f * a[i]; }
    unsigned b = seed | 1u; double db = 1.0000001; float fb = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
                if constexpr (KIND == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
                if constexpr (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fb));
                if constexpr (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
                if constexpr (KIND == 8) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 9) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[i]));
            }
        }
    }
    unsigned s = 0; double ds = 0; float fs = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; ds += d[i]; fs += f[i]; }
    if (s == 0x12345 && ds == 3.0 && fs == 2.0f) out[0] = s;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    unsigned* out; CK(hipMalloc(&out, 4));
    int iters = 4096;
    int cus = 256;
    // waves_per_simd waves on each of the 4 SIMDs of every CU: blocks of 256 threads = 4 waves = one per SIMD
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<KIND><<<grid, block>>>(out, 64, 1); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0)); k<KIND><<<grid, block>>>(out, iters, 1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double insts = double(grid.x) * 4 /*waves*/ * iters * 32.0;
    double rate = insts / (best * 1e-3);
    printf("%-16s waves/SIMD %2d : %8.3f ms  %.3e wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, best, rate,
           1024.0 * 2.4e9 / rate);
    CK(hipFree(out));
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", w); run<1>("v_xor_b32", w); run<6>("v_lshl_add_u32", w); run<8>("v_and_or_b32", w); run<9>("v_bfe_u32", w);
        run<5>("v_mul_lo_u32", w);
        run<4>("v_fma_f32", w); run<2>("v_fma_f64", w); run<3>("v_mul_f64", w); run<7>("v_add_f64", w);
    }
    return 0;
}
