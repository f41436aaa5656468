// Vector-memory pipeline microbenchmark for gfx950: cycles per global_load instruction per CU as a function of the
// address pattern (L2-resident table), the load width and the number of active lanes; and per store instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gather_rate.hip -o tools/micro/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every wave issues `iters` x 8 independent loads (plain C++ loads: the compiler tracks their completion; an earlier version used
// inline-asm loads, whose results it does not track -- registers were reused while loads were in flight and the W = 4 variant faulted);
// offsets (in 8-byte elements) come from a per-lane table
template <int W> struct LoadT;
template <> struct LoadT<4> { typedef uint32_t type; };
template <> struct LoadT<8> { typedef uint64_t type; };
template <> struct LoadT<16> { typedef uint4 type; };
__device__ __forceinline__ uint64_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint64_t fold(uint64_t v) { return v; }
__device__ __forceinline__ uint64_t fold(uint4 v) { return (uint64_t)v.x + v.w; }

template <int W, int ACTIVE>
__global__ __launch_bounds__(256) void kload(const char* __restrict__ base, const uint32_t* __restrict__ offs, int iters, uint64_t* out) {
    typedef typename LoadT<W>::type T;
    const int lane = threadIdx.x & 63;
    uint32_t o[8];
    for (int i = 0; i < 8; ++i) o[i] = offs[(threadIdx.x * 8 + i) & 2047];
    uint64_t acc = 0;
    if (lane < ACTIVE) {
        for (int it = 0; it < iters; ++it) {
            T v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const T*>(base + (size_t)((o[i] + it * 64u) & 0x3ffffu) * 8u);  // inside 2 MiB (+ 16 B)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += fold(v[i]);
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int W>
__global__ __launch_bounds__(256) void kstore(char* __restrict__ base, int iters) {
    // each wave writes its own contiguous spans: 64 lanes x W bytes per instruction
    typedef typename LoadT<W>::type T;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    T* p = reinterpret_cast<T*>(base + (wave * (size_t)iters * 8) * 64 * W) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            T v;
            if constexpr (W == 16) v = make_uint4((uint32_t)it, (uint32_t)i, 0u, 1u); else v = (T)(it + i);
            p[(size_t)(it * 8 + i) * 64] = v;
        }
    }
}

static std::vector<uint32_t> pattern(int kind) {
    std::vector<uint32_t> o(2048);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (int t = 0; t < 256; ++t)
        for (int i = 0; i < 8; ++i) {
            const int lane = t & 63;
            uint32_t v = 0;
            if (kind == 0) v = lane + 64 * i + 512 * (t >> 6);                       // coalesced: 64 consecutive elements
            if (kind == 1) v = (rnd() % 400) + 400 * i + 3200 * (t >> 6);           // random inside a 3.2 KB window (25 lines)
            if (kind == 2) v = rnd() & 0x3ffff;                                      // random over 2 MiB: 64 different lines
            if (kind == 3) v = 16 * lane + 1024 * i + 8192 * (t >> 6);              // one element per 128-byte line, regular
            if (kind == 4) v = (lane >> 2) * 16 + (lane & 3) + 1024 * i;            // 4 lanes per line, 16 lines
            o[t * 8 + i] = v;
        }
    return o;
}

template <int W, int ACTIVE>
void runload(const char* name, int kind, char* buf, uint32_t* doffs, uint64_t* out) {
    auto o = pattern(kind);
    CK(hipMemcpy(doffs, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    const int iters = 512;
    dim3 grid(256 * 8), block(256);  // 8 workgroups = 32 waves per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kload<W, ACTIVE><<<grid, block>>>(buf, doffs, 8, out); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); kload<W, ACTIVE><<<grid, block>>>(buf, doffs, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double instr_per_cu = 8.0 * 4 * iters * 8;  // workgroups x waves x iters x loads
    fflush(stdout); printf("load  %-44s W=%2d active=%2d : %7.3f ms  %.1f cycles per instruction per CU (2.4 GHz)  %.2f TB/s useful\n", name, W, ACTIVE, best,
           best * 1e-3 * 2.4e9 / instr_per_cu, 256.0 * instr_per_cu * ACTIVE * W / (best * 1e-3) / 1e12);
}

template <int W>
void runstore(char* buf) {
    const int iters = 16;
    dim3 grid(256 * 8), block(256);
    const size_t need = (size_t)grid.x * 4 * iters * 8 * 64 * W;  // bytes the kernel writes
    if (need > ((size_t)1 << 31)) { printf("store test would overrun the buffer\n"); exit(1); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kstore<W><<<grid, block>>>(buf, 4); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); kstore<W><<<grid, block>>>(buf, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double instr_per_cu = 8.0 * 4 * iters * 8;
    printf("store dense spans W=%2d : %7.3f ms  %.1f cycles per instruction per CU  %.2f TB/s\n", W, best, best * 1e-3 * 2.4e9 / instr_per_cu,
           256.0 * instr_per_cu * 64 * W / (best * 1e-3) / 1e12);
}

int main() {
    char* buf; CK(hipMalloc(&buf, (size_t)1 << 31)); CK(hipMemset(buf, 1, (size_t)1 << 22));
    uint32_t* doffs; CK(hipMalloc(&doffs, 2048 * 4));
    uint64_t* out; CK(hipMalloc(&out, 8));
    runload<8, 64>("coalesced 512 B", 0, buf, doffs, out);
    runload<8, 64>("random in a 3.2 KB window (<= 25 lines)", 1, buf, doffs, out);
    runload<8, 64>("random over 2 MiB (64 lines)", 2, buf, doffs, out);
    runload<8, 64>("one element per line, regular (64 lines)", 3, buf, doffs, out);
    runload<8, 64>("4 lanes per line (16 lines)", 4, buf, doffs, out);
    runload<8, 11>("random in a 3.2 KB window, 11 lanes", 1, buf, doffs, out);
    runload<8, 32>("random in a 3.2 KB window, 32 lanes", 1, buf, doffs, out);
    runload<8, 11>("random over 2 MiB, 11 lanes", 2, buf, doffs, out);
    runload<4, 64>("random in a 3.2 KB window", 1, buf, doffs, out);
    runload<16, 64>("random in a 3.2 KB window", 1, buf, doffs, out);
    runload<16, 64>("coalesced 1 KB", 0, buf, doffs, out);
    runload<4, 64>("random over 2 MiB", 2, buf, doffs, out);
    runload<16, 64>("random over 2 MiB", 2, buf, doffs, out);
    runstore<8>(buf);
    runstore<16>(buf);
    return 0;
}
