// How many 256-thread workgroups are REALLY resident per CU for a given dynamic LDS size (and SGPR/VGPR footprint of a small kernel)?
// Every workgroup notes wall_clock64() at its start and end and busy-waits ~30 us in between; the peak number alive at once / 256 CUs is
// the residency.  Beside it, what hipOccupancyMaxActiveBlocksPerMultiprocessor promises.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
extern __shared__ unsigned char smem[];
__global__ __launch_bounds__(256) void k(unsigned long long *t, int lds) {
  __shared__ unsigned int few[19];  // (a few static words beside the dynamic region, as the real kernels have)
  if (threadIdx.x < 19) few[threadIdx.x] = threadIdx.x;
  smem[threadIdx.x] = 1; smem[lds - 1 - threadIdx.x] = 2;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(8);  // 30 us at 100 MHz
  __syncthreads();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64() + smem[0] - 1 + few[0]; }
}
int main() {
  const int n = 8192;
  unsigned long long *d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * n);
  std::vector<unsigned long long> h(2 * n);
  for (int lds : {16384, 19456, 19968, 20224, 20480, 22208, 22528, 23040, 23296, 23552, 26624}) {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), lds, 0, d, lds);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < n; ++i) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
    std::sort(ev.begin(), ev.end());
    int alive = 0, peak = 0;
    for (auto &e : ev) { alive += e.second; peak = std::max(peak, alive); }
    printf("dynamic LDS %5d B: API %d per CU, measured peak %d alive = %.2f per CU\n", lds, nb, peak, peak / 256.0);
  }
  return 0;
}
