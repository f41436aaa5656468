// kernels_gfmc.hip -- the move of a Green's-function Monte-Carlo step (PyNQS gfmc/walker.py:260-279, sample_update):
//   beta = sum_k G[x][k];  index = first k with cumsum(G[x])[k] / beta >= u[x];  x_new = comb[x][index]
// The reference runs sum, cumsum, division, searchsorted and an advanced-index gather as five passes over the
// [n, ncomb] Green's-function matrix; here one workgroup per walker reads its row twice at most (once for the
// tile sums, then only the tile that holds the target), the row never leaves L2 in between.
//   1. every wave sums whole tiles of the row (a lane reads kPerLane consecutive values, 16-byte loads) -> LDS
//   2. the first wave scans the tile sums, finds the tile in which the running sum reaches u * beta,
//   3. re-reads that tile, scans it in column order and takes the first column that reaches the target.
// Memory-bound: 8 B per matrix element, once.
#include "detcore.h"
#include "launch.h"

namespace pynqs {

constexpr int kGfmcMaxTiles = 1024;

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

template <int LEN>
__global__ __launch_bounds__(kBlock) void gfmc_sample_kernel(const double *__restrict__ gk, int64_t m, uint32_t tile, uint32_t ntiles,
                                                             const double *__restrict__ rnd, const uint64_t *__restrict__ comb,
                                                             int64_t *__restrict__ index, double *__restrict__ beta,
                                                             uint64_t *__restrict__ x_new) {
  __shared__ double tsum[kGfmcMaxTiles];
  __shared__ double s_target, s_before;
  __shared__ uint32_t s_tile;
  const int64_t walker = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double *__restrict__ row = gk + walker * m;
  // 1. tile sums: a wave per tile, lanes over consecutive columns
  for (uint32_t t = wave; t < ntiles; t += kBlock / 64) {
    const int64_t c0 = (int64_t)t * tile, c1 = min(c0 + (int64_t)tile, m);
    double acc = 0.0;
    for (int64_t c = c0 + lane; c < c1; c += 64) acc += row[c];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) tsum[t] = acc;
  }
  __syncthreads();
  // 2. scan of the tile sums by the first wave (kGfmcMaxTiles / 64 per lane, in tile order)
  if (wave == 0) {
    constexpr int kPer = kGfmcMaxTiles / 64;
    double loc[kPer], s = 0.0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const uint32_t t = lane * kPer + i;
      loc[i] = t < ntiles ? tsum[t] : 0.0;
      s += loc[i];
    }
    const double incl = wave_incl_scan(s, lane);
    const double total = __shfl(incl, 63);
    const double target = rnd[walker] * total;
    double before = incl - s;  // sum of the tiles of the lanes in front
    // first tile whose inclusive sum reaches the target (tiles in order: lanes in order, then i in order)
    uint32_t mine = 0xffffffffu;
    double mine_before = 0.0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const uint32_t t = lane * kPer + i;
      if (mine == 0xffffffffu && t < ntiles && before + loc[i] >= target) { mine = t; mine_before = before; }
      before += loc[i];
    }
    const uint64_t have = __ballot(mine != 0xffffffffu);
    const int first = have ? __ffsll((long long)have) - 1 : -1;
    if (lane == (first < 0 ? 0 : first)) {
      // first < 0: rounding put the target beyond the last partial sum -> last tile, whose scan then ends on the
      // row's last column
      s_tile = first < 0 ? ntiles - 1 : mine;
      s_before = first < 0 ? total - tsum[ntiles - 1] : mine_before;
      s_target = target;
    }
    if (lane == 0) beta[walker] = total;
  }
  __syncthreads();
  // 3. inside the tile, in column order: chunks of 64 columns by the first wave
  if (wave == 0) {
    const uint32_t t = s_tile;
    const double target = s_target;
    double run = s_before;
    const int64_t c0 = (int64_t)t * tile, c1 = min(c0 + (int64_t)tile, m);
    int64_t found = -1;
    for (int64_t c = c0; c < c1 && found < 0; c += 64) {
      const double v = c + lane < c1 ? row[c + lane] : 0.0;
      const double incl = wave_incl_scan(v, lane) + run;
      const uint64_t hit = __ballot(c + lane < c1 && incl >= target);
      if (hit) found = c + __ffsll((long long)hit) - 1;
      run = __shfl(incl, 63);
    }
    if (found < 0) found = c1 - 1;  // rounding at the very end of the row
    if (lane == 0) index[walker] = found;
    if (lane < LEN) x_new[walker * LEN + lane] = comb[(walker * m + found) * LEN + lane];
  }
}

}  // namespace pynqs

using namespace pynqs;

extern "C" int pynqs_gfmc_sample(const double *green, int64_t n, int64_t ncomb, const double *rand_num, const uint64_t *comb, int sorb,
                                 int64_t *index, double *beta, uint64_t *x_new, void *stream) {
  if (n < 0 || ncomb < 1 || sorb < 1 || sorb > kMaxSorb) return set_error(PYNQS_EINVAL, "bad n / ncomb / sorb");
  if (n == 0) return PYNQS_OK;
  if (!green || !rand_num || !comb || !index || !beta || !x_new) return set_error(PYNQS_EINVAL, "null pointer");
  if (n > 0x7fffffffll) return set_error(PYNQS_EINVAL, "n too large for one launch");
  // tiles of a multiple of 256 columns, at most kGfmcMaxTiles of them
  uint32_t tile = 256;
  while ((ncomb + tile - 1) / tile > kGfmcMaxTiles) tile *= 2;
  const uint32_t ntiles = (uint32_t)((ncomb + tile - 1) / tile);
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_LEN(len, hipLaunchKernelGGL((gfmc_sample_kernel<LEN>), dim3((uint32_t)n), dim3(kBlock), 0, st, green, ncomb, tile, ntiles,
                                       rand_num, comb, index, beta, x_new));
  return check_launch("gfmc_sample");
}
