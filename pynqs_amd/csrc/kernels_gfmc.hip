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

// FROM_RANK: there is no comb array (the row came from the fused Green's-function kernel, pynqs_green_rbm): `comb` then holds the
// n walkers themselves and x_new is the excitation of rank index - 1 of the walker (column 0 = the walker).
template <int LEN, bool FROM_RANK>
__global__ __launch_bounds__(kBlock) void gfmc_sample_kernel(const double *__restrict__ gk, int64_t m, uint32_t tile, uint32_t ntiles,
                                                             const double *__restrict__ rnd, const uint64_t *__restrict__ comb, SDParams p,
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
    int64_t found = -1, last_pos = -1;
    for (int64_t c = c0; c < c1 && found < 0; c += 64) {
      const double v = c + lane < c1 ? row[c + lane] : 0.0;
      const double incl = wave_incl_scan(v, lane) + run;
      const uint64_t hit = __ballot(c + lane < c1 && incl >= target);
      if (hit) found = c + __ffsll((long long)hit) - 1;
      const uint64_t pos = __ballot(v > 0.0);
      if (pos) last_pos = c + 63 - __clzll((long long)pos);
      run = __shfl(incl, 63);
    }
    // The tile was chosen from lane-strided partial sums, the scan above adds in column order: when the two roundings
    // disagree at the tile's end the target is "just behind" the tile's last weight -> take the last column that HAS
    // weight (never a zero-weight column, which the reference's searchsorted cannot return either)
    if (found < 0) found = last_pos >= 0 ? last_pos : c1 - 1;
    if (lane == 0) index[walker] = found;
    if constexpr (FROM_RANK) {
      if (lane == 0) {
        uint64_t x[LEN];
#pragma unroll
        for (int i = 0; i < LEN; ++i) x[i] = comb[walker * LEN + i];
        if (found > 0) excite_by_rank<LEN>(x, (uint32_t)(found - 1), p);
#pragma unroll
        for (int i = 0; i < LEN; ++i) x_new[walker * LEN + i] = x[i];
      }
    } else {
      if (lane < LEN) x_new[walker * LEN + lane] = comb[(walker * m + found) * LEN + lane];
    }
  }
}

}  // namespace pynqs

using namespace pynqs;

template <bool FROM_RANK>
static int gfmc_sample_impl(const double *green, int64_t n, int64_t ncomb, const double *rand_num, const uint64_t *comb, int sorb,
                            const SDParams &p, int64_t *index, double *beta, uint64_t *x_new, void *stream) {
  pynqs::DeviceScope device_scope_(green);
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
  DISPATCH_LEN(len, hipLaunchKernelGGL((gfmc_sample_kernel<LEN, FROM_RANK>), dim3((uint32_t)n), dim3(kBlock), 0, st, green, ncomb, tile, ntiles,
                                       rand_num, comb, p, index, beta, x_new));
  return check_launch("gfmc_sample");
}

extern "C" int pynqs_gfmc_sample(const double *green, int64_t n, int64_t ncomb, const double *rand_num, const uint64_t *comb, int sorb,
                                 int64_t *index, double *beta, uint64_t *x_new, void *stream) {
  SDParams p = {};
  return gfmc_sample_impl<false>(green, n, ncomb, rand_num, comb, sorb, p, index, beta, x_new, stream);
}

extern "C" int pynqs_gfmc_sample_rank(const double *green, int64_t n, const double *rand_num, const uint64_t *bra, int sorb, int nele,
                                      int noA, int noB, int64_t *index, double *beta, uint64_t *x_new, void *stream) {
  SDParams p;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  return gfmc_sample_impl<true>(green, n, (int64_t)p.nsd + 1, rand_num, bra, sorb, p, index, beta, x_new, stream);
}

// -------------------------------------------------------------------------------------------------
// Weighted moments of the local energies for the statistics all-reduce (utils/stats/dist_stats.py:18-79):
//   out[0..3] = sum_i p_i Re x_i,  sum_i p_i Im x_i,  sum_i p_i |x_i|^2,  sum_i p_i
// in one pass and a fixed order of additions (per-lane strided sums, wave butterfly, waves, then the blocks' partial
// sums by the last block to finish): the ~10 small torch kernels this replaces cost 0.2 ms per step, a third of the
// fused sample-space step.  out has room for 4 * (kMomentBlocks + 1) doubles (the partials follow the result) and
// one counter word.
namespace pynqs {

constexpr int kMomentBlocks = 128;

template <bool CPLX>
__global__ __launch_bounds__(kBlock) void moments_kernel(const double *__restrict__ x, const double *__restrict__ prob, int64_t n,
                                                         double *__restrict__ out, unsigned int *__restrict__ done) {
  __shared__ double red[4][kBlock / 64];
  __shared__ bool last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + tid; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double p = prob[i];
    const double re = CPLX ? x[2 * i] : x[i], im = CPLX ? x[2 * i + 1] : 0.0;
    s[0] += p * re; s[1] += p * im; s[2] += p * (re * re + im * im); s[3] += p;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s[k] += __shfl_xor(s[k], d);
    if (lane == 0) red[k][wave] = s[k];
  }
  __syncthreads();
  if (tid < 4) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) t += red[tid][w];
    out[4 * (1 + blockIdx.x) + tid] = t;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) last = atomicAdd(done, 1u) == gridDim.x - 1;
  __syncthreads();
  if (last) {
    __threadfence();
    if (tid < 4) {
      double t = 0.0;
      for (unsigned b = 0; b < gridDim.x; ++b) t += out[4 * (1 + b) + tid];  // block order: reproducible
      out[tid] = t;
    }
    if (tid == 0) *done = 0;  // ready for the next call
  }
}

}  // namespace pynqs

namespace pynqs {
// mean, var = E|O|^2 - |E O|^2 (2 - sum p), sd, se from the (all-reduced) moments: one thread instead of ten torch launches
__global__ void stats_finish_kernel(const double *__restrict__ m, double inv_world, double counts, double *__restrict__ out) {
  const double re = m[0] * inv_world, im = m[1] * inv_world, m2 = m[2] * inv_world, ps = m[3] * inv_world;
  double var = m2 - (re * re + im * im) * (2.0 - ps);
  var = var > 0.0 ? var : 0.0;
  const double sd = sqrt(var);
  out[0] = re; out[1] = im; out[2] = var; out[3] = sd; out[4] = sd / sqrt(counts);
}
}  // namespace pynqs

extern "C" int pynqs_stats_finish(const double *moments, double inv_world, double counts, double *out5, void *stream) {
  pynqs::DeviceScope device_scope_(moments);
  if (!moments || !out5 || !(counts > 0.0)) return set_error(PYNQS_EINVAL, "bad arguments");
  hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, moments, inv_world, counts, out5);
  return check_launch("stats_finish");
}

extern "C" int64_t pynqs_moments_workspace(void) { return 8 * 4 * (pynqs::kMomentBlocks + 1) + 8; }

extern "C" int pynqs_weighted_moments(const double *x, int is_complex, const double *prob, int64_t n, void *workspace, void *stream) {
  pynqs::DeviceScope device_scope_(x);
  if (n < 0) return set_error(PYNQS_EINVAL, "bad n");
  if (!workspace || (n > 0 && (!x || !prob))) return set_error(PYNQS_EINVAL, "null pointer");
  double *out = (double *)workspace;
  unsigned int *done = (unsigned int *)(out + 4 * (kMomentBlocks + 1));
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > kMomentBlocks) blocks = kMomentBlocks;
  if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  if (is_complex) hipLaunchKernelGGL((moments_kernel<true>), dim3((uint32_t)blocks), dim3(kBlock), 0, st, x, prob, n, out, done);
  else hipLaunchKernelGGL((moments_kernel<false>), dim3((uint32_t)blocks), dim3(kBlock), 0, st, x, prob, n, out, done);
  return check_launch("weighted_moments");
}
