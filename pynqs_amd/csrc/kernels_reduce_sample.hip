// kernels_reduce_sample.hip -- the semi-stochastic part of the REDUCE local energy (vmc/energy/eloc.py:257-296, the
// Fe2S2 example's eps = 1e-2, eps_sample = 1000) without materialising the [nbatch, ncomb] matrices.
//
// Reference: columns with |<x|H|x'>| >= eps are kept; from the others N = eps_sample columns are drawn with
// replacement, p_m = |H_m| / S (S = sum of the sub-eps |H|, torch.multinomial), and a column drawn c times enters
// with the weight (c / N) H_m / p_m = (c / N) sign(H_m) S.
//
// Here the multinomial is drawn hierarchically, which is the same distribution:
//   1. pynqs_reduce_count_sums : per tile of a walker's row (plan_tiles.h) the kept count and the sub-eps sum s_t
//   2. host (torch)            : N draws over the tiles with p_t = s_t / S  -> draws per tile, offsets
//   3. pynqs_reduce_sample     : a wave re-visits its tile, builds the running sum of the sub-eps |H| in the tile's
//                                fixed column order in LDS, draws its share of uniforms from a counter-based
//                                generator keyed (seed, walker, chunk, tile, k), finds each one's column by binary
//                                search, counts hits per column (LDS atomics) and emits one record per distinct
//                                column: (column, ket, sign(H) * hits * S / N).
// The kept columns are emitted by pynqs_reduce_emit as before.  Nothing of size nbatch x ncomb exists at any point.
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "plan_tiles.h"

namespace pynqs {

constexpr int kTileCols = 128 * PYNQS_U;  // columns of the largest tile

__device__ __forceinline__ uint64_t rs_mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// inclusive scan over the lanes 0 .. (active prefix); lanes must be a contiguous set starting at 0
__device__ __forceinline__ double rs_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// ---- pass 1: kept count and sub-eps sum per tile ---------------------------------------------------------------
template <int LEN, typename T>
struct CountSumSink {
  T eps;
  uint32_t *__restrict__ tile_counts;
  double *__restrict__ tile_sums;
  uint32_t tile;
  uint32_t cnt;  // per lane
  double sub;    // per lane
  __device__ __forceinline__ void add(T h) {
    const T a = fabs(h);
    if (a >= eps) ++cnt;
    else sub += (double)a;
  }
  __device__ __forceinline__ void flush() {  // wave-uniform
    if (tile == 0xffffffffu) return;
    uint32_t c = cnt;
    double s = sub;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { c += __shfl_xor(c, d); s += __shfl_xor(s, d); }
    if ((threadIdx.x & 63) == 0) { tile_counts[tile] = c; tile_sums[tile] = s; }
  }
  __device__ __forceinline__ void tile_begin(uint32_t t) { flush(); tile = t; cnt = 0; sub = 0.0; }
  __device__ __forceinline__ void one(uint32_t, T h, const uint64_t (&)[LEN]) { add(h); }
  __device__ __forceinline__ void pair(uint32_t, T h0, T h1, const uint64_t (&)[LEN], const uint64_t (&)[LEN]) { add(h0); add(h1); }
  __device__ __forceinline__ void two(uint32_t, T h0, const uint64_t (&)[LEN], uint32_t, T h1, const uint64_t (&)[LEN]) { add(h0); add(h1); }
};

// ---- pass 3: the draws ---------------------------------------------------------------------------------------------
// Wave-private LDS: prefix[kTileCols] running sums, cs[kTileCols] column | sign << 31, hits[kTileCols], and two words
// (number of columns so far, as uint32; running sum, as double) that outlive the divergent code some columns come from.
struct SampleLds {
  double *prefix;
  uint32_t *cs;
  uint32_t *hits;
  volatile uint32_t *ncols;
  volatile double *run;
};

template <int LEN, typename T>
struct SampleSink {
  T eps;
  SampleLds S;
  const SDParams *p;
  const LdsLayout *L;
  const Walker<LEN> *wk;
  const int32_t *__restrict__ tile_draws;   // this workgroup's slice
  const int64_t *__restrict__ sample_off;   // this workgroup's slice
  double scale;                             // S_walker / N
  uint64_t key;                             // seed mixed with (walker, chunk)
  int32_t *__restrict__ s_col;
  uint64_t *__restrict__ s_onv;
  T *__restrict__ s_h;
  uint32_t tile;

  // column `col` with matrix element h becomes entry `idx` of the tile, the running sum before it being `before`
  __device__ __forceinline__ void entry(uint32_t idx, uint32_t col, T h, double incl) const {
    S.prefix[idx] = incl;
    S.cs[idx] = col | (h < T(0) ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ double width(T h) const {
    const T a = fabs(h);
    return a >= eps ? 0.0 : (double)a;
  }
  // any set of active lanes (tile 0, singles): one lane after the other
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(1);
    const double w = width(h);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (lane == b) {
        const uint32_t idx = *S.ncols;
        const double incl = *S.run + w;
        entry(idx, col, h, incl);
        *S.ncols = idx + 1;
        *S.run = incl;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // lanes 0 .. k-1 active (k = 64 in full tiles); column order (lane, v): col, col + 1
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&)[LEN], const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const uint32_t nact = (uint32_t)__popcll(__ballot(1));
    const double w0 = width(h0), w1 = width(h1);
    const double incl = rs_scan(w0 + w1, lane);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + 2 * lane, col, h0, run + incl - w1);
    entry(base + 2 * lane + 1, col + 1, h1, run + incl);
    const double total = __shfl(incl, (int)nact - 1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 2 * nact; *S.run = run + total; }
    __builtin_amdgcn_wave_barrier();
  }
  // all 64 lanes active; column order: the c0 of every lane, then the c1
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&)[LEN], uint32_t c1, T h1, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const double w0 = width(h0), w1 = width(h1);
    const double i0 = rs_scan(w0, lane), t0 = __shfl(i0, 63);
    const double i1 = rs_scan(w1, lane), t1 = __shfl(i1, 63);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + lane, c0, h0, run + i0);
    entry(base + 64 + lane, c1, h1, run + t0 + i1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 128; *S.run = run + t0 + t1; }
    __builtin_amdgcn_wave_barrier();
  }

  __device__ __forceinline__ void flush() {  // wave-uniform
    if (tile == 0xffffffffu) return;
    const int lane = threadIdx.x & 63;
    const uint32_t draws = (uint32_t)tile_draws[tile];
    const uint32_t ncols = *S.ncols;
    const double total = *S.run;
    if (draws == 0 || ncols == 0 || !(total > 0.0)) return;
    // draw, locate, count
    for (uint32_t k = lane; k < draws; k += 64) {
      const uint64_t r = rs_mix64(key ^ rs_mix64(((uint64_t)tile << 32) | k));
      const double target = (double)(r >> 11) * 0x1.0p-53 * total;
      uint32_t lo = 0, hi = ncols;  // first idx with prefix[idx] > target
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.prefix[mid] > target) hi = mid; else lo = mid + 1;
      }
      if (lo >= ncols) lo = ncols - 1;
      while (lo > 0 && !(S.prefix[lo] > S.prefix[lo - 1])) --lo;  // rounding at the end: back to a column of positive width
      atomicAdd(&S.hits[lo], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    // one record per distinct column, in tile order
    int64_t pos = sample_off[tile];
    for (uint32_t i0 = 0; i0 < ncols; i0 += 64) {
      const uint32_t idx = i0 + lane;
      const uint32_t hc = idx < ncols ? S.hits[idx] : 0u;
      const uint64_t m = __ballot(hc != 0u);
      if (hc) {
        const uint32_t e = S.cs[idx], col = e & 0x7fffffffu;
        const int64_t at = pos + __popcll(m & ((1ull << lane) - 1ull));
        uint64_t ket[LEN];
        if (col == 0) {
#pragma unroll
          for (int i = 0; i < LEN; ++i) ket[i] = wk->w[i];
        } else {
          const Excitation x = decode(col - 1, *p, *L);
          make_ket<LEN>(*wk, x, ket);
        }
        s_col[at] = (int32_t)col;
        const double v = scale * (double)hc;
        s_h[at] = (T)((e >> 31) ? -v : v);
#pragma unroll
        for (int i = 0; i < LEN; ++i) s_onv[at * LEN + i] = ket[i];
      }
      pos += __popcll(m);
    }
  }
  // a tile without draws is not enumerated at all (plan_tiles.h); with 1000 draws over the 4650 tiles of a sorb-120 row
  // that is four tiles out of five
  __device__ __forceinline__ bool skip_tile(uint32_t t) const { return tile_draws[t] == 0; }
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush();
    tile = t;
    if (tile_draws[t] == 0) return;
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kTileCols; i += 64) S.hits[i] = 0u;
    if (lane == 0) { *S.ncols = 0u; *S.run = 0.0; }
    __builtin_amdgcn_wave_barrier();
  }
};

constexpr size_t kSampleLdsPerWave = (size_t)kTileCols * (8 + 4 + 4) + 16;

template <int LEN, typename T, bool SAMPLE>
__global__ __launch_bounds__(kBlock) void reduce_sample_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                               uint32_t nchunks, uint32_t chunk_len, uint32_t max_tiles,
                                                               const T *__restrict__ plan, T eps, uint32_t *__restrict__ tile_counts,
                                                               double *__restrict__ tile_sums, const int32_t *__restrict__ tile_draws,
                                                               const int64_t *__restrict__ sample_off,
                                                               const double *__restrict__ walker_scale, uint64_t seed,
                                                               int32_t *__restrict__ s_col, uint64_t *__restrict__ s_onv,
                                                               T *__restrict__ s_h) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t next_tile;
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, false, walker, chunk);
  const uint64_t slot = walker * nchunks + chunk;
  const int tid = threadIdx.x;
  if (tid == 0) next_tile = 0;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  if constexpr (SAMPLE) {
    unsigned char *mine = smem + ((lds_bytes(p, sizeof(T)) + 15) & ~(size_t)15) + (size_t)(tid >> 6) * kSampleLdsPerWave;
    SampleLds S;
    S.prefix = reinterpret_cast<double *>(mine);
    S.run = reinterpret_cast<volatile double *>(mine + (size_t)kTileCols * 8);
    S.cs = reinterpret_cast<uint32_t *>(mine + (size_t)kTileCols * 8 + 8);
    S.hits = S.cs + kTileCols;
    S.ncols = reinterpret_cast<volatile uint32_t *>(S.hits + kTileCols);
    SampleSink<LEN, T> sink{eps, S, &p, &L, &wk, tile_draws + slot * max_tiles, sample_off + slot * max_tiles, walker_scale[walker],
                            rs_mix64(seed ^ rs_mix64(slot)), s_col, s_onv, s_h, 0xffffffffu};
    visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
    sink.flush();
  } else {
    CountSumSink<LEN, T> sink{eps, tile_counts + slot * max_tiles, tile_sums + slot * max_tiles, 0xffffffffu, 0u, 0.0};
    visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
    sink.flush();
  }
}

}  // namespace pynqs

using namespace pynqs;

template <bool SAMPLE>
static int launch_rs(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan, int dtype, double eps,
                     uint32_t *tile_counts, double *tile_sums, const int32_t *tile_draws, const int64_t *sample_off,
                     const double *walker_scale, uint64_t seed, int32_t *s_col, uint64_t *s_onv, void *s_h, void *stream) {
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || nbatch > 0x7fffffffll || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return set_error(PYNQS_EINVAL, "bad nbatch/dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  if (SAMPLE ? (!tile_draws || !sample_off || !walker_scale || !s_col || !s_onv || !s_h) : (!tile_counts || !tile_sums))
    return set_error(PYNQS_EINVAL, "null pointer");
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, p.nsd + 1, &nchunks, &chunk_len);
  const uint32_t max_tiles = max_tiles_per_chunk(p, nchunks, chunk_len);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  const int len = (sorb - 1) / 64 + 1;
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  const size_t lds = ((lds_bytes(p, esz) + 15) & ~(size_t)15) + (SAMPLE ? (kBlock / 64) * kSampleLdsPerWave : 0);
  hipStream_t st = (hipStream_t)stream;
  if (!SAMPLE) {
    if (hipMemsetAsync(tile_counts, 0, 4 * (size_t)grid * max_tiles, st) != hipSuccess) return check_launch("memset");
    if (hipMemsetAsync(tile_sums, 0, 8 * (size_t)grid * max_tiles, st) != hipSuccess) return check_launch("memset");
  }
  DISPATCH_LEN(len, {
    if (lds > 64 * 1024) {  // beyond the default dynamic LDS limit (large tables, sorb >~ 150)
      const void *fn = dtype == PYNQS_F64 ? reinterpret_cast<const void *>(&reduce_sample_kernel<LEN, double, SAMPLE>)
                                          : reinterpret_cast<const void *>(&reduce_sample_kernel<LEN, float, SAMPLE>);
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("hipFuncSetAttribute");
    }
    if (dtype == PYNQS_F64)
      hipLaunchKernelGGL((reduce_sample_kernel<LEN, double, SAMPLE>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, max_tiles, (const double *)plan, eps, tile_counts, tile_sums, tile_draws, sample_off, walker_scale,
                         seed, s_col, s_onv, (double *)s_h);
    else
      hipLaunchKernelGGL((reduce_sample_kernel<LEN, float, SAMPLE>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, max_tiles, (const float *)plan, (float)eps, tile_counts, tile_sums, tile_draws, sample_off,
                         walker_scale, seed, s_col, s_onv, (float *)s_h);
  });
  return check_launch(SAMPLE ? "reduce_sample" : "reduce_count_sums");
}

extern "C" int pynqs_reduce_count_sums(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                       int dtype, double eps, uint32_t *tile_counts, double *tile_sums, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return launch_rs<false>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, tile_counts, tile_sums, nullptr, nullptr, nullptr, 0,
                          nullptr, nullptr, nullptr, stream);
}

extern "C" int pynqs_reduce_sample(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                   int dtype, double eps, const int32_t *tile_draws, const int64_t *sample_offsets,
                                   const double *walker_scale, uint64_t seed, int32_t *s_col, uint64_t *s_onv, void *s_h,
                                   void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return launch_rs<true>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, nullptr, nullptr, tile_draws, sample_offsets,
                         walker_scale, seed, s_col, s_onv, s_h, stream);
}
