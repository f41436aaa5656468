// kernels_eloc.hip -- fused local-energy kernels on the integral plan: the excitation list and its matrix
// elements are consumed on chip, comb / Hmat are never written to HBM.
//   eloc_sample_space_kernel : E_loc(x) = sum_x' <x|H|x'> psi(x') / psi(x), psi from a sorted sample table
//                              (vmc/energy/eloc.py:326-401 + utils/public_function.py:817-838)
//   reduce_count / reduce_emit : keep only |<x|H|x'>| >= eps (vmc/energy/eloc.py:297-298), compacted in
//                              ascending column order
// Matrix elements are bit-identical to kernels_plan.hip (same helpers, same order of additions).
#include "detcore.h"
#include "launch.h"
#include "plan.h"

namespace pynqs {

// One excitation of the walker: column k = rank + 1, value, ket.
template <int LEN, typename T>
struct Item {
  uint64_t ket[LEN];
  T h;
};

// Visits every column of the block's range in a fixed schedule: `rounds` of kBlock consecutive columns,
// lane `tid` of round i handles column lo + i*kBlock + tid.  f(col, valid, h, ket) is called by ALL lanes
// every round (valid == false past the end) so that f may use barriers and wave-wide operations.
// Singles and the diagonal need the workgroup-cooperative phases of kernels_plan.hip; here they are computed
// first into LDS (`hs` = one T per column < first_double) and replayed from there.
template <int LEN, typename T>
__device__ __forceinline__ T double_element(uint32_t r, const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                            const T *__restrict__ plan, const Walker<LEN> &wk, uint64_t (&ket)[LEN]) {
  const uint32_t K = (uint32_t)pl.K, NP = (uint32_t)pl.NP;
  int a, b, c, d;
  T v;
  uint32_t par;
  if (r < p.d3) {
    const bool beta = r >= p.d2;
    const uint32_t t = r - (beta ? p.d2 : p.d1);
    const uint32_t npair = beta ? p.noBB : p.noAA;
    const uint32_t ab = mdiv(t, beta ? p.divNoBB : p.divNoAA);
    uint32_t ij = t - ab * npair + (beta ? p.rotB : p.rotA);
    ij = ij >= npair ? ij - npair : ij;
    const uint32_t eh = L.tab[(beta ? p.offHPb : p.offHPa) + ij];
    const uint32_t ep = L.tab[(beta ? p.offPPb : p.offPPa) + ab];
    v = plan[pl.offVss + (size_t)(beta ? 1 : 0) * NP * NP + __umul24((ep >> 17) & 0x1fffu, NP) + ((eh >> 17) & 0x1fffu)];
    a = eh & 0xff; b = (eh >> 8) & 0xff; c = ep & 0xff; d = (ep >> 8) & 0xff;
    par = (((eh ^ ep) >> 16) & 1u) ^ (uint32_t)(a < c) ^ (uint32_t)(b < c) ^ (uint32_t)(a < d) ^ (uint32_t)(b < d);
  } else {
    const uint32_t t = r - p.d3;
    const uint32_t jb = mdiv(t, p.divNSa);
    const uint32_t ia = t - jb * (uint32_t)p.nSa;
    const uint32_t ea = L.tab[p.offSa + ia], eb = L.tab[p.offSb + jb];
    v = plan[pl.offVab + (size_t)__umul24(eb >> 17, K * K) + (ea >> 17)];
    a = ea & 0xff; c = (ea >> 8) & 0xff; b = eb & 0xff; d = (eb >> 8) & 0xff;
    par = (((ea ^ eb) >> 16) & 1u) ^ (uint32_t)(a < d) ^ (uint32_t)(b < c) ^ 1u;
  }
#pragma unroll
  for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
  toggle<LEN>(ket, a); toggle<LEN>(ket, b); toggle<LEN>(ket, c); toggle<LEN>(ket, d);
  return par ? -v : v;
}

// Computes <x|H|x'> of the singles [0, d1) and of the diagonal into LDS: hs[0] = <x|H|x>, hs[1 + r] = single r.
// Same arithmetic as comb_hij_plan_kernel.  `hs` must hold 1 + d1 values; uses L.scratch as staging.
template <int LEN, typename T>
__device__ __forceinline__ void singles_and_diag_to_lds(const SDParams &p, const PlanLayout &pl, const LdsLayout &L, int nocc,
                                                        const T *__restrict__ plan, T *__restrict__ hs) {
  const int tid = threadIdx.x;
  const uint32_t K = (uint32_t)pl.K;
  T *tile = reinterpret_cast<T *>(L.scratch);
  const int stride = nocc | 1;
  const int per_tile = max(1, min(kBlock, kDiagTile / stride));
  const int wave = tid >> 6, lane = tid & 63;
  const T *__restrict__ S2 = plan + pl.offS2;
  const T *__restrict__ S1 = plan + pl.offS1;
  for (uint32_t t0 = 0; t0 < p.d1; t0 += per_tile) {
    const int cnt = (int)min((uint32_t)per_tile, p.d1 - t0);
    __syncthreads();
    for (int sl = wave; sl < cnt; sl += kBlock / 64) {
      const uint32_t r = t0 + sl;
      const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
      const uint32_t spin = r >= p.d0;
      const uint32_t hm = (e & 0xff) >> 1, qm = ((e >> 8) & 0xff) >> 1;
      const T *__restrict__ rowp = S2 + ((size_t)(spin * K + hm) * K + qm) * p.sorb;
      for (int j = lane; j < nocc; j += 64) tile[sl * stride + j] = rowp[L.occv[j]];
    }
    __syncthreads();
    if (tid < cnt) {
      const uint32_t r = t0 + tid;
      const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
      const uint32_t spin = r >= p.d0;
      const int h = e & 0xff, q = (e >> 8) & 0xff;
      T acc = T(0);
      acc += S1[(size_t)(spin * K + (h >> 1)) * K + (q >> 1)];
      const T *__restrict__ mine = tile + tid * stride;
      for (int j = 0; j < nocc; ++j) acc += mine[j];
      hs[1 + r] = ((e >> 16) & 1u) ? -acc : acc;
    }
  }
  // diagonal (hamiltonian.cpp:41-48 order)
  const T *__restrict__ D1 = plan + pl.offD1;
  const T *__restrict__ D2 = plan + pl.offD2;
  const int nele = p.nele, nterms = nele * (nele + 1) / 2;
  T acc = T(0);
  for (int base = 0; base < nterms; base += kDiagTile) {
    const int end = min(base + kDiagTile, nterms);
    __syncthreads();
    for (int t = base + tid; t < end; t += kBlock) {
      int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > t) --a;
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      const int pos = t - a * (a + 1) / 2;
      const int pa = L.occa[a];
      tile[t - base] = pos == 0 ? D1[pa] : D2[pa * p.sorb + L.occa[pos - 1]];
    }
    __syncthreads();
    if (tid == kBlock - 1)
      for (int t = 0; t < end - base; ++t) acc += tile[t];
  }
  if (tid == kBlock - 1) hs[0] = acc;
  __syncthreads();
}

// column -> (h, ket) for any column of the walker, given hs for the singles/diagonal
template <int LEN, typename T>
__device__ __forceinline__ T column_element(uint32_t col, const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                            const T *__restrict__ plan, const Walker<LEN> &wk, const T *__restrict__ hs,
                                            uint64_t (&ket)[LEN]) {
  if (col == 0) {
#pragma unroll
    for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
    return hs[0];
  }
  const uint32_t r = col - 1;
  if (r < p.d1) {
    const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
#pragma unroll
    for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
    toggle<LEN>(ket, e & 0xff); toggle<LEN>(ket, (e >> 8) & 0xff);
    return hs[col];
  }
  return double_element<LEN, T>(r, p, pl, L, plan, wk, ket);
}

__host__ __device__ inline size_t lds_bytes_eloc(const SDParams &p, size_t elem) {
  // fixed part + staging tile + hs[1 + d1]
  return lds_bytes(p, elem) + (size_t)(p.d1 + 2) * elem;
}

// -------------------------------------------------------------------------------------------------
// SAMPLE_SPACE local energy.  acc[walker] += sum_cols h * psi(ket); the first chunk also stores psi(x).
// One workgroup per (walker, chunk); with more than one chunk per walker partial sums meet through
// float atomics (then the summation order, and the last bits, depend on arrival order).
template <int LEN, bool CPLX>
__global__ __launch_bounds__(kBlock) void eloc_sample_space_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                                   uint32_t nchunks, uint32_t chunk_len,
                                                                   const double *__restrict__ plan,
                                                                   const uint64_t *__restrict__ keys, int64_t nkeys,
                                                                   const double *__restrict__ wf, double *__restrict__ acc,
                                                                   double *__restrict__ psi0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double red[2][kBlock / 64];
  const uint64_t wg = blockIdx.x;
  const uint64_t walker = wg / nchunks;
  const uint32_t chunk = (uint32_t)(wg - walker * nchunks);
  const int tid = threadIdx.x;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  double *hs = reinterpret_cast<double *>(smem + lds_bytes(p, sizeof(double)));
  const uint32_t ncomb = p.nsd + 1;
  const uint32_t lo = chunk * chunk_len, hi = min(lo + chunk_len, ncomb);
  if (lo <= p.d1) singles_and_diag_to_lds<LEN, double>(p, pl, L, nocc, plan, hs);

  double re = 0.0, im = 0.0;
  for (uint32_t col = lo + tid; col < hi; col += kBlock) {
    uint64_t ket[LEN];
    const double h = column_element<LEN, double>(col, p, pl, L, plan, wk, hs, ket);
    const int64_t pos = lut_find<LEN>(keys, nkeys, ket);
    if (pos >= 0) {
      if constexpr (CPLX) { re += h * wf[2 * pos]; im += h * wf[2 * pos + 1]; }
      else re += h * wf[pos];
    }
    if (col == 0) {
      if constexpr (CPLX) { psi0[2 * walker] = pos >= 0 ? wf[2 * pos] : 0.0; psi0[2 * walker + 1] = pos >= 0 ? wf[2 * pos + 1] : 0.0; }
      else psi0[walker] = pos >= 0 ? wf[pos] : 0.0;
    }
  }
  // fixed-order reduction: lanes (xor butterfly), then waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    re += __shfl_xor(re, o);
    if constexpr (CPLX) im += __shfl_xor(im, o);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = re; red[1][tid >> 6] = im; }
  __syncthreads();
  if (tid == 0) {
    double sr = 0.0, si = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) { sr += red[0][w]; si += red[1][w]; }
    if (nchunks == 1) {
      if constexpr (CPLX) { acc[2 * walker] = sr; acc[2 * walker + 1] = si; }
      else acc[walker] = sr;
    } else {
      if constexpr (CPLX) { atomicAdd(acc + 2 * walker, sr); atomicAdd(acc + 2 * walker + 1, si); }
      else atomicAdd(acc + walker, sr);
    }
  }
}

// eloc = acc / psi0 (complex division when CPLX), in place on acc
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void eloc_divide_kernel(double *__restrict__ acc, const double *__restrict__ psi0, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if constexpr (CPLX) {
    const double ar = acc[2 * i], ai = acc[2 * i + 1], br = psi0[2 * i], bi = psi0[2 * i + 1];
    const double d = br * br + bi * bi;
    acc[2 * i] = (ar * br + ai * bi) / d;
    acc[2 * i + 1] = (ai * br - ar * bi) / d;
  } else {
    acc[i] = acc[i] / psi0[i];
  }
}

// -------------------------------------------------------------------------------------------------
// REDUCE front end: keep |h| >= eps.  One workgroup per walker; columns are visited in rounds of kBlock
// consecutive columns so that a workgroup-wide exclusive scan of the keep flags gives ascending positions.
template <int LEN, typename T, bool EMIT>
__global__ __launch_bounds__(kBlock) void reduce_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                        const T *__restrict__ plan, T eps, int64_t *__restrict__ counts,
                                                        const int64_t *__restrict__ offsets, int32_t *__restrict__ kept_col,
                                                        uint64_t *__restrict__ kept_onv, T *__restrict__ kept_h) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const uint64_t walker = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  T *hs = reinterpret_cast<T *>(smem + lds_bytes(p, sizeof(T)));
  singles_and_diag_to_lds<LEN, T>(p, pl, L, nocc, plan, hs);
  const uint32_t ncomb = p.nsd + 1;
  uint64_t base = EMIT ? (uint64_t)offsets[walker] : 0;
  uint32_t total = 0;
  for (uint32_t c0 = 0; c0 < ncomb; c0 += kBlock) {
    const uint32_t col = c0 + tid;
    uint64_t ket[LEN];
    T h = T(0);
    bool keep = false;
    if (col < ncomb) {
      h = column_element<LEN, T>(col, p, pl, L, plan, wk, hs, ket);
      keep = fabs(h) >= eps;
    }
    const uint64_t m = __ballot(keep);
    if constexpr (EMIT) {
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) { const uint32_t c = wave_cnt[w]; all += c; if (w < wave) before += c; }
      if (keep) {
        const uint64_t pos = base + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        kept_col[pos] = (int32_t)col;
        kept_h[pos] = h;
#pragma unroll
        for (int i = 0; i < LEN; ++i) kept_onv[pos * LEN + i] = ket[i];
      }
      base += all;
      __syncthreads();
    } else {
      total += (uint32_t)__popcll(m);  // identical in every lane of the wave
    }
  }
  if constexpr (!EMIT) {
    if (lane == 0) wave_cnt[wave] = total;
    __syncthreads();
    if (tid == 0) {
      uint32_t s = 0;
      for (int w = 0; w < kBlock / 64; ++w) s += wave_cnt[w];
      counts[walker] = s;
    }
  }
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

#define DISPATCH_LEN(len, ...)                                  \
  switch (len) {                                                \
    case 1: { constexpr int LEN = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int LEN = 2; __VA_ARGS__; } break;      \
    default: { constexpr int LEN = 3; __VA_ARGS__; } break;     \
  }

static int eloc_common_checks(int sorb, int nele, int noA, int noB, int64_t nbatch, SDParams *p, PlanLayout *pl) {
  if (!make_sd_params(sorb, nele, noA, noB, p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "bad nbatch");
  return PYNQS_OK;
}

extern "C" int pynqs_eloc_sample_space(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                       const void *plan, const uint64_t *keys, int64_t nkeys, const double *wf,
                                       int wf_is_complex, double *eloc, double *psi0, void *stream) {
  SDParams p;
  PlanLayout pl;
  int rc = eloc_common_checks(sorb, nele, noA, noB, nbatch, &p, &pl);
  if (rc != PYNQS_OK) return rc;
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !eloc || !psi0 || nkeys < 0 || (nkeys > 0 && (!keys || !wf))) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, p.nsd + 1, &nchunks, &chunk_len);
  const size_t lds = lds_bytes_eloc(p, sizeof(double));
  if (lds > 64 * 1024) return set_error(PYNQS_EINVAL, "too many single excitations for the LDS staging buffer");
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  const size_t esz = wf_is_complex ? 16 : 8;
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");
  const double *pd = (const double *)plan;
  DISPATCH_LEN(len, {
    if (wf_is_complex)
      hipLaunchKernelGGL((eloc_sample_space_kernel<LEN, true>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, pd, keys, nkeys, wf, eloc, psi0);
    else
      hipLaunchKernelGGL((eloc_sample_space_kernel<LEN, false>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, pd, keys, nkeys, wf, eloc, psi0);
  });
  const uint32_t g2 = (uint32_t)((nbatch + kBlock - 1) / kBlock);
  if (wf_is_complex) hipLaunchKernelGGL((eloc_divide_kernel<true>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  else hipLaunchKernelGGL((eloc_divide_kernel<false>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  return check_launch("eloc_sample_space");
}

template <bool EMIT>
static int launch_reduce(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan, int dtype,
                         double eps, int64_t *counts, const int64_t *offsets, int32_t *kept_col, uint64_t *kept_onv, void *kept_h,
                         void *stream) {
  SDParams p;
  PlanLayout pl;
  int rc = eloc_common_checks(sorb, nele, noA, noB, nbatch, &p, &pl);
  if (rc != PYNQS_OK) return rc;
  if (dtype != PYNQS_F32 && dtype != PYNQS_F64) return set_error(PYNQS_EINVAL, "bad dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  if (EMIT ? (!offsets || !kept_col || !kept_onv || !kept_h) : !counts) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  const size_t lds = lds_bytes_eloc(p, esz);
  if (lds > 64 * 1024) return set_error(PYNQS_EINVAL, "too many single excitations for the LDS staging buffer");
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64)
      hipLaunchKernelGGL((reduce_kernel<LEN, double, EMIT>), dim3((uint32_t)nbatch), dim3(kBlock), lds, st, bra, p, pl,
                         (const double *)plan, eps, counts, offsets, kept_col, kept_onv, (double *)kept_h);
    else
      hipLaunchKernelGGL((reduce_kernel<LEN, float, EMIT>), dim3((uint32_t)nbatch), dim3(kBlock), lds, st, bra, p, pl,
                         (const float *)plan, (float)eps, counts, offsets, kept_col, kept_onv, (float *)kept_h);
  });
  return check_launch(EMIT ? "reduce_emit" : "reduce_count");
}

extern "C" int pynqs_reduce_count(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                  int dtype, double eps, int64_t *counts, void *stream) {
  return launch_reduce<false>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, counts, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int pynqs_reduce_emit(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                 int dtype, double eps, const int64_t *offsets, int32_t *kept_col, uint64_t *kept_onv,
                                 void *kept_h, void *stream) {
  return launch_reduce<true>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, nullptr, offsets, kept_col, kept_onv, kept_h, stream);
}
