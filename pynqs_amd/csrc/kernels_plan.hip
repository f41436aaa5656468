// kernels_plan.hip -- the fast path of libpynqs_amd: enumeration + matrix elements on the integral plan
// (plan.h).  Same results, bit for bit, as kernels_det.hip / the reference CPU path
// (cpp_src/cpu/excitation.cpp:125-169, hamiltonian.cpp:34-50); what changes is where the bytes live:
//   * doubles read one element of the dense spin-blocked tables whose fast index follows the lanes;
//   * singles gather their nele+1 terms with lanes over the occupied orbitals (one contiguous table row
//     per single), stage them in LDS and add them in the reference's order;
//   * each excitation class has its own loop, so class parameters are scalar and no lane diverges.
#include "detcore.h"
#include "launch.h"
#include "plan.h"

namespace pynqs {

// -------------------------------------------------------------------------------------------------
// Plan construction: one lane per table element; every element is h2e/h1e[...] or its negation.
template <typename T>
__global__ __launch_bounds__(kBlock) void plan_build_kernel(const T *__restrict__ h1e, const T *__restrict__ h2e, PlanLayout pl,
                                                            T *__restrict__ plan) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= pl.total) return;
  const int64_t K = pl.K, NP = pl.NP, sorb = pl.sorb;
  T v = T(0);
  if (e < pl.offVss) {  // Vab[pb][pj][pa][pi]
    int64_t r = e;
    const int pi = (int)(r % K); r /= K;
    const int pa = (int)(r % K); r /= K;
    const int pj = (int)(r % K); r /= K;
    const int pb = (int)r;
    const int hA = 2 * pi, hB = 2 * pj + 1, qA = 2 * pa, qB = 2 * pb + 1;
    const int p0 = max(hA, hB), p1 = min(hA, hB), q0 = max(qA, qB), q1 = min(qA, qB);
    v = two_body<T>(h2e, p0, p1, q0, q1);
  } else if (e < pl.offS2) {  // Vss[spin][ab][ij]
    int64_t r = e - pl.offVss;
    const int spin = (int)(r / (NP * NP)); r -= (int64_t)spin * NP * NP;
    const int ab = (int)(r / NP), ij = (int)(r - (int64_t)ab * NP);
    int m1, m0, n1, n0;
    pair_unrank(ij, m1, m0);
    pair_unrank(ab, n1, n0);
    v = two_body<T>(h2e, 2 * m1 + spin, 2 * m0 + spin, 2 * n1 + spin, 2 * n0 + spin);
  } else if (e < pl.offS1) {  // S2[spin][pm][qm][k]
    int64_t r = e - pl.offS2;
    const int k = (int)(r % sorb); r /= sorb;
    const int qm = (int)(r % K); r /= K;
    const int pm = (int)(r % K); r /= K;
    const int spin = (int)r;
    const int pp = 2 * pm + spin, q = 2 * qm + spin;
    v = two_body<T>(h2e, pp, k, q, k);
  } else if (e < pl.offD2) {  // S1[spin][pm][qm] = h1e_get(p, q) = h1e[q*sorb + p]
    int64_t r = e - pl.offS1;
    const int qm = (int)(r % K); r /= K;
    const int pm = (int)(r % K); r /= K;
    const int spin = (int)r;
    v = h1e[(int64_t)(2 * qm + spin) * sorb + (2 * pm + spin)];
  } else if (e < pl.offD1) {  // D2[p][q] = <pq||pq>
    const int64_t r = e - pl.offD2;
    const int pp = (int)(r / sorb), q = (int)(r % sorb);
    v = two_body<T>(h2e, pp, q, pp, q);
  } else if (e < pl.offD1 + sorb) {
    const int64_t pp = e - pl.offD1;
    v = h1e[pp * sorb + pp];
  }
  plan[e] = v;
}

// -------------------------------------------------------------------------------------------------
template <int LEN>
__device__ __forceinline__ void ket_from(const Walker<LEN> &wk, int a, int b, int c, int d, bool four, uint64_t (&ket)[LEN]) {
#pragma unroll
  for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
  toggle<LEN>(ket, a);
  toggle<LEN>(ket, b);
  if (four) {
    toggle<LEN>(ket, c);
    toggle<LEN>(ket, d);
  }
}

// Output stores with a 32-bit lane offset on top of the walker's (wave-uniform) row pointer: one shift
// instead of 64-bit address arithmetic per lane.  A row is < 2^32 bytes (ncomb < 2^24, <= 24 B per ket).
template <typename T>
__device__ __forceinline__ void store_h(T *__restrict__ hrow, uint32_t col, T v) {
  *reinterpret_cast<T *>(reinterpret_cast<char *>(hrow) + (size_t)(col * (uint32_t)sizeof(T))) = v;
}
template <int LEN>
__device__ __forceinline__ void store_ket(uint64_t *__restrict__ crow, uint32_t col, const uint64_t (&ket)[LEN]) {
  uint64_t *dst = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(crow) + (size_t)(col * (uint32_t)(8 * LEN)));
#pragma unroll
  for (int i = 0; i < LEN; ++i) dst[i] = ket[i];
}

// Diagonal element from the plan's D1/D2 (same order of additions as hamiltonian.cpp:41-48).
template <typename T>
__device__ __forceinline__ void diag_phase_plan(const SDParams &p, const LdsLayout &L, const T *__restrict__ D1,
                                                const T *__restrict__ D2, T *__restrict__ out) {
  T *tile = reinterpret_cast<T *>(L.scratch);
  const int tid = threadIdx.x;
  const int nele = p.nele;
  const int nterms = nele * (nele + 1) / 2;
  T acc = T(0);
  for (int base = 0; base < nterms; base += kDiagTile) {
    const int end = min(base + kDiagTile, nterms);
    if (base) __syncthreads();
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
  if (tid == kBlock - 1) *out = acc;
}

template <int LEN, typename T, bool WRITE_COMB>
__global__ __launch_bounds__(kBlock) void comb_hij_plan_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                               uint32_t nchunks, uint32_t chunk_len,
                                                               const T *__restrict__ plan, uint64_t *__restrict__ comb,
                                                               T *__restrict__ hmat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t wg = blockIdx.x;
  const uint64_t walker = wg / nchunks;
  const uint32_t chunk = (uint32_t)(wg - walker * nchunks);
  const int tid = threadIdx.x;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);

  const uint32_t ncomb = p.nsd + 1;
  const uint32_t lo = chunk * chunk_len;
  const uint32_t hi = min(lo + chunk_len, ncomb);
  // excitation ranks handled here: [rlo, rhi) (column k = rank + 1)
  const uint32_t rlo = lo == 0 ? 0 : lo - 1, rhi = hi - 1;
  T *__restrict__ hrow = hmat + (size_t)walker * ncomb;
  uint64_t *__restrict__ crow = comb + (size_t)walker * ncomb * LEN;
  const uint32_t K = (uint32_t)pl.K, NP = (uint32_t)pl.NP;

  // ---- singles: ranks [0, d1) --------------------------------------------------------------------------
  {
    const uint32_t s_lo = rlo, s_hi = min(rhi, p.d1);
    if (s_lo < s_hi) {
      T *tile = reinterpret_cast<T *>(L.scratch);
      const int stride = nocc | 1;  // odd: conflict-free column reads in the summation
      const int per_tile = max(1, min(kBlock, kDiagTile / stride));  // one summing lane per staged single
      const int wave = tid >> 6, lane = tid & 63;
      const T *__restrict__ S2 = plan + pl.offS2;
      const T *__restrict__ S1 = plan + pl.offS1;
      for (uint32_t t0 = s_lo; t0 < s_hi; t0 += per_tile) {
        const int cnt = (int)min((uint32_t)per_tile, s_hi - t0);
        __syncthreads();  // scratch free (diag sum / previous tile consumed)
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
          hrow[r + 1] = ((e >> 16) & 1u) ? -acc : acc;
          if constexpr (WRITE_COMB) {
            uint64_t ket[LEN];
            ket_from<LEN>(wk, h, q, 0, 0, false, ket);
#pragma unroll
            for (int i = 0; i < LEN; ++i) crow[(size_t)(r + 1) * LEN + i] = ket[i];
          }
        }
      }
    }
  }

  // ---- column 0: x itself and <x|H|x>.  Placed after the singles so that the single lane doing the
  // ordered summation overlaps with the doubles loops of the other waves.
  if (lo == 0) {
    if constexpr (WRITE_COMB) {
      if (tid < LEN) crow[tid] = pick<LEN>(wk.w, tid);
    }
    __syncthreads();  // scratch free (last singles tile consumed)
    diag_phase_plan<T>(p, L, plan + pl.offD1, plan + pl.offD2, hrow);
  }

  // ---- doubles ------------------------------------------------------------------------------------------
  // Software-pipelined: the table element of the NEXT excitations is requested before the stores of the
  // current ones are issued.  gfx950 retires vector-memory operations in order for s_waitcnt purposes
  // (stores included), so a load issued behind stores cannot be consumed until those stores have been
  // acknowledged by HBM; issued ahead of them it only waits for itself.
#ifndef PYNQS_U
#define PYNQS_U 4
#endif
  constexpr int U = PYNQS_U;  // excitations in flight per lane

  // same-spin: [d1, d2) alpha, [d2, d3) beta
#pragma unroll
  for (int spin = 0; spin < 2; ++spin) {
    const uint32_t b0 = spin ? p.d2 : p.d1, b1 = spin ? p.d3 : p.d2;
    const uint32_t a0 = max(rlo, b0), a1 = min(rhi, b1);
    if (a0 >= a1) continue;
    const uint32_t npair = spin ? p.noBB : p.noAA;
    const uint32_t rot = spin ? p.rotB : p.rotA;
    const MagicDiv dv = spin ? p.divNoBB : p.divNoAA;
    const uint32_t *__restrict__ HP = L.tab + (spin ? p.offHPb : p.offHPa);
    const uint32_t *__restrict__ PP = L.tab + (spin ? p.offPPb : p.offPPa);
    const T *__restrict__ V = plan + pl.offVss + (size_t)spin * NP * NP;
    auto fetch = [&](uint32_t r, uint32_t &eh, uint32_t &ep, T &v) {
      const uint32_t t = r - b0;
      const uint32_t ab = mdiv(t, dv);
      uint32_t ij = t - ab * npair + rot;  // == r % npair (excitation.cpp:63,79)
      ij = ij >= npair ? ij - npair : ij;
      eh = HP[ij]; ep = PP[ab];
      v = V[__umul24((ep >> 17) & 0x1fffu, NP) + ((eh >> 17) & 0x1fffu)];
    };
    uint32_t eh[U], ep[U];
    T v[U];
    uint32_t r = a0 + tid;
#pragma unroll
    for (int u = 0; u < U; ++u) { eh[u] = ep[u] = 0; v[u] = T(0); if (r + u * kBlock < a1) fetch(r + u * kBlock, eh[u], ep[u], v[u]); }
    while (r < a1) {
      uint32_t neh[U], nep[U];
      T nv[U];
      const uint32_t rn = r + U * kBlock;
#pragma unroll
      for (int u = 0; u < U; ++u) { neh[u] = nep[u] = 0; nv[u] = T(0); if (rn + u * kBlock < a1) fetch(rn + u * kBlock, neh[u], nep[u], nv[u]); }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t rr = r + u * kBlock;
        if (rr < a1) {
          const int h0 = eh[u] & 0xff, h1 = (eh[u] >> 8) & 0xff, q0 = ep[u] & 0xff, q1 = (ep[u] >> 8) & 0xff;
          const uint32_t par = (((eh[u] ^ ep[u]) >> 16) & 1u) ^ (uint32_t)(h0 < q0) ^ (uint32_t)(h1 < q0) ^
                               (uint32_t)(h0 < q1) ^ (uint32_t)(h1 < q1);
          store_h<T>(hrow, rr + 1, par ? -v[u] : v[u]);
          if constexpr (WRITE_COMB) {
            uint64_t ket[LEN];
            ket_from<LEN>(wk, h0, h1, q0, q1, true, ket);
            store_ket<LEN>(crow, rr + 1, ket);
          }
        }
      }
      r = rn;
#pragma unroll
      for (int u = 0; u < U; ++u) { eh[u] = neh[u]; ep[u] = nep[u]; v[u] = nv[u]; }
    }
  }

  // opposite-spin: [d3, nsd)
  {
    const uint32_t a0 = max(rlo, p.d3), a1 = min(rhi, p.nsd);
    const uint32_t *__restrict__ SA = L.tab + p.offSa;
    const uint32_t *__restrict__ SB = L.tab + p.offSb;
    const T *__restrict__ V = plan + pl.offVab;
    const uint32_t K2 = K * K;
    auto fetch = [&](uint32_t r, uint32_t &ea, uint32_t &eb, T &v) {
      const uint32_t t = r - p.d3;
      const uint32_t jb = mdiv(t, p.divNSa);
      const uint32_t ia = t - jb * (uint32_t)p.nSa;
      ea = SA[ia]; eb = SB[jb];
#ifdef PYNQS_ABL_NOGATHER
      v = T(1) + T((eb >> 17) + (ea >> 17));
#else
      v = V[__umul24(eb >> 17, K2) + (ea >> 17)];
#endif
    };
    uint32_t ea[U], eb[U];
    T v[U];
    uint32_t r = a0 + tid;
#pragma unroll
    for (int u = 0; u < U; ++u) { ea[u] = eb[u] = 0; v[u] = T(0); if (r + u * kBlock < a1) fetch(r + u * kBlock, ea[u], eb[u], v[u]); }
    while (r < a1) {
      uint32_t nea[U], neb[U];
      T nv[U];
      const uint32_t rn = r + U * kBlock;
#pragma unroll
      for (int u = 0; u < U; ++u) { nea[u] = neb[u] = 0; nv[u] = T(0); if (rn + u * kBlock < a1) fetch(rn + u * kBlock, nea[u], neb[u], nv[u]); }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t rr = r + u * kBlock;
        if (rr < a1) {
          const int ha = ea[u] & 0xff, qa = (ea[u] >> 8) & 0xff, hb = eb[u] & 0xff, qb = (eb[u] >> 8) & 0xff;
          const uint32_t par = (((ea[u] ^ eb[u]) >> 16) & 1u) ^ (uint32_t)(ha < qb) ^ (uint32_t)(hb < qa) ^ 1u;
#ifdef PYNQS_ABL_NOSTORE
          const T vv = par ? -v[u] : v[u];
          uint64_t ket[LEN];
          ket_from<LEN>(wk, ha, qa, hb, qb, true, ket);
          asm volatile("" ::"v"(vv), "v"(ket[0]));
#else
          store_h<T>(hrow, rr + 1, par ? -v[u] : v[u]);
          if constexpr (WRITE_COMB) {
            uint64_t ket[LEN];
            ket_from<LEN>(wk, ha, qa, hb, qb, true, ket);
            store_ket<LEN>(crow, rr + 1, ket);
          }
#endif
        }
      }
      r = rn;
#pragma unroll
      for (int u = 0; u < U; ++u) { ea[u] = nea[u]; eb[u] = neb[u]; v[u] = nv[u]; }
    }
  }
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

extern "C" int64_t pynqs_plan_bytes(int sorb, int dtype) {
  PlanLayout pl;
  if (!make_plan_layout(sorb, &pl) || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return -1;
  return pl.total * (dtype == PYNQS_F64 ? 8 : 4);
}

extern "C" int pynqs_plan_build(const void *h1e, const void *h2e, int sorb, int dtype, void *plan, void *stream) {
  PlanLayout pl;
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (dtype != PYNQS_F32 && dtype != PYNQS_F64) return set_error(PYNQS_EINVAL, "bad dtype");
  if (!h1e || !h2e || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  const uint64_t grid = ((uint64_t)pl.total + kBlock - 1) / kBlock;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYNQS_F64)
    hipLaunchKernelGGL((plan_build_kernel<double>), dim3((uint32_t)grid), dim3(kBlock), 0, st, (const double *)h1e,
                       (const double *)h2e, pl, (double *)plan);
  else
    hipLaunchKernelGGL((plan_build_kernel<float>), dim3((uint32_t)grid), dim3(kBlock), 0, st, (const float *)h1e,
                       (const float *)h2e, pl, (float *)plan);
  return check_launch("plan_build");
}

#define DISPATCH_LEN(len, ...)                                  \
  switch (len) {                                                \
    case 1: { constexpr int LEN = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int LEN = 2; __VA_ARGS__; } break;      \
    default: { constexpr int LEN = 3; __VA_ARGS__; } break;     \
  }

template <int LEN, typename T>
static int launch_plan(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, const T *plan,
                       uint64_t *comb, T *hmat, hipStream_t st) {
  const uint32_t ncomb = p.nsd + 1;
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, ncomb, &nchunks, &chunk_len);
  const size_t lds = lds_bytes(p, sizeof(T));
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large: nbatch*nchunks > 2^31-1");
  if (comb)
    hipLaunchKernelGGL((comb_hij_plan_kernel<LEN, T, true>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                       chunk_len, plan, comb, hmat);
  else
    hipLaunchKernelGGL((comb_hij_plan_kernel<LEN, T, false>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                       chunk_len, plan, comb, hmat);
  return check_launch("comb_hij_plan");
}

extern "C" int pynqs_comb_hij_fused_plan(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                         const void *plan, int dtype, uint64_t *comb, void *hmat, void *stream) {
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return set_error(PYNQS_EINVAL, "bad nbatch/dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !hmat) return set_error(PYNQS_EINVAL, "null pointer");
  if (nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "nbatch too large");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  DISPATCH_LEN(len, rc = dtype == PYNQS_F64 ? launch_plan<LEN, double>(bra, nbatch, p, pl, (const double *)plan, comb, (double *)hmat, st)
                                            : launch_plan<LEN, float>(bra, nbatch, p, pl, (const float *)plan, comb, (float *)hmat, st));
  return rc;
}
