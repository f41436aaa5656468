// plan_dev.h -- device routines shared by the plan-based kernels (kernels_plan.hip, kernels_eloc.hip).
// Everything here reproduces the reference's arithmetic exactly (cpp_src/cpu/excitation.cpp:125-169,
// hamiltonian.cpp:34-50); only the data movement is ours.
#pragma once

#include "detcore.h"
#include "plan.h"

namespace pynqs {

// ---- singles -------------------------------------------------------------------------------------------
// <x|H|x'> of the single excitations with ranks [s_lo, s_hi) (all < d1).  A single p -> q needs
// h(p,q) + sum_{k in occ(x)} <pk||qk> added in the reference's order (k: word ascending, bit 63 -> 0,
// excitation.cpp:141-157).  The nocc terms of one single are one row of the plan's S2 table: lanes gather
// (single, k) pairs with full lane utilisation, stage them in LDS, then one lane per single adds them in order.
// sink(rank, value, hole, particle) is called once per single by the summing lane.
// All threads of the workgroup must call this (it contains barriers).
template <typename T, typename Sink>
__device__ __forceinline__ void singles_phase(const SDParams &p, const PlanLayout &pl, const LdsLayout &L, int nocc,
                                              const T *__restrict__ plan, uint32_t s_lo, uint32_t s_hi, Sink sink) {
  if (s_lo >= s_hi) return;
  const int tid = threadIdx.x;
  const uint32_t K = (uint32_t)pl.K;
  T *tile = reinterpret_cast<T *>(L.scratch);
  const int stride = nocc | 1;  // odd: conflict-free column reads in the summation
  const int per_tile = max(1, min(kBlock, kDiagTile / stride));
  const T *__restrict__ S2 = plan + pl.offS2;
  const T *__restrict__ S1 = plan + pl.offS1;
  // lanes are grouped G per single (G = power of two >= nocc, at most 64)
  const int G = nocc <= 16 ? 16 : (nocc <= 32 ? 32 : 64);
  const int gshift = nocc <= 16 ? 4 : (nocc <= 32 ? 5 : 6);
  const int per_iter = kBlock >> gshift;
  const int my_s = tid >> gshift, my_j = tid & (G - 1);
  for (uint32_t t0 = s_lo; t0 < s_hi; t0 += per_tile) {
    const int cnt = (int)min((uint32_t)per_tile, s_hi - t0);
    __syncthreads();  // scratch free
    for (int sl = my_s; sl < cnt; sl += per_iter) {
      const uint32_t r = t0 + sl;
      const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
      const uint32_t spin = r >= p.d0;
      const uint32_t hm = (e & 0xff) >> 1, qm = ((e >> 8) & 0xff) >> 1;
      const T *__restrict__ rowp = S2 + ((size_t)(spin * K + hm) * K + qm) * p.sorb;
      for (int j = my_j; j < nocc; j += G) tile[sl * stride + j] = rowp[L.occv[j]];
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
      sink(r, ((e >> 16) & 1u) ? -acc : acc, h, q);
    }
  }
}

// ---- diagonal ---------------------------------------------------------------------------------------------
// <x|H|x> from the plan's D1/D2 with hamiltonian.cpp:41-48's order of additions: the workgroup gathers the
// nele(nele+1)/2 terms into LDS, the LAST lane of the workgroup adds them in order and calls sink(value).
// Begins with a barrier (scratch must be free); the other lanes return right after the last gather.
template <typename T, typename Sink>
__device__ __forceinline__ void diag_phase_plan(const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                                const T *__restrict__ plan, Sink sink) {
  const T *__restrict__ D1 = plan + pl.offD1;
  const T *__restrict__ D2 = plan + pl.offD2;
  T *tile = reinterpret_cast<T *>(L.scratch);
  const int tid = threadIdx.x;
  const int nele = p.nele;
  const int nterms = nele * (nele + 1) / 2;
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
    if (tid == kBlock - 1) {
      int t = 0;
      const int n = end - base;
      for (; t + 4 <= n; t += 4) {  // same order, fewer loop instructions
        acc += tile[t]; acc += tile[t + 1]; acc += tile[t + 2]; acc += tile[t + 3];
      }
      for (; t < n; ++t) acc += tile[t];
    }
  }
  if (tid == kBlock - 1) sink(acc);
}

// ---- doubles ----------------------------------------------------------------------------------------------
// A double excitation whose table element has been requested: the two LDS table entries and the value.
template <typename T>
struct PendingDouble {
  uint32_t e0, e1;
  T v;
};

struct SameSpinClass {
  uint32_t b0;       // first rank of the class
  uint32_t npair;    // number of hole pairs
  uint32_t rot;      // b0 % npair : the reference's `idx % noAA` quirk as a rotation (excitation.cpp:63,79)
  MagicDiv dv;
  const uint32_t *HP, *PP;
  uint32_t NP;
};

template <typename T>
__device__ __forceinline__ PendingDouble<T> fetch_same_spin(uint32_t r, const SameSpinClass &c, const T *__restrict__ V) {
  PendingDouble<T> d;
  const uint32_t t = r - c.b0;
  const uint32_t ab = mdiv(t, c.dv);
  uint32_t ij = t - ab * c.npair + c.rot;
  ij = ij >= c.npair ? ij - c.npair : ij;
  d.e0 = c.HP[ij];
  d.e1 = c.PP[ab];
  d.v = V[__umul24((d.e1 >> 17) & 0x1fffu, c.NP) + ((d.e0 >> 17) & 0x1fffu)];
  return d;
}

template <int LEN, typename T>
__device__ __forceinline__ T finish_same_spin(const PendingDouble<T> &d, const Walker<LEN> &wk, uint64_t (&ket)[LEN]) {
  const int h0 = d.e0 & 0xff, h1 = (d.e0 >> 8) & 0xff, q0 = d.e1 & 0xff, q1 = (d.e1 >> 8) & 0xff;
  const uint32_t par = (((d.e0 ^ d.e1) >> 16) & 1u) ^ (uint32_t)(h0 < q0) ^ (uint32_t)(h1 < q0) ^ (uint32_t)(h0 < q1) ^
                       (uint32_t)(h1 < q1);
#pragma unroll
  for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
  toggle<LEN>(ket, h0); toggle<LEN>(ket, h1); toggle<LEN>(ket, q0); toggle<LEN>(ket, q1);
  return par ? -d.v : d.v;
}

struct OppSpinClass {
  uint32_t b0;  // d3
  uint32_t nSa;
  MagicDiv dv;
  const uint32_t *SA, *SB;
  uint32_t K2;
};

template <typename T>
__device__ __forceinline__ PendingDouble<T> fetch_opp_spin(uint32_t r, const OppSpinClass &c, const T *__restrict__ V) {
  PendingDouble<T> d;
  const uint32_t t = r - c.b0;
  const uint32_t jb = mdiv(t, c.dv);
  const uint32_t ia = t - jb * c.nSa;
  d.e0 = c.SA[ia];
  d.e1 = c.SB[jb];
  d.v = V[__umul24(d.e1 >> 17, c.K2) + (d.e0 >> 17)];
  return d;
}

template <int LEN, typename T>
__device__ __forceinline__ T finish_opp_spin(const PendingDouble<T> &d, const Walker<LEN> &wk, uint64_t (&ket)[LEN]) {
  const int ha = d.e0 & 0xff, qa = (d.e0 >> 8) & 0xff, hb = d.e1 & 0xff, qb = (d.e1 >> 8) & 0xff;
  const uint32_t par = (((d.e0 ^ d.e1) >> 16) & 1u) ^ (uint32_t)(ha < qb) ^ (uint32_t)(hb < qa) ^ 1u;
#pragma unroll
  for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
  toggle<LEN>(ket, ha); toggle<LEN>(ket, qa); toggle<LEN>(ket, hb); toggle<LEN>(ket, qb);
  return par ? -d.v : d.v;
}

inline __device__ SameSpinClass make_same_spin(const SDParams &p, const PlanLayout &pl, const LdsLayout &L, int spin) {
  SameSpinClass c;
  c.b0 = spin ? p.d2 : p.d1;
  c.npair = spin ? p.noBB : p.noAA;
  c.rot = spin ? p.rotB : p.rotA;
  c.dv = spin ? p.divNoBB : p.divNoAA;
  c.HP = L.tab + (spin ? p.offHPb : p.offHPa);
  c.PP = L.tab + (spin ? p.offPPb : p.offPPa);
  c.NP = (uint32_t)pl.NP;
  return c;
}

inline __device__ OppSpinClass make_opp_spin(const SDParams &p, const PlanLayout &pl, const LdsLayout &L) {
  OppSpinClass c;
  c.b0 = p.d3;
  c.nSa = (uint32_t)p.nSa;
  c.dv = p.divNSa;
  c.SA = L.tab + p.offSa;
  c.SB = L.tab + p.offSb;
  c.K2 = (uint32_t)(pl.K * pl.K);
  return c;
}

// Any double rank r in [d1, nsd): element + ket (used by the fused E_loc kernels).
template <int LEN, typename T>
__device__ __forceinline__ T double_element(uint32_t r, const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                            const T *__restrict__ plan, const Walker<LEN> &wk, uint64_t (&ket)[LEN]) {
  if (r < p.d3) {
    const int spin = r >= p.d2;
    const SameSpinClass c = make_same_spin(p, pl, L, spin);
    const PendingDouble<T> d = fetch_same_spin<T>(r, c, plan + pl.offVss + (size_t)spin * pl.NP * pl.NP);
    return finish_same_spin<LEN, T>(d, wk, ket);
  }
  const OppSpinClass c = make_opp_spin(p, pl, L);
  const PendingDouble<T> d = fetch_opp_spin<T>(r, c, plan + pl.offVab);
  return finish_opp_spin<LEN, T>(d, wk, ket);
}

}  // namespace pynqs
