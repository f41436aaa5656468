// plan_dev.h -- device routines shared by the plan-based kernels (kernels_plan.hip, kernels_eloc.hip).
// Everything here reproduces the reference's arithmetic exactly (cpp_src/cpu/excitation.cpp:125-169,
// hamiltonian.cpp:34-50); only the data movement is ours.
#pragma once

#include "detcore.h"
#include "plan.h"

namespace pynqs {

// ---- singles and diagonal (used by the tile scheduler, plan_tiles.h) -------------------------------------------
// A single p -> q needs h(p,q) + sum_{k in occ(x)} <pk||qk> added in the reference's order (k: word ascending, bit
// 63 -> 0, excitation.cpp:141-157), the diagonal the nele(nele+1)/2 terms of hamiltonian.cpp:41-48 in theirs: the
// terms are gathered with all lanes, staged in LDS and added by one lane per sum (bit-identical results).
// Both are carried out by ONE wave with no workgroup barrier, so that the other waves of the
// workgroup can run the doubles meanwhile (ablation, profiles/: with workgroup-wide phases the singles and
// the diagonal, 2 % of the columns, cost 0.08 ms of a 0.26 ms kernel in barrier-separated, mostly idle steps).
// LDS operations of one wave execute in order; wave_sync() keeps the compiler from moving LDS accesses
// across the hand-over points and drains the LDS queue.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// QUARTER: elements of the wave's private part of L.scratch (default: a quarter of the kDiagTile elements a 256-thread workgroup
// reserves; kernels that are short of LDS pass less and take more rounds)
template <typename T, int QUARTER = kDiagTile / 4, typename Sink>
__device__ __forceinline__ void diag_wave(const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                          const T *__restrict__ plan, Sink sink) {
  const T *__restrict__ D1 = plan + pl.offD1;
  const T *__restrict__ D2 = plan + pl.offD2;
  // the wave's private quarter of L.scratch (other waves may be staging singles in theirs)
  constexpr int kQuarter = QUARTER;
  T *tile = reinterpret_cast<T *>(L.scratch) + (threadIdx.x >> 6) * kQuarter;
  const int lane = threadIdx.x & 63;
  const int nele = p.nele;
  const int nterms = nele * (nele + 1) / 2;
  T acc = T(0);
  for (int base = 0; base < nterms; base += kQuarter) {
    const int end = min(base + kQuarter, nterms);
    // gather, 4 requests in flight per lane
    for (int t0 = base + lane; t0 < end; t0 += 4 * 64) {
      T val[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * 64;
        val[u] = T(0);
        if (t < end) {
          int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
          while (a * (a + 1) / 2 > t) --a;
          while ((a + 1) * (a + 2) / 2 <= t) ++a;
          const int pos = t - a * (a + 1) / 2;
          const int pa = L.occa[a];
          val[u] = pos == 0 ? D1[pa] : D2[pa * p.sorb + L.occa[pos - 1]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t0 + u * 64 < end) tile[t0 + u * 64 - base] = val[u];
    }
    wave_sync();
    if (lane == 63) {
      // ordered sum; the next 16 LDS values are requested while the current 16 are being added, so the
      // ~100-cycle LDS latency is paid once, not per group (it was 80 % of this loop's time)
      const int n = end - base;
      constexpr int Gp = 8;
      T cur[Gp], nxt[Gp];
      int t = 0;
      if (n >= Gp) {
#pragma unroll
        for (int u = 0; u < Gp; ++u) cur[u] = tile[u];
        for (; t + 2 * Gp <= n; t += Gp) {
#pragma unroll
          for (int u = 0; u < Gp; ++u) nxt[u] = tile[t + Gp + u];
#pragma unroll
          for (int u = 0; u < Gp; ++u) acc += cur[u];
#pragma unroll
          for (int u = 0; u < Gp; ++u) cur[u] = nxt[u];
        }
#pragma unroll
        for (int u = 0; u < Gp; ++u) acc += cur[u];
        t += Gp;
      }
      for (; t < n; ++t) acc += tile[t];
    }
    wave_sync();
  }
  if (lane == 63) sink(acc);
}

// A tile of up to kSinglesPerTile singles handled by ONE wave, no workgroup barrier: coalesced gather (G lanes
// walk one S2 row -> ~3 lines per single instead of 64 different lines per instruction, which made the vector
// L1 the bottleneck again when every lane walked its own row), staged in the wave's private quarter of
// L.scratch, then one lane per single adds its terms in the reference's order.
// sink(rank, value, table entry) is called by the summing lanes.
constexpr int kSinglesPerTile = 16;

template <typename T, int QUARTER = kDiagTile / 4, typename Sink>
__device__ __forceinline__ void singles_tile(uint32_t r0, uint32_t r_end, const SDParams &p, const PlanLayout &pl,
                                             const LdsLayout &L, int nocc, const T *__restrict__ plan, Sink sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t K = (uint32_t)pl.K;
  T *tile = reinterpret_cast<T *>(L.scratch) + wave * QUARTER;
  const int stride = nocc | 1;
  const int cap = max(1, min(kSinglesPerTile, QUARTER / stride));  // singles per pass
  const T *__restrict__ S2 = plan + pl.offS2;
  const int G = nocc <= 16 ? 16 : (nocc <= 32 ? 32 : 64);
  const int gshift = nocc <= 16 ? 4 : (nocc <= 32 ? 5 : 6);
  const int per_iter = 64 >> gshift;
  const int my_s = lane >> gshift, my_j = lane & (G - 1);
  for (uint32_t t0 = r0; t0 < r_end; t0 += cap) {
    const int cnt = (int)min((uint32_t)cap, r_end - t0);
    for (int sl0 = my_s; sl0 < cnt; sl0 += 4 * per_iter) {  // 4 requests in flight per lane
      T val[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sl = sl0 + u * per_iter;
        ok[u] = sl < cnt && my_j < nocc;
        val[u] = T(0);
        if (ok[u]) {
          const uint32_t r = t0 + sl;
          const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
          const uint32_t spin = r >= p.d0;
          const uint32_t hm = (e & 0xff) >> 1, qm = ((e >> 8) & 0xff) >> 1;
          val[u] = S2[((size_t)(spin * K + hm) * K + qm) * p.sorb + L.occv[my_j]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) tile[(sl0 + u * per_iter) * stride + my_j] = val[u];
    }
    if (nocc > 64) {  // more electrons than lanes: the remaining terms of each single
      for (int sl = 0; sl < cnt; ++sl) {
        const uint32_t r = t0 + sl;
        const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
        const uint32_t spin = r >= p.d0;
        const uint32_t hm = (e & 0xff) >> 1, qm = ((e >> 8) & 0xff) >> 1;
        const T *__restrict__ rowp = S2 + ((size_t)(spin * K + hm) * K + qm) * p.sorb;
        for (int j = 64 + lane; j < nocc; j += 64) tile[sl * stride + j] = rowp[L.occv[j]];
      }
    }
    wave_sync();
    if (lane < cnt) {
      const uint32_t r = t0 + lane;
      const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
      const uint32_t spin = r >= p.d0;
      T acc = T(0);
      acc += plan[pl.offS1 + (size_t)(spin * K + ((e & 0xff) >> 1)) * K + (((e >> 8) & 0xff) >> 1)];
      const T *__restrict__ mine = tile + lane * stride;
      int j = 0;
      for (; j + 8 <= nocc; j += 8) {  // reference order; 8 LDS reads issued together
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = mine[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; j < nocc; ++j) acc += mine[j];
      sink(r, ((e >> 16) & 1u) ? -acc : acc, e);
    }
    wave_sync();
  }
}

// ---- order-free variants -----------------------------------------------------------------------------------
// For kernels whose result is a rounded sum over all columns anyway (fused local energies, tolerance 1e-8 Ha): the
// terms of <x|H|x> and of a single are added in whatever order is cheapest, no LDS staging:
//   fast_diag  : lane a adds h(p_a,p_a) + sum_{b<a} <p_a p_b||p_a p_b>, then a butterfly over the wave (all lanes get it);
//   fast_single: one lane walks the single's S2 row over the occupied orbitals (its 2-3 cache lines stay in L1).
template <typename T>
__device__ __forceinline__ T fast_diag(const SDParams &p, const PlanLayout &pl, const LdsLayout &L, const T *__restrict__ plan) {
  const int lane = threadIdx.x & 63;
  const T *__restrict__ D1 = plan + pl.offD1;
  const T *__restrict__ D2 = plan + pl.offD2;
  T acc = T(0);
  for (int a = lane; a < p.nele; a += 64) {
    const uint32_t pa = L.occa[a];
    acc += D1[pa];
    const T *__restrict__ row = D2 + pa * (uint32_t)p.sorb;
#pragma unroll 4
    for (int b = 0; b < a; ++b) acc += row[L.occa[b]];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  return acc;
}

// single excitation rank r < d1 (signed matrix element)
template <typename T>
__device__ __forceinline__ T fast_single(uint32_t r, const SDParams &p, const PlanLayout &pl, const LdsLayout &L, int nocc,
                                         const T *__restrict__ plan) {
  const uint32_t e = L.tab[p.offSa + r], K = (uint32_t)pl.K;
  const uint32_t pq = ((r >= p.d0 ? K : 0u) + ((e & 0xff) >> 1)) * K + (((e >> 8) & 0xff) >> 1);
  const T *__restrict__ row = plan + pl.offS2 + (size_t)pq * p.sorb;
  T acc = plan[pl.offS1 + pq];
#pragma unroll 8
  for (int j = 0; j < nocc; ++j) acc += row[L.occv[j]];
  return ((e >> 16) & 1u) ? -acc : acc;
}

// ---- doubles ----------------------------------------------------------------------------------------------
// A double excitation whose table element has been requested: the two LDS table entries and the value.
// For single-word ONVs (LEN == 1) the ket comes from the LDS mask tables (detcore.h: msk): k0 ^ k1.
template <typename T>
struct PendingDouble {
  uint32_t e0, e1;
  uint64_t k0, k1;
  T v;
};

// A class of doubles = a 2-D index space (slow, fast) with rank = b0 + slow * nfast + fast, two LDS tables and
// one dense plan table.  Same-spin: fast = hole pair (rotated: the reference's `idx % noAA` quirk,
// excitation.cpp:63,79), slow = particle pair.  Opposite spin: fast = alpha single, slow = beta single.
struct DoubleClass {
  uint32_t b0;      // first rank of the class
  uint32_t nfast;   // size of the fast index
  uint32_t rot;     // rotation of the fast index (0 for opposite spin)
  MagicDiv dv;      // division by nfast
  uint32_t off_fast, off_slow;  // table offsets (entries) inside L.tab / L.msk
  uint32_t mul;     // plan offset = slow_part * mul + fast_part
  uint32_t mask;    // mask of the plan offset parts inside a table entry (after >> 17)
  bool opposite;
};

inline __device__ DoubleClass make_same_spin(const SDParams &p, const PlanLayout &pl, int spin) {
  DoubleClass c;
  c.b0 = spin ? p.d2 : p.d1;
  c.nfast = spin ? p.noBB : p.noAA;
  c.rot = spin ? p.rotB : p.rotA;
  c.dv = spin ? p.divNoBB : p.divNoAA;
  c.off_fast = spin ? p.offHPb : p.offHPa;
  c.off_slow = spin ? p.offPPb : p.offPPa;
  c.mul = (uint32_t)pl.NP;
  c.mask = 0x1fffu;
  c.opposite = false;
  return c;
}

inline __device__ DoubleClass make_opp_spin(const SDParams &p, const PlanLayout &pl) {
  DoubleClass c;
  c.b0 = p.d3;
  c.nfast = (uint32_t)p.nSa;
  c.rot = 0;
  c.dv = p.divNSa;
  c.off_fast = p.offSa;
  c.off_slow = p.offSb;
  c.mul = (uint32_t)(pl.K * pl.K);
  c.mask = 0x7fffu;
  c.opposite = true;
  return c;
}

// (slow, fast-table index) of rank r
__device__ __forceinline__ void class_split(uint32_t r, const DoubleClass &c, uint32_t &slow, uint32_t &u) {
  const uint32_t t = r - c.b0;
  slow = mdiv(t, c.dv);
  u = t - slow * c.nfast;
}

template <int LEN, typename T>
__device__ __forceinline__ PendingDouble<T> fetch_at(uint32_t slow, uint32_t u, const DoubleClass &c, const LdsLayout &L,
                                                     const T *__restrict__ V) {
  PendingDouble<T> d;
  uint32_t f = u + c.rot;
  f = f >= c.nfast ? f - c.nfast : f;
  d.e0 = L.tab[c.off_fast + f];
  d.e1 = L.tab[c.off_slow + slow];
  if constexpr (LEN == 1) {
    d.k0 = L.msk[c.off_fast + f];
    d.k1 = L.msk[c.off_slow + slow];
  }
  d.v = V[__umul24((d.e1 >> 17) & c.mask, c.mul) + ((d.e0 >> 17) & c.mask)];
  return d;
}

template <int LEN, typename T>
__device__ __forceinline__ PendingDouble<T> fetch_double(uint32_t r, const DoubleClass &c, const LdsLayout &L,
                                                         const T *__restrict__ V) {
  uint32_t slow, u;
  class_split(r, c, slow, u);
  return fetch_at<LEN, T>(slow, u, c, L, V);
}

// ranks r and r + 1 with one division
template <int LEN, typename T>
__device__ __forceinline__ void fetch_double2(uint32_t r, const DoubleClass &c, const LdsLayout &L, const T *__restrict__ V,
                                              PendingDouble<T> &d0, PendingDouble<T> &d1) {
  uint32_t slow, u;
  class_split(r, c, slow, u);
  d0 = fetch_at<LEN, T>(slow, u, c, L, V);
  const bool wrap = u + 1 == c.nfast;
  d1 = fetch_at<LEN, T>(wrap ? slow + 1 : slow, wrap ? 0u : u + 1, c, L, V);
}

// sign, value and ket of a fetched double.  Table entries: orbital | orbital << 8 | parity << 16.
//   same spin    : e0 = hole pair (h0 > h1), e1 = particle pair (q0 > q1); parity bits carry P(h0)^P(h1) and
//                  P(q0)^P(q1)^1; cross term [h0<q0]^[h1<q0]^[h0<q1]^[h1<q1]
//   opposite spin: e0 = alpha (hole, particle), e1 = beta (hole, particle); parity bits carry
//                  P(h)^P(q)^[h<q] per spin; cross term [ha<qb]^[hb<qa]^1
template <int LEN, typename T>
__device__ __forceinline__ T finish_double(const PendingDouble<T> &d, const DoubleClass &c, const Walker<LEN> &wk,
                                           uint64_t (&ket)[LEN]) {
  const int a0 = d.e0 & 0xff, a1 = (d.e0 >> 8) & 0xff, b0 = d.e1 & 0xff, b1 = (d.e1 >> 8) & 0xff;
  uint32_t par = ((d.e0 ^ d.e1) >> 16) & 1u;
  if (c.opposite) par ^= (uint32_t)(a0 < b1) ^ (uint32_t)(b0 < a1) ^ 1u;
  else par ^= (uint32_t)(a0 < b0) ^ (uint32_t)(a1 < b0) ^ (uint32_t)(a0 < b1) ^ (uint32_t)(a1 < b1);
  if constexpr (LEN == 1) {
    ket[0] = d.k0 ^ d.k1;
  } else {
#pragma unroll
    for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
    toggle<LEN>(ket, a0); toggle<LEN>(ket, a1); toggle<LEN>(ket, b0); toggle<LEN>(ket, b1);
  }
  return par ? -d.v : d.v;
}

// Any double rank r in [d1, nsd): element + ket (used by the fused E_loc kernels).
template <int LEN, typename T>
__device__ __forceinline__ T double_element(uint32_t r, const SDParams &p, const PlanLayout &pl, const LdsLayout &L,
                                            const T *__restrict__ plan, const Walker<LEN> &wk, uint64_t (&ket)[LEN]) {
  if (r < p.d3) {
    const int spin = r >= p.d2;
    const DoubleClass c = make_same_spin(p, pl, spin);
    const PendingDouble<T> d = fetch_double<LEN, T>(r, c, L, plan + pl.offVss + (size_t)spin * pl.NP * pl.NP);
    return finish_double<LEN, T>(d, c, wk, ket);
  }
  const DoubleClass c = make_opp_spin(p, pl);
  const PendingDouble<T> d = fetch_double<LEN, T>(r, c, L, plan + pl.offVab);
  return finish_double<LEN, T>(d, c, wk, ket);
}

}  // namespace pynqs
