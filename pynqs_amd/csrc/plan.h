// plan.h -- "integral plan": a spin-blocked, dense re-layout of (h1e, h2e) in device memory.
//
// Why.  The reference API hands over h2e as one packed triangle over ALL spin-orbital pairs
// (cpp_src/tensor/integral.cpp:6-60).  Gathering from it, the 64 lanes of a wave touch ~64 different
// 128-byte lines per load instruction and the CU's vector L1 (TCP) becomes the bottleneck (rocprof,
// profiles/r01_fe2s2_dropin_v1: TCP_PENDING_STALL 73 % of the kernel).  In the plan the index that
// varies fastest across lanes in the reference's enumeration order (the hole orbital / hole pair) is the
// fastest index in memory, over ALL orbitals of that spin, so a wave's gathers fall into a few lines.
// Every table entry is a (possibly negated) copy of one h2e/h1e element: values stay bit-identical.
//
// Tables (T = integral dtype, K = sorb/2 spatial orbitals, NP = K(K-1)/2):
//   Vab [pb][pj][pa][pi]        K^4       <p0 p1||q0 q1>, holes (2pi, 2pj+1), particles (2pa, 2pb+1)
//   Vss [spin][ab_pair][ij_pair] 2 NP^2   same-spin doubles, pair rank m1(m1-1)/2+m0 over spatial orbitals
//   S2  [spin][pm][qm][k]       2 K^2 sorb <p k||q k>, p = 2pm+spin, q = 2qm+spin, k any spin orbital
//   S1  [spin][pm][qm]          2 K^2     h1e_get(p, q) = h1e[q*sorb + p]
//   D2  [p][q]                  sorb^2    <p q||p q>
//   D1  [p]                     sorb      h1e[p*sorb + p]
// The plan needs an even sorb (alpha = even, beta = odd spin orbitals, excitation.cpp:47-56).
#pragma once

#include <stdint.h>

namespace pynqs {

struct PlanLayout {
  int sorb, K, NP;
  int64_t offVab, offVss, offS2, offS1, offD2, offD1, total;  // in elements
};

inline bool make_plan_layout(int sorb, PlanLayout *L) {
  if (sorb < 2 || sorb > 192 || (sorb & 1)) return false;
  const int64_t K = sorb / 2, NP = K * (K - 1) / 2;
  L->sorb = sorb; L->K = (int)K; L->NP = (int)NP;
  int64_t o = 0;
  L->offVab = o; o += K * K * K * K;
  L->offVss = o; o += 2 * NP * NP;
  L->offS2 = o; o += 2 * K * K * sorb;
  L->offS1 = o; o += 2 * K * K;
  L->offD2 = o; o += (int64_t)sorb * sorb;
  L->offD1 = o; o += sorb;
  L->total = (o + 1) & ~(int64_t)1;
  return true;
}

}  // namespace pynqs
