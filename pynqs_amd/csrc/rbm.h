// rbm.h -- device-memory layout of the "RBM table": the parameters of a real restricted-Boltzmann-machine
// amplitude (PyNQS vmc/ansatz/rbm/rbm.py:186-211, rbm_type "real")
//     psi(x) = exp(a.x) * prod_h 2 cosh(theta_h(x)),   theta_h = b_h + sum_o W[h][o] x_o,   x_o = +-1,
// re-laid out for the fused local-energy kernel (kernels_rbm.hip), in caller-owned memory like the integral plan.
// All tables are double, hidden index fastest (a wave reads a row of one orbital with consecutive lanes):
//   Wt  [sorb][Hq]   W transposed (padding 0)
//   E4p [sorb][Hq]   exp(+4 W[h][o])   (padding 1)
//   E4m [sorb][Hq]   exp(-4 W[h][o])   (padding 1)
//   hb  [Hq]         hidden bias b (padding 0)
//   vb  [sorb]       visible bias a (0 when the caller passes none)
// Hq = row stride = H rounded up to a multiple of 8, plus 1: odd, so that in LDS (read with ds_read_b64: banks
// = 8-byte pairs of a 256-byte row, conflicts among 32 lanes) 32 consecutive rows start in 32 different pairs.
#pragma once

#include <stdint.h>

namespace pynqs {

struct RbmLayout {
  int sorb, H, Hq, Hloop;  // Hloop = H rounded up to a multiple of 8 (what the hidden-unit loop runs over)
  int64_t offWt, offE4p, offE4m, offHb, offVb, total;  // in doubles
};

inline bool make_rbm_layout(int sorb, int H, RbmLayout *L) {
  if (sorb < 1 || sorb > 192 || H < 1 || H > 8192) return false;
  L->sorb = sorb; L->H = H;
  L->Hloop = (H + 7) & ~7;
  L->Hq = L->Hloop + 1;
  const int64_t row = (int64_t)sorb * L->Hq;
  L->offWt = 0; L->offE4p = row; L->offE4m = 2 * row; L->offHb = 3 * row;
  L->offVb = L->offHb + L->Hq;
  L->total = (L->offVb + sorb + 1) & ~(int64_t)1;
  return true;
}

}  // namespace pynqs
