// kernels_rbm_forward.hip -- psi(x) of the reference's RBM amplitudes (vmc/ansatz/rbm/rbm.py:186-211) on a LIST of determinants, straight
// from the packed bits: the amplitude forward of the REDUCE local energy (vmc/energy/eloc.py:299-303, flip.py:44-50: `Func` on the distinct
// x') when the ansatz is an RBM.
//
// The PyTorch module does theta = X W^T as a GEMM on the +-1 matrix X [n, sorb] and then ~15 element-wise kernels on [n, H] arrays: for the
// 1.5 M distinct x' of 8192 Fe2S2 walkers that is a 1 GB theta array read and written a dozen times (3.5 ms per step, 70 % of the REDUCE
// step).  Here one lane owns one determinant and never leaves registers: x_o = +-1, so theta_h = b_h + sum_o (+-W_ho) is sorb fused
// multiply-adds per hidden unit with W_ho wave-uniform (scalar loads, no LDS, no bank conflicts), eight hidden units at a time; then
// ln 2cosh(theta_h) is summed and psi = exp(a.x + sum) written: 8 or 16 bytes of output per determinant, 8 len bytes of input.
//   real parameters :  psi = exp(a.x) prod_h 2cosh(theta_h)         (flavour REAL)
//                      psi = tanh(a.x) prod_h 2cosh(theta_h)        (TANH)
//                      psi = exp(i (a.x + sum_h ln 2cosh(theta_h))) (PHASE, complex output)
//   complex parameters (re, im pairs):  psi = exp(a.x) prod_h 2cosh(theta_h), complex output (PYNQS_RBM_COMPLEX)
// The product prod_h 2cosh(theta_h) is formed as a product, not as exp(sum ln ...): 2cosh t = e^|t| (1 + e^{-2|t|}) keeps the large
// factor as an exponent that is summed, and the bounded factors (1 + e^{-2|t|}) in (1, 2] -- complex: (1 + rho cos phi) + i rho sin phi,
// modulus in [0, 2] -- are multiplied up and renormalised by a power of two every eight hidden units, so nothing overflows or
// underflows for any theta.  One exp (+ one sincos) per hidden unit instead of exp + log (+ sincos + atan2): 0.98 -> 0.5 ms per
// 1.6 M determinants with 40 complex hidden units.
#include "detcore.h"
#include "launch.h"
#include "rbm_math.h"

namespace pynqs {

constexpr int kHChunk = 8;

// running product of the bounded factors with its binary exponent split off (renormalised by the caller every few factors)
struct Prod {
  double re = 1.0, im = 0.0;  // mantissa
  double lin = 0.0, ang = 0.0;  // sum of the |Re theta| (natural-log scale of the modulus) and of the s * Im theta (phase)
  int e2 = 0;
  __device__ __forceinline__ void renorm() {
    int k;
    (void)frexp(fmax(fabs(re), fabs(im)), &k);
    re = ldexp(re, -k); im = ldexp(im, -k);
    e2 += k;
  }
  // *= 2cosh(x + iy) = e^{s(x + iy)} (1 + e^{-2s(x + iy)}),  s = sign(x)
  template <bool CPLX>
  __device__ __forceinline__ void times_2cosh(double x, double y) {
    const double ax = fabs(x);
    const double rho = exp(-2.0 * ax);
    lin += ax;
    if constexpr (CPLX) {
      const double sy = x < 0.0 ? -y : y;
      double sn, cs;
      sincos_mod(-2.0 * sy, sn, cs);
      const double u = fma(rho, cs, 1.0), v = rho * sn;
      const double nr = re * u - im * v;
      im = fma(re, v, im * u);
      re = nr;
      ang += sy;
    } else {
      re *= 1.0 + rho;
    }
  }
};

// psi from the product of the hidden units' factors and a.x = axr + i axi
template <int FLAVOUR>
__device__ __forceinline__ void write_psi(double *__restrict__ psi, int64_t i, const Prod &P, double axr, double axi) {
  const double kLn2 = 0.693147180559945309417;
  if constexpr (FLAVOUR == PYNQS_RBM_REAL) {
    psi[i] = P.re * exp(axr + P.lin + kLn2 * (double)P.e2);
  } else if constexpr (FLAVOUR == PYNQS_RBM_TANH) {
    psi[i] = tanh(axr) * P.re * exp(P.lin + kLn2 * (double)P.e2);
  } else if constexpr (FLAVOUR == PYNQS_RBM_PHASE) {
    double sn, cs;
    sincos(axr + P.lin + log(P.re) + kLn2 * (double)P.e2, &sn, &cs);  // (the phase IS the logarithm of the real flavour's amplitude)
    psi[2 * i] = cs; psi[2 * i + 1] = sn;
  } else {
    const double m = exp(axr + P.lin + kLn2 * (double)P.e2);
    double sn, cs;
    sincos(axi + P.ang, &sn, &cs);
    psi[2 * i] = m * (P.re * cs - P.im * sn);
    psi[2 * i + 1] = m * (P.re * sn + P.im * cs);
  }
}

// psi of one determinant from scratch (every lane of the wave takes part: the parameter loads are wave-uniform scalar loads)
template <int LEN, int FLAVOUR>
__device__ __forceinline__ void rbm_forward_row(const uint64_t (&ket)[LEN], int sorb, int H, const double *__restrict__ W,
                                                const double *__restrict__ hb, const double *__restrict__ vb, Prod &P, double &axr, double &axi) {
  constexpr bool CPLX = FLAVOUR == PYNQS_RBM_COMPLEX;
  for (int h0 = 0; h0 < H; h0 += kHChunk) {
    double tr[kHChunk], ti[kHChunk];
#pragma unroll
    for (int j = 0; j < kHChunk; ++j) {
      const int h = min(h0 + j, H - 1);
      tr[j] = CPLX ? hb[2 * h] : hb[h];
      ti[j] = CPLX ? hb[2 * h + 1] : 0.0;
    }
    for (int o = 0; o < sorb; ++o) {
      const double x = pm1_of<LEN>(ket, o);
#pragma unroll
      for (int j = 0; j < kHChunk; ++j) {
        const int h = min(h0 + j, H - 1);  // (wave-uniform address: a scalar load)
        if constexpr (CPLX) {
          tr[j] = fma(x, W[((size_t)h * sorb + o) * 2], tr[j]);
          ti[j] = fma(x, W[((size_t)h * sorb + o) * 2 + 1], ti[j]);
        } else {
          tr[j] = fma(x, W[(size_t)h * sorb + o], tr[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kHChunk; ++j)
      if (h0 + j < H) P.template times_2cosh<CPLX>(tr[j], ti[j]);
    P.renorm();
  }
  axr = 0.0; axi = 0.0;
  if (vb) {
    for (int o = 0; o < sorb; ++o) {
      const double x = pm1_of<LEN>(ket, o);
      if constexpr (CPLX) { axr = fma(x, vb[2 * o], axr); axi = fma(x, vb[2 * o + 1], axi); }
      else axr = fma(x, vb[o], axr);
    }
  }
}

template <int LEN, int FLAVOUR>
__global__ __launch_bounds__(kBlock) void rbm_forward_kernel(const uint64_t *__restrict__ onv, int64_t n, int sorb, int H,
                                                             const double *__restrict__ W, const double *__restrict__ hb,
                                                             const double *__restrict__ vb, double *__restrict__ psi) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = i < n ? i : n - 1;  // (idle lanes repeat the last determinant: the parameter loads must stay wave-uniform)
  uint64_t ket[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) ket[w] = onv[row * LEN + w];
  Prod P;  // prod_h 2cosh(theta_h) = exp(P.lin + i P.ang) * (P.re + i P.im) * 2^P.e2
  double axr, axi;
  rbm_forward_row<LEN, FLAVOUR>(ket, sorb, H, W, hb, vb, P, axr, axi);
  if (i >= n) return;
  write_psi<FLAVOUR>(psi, i, P, axr, axi);
}

// ---- psi on the distinct x' of a REDUCE front end, each from its parent walker -----------------------------------------------------
// x' = x with <= 4 orbitals flipped, and   prod_h 2cosh(theta_h) = exp(sum_h theta_h) prod_h (1 + q_h),   q_h = exp(-2 theta_h).
// Flipping orbital o to x'_o = +-1 changes theta_h by +-2 W_ho:  q_h *= exp(-+4 W_ho),  sum_h theta_h += +-2 sum_h W_ho,  a.x += +-2 a_o.
// So a child costs, per hidden unit, 4 (complex) multiplications by table entries and one by (1 + q): no exponential, no sine, no loop over
// the orbitals.  The table (caller-owned, pynqs_rbm_children_table_bytes) holds
//   parents [nwalkers][H + 2] : q_h(x) for h < H, then sum_h theta_h(x), then a.x
//   factors [2 sorb + 1][HP]  : row 2 o (x'_o = +1) / 2 o + 1 (x'_o = -1): exp(-+4 W_ho) for h < H, +-2 sum_h W_ho, +-2 a_o;
//                               the last row (1, ..., 1, 0, 0) stands for "no flip"; HP = H + 2 made odd (rows start in different banks)
//   flag    one double after the factors: non-zero if some parent has Re theta_h < -340, where q_h = exp(-2 theta_h) leaves the range of a
//           double: the children kernel then computes every row from scratch (rbm_forward_row, the plain kernel's body)
// all entries real or (re, im) by the flavour.
__host__ __device__ inline int children_hp(int H) { return (H + 2) | 1; }

template <bool CPLX>
__global__ __launch_bounds__(kBlock) void rbm_children_factors_kernel(int sorb, int H, const double *__restrict__ W, const double *__restrict__ vb,
                                                                      double *__restrict__ factors) {
  constexpr int C = CPLX ? 2 : 1;
  const int HP = children_hp(H);
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx == 0) factors[(size_t)(2 * sorb + 1) * HP * C] = 0.0;  // the flag (the parents kernel, launched next, may raise it)
  if (idx >= (2 * sorb + 1) * HP) return;
  const int row = idx / HP, h = idx - row * HP, o = row >> 1;
  const double sign = (row & 1) ? -1.0 : 1.0;  // x'_o
  double re = 0.0, im = 0.0;
  if (row == 2 * sorb) {
    re = h < H ? 1.0 : 0.0;
  } else if (h < H) {
    const double wr = W[((size_t)h * sorb + o) * C], wi = CPLX ? W[((size_t)h * sorb + o) * C + 1] : 0.0;
    const double m = exp(-4.0 * sign * wr);
    if constexpr (CPLX) {
      double sn, cs;
      sincos(-4.0 * sign * wi, &sn, &cs);
      re = m * cs; im = m * sn;
    } else {
      re = m;
    }
  } else if (h == H) {
#pragma unroll 16
    for (int k = 0; k < H; ++k) {  // (independent loads: 16 in flight)
      re += W[((size_t)k * sorb + o) * C];
      if constexpr (CPLX) im += W[((size_t)k * sorb + o) * C + 1];
    }
    re *= 2.0 * sign; im *= 2.0 * sign;
  } else if (h == H + 1 && vb) {
    re = 2.0 * sign * vb[(size_t)o * C];
    if constexpr (CPLX) im = 2.0 * sign * vb[(size_t)o * C + 1];
  }
  factors[(size_t)idx * C] = re;
  if constexpr (CPLX) factors[(size_t)idx * C + 1] = im;
}

constexpr int kParentChunk = 4;

// one lane per (walker, chunk of kParentChunk = 4 hidden units): blockIdx.y is the chunk, so that W_ho stays wave-uniform and 8192
// walkers are 10 x 128 waves, not 128 (80 -> 28 us with chunks of 8 -> 17 us for Fe2S2).  Chunk 0 also leaves sum_h theta_h = sum_h b_h + sum_o x_o sum_h W_ho (from the factor
// table's sum_h W_ho, built by the launch before this one) and a.x.
template <int LEN, bool CPLX>
__global__ __launch_bounds__(kBlock) void rbm_children_parents_kernel(const uint64_t *__restrict__ onv, int64_t n, int sorb, int H,
                                                                      const double *__restrict__ W, const double *__restrict__ hb,
                                                                      const double *__restrict__ vb, const double *__restrict__ factors,
                                                                      double *__restrict__ table) {
  constexpr int C = CPLX ? 2 : 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = i < n ? i : n - 1;
  const int h0 = (int)blockIdx.y * kParentChunk;
  uint64_t ket[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) ket[w] = onv[row * LEN + w];
  double *__restrict__ out = table + (size_t)row * (size_t)(H + 2) * C;
  double tr[kParentChunk], ti[kParentChunk];
#pragma unroll
  for (int j = 0; j < kParentChunk; ++j) {
    const int h = min(h0 + j, H - 1);
    tr[j] = CPLX ? hb[2 * h] : hb[h];
    ti[j] = CPLX ? hb[2 * h + 1] : 0.0;
  }
  for (int o = 0; o < sorb; ++o) {
    const double x = pm1_of<LEN>(ket, o);
#pragma unroll
    for (int j = 0; j < kParentChunk; ++j) {
      const int h = min(h0 + j, H - 1);  // (wave-uniform address: a scalar load)
      tr[j] = fma(x, W[((size_t)h * sorb + o) * C], tr[j]);
      if constexpr (CPLX) ti[j] = fma(x, W[((size_t)h * sorb + o) * C + 1], ti[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < kParentChunk; ++j) {
    if (h0 + j < H && i < n) {
      if (!(tr[j] > -340.0)) const_cast<double *>(factors)[(size_t)(2 * sorb + 1) * children_hp(H) * C] = 1.0;  // (also for nan)
      const double m = exp(-2.0 * tr[j]);
      if constexpr (CPLX) {
        double sn, cs;
        sincos_mod(-2.0 * ti[j], sn, cs);
        out[(size_t)(h0 + j) * 2] = m * cs;
        out[(size_t)(h0 + j) * 2 + 1] = m * sn;
      } else {
        out[h0 + j] = m;
      }
    }
  }
  if (blockIdx.y != 0) return;
  const int HP = children_hp(H);
  double sr = 0.0, si = 0.0, axr = 0.0, axi = 0.0;
  for (int h = 0; h < H; ++h) {
    sr += hb[(size_t)h * C];
    if constexpr (CPLX) si += hb[(size_t)h * C + 1];
  }
  for (int o = 0; o < sorb; ++o) {
    const double x = pm1_of<LEN>(ket, o);
    // rows 2 o of the factor table: entry H = +2 sum_h W_ho, entry H + 1 = +2 a_o  (wave-uniform addresses)
    const double *__restrict__ f = factors + ((size_t)(2 * o) * HP + H) * C;
    sr = fma(0.5 * x, f[0], sr);
    axr = fma(0.5 * x, f[C], axr);
    if constexpr (CPLX) { si = fma(0.5 * x, f[1], si); axi = fma(0.5 * x, f[C + 1], axi); }
  }
  if (i >= n) return;
  out[(size_t)H * C] = sr;
  out[(size_t)(H + 1) * C] = axr;
  if constexpr (CPLX) { out[(size_t)H * C + 1] = si; out[(size_t)(H + 1) * C + 1] = axi; }
}

#ifndef PYNQS_CHILD_BLOCK
#define PYNQS_CHILD_BLOCK 1024
#endif
constexpr int kChildBlock = PYNQS_CHILD_BLOCK;  // the factor table (up to 64 KB of LDS) is shared by the workgroup's waves: large workgroups, more waves per CU

template <int LEN, int FLAVOUR>
__global__ __launch_bounds__(kChildBlock) void rbm_forward_children_kernel(const uint64_t *__restrict__ onv, int64_t n, const int32_t *__restrict__ count_dev,
                                                                      const int32_t *__restrict__ parent, const uint64_t *__restrict__ walkers,
                                                                      int64_t nwalkers, const double *__restrict__ table,
                                                                      const double *__restrict__ factors, int sorb, int H,
                                                                      const double *__restrict__ W, const double *__restrict__ hb,
                                                                      const double *__restrict__ vb, double *__restrict__ psi) {
  constexpr bool CPLX = FLAVOUR == PYNQS_RBM_COMPLEX;
  constexpr int C = CPLX ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double wl[];
  const int HP = children_hp(H);
  int64_t cnt = n;
  if (count_dev) cnt = min((int64_t)max(*count_dev, 0), n);
  if (factors[(size_t)(2 * sorb + 1) * HP * C] != 0.0) {  // (grid-uniform) parents out of range: every row from scratch
    const int64_t rounds = (cnt + (int64_t)gridDim.x * kChildBlock - 1) / ((int64_t)gridDim.x * kChildBlock);
    for (int64_t r = 0; r < rounds; ++r) {  // (whole waves iterate together: the parameter loads are wave-uniform)
      const int64_t i = ((int64_t)r * gridDim.x + blockIdx.x) * kChildBlock + threadIdx.x;
      const int64_t row = i < cnt ? i : cnt - 1;
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = onv[row * LEN + w];
      Prod P;
      double axr, axi;
      rbm_forward_row<LEN, FLAVOUR>(ket, sorb, H, W, hb, vb, P, axr, axi);
      if (i < cnt) write_psi<FLAVOUR>(psi, i, P, axr, axi);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < (2 * sorb + 1) * HP * C; idx += kChildBlock) wl[idx] = factors[idx];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kChildBlock + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * kChildBlock) {
    int64_t p = parent[i];
    // a row that is NOT its parent with at most four orbitals flipped (a parent index out of range, a row the caller put there itself) is
    // computed from scratch below instead of being truncated to four flips silently
    bool stranger = p < 0 || p >= nwalkers;
    p = stranger ? 0 : p;
    // the rows of the factor table for the flipped orbitals ("no flip" for the unused slots)
    int at[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) at[q] = 2 * sorb * HP;
    int k = 0, nflip = 0;
#pragma unroll
    for (int w = 0; w < LEN; ++w) {
      const uint64_t xc = onv[i * LEN + w];
      uint64_t d = xc ^ walkers[p * LEN + w];
      nflip += __popcll(d);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (d && k < 4) {
          const int b = __builtin_ctzll(d);
          d &= d - 1;
          const int r = (2 * (64 * w + b) + (((xc >> b) & 1ull) ? 0 : 1)) * HP;
          // (k is a small per-lane counter: the slots are selected, not indexed)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == k) at[q] = r;
          ++k;
        }
      }
    }
    const double *__restrict__ tp = table + (size_t)p * (size_t)(H + 2) * C;
    Prod P;
    for (int h0 = 0; h0 < H; h0 += kHChunk) {
#pragma unroll
      for (int j = 0; j < kHChunk; ++j) {
        const int h = h0 + j;
        if (h < H) {
          double qr = tp[(size_t)h * C], qi = CPLX ? tp[(size_t)h * C + 1] : 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double fr = wl[(size_t)(at[q] + h) * C];
            if constexpr (CPLX) {
              const double fi = wl[(size_t)(at[q] + h) * C + 1];
              const double nr = qr * fr - qi * fi;
              qi = fma(qr, fi, qi * fr);
              qr = nr;
            } else {
              qr *= fr;
            }
          }
          if constexpr (CPLX) {  // P *= 1 + q
            const double u = 1.0 + qr;
            const double nr = P.re * u - P.im * qi;
            P.im = fma(P.re, qi, P.im * u);
            P.re = nr;
          } else {
            P.re *= 1.0 + qr;
          }
        }
      }
      P.renorm();
    }
    double axr = tp[(size_t)(H + 1) * C], axi = CPLX ? tp[(size_t)(H + 1) * C + 1] : 0.0;
    P.lin = tp[(size_t)H * C];
    P.ang = CPLX ? tp[(size_t)H * C + 1] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      P.lin += wl[(size_t)(at[q] + H) * C];
      axr += wl[(size_t)(at[q] + H + 1) * C];
      if constexpr (CPLX) {
        P.ang += wl[(size_t)(at[q] + H) * C + 1];
        axi += wl[(size_t)(at[q] + H + 1) * C + 1];
      }
    }
    stranger = stranger || nflip > 4;
    if (__ballot(stranger)) {  // (rare: the whole wave walks the parameters, the strangers keep the result)
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = onv[i * LEN + w];
      Prod P2;
      double a2r, a2i;
      rbm_forward_row<LEN, FLAVOUR>(ket, sorb, H, W, hb, vb, P2, a2r, a2i);
      if (stranger) { P = P2; axr = a2r; axi = a2i; }
    }
    write_psi<FLAVOUR>(psi, i, P, axr, axi);
  }
}

// The same for factor tables beyond the LDS (sorb x num_hidden above ~64 x 64: 235 KB at 120 x 120): ONE WAVE PER ROW, the lanes over the
// hidden units.  Lane l forms q_h = q_h(parent) prod_{o in F} f_h(o) and its product of (1 + q_h) for h = l, l + 64, ...: the parent's row
// and the (at most four) factor rows are read as consecutive 8- / 16-byte words by consecutive lanes, from the L2 (the factor table is shared
// by every row, a parent's row by its few hundred children); the 64 partial products meet in a butterfly (mantissa and exponent kept apart:
// Prod).  4 row loads per hidden unit instead of sorb multiply-adds.  What is the same for all lanes of the row (the parent, the flipped
// orbitals, the table offsets) is computed on the scalar unit: the row number goes through readfirstlane, which tells the compiler so.
template <int LEN, int FLAVOUR>
__global__ __launch_bounds__(kBlock) void rbm_forward_children_wave_kernel(const uint64_t *__restrict__ onv, int64_t n, const int32_t *__restrict__ count_dev,
                                                                           const int32_t *__restrict__ parent, const uint64_t *__restrict__ walkers,
                                                                           int64_t nwalkers, const double *__restrict__ table,
                                                                           const double *__restrict__ factors, int sorb, int H,
                                                                           const double *__restrict__ W, const double *__restrict__ hb,
                                                                           const double *__restrict__ vb, double *__restrict__ psi) {
  constexpr bool CPLX = FLAVOUR == PYNQS_RBM_COMPLEX;
  constexpr int C = CPLX ? 2 : 1;
  const int HP = children_hp(H);
  int64_t cnt = n;
  if (count_dev) cnt = min((int64_t)max(*count_dev, 0), n);
  if (factors[(size_t)(2 * sorb + 1) * HP * C] != 0.0) {  // (grid-uniform) parents out of range: every row from scratch, a thread per row
    const int64_t rounds = (cnt + (int64_t)gridDim.x * kBlock - 1) / ((int64_t)gridDim.x * kBlock);
    for (int64_t r = 0; r < rounds; ++r) {
      const int64_t i = ((int64_t)r * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      const int64_t row = i < cnt ? i : cnt - 1;
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = onv[row * LEN + w];
      Prod P;
      double axr, axi;
      rbm_forward_row<LEN, FLAVOUR>(ket, sorb, H, W, hb, vb, P, axr, axi);
      if (i < cnt) write_psi<FLAVOUR>(psi, i, P, axr, axi);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  // a wave takes kWaveRows consecutive rows at a time (the children of a walker follow each other in the distinct list: the parent's row and
  // the factor rows of its occupied orbitals stay in the CU's L1)
  constexpr int kWaveRows = 16;
  for (int64_t i0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kWaveRows; i0 < cnt; i0 += nwaves * kWaveRows)
  for (int64_t i = i0; i < min(i0 + kWaveRows, cnt); ++i) {  // (wave-uniform)
    int64_t p = parent[i];
    bool stranger = p < 0 || p >= nwalkers;  // (see the kernel above)
    p = stranger ? 0 : p;
    int at[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) at[q] = 2 * sorb * HP;
    int k = 0, nflip = 0;
#pragma unroll
    for (int w = 0; w < LEN; ++w) {
      const uint64_t xc = onv[i * LEN + w];
      uint64_t d = xc ^ walkers[p * LEN + w];
      nflip += __popcll(d);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (d && k < 4) {
          const int b = __builtin_ctzll(d);
          d &= d - 1;
          const int r = (2 * (64 * w + b) + (((xc >> b) & 1ull) ? 0 : 1)) * HP;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == k) at[q] = r;
          ++k;
        }
      }
    }
    if (stranger || nflip > 4) {  // (wave-uniform, rare: from scratch, every lane the same row)
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = onv[i * LEN + w];
      Prod P2;
      double a2r, a2i;
      rbm_forward_row<LEN, FLAVOUR>(ket, sorb, H, W, hb, vb, P2, a2r, a2i);
      if (lane == 0) write_psi<FLAVOUR>(psi, i, P2, a2r, a2i);
      continue;
    }
    const double *__restrict__ tp = table + (size_t)p * (size_t)(H + 2) * C;
    Prod P;
    for (int h = lane; h < H; h += 64) {
      double qr = tp[(size_t)h * C], qi = CPLX ? tp[(size_t)h * C + 1] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double fr = factors[(size_t)(at[q] + h) * C];
        if constexpr (CPLX) {
          const double fi = factors[(size_t)(at[q] + h) * C + 1];
          const double nr = qr * fr - qi * fi;
          qi = fma(qr, fi, qi * fr);
          qr = nr;
        } else {
          qr *= fr;
        }
      }
      if constexpr (CPLX) {  // P *= 1 + q
        const double u = 1.0 + qr;
        const double nr = P.re * u - P.im * qi;
        P.im = fma(P.re, qi, P.im * u);
        P.re = nr;
      } else {
        P.re *= 1.0 + qr;
      }
      P.renorm();
    }
    // the lanes' products: mantissas multiplied, exponents added, in a butterfly (every lane ends with the full product)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const double orr = __shfl_xor(P.re, d), oi = CPLX ? __shfl_xor(P.im, d) : 0.0;
      const int oe = __shfl_xor(P.e2, d);
      if constexpr (CPLX) {
        const double nr = P.re * orr - P.im * oi;
        P.im = fma(P.re, oi, P.im * orr);
        P.re = nr;
      } else {
        P.re *= orr;
      }
      P.e2 += oe;   // (mantissas in [0.5, 1): 64 of them multiply to >= 2^-64, no renormalisation on the way)
    }
    P.renorm();
    if (lane == 0) {
      double axr = tp[(size_t)(H + 1) * C], axi = CPLX ? tp[(size_t)(H + 1) * C + 1] : 0.0;
      P.lin = tp[(size_t)H * C];
      P.ang = CPLX ? tp[(size_t)H * C + 1] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        P.lin += factors[(size_t)(at[q] + H) * C];
        axr += factors[(size_t)(at[q] + H + 1) * C];
        if constexpr (CPLX) {
          P.ang += factors[(size_t)(at[q] + H) * C + 1];
          axi += factors[(size_t)(at[q] + H + 1) * C + 1];
        }
      }
      write_psi<FLAVOUR>(psi, i, P, axr, axi);
    }
  }
}

}  // namespace pynqs

using namespace pynqs;

extern "C" int pynqs_rbm_forward(const uint64_t *onv, int64_t n, int sorb, const double *weights, const double *hidden_bias,
                                 const double *visible_bias, int nhidden, int flavour, double *psi, void *stream) {
  pynqs::DeviceScope device_scope_(onv);
  if (n < 0 || n > 0x7fffffffll * kBlock || sorb < 1 || sorb > kMaxSorb || nhidden < 1) return set_error(PYNQS_EINVAL, "bad n/sorb/nhidden");
  if (flavour != PYNQS_RBM_REAL && flavour != PYNQS_RBM_TANH && flavour != PYNQS_RBM_PHASE && flavour != PYNQS_RBM_COMPLEX)
    return set_error(PYNQS_EINVAL, "bad flavour");
  if (n == 0) return PYNQS_OK;
  if (!onv || !weights || !hidden_bias || !psi) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint32_t grid = (uint32_t)((n + kBlock - 1) / kBlock);
  hipStream_t st = (hipStream_t)stream;
#define PYNQS_RF(F) hipLaunchKernelGGL((rbm_forward_kernel<LEN, F>), dim3(grid), dim3(kBlock), 0, st, onv, n, sorb, nhidden, weights, hidden_bias, visible_bias, psi)
  DISPATCH_LEN(len, {
    switch (flavour) {
      case PYNQS_RBM_REAL: PYNQS_RF(PYNQS_RBM_REAL); break;
      case PYNQS_RBM_TANH: PYNQS_RF(PYNQS_RBM_TANH); break;
      case PYNQS_RBM_PHASE: PYNQS_RF(PYNQS_RBM_PHASE); break;
      default: PYNQS_RF(PYNQS_RBM_COMPLEX); break;
    }
  });
#undef PYNQS_RF
  return check_launch("rbm_forward");
}

static size_t children_lds_bytes(int sorb, int nhidden, int flavour) {
  return (size_t)(2 * sorb + 1) * (size_t)children_hp(nhidden) * (flavour == PYNQS_RBM_COMPLEX ? 16 : 8);
}

static bool children_flavour_ok(int flavour) {
  return flavour == PYNQS_RBM_REAL || flavour == PYNQS_RBM_TANH || flavour == PYNQS_RBM_PHASE || flavour == PYNQS_RBM_COMPLEX;
}

// (every shape: the factor table in LDS when it fits 64 KB, else a wave per row with the table read from the L2)
extern "C" int pynqs_rbm_forward_children_supported(int sorb, int nhidden, int flavour) {
  if (sorb < 1 || sorb > kMaxSorb || nhidden < 1 || !children_flavour_ok(flavour)) return 0;
  return 1;
}
static bool children_table_in_lds(int sorb, int nhidden, int flavour) {
  static const int force = getenv("PYNQS_RBM_CHILDREN_WAVE") ? atoi(getenv("PYNQS_RBM_CHILDREN_WAVE")) : -1;  // 1: the wave form everywhere
  return force != 1 && children_lds_bytes(sorb, nhidden, flavour) <= 64 * 1024;
}

extern "C" int64_t pynqs_rbm_children_table_bytes(int64_t nwalkers, int sorb, int nhidden, int flavour) {
  if (nwalkers < 0 || sorb < 1 || sorb > kMaxSorb || nhidden < 1 || !children_flavour_ok(flavour)) return -1;
  const int64_t c = flavour == PYNQS_RBM_COMPLEX ? 16 : 8;
  return nwalkers * (nhidden + 2) * c + (int64_t)children_lds_bytes(sorb, nhidden, flavour) + 8;
}

extern "C" int pynqs_rbm_children_prepare(const uint64_t *walkers, int64_t nwalkers, int sorb, const double *weights, const double *hidden_bias,
                                          const double *visible_bias, int nhidden, int flavour, void *table, void *stream) {
  pynqs::DeviceScope device_scope_(table);
  if (nwalkers < 0 || nwalkers > 0x7fffffffll * kBlock || sorb < 1 || sorb > kMaxSorb || nhidden < 1 || !children_flavour_ok(flavour))
    return set_error(PYNQS_EINVAL, "bad nwalkers/sorb/nhidden/flavour");
  if (!weights || !hidden_bias || !table || (nwalkers > 0 && !walkers)) return set_error(PYNQS_EINVAL, "null pointer");
  const bool cplx = flavour == PYNQS_RBM_COMPLEX;
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  double *parents = (double *)table;
  double *factors = parents + (size_t)nwalkers * (size_t)(nhidden + 2) * (cplx ? 2 : 1);
  const uint32_t gf = (uint32_t)(((2 * sorb + 1) * children_hp(nhidden) + kBlock - 1) / kBlock);
  if (cplx) hipLaunchKernelGGL((rbm_children_factors_kernel<true>), dim3(gf), dim3(kBlock), 0, st, sorb, nhidden, weights, visible_bias, factors);
  else hipLaunchKernelGGL((rbm_children_factors_kernel<false>), dim3(gf), dim3(kBlock), 0, st, sorb, nhidden, weights, visible_bias, factors);
  if (nwalkers > 0) {
    const dim3 grid((uint32_t)((nwalkers + kBlock - 1) / kBlock), (uint32_t)((nhidden + kParentChunk - 1) / kParentChunk));
    if (grid.y > 65535u) return set_error(PYNQS_EINVAL, "too many hidden units");
    DISPATCH_LEN(len, {
      if (cplx)
        hipLaunchKernelGGL((rbm_children_parents_kernel<LEN, true>), grid, dim3(kBlock), 0, st, walkers, nwalkers, sorb, nhidden, weights,
                           hidden_bias, visible_bias, factors, parents);
      else
        hipLaunchKernelGGL((rbm_children_parents_kernel<LEN, false>), grid, dim3(kBlock), 0, st, walkers, nwalkers, sorb, nhidden, weights,
                           hidden_bias, visible_bias, factors, parents);
    });
  }
  return check_launch("rbm_children_prepare");
}

extern "C" int pynqs_rbm_forward_children(const uint64_t *onv, int64_t n, const int32_t *count_dev, const int32_t *parent,
                                          const uint64_t *walkers, int64_t nwalkers, const void *table, int sorb, const double *weights,
                                          const double *hidden_bias, const double *visible_bias, int nhidden, int flavour, double *psi,
                                          void *stream) {
  pynqs::DeviceScope device_scope_(onv);
  if (n < 0 || n > 0x7fffffffll * kBlock || nwalkers < 0 || sorb < 1 || sorb > kMaxSorb || nhidden < 1) return set_error(PYNQS_EINVAL, "bad n/sorb/nhidden");
  if (!pynqs_rbm_forward_children_supported(sorb, nhidden, flavour))
    return set_error(PYNQS_EINVAL, "rbm_forward_children: bad flavour");
  if (n == 0) return PYNQS_OK;
  if (!onv || !parent || !walkers || !table || !psi || !weights || !hidden_bias || nwalkers == 0) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  int64_t blocks = (n + kChildBlock - 1) / kChildBlock;
  if (blocks > 1024) blocks = 1024;  // (a workgroup copies the factor table once and strides over the rows)
  const uint32_t grid = (uint32_t)blocks;
  const size_t lds = children_lds_bytes(sorb, nhidden, flavour);
  const double *parents = (const double *)table;
  const double *factors = parents + (size_t)nwalkers * (size_t)(nhidden + 2) * (flavour == PYNQS_RBM_COMPLEX ? 2 : 1);
  hipStream_t st = (hipStream_t)stream;
  const bool in_lds = children_table_in_lds(sorb, nhidden, flavour);
  int64_t wblocks = (n + (kBlock / 64) * 16 - 1) / ((kBlock / 64) * 16);   // the wave form: a wave per 16 consecutive rows, strided
  if (wblocks > 256 * 32) wblocks = 256 * 32;
#define PYNQS_RC(F)                                                                                                                          \
  do {                                                                                                                                       \
    if (in_lds)                                                                                                                              \
      hipLaunchKernelGGL((rbm_forward_children_kernel<LEN, F>), dim3(grid), dim3(kChildBlock), lds, st, onv, n, count_dev, parent, walkers,  \
                         nwalkers, parents, factors, sorb, nhidden, weights, hidden_bias, visible_bias, psi);                                \
    else                                                                                                                                     \
      hipLaunchKernelGGL((rbm_forward_children_wave_kernel<LEN, F>), dim3((uint32_t)wblocks), dim3(kBlock), 0, st, onv, n, count_dev,        \
                         parent, walkers, nwalkers, parents, factors, sorb, nhidden, weights, hidden_bias, visible_bias, psi);               \
  } while (0)
  DISPATCH_LEN(len, {
    switch (flavour) {
      case PYNQS_RBM_REAL: PYNQS_RC(PYNQS_RBM_REAL); break;
      case PYNQS_RBM_TANH: PYNQS_RC(PYNQS_RBM_TANH); break;
      case PYNQS_RBM_PHASE: PYNQS_RC(PYNQS_RBM_PHASE); break;
      default: PYNQS_RC(PYNQS_RBM_COMPLEX); break;
    }
  });
#undef PYNQS_RC
  return check_launch("rbm_forward_children");
}
