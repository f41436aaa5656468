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

namespace pynqs {

constexpr int kHChunk = 8;

template <int LEN>
__device__ __forceinline__ double pm1_of(const uint64_t (&ket)[LEN], int o) {
  // +1.0 / -1.0 from the occupation bit: only the sign bit of the double differs
  const uint32_t bit = (uint32_t)(ket[o >> 6] >> (o & 63)) & 1u;
  const uint64_t u = 0x3ff0000000000000ull | ((uint64_t)(bit ^ 1u) << 63);
  return __longlong_as_double((long long)u);
}

// x = k pi/2 + r by two fmas (pi/2 split in two doubles), then the fdlibm kernels on |r| <= pi/4: the arguments here are sums of a few
// dozen parameters, far from the library sincos' large-argument path
__device__ __forceinline__ void sincos_mod(double x, double &sn, double &cs) {
  const double k = rint(x * 0.63661977236758134308);  // 2 / pi
  double r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                     2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s0 = fma(r * z, ps, r), c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k;
  const double a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}

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
};

template <int LEN, int FLAVOUR>
__global__ __launch_bounds__(kBlock) void rbm_forward_kernel(const uint64_t *__restrict__ onv, int64_t n, int sorb, int H,
                                                             const double *__restrict__ W, const double *__restrict__ hb,
                                                             const double *__restrict__ vb, double *__restrict__ psi) {
  constexpr bool CPLX = FLAVOUR == PYNQS_RBM_COMPLEX;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = i < n ? i : n - 1;  // (idle lanes repeat the last determinant: the parameter loads below must stay wave-uniform)
  uint64_t ket[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) ket[w] = onv[row * LEN + w];
  Prod P;  // prod_h 2cosh(theta_h) = exp(P.lin + i P.ang) * (P.re + i P.im) * 2^P.e2
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
    for (int j = 0; j < kHChunk; ++j) {
      if (h0 + j < H) {
        // 2cosh(x + iy) = e^{s(x + iy)} (1 + e^{-2s(x + iy)}),  s = sign(x)
        const double ax = fabs(tr[j]);
        const double rho = exp(-2.0 * ax);
        P.lin += ax;
        if constexpr (CPLX) {
          const double sy = tr[j] < 0.0 ? -ti[j] : ti[j];
          double sn, cs;
          sincos_mod(-2.0 * sy, sn, cs);
          const double u = fma(rho, cs, 1.0), v = rho * sn;
          const double nr = P.re * u - P.im * v;
          P.im = fma(P.re, v, P.im * u);
          P.re = nr;
          P.ang += sy;
        } else {
          P.re *= 1.0 + rho;
        }
      }
    }
    P.renorm();
  }
  double axr = 0.0, axi = 0.0;
  if (vb) {
    for (int o = 0; o < sorb; ++o) {
      const double x = pm1_of<LEN>(ket, o);
      if constexpr (CPLX) { axr = fma(x, vb[2 * o], axr); axi = fma(x, vb[2 * o + 1], axi); }
      else axr = fma(x, vb[o], axr);
    }
  }
  if (i >= n) return;
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
