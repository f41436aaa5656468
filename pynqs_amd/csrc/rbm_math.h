// rbm_math.h -- device helpers shared by the RBM amplitude kernels that work from the packed bits (kernels_rbm_forward.hip,
// kernels_rbm_grad.hip; vmc/ansatz/rbm/rbm.py:186-211)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pynqs {

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

}  // namespace pynqs
