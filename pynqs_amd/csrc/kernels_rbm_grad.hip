// kernels_rbm_grad.hip -- the energy-gradient estimator of vmc/grad/energy_grad.py:118-184 ("AD" method)
//     loss = 2 Re sum_n p_n conj(ln psi(x_n)) (E_loc(x_n) - <E> c_n),      grad = d loss / d parameters
// for the reference's RBM amplitudes (vmc/ansatz/rbm/rbm.py:186-211), analytically and straight from the packed bits: what
// loss.backward() computes through ~150 autograd kernels (0.3 ms per 8192 walkers even when replayed from a HIP graph) is
//     G_k = sum_n conj(f_n) O_k(x_n),   f_n = p_n (E_loc(x_n) - <E> c_n),   O_k = d ln psi / d theta_k:
//     O_{a_o} = x_o,   O_{b_h} = tanh(theta_h),   O_{W_ho} = tanh(theta_h) x_o,   theta_h = b_h + sum_o W_ho x_o,
// with  grad = 2 Re G  for real parameters and  (d/d Re, d/d Im) = (2 Re G, -2 Im G)  for complex parameters stored as (re, im) pairs
// (ln psi is holomorphic in them).  [n x H] x [n x sorb] outer products -- 26 MFLOP for 8192 Fe2S2 walkers.
//   kernel 1: a workgroup takes 32 walkers.  Thread (walker w, group q of 8) computes theta and tanh for 4 of every 32 hidden units
//             and leaves tanh(theta_h) conj(f_w) in LDS; then the 256 threads share out the 32 (sorb + 1) outputs of those hidden units
//             and each sums its output over the 32 walkers (LDS broadcasts, +- from the walker's bit).  Partial sums per workgroup go
//             to the workspace; the loss' share of the 32 walkers too.  (64 walkers per workgroup, 8 hidden units per thread: 58 us
//             instead of 30 for 8192 Fe2S2 walkers -- 512 waves, each alone on its SIMD with twice the chain.)
//   kernel 2: one thread per parameter adds the workgroups' partial sums in their fixed order: the gradient is bit-reproducible.
#include "detcore.h"
#include "launch.h"
#include "rbm_math.h"

namespace pynqs {

constexpr int kGradWalkers = 32;  // per workgroup
constexpr int kGradHidden = 32;   // per pass: 4 per thread, 8 threads per walker

template <int LEN, bool CPLX>
__global__ __launch_bounds__(kBlock) void rbm_grad_partial_kernel(const uint64_t *__restrict__ onv, int64_t n, int sorb, int H,
                                                                  const double *__restrict__ W, const double *__restrict__ hb,
                                                                  const double *__restrict__ vb, const double *__restrict__ prob,
                                                                  const double *__restrict__ eloc, bool eloc_cplx,
                                                                  const double *__restrict__ e_total, const double *__restrict__ pw,
                                                                  double *__restrict__ partial, int64_t stride) {
  constexpr int C = CPLX ? 2 : 1;
  __shared__ uint64_t xs[kGradWalkers][LEN];
  __shared__ double tc[kGradWalkers][kGradHidden + 1][C];  // tanh(theta_h) conj(f_w) (real parameters: tanh(theta_h) Re f_w); +1: bank spread
  __shared__ double cf[kGradWalkers][C];                   // conj(f_w)
  constexpr int NQ = kBlock / kGradWalkers;  // threads per walker
  const int tid = threadIdx.x, w = tid % kGradWalkers, q = tid / kGradWalkers;
  const int64_t i = (int64_t)blockIdx.x * kGradWalkers + w;
  const bool valid = i < n;
  const int64_t row = valid ? i : n - 1;
  uint64_t ket[LEN];
#pragma unroll
  for (int k = 0; k < LEN; ++k) ket[k] = onv[row * LEN + k];
  // f = p (E_loc - <E> c)
  double fr = 0.0, fi = 0.0;
  if (valid) {
    const double pr = prob[i], c = pw ? pw[i] : 1.0;
    const double er = eloc_cplx ? eloc[2 * i] : eloc[i], ei = eloc_cplx ? eloc[2 * i + 1] : 0.0;
    const double tr = e_total[0], ti = eloc_cplx ? e_total[1] : 0.0;
    fr = pr * (er - tr * c);
    fi = pr * (ei - ti * c);
  }
  if (q == 0) {
#pragma unroll
    for (int k = 0; k < LEN; ++k) xs[w][k] = ket[k];
    cf[w][0] = fr;
    if constexpr (CPLX) cf[w][1] = -fi;
  }
  double *__restrict__ out = partial + (int64_t)blockIdx.x * stride;
  // ln psi of this thread's hidden units (for the loss): Re and Im accumulated per hidden unit, Im wrapped at the end
  double lre = 0.0, lim = 0.0;
  const int SP = sorb + 1;  // outputs per hidden unit: W[h][0..sorb-1], b[h]
  for (int h0 = 0; h0 < H; h0 += kGradHidden) {
    // ---- theta, tanh for hidden units h0 + 8 q .. + 8 of walker w
    constexpr int HC = kGradHidden / NQ;
    double tr[HC], ti[HC];
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const int h = min(h0 + HC * q + j, H - 1);
      tr[j] = CPLX ? hb[2 * h] : hb[h];
      ti[j] = CPLX ? hb[2 * h + 1] : 0.0;
    }
    for (int o = 0; o < sorb; ++o) {
      const double x = pm1_of<LEN>(ket, o);
#pragma unroll
      for (int j = 0; j < HC; ++j) {
        const int h = min(h0 + HC * q + j, H - 1);  // (two addresses per wave)
        if constexpr (CPLX) {
          tr[j] = fma(x, W[((size_t)h * sorb + o) * 2], tr[j]);
          ti[j] = fma(x, W[((size_t)h * sorb + o) * 2 + 1], ti[j]);
        } else {
          tr[j] = fma(x, W[(size_t)h * sorb + o], tr[j]);
        }
      }
    }
    if (h0) __syncthreads();  // the previous pass' sums have read tc
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const int hh = HC * q + j;
      const bool live = h0 + hh < H;
      // tanh(a + ib) = (s (1 - e^2) + 2 i e sin 2b) / (1 + e^2 + 2 e cos 2b),  e = exp(-2 |a|), s = sign(a)
      // 2cosh(a + ib) = exp(s (a + ib)) (1 + e exp(-2 i s b))
      const double ax = fabs(tr[j]), e = exp(-2.0 * ax), s = tr[j] < 0.0 ? -1.0 : 1.0;
      double yr, yi = 0.0;
      if constexpr (CPLX) {
        double sn, cs;
        sincos_mod(2.0 * ti[j], sn, cs);
        const double den = fma(2.0 * e, cs, fma(e, e, 1.0));
        yr = s * (1.0 - e * e) / den;
        yi = 2.0 * e * sn / den;
        if (live) {
          const double u = fma(e, cs, 1.0), v = -s * e * sn;  // 1 + e exp(-2 i s b)
          lre += ax + 0.5 * log(u * u + v * v);
          lim += s * ti[j] + atan2(v, u);
        }
        tc[w][hh][0] = live ? yr * fr + yi * fi : 0.0;  // (yr + i yi) (fr - i fi)
        tc[w][hh][1] = live ? yi * fr - yr * fi : 0.0;
      } else {
        yr = s * (1.0 - e) / (1.0 + e);
        if (live) lre += ax + log1p(e);
        tc[w][hh][0] = live ? yr * fr : 0.0;
      }
    }
    __syncthreads();
    // ---- this pass' outputs: (hh, o), o = sorb: the hidden bias
    const int nout = min(kGradHidden, H - h0) * SP;
    for (int k = tid; k < nout; k += kBlock) {
      const int hh = k / SP, o = k - hh * SP;
      double ar = 0.0, ai = 0.0;
      if (o < sorb) {
        const int word = o >> 6, bit = o & 63;
#pragma unroll 8
        for (int v = 0; v < kGradWalkers; ++v) {
          const bool up = (xs[v][word] >> bit) & 1ull;
          const double a = tc[v][hh][0];
          ar += up ? a : -a;
          if constexpr (CPLX) { const double b = tc[v][hh][1]; ai += up ? b : -b; }
        }
      } else {
#pragma unroll 8
        for (int v = 0; v < kGradWalkers; ++v) {
          ar += tc[v][hh][0];
          if constexpr (CPLX) ai += tc[v][hh][1];
        }
      }
      const int64_t at = (int64_t)(h0 + hh) * SP + o;
      out[C * at] = ar;
      if constexpr (CPLX) out[C * at + 1] = ai;
    }
  }
  // ---- visible bias: sum_w conj(f_w) x_wo, and a.x for the loss
  const int64_t off_vb = (int64_t)H * SP;
  for (int o = tid; o < sorb; o += kBlock) {
    const int word = o >> 6, bit = o & 63;
    double ar = 0.0, ai = 0.0;
    for (int v = 0; v < kGradWalkers; ++v) {
      const bool up = (xs[v][word] >> bit) & 1ull;
      ar += up ? cf[v][0] : -cf[v][0];
      if constexpr (CPLX) ai += up ? cf[v][1] : -cf[v][1];
    }
    out[C * (off_vb + o)] = ar;
    if constexpr (CPLX) out[C * (off_vb + o) + 1] = ai;
  }
  // ---- loss: 2 Re conj(ln psi) f = 2 (Re L Re f + Im L Im f); the four waves hold four shares of the hidden units' sum
  if (vb && q == 0) {
    for (int o = 0; o < sorb; ++o) {
      const double x = pm1_of<LEN>(ket, o);
      lre = fma(x, CPLX ? vb[2 * o] : vb[o], lre);
      if constexpr (CPLX) lim = fma(x, vb[2 * o + 1], lim);
    }
  }
  __syncthreads();
  double *lr = &tc[0][0][0];  // (tc is free now) [NQ][walkers][2]
  lr[(q * kGradWalkers + w) * 2] = lre;
  lr[(q * kGradWalkers + w) * 2 + 1] = lim;
  __syncthreads();
  if (tid < 64) {  // (one wave; lanes past the walkers add 0)
    double l = 0.0;
    if (q == 0) {
      double re = 0.0, im = 0.0;
#pragma unroll
      for (int k = 0; k < NQ; ++k) { re += lr[(k * kGradWalkers + w) * 2]; im += lr[(k * kGradWalkers + w) * 2 + 1]; }
      im -= 6.283185307179586476925 * rint(im * 0.15915494309189533577);  // the principal value torch.log takes
      l = valid ? 2.0 * (re * fr + im * fi) : 0.0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) l += __shfl_xor(l, d);
    if (tid == 0) out[C * (off_vb + sorb)] = l;
  }
}

// grad[k] = 2 Re G_k (and -2 Im G_k), G_k = the workgroups' partial sums in a fixed order; parameter layout of the module:
// weights [H][sorb](x2), hidden_bias [H](x2), visible_bias [sorb](x2)
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void rbm_grad_reduce_kernel(const double *__restrict__ partial, int64_t stride, int ngroups, int sorb, int H,
                                                                 double *__restrict__ gw, double *__restrict__ ghb, double *__restrict__ gvb,
                                                                 double *__restrict__ loss) {
  constexpr int C = CPLX ? 2 : 1;
  constexpr int NS = kBlock / 64;  // a block owns 64 outputs; its NS waves take contiguous slices of the workgroups' partial sums
  __shared__ double part[NS][64][2];
  const int SP = sorb + 1;
  const int64_t nout = (int64_t)H * SP + sorb + 1;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.x * 64 + lane;
  const int per = (ngroups + NS - 1) / NS, g_lo = min(slice * per, ngroups), g_hi = min(g_lo + per, ngroups);
  double re = 0.0, im = 0.0;
  if (k < nout) {
    constexpr int RU = 16;  // loads in flight (the additions keep their order)
    for (int g0 = g_lo; g0 < g_hi; g0 += RU) {
      double vr[RU], vi[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const bool in = g0 + u < g_hi;
        vr[u] = in ? partial[(int64_t)(g0 + u) * stride + C * k] : 0.0;
        vi[u] = CPLX && in ? partial[(int64_t)(g0 + u) * stride + C * k + 1] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) { re += vr[u]; im += vi[u]; }
    }
  }
  part[slice][lane][0] = re;
  part[slice][lane][1] = im;
  __syncthreads();
  if (slice != 0 || k >= nout) return;
  re = 0.0; im = 0.0;
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) { re += part[sl][lane][0]; im += part[sl][lane][1]; }  // fixed order: reproducible
  if (k == nout - 1) {
    if (loss) loss[0] = re;
    return;
  }
  double *dst;
  int64_t at;
  if (k < (int64_t)H * SP) {
    const int64_t h = k / SP;
    const int o = (int)(k - h * SP);
    if (o < sorb) { dst = gw; at = h * sorb + o; }
    else { dst = ghb; at = h; }
  } else {
    dst = gvb; at = k - (int64_t)H * SP;
  }
  if (!dst) return;
  dst[C * at] = 2.0 * re;
  if constexpr (CPLX) dst[C * at + 1] = -2.0 * im;
}

static inline int64_t grad_stride(int sorb, int H, bool cplx) {  // doubles per workgroup
  return ((int64_t)H * (sorb + 1) + sorb + 1) * (cplx ? 2 : 1);
}

}  // namespace pynqs

using namespace pynqs;

extern "C" int64_t pynqs_rbm_grad_workspace(int64_t n, int sorb, int nhidden, int flavour) {
  if (n < 0 || sorb < 1 || sorb > kMaxSorb || nhidden < 1 || (flavour != PYNQS_RBM_REAL && flavour != PYNQS_RBM_COMPLEX)) return -1;
  const int64_t groups = (n + kGradWalkers - 1) / kGradWalkers;
  return groups * grad_stride(sorb, nhidden, flavour == PYNQS_RBM_COMPLEX) * 8;
}

extern "C" int pynqs_rbm_grad(const uint64_t *onv, int64_t n, int sorb, const double *weights, const double *hidden_bias,
                              const double *visible_bias, int nhidden, int flavour, const double *prob, const double *eloc,
                              int eloc_is_complex, const double *e_total, const double *pow, double *grad_weights, double *grad_hidden_bias,
                              double *grad_visible_bias, double *loss, void *workspace, void *stream) {
  pynqs::DeviceScope device_scope_(onv);
  if (n < 0 || n > 0x7fffffffll * kGradWalkers || sorb < 1 || sorb > kMaxSorb || nhidden < 1) return set_error(PYNQS_EINVAL, "bad n/sorb/nhidden");
  if (flavour != PYNQS_RBM_REAL && flavour != PYNQS_RBM_COMPLEX) return set_error(PYNQS_EINVAL, "rbm_grad: flavour must be PYNQS_RBM_REAL or PYNQS_RBM_COMPLEX");
  if (!weights || !hidden_bias || !grad_weights || !grad_hidden_bias || (n > 0 && (!onv || !prob || !eloc || !e_total || !workspace)))
    return set_error(PYNQS_EINVAL, "null pointer");
  const bool cplx = flavour == PYNQS_RBM_COMPLEX;
  const int len = (sorb - 1) / 64 + 1;
  const int64_t groups = (n + kGradWalkers - 1) / kGradWalkers, stride = grad_stride(sorb, nhidden, cplx);
  hipStream_t st = (hipStream_t)stream;
  double *partial = (double *)workspace;
  if (groups > 0) {
#define PYNQS_RG(C)                                                                                                                          \
  hipLaunchKernelGGL((rbm_grad_partial_kernel<LEN, C>), dim3((uint32_t)groups), dim3(kBlock), 0, st, onv, n, sorb, nhidden, weights, hidden_bias, \
                     visible_bias, prob, eloc, eloc_is_complex != 0, e_total, pow, partial, stride)
    DISPATCH_LEN(len, {
      if (cplx) PYNQS_RG(true);
      else PYNQS_RG(false);
    });
#undef PYNQS_RG
  }
  const int64_t nout = (int64_t)nhidden * (sorb + 1) + sorb + 1;
  const uint32_t g2 = (uint32_t)((nout + 63) / 64);
  if (cplx)
    hipLaunchKernelGGL((rbm_grad_reduce_kernel<true>), dim3(g2), dim3(kBlock), 0, st, partial, stride, (int)groups, sorb, nhidden, grad_weights,
                       grad_hidden_bias, grad_visible_bias, loss);
  else
    hipLaunchKernelGGL((rbm_grad_reduce_kernel<false>), dim3(g2), dim3(kBlock), 0, st, partial, stride, (int)groups, sorb, nhidden, grad_weights,
                       grad_hidden_bias, grad_visible_bias, loss);
  return check_launch("rbm_grad");
}
