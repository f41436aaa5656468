// kernels_rbm.hip -- SIMPLE local energy (vmc/energy/eloc.py:121-203, _simple) with the amplitude ratio of a real
// RBM (vmc/ansatz/rbm/rbm.py:186-211) evaluated on chip: nothing of size nbatch x ncomb ever reaches HBM.
//
//   E_loc(x) = sum_x' <x|H|x'> psi(x') / psi(x),     psi(x) = exp(a.x) prod_h 2 cosh(theta_h(x))
//
// An excitation flips 2 or 4 orbitals F (occupied x_o = +1 -> -1, empty -1 -> +1): theta'_h = theta_h - delta_h,
// delta_h = 2 sum_{o in F} W[h][o] x_o.  With s_h = sign(theta_h), rho_h = exp(-2|theta_h|), m_h = 1/(1+rho_h):
//   cosh(theta_h - delta_h) / cosh(theta_h) = exp(-s_h delta_h) * (m_h + m_h rho_h * prod_{o in F} q_h(o)),
//   q_h(o) = exp(4 s_h W[h][o] x_o)
// (no overflow for any theta; the reference's psi(x')/psi(x) of two products over-/underflows first), so
//   psi(x')/psi(x) = prod_{o in F} C(o) * prod_h (m_h + n_h prod_{o in F} q_h(o)),
//   C(o) = exp(-2 x_o (a_o + sum_h s_h W[h][o])),  n_h = m_h rho_h.
// Every excitation multiplies four rows (a single: its two orbitals and a dummy row twice), so the rows kept in
// LDS are q'_h(o) = (m_h rho_h)^(1/4) q_h(o) and a factor is m_h + prod_{o in F} q'_h(o).
// Per walker the workgroup builds q'[o][h] in LDS (one read of exp(+-4W) per element, no transcendental in the
// inner loop) and C(o); then every lane owns a 4 x 4 block of excitations -- 4 entries of a class's "fast"
// excitation table x 4 entries of its "slow" table (hole pairs x particle pairs, alpha singles x beta singles:
// detcore.h) -- and runs over the hidden units with 16 running products in registers:
//   per hidden unit and lane: 16 LDS reads, 8 multiplications for the 8 pair products, 16 x (fma + mul).
// The kernel is bound by the f64 vector rate (84 % busy), not by HBM.  When sorb x
// num_hidden does not fit the LDS the WINDOWED variant streams q' through it (see eloc_rbm_kernel).
// Matrix elements come from the integral plan exactly as in kernels_plan.hip.
//
// FLAVOUR selects the reference's other amplitudes that share the hidden-unit product (rbm.py:199-211):
//   kRbmReal  psi = exp(a.x)  prod_h 2cosh(theta_h)
//   kRbmTanh  psi = tanh(a.x) prod_h 2cosh(theta_h)            -- C(o) without a_o; tanh(a.x') = 1 - 2 / (exp(2 a.x') + 1) with
//             exp(2 a.x') = exp(2 a.x) prod_{o in F} exp(-4 x_o a_o): table products and one division per column
//   kRbmPhase psi = exp(i (a.x + sum_h ln 2cosh(theta_h)))     -- "pRBM": the phase difference of x' and x IS the logarithm of the
//             real flavour's ratio, so psi(x')/psi(x) = exp(i ln t) with t from the same running products; E_loc is complex
// ("cos" and "complex" need complex running products: they take the module path.)
//
// GREEN: the fixed-node Green's-function row of a GFMC step (gfmc/walker.py:167-235, _calculate_green_kernel) from the same
// pass, for the real-valued flavours: with r_k = psi(x'_k)/psi(x) and h_k = <x|H|x'_k>, a move k >= 1 keeps the sign when
// h_k r_k < 0 (the reference's cos(phase difference + gamma) < 0) and then has weight g_k = -h_k r_k; the others add to the
// sign-flip potential v_sf = sum h_k r_k on the diagonal: g_0 = max(0, Lambda - h_0 - v_sf).  The row is written in the
// reference's column order (column = 1 + rank, excitation.cpp:43-109 incl. the `idx % noAA` rotation); E_loc is unchanged.
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "rbm.h"

namespace pynqs {

// W[H][sorb], hb[H], vb[sorb] (row-major, the reference's parameter shapes) -> RBM table (rbm.h)
__global__ __launch_bounds__(kBlock) void rbm_table_kernel(const double *__restrict__ W, const double *__restrict__ hb,
                                                           const double *__restrict__ vb, RbmLayout rl, double *__restrict__ tab) {
  const int64_t row = (int64_t)rl.sorb * rl.Hq;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < row) {
    const int o = (int)(i / rl.Hq), h = (int)(i - (int64_t)o * rl.Hq);
    const double w = h < rl.H ? W[(int64_t)h * rl.sorb + o] : 0.0;
    tab[rl.offWt + i] = w;
    tab[rl.offE4p + i] = exp(4.0 * w);
    tab[rl.offE4m + i] = exp(-4.0 * w);
  }
  if (i < rl.Hq) tab[rl.offHb + i] = i < rl.H ? hb[i] : 0.0;
  if (i < rl.total - rl.offVb) tab[rl.offVb + i] = (vb && i < rl.sorb) ? vb[i] : 0.0;
}

// How the excitations of one walker are cut into 4 x 4 blocks: class k has nbf[k] x nbs[k] blocks
// (k = 0 singles x nothing, 1 alpha-alpha, 2 beta-beta, 3 alpha-beta); b[k] = cumulative block counts.
struct RbmBlocks {
  uint32_t nbf[4];
  uint32_t b[4];
  uint32_t ntiles;  // tiles of 64 blocks
  MagicDiv dv[4];   // division by nbf[k]
};

static inline RbmBlocks make_rbm_blocks(const SDParams &p) {
  RbmBlocks B;
  const uint32_t nf[4] = {p.d1, (uint32_t)p.noAA, (uint32_t)p.noBB, (uint32_t)p.nSa};
  const uint32_t ns[4] = {p.d1 ? 1u : 0u, (uint32_t)p.nvAA, (uint32_t)p.nvBB, (uint32_t)p.nSb};
  uint32_t acc = 0;
  for (int k = 0; k < 4; ++k) {
    B.nbf[k] = (nf[k] + 3) / 4;
    B.dv[k] = make_magic(B.nbf[k]);
    acc += B.nbf[k] * ((ns[k] + 3) / 4);
    B.b[k] = acc;
  }
  B.ntiles = (acc + 63) / 64;
  return B;
}

// LDS after the walker tables (16-byte aligned): [q / staging][mn][sh][Cq][hs]
//   q   [sorb + 1][hw+1] row r(o) = o/2 for alpha, sorb/2 + o/2 for beta orbitals (a wave mostly reads rows of one
//                       spin: consecutive rows -> different bank pairs); row `sorb` belongs to the dummy orbital
//                       (the partner of a single); hw hidden units at a time (all of them in the fast kernel)
//   mn  [2][Hq]         m_h, then (m_h rho_h)^(1/4) (1, 0 in the padding)
//   sh  [Hq]            s_h
//   Cq  [sorb + 2]      C(o) by orbital, 1 for the dummy orbital `sorb`
//   hs  [d1 + 2]        <x|H|x>, then the singles
typedef __attribute__((address_space(3))) const double lds_cdouble;  // read through a 32-bit LDS address

struct RbmLds {
  double *q, *mn, *sh, *Cq, *Aq, *hs;  // Aq [sorb + 2]: exp(-4 x_o a_o) (tanh flavour: exp(2 a.x') = exp(2 a.x) prod over the flipped orbitals)
  uint32_t *rowaddr;  // [sorb + 2]: LDS byte address of an orbital's q row (the dummy's at index sorb)
};

__host__ __device__ inline size_t rbm_q_offset(const SDParams &p) { return (lds_fixed_bytes(p) + 15) & ~(size_t)15; }

// `hw` = hidden units resident in LDS at a time: rl.Hloop (all of them, the fast kernel) or a multiple of 8 (the
// windowed kernel for sorb x num_hidden beyond the LDS); row stride hw + 1 doubles (odd: see rbm.h)
__host__ __device__ inline size_t rbm_region_bytes(const SDParams &p, uint32_t hw) {
  return (((size_t)(p.sorb + 1) * (hw + 1) * 8) + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t lds_bytes_rbm(const SDParams &p, const RbmLayout &rl, uint32_t hw) {
  return rbm_q_offset(p) + rbm_region_bytes(p, hw) +
         8 * (3 * (size_t)rl.Hq + 2 * (size_t)(p.sorb + 2) + (size_t)(p.d1 + 2)) + 4 * (((size_t)p.sorb + 2 + 3) & ~(size_t)3) + 144;  // + red (up to 16 waves), counters
}

__device__ __forceinline__ uint32_t rbm_row(uint32_t o, uint32_t K) { return (o >> 1) + ((o & 1u) ? K : 0u); }

// WINDOWED = false: all hidden units of q' live in LDS, waves pull tiles from a counter (the fast kernel).
// WINDOWED = true : sorb x num_hidden does not fit: the workgroup streams q' through LDS `hw` hidden units at a time;
//                   in every round each wave holds the 16 x 64 running products of ONE tile in registers across the
//                   windows (two barriers per window).
enum : int { kRbmReal = 0, kRbmTanh = 1, kRbmPhase = 2 };

// sin and cos of a moderate argument (|x| < 1e5: here the logarithm of an amplitude ratio) without the library's large-argument
// path, whose code and registers the 16-column epilogue of the phase flavour pays for on every call: x = k pi/2 + r by two fmas
// (pi/2 split in two doubles), then the fdlibm kernels on |r| <= pi/4 (errors < 1 ulp of the result for these magnitudes).
__device__ __forceinline__ void sincos_moderate(double x, double &sn, double &cs) {
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

template <int LEN, bool WINDOWED, int FLAVOUR, bool GREEN>
__global__ __launch_bounds__(1024, 4) void eloc_rbm_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, RbmLayout rl,
                                                          RbmBlocks B, uint32_t nchunks, uint32_t hw, const double *__restrict__ plan,
                                                          const double *__restrict__ rbm, double *__restrict__ eloc,
                                                          double *__restrict__ psi, double lambda, double *__restrict__ green,
                                                          uint8_t *__restrict__ clamped) {
  static_assert(!GREEN || FLAVOUR != kRbmPhase, "the fixed-node row needs a real-valued amplitude");
  // no static __shared__ here: with the dynamic region at LDS address 0 the row offsets below are the addresses and
  // the ds_read immediates carry the rest (a static in front costs one v_add per read)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *red = reinterpret_cast<double *>(smem + lds_bytes_rbm(p, rl, hw) - 144);  // [16]: one per wave
  uint32_t *next_tile_p = reinterpret_cast<uint32_t *>(red + 16);
  uint32_t *next_single_p = next_tile_p + 1;
  const uint64_t wg = blockIdx.x;
  const uint64_t walker = wg / nchunks;
  const uint32_t chunk = (uint32_t)(wg - walker * nchunks);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;  // 3 or 4 waves: whichever divides the walker's tiles better
  if (tid == 0) { *next_tile_p = 0; *next_single_p = 0; }
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  const int sorb = p.sorb, H = rl.H, Hq = rl.Hq;
  const uint32_t K = (uint32_t)sorb >> 1;
  RbmLds R;
  {
    // (plain offsets from the LDS array: a pointer that went through an integer cast is no longer known to be LDS
    // and its loads become flat_load with full waits)
    R.q = reinterpret_cast<double *>(smem + rbm_q_offset(p));
    R.mn = reinterpret_cast<double *>(smem + rbm_q_offset(p) + rbm_region_bytes(p, hw));
    R.sh = R.mn + 2 * Hq;
    R.Cq = R.sh + Hq;
    R.Aq = R.Cq + (sorb + 2);
    R.hs = R.Aq + (sorb + 2);
    R.rowaddr = reinterpret_cast<uint32_t *>(R.hs + (p.d1 + 2));
  }
#if defined(PYNQS_RBM_STOP) && PYNQS_RBM_STOP == 1
  if (tid == 0) eloc[walker] = (double)nocc;
  return;
#endif
  // ---- phase A, no barrier inside: three independent jobs on different waves -----------------------------------
  //   last wave  : <x|H|x> (ordered sum of nele(nele+1)/2 terms by one lane: the longest serial job)
  //   other waves: theta_h -> m_h, n_h, s_h (and ln psi(x)), then the singles' matrix elements in tiles of 16
  // The singles / diagonal are staged in wave-private quarters of the region q will occupy afterwards.
  const uint32_t tS = (B.b[0] + 63) / 64;
  const bool need_hs = chunk < max(tS, 1u);
  const double *__restrict__ Wt = rbm + rl.offWt;
  double lnpsi = 0.0;
  const int kThetaThreads = nthreads - 64;
  if (wave == nwaves - 1) {
    if (need_hs) {
      const double hii = fast_diag<double>(p, pl, L, plan);
      if (lane == 0) R.hs[0] = hii;
    }
  } else {
    for (int h = tid; h < Hq; h += kThetaThreads) {
      double m = 1.0, n = 0.0, s = 1.0;
      if (h < H) {
        double th = rbm[rl.offHb + h];
#pragma unroll 16
        for (int o = 0; o < sorb; ++o) {  // independent loads: 16 in flight per round trip (L2 latency ~0.6 us)
          const double w = Wt[(size_t)o * Hq + h];
          th += bit_of<LEN>(wk.w, o) ? w : -w;
        }
        const double a = fabs(th), rho = exp(-2.0 * a), lc = a + log1p(rho);  // lc = ln 2cosh(theta)
        m = 1.0 / (1.0 + rho);
        n = exp(-0.25 * (a + lc));  // (m rho)^(1/4): every excitation multiplies four rows (singles: two + dummy twice)
        s = th >= 0.0 ? 1.0 : -1.0;
        lnpsi += lc;
      }
      R.mn[h] = m; R.mn[Hq + h] = n; R.sh[h] = s;
    }
  }
  if (need_hs) {  // every wave, as it becomes free: 64 singles per pass
    const uint32_t nst = (p.d1 + 63) / 64;
    for (;;) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(next_single_p, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nst) break;
      if (t * 64 + lane < p.d1) R.hs[1 + t * 64 + lane] = fast_single<double>(t * 64 + lane, p, pl, L, nocc, plan);
    }
  }
  __syncthreads();
#if defined(PYNQS_RBM_STOP) && PYNQS_RBM_STOP == 2
  if (tid == 0) eloc[walker] = R.hs[0];
  return;
#endif
  // ---- phase B: q'[o][h] = (m_h rho_h)^(1/4) exp(4 s_h x_o W[h][o]) for the hidden units [h0, h0 + hw) into LDS, a wave
  // per row, and (with_sum) sum_h s_h W[h][o] -> Cq[o].  kRowBatch rows x 2 columns per lane are requested together:
  // an un-batched loop pays one L2 round trip per row.
  constexpr int kRowBatch = 4;
  const double *__restrict__ E4 = rbm + rl.offE4p;
  const uint32_t dE4 = (uint32_t)(rl.offE4m - rl.offE4p), uHq = (uint32_t)Hq, stride = hw + 1;
  auto build_window = [&](uint32_t h0, bool with_sum) {
    bool pos[2];   // s_h > 0 for this lane's two columns
    double fq[2];  // (m_h rho_h)^(1/4), 0 in the padding
    double sgn[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t j = lane + 64 * c, h = h0 + j;
      const bool in = j < hw && h < (uint32_t)H;
      sgn[c] = in ? R.sh[h] : 0.0;
      pos[c] = sgn[c] > 0.0;
      fq[c] = in ? R.mn[Hq + h] : 0.0;
    }
    for (int o0 = wave; o0 <= sorb; o0 += nwaves * kRowBatch) {
      double e4[kRowBatch][2], wv[kRowBatch][2];
#pragma unroll
      for (int b = 0; b < kRowBatch; ++b) {
        const int o = o0 + b * nwaves;
        const bool occ = o < sorb && bit_of<LEN>(wk.w, o);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint32_t j = lane + 64 * c, h = h0 + j, idx = (uint32_t)o * uHq + h;
          e4[b][c] = 1.0; wv[b][c] = 0.0;
          if (o < sorb && j < hw && h < (uint32_t)H) {
            e4[b][c] = E4[idx + (pos[c] == occ ? 0u : dE4)];
            if (with_sum) wv[b][c] = Wt[idx];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < kRowBatch; ++b) {
        const int o = o0 + b * nwaves;
        if (o > sorb) break;  // wave-uniform
        const bool occ = o < sorb && bit_of<LEN>(wk.w, o);
        const uint32_t rowq = (o < sorb ? rbm_row(o, K) : (uint32_t)sorb) * stride;
        double S = sgn[0] * wv[b][0] + sgn[1] * wv[b][1];
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (lane + 64 * c < stride) R.q[rowq + lane + 64 * c] = e4[b][c] * fq[c];
        for (uint32_t j = lane + 128; j < stride; j += 64) {  // windows wider than 128 hidden units: the rest of the row
          const uint32_t h = h0 + j;
          double v = 0.0;
          if (j < hw && h < (uint32_t)H) {
            const double s2 = R.sh[h];
            v = R.mn[Hq + h];
            if (o < sorb) {
              const uint32_t idx = (uint32_t)o * uHq + h;
              v *= E4[idx + ((s2 > 0.0) == occ ? 0u : dE4)];
              if (with_sum) S += s2 * Wt[idx];
            }
          }
          R.q[rowq + j] = v;
        }
        if (with_sum) {
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) S += __shfl_xor(S, d);
          if (lane == 0) R.Cq[o] = S;  // sum_h s_h W[h][o]; turned into C(o) below, all orbitals at once
        }
      }
    }
  };
  if constexpr (!WINDOWED) {
    build_window(0u, true);
  } else {
    // sum_h s_h W[h][o] over ALL hidden units, a wave per orbital (the windows are built inside the rounds below)
    for (int o = wave; o <= sorb; o += nwaves) {
      double S = 0.0;
      if (o < sorb)
        for (int h = lane; h < H; h += 64) S += R.sh[h] * Wt[(uint32_t)o * uHq + h];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) S += __shfl_xor(S, d);
      if (lane == 0) R.Cq[o] = S;
    }
  }
  __syncthreads();
  const uint32_t qbase = __builtin_amdgcn_groupstaticsize() + (uint32_t)rbm_q_offset(p), rowB = stride * 8u;
  for (int o = tid; o <= sorb; o += nthreads) {
    R.rowaddr[o] = qbase + (o < sorb ? rbm_row(o, K) : (uint32_t)sorb) * rowB;
    double c = 1.0, da = 1.0;
    if (o < sorb) {
      const double x = bit_of<LEN>(wk.w, o) ? 1.0 : -1.0, a = rbm[rl.offVb + o];
      if constexpr (FLAVOUR == kRbmTanh) {
        c = exp(-2.0 * x * R.Cq[o]);
        da = exp(-4.0 * x * a);
      } else {
        c = exp(-2.0 * x * (a + R.Cq[o]));
        lnpsi += x * a;
      }
    }
    R.Cq[o] = c;
    if constexpr (FLAVOUR == kRbmTanh) R.Aq[o] = da;
  }
  double ax = 0.0, e2ax = 1.0, inv_tanh_ax = 1.0;  // tanh flavour: a.x (wave-uniform: scalar loads), exp(2 a.x), 1 / tanh(a.x)
  if constexpr (FLAVOUR == kRbmTanh) {
    for (int o = 0; o < sorb; ++o) {
      const double a = rbm[rl.offVb + o];
      ax += bit_of<LEN>(wk.w, o) ? a : -a;
    }
    e2ax = exp(2.0 * ax);
    inv_tanh_ax = 1.0 / tanh(ax);
  }
  __syncthreads();
#if defined(PYNQS_RBM_STOP) && PYNQS_RBM_STOP == 3
  if (tid == 0) eloc[walker] = R.hs[0] + R.Cq[0] + R.q[5];
  return;
#endif

  // ---- tiles of 64 blocks: pulled by the waves from an LDS counter, or (WINDOWED) one per wave and round ------------
  const double *__restrict__ Vss = plan + pl.offVss;
  const double *__restrict__ Vab = plan + pl.offVab;
  const uint32_t my_tiles = B.ntiles > chunk ? (B.ntiles - chunk + nchunks - 1) / nchunks : 0;
  const uint32_t nrounds = (my_tiles + nwaves - 1) / nwaves;  // WINDOWED only
  double esum = 0.0, esum_im = 0.0;  // GREEN: esum_im collects the sign-flip potential
  double *__restrict__ grow = GREEN ? green + (size_t)walker * (p.nsd + 1) : nullptr;
  // one column's contribution: h = <x|H|x'>, t = the real flavour's psi(x')/psi(x) (for tanh: without the visible factor),
  // da = exp(2 (a.x' - a.x)), col = the reference's column of x' (GREEN)
  // (branch-free but for the GREEN store: a branch per column keeps the 16 divisions / logarithms of a block from overlapping --
  // the phase flavour ran 1.70 ms with `if (ok) add_column(...)`, 1.12 ms without)
  auto add_column = [&](bool ok, double h, double t, double da, uint32_t col) {
    if constexpr (FLAVOUR == kRbmReal || FLAVOUR == kRbmTanh) {
      double r = t;
      if constexpr (FLAVOUR == kRbmTanh) r = t * ((1.0 - 2.0 / fma(e2ax, da, 1.0)) * inv_tanh_ax);  // tanh(a.x'): exp(2 a.x') = inf -> 1, 0 -> -1
      const double hr = ok ? h * r : 0.0;
      esum += hr;
      if constexpr (GREEN) {
        const bool keeps_sign = (r < 0.0) != (h < 0.0);
        if (ok) grow[col] = keeps_sign ? -hr : 0.0;
        esum_im += keeps_sign ? 0.0 : hr;
      }
    } else {
      double sn, cs;
      sincos_moderate(log(ok ? t : 1.0), sn, cs);
      esum += (ok ? h : 0.0) * cs;
      esum_im += (ok ? h : 0.0) * sn;
    }
  };
  for (uint32_t round = 0;; ++round) {
    uint32_t lt = 0;
    if constexpr (WINDOWED) {
      if (round >= nrounds) break;       // workgroup-uniform: the windows below contain barriers
      lt = round * nwaves + wave;        // >= my_tiles: a wave without a tile still helps to build the windows
    } else {
      if (lane == 0) lt = atomicAdd(next_tile_p, 1u);
      lt = __builtin_amdgcn_readfirstlane(lt);
      if (lt >= my_tiles) break;
    }
    const uint32_t id = lt < my_tiles ? (chunk + lt * nchunks) * 64u + (uint32_t)lane : 0xffffffffu;
    // class and block of this lane
    int cls = 4;
    uint32_t bid = 0, nbf = 1, offF = 0, offS = 0, nF = 1, nS = 1;
    MagicDiv dv = B.dv[0];
    if (id < B.b[0]) { cls = 0; bid = id; nbf = B.nbf[0]; offF = p.offSa; nF = p.d1; }
    else if (id < B.b[1]) { cls = 1; bid = id - B.b[0]; nbf = B.nbf[1]; dv = B.dv[1]; offF = p.offHPa; offS = p.offPPa; nF = p.noAA; nS = p.nvAA; }
    else if (id < B.b[2]) { cls = 2; bid = id - B.b[1]; nbf = B.nbf[2]; dv = B.dv[2]; offF = p.offHPb; offS = p.offPPb; nF = p.noBB; nS = p.nvBB; }
    else if (id < B.b[3]) { cls = 3; bid = id - B.b[2]; nbf = B.nbf[3]; dv = B.dv[3]; offF = p.offSa; offS = p.offSb; nF = p.nSa; nS = p.nSb; }
    const uint32_t bs = mdiv(bid, dv), bf = bid - bs * nbf;
    const bool real_fast = cls < 4, real_slow = cls >= 1 && cls < 4;
    // the 4 + 4 table entries of this lane's block; re-read where needed rather than kept in registers over the
    // hidden-unit loop (the kernel sits at the 128-VGPR line)
    auto entries = [&](uint32_t (&ef)[4], uint32_t (&es)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ef[i] = real_fast ? L.tab[offF + min(4 * bf + i, nF - 1)] : 0u;
        es[i] = real_slow ? L.tab[offS + min(4 * bs + i, nS - 1)] : 0u;
      }
    };
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 1.0;
    // One hidden unit per sub-step: 16 ds_read_b64 (2 LDS cycles each, 256 B/clk) feed 8 + 32 f64 operations
    // (the rows carry (m_h rho_h)^(1/4), so the product of an excitation's four rows is n_h prod q).
    // Eight sub-steps share one update of the row addresses (immediate offsets); the empty asm keeps the compiler
    // from fusing the loads of neighbouring hidden units into ds_read2_b64 (half the LDS rate) or into 16-byte
    // loads (twice the registers: the kernel must stay below 128 VGPRs for 4 waves per SIMD).
    auto hidden_units = [&](uint32_t h0, uint32_t count) {
      // the q' rows of the 8 entries' orbitals as 32-bit LDS addresses (the dynamic region starts at the static
      // size; an array of generic pointers loses the address space and its loads become flat_load)
      uint32_t rb[16];
      {
        uint32_t ef[4], es[4];
        entries(ef, es);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rb[2 * i] = R.rowaddr[real_fast ? (ef[i] & 0xff) : (uint32_t)sorb];
          rb[2 * i + 1] = R.rowaddr[real_fast ? ((ef[i] >> 8) & 0xff) : (uint32_t)sorb];
          rb[8 + 2 * i] = R.rowaddr[real_slow ? (es[i] & 0xff) : (uint32_t)sorb];
          rb[8 + 2 * i + 1] = R.rowaddr[real_slow ? ((es[i] >> 8) & 0xff) : (uint32_t)sorb];
        }
      }
      for (uint32_t h = 0; h < count; h += 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double v[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<lds_cdouble *>(rb[k] + 8 * c);
          const double m = R.mn[h0 + h + c];
          double gf[4], gs[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            gf[i] = v[2 * i] * v[2 * i + 1];
            gs[i] = v[8 + 2 * i] * v[8 + 2 * i + 1];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * i + j] *= fma(gf[i], gs[j], m);
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) rb[k] += 64;
      }
    };
    if constexpr (WINDOWED) {
      for (uint32_t h0 = 0; h0 < (uint32_t)rl.Hloop; h0 += hw) {
        __syncthreads();  // the previous window has been consumed by every wave
        build_window(h0, false);
        __syncthreads();
        hidden_units(h0, min(hw, (uint32_t)rl.Hloop - h0));
      }
    } else {
      hidden_units(0u, (uint32_t)rl.Hloop);
    }
    // matrix elements, prefactors, sum
    if (cls < 4) {
      uint32_t ef[4], es[4];
      entries(ef, es);
      double cf[4], cs[4], af[4], as[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cf[i] = R.Cq[ef[i] & 0xff] * R.Cq[(ef[i] >> 8) & 0xff];
        cs[i] = cls == 0 ? 1.0 : R.Cq[es[i] & 0xff] * R.Cq[(es[i] >> 8) & 0xff];
        af[i] = as[i] = 1.0;
        if constexpr (FLAVOUR == kRbmTanh) {
          af[i] = R.Aq[ef[i] & 0xff] * R.Aq[(ef[i] >> 8) & 0xff];
          as[i] = cls == 0 ? 1.0 : R.Aq[es[i] & 0xff] * R.Aq[(es[i] >> 8) & 0xff];
        }
      }
      if (cls == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t f = 4 * bf + i;
          add_column(f < nF, R.hs[1 + min(f, nF - 1)], acc[4 * i] * cf[i], af[i], 1u + f);
        }
      } else {
        const bool opp = cls == 3;
        const double *__restrict__ V = opp ? Vab : Vss + (size_t)(cls - 1) * pl.NP * pl.NP;
        const uint32_t mul = opp ? (uint32_t)(pl.K * pl.K) : (uint32_t)pl.NP;
        const uint32_t mask = opp ? 0x7fffu : 0x1fffu;
        double hv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) hv[4 * i + j] = V[((es[j] >> 17) & mask) * mul + ((ef[i] >> 17) & mask)];
        // GREEN: column of (fast f, slow s) = 1 + class base + s * nF + u, u = f - rot (mod nF) for the same-spin classes
        const uint32_t cbase = 1u + (cls == 1 ? p.d1 : (cls == 2 ? p.d2 : p.d3));
        const uint32_t rot = cls == 1 ? (uint32_t)p.rotA : (cls == 2 ? (uint32_t)p.rotB : 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a0 = ef[i] & 0xff, a1 = (ef[i] >> 8) & 0xff;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b0 = es[j] & 0xff, b1 = (es[j] >> 8) & 0xff;
            uint32_t par = ((ef[i] ^ es[j]) >> 16) & 1u;  // as plan_dev.h: finish_double
            if (opp) par ^= (uint32_t)(a0 < b1) ^ (uint32_t)(b0 < a1) ^ 1u;
            else par ^= (uint32_t)(a0 < b0) ^ (uint32_t)(a1 < b0) ^ (uint32_t)(a0 < b1) ^ (uint32_t)(a1 < b1);
            const bool ok = 4 * bf + i < nF && 4 * bs + j < nS;
            const double t = (acc[4 * i + j] * cf[i]) * cs[j];
            uint32_t col = 0;
            if constexpr (GREEN) {
              const uint32_t f = 4 * bf + i;
              col = cbase + (4 * bs + j) * nF + (f >= rot ? f - rot : f + nF - rot);
            }
            if constexpr (FLAVOUR == kRbmPhase) {
              // branch-free first pass: the phase ln t and the signed matrix element in place (0 for the padding columns of the block)
              acc[4 * i + j] = log(ok ? t : 1.0);
              hv[4 * i + j] = ok ? (par ? -hv[4 * i + j] : hv[4 * i + j]) : 0.0;
            } else {
              add_column(ok, par ? -hv[4 * i + j] : hv[4 * i + j], t, af[i] * as[j], col);
            }
          }
        }
        if constexpr (FLAVOUR == kRbmPhase) {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            double sn, cs;
            sincos_moderate(acc[k], sn, cs);
            esum += hv[k] * cs;
            esum_im += hv[k] * sn;
          }
        }
      }
    }
  }
  if (chunk == 0 && tid == 0) esum += R.hs[0];  // x' = x
  // fixed-order reductions: lanes, then waves.  kRbmPhase: eloc / psi hold (re, im) pairs
  constexpr int kOut = FLAVOUR == kRbmPhase ? 2 : 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    esum += __shfl_xor(esum, o);
    lnpsi += __shfl_xor(lnpsi, o);
    if constexpr (FLAVOUR == kRbmPhase || GREEN) esum_im += __shfl_xor(esum_im, o);
  }
  auto over_waves = [&](double v) {  // workgroup-uniform calls; the sum is valid in thread 0
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0)
      for (int w = 0; w < nwaves; ++w) s += red[w];
    return s;
  };
  const double e_re = over_waves(esum);
  if (tid == 0) {
    if (nchunks == 1) eloc[kOut * walker] = e_re;
    else atomicAdd(eloc + kOut * walker, e_re);
  }
  if constexpr (GREEN) {  // (launched with one workgroup per walker)
    const double v_sf = over_waves(esum_im);
    if (tid == 0) {
      const double k0 = lambda - (R.hs[0] + v_sf);
      grow[0] = k0 < 0.0 ? 0.0 : k0;
      clamped[walker] = k0 < 0.0 ? 1 : 0;
    }
  }
  if constexpr (FLAVOUR == kRbmPhase) {
    const double e_im = over_waves(esum_im);
    if (tid == 0) {
      if (nchunks == 1) eloc[2 * walker + 1] = e_im;
      else atomicAdd(eloc + 2 * walker + 1, e_im);
    }
  }
  if (psi != nullptr && chunk == 0) {  // workgroup-uniform
    const double s = over_waves(lnpsi);
    if (tid == 0) {
      if constexpr (FLAVOUR == kRbmReal) psi[walker] = exp(s);
      else if constexpr (FLAVOUR == kRbmTanh) psi[walker] = tanh(ax) * exp(s);
      else sincos(s, &psi[2 * walker + 1], &psi[2 * walker]);
    }
  }
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

static constexpr size_t kRbmMaxLds = 158 * 1024;  // of the CU's 160 KiB

// Shape of the windowed kernel (sorb x num_hidden does not fit in LDS), by the number of tiles a walker's row has:
// many tiles -> ONE workgroup of 1024 threads per CU around a 136 KiB window (16 waves share it, half as many windows
// and barriers: sorb 120, 240 hidden units, 1024 walkers: 70.3 ms with 256 threads and 64 KiB, 41.6 with 512 and 64 KiB,
// 32.6 with 1024 and 136 KiB); fewer tiles than waves would idle -> 512 or 256 threads, two workgroups per CU around
// 64 KiB windows.  PYNQS_RBM_BLOCK / PYNQS_RBM_WINDOW_KB override.
struct RbmShape {
  uint32_t threads;
  size_t window_lds;
};
static RbmShape rbm_windowed_shape(uint32_t ntiles) {
  static const int blk_env = getenv("PYNQS_RBM_BLOCK") ? atoi(getenv("PYNQS_RBM_BLOCK")) : 0;
  static const int win_env = getenv("PYNQS_RBM_WINDOW_KB") ? atoi(getenv("PYNQS_RBM_WINDOW_KB")) : 0;
  RbmShape s;
  s.threads = (blk_env == 256 || blk_env == 512 || blk_env == 1024) ? (uint32_t)blk_env : (ntiles >= 64 ? 1024u : (ntiles >= 24 ? 512u : 256u));
  s.window_lds = (size_t)(win_env > 0 ? win_env : (s.threads == 1024 ? 136 : 64)) * 1024;
  if (s.window_lds > kRbmMaxLds) s.window_lds = kRbmMaxLds;
  return s;
}

// hidden units resident in LDS: all of them (rl.Hloop) when that fits, else the largest multiple of 8 that keeps the
// workgroup within window_lds (at least 8); 0 if not even that fits
static uint32_t rbm_window(const SDParams &p, const RbmLayout &rl, size_t window_lds) {
  if (lds_bytes_rbm(p, rl, (uint32_t)rl.Hloop) <= kRbmMaxLds) return (uint32_t)rl.Hloop;
  const size_t other = lds_bytes_rbm(p, rl, 0u) - rbm_region_bytes(p, 0u);
  for (uint32_t hw = 256; hw >= 8; hw -= 8)
    if (other + rbm_region_bytes(p, hw) <= (hw > 8 ? window_lds : kRbmMaxLds)) return hw;
  return 0;
}

extern "C" int64_t pynqs_rbm_table_bytes(int sorb, int nhidden) {
  RbmLayout rl;
  if (!make_rbm_layout(sorb, nhidden, &rl)) return -1;
  return rl.total * 8;
}

extern "C" int pynqs_eloc_rbm_supported(int sorb, int nele, int noA, int noB, int nhidden) {
  SDParams p;
  PlanLayout pl;
  RbmLayout rl;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl) || !make_rbm_layout(sorb, nhidden, &rl)) return 0;
  return rbm_window(p, rl, kRbmMaxLds) > 0 ? 1 : 0;
}

extern "C" int pynqs_rbm_table_build(const double *weights, const double *hidden_bias, const double *visible_bias, int sorb,
                                     int nhidden, void *table, void *stream) {
  pynqs::DeviceScope device_scope_(weights);
  RbmLayout rl;
  if (!make_rbm_layout(sorb, nhidden, &rl)) return set_error(PYNQS_EINVAL, "bad sorb / nhidden");
  if (!weights || !hidden_bias || !table) return set_error(PYNQS_EINVAL, "null pointer");
  const int64_t n = (int64_t)rl.sorb * rl.Hq;
  const uint32_t grid = (uint32_t)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(rbm_table_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, weights, hidden_bias, visible_bias, rl,
                     (double *)table);
  return check_launch("rbm_table_build");
}

static int eloc_rbm_impl(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                         const void *rbm_table, int nhidden, int flavour, double *eloc, double *psi, double lambda, double *green,
                         uint8_t *clamped, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  if (flavour < PYNQS_RBM_REAL || flavour > PYNQS_RBM_PHASE) return set_error(PYNQS_EINVAL, "unknown RBM flavour");
  if (green && (flavour == PYNQS_RBM_PHASE || !clamped)) return set_error(PYNQS_EINVAL, "the Green's-function row needs a real-valued flavour");
  SDParams p;
  PlanLayout pl;
  RbmLayout rl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (!make_rbm_layout(sorb, nhidden, &rl)) return set_error(PYNQS_EINVAL, "bad nhidden");
  if (nbatch < 0 || nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "bad nbatch");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !rbm_table || !eloc) return set_error(PYNQS_EINVAL, "null pointer");
  const RbmBlocks B = make_rbm_blocks(p);
  // few walkers: cut a walker's tiles over several workgroups (each repeats the per-walker set-up)
  uint32_t nchunks = 1;
  if (nbatch < 1024 && !green) {  // (the Green's-function row finishes its diagonal in the kernel: one workgroup per walker)
    nchunks = (uint32_t)((1024 + nbatch - 1) / nbatch);
    const uint32_t maxc = B.ntiles / 4 > 0 ? B.ntiles / 4 : 1;
    if (nchunks > maxc) nchunks = maxc;
  }
  const RbmShape shape = rbm_windowed_shape(B.ntiles / nchunks);
  const uint32_t hw = rbm_window(p, rl, shape.window_lds);
  if (hw == 0) return set_error(PYNQS_EINVAL, "the walker tables of this system leave no LDS for the RBM rows");
  const bool windowed = hw < (uint32_t)rl.Hloop;
  const size_t lds = lds_bytes_rbm(p, rl, hw);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, (flavour == PYNQS_RBM_PHASE ? 16 : 8) * (size_t)nbatch, st) != hipSuccess)
    return check_launch("memset");
  const int len = (sorb - 1) / 64 + 1;
  // (3-wave workgroups divide Fe2S2's 9 tiles evenly but leave only 12 waves per CU -- LDS allows 4 workgroups --
  // and were 8 % slower at 80 hidden units; the kernel itself runs with any multiple of 64 threads >= 128)
  // resident q': the workgroup size that puts the most waves on a CU (the registers allow 16; a workgroup's waves share
  // its LDS), as long as the walker has at least two tiles per wave.  Fe2S2 (31 KiB): 256 threads, 4 workgroups per CU;
  // sorb 56 with 112 hidden units (58 KiB): two workgroups per CU -> 512 threads.
  uint32_t threads = shape.threads;
  if (!windowed) {
    static const int blk_env = getenv("PYNQS_RBM_BLOCK") ? atoi(getenv("PYNQS_RBM_BLOCK")) : 0;
    threads = kBlock;
    size_t best = 0;
    for (uint32_t b = kBlock; b <= 1024; b *= 2) {
      size_t waves = (160 * 1024 / (lds + 256)) * (b / 64);
      if (waves > 16) waves = 16;
      if (b > kBlock && B.ntiles / nchunks < 2 * (b / 64)) break;
      if (waves > best) { best = waves; threads = b; }
    }
    if (blk_env == 256 || blk_env == 512 || blk_env == 1024) threads = (uint32_t)blk_env;
  }
#define PYNQS_RBM_LAUNCH(W, F, G)                                                                                                     \
  do {                                                                                                                                \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&eloc_rbm_kernel<LEN, W, F, G>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                  \
      return check_launch("hipFuncSetAttribute");                                                                                     \
    hipLaunchKernelGGL((eloc_rbm_kernel<LEN, W, F, G>), dim3((uint32_t)grid), dim3(threads), lds, st, bra, p, pl, rl, B, nchunks, hw, \
                       (const double *)plan, (const double *)rbm_table, eloc, psi, lambda, green, clamped);                          \
  } while (0)
#define PYNQS_RBM_FLAVOURS(W)                                                \
  do {                                                                       \
    if (green && flavour == PYNQS_RBM_REAL) PYNQS_RBM_LAUNCH(W, kRbmReal, true);  \
    else if (green) PYNQS_RBM_LAUNCH(W, kRbmTanh, true);                     \
    else if (flavour == PYNQS_RBM_REAL) PYNQS_RBM_LAUNCH(W, kRbmReal, false); \
    else if (flavour == PYNQS_RBM_TANH) PYNQS_RBM_LAUNCH(W, kRbmTanh, false); \
    else PYNQS_RBM_LAUNCH(W, kRbmPhase, false);                              \
  } while (0)
  DISPATCH_LEN(len, {
    if (windowed) PYNQS_RBM_FLAVOURS(true);
    else PYNQS_RBM_FLAVOURS(false);
  });
#undef PYNQS_RBM_FLAVOURS
#undef PYNQS_RBM_LAUNCH
  return check_launch("eloc_rbm");
}

extern "C" int pynqs_eloc_rbm_flavour(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                      const void *rbm_table, int nhidden, int flavour, double *eloc, double *psi, void *stream) {
  return eloc_rbm_impl(bra, nbatch, sorb, nele, noA, noB, plan, rbm_table, nhidden, flavour, eloc, psi, 0.0, nullptr, nullptr, stream);
}

extern "C" int pynqs_eloc_rbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                              const void *rbm_table, int nhidden, double *eloc, double *psi, void *stream) {
  return eloc_rbm_impl(bra, nbatch, sorb, nele, noA, noB, plan, rbm_table, nhidden, PYNQS_RBM_REAL, eloc, psi, 0.0, nullptr, nullptr, stream);
}

extern "C" int pynqs_green_rbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                               const void *rbm_table, int nhidden, int flavour, double lambda, double *eloc, double *psi, double *green,
                               uint8_t *clamped, void *stream) {
  if (!green || !clamped) return set_error(PYNQS_EINVAL, "null pointer");
  return eloc_rbm_impl(bra, nbatch, sorb, nele, noA, noB, plan, rbm_table, nhidden, flavour, eloc, psi, lambda, green, clamped, stream);
}
