// kernels_rbm_complex.hip -- SIMPLE local energy (vmc/energy/eloc.py:121-203) with the amplitude ratio of an RBM with COMPLEX
// parameters evaluated on chip:
//   psi(x) = exp(a.x) prod_h 2 cosh(theta_h(x)),  theta_h = b_h + sum_o W[h][o] x_o,   a, b, W complex, x_o = +-1
// (vmc/ansatz/rbm/rbm.py:199-211, rbm_type "complex"; rbm_type "cos", prod_h cos(theta_h) with real parameters, is the same
// function of i*W, i*b up to the constant 2^H: cos t = cosh(i t)).
// Same algebra as kernels_rbm.hip with complex numbers: an excitation flips the orbitals F, theta' = theta - delta,
// delta_h = 2 sum_{o in F} W[h][o] x_o; with s_h = sign(Re theta_h), rho_h = exp(-2 s_h theta_h) (|rho| <= 1), m_h = 1/(1 + rho_h):
//   cosh(theta_h - delta_h)/cosh(theta_h) = exp(-s_h delta_h) (m_h + m_h rho_h prod_{o in F} q_h(o)),  q_h(o) = exp(4 s_h W[h][o] x_o),
//   psi(x')/psi(x) = prod_{o in F} C(o) * prod_h (m_h + prod_{o in F} q'_h(o)),   q'_h(o) = (m_h rho_h)^(1/4) q_h(o),
//   C(o) = exp(-2 x_o (a_o + sum_h s_h W[h][o]))          (any branch of the fourth root: four rows are multiplied).
// Per walker the workgroup builds q'[o][h] (complex, 16 B) in LDS; a lane owns a 2 x 4 block of excitations (2 entries of a
// class's fast table x 4 of its slow table) with 8 complex running products:
//   per hidden unit and lane: 12 ds_read_b128, 6 complex products for the pairs, 8 x (complex fma + complex product)
//   = 88 f64 instructions per 8 columns (the real kernel: 40 per 16).
// When sorb x num_hidden rows do not fit the LDS the kernel runs WINDOWED (round 3), like the real-parameter kernel: the workgroup
// streams q' through the LDS `hw` hidden units at a time, and in every round each wave keeps the 8 x 64 running products of ONE tile in
// registers across the windows (two barriers per window; the rows of a window are rebuilt every round: ~1 % of a round's work).
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"

namespace pynqs {

typedef double cplx __attribute__((ext_vector_type(2)));  // (re, im)
typedef __attribute__((address_space(3))) const cplx lds_ccplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{fma(-a.y, b.y, a.x * b.x), fma(a.x, b.y, a.y * b.x)}; }
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) { return cplx{fma(-a.y, b.y, fma(a.x, b.x, c.x)), fma(a.y, b.x, fma(a.x, b.y, c.y))}; }
__device__ __forceinline__ cplx cexp(cplx z) {
  double sn, cs;
  sincos(z.y, &sn, &cs);
  const double e = exp(z.x);
  return cplx{e * cs, e * sn};
}

// Table in caller-owned memory, complex double = 2 doubles, hidden index fastest, row stride Hs (odd: consecutive rows start in
// different 16-byte bank groups):  Wt [sorb][Hs] | E4p = exp(+4W) [sorb][Hs] | E4m = exp(-4W) [sorb][Hs] | hb [Hs] | vb [sorb]
struct CrbmLayout {
  int sorb, H, Hloop, Hs;  // Hloop = H rounded up to 2 (the hidden-unit loop), Hs = Hloop + 1
  int64_t offWt, offE4p, offE4m, offHb, offVb, total;  // in complex elements
};

static inline bool make_crbm_layout(int sorb, int H, CrbmLayout *L) {
  if (sorb < 1 || sorb > 192 || H < 1 || H > 4096) return false;
  L->sorb = sorb; L->H = H;
  L->Hloop = (H + 1) & ~1;
  L->Hs = L->Hloop + 1;
  const int64_t row = (int64_t)sorb * L->Hs;
  L->offWt = 0; L->offE4p = row; L->offE4m = 2 * row; L->offHb = 3 * row;
  L->offVb = L->offHb + L->Hs;
  L->total = L->offVb + sorb;
  return true;
}

// W[H][sorb][2], hb[H][2], vb[sorb][2] (the reference's params_weights / params_hidden_bias / params_visible_bias) -> table
__global__ __launch_bounds__(kBlock) void crbm_table_kernel(const cplx *__restrict__ W, const cplx *__restrict__ hb, const cplx *__restrict__ vb,
                                                            CrbmLayout cl, cplx *__restrict__ tab) {
  const int64_t row = (int64_t)cl.sorb * cl.Hs;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < row) {
    const int o = (int)(i / cl.Hs), h = (int)(i - (int64_t)o * cl.Hs);
    const cplx w = h < cl.H ? W[(int64_t)h * cl.sorb + o] : cplx{0.0, 0.0};
    tab[cl.offWt + i] = w;
    tab[cl.offE4p + i] = cexp(4.0 * w);
    tab[cl.offE4m + i] = cexp(-4.0 * w);
  }
  if (i < cl.Hs) tab[cl.offHb + i] = i < cl.H ? hb[i] : cplx{0.0, 0.0};
  if (i < cl.sorb) tab[cl.offVb + i] = vb ? vb[i] : cplx{0.0, 0.0};
}

// 2 x 4 blocks: class k (0 singles x nothing, 1 alpha-alpha, 2 beta-beta, 3 alpha-beta) has nbf[k] x nbs blocks
struct CrbmBlocks {
  uint32_t nbf[4];
  uint32_t b[4];  // cumulative block counts
  uint32_t ntiles;
  MagicDiv dv[4];
};

static inline CrbmBlocks make_crbm_blocks(const SDParams &p) {
  CrbmBlocks B;
  const uint32_t nf[4] = {p.d1, (uint32_t)p.noAA, (uint32_t)p.noBB, (uint32_t)p.nSa};
  const uint32_t ns[4] = {p.d1 ? 1u : 0u, (uint32_t)p.nvAA, (uint32_t)p.nvBB, (uint32_t)p.nSb};
  uint32_t acc = 0;
  for (int k = 0; k < 4; ++k) {
    B.nbf[k] = (nf[k] + 1) / 2;
    B.dv[k] = make_magic(B.nbf[k]);
    acc += B.nbf[k] * ((ns[k] + 3) / 4);
    B.b[k] = acc;
  }
  B.ntiles = (acc + 63) / 64;
  return B;
}

// LDS after the walker tables (16-byte aligned): q [sorb + 1][Hs] cplx | m [Hs] cplx | n4 [Hs] cplx | Cq [sorb + 2] cplx |
// sh [Hs] double | hs [d1 + 2] double | rowaddr [sorb + 2] u32 | red [2 * 16] double, counters
__host__ __device__ inline size_t crbm_q_offset(const SDParams &p) { return (lds_fixed_bytes(p) + 15) & ~(size_t)15; }
// `hw`: hidden units of q' resident at a time (even): cl.Hloop (all of them) or the window of the WINDOWED kernel; row stride hw + 1
__host__ __device__ inline size_t lds_bytes_crbm(const SDParams &p, const CrbmLayout &cl, uint32_t hw) {
  return crbm_q_offset(p) + 16 * ((size_t)(p.sorb + 1) * (hw + 1) + 2 * (size_t)cl.Hs + (size_t)(p.sorb + 2)) +
         8 * ((size_t)cl.Hs + (size_t)(p.d1 + 2) + 1) + 4 * (((size_t)p.sorb + 2 + 3) & ~(size_t)3) + 8 * 32 + 16;
}

template <int LEN, bool WINDOWED>
__global__ __launch_bounds__(512) void eloc_crbm_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, CrbmLayout cl, CrbmBlocks B,
                                                        uint32_t nchunks, uint32_t hw, const double *__restrict__ plan,
                                                        const cplx *__restrict__ rbm, double log_scale, double *__restrict__ eloc,
                                                        double *__restrict__ psi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t wg = blockIdx.x;
  const uint64_t walker = wg / nchunks;
  const uint32_t chunk = (uint32_t)(wg - walker * nchunks);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;
  const int sorb = p.sorb, H = cl.H, Hs = cl.Hs;
  const uint32_t K = (uint32_t)sorb >> 1;
  const uint32_t stride = hw + 1;  // complex elements per q' row
  cplx *q = reinterpret_cast<cplx *>(smem + crbm_q_offset(p));
  cplx *mm = q + (size_t)(sorb + 1) * stride;
  cplx *n4 = mm + Hs;
  cplx *Cq = n4 + Hs;
  double *sh = reinterpret_cast<double *>(Cq + (sorb + 2));
  double *hs = sh + Hs;
  uint32_t *rowaddr = reinterpret_cast<uint32_t *>(hs + (p.d1 + 2) + 1);
  double *red = reinterpret_cast<double *>(smem + lds_bytes_crbm(p, cl, hw) - (8 * 32 + 16));
  uint32_t *next_tile_p = reinterpret_cast<uint32_t *>(red + 32);
  uint32_t *next_single_p = next_tile_p + 1;
  if (tid == 0) { *next_tile_p = 0; *next_single_p = 0; }
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);

  // ---- phase A (no barrier inside): last wave <x|H|x>; the others theta_h -> m_h, (m_h rho_h)^(1/4), s_h, ln 2cosh(theta_h); then
  // every wave the singles' matrix elements
  const uint32_t tS = (B.b[0] + 63) / 64;
  const bool need_hs = chunk < max(tS, 1u);
  const cplx *__restrict__ Wt = rbm + cl.offWt;
  cplx lnpsi = {0.0, 0.0};
  const int kThetaThreads = nthreads - 64;
  if (wave == nwaves - 1) {
    if (need_hs) {
      const double hii = fast_diag<double>(p, pl, L, plan);
      if (lane == 0) hs[0] = hii;
    }
  } else {
    for (int h = tid; h < Hs; h += kThetaThreads) {
      cplx m = {1.0, 0.0}, nq = {0.0, 0.0};
      double s = 1.0;
      if (h < H) {
        cplx th = rbm[cl.offHb + h];
#pragma unroll 8
        for (int o = 0; o < sorb; ++o) {
          const cplx w = Wt[(size_t)o * Hs + h];
          th += bit_of<LEN>(wk.w, o) ? w : -w;
        }
        s = th.x >= 0.0 ? 1.0 : -1.0;
        const cplx st = s * th;                       // Re >= 0
        const cplx rho = cexp(-2.0 * st);             // |rho| <= 1
        const double pr = 1.0 + rho.x, pi = rho.y, d = pr * pr + pi * pi;
        m = cplx{pr / d, -pi / d};                    // 1 / (1 + rho)
        const cplx lc = st + cplx{0.5 * log(d), atan2(pi, pr)};  // ln 2cosh(theta) = s theta + ln(1 + rho)
        nq = cexp(-0.25 * (st + lc));                 // (m rho)^(1/4): its fourth power is exp(-2 s theta - ln(1 + rho)) on every branch
        lnpsi += lc;
      }
      mm[h] = m; n4[h] = nq; sh[h] = s;
    }
  }
  if (need_hs) {
    const uint32_t nst = (p.d1 + 63) / 64;
    for (;;) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(next_single_p, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nst) break;
      if (t * 64 + lane < p.d1) hs[1 + t * 64 + lane] = fast_single<double>(t * 64 + lane, p, pl, L, nocc, plan);
    }
  }
  __syncthreads();
  // ---- phase B: q'[o][h] = (m_h rho_h)^(1/4) exp(4 s_h x_o W[h][o]) for the hidden units [h0, h0 + hw), a wave per row, and (with_sum:
  // the window is all of them) sum_h s_h W[h][o]
  const cplx *__restrict__ E4 = rbm + cl.offE4p;
  const uint32_t dE4 = (uint32_t)(cl.offE4m - cl.offE4p);
  auto rbm_row = [&](uint32_t o) { return o < (uint32_t)sorb ? (o >> 1) + ((o & 1u) ? K : 0u) : (uint32_t)sorb; };
  auto build_window = [&](uint32_t h0, bool with_sum) {
    for (int o = wave; o <= sorb; o += nwaves) {
      const bool occ = o < sorb && bit_of<LEN>(wk.w, o);
      cplx S = {0.0, 0.0};
      for (uint32_t j = lane; j < stride; j += 64) {
        const uint32_t h = h0 + j;
        cplx v = {0.0, 0.0};
        if (j < hw && h < (uint32_t)H) {
          const double s = sh[h];
          v = n4[h];
          if (o < sorb) {
            const uint32_t idx = (uint32_t)o * (uint32_t)Hs + h;
            v = cmul(v, E4[idx + ((s > 0.0) == occ ? 0u : dE4)]);
            if (with_sum) S += s * Wt[idx];
          }
        }
        q[(size_t)rbm_row((uint32_t)o) * stride + j] = v;
      }
      if (with_sum) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { S.x += __shfl_xor(S.x, d); S.y += __shfl_xor(S.y, d); }
        if (lane == 0) Cq[o] = S;
      }
    }
  };
  if constexpr (!WINDOWED) {
    build_window(0u, true);
  } else {
    for (int o = wave; o <= sorb; o += nwaves) {  // sum_h s_h W[h][o] over ALL hidden units (the windows are built inside the rounds)
      cplx S = {0.0, 0.0};
      if (o < sorb)
        for (int h = lane; h < H; h += 64) S += sh[h] * Wt[(uint32_t)o * (uint32_t)Hs + (uint32_t)h];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { S.x += __shfl_xor(S.x, d); S.y += __shfl_xor(S.y, d); }
      if (lane == 0) Cq[o] = S;
    }
  }
  __syncthreads();
  const uint32_t qbase = __builtin_amdgcn_groupstaticsize() + (uint32_t)crbm_q_offset(p), rowB = stride * 16u;
  for (int o = tid; o <= sorb; o += nthreads) {
    rowaddr[o] = qbase + rbm_row((uint32_t)o) * rowB;
    cplx c = {1.0, 0.0};
    if (o < sorb) {
      const double x = bit_of<LEN>(wk.w, o) ? 1.0 : -1.0;
      const cplx a = rbm[cl.offVb + o];
      c = cexp((-2.0 * x) * (a + Cq[o]));
      lnpsi += x * a;
    }
    Cq[o] = c;
  }
  __syncthreads();

  // ---- tiles of 64 blocks, pulled by the waves from an LDS counter
  const double *__restrict__ Vss = plan + pl.offVss;
  const double *__restrict__ Vab = plan + pl.offVab;
  const uint32_t my_tiles = B.ntiles > chunk ? (B.ntiles - chunk + nchunks - 1) / nchunks : 0;
  cplx esum = {0.0, 0.0};
  for (uint32_t round = 0;; ++round) {
    uint32_t lt = 0;
    bool active = true;  // (WINDOWED: wave-uniform; a wave without a tile still meets the round's barriers)
    if constexpr (WINDOWED) {
      if (round * (uint32_t)nwaves >= my_tiles) break;  // workgroup-uniform
      lt = round * (uint32_t)nwaves + (uint32_t)wave;
      active = lt < my_tiles;
    } else {
      if (lane == 0) lt = atomicAdd(next_tile_p, 1u);
      lt = __builtin_amdgcn_readfirstlane(lt);
      if (lt >= my_tiles) break;
    }
    const uint32_t id = active ? (chunk + lt * nchunks) * 64u + (uint32_t)lane : 0xffffffffu;
    int cls = 4;
    uint32_t bid = 0, nbf = 1, offF = 0, offS = 0, nF = 1, nS = 1;
    MagicDiv dv = B.dv[0];
    if (id < B.b[0]) { cls = 0; bid = id; nbf = B.nbf[0]; offF = p.offSa; nF = p.d1; }
    else if (id < B.b[1]) { cls = 1; bid = id - B.b[0]; nbf = B.nbf[1]; dv = B.dv[1]; offF = p.offHPa; offS = p.offPPa; nF = p.noAA; nS = p.nvAA; }
    else if (id < B.b[2]) { cls = 2; bid = id - B.b[1]; nbf = B.nbf[2]; dv = B.dv[2]; offF = p.offHPb; offS = p.offPPb; nF = p.noBB; nS = p.nvBB; }
    else if (id < B.b[3]) { cls = 3; bid = id - B.b[2]; nbf = B.nbf[3]; dv = B.dv[3]; offF = p.offSa; offS = p.offSb; nF = p.nSa; nS = p.nSb; }
    const uint32_t bs = mdiv(bid, dv), bf = bid - bs * nbf;
    const bool real_fast = cls < 4, real_slow = cls >= 1 && cls < 4;
    uint32_t ef[2], es[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) ef[i] = real_fast ? L.tab[offF + min(2 * bf + i, nF - 1)] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) es[j] = real_slow ? L.tab[offS + min(4 * bs + j, nS - 1)] : 0u;
    uint32_t rb0[12];  // LDS addresses of the q' rows: fast entry i -> rb[2i], rb[2i+1]; slow entry j -> rb[4+2j], rb[5+2j]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rb0[2 * i] = rowaddr[real_fast ? (ef[i] & 0xff) : (uint32_t)sorb];
      rb0[2 * i + 1] = rowaddr[real_fast ? ((ef[i] >> 8) & 0xff) : (uint32_t)sorb];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb0[4 + 2 * j] = rowaddr[real_slow ? (es[j] & 0xff) : (uint32_t)sorb];
      rb0[5 + 2 * j] = rowaddr[real_slow ? ((es[j] >> 8) & 0xff) : (uint32_t)sorb];
    }
    cplx acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = cplx{1.0, 0.0};
    for (uint32_t h0 = 0; h0 < (uint32_t)cl.Hloop; h0 += hw) {
      if constexpr (WINDOWED) {
        __syncthreads();  // everybody is done with the previous window
        build_window(h0, false);
        __syncthreads();
      }
      if (active) {
        uint32_t rb[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) rb[k] = rb0[k];
        const uint32_t wlen = min(hw, (uint32_t)cl.Hloop - h0);
        for (uint32_t j = 0; j < wlen; j += 2) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            cplx v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = *reinterpret_cast<lds_ccplx *>(rb[k] + 16 * c);
            const cplx m = mm[h0 + j + c];
            cplx gf[2], gs[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) gf[i] = cmul(v[2 * i], v[2 * i + 1]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) gs[jj] = cmul(v[4 + 2 * jj], v[5 + 2 * jj]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) acc[4 * i + jj] = cmul(acc[4 * i + jj], cfma(gf[i], gs[jj], m));
          }
#pragma unroll
          for (int k = 0; k < 12; ++k) rb[k] += 32;
        }
      }
    }
    if (cls < 4) {
      cplx cf[2], cs[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) cf[i] = cmul(Cq[ef[i] & 0xff], Cq[(ef[i] >> 8) & 0xff]);
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] = cls == 0 ? cplx{1.0, 0.0} : cmul(Cq[es[j] & 0xff], Cq[(es[j] >> 8) & 0xff]);
      if (cls == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t f = 2 * bf + i;
          if (f < nF) esum += hs[1 + f] * cmul(acc[4 * i], cf[i]);
        }
      } else {
        const bool opp = cls == 3;
        const double *__restrict__ V = opp ? Vab : Vss + (size_t)(cls - 1) * pl.NP * pl.NP;
        const uint32_t mul = opp ? (uint32_t)(pl.K * pl.K) : (uint32_t)pl.NP;
        const uint32_t mask = opp ? 0x7fffu : 0x1fffu;
        double hv[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) hv[4 * i + j] = V[((es[j] >> 17) & mask) * mul + ((ef[i] >> 17) & mask)];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int a0 = ef[i] & 0xff, a1 = (ef[i] >> 8) & 0xff;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b0 = es[j] & 0xff, b1 = (es[j] >> 8) & 0xff;
            uint32_t par = ((ef[i] ^ es[j]) >> 16) & 1u;  // as plan_dev.h: finish_double
            if (opp) par ^= (uint32_t)(a0 < b1) ^ (uint32_t)(b0 < a1) ^ 1u;
            else par ^= (uint32_t)(a0 < b0) ^ (uint32_t)(a1 < b0) ^ (uint32_t)(a0 < b1) ^ (uint32_t)(a1 < b1);
            const bool ok = 2 * bf + i < nF && 4 * bs + j < nS;
            const cplx t = cmul(cmul(acc[4 * i + j], cf[i]), cs[j]);
            if (ok) esum += (par ? -hv[4 * i + j] : hv[4 * i + j]) * t;
          }
        }
      }
    }
  }
  if (chunk == 0 && tid == 0) esum.x += hs[0];  // x' = x
  // fixed-order reductions: lanes, then waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    esum.x += __shfl_xor(esum.x, o); esum.y += __shfl_xor(esum.y, o);
    lnpsi.x += __shfl_xor(lnpsi.x, o); lnpsi.y += __shfl_xor(lnpsi.y, o);
  }
  auto over_waves = [&](cplx v) {  // workgroup-uniform calls; valid in thread 0
    __syncthreads();
    if (lane == 0) { red[2 * wave] = v.x; red[2 * wave + 1] = v.y; }
    __syncthreads();
    cplx s = {0.0, 0.0};
    if (tid == 0)
      for (int w = 0; w < nwaves; ++w) { s.x += red[2 * w]; s.y += red[2 * w + 1]; }
    return s;
  };
  const cplx e = over_waves(esum);
  if (tid == 0) {
    if (nchunks == 1) { eloc[2 * walker] = e.x; eloc[2 * walker + 1] = e.y; }
    else { atomicAdd(eloc + 2 * walker, e.x); atomicAdd(eloc + 2 * walker + 1, e.y); }
  }
  if (psi != nullptr && chunk == 0) {  // workgroup-uniform
    const cplx s = over_waves(lnpsi);
    if (tid == 0) {
      const cplx v = cexp(cplx{s.x - log_scale, s.y});
      psi[2 * walker] = v.x; psi[2 * walker + 1] = v.y;
    }
  }
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

static constexpr size_t kCrbmMaxLds = 158 * 1024;

// hidden units of q' resident at a time: all of them (cl.Hloop) if they fit, else the largest even window that leaves room for two
// workgroups per CU when it can (PYNQS_CRBM_WINDOW forces a window, for tests); 0: not even a window of two fits
static uint32_t crbm_window(const SDParams &p, const CrbmLayout &cl) {
  const int win_env = getenv("PYNQS_CRBM_WINDOW") ? atoi(getenv("PYNQS_CRBM_WINDOW")) : 0;  // (read per call: tests switch it)
  if (win_env >= 2) {
    const uint32_t hw = (uint32_t)win_env & ~1u;
    return hw >= (uint32_t)cl.Hloop ? (uint32_t)cl.Hloop : (lds_bytes_crbm(p, cl, hw) <= kCrbmMaxLds ? hw : 0u);
  }
  if (lds_bytes_crbm(p, cl, (uint32_t)cl.Hloop) <= kCrbmMaxLds) return (uint32_t)cl.Hloop;
  const size_t fixed = lds_bytes_crbm(p, cl, 0u), row = 16 * (size_t)(p.sorb + 1);
  for (size_t budget : {(size_t)(79 * 1024), kCrbmMaxLds}) {
    if (fixed + 3 * row > budget) continue;
    uint32_t hw = (uint32_t)((budget - fixed) / row - 1) & ~1u;
    if (hw >= 16u || budget == kCrbmMaxLds) return hw >= 2u ? hw : 0u;
  }
  return 0u;
}

extern "C" int64_t pynqs_crbm_table_bytes(int sorb, int nhidden) {
  CrbmLayout cl;
  if (!make_crbm_layout(sorb, nhidden, &cl)) return -1;
  return cl.total * 16;
}

extern "C" int pynqs_eloc_crbm_supported(int sorb, int nele, int noA, int noB, int nhidden) {
  SDParams p;
  PlanLayout pl;
  CrbmLayout cl;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl) || !make_crbm_layout(sorb, nhidden, &cl)) return 0;
  return crbm_window(p, cl) > 0 ? 1 : 0;
}

extern "C" int pynqs_crbm_table_build(const double *weights, const double *hidden_bias, const double *visible_bias, int sorb, int nhidden,
                                      void *table, void *stream) {
  pynqs::DeviceScope device_scope_(weights);
  CrbmLayout cl;
  if (!make_crbm_layout(sorb, nhidden, &cl)) return set_error(PYNQS_EINVAL, "bad sorb / nhidden");
  if (!weights || !hidden_bias || !table) return set_error(PYNQS_EINVAL, "null pointer");
  const int64_t n = (int64_t)cl.sorb * cl.Hs;
  const uint32_t grid = (uint32_t)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(crbm_table_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const cplx *)weights, (const cplx *)hidden_bias,
                     (const cplx *)visible_bias, cl, (cplx *)table);
  return check_launch("crbm_table_build");
}

extern "C" int pynqs_eloc_crbm(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                               const void *crbm_table, int nhidden, double log_scale, double *eloc, double *psi, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  PlanLayout pl;
  CrbmLayout cl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (!make_crbm_layout(sorb, nhidden, &cl)) return set_error(PYNQS_EINVAL, "bad nhidden");
  if (nbatch < 0 || nbatch > 0x3fffffffll) return set_error(PYNQS_EINVAL, "bad nbatch");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !crbm_table || !eloc) return set_error(PYNQS_EINVAL, "null pointer");
  const uint32_t hw = crbm_window(p, cl);
  if (hw == 0) return set_error(PYNQS_EINVAL, "the per-hidden-unit arrays of this RBM do not fit the LDS (pynqs_eloc_crbm_supported)");
  const bool windowed = hw < (uint32_t)cl.Hloop;
  const size_t lds = lds_bytes_crbm(p, cl, hw);
  const CrbmBlocks B = make_crbm_blocks(p);
  uint32_t nchunks = 1;  // few walkers: a walker's tiles over several workgroups (each repeats the per-walker set-up)
  if (nbatch < 1024) {
    nchunks = (uint32_t)((1024 + nbatch - 1) / nbatch);
    const uint32_t maxc = B.ntiles / 4 > 0 ? B.ntiles / 4 : 1;
    if (nchunks > maxc) nchunks = maxc;
  }
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, 16 * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");
  // workgroup size: the one that puts the most waves on a CU (a workgroup's waves share its LDS), as long as the walker has at
  // least two tiles per wave; PYNQS_CRBM_BLOCK overrides
  static const int blk_env = getenv("PYNQS_CRBM_BLOCK") ? atoi(getenv("PYNQS_CRBM_BLOCK")) : 0;
  uint32_t threads = kBlock;
  size_t best = 0;
  for (uint32_t b = kBlock; b <= 512; b *= 2) {
    size_t waves = (160 * 1024 / (lds + 256)) * (b / 64);
    if (waves > 16) waves = 16;
    if (b > kBlock && B.ntiles / nchunks < 2 * (b / 64)) break;
    if (waves > best) { best = waves; threads = b; }
  }
  if (blk_env == 128 || blk_env == 256 || blk_env == 512) threads = (uint32_t)blk_env;
  const int len = (sorb - 1) / 64 + 1;
  DISPATCH_LEN(len, {
    auto kfn = windowed ? eloc_crbm_kernel<LEN, true> : eloc_crbm_kernel<LEN, false>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    hipLaunchKernelGGL(kfn, dim3((uint32_t)grid), dim3(threads), lds, st, bra, p, pl, cl, B, nchunks, hw, (const double *)plan,
                       (const cplx *)crbm_table, log_scale, eloc, psi);
  });
  return check_launch("eloc_crbm");
}
