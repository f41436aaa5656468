// kernels_eloc_keys.hip -- SAMPLE_SPACE local energy, KEY-MAJOR:
//   E_loc(x) = sum_{y in S, y = x or a single / double excitation of x} <x|H|y> psi(y) / psi(x)
// (vmc/energy/eloc.py:326-401: psi(x') is taken from the table of the sample space S and is 0 outside it).  The column-major kernels
// (kernels_eloc.hip) enumerate the ncomb excitations of x and ask the table for each; this one walks the TABLE and asks of every key
// whether it is within a double excitation of x -- popcount(x ^ y) <= 4, five vector instructions per (walker, key) for one-word
// determinants -- and evaluates <x|H|y> from the two bit patterns for the few keys that are.  Work per walker is |S| instead of ncomb:
// the sample space of a run is 10^4 - 10^6 determinants whatever the orbital count, while ncomb grows like sorb^4 (7.9e3 for Fe2S2,
// 1.2e6 at sorb 120, 6.6e6 at sorb 184), so this is the kernel for large orbital spaces and for small tables; the host picks
// (pynqs_amd/energy.py) by |S| against ncomb.
//   - a wave owns W walkers (words in LDS / scalar registers) and streams the workgroup's chunk of keys: a key is loaded once per
//     lane and compared with the W walkers; candidates (walker, key index) are parked in a wave-private LDS queue and evaluated 64 at
//     a time, all lanes busy (kernels_eloc.hip's scheme);
//   - evaluation: holes = (x ^ y) & x, particles = (x ^ y) & y; degree, spin sectors, orbitals by ctz / clz; the matrix element from
//     the integral plan with the indices the excitation tables would have produced (detcore.h: build_walker_tables) and the same sign
//     rules (plan_dev.h: finish_double); singles add their nele terms from the walker's occupied list in LDS; <x|H|x> is computed by the
//     wave that meets the key equal to x, when it meets it;
//   - the keys need not be sorted and no hash table is involved; psi(x) is the table value of the key equal to x (0 if x is not in S).
// flip: the projected form's partner sum (flip.py:322-418): sum_{x'} <x|H|x'> eta_m(x') psi(flip x') -- the key y stands for x' = flip(y).
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"

#include <type_traits>

namespace pynqs {

#ifndef PYNQS_KEYS_W
#define PYNQS_KEYS_W 4
#endif
constexpr int kKeysWalkers = PYNQS_KEYS_W;   // walkers per wave (their 32-bit folds live in scalar registers, the words in LDS)
constexpr uint32_t kKeysQueue = 128;  // < 64 left over + 64 parked by one comparison

// alpha <-> beta occupations exchanged in place; returns true if eta_m = (-1)^(doubly occupied spatial orbitals) is -1 (the same for
// a determinant and its partner)
template <int LEN>
__device__ __forceinline__ bool spin_flip_ket_keys(uint64_t (&ket)[LEN]) {
  uint32_t pairs = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    const uint64_t w = ket[i];
    pairs += (uint32_t)__popcll(w & (w >> 1) & 0x5555555555555555ull);
    ket[i] = ((w >> 1) & 0x5555555555555555ull) | ((w & 0x5555555555555555ull) << 1);
  }
  return pairs & 1u;
}

template <int LEN>
__device__ __forceinline__ int lowest_bit(const uint64_t (&m)[LEN]) {
#pragma unroll
  for (int i = 0; i < LEN; ++i)
    if (m[i]) return 64 * i + __builtin_ctzll(m[i]);
  return 0;
}
template <int LEN>
__device__ __forceinline__ int highest_bit(const uint64_t (&m)[LEN]) {
#pragma unroll
  for (int i = LEN - 1; i >= 0; --i)
    if (m[i]) return 64 * i + 63 - __builtin_clzll(m[i]);
  return 0;
}
// parity of the number of occupied orbitals of x below orbital n (detcore.h: Walker::pm holds the same bit)
template <int LEN>
__device__ __forceinline__ uint32_t parity_below(const uint64_t (&x)[LEN], int n) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    const int word = n >> 6;
    const uint64_t m = i < word ? ~0ull : (i == word ? ((1ull << (n & 63)) - 1ull) : 0ull);
    c += (uint32_t)__popcll(x[i] & m);
  }
  return c & 1u;
}

template <int LEN, bool CPLX>
__global__ __launch_bounds__(kBlock) void eloc_sample_space_keys_kernel(const uint64_t *__restrict__ bra, int64_t nbatch, SDParams p, PlanLayout pl,
                                                                        uint32_t nchunks, int64_t chunk_len, const double *__restrict__ plan,
                                                                        const uint64_t *__restrict__ keys, int64_t nkeys,
                                                                        const double *__restrict__ wf, double *__restrict__ acc,
                                                                        double *__restrict__ psi0, bool flip) {
  constexpr int W = kKeysWalkers, NW = kBlock / 64;
  __shared__ uint64_t xs[NW][W][LEN];
  __shared__ uint8_t occ[NW][W][192];
  __shared__ uint32_t queue[NW][2][kKeysQueue];  // [0]: doubles (popcount(x ^ y) == 4), [1]: singles (== 2)
  __shared__ uint64_t qkeys[NW][2][kKeysQueue][LEN];  // the parked determinant itself (x' = the key, or its spin-flip partner): the
                                                       // evaluation then has ONE memory round trip (integral and psi together)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t group = blockIdx.x / nchunks;
  const uint32_t chunk = (uint32_t)(blockIdx.x - group * nchunks);
  const int64_t wbase = ((int64_t)group * NW + wave) * W;  // this wave's first walker
  const int sorb = p.sorb;
  // ---- the wave's walkers: words (scalar), occupied lists, <x|H|x>
  uint32_t fx[W];  // 32-bit folds of the walkers (scalar): all the common path needs
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const bool valid = wbase + w < nbatch;
    int nocc = 0;
    fx[w] = 0;
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      const uint64_t v = valid ? bra[(wbase + w) * LEN + i] : ~0ull;  // (no key is within four bits of all-ones)
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      const uint64_t xw = ((uint64_t)hi << 32) | lo;
      if (lane == 0) xs[wave][w][i] = xw;
      if ((xw >> lane) & 1ull) occ[wave][w][nocc + __popcll(xw & ((1ull << lane) - 1ull))] = (uint8_t)(64 * i + lane);
      nocc += __popcll(xw);
      fx[w] ^= lo ^ hi;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // <x|H|x> of the wave's walker w, by the whole wave, when (and where) the key equal to x turns up: once per walker over the whole
  // grid -- computed up front in every workgroup it cost sorb 184 (4278 terms per walker) a third of the kernel.
  // The nocc (nocc + 1) / 2 terms h(p,p), <pq||pq> (q < p) are dealt over the lanes: independent loads, one round trip per 64 terms.
  auto diagonal = [&](int w) -> double {
    const double *__restrict__ D1 = plan + pl.offD1;
    const double *__restrict__ D2 = plan + pl.offD2;
    double s = 0.0;
    int no = 0;
#pragma unroll
    for (int i = 0; i < LEN; ++i) no += __popcll(xs[wave][w][i]);
    const int nterms = no * (no + 1) / 2;
    for (int t = lane; t < nterms; t += 64) {
      int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > t) --a;
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      const int pos = t - a * (a + 1) / 2;
      const uint32_t pa = occ[wave][w][a];
      s += pos == 0 ? D1[pa] : D2[pa * (uint32_t)sorb + occ[wave][w][pos - 1]];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    return s;
  };

  double are[W], aim[W];
#pragma unroll
  for (int w = 0; w < W; ++w) { are[w] = 0.0; aim[w] = 0.0; }
  uint32_t cnts[2] = {0u, 0u};  // entries in this wave's queues (wave-uniform)

  // evaluation of the top n <= 64 parked (walker, key) pairs of queue S.  Singles have their own queue: one costs nele gathers, and a
  // lane-per-candidate loop lasts as long for one single among 63 doubles as for 64 singles.
  auto evaluate = [&](auto S_, uint32_t n) {
    constexpr int S = decltype(S_)::value;
    uint32_t &count = cnts[S];
    __builtin_amdgcn_wave_barrier();
    double h = 0.0, vr = 0.0, vi = 0.0;
    uint32_t w = 0;
    if ((uint32_t)lane < n) {
      const uint32_t slot = count - n + (uint32_t)lane;
      const uint32_t code = queue[wave][S][slot];
      w = code >> 28;
      const int64_t k = (int64_t)(code & 0x07ffffffu);
      const bool minus = (code >> 27) & 1u;  // flip: eta_m(x') = -1
      uint64_t xx[LEN], y[LEN], hx[LEN], py[LEN];
      int nh = 0, np = 0, nha = 0, npa = 0;
#pragma unroll
      for (int i = 0; i < LEN; ++i) {
        xx[i] = xs[wave][w][i];
        y[i] = qkeys[wave][S][slot][i];
      }
#pragma unroll
      for (int i = 0; i < LEN; ++i) {
        const uint64_t d = xx[i] ^ y[i];
        hx[i] = d & xx[i];
        py[i] = d & y[i];
        nh += __popcll(hx[i]); np += __popcll(py[i]);
        nha += __popcll(hx[i] & 0x5555555555555555ull); npa += __popcll(py[i] & 0x5555555555555555ull);
      }
      if (nh == np && nha == npa && nh == (S == 1 ? 1 : 2)) {
        if (S == 1) {
          const int ho = lowest_bit<LEN>(hx), q = lowest_bit<LEN>(py);
          const uint32_t K = (uint32_t)pl.K;
          const uint32_t pq = (((ho & 1) ? K : 0u) + ((uint32_t)ho >> 1)) * K + ((uint32_t)q >> 1);
          const double *__restrict__ row = plan + pl.offS2 + (size_t)pq * sorb;
          double s = plan[pl.offS1 + pq];
          int no = 0;
#pragma unroll
          for (int i = 0; i < LEN; ++i) no += __popcll(xx[i]);
#pragma unroll 8
          for (int j = 0; j < no; ++j) s += row[occ[wave][w][j]];
          const uint32_t par = parity_below<LEN>(xx, ho) ^ parity_below<LEN>(xx, q) ^ (uint32_t)(ho < q);
          h = par ? -s : s;
        } else {
          const int h0 = highest_bit<LEN>(hx), h1 = lowest_bit<LEN>(hx), q0 = highest_bit<LEN>(py), q1 = lowest_bit<LEN>(py);
          const uint32_t P = parity_below<LEN>(xx, h0) ^ parity_below<LEN>(xx, h1) ^ parity_below<LEN>(xx, q0) ^ parity_below<LEN>(xx, q1);
          if (nha == 1) {  // one alpha, one beta: the alpha single (ha -> qa) and the beta single (hb -> qb)
            const int ha = (h0 & 1) ? h1 : h0, hb = (h0 & 1) ? h0 : h1, qa = (q0 & 1) ? q1 : q0, qb = (q0 & 1) ? q0 : q1;
            const uint32_t K = (uint32_t)pl.K;
            const size_t idx = ((size_t)(((uint32_t)qb >> 1) * K + ((uint32_t)hb >> 1)) * K + ((uint32_t)qa >> 1)) * K + ((uint32_t)ha >> 1);
            const double v = plan[pl.offVab + idx];
            const uint32_t par = P ^ (uint32_t)(ha < qa) ^ (uint32_t)(hb < qb) ^ (uint32_t)(ha < qb) ^ (uint32_t)(hb < qa) ^ 1u;
            h = par ? -v : v;
          } else {  // same spin: hole pair (h0 > h1), particle pair (q0 > q1)
            const uint32_t spin = (uint32_t)h0 & 1u, NP = (uint32_t)pl.NP;
            const uint32_t mh0 = (uint32_t)h0 >> 1, mh1 = (uint32_t)h1 >> 1, mq0 = (uint32_t)q0 >> 1, mq1 = (uint32_t)q1 >> 1;
            const uint32_t ij = mh0 * (mh0 - 1) / 2 + mh1, ab = mq0 * (mq0 - 1) / 2 + mq1;
            const double v = plan[pl.offVss + ((size_t)spin * NP + ab) * NP + ij];
            const uint32_t par = P ^ 1u ^ (uint32_t)(h0 < q0) ^ (uint32_t)(h1 < q0) ^ (uint32_t)(h0 < q1) ^ (uint32_t)(h1 < q1);
            h = par ? -v : v;
          }
        }
        if (minus) h = -h;
        if constexpr (CPLX) { vr = wf[2 * k]; vi = wf[2 * k + 1]; }
        else vr = wf[k];
      }
    }
    count = __builtin_amdgcn_readfirstlane(count - n);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const double hh = w == (uint32_t)i ? h : 0.0;
      are[i] = fma(hh, vr, are[i]);
      if constexpr (CPLX) aim[i] = fma(hh, vi, aim[i]);
    }
  };
  auto park = [&](auto S_, uint32_t code, const uint64_t (&y)[LEN], bool pass) {
    constexpr int S = decltype(S_)::value;
    const uint64_t m = __ballot(pass);
    if (!m) return;
    if (pass) {
      const uint32_t slot = cnts[S] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      queue[wave][S][slot] = code;
#pragma unroll
      for (int i = 0; i < LEN; ++i) qkeys[wave][S][slot][i] = y[i];
    }
    cnts[S] = __builtin_amdgcn_readfirstlane(cnts[S] + (uint32_t)__popcll(m));
    if (cnts[S] >= 64u) evaluate(S_, 64u);
  };
  typedef std::integral_constant<int, 0> Doubles;
  typedef std::integral_constant<int, 1> Singles;

  // ---- the workgroup's chunk of keys: U keys per lane and step, the next step's keys requested before this step's are compared
  // (a wave alone with one dependent load per step waits an L2 round trip per 64 keys: 0.50 ms for 8192 walkers x 2.8e4 keys)
#ifndef PYNQS_KEYS_U3
#define PYNQS_KEYS_U3 2
#endif
  constexpr int U = LEN == 3 ? PYNQS_KEYS_U3 : 4;
  const int64_t k_lo = (int64_t)chunk * chunk_len, k_hi = min(k_lo + chunk_len, nkeys);
  uint64_t ynext[U][LEN];
  auto request = [&](int64_t k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = k0 + 64 * u + lane;
#pragma unroll
      for (int i = 0; i < LEN; ++i) ynext[u][i] = k < k_hi ? keys[k * LEN + i] : 0ull;
    }
  };
  request(k_lo);
  for (int64_t k0 = k_lo; k0 < k_hi; k0 += 64 * U) {
    uint64_t y[U][LEN];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < LEN; ++i) y[u][i] = ynext[u][i];
    if (k0 + 64 * U < k_hi) request(k0 + 64 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = k0 + 64 * u + lane;
      const bool in = k < k_hi;
      const uint32_t minus = flip && spin_flip_ket_keys<LEN>(y[u]) ? (1u << 27) : 0u;  // y <- x' = flip(key); eta_m(x') = eta_m(key)
      // the cheap test (spin sectors and hole / particle balance are checked by the evaluation).  Common path: the 32-bit FOLD of a
      // determinant (XOR of its 32-bit quarters): folding never increases a Hamming distance, so fold(x) ^ fold(y) with more than four
      // bits set rules the pair out -- one xor and one popcount per (walker, key) whatever the word count, and for the W walkers of a
      // key together one minimum and ONE wave-wide question (a branch per (walker, key group) cost more than the popcounts; one
      // question for all U key groups of a step was tried in round 3 and is slower: 0.19 -> 0.22 ms at two words, the folds of
      // U x W pairs held across the branch).  Unrelated determinants differ in tens of bits and their folds in ~16 +- 3.  (Keys past
      // the chunk's end were loaded as 0; walkers past the batch's end are all-ones patterns: whatever their folds say, the full
      // comparison below rejects them.)
      uint32_t fy = 0;
#pragma unroll
      for (int i = 0; i < LEN; ++i) fy ^= (uint32_t)y[u][i] ^ (uint32_t)(y[u][i] >> 32);
      int folded[W], least = 1 << 20;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        folded[w] = __popc(fx[w] ^ fy);
        least = min(least, folded[w]);
      }
      if (!__ballot(least <= 4)) continue;  // (wave-uniform)
#pragma unroll
      for (int w = 0; w < W; ++w) {
        if (!__ballot(folded[w] <= 4)) continue;  // (wave-uniform)
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < LEN; ++i) cnt += __popcll(xs[wave][w][i] ^ y[u][i]);  // (rare path: the walker's words from LDS, one address per wave)
        if (__ballot(cnt <= 4)) {  // (wave-uniform)
          const uint32_t code = ((uint32_t)w << 28) | minus | (uint32_t)k;
          if (__ballot(in && cnt == 0)) {  // the key equal to the walker itself (keys are distinct: one lane, once per walker and launch)
            const double hd = diagonal(w);
            if (in && cnt == 0) {
              double vr, vi = 0.0;
              if constexpr (CPLX) { vr = wf[2 * k]; vi = wf[2 * k + 1]; }
              else vr = wf[k];
              const double hh = minus ? -hd : hd;
              are[w] = fma(hh, vr, are[w]);
              if constexpr (CPLX) aim[w] = fma(hh, vi, aim[w]);
              if (!flip) {
                double *__restrict__ out = psi0 + (CPLX ? 2 : 1) * (wbase + w);
                out[0] = vr;
                if constexpr (CPLX) out[1] = vi;
              }
            }
          }
          park(Doubles{}, code, y[u], in && cnt == 4);
          park(Singles{}, code, y[u], in && cnt == 2);
        }
      }
    }
  }
  while (cnts[0]) evaluate(Doubles{}, min(cnts[0], 64u));
  while (cnts[1]) evaluate(Singles{}, min(cnts[1], 64u));

  // ---- per-walker sums over the lanes; chunks meet through atomics
#pragma unroll
  for (int w = 0; w < W; ++w) {
    double re = are[w], im = aim[w];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      re += __shfl_xor(re, o);
      if constexpr (CPLX) im += __shfl_xor(im, o);
    }
    if (lane == 0 && wbase + w < nbatch) {
      double *__restrict__ out = acc + (CPLX ? 2 : 1) * (wbase + w);
      if (nchunks == 1) {
        out[0] = re;
        if constexpr (CPLX) out[1] = im;
      } else {
        atomicAdd(out, re);
        if constexpr (CPLX) atomicAdd(out + 1, im);
      }
    }
  }
}

template <bool CPLX>
__global__ __launch_bounds__(kBlock) void eloc_divide_keys_kernel(double *__restrict__ acc, const double *__restrict__ psi0, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if constexpr (CPLX) {
    const double a = acc[2 * i], b = acc[2 * i + 1], c = psi0[2 * i], d = psi0[2 * i + 1], den = c * c + d * d;
    acc[2 * i] = (a * c + b * d) / den;
    acc[2 * i + 1] = (b * c - a * d) / den;
  } else {
    acc[i] = acc[i] / psi0[i];
  }
}

}  // namespace pynqs

using namespace pynqs;

extern "C" int pynqs_eloc_sample_space_keys(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                            const uint64_t *keys, int64_t nkeys, const double *wf, int wf_is_complex, int flip,
                                            double *eloc, double *psi0, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || nkeys < 0 || nkeys >= (1ll << 27)) return set_error(PYNQS_EINVAL, "bad nbatch / nkeys (nkeys < 2^27)");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !eloc || !psi0 || (nkeys > 0 && (!keys || !wf))) return set_error(PYNQS_EINVAL, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = wf_is_complex ? 16 : 8;
  constexpr int kPerGroup = (kBlock / 64) * kKeysWalkers;  // walkers per workgroup
  const int64_t groups = (nbatch + kPerGroup - 1) / kPerGroup;
  // enough workgroups to fill the chip, chunks of at least 2048 keys
  static const int64_t want = getenv("PYNQS_KEYS_WG") ? atoll(getenv("PYNQS_KEYS_WG")) : 4096;
  int64_t nchunks = groups >= want ? 1 : (want + groups - 1) / groups;
  const int64_t maxc = (nkeys + 2047) / 2048;
  if (nchunks > maxc) nchunks = maxc;
  if (nchunks < 1) nchunks = 1;
  int64_t chunk_len = (nkeys + nchunks - 1) / nchunks;
  chunk_len = (chunk_len + 63) & ~(int64_t)63;
  if (chunk_len < 64) chunk_len = 64;
  nchunks = nkeys > 0 ? (nkeys + chunk_len - 1) / chunk_len : 1;
  const uint64_t grid = (uint64_t)groups * (uint64_t)nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");
  if (!flip && hipMemsetAsync(psi0, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");  // x not in S: psi(x) = 0
  const int len = (sorb - 1) / 64 + 1;
  DISPATCH_LEN(len, {
    if (wf_is_complex)
      hipLaunchKernelGGL((eloc_sample_space_keys_kernel<LEN, true>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, nbatch, p, pl, (uint32_t)nchunks,
                         chunk_len, (const double *)plan, keys, nkeys, wf, eloc, psi0, flip != 0);
    else
      hipLaunchKernelGGL((eloc_sample_space_keys_kernel<LEN, false>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, nbatch, p, pl, (uint32_t)nchunks,
                         chunk_len, (const double *)plan, keys, nkeys, wf, eloc, psi0, flip != 0);
  });
  const uint32_t g2 = (uint32_t)((nbatch + kBlock - 1) / kBlock);
  if (wf_is_complex) hipLaunchKernelGGL((eloc_divide_keys_kernel<true>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  else hipLaunchKernelGGL((eloc_divide_keys_kernel<false>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  return check_launch("eloc_sample_space_keys");
}
