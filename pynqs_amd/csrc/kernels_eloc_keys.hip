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
// INDEXED (round 3): the same evaluation behind a BLOCK INDEX of the keys instead of the stream over all of them.  The sorb bits are cut
// into five blocks; a determinant within a double excitation of x differs from it in at most four bits, so it agrees with x in at least
// one whole block (multi-index hashing).  The index (pynqs_keys_index_build) holds, per block, the keys' block values sorted with the
// key numbers beside them; a walker looks its own five block values up (binary searches, all walkers and blocks of a wave at once, one
// per lane), and only the keys found there are loaded and compared -- each counted in the first block it agrees in.  Work per walker is
// the number of such keys (tens for a table of samples at sorb 120 / 184; ~|S| / 8 for Fe2S2's CAS-like table) instead of |S|.
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
#ifndef PYNQS_INDEX_W
#define PYNQS_INDEX_W 1
#endif
constexpr int kIndexWalkers = PYNQS_INDEX_W;  // ... of the INDEXED form: nothing is shared between the walkers of a wave there but the queues
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

template <int LEN, bool CPLX, bool INDEXED>
__global__ __launch_bounds__(kBlock) void eloc_sample_space_keys_kernel(const uint64_t *__restrict__ bra, int64_t nbatch, SDParams p, PlanLayout pl,
                                                                        uint32_t nchunks, int64_t chunk_len, const double *__restrict__ plan,
                                                                        const uint64_t *__restrict__ keys, int64_t nkeys,
                                                                        const uint64_t *__restrict__ svals, const uint32_t *__restrict__ perm,
                                                                        const double *__restrict__ wf, double *__restrict__ acc,
                                                                        double *__restrict__ psi0, bool flip) {
  constexpr int W = INDEXED ? kIndexWalkers : kKeysWalkers, NW = kBlock / 64, NB = kIndexBlocks;
  __shared__ uint32_t rng[INDEXED ? NW : 1][W][NB][2];  // INDEXED: [first, last) of the walker's block value in the block's sorted list
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
  // h(p,p) for the occupied p, <pq||pq> for the occupied pairs q < p: lane l keeps the orbitals number l, l + 64, l + 128 of the occupied
  // list and meets every later orbital p of the list in turn -- p is wave-uniform (its row of the table is one base address), the loads of
  // successive p are independent, nothing is decoded (round 3; until then the nocc (nocc + 1) / 2 terms were dealt over the lanes by a
  // triangular-number inversion per term: 2.5 x the time at sorb 184, where this sum was a third of the indexed kernel).
  auto diagonal = [&](int w) -> double {
    const double *__restrict__ D1 = plan + pl.offD1;
    const double *__restrict__ D2 = plan + pl.offD2;
    int no = 0;
#pragma unroll
    for (int i = 0; i < LEN; ++i) no += __popcll(xs[wave][w][i]);
    constexpr int SL = LEN;  // occupied orbitals per lane: no <= 64 LEN
    uint32_t mine[SL];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      mine[j] = lane + 64 * j < no ? occ[wave][w][lane + 64 * j] : 0u;
      if (lane + 64 * j < no) s += D1[mine[j]];
    }
    constexpr int DU = 4;  // rows in flight per batch (8 / 16: fewer round trips but 4 / 3 waves per SIMD instead of 5 at three words; slower)
    for (int a0 = 1; a0 < no; a0 += DU) {
      double v[DU][SL];
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int a = a0 + u;
        const double *__restrict__ row = D2 + (size_t)occ[wave][w][min(a, no - 1)] * (uint32_t)sorb;  // (one LDS address per wave)
#pragma unroll
        for (int j = 0; j < SL; ++j) v[u][j] = a < no && lane + 64 * j < a ? row[mine[j]] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < DU; ++u)
#pragma unroll
        for (int j = 0; j < SL; ++j) s += v[u][j];
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

  // one (walker w, x' = y) pair per lane with popcount(x ^ y) = cnt <= 4 somewhere in the wave: the key equal to the walker brings <x|H|x>
  // and psi(x); doubles and singles are parked
  auto take = [&](int w, uint32_t minus, int64_t k, const uint64_t (&y)[LEN], bool in, int cnt) {
    const uint32_t code = ((uint32_t)w << 28) | minus | (uint32_t)k;
    if (__ballot(in && cnt == 0)) {  // the key equal to the walker itself (keys are distinct: one lane, once per walker and launch)
      const double hd = diagonal(w);
      if (in && cnt == 0) {
        double vr, vi = 0.0;
        if constexpr (CPLX) { vr = wf[2 * k]; vi = wf[2 * k + 1]; }
        else vr = wf[k];
        const double hh = minus ? -hd : hd;
#pragma unroll
        for (int i = 0; i < W; ++i) {
          if (i == w) {  // (w is wave-uniform)
            are[i] = fma(hh, vr, are[i]);
            if constexpr (CPLX) aim[i] = fma(hh, vi, aim[i]);
          }
        }
        if (!flip) {
          double *__restrict__ out = psi0 + (CPLX ? 2 : 1) * (wbase + w);
          out[0] = vr;
          if constexpr (CPLX) out[1] = vi;
        }
      }
    }
    park(Doubles{}, code, y, in && cnt == 4);
    park(Singles{}, code, y, in && cnt == 2);
  };

  if constexpr (INDEXED) {
    // ---- the walkers' block values looked up: lane t = (walker, block, first / one past last), a binary search each
    int blo[NB + 1];
#pragma unroll
    for (int b = 0; b <= NB; ++b) blo[b] = index_block_lo(sorb, b);
    if (lane < W * NB * 2) {
      const int w = lane / (2 * NB), b = (lane >> 1) % NB, upper = lane & 1;
      uint64_t xw[LEN];
#pragma unroll
      for (int i = 0; i < LEN; ++i) xw[i] = xs[wave][w][i];
      if (flip) (void)spin_flip_ket_keys<LEN>(xw);  // flip(key) agrees with x in a block  <=>  the key agrees with flip(x) in it
      int lo_bit = 0, hi_bit = 0;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
        if (bb == b) { lo_bit = blo[bb]; hi_bit = blo[bb + 1]; }
      const uint64_t v = ((uint64_t)b << kIndexTagShift) | index_block_value<LEN>(xw, lo_bit, hi_bit);
      const uint64_t *__restrict__ sv = svals + (size_t)b * (size_t)nkeys;
      int64_t first = 0, n = wbase + w < nbatch ? nkeys : 0;
      while (n > 0) {  // first position with sv[pos] >= v (upper: > v)
        const int64_t half = n >> 1;
        const uint64_t m = sv[first + half];
        const bool right = upper ? m <= v : m < v;
        first = right ? first + half + 1 : first;
        n = right ? n - half - 1 : half;
      }
      rng[wave][w][b][upper] = (uint32_t)first;
    }
    __builtin_amdgcn_wave_barrier();
    // block masks (wave-uniform): a candidate found through block b counts only if it disagrees with x in every block before b
    uint64_t bmask[NB - 1][LEN];
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
#pragma unroll
      for (int i = 0; i < LEN; ++i) {
        const int lo = max(blo[b] - 64 * i, 0), hi = min(blo[b + 1] - 64 * i, 64);
        bmask[b][i] = hi > lo ? ((hi == 64 ? 0ull : (1ull << hi)) - (1ull << lo)) : 0ull;
      }
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
      if (wbase + w >= nbatch) break;  // (wave-uniform)
      uint32_t first[NB], off[NB + 1];
      off[0] = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        first[b] = rng[wave][w][b][0];
        off[b + 1] = off[b] + (rng[wave][w][b][1] - first[b]);
      }
      uint64_t xw[LEN];
#pragma unroll
      for (int i = 0; i < LEN; ++i) xw[i] = xs[wave][w][i];
      for (uint32_t j0 = 0; j0 < off[NB]; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        const bool in = j < off[NB];
        int b = 0;
        uint32_t pos = first[0] + j;
#pragma unroll
        for (int bb = 1; bb < NB; ++bb)
          if (j >= off[bb]) { b = bb; pos = first[bb] + (j - off[bb]); }
        const int64_t k = in ? (int64_t)perm[(size_t)b * (size_t)nkeys + pos] : 0;
        uint64_t y[LEN];
#pragma unroll
        for (int i = 0; i < LEN; ++i) y[i] = in ? keys[k * LEN + i] : 0ull;
        const uint32_t minus = flip && spin_flip_ket_keys<LEN>(y) ? (1u << 27) : 0u;
        int cnt = 0;
        bool counted_before = false;
        uint64_t d[LEN];
#pragma unroll
        for (int i = 0; i < LEN; ++i) { d[i] = xw[i] ^ y[i]; cnt += __popcll(d[i]); }
#pragma unroll
        for (int bb = 0; bb < NB - 1; ++bb) {
          uint64_t any = 0;
#pragma unroll
          for (int i = 0; i < LEN; ++i) any |= d[i] & bmask[bb][i];
          counted_before = counted_before || (bb < b && any == 0);
        }
        const bool mine = in && !counted_before && cnt <= 4;
        if (__ballot(mine)) take(w, minus, k, y, mine, cnt);
      }
    }
  } else {
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
          take(w, minus, k, y[u], in, cnt);
        }
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

static int launch_keys(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan, const uint64_t *keys,
                       int64_t nkeys, const void *index, const double *wf, int wf_is_complex, int flip, double *eloc, double *psi0,
                       void *stream) {
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || nkeys < 0 || nkeys >= (1ll << 27)) return set_error(PYNQS_EINVAL, "bad nbatch / nkeys (nkeys < 2^27)");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !eloc || !psi0 || (nkeys > 0 && (!keys || !wf))) return set_error(PYNQS_EINVAL, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = wf_is_complex ? 16 : 8;
  const int kPerGroup = (kBlock / 64) * (index ? kIndexWalkers : kKeysWalkers);  // walkers per workgroup
  const int64_t groups = (nbatch + kPerGroup - 1) / kPerGroup;
  // streamed form: enough workgroups to fill the chip, chunks of at least 2048 keys (the indexed form has one workgroup per group)
  static const int64_t want = getenv("PYNQS_KEYS_WG") ? atoll(getenv("PYNQS_KEYS_WG")) : 4096;
  int64_t nchunks = groups >= want || index ? 1 : (want + groups - 1) / groups;
  const int64_t maxc = (nkeys + 2047) / 2048;
  if (nchunks > maxc) nchunks = maxc;
  if (nchunks < 1) nchunks = 1;
  int64_t chunk_len = (nkeys + nchunks - 1) / nchunks;
  chunk_len = (chunk_len + 63) & ~(int64_t)63;
  if (chunk_len < 64) chunk_len = 64;
  nchunks = nkeys > 0 && !index ? (nkeys + chunk_len - 1) / chunk_len : 1;
  const uint64_t grid = (uint64_t)groups * (uint64_t)nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");
  if (!flip && hipMemsetAsync(psi0, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");  // x not in S: psi(x) = 0
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t *svals = (const uint64_t *)index;
  const uint32_t *perm = index ? (const uint32_t *)(svals + (size_t)kIndexBlocks * (size_t)nkeys) : nullptr;
#define PYNQS_KEYS_LAUNCH(C, I)                                                                                                             \
  hipLaunchKernelGGL((eloc_sample_space_keys_kernel<LEN, C, I>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, nbatch, p, pl, (uint32_t)nchunks, \
                     chunk_len, (const double *)plan, keys, nkeys, svals, perm, wf, eloc, psi0, flip != 0)
  DISPATCH_LEN(len, {
    if (index) {
      if (wf_is_complex) PYNQS_KEYS_LAUNCH(true, true);
      else PYNQS_KEYS_LAUNCH(false, true);
    } else {
      if (wf_is_complex) PYNQS_KEYS_LAUNCH(true, false);
      else PYNQS_KEYS_LAUNCH(false, false);
    }
  });
#undef PYNQS_KEYS_LAUNCH
  const uint32_t g2 = (uint32_t)((nbatch + kBlock - 1) / kBlock);
  if (wf_is_complex) hipLaunchKernelGGL((eloc_divide_keys_kernel<true>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  else hipLaunchKernelGGL((eloc_divide_keys_kernel<false>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  return check_launch(index ? "eloc_sample_space_indexed" : "eloc_sample_space_keys");
}

extern "C" int pynqs_eloc_sample_space_keys(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                            const uint64_t *keys, int64_t nkeys, const double *wf, int wf_is_complex, int flip,
                                            double *eloc, double *psi0, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return launch_keys(bra, nbatch, sorb, nele, noA, noB, plan, keys, nkeys, nullptr, wf, wf_is_complex, flip, eloc, psi0, stream);
}

extern "C" int pynqs_eloc_sample_space_indexed(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                               const uint64_t *keys, int64_t nkeys, const void *index, const double *wf,
                                               int wf_is_complex, int flip, double *eloc, double *psi0, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  if (nkeys > 0 && !index) return set_error(PYNQS_EINVAL, "null index");
  if (nkeys == 0) return launch_keys(bra, nbatch, sorb, nele, noA, noB, plan, keys, nkeys, nullptr, wf, wf_is_complex, flip, eloc, psi0, stream);
  return launch_keys(bra, nbatch, sorb, nele, noA, noB, plan, keys, nkeys, index, wf, wf_is_complex, flip, eloc, psi0, stream);
}
