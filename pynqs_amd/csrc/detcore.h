// detcore.h -- device-side building blocks shared by every kernel of libpynqs_amd (gfx950 only).
//
// Design (MI355X-first, not a translation of cpp_src/cuda):
//  * one workgroup works on ONE walker, so the walker's occupation words and its prefix-parity
//    masks are wave-uniform (SGPRs); fermionic signs are bit tests on those masks instead of the
//    reference's popcount loops (cpp_src/cpu/onstate.cpp:22-32);
//  * the walker's excitation tables (singles lists, hole-pair / particle-pair lists) are built once
//    per workgroup in LDS; an excitation rank is then ONE magic-number division plus two LDS reads,
//    instead of the div/mod/sqrt chain of cpp_src/cpu/excitation.cpp:18-110 per element;
//  * consecutive lanes take consecutive excitation ranks, so Hmat/comb stores are fully coalesced;
//  * all global indexing is 64-bit (the reference overflows int32 at cuda/kernel.cu:243-263).
#pragma once
#include <stdlib.h>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pynqs {

constexpr int kMaxSorb = 192;
constexpr int kBlock = 256;

// Exact n / d for n < 2^24 by multiply-shift: m = ceil(2^s / d), s = 24 + ceil(log2 d).
struct MagicDiv {
  uint32_t m;
  uint32_t s;
  uint32_t d;
};

static inline MagicDiv make_magic(uint32_t d) {
  MagicDiv r;
  if (d == 0) d = 1;  // never used for division in that case (empty block)
  uint32_t l = 0;
  while ((1u << l) < d) ++l;
  r.s = 24 + l;
  r.m = (uint32_t)(((1ull << r.s) + d - 1) / d);
  r.d = d;
  return r;
}

__device__ __forceinline__ uint32_t mdiv(uint32_t n, const MagicDiv &k) {
  return (uint32_t)(((uint64_t)n * k.m) >> k.s);
}

// Everything the enumeration needs, computed once on the host (excitation.cpp:20-42).
struct SDParams {
  int sorb, nele, noA, noB, nvA, nvB;
  int noAA, noBB, nvAA, nvBB;  // pair counts
  int nSa, nSb;                // noA*nvA, noB*nvB
  uint32_t d0, d1, d2, d3;     // cumulative block ends [Sa, Sb, Daa, Dbb)
  uint32_t nsd;                // total singles+doubles
  uint32_t rotA, rotB;         // d1 % noAA, d2 % noBB : the `idx % noAA` quirk as a cyclic rotation
  MagicDiv divNoA, divNoB, divNoAA, divNoBB, divNSa;
  // LDS table offsets (in uint32 entries)
  int offSa, offSb, offHPa, offPPa, offHPb, offPPb, tabEntries;
};

static inline bool make_sd_params(int sorb, int nele, int noA, int noB, SDParams *p) {
  if (sorb < 1 || sorb > kMaxSorb || noA < 0 || noB < 0) return false;
  int k = sorb / 2;
  int nvA = k - noA, nvB = k - noB;
  if (nvA < 0 || nvB < 0) return false;
  p->sorb = sorb; p->nele = nele; p->noA = noA; p->noB = noB; p->nvA = nvA; p->nvB = nvB;
  p->noAA = noA * (noA - 1) / 2; p->noBB = noB * (noB - 1) / 2;
  p->nvAA = nvA * (nvA - 1) / 2; p->nvBB = nvB * (nvB - 1) / 2;
  p->nSa = noA * nvA; p->nSb = noB * nvB;
  int64_t nDaa = (int64_t)p->noAA * p->nvAA, nDbb = (int64_t)p->noBB * p->nvBB;
  int64_t nDab = (int64_t)p->nSa * p->nSb;
  int64_t tot = p->nSa + p->nSb + nDaa + nDbb + nDab;
  if (tot >= (1ll << 24)) return false;  // magic division range; sorb <= 192 never reaches it
  p->d0 = p->nSa; p->d1 = p->d0 + p->nSb; p->d2 = p->d1 + (uint32_t)nDaa; p->d3 = p->d2 + (uint32_t)nDbb;
  p->nsd = (uint32_t)tot;
  p->rotA = p->noAA ? p->d1 % p->noAA : 0;
  p->rotB = p->noBB ? p->d2 % p->noBB : 0;
  p->divNoA = make_magic(noA); p->divNoB = make_magic(noB);
  p->divNoAA = make_magic(p->noAA); p->divNoBB = make_magic(p->noBB); p->divNSa = make_magic(p->nSa);
  int o = 0;
  p->offSa = o; o += p->nSa;
  p->offSb = o; o += p->nSb;
  p->offHPa = o; o += p->noAA;
  p->offPPa = o; o += p->nvAA;
  p->offHPb = o; o += p->noBB;
  p->offPPb = o; o += p->nvBB;
  p->tabEntries = o;
  return true;
}

// ---- bit helpers ---------------------------------------------------------------------------------

// bit n of the result = parity of the bits of x strictly below n
__device__ __forceinline__ uint64_t prefix_parity_excl(uint64_t x) {
  uint64_t y = x << 1;
  y ^= y << 1; y ^= y << 2; y ^= y << 4; y ^= y << 8; y ^= y << 16; y ^= y << 32;
  return y;
}

template <int LEN>
struct Walker {
  uint64_t w[LEN];   // occupation words (wave-uniform)
  uint64_t pm[LEN];  // prefix-parity masks: bit n of pm[n/64] = parity(#occupied below n)
};

template <int LEN>
__device__ __forceinline__ void load_walker(const uint64_t *__restrict__ bra, Walker<LEN> &wk) {
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    uint64_t v = bra[i];
    // the address is workgroup-uniform: keep the value in scalar registers
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    v = ((uint64_t)hi << 32) | lo;
    wk.w[i] = v;
    uint64_t p = prefix_parity_excl(v);
    wk.pm[i] = carry ? ~p : p;
    carry ^= (uint32_t)__popcll(v) & 1u;
  }
}

template <int LEN>
__device__ __forceinline__ uint64_t pick(const uint64_t (&a)[LEN], int word) {
  if constexpr (LEN == 1) return a[0];
  else if constexpr (LEN == 2) return word ? a[1] : a[0];
  else return word == 0 ? a[0] : (word == 1 ? a[1] : a[2]);
}

template <int LEN>
__device__ __forceinline__ uint32_t bit_of(const uint64_t (&a)[LEN], int n) {
  return (uint32_t)(pick<LEN>(a, n >> 6) >> (n & 63)) & 1u;
}

template <int LEN>
__device__ __forceinline__ void toggle(uint64_t (&a)[LEN], int n) {
  uint64_t b = 1ull << (n & 63);
  if constexpr (LEN == 1) a[0] ^= b;
  else {
#pragma unroll
    for (int i = 0; i < LEN; ++i) a[i] ^= ((n >> 6) == i) ? b : 0ull;
  }
}

// ---- integrals ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t pair_index(uint32_t hi, uint32_t lo) { return hi * (hi - 1) / 2 + lo; }

// packed-triangle offset of (ij, kl); fits 32 bits for sorb <= 192 (max 1.68e8)
__device__ __forceinline__ uint32_t tri_index(uint32_t ij, uint32_t kl) {
  uint32_t P = ij > kl ? ij : kl, Q = ij > kl ? kl : ij;
  return P * (P + 1) / 2 + Q;
}

// <ij||kl> for arbitrary order, hamiltonian.cpp:14-31
template <typename T>
__device__ __forceinline__ T two_body(const T *__restrict__ h2e, int i, int j, int k, int l) {
  if (i == j || k == l) return T(0);
  uint32_t ij = i > j ? pair_index(i, j) : pair_index(j, i);
  uint32_t kl = k > l ? pair_index(k, l) : pair_index(l, k);
  T v = h2e[tri_index(ij, kl)];
  bool neg = (i > j) != (k > l);
  return neg ? -v : v;
}

// ---- sorted-key table (WavefunctionLUT) -----------------------------------------------------------
// cpu_tensor.cpp:589-637, little-endian branch: keys compare as multi-word integers, most significant
// word LAST.  Returns the position of q or -1.
template <int LEN>
__device__ __forceinline__ int64_t lut_find(const uint64_t *__restrict__ keys, int64_t nkeys, const uint64_t (&q)[LEN]) {
  int64_t lo = 0, hi = nkeys - 1;
  while (lo <= hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    int c = 0;
#pragma unroll
    for (int w = LEN - 1; w >= 0; --w) {
      const uint64_t kv = keys[mid * LEN + w];
      if (c == 0) c = kv < q[w] ? -1 : (kv > q[w] ? 1 : 0);
    }
    if (c == 0) return mid;
    if (c < 0) lo = mid + 1;
    else hi = mid - 1;
  }
  return -1;
}

// ---- hash table over the same keys (open addressing, linear probing, load factor <= 1/4) -----------
// slot = 2 (one-word keys) or 4 (two/three-word keys) 64-bit words: the key words first, the LAST word is the
// index into the sorted key array (-1 = empty).  A slot is read with 16-byte loads: one vector-memory
// instruction per probe for one-word keys -- the fused local-energy kernel is bound by the number of such
// instructions (TA busy 82 %, profiles/r01_fe2s2_eloc_sample_space_v1.txt), not by bytes.  The low load factor
// keeps the wave-wide maximum probe count (which is what a wave pays) near 2.
// Replaces the 15-24 dependent probes of the binary search (the reference has an optional GPU hash table for the
// same purpose, cuda/hashTable.cu, disabled by USE_HASH = False in utils/public_function.py:23).
typedef uint64_t hash_u64x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline uint64_t hash_capacity(int64_t nkeys) {
  uint64_t c = 64;
  while (c < 4 * (uint64_t)(nkeys > 0 ? nkeys : 1)) c <<= 1;
  return c;
}

__host__ __device__ constexpr int hash_slot_words(int len) { return len == 1 ? 2 : 4; }

// One 64-bit multiplication per key word (multiply-xorshift) instead of the two of the splitmix finaliser: the
// fused sample-space kernel hashes every x' (64.5 M per launch on Fe2S2) and is close to VALU-bound, a 64-bit
// multiplication is ~6 vector instructions.  The xor-shift brings the well-mixed high half of the product down to
// the bits the table index is taken from.
template <int LEN>
__device__ __forceinline__ uint64_t hash_of(const uint64_t (&q)[LEN]) {
  uint64_t h = q[0] * 0x9e3779b97f4a7c15ull;
#pragma unroll
  for (int w = 1; w < LEN; ++w) h = (h ^ (h >> 32) ^ q[w]) * 0xbf58476d1ce4e5b9ull;
  return h ^ (h >> 29);
}

// Bloom prefilter of the same keys (2 bits per key), appended to the table: the fused sample-space kernel copies it
// into LDS and asks it before anything else is done for a column.  Most x' are not in the sample table (85 % for
// the Fe2S2 CI space, far more for larger systems), so the filter decides what a column costs.  Its hash is a
// Zobrist hash -- the XOR of one fixed 32-bit value per occupied orbital -- because that one follows from the
// walker's hash and the 2 or 4 orbitals an excitation flips, without forming x' at all.
constexpr uint32_t kFilterMaxBits = 1u << 18;  // 32 KiB of LDS; also the most the 32-bit hash can address twice

__host__ __device__ inline uint32_t zobrist32(uint32_t orbital) {
  uint32_t z = (orbital + 1u) * 0x9e3779b1u;
  z ^= z >> 15; z *= 0x85ebca77u;
  z ^= z >> 13; z *= 0xc2b2ae3du;
  return z ^ (z >> 16);
}

// a second, independent value per orbital for the second-level filter
__host__ __device__ inline uint32_t zobrist32b(uint32_t orbital) {
  uint32_t z = (orbital + 0x51u) * 0x7feb352du;
  z ^= z >> 16; z *= 0x846ca68bu;
  z ^= z >> 15; z *= 0x9e3779b1u;
  return z ^ (z >> 13);
}

template <int LEN>
__host__ __device__ inline void zobrist_of(const uint64_t (&q)[LEN], uint32_t &z, uint32_t &z2) {
  z = 0; z2 = 0;
  for (int w = 0; w < LEN; ++w)
    for (uint64_t b = q[w]; b; b &= b - 1) {
      const uint32_t o = 64u * w + (uint32_t)__builtin_ctzll(b);
      z ^= zobrist32(o);
      z2 ^= zobrist32b(o);
    }
}

inline uint32_t hash_filter_bits(int64_t nkeys) {  // host side
  if (nkeys <= 0) return 0;
  // largest power of two <= 8 bits per key, in [1024, kFilterMaxBits] (PYNQS_FILTER_BITS lowers the cap, 0 = no filter).
  // Measured on Fe2S2 (18496 keys, 8192 walkers): 2 / 4 / 8 / 16 bits per key -> 0.345 / 0.298 / 0.271 / 0.294 ms
  // (false positives against LDS occupancy); no filter 0.447 ms.
  static const uint64_t maxbits = getenv("PYNQS_FILTER_BITS") ? strtoull(getenv("PYNQS_FILTER_BITS"), nullptr, 10) : kFilterMaxBits;
  if (maxbits == 0) return 0;
  const uint64_t cap = maxbits < kFilterMaxBits ? maxbits : kFilterMaxBits;
  uint64_t b = 1024;
  static const uint64_t per_key = getenv("PYNQS_FILTER_PER_KEY") ? strtoull(getenv("PYNQS_FILTER_PER_KEY"), nullptr, 10) : 8;
  while (2 * b <= per_key * (uint64_t)nkeys && 2 * b <= cap) b <<= 1;
  return b >= (uint64_t)nkeys ? (uint32_t)b : 0u;  // below one bit per key it rejects too little
}

// Second-level filter (stays in global memory, L2-resident): 32 bits per key on the second Zobrist hash, asked only
// for the columns that passed the LDS filter, 64 of them at a time.  With 2-8 bits per key the LDS filter lets 5-15 %
// of the columns through; at sorb 120 the integral gathers and table probes of those false positives were the
// kernel's HBM traffic.  Tables too large for an LDS filter (> 2^18 keys: less than one bit per key) use this one alone,
// asked for every column.
inline uint32_t hash_filter2_bits(int64_t nkeys) {  // host side
  static const bool off = getenv("PYNQS_FILTER_BITS") && strtoull(getenv("PYNQS_FILTER_BITS"), nullptr, 10) == 0;
  if (nkeys <= 0 || off) return 0;
  uint64_t b = 1u << 15;
  while (b < 32ull * (uint64_t)nkeys && b < (1u << 26)) b <<= 1;
  return (uint32_t)b;
}

// String filters (round 3): one Bloom filter over the ALPHA strings of the keys and one over their BETA strings (the Zobrist hash of the
// even / of the odd orbitals; two positions each), appended after the second-level filter.  An alpha single of x can only lead into the
// table -- alone or combined with any beta single -- if alpha(x) with that single applied is the alpha string of some key, so the
// alpha-beta class (71 % of Fe2S2's columns) shrinks to (passing alpha singles) x (passing beta singles) before any column is visited.
// At most 1/8 full: a string that is not in the table passes with probability < 2 %.
inline uint32_t hash_string_bits(int64_t nkeys) {  // host side; per spin
  if (nkeys <= 0) return 0;
  uint64_t b = 1u << 13;
  while (b < 16ull * (uint64_t)nkeys && b < (1u << 22)) b <<= 1;
  return (uint32_t)b;
}

template <int LEN>
__host__ __device__ inline void zobrist_strings(const uint64_t (&q)[LEN], uint32_t &za, uint32_t &zb) {
  za = 0; zb = 0;
  for (int w = 0; w < LEN; ++w)
    for (uint64_t b = q[w]; b; b &= b - 1) {
      const uint32_t o = 64u * w + (uint32_t)__builtin_ctzll(b);
      if (o & 1u) zb ^= zobrist32(o);
      else za ^= zobrist32(o);
    }
}

// The second-level filter is blocked: both bits of a key lie in ONE 32-bit word (word from the low bits of the hash, the
// two bit numbers from its top 10 bits), so a query is one load.  (At 32 bits per key the false-positive rate stays
// below 1 %; the LDS filter, with 2-8 bits per key, keeps two independent positions.)
__host__ __device__ inline void filter2_position(uint32_t z2, uint32_t f2bits, uint32_t &word, uint32_t &mask) {
  word = z2 & (f2bits / 32u - 1u);  // f2bits <= 2^26: bits 0..20
  mask = (1u << ((z2 >> 22) & 31u)) | (1u << (z2 >> 27));
}

// the two filter bits of a key: the low and the high log2(fbits) bits of its Zobrist hash (fbits = 2^k, 10 <= k <= 18)
__host__ __device__ inline void filter_positions(uint32_t z, uint32_t fbits, uint32_t &b0, uint32_t &b1) {
  b0 = z & (fbits - 1u);
  b1 = z >> (uint32_t)__builtin_clz(fbits - 1u);  // 32 - k
}

// ---- block index of a key table (kernels_keys_index.hip builds it, kernels_eloc_keys.hip's INDEXED form reads it)
constexpr int kIndexBlocks = 5;  // one more than the bits a double excitation changes: x and x' agree in a whole block

constexpr int kIndexTagShift = 40;  // sorted values carry their block number above the widest block (2 * ceil(96 / 5) = 40 bits)

// first bit of block b (b = kIndexBlocks: sorb).  Even, so that a block holds whole spatial orbitals and the alpha <-> beta exchange of
// the projected form maps a block onto itself.
__host__ __device__ inline int index_block_lo(int sorb, int b) { return 2 * ((b * (sorb / 2)) / kIndexBlocks); }

// bits [lo, hi) of a determinant (hi - lo <= 40)
template <int LEN>
__host__ __device__ inline uint64_t index_block_value(const uint64_t (&x)[LEN], int lo, int hi) {
  const int w = lo >> 6, sh = lo & 63;
  uint64_t a = 0, next = 0;  // (selected, not indexed: x lives in registers)
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    if (i == w) a = x[i];
    if (i == w + 1) next = x[i];
  }
  uint64_t v = a >> sh;
  if (sh) v |= next << (64 - sh);
  return v & ((1ull << (hi - lo)) - 1ull);
}

// First probe only: the slot content (to let a caller issue several independent first probes back to back).
template <int LEN>
struct HashProbe {
  uint64_t w[hash_slot_words(LEN)];
  uint64_t s;
};

template <int LEN>
__device__ __forceinline__ HashProbe<LEN> hash_probe_first_h(const uint64_t *__restrict__ table, uint64_t cap, uint64_t h) {
  constexpr int W = hash_slot_words(LEN);
  HashProbe<LEN> pr;
  pr.s = h & (cap - 1);
  const hash_u64x2 *slot = reinterpret_cast<const hash_u64x2 *>(table + pr.s * W);
#pragma unroll
  for (int i = 0; i < W / 2; ++i) { const hash_u64x2 v = slot[i]; pr.w[2 * i] = v[0]; pr.w[2 * i + 1] = v[1]; }
  return pr;
}

template <int LEN>
__device__ __forceinline__ HashProbe<LEN> hash_probe_first(const uint64_t *__restrict__ table, uint64_t cap,
                                                           const uint64_t (&q)[LEN]) {
  constexpr int W = hash_slot_words(LEN);
  HashProbe<LEN> pr;
  pr.s = hash_of<LEN>(q) & (cap - 1);
  const hash_u64x2 *slot = reinterpret_cast<const hash_u64x2 *>(table + pr.s * W);
#pragma unroll
  for (int i = 0; i < W / 2; ++i) { const hash_u64x2 v = slot[i]; pr.w[2 * i] = v[0]; pr.w[2 * i + 1] = v[1]; }
  return pr;
}

// Finish a lookup whose first slot has been read: almost always decided right away (load factor 1/4).
template <int LEN>
__device__ __forceinline__ int64_t hash_resolve(const HashProbe<LEN> &pr, const uint64_t *__restrict__ table, uint64_t cap,
                                                const uint64_t (&q)[LEN]) {
  constexpr int W = hash_slot_words(LEN);
  uint64_t w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = pr.w[i];
  uint64_t s = pr.s;
  for (uint64_t probes = 0; probes < cap; ++probes) {
    const int64_t idx = (int64_t)w[W - 1];
    if (idx < 0) return -1;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < LEN; ++i) eq = eq && w[i] == q[i];
    if (eq) return idx;
    s = (s + 1) & (cap - 1);
    const hash_u64x2 *slot = reinterpret_cast<const hash_u64x2 *>(table + s * W);
#pragma unroll
    for (int i = 0; i < W / 2; ++i) { const hash_u64x2 v = slot[i]; w[2 * i] = v[0]; w[2 * i + 1] = v[1]; }
  }
  return -1;
}

template <int LEN>
__device__ __forceinline__ int64_t hash_find(const uint64_t *__restrict__ table, uint64_t cap, const uint64_t (&q)[LEN]) {
  constexpr int W = hash_slot_words(LEN);
  uint64_t s = hash_of<LEN>(q) & (cap - 1);
  for (uint64_t probes = 0; probes < cap; ++probes) {  // bounded: terminates even on a full table
    const hash_u64x2 *slot = reinterpret_cast<const hash_u64x2 *>(table + s * W);
    uint64_t w[W];
#pragma unroll
    for (int i = 0; i < W / 2; ++i) { const hash_u64x2 v = slot[i]; w[2 * i] = v[0]; w[2 * i + 1] = v[1]; }
    const int64_t idx = (int64_t)w[W - 1];
    if (idx < 0) return -1;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < LEN; ++i) eq = eq && w[i] == q[i];
    if (eq) return idx;
    s = (s + 1) & (cap - 1);
  }
  return -1;
}

// ---- per-walker LDS state ------------------------------------------------------------------------
// merged[sorb]  : onstate.cpp:147-193 slot list (u8 orbitals)
// occv[nele]    : occupied orbitals in the order singles visit them (word ascending, bit 63 -> 0)
// occa[192]     : occupied orbitals ascending, zero-padded (the reference's olst, hamiltonian.cpp:38-39)
// tab[...]      : excitation tables, entry = orbX | orbY << 8 | parity << 16 | plan offset part << 17
// scratch       : kDiagTile elements of the integral dtype (diagonal-element terms), 8-byte aligned
// msk[...]      : (sorb <= 64 only) one 64-bit word per table entry: the entry's two orbital bits, XORed with
//                 the walker for the alpha-singles and hole-pair tables, so that a ket is msk[x] ^ msk[y]
struct LdsLayout {
  uint8_t *merged;
  uint8_t *occv;
  uint8_t *occa;
  uint32_t *tab;
  uint64_t *msk;
  unsigned char *scratch;
};

#ifndef PYNQS_DIAG_TILE
#define PYNQS_DIAG_TILE 2048
#endif
constexpr int kDiagTile = PYNQS_DIAG_TILE;

__host__ __device__ inline size_t lds_tab_bytes(const SDParams &p) {
  return (((size_t)p.tabEntries * 4 + 3 * 192) + 7) & ~(size_t)7;
}

__host__ __device__ inline size_t lds_fixed_bytes(const SDParams &p) {
  return lds_tab_bytes(p) + (p.sorb <= 64 ? (size_t)p.tabEntries * 8 : 0);
}

// with_diag_scratch: room for kDiagTile values of `elem` bytes after the fixed part
__host__ __device__ inline size_t lds_bytes(const SDParams &p, size_t elem = 0) {
  return lds_fixed_bytes(p) + elem * kDiagTile;
}

__device__ __forceinline__ LdsLayout carve_lds(unsigned char *base, const SDParams &p) {
  LdsLayout L;
  L.tab = reinterpret_cast<uint32_t *>(base);
  L.merged = base + (size_t)p.tabEntries * 4;
  L.occv = L.merged + 192;
  L.occa = L.occv + 192;
  L.msk = reinterpret_cast<uint64_t *>(base + lds_tab_bytes(p));
  L.scratch = base + lds_fixed_bytes(p);
  return L;
}

// triangular pair rank -> (hi > lo).  The reference uses int(sqrt(2(q+1)) + 0.5) in double
// (excitation.h:6-11); for q < 2^24 the float estimate corrected by one step gives the same integers.
__device__ __forceinline__ void pair_unrank(int q, int &hi, int &lo) {
  int i = (int)(sqrtf(2.0f * (float)(q + 1)) + 0.5f);
  // exact: i is the unique integer with i(i-1)/2 <= q < i(i+1)/2
  while (i * (i - 1) / 2 > q) --i;
  while (i * (i + 1) / 2 <= q) ++i;
  hi = i;
  lo = q - i * (i - 1) / 2;
}

// x -> the rank-r single / double excitation of x (excitation.cpp:43-109 incl. the global `idx % noAA` modulo), without LDS tables:
// for kernels that need ONE excitation per walker (spin_flip_rand, the GFMC move from a column index).
template <int LEN>
__device__ __forceinline__ void excite_by_rank(uint64_t (&x)[LEN], uint32_t r, const SDParams &p) {
  // (slot indices) -> orbitals: slot 2k (+1) = k-th alpha (beta) orbital of [occupied ascending | virtual ascending]
  auto orbital_of_slot = [&](int slot) -> int {
    const int beta = slot & 1;
    int k = slot >> 1;
    const uint64_t spin = beta ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
    int nocc = 0;
#pragma unroll
    for (int i = 0; i < LEN; ++i) nocc += __popcll(x[i] & spin);
    const bool want_occ = k < nocc;
    if (!want_occ) k -= nocc;
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      uint64_t bits = (want_occ ? x[i] : ~x[i]) & spin;
      if (i == LEN - 1 && (p.sorb & 63)) bits &= (1ull << (p.sorb & 63)) - 1ull;
      const int c = __popcll(bits);
      if (k < c) {
        for (int t = 0; t < k; ++t) bits &= bits - 1;
        return i * 64 + __builtin_ctzll(bits);
      }
      k -= c;
    }
    return 0;
  };
  int si, sa, sj = -1, sb = -1;
  if (r < p.d0) { si = 2 * (int)(r % p.noA); sa = 2 * (int)(r / p.noA + p.noA); }
  else if (r < p.d1) { const uint32_t t = r - p.d0; si = 2 * (int)(t % p.noB) + 1; sa = 2 * (int)(t / p.noB + p.noB) + 1; }
  else if (r < p.d3) {
    const bool beta = r >= p.d2;
    const uint32_t t = r - (beta ? p.d2 : p.d1);
    const int npair = beta ? p.noBB : p.noAA, no = beta ? p.noB : p.noA;
    int h1, h0, v1, v0;
    pair_unrank((int)(r % npair), h1, h0);  // global rank modulo: excitation.cpp:63,79
    pair_unrank((int)(t / npair), v1, v0);
    si = 2 * h1 + beta; sj = 2 * h0 + beta; sa = 2 * (v1 + no) + beta; sb = 2 * (v0 + no) + beta;
  } else {
    const uint32_t t = r - p.d3;
    const uint32_t ia = t % p.nSa, jb = t / p.nSa;
    si = 2 * (int)(ia % p.noA); sa = 2 * (int)(ia / p.noA + p.noA);
    sj = 2 * (int)(jb % p.noB) + 1; sb = 2 * (int)(jb / p.noB + p.noB) + 1;
  }
  const int oi = orbital_of_slot(si), oa = orbital_of_slot(sa);
  int oj = 0, ob = 0;
  if (sj >= 0) { oj = orbital_of_slot(sj); ob = orbital_of_slot(sb); }
  toggle<LEN>(x, oi); toggle<LEN>(x, oa);
  if (sj >= 0) { toggle<LEN>(x, oj); toggle<LEN>(x, ob); }
}

// Builds merged / occv / tables for the workgroup's walker.  All threads of the block must call it;
// ends with a barrier.  Returns the walker's electron count (length of occv).
template <int LEN>
// zorb (one-word determinants only): a per-orbital 32-bit value in LDS, written before the call; the msk area then
// receives, instead of the ket masks, the XOR of the two orbitals' values per table entry as uint32 (the
// filter-first sample-space kernel, kernels_eloc.hip).
__device__ __forceinline__ int build_walker_tables(const Walker<LEN> &wk, const SDParams &p, const LdsLayout &L,
                                                   const uint32_t *zorb = nullptr) {
  const int tid = threadIdx.x;
  const int sorb = p.sorb;
  // actual alpha / beta electron counts of this walker (the reference's slot counters run on the
  // determinant itself, not on noA/noB)
  int occA = 0, occB = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    occA += __popcll(wk.w[i] & 0x5555555555555555ull);
    occB += __popcll(wk.w[i] & 0xAAAAAAAAAAAAAAAAull);
  }
  for (int s = tid; s < 192; s += blockDim.x) L.occa[s] = 0;
  __syncthreads();
  for (int s = tid; s < sorb; s += blockDim.x) {
    const int word = s >> 6, b = s & 63;
    const uint64_t spin = (s & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
    const uint64_t below = (1ull << b) - 1ull;
    int rank_occ = 0, rank_all = 0;  // same-spin occupied / all same-spin orbitals below s
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      uint64_t m = i < word ? ~0ull : (i == word ? below : 0ull);
      rank_occ += __popcll(wk.w[i] & spin & m);
      rank_all += __popcll(spin & m);
    }
    const bool occ = (pick<LEN>(wk.w, word) >> b) & 1ull;
    const int nocc = (s & 1) ? occB : occA;
    const int r = occ ? rank_occ : nocc + (rank_all - rank_occ);
    L.merged[2 * r + (s & 1)] = (uint8_t)s;
    if (occ) {
      // visiting order of singles: words ascending, bits descending inside a word
      int before = 0, lower = 0;
#pragma unroll
      for (int i = 0; i < LEN; ++i) {
        if (i < word) { before += __popcll(wk.w[i]); lower += __popcll(wk.w[i]); }
        if (i == word) { before += __popcll(wk.w[i] & ~below & ~(1ull << b)); lower += __popcll(wk.w[i] & below); }
      }
      L.occv[before] = (uint8_t)s;
      L.occa[lower] = (uint8_t)s;
    }
  }
  __syncthreads();
  // singles tables: entry ia = a_idx*no + i_idx  ->  hole | particle << 8 | sign parity << 16
  for (int e = tid; e < p.nSa + p.nSb; e += blockDim.x) {
    const bool beta = e >= p.nSa;
    const int ia = beta ? e - p.nSa : e;
    const int no = beta ? p.noB : p.noA;
    const int a_idx = (int)mdiv((uint32_t)ia, beta ? p.divNoB : p.divNoA);
    const int i_idx = ia - a_idx * no;
    const int h = L.merged[2 * i_idx + beta];
    const int q = L.merged[2 * (a_idx + no) + beta];
    const uint32_t par = bit_of<LEN>(wk.pm, h) ^ bit_of<LEN>(wk.pm, q) ^ (uint32_t)(h < q);
    // bits 17..30: (particle spatial index) * K + (hole spatial index): the plan kernels' Vab offset parts
    const uint32_t sp = (uint32_t)(q >> 1) * (uint32_t)(p.sorb >> 1) + (uint32_t)(h >> 1);
    L.tab[(beta ? p.offSb : p.offSa) + ia] = (uint32_t)h | ((uint32_t)q << 8) | (par << 16) | (sp << 17);
    if constexpr (LEN == 1) {
      if (zorb) {
        reinterpret_cast<uint32_t *>(L.msk)[(beta ? p.offSb : p.offSa) + ia] = zorb[h] ^ zorb[q];
      } else {
        const uint64_t bits = (1ull << h) ^ (1ull << q);
        L.msk[(beta ? p.offSb : p.offSa) + ia] = beta ? bits : (bits ^ wk.w[0]);
      }
    }
  }
  // pair tables: hole pairs (hi > lo among occupied slots) and particle pairs (virtual slots)
  const int nPairs = p.noAA + p.nvAA + p.noBB + p.nvBB;
  for (int e = tid; e < nPairs; e += blockDim.x) {
    int q = e, off, base, beta, extra;
    if (q < p.noAA) { off = p.offHPa; base = 0; beta = 0; extra = 0; }
    else if ((q -= p.noAA) < p.nvAA) { off = p.offPPa; base = p.noA; beta = 0; extra = 1; }
    else if ((q -= p.nvAA) < p.noBB) { off = p.offHPb; base = 0; beta = 1; extra = 0; }
    else { q -= p.noBB; off = p.offPPb; base = p.noB; beta = 1; extra = 1; }
    int hi, lo;
    pair_unrank(q, hi, lo);
    const int o1 = L.merged[2 * (hi + base) + beta];
    const int o0 = L.merged[2 * (lo + base) + beta];
    // holes: P(p0)^P(p1); particles: P(q0)^P(q1)^1 (q1 < q0 is one of the flipped bits below q0)
    const uint32_t par = bit_of<LEN>(wk.pm, o1) ^ bit_of<LEN>(wk.pm, o0) ^ (uint32_t)extra;
    // bits 17..29: pair rank over the spatial orbitals of this spin (the plan kernels' Vss row / column)
    const uint32_t m1 = (uint32_t)o1 >> 1, m0 = (uint32_t)o0 >> 1;
    L.tab[off + q] = (uint32_t)o1 | ((uint32_t)o0 << 8) | (par << 16) | ((m1 * (m1 - 1) / 2 + m0) << 17);
    if constexpr (LEN == 1) {
      if (zorb) {
        reinterpret_cast<uint32_t *>(L.msk)[off + q] = zorb[o1] ^ zorb[o0];
      } else {
        const uint64_t bits = (1ull << o1) ^ (1ull << o0);
        L.msk[off + q] = extra ? bits : (bits ^ wk.w[0]);
      }
    }
  }
  __syncthreads();
  return occA + occB;
}

// One decoded excitation.
struct Excitation {
  int h0, h1;     // holes   (h0 > h1 for doubles; h1 unused for singles)
  int q0, q1;     // particles (q0 > q1 for doubles)
  uint32_t par;   // 1 -> matrix element gets a minus sign
  bool is_double;
};

// rank r in [0, nsd) -> excitation, using the LDS tables (excitation.cpp:43-109 incl. the quirk).
__device__ __forceinline__ Excitation decode(uint32_t r, const SDParams &p, const LdsLayout &L) {
  Excitation x;
  if (r < p.d1) {
    const uint32_t e = r < p.d0 ? L.tab[p.offSa + r] : L.tab[p.offSb + (r - p.d0)];
    x.h0 = e & 0xff; x.q0 = (e >> 8) & 0xff; x.h1 = x.q1 = 0;
    x.par = (e >> 16) & 1u;
    x.is_double = false;
    return x;
  }
  x.is_double = true;
  uint32_t eh, ep;
  if (r < p.d3) {
    const bool beta = r >= p.d2;
    const uint32_t t = r - (beta ? p.d2 : p.d1);
    const uint32_t npair = beta ? p.noBB : p.noAA;
    const uint32_t ab = mdiv(t, beta ? p.divNoBB : p.divNoAA);
    uint32_t ij = t - ab * npair + (beta ? p.rotB : p.rotA);  // == r % npair
    ij = ij >= npair ? ij - npair : ij;
    eh = L.tab[(beta ? p.offHPb : p.offHPa) + ij];
    ep = L.tab[(beta ? p.offPPb : p.offPPa) + ab];
    x.h0 = eh & 0xff; x.h1 = (eh >> 8) & 0xff;
    x.q0 = ep & 0xff; x.q1 = (ep >> 8) & 0xff;
    x.par = ((eh ^ ep) >> 16) & 1u;
    x.par ^= (uint32_t)(x.h0 < x.q0) ^ (uint32_t)(x.h1 < x.q0) ^ (uint32_t)(x.h0 < x.q1) ^ (uint32_t)(x.h1 < x.q1);
  } else {
    const uint32_t t = r - p.d3;
    const uint32_t jb = mdiv(t, p.divNSa);
    const uint32_t ia = t - jb * (uint32_t)p.nSa;
    eh = L.tab[p.offSa + ia];  // alpha hole/particle
    ep = L.tab[p.offSb + jb];  // beta hole/particle
    const int ha = eh & 0xff, qa = (eh >> 8) & 0xff, hb = ep & 0xff, qb = (ep >> 8) & 0xff;
    // entries carry P(h)^P(q)^[h<q] per spin; add the cross terms and the constant 1
    x.par = (((eh ^ ep) >> 16) & 1u) ^ (uint32_t)(ha < qb) ^ (uint32_t)(hb < qa) ^ 1u;
    x.h0 = ha > hb ? ha : hb; x.h1 = ha > hb ? hb : ha;
    x.q0 = qa > qb ? qa : qb; x.q1 = qa > qb ? qb : qa;
  }
  return x;
}

template <int LEN>
__device__ __forceinline__ void make_ket(const Walker<LEN> &wk, const Excitation &x, uint64_t (&ket)[LEN]) {
#pragma unroll
  for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
  toggle<LEN>(ket, x.h0);
  toggle<LEN>(ket, x.q0);
  if (x.is_double) {
    toggle<LEN>(ket, x.h1);
    toggle<LEN>(ket, x.q1);
  }
}

// <x|H|x'> for a decoded excitation, bit-identical to excitation.cpp:141-167 (same operation order).
template <typename T>
__device__ __forceinline__ T element(const Excitation &x, const SDParams &p, const LdsLayout &L, int nocc,
                                     const T *__restrict__ h1e, const T *__restrict__ h2e) {
  if (x.is_double) {
    const T v = h2e[tri_index(pair_index(x.h0, x.h1), pair_index(x.q0, x.q1))];
    return x.par ? -v : v;
  }
  const int hp = x.h0, q = x.q0;
  T acc = T(0);
  acc += h1e[(size_t)q * p.sorb + hp];
#pragma unroll 4
  for (int t = 0; t < nocc; ++t) {
    const int k = L.occv[t];
    acc += two_body<T>(h2e, hp, k, q, k);
  }
  return x.par ? -acc : acc;
}

// Diagonal element <x|H|x>, hamiltonian.cpp:34-50.  The reference adds nele(nele+1)/2 terms in a fixed
// order (for p ascending: h(p,p), then <pq||pq> for q < p ascending).  The whole workgroup gathers the
// terms into LDS (coalescing does not matter: they are L2 hits), then ONE lane (the last of the block)
// adds them in the reference's order, so the value is bit-identical; the other waves go on.
// Must be called by every thread of the block, after build_walker_tables.
template <typename T>
__device__ __forceinline__ void diag_phase(const SDParams &p, const LdsLayout &L, const T *__restrict__ h1e,
                                           const T *__restrict__ h2e, T *__restrict__ out) {
  T *tile = reinterpret_cast<T *>(L.scratch);
  const int tid = threadIdx.x;
  const int nele = p.nele;
  const int nterms = nele * (nele + 1) / 2;
  T acc = T(0);
  for (int base = 0; base < nterms; base += kDiagTile) {
    const int end = min(base + kDiagTile, nterms);
    if (base) __syncthreads();  // previous tile fully consumed
    for (int t = base + tid; t < end; t += blockDim.x) {
      int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > t) --a;
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      const int pos = t - a * (a + 1) / 2;
      const int pa = L.occa[a];
      T v;
      if (pos == 0) v = h1e[(size_t)pa * p.sorb + pa];
      else v = two_body<T>(h2e, pa, L.occa[pos - 1], pa, L.occa[pos - 1]);
      tile[t - base] = v;
    }
    __syncthreads();
    if (tid == (int)blockDim.x - 1)
      for (int t = 0; t < end - base; ++t) acc += tile[t];
  }
  if (tid == (int)blockDim.x - 1) *out = acc;
}

}  // namespace pynqs
