// kernels_eloc.hip -- fused local-energy kernels on the integral plan: the excitation list and its matrix
// elements are consumed on chip, comb / Hmat are never written to HBM.
//   eloc_sample_space_filtered_kernel : E_loc(x) = sum_x' <x|H|x'> psi(x') / psi(x), psi from the hash table of the sample
//                              space (vmc/energy/eloc.py:326-401 + utils/public_function.py:817-838); a Zobrist-hash filter
//                              decides per column before any other work, candidates are queued and evaluated 64 at a time
//   eloc_sample_space_kernel : the same sum with every column evaluated in place (tile scheduler + LookupSink): sorted
//                              keys (binary search), or a hash table built without filters
//   hash_build / hash_lookup : the table (the reference's optional GPU table, cuda_tensor.cpp:489-559) and its filters
//   reduce_count / reduce_emit : keep only |<x|H|x'>| >= eps (vmc/energy/eloc.py:297-298), compacted in a reproducible order
// Matrix elements of the doubles are bit-identical to kernels_plan.hip (same helpers); the sample-space kernels add the
// terms of <x|H|x> and of the singles in an order-free way (E_loc is a rounded sum, tolerance 1e-8 Ha).
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "plan_tiles.h"

namespace pynqs {

// -------------------------------------------------------------------------------------------------
// SAMPLE_SPACE local energy: the tile scheduler of the drop-in kernel (plan_tiles.h) with a sink that, instead
// of storing the column, looks psi(x') up and accumulates h * psi(x') in registers.  acc[walker] receives the
// sum; the workgroup that owns column 0 also stores psi(x).  One workgroup per (walker, chunk); with more than
// one chunk per walker partial sums meet through float atomics (the last bits then depend on arrival order).
// HASH: `keys` is a hash table built by pynqs_hash_build (nkeys = its capacity) instead of the sorted keys.
// Spin-flip partner of a determinant (vmc/energy/flip.py:322-418, utils/public_function.py:966-1007): alpha <-> beta occupations
// exchanged (orbitals 2k <-> 2k + 1 live in the same word) and the sign (-1)^(doubly occupied spatial orbitals) of x' itself.
template <int LEN>
__device__ __forceinline__ bool spin_flip_ket(uint64_t (&ket)[LEN]) {
  uint32_t pairs = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    const uint64_t w = ket[i];
    pairs += (uint32_t)__popcll(w & (w >> 1) & 0x5555555555555555ull);
    ket[i] = ((w >> 1) & 0x5555555555555555ull) | ((w & 0x5555555555555555ull) << 1);
  }
  return pairs & 1u;  // true: eta_m = -1
}

template <int LEN, bool CPLX, bool HASH>
struct LookupSink {
  const uint64_t *__restrict__ keys;
  int64_t nkeys;
  const double *__restrict__ wf;
  double *__restrict__ psi0;  // this walker's psi(x) slot
  double re, im;
  bool flip;  // wave-uniform: look flip(x') up and weight with eta_m(x') (the projected form's second sum); psi0 is not written
  // the two look-ups of a sink call: both first probes in flight together
  __device__ __forceinline__ void lookup2(uint32_t c0, double h0, const uint64_t (&k0)[LEN], uint32_t c1, double h1, const uint64_t (&k1)[LEN]) {
    if (flip) {
      add(c0, h0, k0);
      add(c1, h1, k1);
      return;
    }
    const HashProbe<LEN> p0 = hash_probe_first_h<LEN>(keys, (uint64_t)nkeys, hash_of<LEN>(k0));
    const HashProbe<LEN> p1 = hash_probe_first_h<LEN>(keys, (uint64_t)nkeys, hash_of<LEN>(k1));
    accumulate(c0, h0, hash_resolve<LEN>(p0, keys, (uint64_t)nkeys, k0));
    accumulate(c1, h1, hash_resolve<LEN>(p1, keys, (uint64_t)nkeys, k1));
  }
  __device__ __forceinline__ void add(uint32_t col, double h, const uint64_t (&ket)[LEN]) {
    uint64_t q[LEN];
#pragma unroll
    for (int i = 0; i < LEN; ++i) q[i] = ket[i];
    if (flip && spin_flip_ket<LEN>(q)) h = -h;
    int64_t pos;
    if constexpr (HASH) pos = hash_find<LEN>(keys, (uint64_t)nkeys, q);
    else pos = lut_find<LEN>(keys, nkeys, q);
    accumulate(col, h, pos);
  }
  __device__ __forceinline__ void accumulate(uint32_t col, double h, int64_t pos) {
    double vr = 0.0, vi = 0.0;
    if (pos >= 0) {
      if constexpr (CPLX) {  // one 16-byte load: the kernel is bound by the number of vector-memory instructions (TD busy 91 %)
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = *reinterpret_cast<const d2 *>(wf + 2 * pos);
        vr = v[0]; vi = v[1];
      } else vr = wf[pos];
    }
    re += h * vr;
    if constexpr (CPLX) im += h * vi;
    if (col == 0 && !flip) {
      psi0[0] = vr;
      if constexpr (CPLX) psi0[1] = vi;
    }
  }
  __device__ __forceinline__ void tile_begin(uint32_t) const {}
  __device__ __forceinline__ void one(uint32_t col, double h, const uint64_t (&ket)[LEN]) { add(col, h, ket); }
  __device__ __forceinline__ void two(uint32_t c0, double h0, const uint64_t (&k0)[LEN], uint32_t c1, double h1, const uint64_t (&k1)[LEN]) {
    if constexpr (HASH) lookup2(c0, h0, k0, c1, h1, k1);
    else {
      add(c0, h0, k0);
      add(c1, h1, k1);
    }
  }
  __device__ __forceinline__ void pair(uint32_t col, double h0, double h1, const uint64_t (&k0)[LEN], const uint64_t (&k1)[LEN]) {
    if constexpr (HASH) lookup2(col, h0, k0, col + 1, h1, k1);
    else {
      add(col, h0, k0);
      add(col + 1, h1, k1);
    }
  }
};

// Sum of a workgroup's per-lane partial sums into acc[walker]: lanes (xor butterfly), then waves, in a fixed order.
// (Which tile a wave gets is dynamic, so the order of the additions inside a lane, and with it the last bits of the
// sum, can vary from run to run; with more than one chunk per walker the chunks meet through float atomics.)
// With one chunk per walker the workgroup holds the whole sum and divides by psi(x) itself (psi0: written by this workgroup's
// column-0 lane before the barrier below, or the caller's input in the flip form); with several chunks eloc_divide_kernel follows.
template <bool CPLX, int NW>
__device__ __forceinline__ void store_walker_sum(double re, double im, double (*red)[NW], uint32_t nchunks, uint64_t walker,
                                                 double *__restrict__ acc, const double *psi0) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    re += __shfl_xor(re, o);
    if constexpr (CPLX) im += __shfl_xor(im, o);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = re; red[1][tid >> 6] = im; }
  __syncthreads();
  if (tid == 0) {
    double sr = 0.0, si = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { sr += red[0][w]; si += red[1][w]; }
    if (nchunks == 1) {
      // (another wave of this workgroup stored psi0 before the barrier above: read it at device scope, not through a line this CU
      // may still hold from before)
      if constexpr (CPLX) {
        const double br = __hip_atomic_load(psi0 + 2 * walker, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double bi = __hip_atomic_load(psi0 + 2 * walker + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), d = br * br + bi * bi;
        acc[2 * walker] = (sr * br + si * bi) / d;
        acc[2 * walker + 1] = (si * br - sr * bi) / d;
      } else acc[walker] = sr / __hip_atomic_load(psi0 + walker, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if constexpr (CPLX) { atomicAdd(acc + 2 * walker, sr); atomicAdd(acc + 2 * walker + 1, si); }
      else atomicAdd(acc + walker, sr);
    }
  }
}

template <int LEN, bool CPLX, bool HASH>
__global__ __launch_bounds__(kBlock) void eloc_sample_space_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                                   uint32_t nchunks, uint32_t chunk_len, bool xcd_map,
                                                                   const double *__restrict__ plan,
                                                                   const uint64_t *__restrict__ keys, int64_t nkeys,
                                                                   const double *__restrict__ wf, double *__restrict__ acc,
                                                                   double *__restrict__ psi0, bool flip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double red[2][kBlock / 64];
  __shared__ uint32_t next_tile;
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, xcd_map, walker, chunk);
  const int tid = threadIdx.x;
  if (tid == 0) next_tile = 0;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);  // ends with a barrier
  LookupSink<LEN, CPLX, HASH> sink{keys, nkeys, wf, psi0 + (CPLX ? 2 : 1) * walker, 0.0, 0.0, flip};
  visit_tiles<LEN, double, LookupSink<LEN, CPLX, HASH>, false>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
  store_walker_sum<CPLX, kBlock / 64>(sink.re, sink.im, red, nchunks, walker, acc, psi0);
}

// -------------------------------------------------------------------------------------------------
// SAMPLE_SPACE with the filter in front (hash table with a Bloom filter, i.e. every table of up to 2^18 keys).
// The kernel above does everything for every column -- decode, integral gather, sign, x', hash, probe -- although
// nearly all x' are not in the table: at sorb 120 that was 177 vector instructions and 9 vector-memory instructions
// per column (TD busy 95 %), 8 of the 9 being probe chains that lasted as long as the slowest lane's.  Here a
// column first costs only its decode and a filter test on the Zobrist hash  zx ^ Z[o1] ^ Z[o2] (^ Z[o3] ^ Z[o4])  of
// the orbitals it flips (detcore.h) -- no integral, no x'.  The ranks that pass are parked in a wave-private LDS
// queue, and whenever 64 are waiting the wave evaluates them together, all lanes busy: matrix element and x' from
// the rank (the same helpers as everywhere else), probe, accumulate.
// Two queues per kind (doubles, singles): ranks that passed the LDS filter wait for the second-level filter (global
// memory, its own Zobrist hash, 32 bits per key), the ones that pass that wait for their evaluation.  Singles are kept
// apart from doubles: one costs nele gathers, and a lane-per-candidate loop over them lasts as long for one single
// among 63 doubles as for 64 singles.
// Queue sizes: < 64 left over + what is parked between two pumps (one level: a pump after every group of 64 ranks).
__host__ __device__ constexpr uint32_t q1_doubles(bool two) { return two ? 320u : 128u; }
constexpr uint32_t kQ1Singles = 128, kQ2 = 128;
// bytes of a wave's queues: doubles are 32-bit entries, singles 16-bit ranks (there are fewer than 2^16 singles)
__host__ __device__ constexpr uint32_t queue_bytes(bool two) { return q1_doubles(two) * 4 + kQ1Singles * 2 + (two ? kQ2 * 4 + kQ2 * 2 : 0u); }
// LDS in front of the filter: the walker tables; for one-word determinants the ket-mask area (8 bytes per table entry)
// shrinks to the precombined Zobrist values (4 bytes per entry)
__host__ __device__ inline size_t filtered_fixed_lds(const SDParams &p) {
  return (lds_tab_bytes(p) + (p.sorb <= 64 ? (size_t)p.tabEntries * 4 : 0) + 15) & ~(size_t)15;
}
__host__ __device__ constexpr uint32_t z_bytes(int sorb) { return 4u * (((uint32_t)sorb + 15u) & ~15u); }  // one Z[orbital] table
// Workgroups of 256 threads, or of 512 / 1024 when tables + filter leave room for only one or two workgroups per CU
// (sorb >~ 100): the waves of a workgroup share them, so that multiplies the waves in flight (multi-word determinants
// only: one-word systems never get there).
constexpr int kBigBlock = 512;
__host__ __device__ constexpr size_t filtered_extra_lds(uint32_t fbits, int sorb, bool two, int block) {
  return fbits / 8 + (two ? 2 : 1) * z_bytes(sorb) + (size_t)(block / 64) * queue_bytes(two);
}

// TWO = false: no second level, queue 1 is evaluated directly (strong LDS filter, or many true hits: the second level
// then only adds work).
template <int LEN, bool CPLX, bool TWO>
struct Candidates {
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const SDParams &p;
  const PlanLayout &pl;
  const LdsLayout &L;
  const Walker<LEN> &wk;
  int nocc;
  const double *__restrict__ plan;
  const uint64_t *__restrict__ table;
  uint64_t cap;
  const double *__restrict__ wf;
  const uint32_t *__restrict__ filt2;  // second-level filter
  uint32_t f2bits;
  uint32_t filt, fbits;  // LDS address and size of the first filter
  uint32_t zorb;         // LDS address of Z[orbital] (z_bytes), followed by the second level's Z2[orbital] if TWO
  uint32_t ztab;         // one-word determinants: LDS address of the per-entry Z[o1] ^ Z[o2] (in place of the ket masks)
  uint32_t queue;        // LDS address of this wave's queues: doubles 1, singles 1, doubles 2, singles 2
  uint32_t zx, zx2;      // Zobrist hashes of the walker
  uint32_t n1[2], n2[2]; // entries of the queues, [0] doubles, [1] singles (wave-uniform)
  double re, im;
  bool flip;             // wave-uniform: the projected form's second sum (the Z tables then hold Z[orbital ^ 1], see the kernel)

  __device__ __forceinline__ uint32_t Z(uint32_t orbital_times_4) const { return *reinterpret_cast<lds_u32 *>(zorb + orbital_times_4); }
  // Zobrist hash of the orbitals of a table entry (orbital | orbital << 8 | ...)
  __device__ __forceinline__ uint32_t flipped(uint32_t e) const { return Z((e << 2) & 0x3fcu) ^ Z((e >> 6) & 0x3fcu); }
  __device__ __forceinline__ uint32_t flipped2(uint32_t e) const {
    const uint32_t z2 = z_bytes(p.sorb);
    return Z(z2 + ((e << 2) & 0x3fcu)) ^ Z(z2 + ((e >> 6) & 0x3fcu));
  }
  // ... or, for one-word determinants, straight from the precombined table (entry index as in L.tab)
  __device__ __forceinline__ uint32_t flipped_at(uint32_t index) const {
    if constexpr (LEN == 1) return *reinterpret_cast<lds_u32 *>(ztab + 4u * index);
    else return flipped(L.tab[index]);
  }
  // a double from its two table entries, without the ket-mask tables
  __device__ __forceinline__ double double_from_entries(uint32_t e0, uint32_t e1, bool opp, const double *__restrict__ V, uint32_t mask, uint32_t mul,
                                                        uint64_t (&ket)[LEN]) const {
    PendingDouble<double> d;
    d.e0 = e0;
    d.e1 = e1;
    if constexpr (LEN == 1) {
      d.k0 = wk.w[0] ^ (1ull << (e0 & 0xff)) ^ (1ull << ((e0 >> 8) & 0xff));
      d.k1 = (1ull << (e1 & 0xff)) ^ (1ull << ((e1 >> 8) & 0xff));
    }
    d.v = V[__umul24((e1 >> 17) & mask, mul) + ((e0 >> 17) & mask)];
    DoubleClass c;
    c.opposite = opp;
    return finish_double<LEN, double>(d, c, wk, ket);
  }
  // any double rank (one level: the queue holds plain ranks)
  __device__ __forceinline__ double double_from_rank(uint32_t r, uint64_t (&ket)[LEN]) const {
    const bool opp = r >= p.d3;
    const int spin = r >= p.d2;
    const DoubleClass c = opp ? make_opp_spin(p, pl) : make_same_spin(p, pl, spin);
    uint32_t slow, u;
    class_split(r, c, slow, u);
    uint32_t f = u + c.rot;
    f = f >= c.nfast ? f - c.nfast : f;
    const double *__restrict__ V = opp ? plan + pl.offVab : plan + pl.offVss + (size_t)spin * pl.NP * pl.NP;
    return double_from_entries(L.tab[c.off_fast + f], L.tab[c.off_slow + slow], opp, V, c.mask, c.mul, ket);
  }
  __device__ __forceinline__ bool maybe(uint32_t z) const {  // false: certainly not in the table
    if (!fbits) return maybe2(z);  // (wave-uniform) no LDS filter: the table is too large for one; Z[] then holds the second hash
    uint32_t b0, b1;
    filter_positions(z, fbits, b0, b1);
    const uint32_t w0 = *reinterpret_cast<lds_u32 *>(filt + 4u * (b0 >> 5)), w1 = *reinterpret_cast<lds_u32 *>(filt + 4u * (b1 >> 5));
    return ((w0 >> (b0 & 31u)) & (w1 >> (b1 & 31u)) & 1u) != 0u;
  }
  __device__ __forceinline__ bool maybe2(uint32_t z2) const {
    uint32_t word, mask;
    filter2_position(z2, f2bits, word, mask);
    return (filt2[word] & mask) == mask;
  }
  __device__ __forceinline__ void value(int64_t pos, double &vr, double &vi) const {  // psi of table entry pos, 0 if pos < 0
    vr = 0.0; vi = 0.0;
    if (pos >= 0) {
      if constexpr (CPLX) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = *reinterpret_cast<const d2 *>(wf + 2 * pos);
        vr = v[0]; vi = v[1];
      } else vr = wf[pos];
    }
  }
  __device__ __forceinline__ void add(double h, int64_t pos) {
    double vr, vi;
    value(pos, vr, vi);
    re += h * vr;
    if constexpr (CPLX) im += h * vi;
  }
  // entry i of a queue (layout of a wave's queues: doubles 1 | singles 1 | doubles 2 | singles 2)
  template <bool SINGLES, int STAGE>
  __device__ __forceinline__ uint32_t qaddr(uint32_t i) const {
    const uint32_t base = STAGE == 1 ? (SINGLES ? q1_doubles(TWO) * 4 : 0u) : q1_doubles(TWO) * 4 + kQ1Singles * 2 + (SINGLES ? kQ2 * 4 : 0u);
    return queue + base + (SINGLES ? 2u : 4u) * i;
  }
  template <bool SINGLES, int STAGE>
  __device__ __forceinline__ uint32_t qread(uint32_t i) const {
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    if constexpr (SINGLES) return *reinterpret_cast<lds_u16 *>(qaddr<SINGLES, STAGE>(i));
    else return *reinterpret_cast<lds_u32 *>(qaddr<SINGLES, STAGE>(i));
  }
  template <bool SINGLES, int STAGE>
  __device__ __forceinline__ void qwrite(uint32_t i, uint32_t r) const {
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    if constexpr (SINGLES) *reinterpret_cast<lds_u16 *>(qaddr<SINGLES, STAGE>(i)) = (uint16_t)r;
    else *reinterpret_cast<lds_u32 *>(qaddr<SINGLES, STAGE>(i)) = r;
  }
  // all 64 lanes active (so are all callers below); LDS operations of a wave execute in order
  template <bool SINGLES, int STAGE>
  __device__ __forceinline__ void park(uint32_t r, bool pass) {
    const uint64_t m = __ballot(pass);
    if (!m) return;
    uint32_t &count = STAGE == 1 ? n1[SINGLES] : n2[SINGLES];
    if (pass) {
      const uint32_t lane = threadIdx.x & 63;
      qwrite<SINGLES, STAGE>(count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), r);
    }
    count = __builtin_amdgcn_readfirstlane(count + (uint32_t)__popcll(m));
  }
  // With two levels a parked double is its class and its two table indices, class << 30 | slow << 15 | fast (class 0 / 1: same spin
  // alpha / beta, 2: opposite spin; both indices < 2^15 for sorb <= 192): no division and no class branches afterwards.
  static __device__ __forceinline__ uint32_t pack(int k, uint32_t slow, uint32_t f) { return ((uint32_t)k << 30) | (slow << 15) | f; }
  __device__ __forceinline__ void double_entries(uint32_t code, uint32_t &i0, uint32_t &i1) const {  // indices into L.tab / L.msk
    const uint32_t k = code >> 30;
    i0 = (uint32_t)(k == 0 ? p.offHPa : (k == 1 ? p.offHPb : p.offSa)) + (code & 0x7fffu);
    i1 = (uint32_t)(k == 0 ? p.offPPa : (k == 1 ? p.offPPb : p.offSb)) + ((code >> 15) & 0x7fffu);
  }
  __device__ __forceinline__ double double_value(uint32_t code, uint64_t (&ket)[LEN]) const {
    const uint32_t k = code >> 30;
    uint32_t i0, i1;
    double_entries(code, i0, i1);
    const uint32_t e0 = L.tab[i0], e1 = L.tab[i1];
    if (k == 2)  // (branches rather than selects: the table bases stay scalar)
      return double_from_entries(e0, e1, true, plan + pl.offVab, 0x7fffu, (uint32_t)(pl.K * pl.K), ket);
    return double_from_entries(e0, e1, false, plan + pl.offVss + (size_t)k * pl.NP * pl.NP, 0x1fffu, (uint32_t)pl.NP, ket);
  }
  // second-level filter for the top n <= 64 ranks of queue 1; survivors move to queue 2
  template <bool SINGLES>
  __device__ __forceinline__ void stage1(uint32_t n) {
    const uint32_t lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    uint32_t r = 0;
    bool pass = false;
    if (lane < n) {
      r = qread<SINGLES, 1>(n1[SINGLES] - n + lane);
      uint32_t z2 = zx2;
      if constexpr (SINGLES) z2 ^= flipped2(L.tab[p.offSa + r]);
      else {
        uint32_t i0, i1;
        double_entries(r, i0, i1);
        z2 ^= flipped2(L.tab[i0]) ^ flipped2(L.tab[i1]);
      }
      pass = maybe2(z2);
    }
    n1[SINGLES] = __builtin_amdgcn_readfirstlane(n1[SINGLES] - n);
    __builtin_amdgcn_wave_barrier();
    park<SINGLES, 2>(r, pass);
  }
  // evaluation of the top n <= 64 ranks of queue STAGE: matrix element, x', probe, accumulate
  template <bool SINGLES, int STAGE>
  __device__ __forceinline__ void evaluate(uint32_t n) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t &count = STAGE == 1 ? n1[SINGLES] : n2[SINGLES];
    __builtin_amdgcn_wave_barrier();
    double h = 0.0;
    int64_t pos = -1;
    if (lane < n) {
      const uint32_t r = qread<SINGLES, STAGE>(count - n + lane);
      uint64_t ket[LEN];
      if constexpr (SINGLES) {
        h = fast_single<double>(r, p, pl, L, nocc, plan);
        const uint32_t e = L.tab[p.offSa + r];
#pragma unroll
        for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
        toggle<LEN>(ket, e & 0xff);
        toggle<LEN>(ket, (e >> 8) & 0xff);
      } else {
        // (one level: the queue holds plain ranks -- the packed form costs this variant 16 more VGPRs and a wave per SIMD)
        if constexpr (TWO) h = double_value(r, ket);
        else h = double_from_rank(r, ket);
      }
      if (flip && spin_flip_ket<LEN>(ket)) h = -h;
      pos = hash_find<LEN>(table, cap, ket);
    }
    count = __builtin_amdgcn_readfirstlane(count - n);
    __builtin_amdgcn_wave_barrier();
    add(h, pos);
  }
  // work off whole batches of 64
  template <bool SINGLES>
  __device__ __forceinline__ void pump() {
    while (n1[SINGLES] >= 64u) {
      if constexpr (TWO) {
        stage1<SINGLES>(64u);
        if (n2[SINGLES] >= 64u) evaluate<SINGLES, 2>(64u);
      } else {
        evaluate<SINGLES, 1>(64u);
      }
    }
  }
  // ... and everything that is left
  template <bool SINGLES>
  __device__ __forceinline__ void flush() {
    pump<SINGLES>();
    if constexpr (TWO) {
      if (n1[SINGLES]) stage1<SINGLES>(n1[SINGLES]);
      while (n2[SINGLES]) evaluate<SINGLES, 2>(min(n2[SINGLES], 64u));
    } else {
      if (n1[SINGLES]) evaluate<SINGLES, 1>(n1[SINGLES]);
    }
  }
};

// SWEEP: the doubles are scanned in block sweeps (long rows) instead of rank by rank; the host picks by the alpha-beta class's row length
// PRE (round 3; rank-by-rank scan only): the alpha-beta class is visited as (alpha singles whose new alpha string is in the table) x
// (beta singles whose new beta string is) -- two short lists made per walker from the table's string filters (detcore.h) -- instead of
// all nSa x nSb pairs: Fe2S2's 5625 columns become ~850, nearly all of them hits.
__host__ __device__ inline size_t pre_lds_bytes(const SDParams &p) { return (8 + 2 * ((size_t)p.nSa + (size_t)p.nSb) + 15) & ~(size_t)15; }

template <int LEN, bool CPLX, bool TWO, int BLOCK, bool SWEEP, bool PRE = false>
__global__ __launch_bounds__(BLOCK) void eloc_sample_space_filtered_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                                            uint32_t nchunks, uint32_t chunk_len, bool xcd_map,
                                                                            const double *__restrict__ plan,
                                                                            const uint64_t *__restrict__ table, int64_t cap,
                                                                            const double *__restrict__ wf, double *__restrict__ acc,
                                                                            double *__restrict__ psi0, bool flip, uint32_t fbits, uint32_t f2bits,
                                                                            uint32_t sbits) {
  static_assert(!(PRE && SWEEP), "the string prefilter belongs to the rank-by-rank scan");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double red[2][BLOCK / 64];
  __shared__ uint32_t next_tile;
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, xcd_map, walker, chunk);
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) next_tile = 0;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  // after the walker tables (no staging scratch: order-free singles / diagonal): filter, Z[orbital], Z2[orbital], the waves' queues
  const uint32_t filt_off = (uint32_t)filtered_fixed_lds(p), z_off = filt_off + fbits / 8, q_off = z_off + (TWO ? 2 : 1) * z_bytes(p.sorb);
  const uint32_t *__restrict__ gf = reinterpret_cast<const uint32_t *>(table + (uint64_t)cap * hash_slot_words(LEN));
  {
    uint32_t *lf = reinterpret_cast<uint32_t *>(smem + filt_off);
    for (uint32_t i = tid; i < fbits / 32; i += BLOCK) lf[i] = gf[i];
    uint32_t *lz = reinterpret_cast<uint32_t *>(smem + z_off);
    // flip: the table is asked for flip(x'), whose Zobrist hash is the XOR of Z[o ^ 1] over the orbitals o of x' -- the same
    // scan with the partner orbital's value in every Z slot
    const uint32_t fx = flip ? 1u : 0u;
    if (tid < p.sorb) {
      lz[tid] = fbits ? zobrist32((uint32_t)tid ^ fx) : zobrist32b((uint32_t)tid ^ fx);
      if constexpr (TWO) lz[z_bytes(p.sorb) / 4 + tid] = zobrist32b((uint32_t)tid ^ fx);
    }
  }
  // (its first barrier publishes Z[] to the table loops; it ends with a barrier: tables and filter are visible)
  const int nocc = build_walker_tables<LEN>(wk, p, L, LEN == 1 ? reinterpret_cast<const uint32_t *>(smem + z_off) : nullptr);
  uint32_t zx = 0, zx2 = 0;
#pragma unroll
  for (int w = 0; w < LEN; ++w)
    if ((wk.w[w] >> lane) & 1ull) {
      const uint32_t o = (64u * w + (uint32_t)lane) ^ (flip ? 1u : 0u);
      zx2 ^= zobrist32b(o);
      zx ^= fbits ? zobrist32(o) : zobrist32b(o);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { zx ^= __shfl_xor(zx, o); zx2 ^= __shfl_xor(zx2, o); }
  const uint32_t dyn = __builtin_amdgcn_groupstaticsize();  // LDS address of smem[0]
  Candidates<LEN, CPLX, TWO> cand{p, pl, L, wk, nocc, plan, table, (uint64_t)cap, wf, gf + fbits / 32, f2bits, dyn + filt_off, fbits, dyn + z_off,
                             dyn + (uint32_t)lds_tab_bytes(p), dyn + q_off + (uint32_t)(tid >> 6) * queue_bytes(TWO), zx, zx2, {0u, 0u}, {0u, 0u}, 0.0, 0.0, flip};

  // ---- PRE: the two lists.  Wave 0 tests the alpha singles, wave 1 the beta singles (ballot compaction: ascending, reproducible)
  uint32_t nA = 0, nB = 0;
  const uint16_t *listA = nullptr, *listB = nullptr;
  if constexpr (PRE) {
    uint32_t zxa = 0;  // the alpha part of zx
#pragma unroll
    for (int w = 0; w < LEN; ++w)
      if (((wk.w[w] >> lane) & 1ull) && !(lane & 1)) zxa ^= zobrist32((64u * w + (uint32_t)lane) ^ (flip ? 1u : 0u));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zxa ^= __shfl_xor(zxa, o);
    uint32_t *pre = reinterpret_cast<uint32_t *>(smem + q_off + (BLOCK / 64) * queue_bytes(TWO));
    uint16_t *la = reinterpret_cast<uint16_t *>(pre + 2), *lb = la + p.nSa;
    const int wv = tid >> 6;
    if (wv < 2) {
      // flip: the table is asked for flip(x'), whose beta string is the alpha string of x' (in the partner orbitals' Zobrist values)
      const uint32_t *__restrict__ sf = gf + fbits / 32 + f2bits / 32 + ((wv == 0) != flip ? 0u : sbits / 32);
      const uint32_t n = wv == 0 ? (uint32_t)p.nSa : (uint32_t)p.nSb, off = wv == 0 ? (uint32_t)p.offSa : (uint32_t)p.offSb;
      const uint32_t zs = wv == 0 ? zxa : zx ^ zxa;
      uint16_t *list = wv == 0 ? la : lb;
      uint32_t cnt = 0;
      for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        bool pass = false;
        if (i < n) {
          uint32_t b0, b1;
          filter_positions(zs ^ cand.flipped_at(off + i), sbits, b0, b1);
          pass = ((sf[b0 >> 5] >> (b0 & 31u)) & (sf[b1 >> 5] >> (b1 & 31u)) & 1u) != 0u;
        }
        const uint64_t m = __ballot(pass);
        if (pass) list[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
        cnt += (uint32_t)__popcll(m);
      }
      if (lane == 0) pre[wv] = cnt;
    }
    __syncthreads();
    nA = pre[0]; nB = pre[1];
    listA = la; listB = lb;
  }

  // tiles: 0 = column 0; 1 = this workgroup's share of the singles (blocks of 64 dealt round-robin over the walker's
  // workgroups, as in plan_tiles.h) -- one wave takes them all, so that its singles queue fills; then 256 ranks of one
  // class of doubles each
  // Doubles: block sweeps.  A class is the index space rank = b0 + slow * nfast + u; a wave takes a block of RB whole rows
  // (RB * nfast <= kScanCols consecutive ranks; rows wider than that are cut into position blocks, RB = 1) and keeps its lanes
  // FIXED to positions of the block: the Zobrist value of the fast entry, its table index and the row offset are loaded once
  // per work item; per row block a lane reads the Z value of its row, XORs, asks the filter and parks.  No division, no
  // fast-table read and no rank arithmetic per column (the rank-by-rank scan spent 16 of its ~42 vector instructions per
  // 64 columns on stepping (slow, u) by 64 ranks, and 2 of its 4 LDS reads on the fast entry).
  constexpr int kScanPasses = 6;
  constexpr uint32_t kScanCols = 64u * kScanPasses;
  struct ScanClass {
    uint32_t t0, t1, row0, nrows, nrb, nfb, RB, W, G, ipf, nitems;
  };
  const uint32_t ncomb = p.nsd + 1;
  const uint32_t lo = chunk * chunk_len, hi = min(lo + chunk_len, ncomb);
  const uint32_t rlo = lo == 0 ? 0 : lo - 1, rhi = hi - 1;
  auto scan_class = [&](uint32_t b0, uint32_t b1, uint32_t nfast, const MagicDiv &dv) {
    ScanClass c;
    const uint32_t a0 = max(rlo, b0), a1 = max(a0, min(rhi, b1));
    c.t0 = a0 - b0; c.t1 = a1 - b0;
    c.row0 = 0; c.nrows = 0;
    if (a1 > a0) {
      c.row0 = mdiv(c.t0, dv);
      c.nrows = mdiv(c.t1 - 1u, dv) - c.row0 + 1u;
    }
    if (nfast > kScanCols) { c.nfb = (nfast + kScanCols - 1) / kScanCols; c.RB = 1; c.W = kScanCols; }
    else { c.nfb = 1; c.RB = nfast ? kScanCols / nfast : 1u; c.W = c.RB * nfast; }
    c.nrb = (c.nrows + c.RB - 1u) / c.RB;
    c.G = min(8u, max(1u, (c.nrb + 7u) / 8u));        // row blocks per work item: ~8 items per class and position block keep the
                                                       // workgroup's waves balanced, up to 8 row blocks share one lane set-up
    c.ipf = (c.nrb + c.G - 1u) / c.G;
    c.nitems = c.ipf * c.nfb;
    return c;
  };
  const ScanClass gA = scan_class(p.d1, p.d2, (uint32_t)p.noAA, p.divNoAA), gB = scan_class(p.d2, p.d3, (uint32_t)p.noBB, p.divNoBB),
                  gO = scan_class(p.d3, p.nsd, (uint32_t)p.nSa, p.divNSa);
  // Short rows (nfast < kSweepMin: Fe2S2 has 75 and 105) keep the rank-by-rank scan, 256 consecutive ranks per tile: a block of
  // several short rows needs the row's Z value per lane again and few, large work items -- measured 0.215 against 0.187 ms for
  // Fe2S2, while sorb 120 / 184 (nfast 435-2116) gain 20-23 % from the sweeps (18.9 -> 15.1 ms, 13.8 -> 10.6 ms).
  constexpr uint32_t kRanks = 256;
  const ClassRange rA = class_range<false>(p.d1, p.d2, rlo, rhi, 0u), rB = class_range<false>(p.d2, p.d3, rlo, rhi, 0u),
                   rO = class_range<false>(p.d3, p.nsd, rlo, rhi, 0u);
  constexpr bool swA = SWEEP, swB = SWEEP, swO = SWEEP;  // one mode per launch: two scan loops in one kernel cost Fe2S2 10 %
  const uint32_t tA = swA ? gA.nitems : (rA.npairs + kRanks - 1) / kRanks, tB = swB ? gB.nitems : (rB.npairs + kRanks - 1) / kRanks,
                 tO = PRE ? (rO.npairs ? (nA * nB + kRanks - 1) / kRanks : 0u) : (swO ? gO.nitems : (rO.npairs + kRanks - 1) / kRanks);
  const uint32_t ntiles = 2 + tA + tB + tO;
  for (;;) {
    uint32_t tile = 0;
    if (lane == 0) tile = atomicAdd(&next_tile, 1u);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (tile >= ntiles) break;
    if (tile == 0) {
      if (lo == 0) {  // column 0: <x|H|x> psi(x), and psi(x) itself
        const double v = fast_diag<double>(p, pl, L, plan);
        if (lane == 0) {
          double vr, vi;
          uint64_t q[LEN];
#pragma unroll
          for (int i = 0; i < LEN; ++i) q[i] = wk.w[i];
          double hv = v;
          if (flip && spin_flip_ket<LEN>(q)) hv = -hv;
          cand.value(hash_find<LEN>(table, (uint64_t)cap, q), vr, vi);
          cand.re += hv * vr;
          if constexpr (CPLX) cand.im += hv * vi;
          if (!flip) {
            double *__restrict__ out = psi0 + (CPLX ? 2 : 1) * walker;
            out[0] = vr;
            if constexpr (CPLX) out[1] = vi;
          }
        }
      }
      continue;
    }
    if (tile == 1) {
      for (uint32_t r0 = chunk * 64u; r0 < p.d1; r0 += nchunks * 64u) {
        const uint32_t r = r0 + (uint32_t)lane;
        cand.template park<true, 1>(r, (r < p.d1) & cand.maybe(zx ^ cand.flipped_at((uint32_t)p.offSa + min(r, p.d1 - 1))));
        cand.template pump<true>();
      }
      cand.template flush<true>();
      continue;
    }
    tile -= 2;
    const int k = tile < tA ? 0 : (tile < tA + tB ? 1 : 2);
    const uint32_t item = tile - (k == 0 ? 0u : (k == 1 ? tA : tA + tB));
    const DoubleClass c = k == 2 ? make_opp_spin(p, pl) : make_same_spin(p, pl, k);
    const uint32_t nslow = (uint32_t)(k == 0 ? p.nvAA : (k == 1 ? p.nvBB : p.nSb));
    if (PRE && k == 2) {
      // ---- the alpha-beta class from the two lists: pair t = (t / nA)-th passing beta single x (t % nA)-th passing alpha single
      const uint32_t npair = nA * nB, first = item * kRanks;
      const uint32_t q64 = 64u / nA, r64 = 64u - q64 * nA;
      uint32_t i = (first + (uint32_t)lane) / nA, j = first + (uint32_t)lane - i * nA;
      uint32_t rr[4], ff[4], ss[4];
      bool pass[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t m = first + 64u * g4 + (uint32_t)lane;
        const uint32_t slow = listB[min(i, nB - 1u)], f = listA[j];
        ff[g4] = f; ss[g4] = slow;
        const uint32_t r = p.d3 + slow * (uint32_t)p.nSa + f;  // (no rotation in this class)
        rr[g4] = TWO ? cand.pack(2, slow, f) : r;
        const uint32_t z = zx ^ cand.flipped_at((uint32_t)p.offSa + f) ^ cand.flipped_at((uint32_t)p.offSb + slow);
        pass[g4] = (m < npair) & (r >= rlo) & (r < rhi) & cand.maybe(z);
        j += r64;
        i += q64 + (j >= nA ? 1u : 0u);
        j = j >= nA ? j - nA : j;
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        // Most pairs of the two lists are in the table when it is a product of string sets (Fe2S2's CI space: all of them): a group
        // with at least half of its lanes passing is evaluated where it stands -- table entries at hand, no queue, no rank decoding
        // (0.145 -> 0.142 ms: the evaluation itself -- hash, probe, integral and psi gathers -- is what costs); sparser groups wait in the
        // queue for company, as everywhere else.
        const uint64_t pm = __ballot(pass[g4]);
        if (!TWO && __popcll(pm) >= 32) {
          double h = 0.0;
          int64_t pos = -1;
          if (pass[g4]) {
            uint64_t ket[LEN];
            h = cand.double_from_entries(L.tab[(uint32_t)p.offSa + ff[g4]], L.tab[(uint32_t)p.offSb + ss[g4]], true, plan + pl.offVab, 0x7fffu,
                                         (uint32_t)(pl.K * pl.K), ket);
            if (flip && spin_flip_ket<LEN>(ket)) h = -h;
            pos = hash_find<LEN>(table, (uint64_t)cap, ket);
          }
          cand.add(h, pos);
          continue;
        }
        cand.template park<false, 1>(rr[g4], pass[g4]);
        if constexpr (!TWO) cand.template pump<false>();
      }
      if constexpr (TWO) cand.template pump<false>();
    } else if constexpr (!SWEEP) {
      // ---- rank-by-rank scan of 256 consecutive ranks: rank -> (slow, fast) by one division for the lane's first rank, then
      // 64 further per group
      const ClassRange g = k == 0 ? rA : (k == 1 ? rB : rO);
      const uint32_t first = item * kRanks;
      const uint32_t q64 = mdiv(64u, c.dv), r64 = 64u - q64 * c.nfast;
      const uint32_t last = g.npairs - 1;
      uint32_t slow, u;
      class_split(g.r_e + min(first + (uint32_t)lane, last), c, slow, u);
      uint32_t rr[4];
      bool pass[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // independent and branch-free: the LDS reads of the four groups overlap
        const uint32_t m = first + 64u * j + (uint32_t)lane;
        uint32_t f = u + c.rot;
        f = f >= c.nfast ? f - c.nfast : f;
        rr[j] = TWO ? cand.pack(k, slow, f) : g.r_e + m;
        // (lanes past the end of the class read a valid but meaningless entry: slow is clamped to the class's last row)
        const uint32_t z = zx ^ cand.flipped_at(c.off_fast + f) ^ cand.flipped_at(c.off_slow + min(slow, nslow - 1u));
        pass[j] = (m <= last) & cand.maybe(z);
        u += r64;
        slow += q64 + (u >= c.nfast ? 1u : 0u);
        u = u >= c.nfast ? u - c.nfast : u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cand.template park<false, 1>(rr[j], pass[j]);
        if constexpr (!TWO) cand.template pump<false>();
      }
      if constexpr (TWO) cand.template pump<false>();
    } else {
    // ---- block sweep (long rows)
    const ScanClass g = k == 0 ? gA : (k == 1 ? gB : gO);
    const uint32_t fb = item / g.ipf, grp = item - fb * g.ipf;
    const uint32_t rb0 = grp * g.G, rb1 = min(rb0 + g.G, g.nrb);
    const uint32_t nfast = c.nfast, RB = g.RB;
    const uint32_t ub = fb * kScanCols;                                 // first position of this position block
    const uint32_t W = RB > 1 ? g.W : min(kScanCols, nfast - ub);       // positions per row block
    const int npass = (int)((W + 63u) >> 6);                            // wave-uniform
    // lane-fixed part
    uint32_t zf[kScanPasses], code0[kScanPasses], rj[kScanPasses];
    bool okq[kScanPasses];
#pragma unroll
    for (int j = 0; j < kScanPasses; ++j) {
      const uint32_t q = 64u * j + (uint32_t)lane;
      okq[j] = q < W;
      const uint32_t qq = okq[j] ? q : 0u;
      rj[j] = RB > 1 ? mdiv(qq, c.dv) : 0u;
      const uint32_t u = ub + qq - rj[j] * nfast;
      uint32_t f = u + c.rot;
      f = f >= nfast ? f - nfast : f;
      zf[j] = zx ^ cand.flipped_at(c.off_fast + f);
      code0[j] = TWO ? cand.pack(k, 0u, f) : c.b0 + u;  // parked entry: class | slow << 15 | fast-table index, or the rank
    }
    for (uint32_t rb = rb0; rb < rb1; ++rb) {
      const uint32_t srow = g.row0 + rb * RB;
      const uint32_t flat0 = srow * nfast + ub;  // flat position of the block's first column (wave-uniform)
      const bool interior = flat0 >= g.t0 && flat0 + W <= g.t1 && srow + RB <= g.row0 + g.nrows;  // wave-uniform
      uint32_t zs1 = 0;
      if (RB == 1) zs1 = cand.flipped_at(c.off_slow + min(srow, nslow - 1u));  // one row: the same value for every pass
#pragma unroll
      for (int j = 0; j < kScanPasses; ++j) {
        if (j >= npass) break;
        const uint32_t row = srow + rj[j];
        const uint32_t zs = RB == 1 ? zs1 : cand.flipped_at(c.off_slow + min(row, nslow - 1u));
        bool valid = okq[j];
        if (!interior) {
          const uint32_t flat = flat0 + 64u * j + (uint32_t)lane;  // rows of a block are consecutive ranks
          valid = valid && row < g.row0 + g.nrows && flat >= g.t0 && flat < g.t1;
        }
        const bool pass = valid & cand.maybe(zf[j] ^ zs);
        cand.template park<false, 1>(TWO ? (code0[j] | (row << 15)) : code0[j] + row * nfast, pass);
        if constexpr (!TWO) cand.template pump<false>();
        else if ((j & 3) == 3) cand.template pump<false>();  // queue 1 holds 320 entries: < 64 left over + four passes
      }
      if constexpr (TWO) cand.template pump<false>();
    }
    }
  }
  cand.template flush<false>();
  store_walker_sum<CPLX, BLOCK / 64>(cand.re, cand.im, red, nchunks, walker, acc, psi0);
}

// the string filters exist beside an LDS filter only (they share its Zobrist values)
static inline uint32_t hash_string_bits_if(int64_t nkeys) { return hash_filter_bits(nkeys) ? hash_string_bits(nkeys) : 0u; }

// Insert key i of the sorted key array: claim a slot by CAS on its index word, then write the key words
// (lookups only start after the build kernel has finished).
template <int LEN>
__global__ __launch_bounds__(kBlock) void hash_build_kernel(const uint64_t *__restrict__ keys, int64_t nkeys, uint64_t cap,
                                                            uint64_t *__restrict__ table, uint32_t fbits, uint32_t f2bits, uint32_t sbits) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nkeys) return;
  uint64_t q[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) q[w] = keys[i * LEN + w];
  constexpr int W = hash_slot_words(LEN);
  const uint64_t hq = hash_of<LEN>(q);
  uint64_t s = hq & (cap - 1);
  for (uint64_t probes = 0; probes < cap; ++probes) {
    unsigned long long *idxp = reinterpret_cast<unsigned long long *>(table + s * W + (W - 1));
    const unsigned long long old = atomicCAS(idxp, ~0ull, (unsigned long long)i);
    if (old == ~0ull) {
#pragma unroll
      for (int w = 0; w < LEN; ++w) table[s * W + w] = q[w];
      if (fbits || f2bits) {
        uint32_t z, z2, b0, b1;
        zobrist_of<LEN>(q, z, z2);
        uint32_t *filter = reinterpret_cast<uint32_t *>(table + cap * W);
        if (fbits) {
          filter_positions(z, fbits, b0, b1);
          atomicOr(filter + (b0 >> 5), 1u << (b0 & 31u));
          atomicOr(filter + (b1 >> 5), 1u << (b1 & 31u));
          filter += fbits / 32;
        }
        if (f2bits) {  // second level
          filter2_position(z2, f2bits, b0, b1);
          atomicOr(filter + b0, b1);
          filter += f2bits / 32;
        }
        if (sbits) {  // the key's alpha and beta strings
          uint32_t za, zb;
          zobrist_strings<LEN>(q, za, zb);
          filter_positions(za, sbits, b0, b1);
          atomicOr(filter + (b0 >> 5), 1u << (b0 & 31u));
          atomicOr(filter + (b1 >> 5), 1u << (b1 & 31u));
          filter += sbits / 32;
          filter_positions(zb, sbits, b0, b1);
          atomicOr(filter + (b0 >> 5), 1u << (b0 & 31u));
          atomicOr(filter + (b1 >> 5), 1u << (b1 & 31u));
        }
      }
      return;
    }
    s = (s + 1) & (cap - 1);
  }
}

template <int LEN>
__global__ __launch_bounds__(kBlock) void hash_lookup_kernel(const uint64_t *__restrict__ table, uint64_t cap,
                                                             const uint64_t *__restrict__ onv, uint64_t n,
                                                             int64_t *__restrict__ idx, uint8_t *__restrict__ mask) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t q[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) q[w] = onv[i * LEN + w];
  const int64_t r = hash_find<LEN>(table, cap, q);
  idx[i] = r;
  mask[i] = r >= 0;
}

// eloc = acc / psi0 (complex division when CPLX), in place on acc
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void eloc_divide_kernel(double *__restrict__ acc, const double *__restrict__ psi0, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if constexpr (CPLX) {
    const double ar = acc[2 * i], ai = acc[2 * i + 1], br = psi0[2 * i], bi = psi0[2 * i + 1];
    const double d = br * br + bi * bi;
    acc[2 * i] = (ar * br + ai * bi) / d;
    acc[2 * i + 1] = (ai * br - ar * bi) / d;
  } else {
    acc[i] = acc[i] / psi0[i];
  }
}

// -------------------------------------------------------------------------------------------------
// REDUCE front end: keep |h| >= eps (vmc/energy/eloc.py:297-298).  Two passes over the tile scheduler of the drop-in
// kernel (plan_tiles.h), one workgroup per (walker, chunk), no workgroup barrier after the table build, no atomics:
//   count: tile_counts[walker][chunk][tile] = kept columns of that tile (a wave owns a tile and visits its columns
//          in a fixed order; the running count lives in a wave-private LDS word because some columns are produced
//          inside divergent code)
//   emit : the caller turns the counts into exclusive offsets; the wave writes its tile's records from there.
// Records of a walker are therefore contiguous and in a reproducible order (tile by tile: diagonal and odd columns,
// singles, the three classes of doubles), not in ascending column order: kept_col says which column each one is.
template <int LEN, typename T, bool EMIT>
struct ReduceSink {
  T eps;
  volatile uint32_t *run;              // this wave's running count inside the current tile (LDS)
  uint32_t *__restrict__ tile_counts;  // count pass: this workgroup's slice
  const int64_t *__restrict__ tile_off;  // emit pass: this workgroup's slice
  int32_t *__restrict__ kept_col;
  uint64_t *__restrict__ kept_onv;
  T *__restrict__ kept_h;
  uint32_t tile;    // current tile (0xffffffff: none)
  int64_t base;     // emit: first record of the current tile

  __device__ __forceinline__ void flush() {
    if constexpr (!EMIT) {
      if (tile != 0xffffffffu && (threadIdx.x & 63) == 0) tile_counts[tile] = *run;
    }
  }
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush();
    tile = t;
    if ((threadIdx.x & 63) == 0) *run = 0;
    if constexpr (EMIT) base = tile_off[t];
  }
  // emit pass: a tile that keeps nothing (the next offset equals this one) is not enumerated again (plan_tiles.h).
  // `tiles_left` = entries of the offset array from this workgroup's slice to its end.
  uint64_t tiles_left;
  __device__ __forceinline__ bool skip_tile(uint32_t t) const {
    if constexpr (EMIT) return (uint64_t)t + 1 < tiles_left && tile_off[t + 1] == tile_off[t];
    else return false;
  }
  __device__ __forceinline__ void put(int64_t pos, uint32_t col, T h, const uint64_t (&ket)[LEN]) const {
    kept_col[pos] = (int32_t)col;
    kept_h[pos] = h;
#pragma unroll
    for (int i = 0; i < LEN; ++i) kept_onv[pos * LEN + i] = ket[i];
  }
  // adds `total` to the wave's running count and returns its previous value to all ACTIVE lanes
  __device__ __forceinline__ uint32_t advance(uint32_t total) const {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)__ballot(1)) - 1;
    uint32_t before = 0;
    if (lane == leader) { before = *run; *run = before + total; }
    return __shfl(before, leader);
  }
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&ket)[LEN]) const {
    const bool k = fabs(h) >= eps;
    const uint64_t m = __ballot(k);
    if (!m) return;
    const uint32_t before = advance((uint32_t)__popcll(m));
    if constexpr (EMIT) {
      const int lane = threadIdx.x & 63;
      if (k) put(base + before + __popcll(m & ((1ull << lane) - 1ull)), col, h, ket);
    }
  }
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&k0)[LEN], uint32_t c1, T h1, const uint64_t (&k1)[LEN]) const {
    const bool a = fabs(h0) >= eps, b = fabs(h1) >= eps;
    const uint64_t ma = __ballot(a), mb = __ballot(b);
    if (!(ma | mb)) return;
    const uint32_t before = advance((uint32_t)(__popcll(ma) + __popcll(mb)));
    if constexpr (EMIT) {
      const int lane = threadIdx.x & 63;
      const uint64_t below = (1ull << lane) - 1ull;
      const int64_t mine = base + before + __popcll(ma & below) + __popcll(mb & below);
      if (a) put(mine, c0, h0, k0);
      if (b) put(mine + (a ? 1 : 0), c1, h1, k1);
    }
  }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&k0)[LEN], const uint64_t (&k1)[LEN]) const {
    two(col, h0, k0, col + 1, h1, k1);
  }
};

template <int LEN, typename T, bool EMIT>
__global__ __launch_bounds__(kBlock) void reduce_tiles_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                              uint32_t nchunks, uint32_t chunk_len, uint32_t max_tiles, bool xcd_map,
                                                              const T *__restrict__ plan, T eps, uint32_t *__restrict__ tile_counts,
                                                              const int64_t *__restrict__ tile_off, int32_t *__restrict__ kept_col,
                                                              uint64_t *__restrict__ kept_onv, T *__restrict__ kept_h) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t wave_run[kBlock / 64];
  __shared__ uint32_t next_tile;
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, xcd_map, walker, chunk);
  const uint64_t slot = walker * nchunks + chunk;  // position of this (walker, chunk) in the per-tile arrays
  const int tid = threadIdx.x;
  if (tid == 0) next_tile = 0;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  ReduceSink<LEN, T, EMIT> sink{eps, wave_run + (tid >> 6), EMIT ? nullptr : tile_counts + slot * max_tiles,
                                EMIT ? tile_off + slot * max_tiles : nullptr, kept_col, kept_onv, kept_h, 0xffffffffu, 0,
                                ((uint64_t)gridDim.x - slot) * max_tiles};
  visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
  sink.flush();
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

static int eloc_common_checks(int sorb, int nele, int noA, int noB, int64_t nbatch, SDParams *p, PlanLayout *pl) {
  if (!make_sd_params(sorb, nele, noA, noB, p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "bad nbatch");
  return PYNQS_OK;
}

static int eloc_sample_space_impl(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                  const uint64_t *keys, int64_t nkeys, bool hash, const double *wf, int wf_is_complex,
                                  double *eloc, double *psi0, void *stream, bool flip = false) {
  SDParams p;
  PlanLayout pl;
  int rc = eloc_common_checks(sorb, nele, noA, noB, nbatch, &p, &pl);
  if (rc != PYNQS_OK) return rc;
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !eloc || !psi0 || nkeys < 0 || (nkeys > 0 && (!keys || !wf))) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, p.nsd + 1, &nchunks, &chunk_len);
  const uint32_t fbits = hash ? hash_filter_bits(nkeys) : 0u;
  // second-level filter when the LDS one has fewer than 6 bits per key (it then lets > 8 % of the columns through);
  // PYNQS_FILTER2=0/1 forces it off / on
  static const int f2env = getenv("PYNQS_FILTER2") ? atoi(getenv("PYNQS_FILTER2")) : -1;
  const uint32_t f2bits = hash ? hash_filter2_bits(nkeys) : 0u;
  const bool filtered = fbits || f2bits;
  const bool two_level = fbits && f2bits && (f2env >= 0 ? f2env != 0 : (uint64_t)fbits < 6ull * (uint64_t)nkeys);
  const size_t lds_fixed = filtered ? filtered_fixed_lds(p) : (lds_bytes(p, 0) + 15) & ~(size_t)15;
  // workgroup size: whatever puts the most waves on a CU (160 KiB of LDS; the kernel's registers allow 7 waves per SIMD);
  // the waves of a workgroup share tables and filter.  One-word systems never need more than 256 threads.
  static const int blk_env = getenv("PYNQS_SS_BLOCK") ? atoi(getenv("PYNQS_SS_BLOCK")) : 0;
  int block = kBlock;
  if (filtered && len > 1) {
    if (blk_env == 256 || blk_env == 512 || blk_env == 1024) block = blk_env;
    else {
      size_t best = 0;
      for (int b = kBlock; b <= 1024; b *= 2) {
        const size_t need = lds_fixed + filtered_extra_lds(fbits, sorb, two_level, b) + 256;
        size_t waves = (160 * 1024 / need) * (size_t)(b / 64);
        if (waves > 28) waves = 28;
        if (waves > best) { best = waves; block = b; }
      }
    }
  }
  // block sweeps once the rows of the alpha-beta class (nSa positions) are long: sorb 120 / 184 gain 20-23 %, Fe2S2 (75) would lose 12 %
  static const int sweep_env = getenv("PYNQS_SS_SWEEP") ? atoi(getenv("PYNQS_SS_SWEEP")) : -1;
  const bool sweep = sweep_env >= 0 ? sweep_env != 0 : p.nSa >= 256;
  // string prefilter of the alpha-beta class: the rank-by-rank scan in 256-thread workgroups (PYNQS_SS_PRE=0: off)
  static const bool pre_env = !(getenv("PYNQS_SS_PRE") && atoi(getenv("PYNQS_SS_PRE")) == 0);
  const uint32_t sbits = hash && fbits ? hash_string_bits(nkeys) : 0u;
  const bool pre = pre_env && filtered && sbits && !sweep && block == kBlock && p.nSa > 0 && p.nSb > 0;
  const size_t lds = lds_fixed + (filtered ? filtered_extra_lds(fbits, sorb, two_level, block) : 0) + (pre ? pre_lds_bytes(p) : 0);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  const size_t esz = wf_is_complex ? 16 : 8;
  if (nchunks > 1 && hipMemsetAsync(eloc, 0, esz * (size_t)nbatch, st) != hipSuccess) return check_launch("memset");
  const double *pd = (const double *)plan;
  const int64_t size_arg = hash ? (int64_t)hash_capacity(nkeys) : nkeys;
#define PYNQS_SS_ARGS dim3((uint32_t)grid), dim3(block), lds, st, bra, p, pl, nchunks, chunk_len, xcd_mapping(nchunks), pd, keys, size_arg, wf, eloc, psi0, flip
#define PYNQS_SS_LAUNCH(KERNEL, ...)                                                                                              \
  do {                                                                                                                            \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds) != hipSuccess)                                                           \
      return check_launch("hipFuncSetAttribute");                                                                                \
    hipLaunchKernelGGL((KERNEL), PYNQS_SS_ARGS, ##__VA_ARGS__);                                                                   \
  } while (0)
#define PYNQS_SS_FILTERED2(B, SW, PR)                                                                                              \
  do {                                                                                                                             \
    if (two_level) {                                                                                                               \
      if (wf_is_complex) PYNQS_SS_LAUNCH((eloc_sample_space_filtered_kernel<LEN, true, true, B, SW, PR>), fbits, f2bits, sbits);    \
      else PYNQS_SS_LAUNCH((eloc_sample_space_filtered_kernel<LEN, false, true, B, SW, PR>), fbits, f2bits, sbits);                \
    } else {                                                                                                                       \
      if (wf_is_complex) PYNQS_SS_LAUNCH((eloc_sample_space_filtered_kernel<LEN, true, false, B, SW, PR>), fbits, f2bits, sbits);   \
      else PYNQS_SS_LAUNCH((eloc_sample_space_filtered_kernel<LEN, false, false, B, SW, PR>), fbits, f2bits, sbits);               \
    }                                                                                                                              \
  } while (0)
#define PYNQS_SS_FILTERED(B) do { if (sweep) PYNQS_SS_FILTERED2(B, true, false); else PYNQS_SS_FILTERED2(B, false, false); } while (0)
#define PYNQS_SS_FILTERED_PRE() PYNQS_SS_FILTERED2(kBlock, false, true)
  DISPATCH_LEN(len, {
    if (filtered) {  // hash table with its filters
      if (pre) {
        PYNQS_SS_FILTERED_PRE();
      } else if constexpr (LEN >= 2) {
        if (block == 1024) PYNQS_SS_FILTERED(1024);
        else if (block == kBigBlock) PYNQS_SS_FILTERED(kBigBlock);
        else PYNQS_SS_FILTERED(kBlock);
      } else {
        PYNQS_SS_FILTERED(kBlock);
      }
    } else if (wf_is_complex) {
      if (hash) PYNQS_SS_LAUNCH((eloc_sample_space_kernel<LEN, true, true>)); else PYNQS_SS_LAUNCH((eloc_sample_space_kernel<LEN, true, false>));
    } else {
      if (hash) PYNQS_SS_LAUNCH((eloc_sample_space_kernel<LEN, false, true>)); else PYNQS_SS_LAUNCH((eloc_sample_space_kernel<LEN, false, false>));
    }
  });
#undef PYNQS_SS_FILTERED_PRE
#undef PYNQS_SS_FILTERED
#undef PYNQS_SS_FILTERED2
#undef PYNQS_SS_ARGS
#undef PYNQS_SS_LAUNCH
  if (nchunks > 1) {  // (one chunk per walker: the kernel divided by psi(x) itself)
    const uint32_t g2 = (uint32_t)((nbatch + kBlock - 1) / kBlock);
    if (wf_is_complex) hipLaunchKernelGGL((eloc_divide_kernel<true>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
    else hipLaunchKernelGGL((eloc_divide_kernel<false>), dim3(g2), dim3(kBlock), 0, st, eloc, psi0, nbatch);
  }
  return check_launch("eloc_sample_space");
}

extern "C" int pynqs_eloc_sample_space(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                       const void *plan, const uint64_t *keys, int64_t nkeys, const double *wf,
                                       int wf_is_complex, double *eloc, double *psi0, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return eloc_sample_space_impl(bra, nbatch, sorb, nele, noA, noB, plan, keys, nkeys, false, wf, wf_is_complex, eloc, psi0, stream);
}

extern "C" int pynqs_eloc_sample_space_hash(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                            const void *plan, const void *table, int64_t nkeys, const double *wf,
                                            int wf_is_complex, double *eloc, double *psi0, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return eloc_sample_space_impl(bra, nbatch, sorb, nele, noA, noB, plan, (const uint64_t *)table, nkeys, true, wf, wf_is_complex,
                                eloc, psi0, stream);
}

// The second sum of the spin-projected SAMPLE_SPACE local energy (vmc/energy/flip.py:322-418):
//   out[x] = sum_x' <x|H|x'> eta_m(x') psi(flip(x')) / psi0[x],   psi0 = psi(x) as returned by the calls above (an INPUT here)
// so that E_loc = (eloc + eta * out) / extra_norm^2.  Same kernels, same filters: the Zobrist tables hold the partner orbitals' values.
extern "C" int pynqs_eloc_sample_space_flip(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                            const void *plan, const uint64_t *keys, int64_t nkeys, const double *wf,
                                            int wf_is_complex, const double *psi0, double *out, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return eloc_sample_space_impl(bra, nbatch, sorb, nele, noA, noB, plan, keys, nkeys, false, wf, wf_is_complex, out, (double *)psi0, stream, true);
}

extern "C" int pynqs_eloc_sample_space_hash_flip(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                                 const void *plan, const void *table, int64_t nkeys, const double *wf,
                                                 int wf_is_complex, const double *psi0, double *out, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return eloc_sample_space_impl(bra, nbatch, sorb, nele, noA, noB, plan, (const uint64_t *)table, nkeys, true, wf, wf_is_complex, out,
                                (double *)psi0, stream, true);
}

extern "C" int64_t pynqs_hash_bytes(int64_t nkeys, int sorb) {
  if (nkeys < 0 || sorb < 1 || sorb > kMaxSorb) return -1;
  const int len = (sorb - 1) / 64 + 1;
  return (int64_t)(hash_capacity(nkeys) * (uint64_t)hash_slot_words(len) * 8 + hash_filter_bits(nkeys) / 8 + hash_filter2_bits(nkeys) / 8 +
                   2 * (size_t)hash_string_bits_if(nkeys) / 8);
}

extern "C" int pynqs_hash_build(const uint64_t *keys, int64_t nkeys, int sorb, void *table, void *stream) {
  pynqs::DeviceScope device_scope_(keys);
  if (nkeys < 0 || sorb < 1 || sorb > kMaxSorb) return set_error(PYNQS_EINVAL, "bad nkeys/sorb");
  if (!table || (nkeys > 0 && !keys)) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t cap = hash_capacity(nkeys);
  hipStream_t st = (hipStream_t)stream;
  if ((uintptr_t)table & 15u) return set_error(PYNQS_EINVAL, "table must be 16-byte aligned");
  const size_t slot_bytes = cap * (size_t)hash_slot_words(len) * 8;
  const uint32_t fbits = hash_filter_bits(nkeys);
  if (hipMemsetAsync(table, 0xFF, slot_bytes, st) != hipSuccess) return check_launch("hash memset");
  const uint32_t f2bits = hash_filter2_bits(nkeys), sbits = hash_string_bits_if(nkeys);
  if ((fbits || f2bits) && hipMemsetAsync((char *)table + slot_bytes, 0, fbits / 8 + f2bits / 8 + 2 * (size_t)sbits / 8, st) != hipSuccess)
    return check_launch("filter memset");
  if (nkeys == 0) return PYNQS_OK;
  const uint32_t grid = (uint32_t)((nkeys + kBlock - 1) / kBlock);
  DISPATCH_LEN(len, hipLaunchKernelGGL((hash_build_kernel<LEN>), dim3(grid), dim3(kBlock), 0, st, keys, nkeys, cap, (uint64_t *)table, fbits, f2bits, sbits));
  return check_launch("hash_build");
}

extern "C" int pynqs_hash_lookup(const void *table, int64_t nkeys, const uint64_t *onv, int64_t n, int sorb, int64_t *idx,
                                 uint8_t *mask, void *stream) {
  pynqs::DeviceScope device_scope_(table);
  if (nkeys < 0 || n < 0 || sorb < 1 || sorb > kMaxSorb) return set_error(PYNQS_EINVAL, "bad nkeys/n/sorb");
  if (n == 0) return PYNQS_OK;
  if (!table || !onv || !idx || !mask) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t grid = ((uint64_t)n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_LEN(len, hipLaunchKernelGGL((hash_lookup_kernel<LEN>), dim3((uint32_t)grid), dim3(kBlock), 0, st, (const uint64_t *)table,
                                       hash_capacity(nkeys), onv, (uint64_t)n, idx, mask));
  return check_launch("hash_lookup");
}

static int reduce_geometry(int64_t nbatch, const SDParams &p, uint32_t *nchunks, uint32_t *chunk_len, uint32_t *max_tiles) {
  plan_chunks(nbatch, p.nsd + 1, nchunks, chunk_len);
  *max_tiles = max_tiles_per_chunk(p, *nchunks, *chunk_len);
  return 0;
}

extern "C" int64_t pynqs_reduce_tiles(int64_t nbatch, int sorb, int nele, int noA, int noB) {
  SDParams p;
  if (nbatch < 0 || !make_sd_params(sorb, nele, noA, noB, &p)) return -1;
  uint32_t nchunks, chunk_len, max_tiles;
  reduce_geometry(nbatch, p, &nchunks, &chunk_len, &max_tiles);
  return (int64_t)nchunks * max_tiles;
}

template <bool EMIT>
static int launch_reduce(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan, int dtype,
                         double eps, uint32_t *tile_counts, const int64_t *tile_off, int32_t *kept_col, uint64_t *kept_onv,
                         void *kept_h, void *stream) {
  SDParams p;
  PlanLayout pl;
  int rc = eloc_common_checks(sorb, nele, noA, noB, nbatch, &p, &pl);
  if (rc != PYNQS_OK) return rc;
  if (dtype != PYNQS_F32 && dtype != PYNQS_F64) return set_error(PYNQS_EINVAL, "bad dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  if (EMIT ? (!tile_off || !kept_col || !kept_onv || !kept_h) : !tile_counts) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  uint32_t nchunks, chunk_len, max_tiles;
  reduce_geometry(nbatch, p, &nchunks, &chunk_len, &max_tiles);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  const size_t lds = lds_bytes(p, esz);
  // tiles a workgroup does not have keep the count 0
  if (!EMIT && hipMemsetAsync(tile_counts, 0, 4 * (size_t)grid * max_tiles, st) != hipSuccess) return check_launch("memset");
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64)
      hipLaunchKernelGGL((reduce_tiles_kernel<LEN, double, EMIT>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, max_tiles, xcd_mapping(nchunks), (const double *)plan, eps, tile_counts, tile_off, kept_col, kept_onv,
                         (double *)kept_h);
    else
      hipLaunchKernelGGL((reduce_tiles_kernel<LEN, float, EMIT>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks,
                         chunk_len, max_tiles, xcd_mapping(nchunks), (const float *)plan, (float)eps, tile_counts, tile_off, kept_col,
                         kept_onv, (float *)kept_h);
  });
  return check_launch(EMIT ? "reduce_emit" : "reduce_count");
}

extern "C" int pynqs_reduce_count(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                  int dtype, double eps, uint32_t *tile_counts, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return launch_reduce<false>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, tile_counts, nullptr, nullptr, nullptr, nullptr,
                              stream);
}

extern "C" int pynqs_reduce_emit(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan,
                                 int dtype, double eps, const int64_t *tile_offsets, int32_t *kept_col, uint64_t *kept_onv,
                                 void *kept_h, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  return launch_reduce<true>(bra, nbatch, sorb, nele, noA, noB, plan, dtype, eps, nullptr, tile_offsets, kept_col, kept_onv, kept_h,
                             stream);
}
