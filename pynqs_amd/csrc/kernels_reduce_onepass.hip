// kernels_reduce_onepass.hip -- the REDUCE front end of the local energy in ONE launch: everything between the walkers and
// the ansatz' forward (vmc/energy/eloc.py:205-324 _reduce_psi, `Func` of vmc/energy/flip.py:29-63, onv_to_tensor).
//
// Round 2 ran this as count -> host read-back -> emit (-> count_sums -> torch.multinomial -> draw) -> unique_insert ->
// unique_first -> onv_to_pm1 plus ~80 small torch kernels, and enumerated every row two or three times.  Here one workgroup
// owns a (walker, chunk) and
//   phase A  visits every column once on the tile scheduler (plan_tiles.h).  Kept columns (|H| >= eps) become records:
//            column 0, the singles and the few unpaired doubles have a FIXED slot each (they are produced by heavy tiles
//            that must not hold anybody up); the kept doubles of a tile wait in the wave's LDS scratch until the tile is
//            finished, get their place by a decoupled look-back over the tiles' counts in LDS (a wave never waits for more
//            than the COUNT of an earlier tile, which is published before anything else), and are written in tile order:
//            no atomics on the output position, no second pass, reproducible.  Sub-eps |H| are summed per tile into LDS.
//   phase B  (eps_sample > 0) scans the tile sums, draws the N uniforms of the reference's torch.multinomial over the tiles
//            (counter-based generator) and turns the per-tile draw counts into slot offsets -- all in LDS.
//   phase C  re-visits only the tiles that received draws and draws inside them (the machinery of kernels_reduce_sample.hip).
// Every record's determinant is looked up in the wave-function table (if given), else inserted into a de-duplication table;
// the winner of a slot takes the next row of the distinct list and the wave writes its +1/-1 row (the ansatz' input).
// A second, small kernel (reduce_contract_kernel) forms E_loc from the records and the amplitudes of the distinct rows.
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "plan_tiles.h"

namespace pynqs {

// -DPYNQS_OP_STAMPS: the LIST kernel notes wall_clock64() at its phase boundaries per workgroup (tools/onepass_stamps.py reads them through
// pynqs_debug_stamps); nothing of this exists in the product build
#ifdef PYNQS_OP_STAMPS
__device__ unsigned long long g_stamps[8192][16];  // 0-9: phase boundaries; 10-13: the flushing form's sums over its rounds
#define PYNQS_STAMP(k) do { if (threadIdx.x == 0 && walker < 8192) g_stamps[walker][k] = wall_clock64(); } while (0)
// (the flushing form: time spent between two marks, summed over the rounds, into stamp slot k)
#define PYNQS_STAMP_MARK() unsigned long long stamp_mark_ = wall_clock64()
#define PYNQS_STAMP_ADD(k) do { const unsigned long long now_ = wall_clock64(); if (threadIdx.x == 0 && walker < 8192) g_stamps[walker][k] += now_ - stamp_mark_; stamp_mark_ = now_; } while (0)
#define PYNQS_STAMP_ZERO(k) do { if (threadIdx.x == 0 && walker < 8192) g_stamps[walker][k] = 0; } while (0)
#else
#define PYNQS_STAMP(k) do { } while (0)
#define PYNQS_STAMP_MARK() do { } while (0)
#define PYNQS_STAMP_ADD(k) do { } while (0)
#define PYNQS_STAMP_ZERO(k) do { } while (0)
#endif

constexpr uint32_t kStatP = 0x80000000u;  // look-back status: inclusive prefix available
constexpr uint32_t kStatA = 0x40000000u;  //                   this tile's count available
constexpr uint32_t kStatMask = 0x3fffffffu;
constexpr int kFixedHead = 8;             // slot 0: column 0; slots 1..6: unpaired doubles; 7: unused
constexpr int kOneTileCols = 128 * PYNQS_U;
constexpr uint32_t kMaxKeptPerTile = kOneTileCols;  // columns of the largest tile (a tile of doubles; singles come 16 per tile, tile 0 has <= 7)
constexpr uint32_t kProbeLimit = 512;     // a de-duplication table at most half full never needs that many
constexpr int32_t kDirectLink = 1 << 30;  // link >= kDirectLink: row of the distinct list = link - kDirectLink (no look at the de-duplication slot)

__device__ __forceinline__ uint64_t op_mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ double op_scan(double v, int lane) {  // inclusive, lanes 0.. contiguous
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ __forceinline__ uint32_t op_wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__device__ __forceinline__ double op_wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// ---- outputs (device pointers, by value) -------------------------------------------------------------------------
template <typename T>
struct OnepassOut {
  int32_t *rec_col;
  T *rec_w;
  uint64_t *rec_onv;
  int32_t *rec_link;
  int32_t *seg_count;
  int32_t *srec_col;
  T *srec_w;
  uint64_t *srec_onv;
  int32_t *srec_link;
  double *row_sum;
  uint64_t *dedup;
  uint32_t dedup_mask;
  const uint64_t *lut;
  uint64_t lut_cap;
  uint64_t *uniq_onv;
  void *uniq_pm1;
  int pm1_f32;
  uint32_t ucap;
  int32_t *counters;
  uint32_t cap_d, fixed;
  const uint64_t *seed_dev;
  T *row_cache;    // [nbatch][ncomb] or NULL
  int32_t *uniq_parent;  // [ucap] or NULL: the walker whose record put the row on the distinct list (x' is a single / double excitation of it)
  int32_t parent;        // this workgroup's walker (set by the kernel)
  uint32_t debug;  // PYNQS_OP_DEBUG ablations (timing only): 1 no amplitude source, 2 no +-1 rows, 4 no look-back, 8 no in-tile draws,
                   // 16 no phase C, 32 phase A only; row-cache form: 64 no tile draws, 128 no search inside a tile, 256 no emission.
                   // Fe2S2, 8192 walkers, 1000 draws (round 3, row-cache form, no +-1 rows): 794 us = enumeration 184 + row cache
                   // written 80 + kept list sorted and resolved 66 + tile sums, tile-level draws, scans 117 + draws inside the tiles
                   // 209 (search 10, emission and resolution of the drawn records 75) + de-duplication 146
  unsigned char *tile_scratch = nullptr;  // GTILE: per walker [max_tiles] f64 tile sums + [max_tiles] u32 draw counts in global memory
  uint32_t tile_stride = 0;               // bytes per walker of tile_scratch
};

// ---- de-duplication table ----------------------------------------------------------------------------------------
// One-word determinants: slot = {key, row | ...}: the key word itself is claimed by a 64-bit CAS (empty = all ones, which no
// determinant with an excitation left can be).  Two / three words: slot = {state | row << 32, key words...}; the state word
// goes EMPTY -> BUSY (CAS) -> READY (after the key words are stored); a reader that meets BUSY polls again -- the owner never
// waits for anybody, and the loop re-converges every iteration, so lanes of one wave cannot starve each other.
// Every access to the table is an agent-scope atomic (coherent across the XCDs' L2s); the row number is written by the
// winner with a plain store and only read by later kernels.
__host__ __device__ constexpr int dedup_slot_words(int len) { return len == 1 ? 2 : 4; }
constexpr uint32_t kSlotEmpty = 0xffffffffu, kSlotBusy = 1u, kSlotReady = 2u;

template <int LEN>
__device__ __forceinline__ uint32_t dedup_insert(uint64_t *__restrict__ tab, uint32_t mask, const uint64_t (&q)[LEN], bool &won) {
  constexpr int W = dedup_slot_words(LEN);
  uint32_t s = (uint32_t)(hash_of<LEN>(q) >> 17) & mask;
  won = false;
  if constexpr (LEN == 1) {
    for (uint32_t probes = 0; probes < kProbeLimit; ++probes) {
      unsigned long long *kp = reinterpret_cast<unsigned long long *>(tab + (size_t)s * W);
      unsigned long long cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == ~0ull) {
        cur = atomicCAS(kp, ~0ull, (unsigned long long)q[0]);
        if (cur == ~0ull) { won = true; return s; }
      }
      if (cur == q[0]) return s;
      s = (s + 1) & mask;
    }
    return 0xffffffffu;
  } else {
    uint32_t probes = 0, polls = 0;
    while (probes < kProbeLimit) {
      uint32_t *sp = reinterpret_cast<uint32_t *>(tab + (size_t)s * W);
      uint32_t st = __hip_atomic_load(sp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (st == kSlotEmpty) {
        uint32_t expect = kSlotEmpty;
        if (__hip_atomic_compare_exchange_strong(sp, &expect, kSlotBusy, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
#pragma unroll
          for (int w = 0; w < LEN; ++w) __hip_atomic_store(tab + (size_t)s * W + 1 + w, q[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp, kSlotReady, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          won = true;
          return s;
        }
        st = expect;
      }
      if (st != kSlotReady) {  // somebody is writing the key: look again (bounded: the owner finishes within its own iteration)
        if (++polls > (1u << 20)) return 0xffffffffu;
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      bool eq = true;
#pragma unroll
      for (int w = 0; w < LEN; ++w)
        eq = eq && __hip_atomic_load(tab + (size_t)s * W + 1 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == q[w];
      if (eq) return s;
      s = (s + 1) & mask;
      ++probes;
    }
    return 0xffffffffu;
  }
}

// the row number of a slot: int32 at this offset (in int32 units) of the slot
__host__ __device__ constexpr int dedup_row_offset(int len) { return len == 1 ? 2 : 1; }

// Rows of the distinct list are handed out per WORKGROUP, not per determinant: a global counter that every new determinant
// increments is one address for 10^5 - 10^6 atomics per launch, and same-address atomics retire at ~4 ns each (measured: the first
// version of this kernel spent 1.26 ms on 245 k of them, 6.3 ms on 1.47 M).  The lane that wins a de-duplication slot only notes
// (slot, column) in an LDS list of its workgroup; at the end of a phase the workgroup takes ONE block of rows from the global counter
// and its waves write the slots' row numbers, the determinants and the +-1 rows.  A full list falls back to one atomic per wave and
// flush step.
struct WinnerList {
  uint32_t *n;      // LDS counter
  uint32_t *slot;   // [cap]
  uint32_t *col;    // [cap]
  uint32_t cap;
};

// Where psi(x') will come from: the wave-function table (link <= -2), or the distinct list through a de-duplication slot
// (link >= 0).  `unlisted`: this lane inserted a new determinant and the workgroup's list was full: the caller allocates its row.
template <int LEN, typename T>
__device__ __forceinline__ int32_t resolve_amplitude(const OnepassOut<T> &o, const WinnerList &wl, const uint64_t (&ket)[LEN], uint32_t col,
                                                     bool &unlisted) {
  unlisted = false;
  if (o.debug & 1u) return -1;
  if (o.lut) {
    const int64_t pos = hash_find<LEN>(o.lut, o.lut_cap, ket);
    if (pos >= 0) return (int32_t)(-2 - pos);
  }
  bool w;
  const uint32_t s = dedup_insert<LEN>(o.dedup, o.dedup_mask, ket, w);
  if (s == 0xffffffffu) {
    atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 2u);
    return -1;
  }
  if (w) {
    const uint32_t k = atomicAdd(wl.n, 1u);
    if (k < wl.cap) { wl.slot[k] = s; wl.col[k] = col; }
    else unlisted = true;
  }
  return (int32_t)s;
}

// row `r` of the distinct list belongs to the determinant in de-duplication slot `s`
template <int LEN, typename T>
__device__ __forceinline__ bool assign_row(const OnepassOut<T> &o, uint32_t s, int32_t r, const uint64_t (&ket)[LEN]) {
  if ((uint32_t)r >= o.ucap) {
    atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 4u);
    return false;
  }
  // (agent scope: other workgroups, on other XCDs, read the row of a determinant they find already inserted -- slot_row() -- to point their
  // records at it directly; one that still reads -1 keeps the slot as its link and the contraction looks the row up)
  if (o.dedup)
    __hip_atomic_store(reinterpret_cast<int32_t *>(o.dedup + (size_t)s * dedup_slot_words(LEN)) + dedup_row_offset(LEN), r, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int i = 0; i < LEN; ++i) o.uniq_onv[(size_t)r * LEN + i] = ket[i];
  if (o.uniq_parent) o.uniq_parent[r] = o.parent;
  return true;
}

// the row a de-duplication slot has been given so far (-1: none yet)
template <int LEN, typename T>
__device__ __forceinline__ int32_t slot_row(const OnepassOut<T> &o, uint32_t s) {
  return __hip_atomic_load(reinterpret_cast<int32_t *>(o.dedup + (size_t)s * dedup_slot_words(LEN)) + dedup_row_offset(LEN), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// The final link of a record whose determinant sits in de-duplication slot `link` (>= 0): the row itself when it is known -- this
// lane's new row `mine`, or the row another record's winner has already written --, else the slot.  Half of the contraction's time was
// the slot look-up: a dependent 4-byte gather from a 64 MB table in front of the amplitude gather.
template <int LEN, typename T>
__device__ __forceinline__ int32_t final_link(const OnepassOut<T> &o, int32_t link, int32_t mine) {
  if (link < 0) return link;
  if (!o.dedup) return mine >= 0 && (uint32_t)mine < o.ucap ? (mine | kDirectLink) : -1;  // (no de-duplication: own row, or none: overflow)
  const int32_t r = mine >= 0 ? mine : slot_row<LEN, T>(o, (uint32_t)link);
  return r >= 0 && (uint32_t)r < o.ucap ? (r | kDirectLink) : link;
}

// The wave writes the +1/-1 rows of the lanes flagged `flag` (all lanes of the wave must call): one coalesced store per row.
template <int LEN, typename T>
__device__ __forceinline__ void emit_rows(const OnepassOut<T> &o, int sorb, bool flag, const uint64_t (&ket)[LEN], int32_t row) {
  if (!o.uniq_pm1 || (o.debug & 2u)) return;
  const int lane = threadIdx.x & 63;
  uint64_t m = __ballot(flag);
  while (m) {
    const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
    m &= m - 1;
    const int32_t r = __builtin_amdgcn_readlane(row, b);
    uint64_t kw[LEN];
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)ket[i], b), hi = __builtin_amdgcn_readlane((uint32_t)(ket[i] >> 32), b);
      kw[i] = ((uint64_t)hi << 32) | lo;
    }
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      const int j = i * 64 + lane;
      if (j < sorb) {
        const bool occ = (kw[i] >> lane) & 1ull;
        if (o.pm1_f32) reinterpret_cast<float *>(o.uniq_pm1)[(size_t)r * sorb + j] = occ ? 1.0f : -1.0f;
        else reinterpret_cast<double *>(o.uniq_pm1)[(size_t)r * sorb + j] = occ ? 1.0 : -1.0;
      }
    }
  }
}

// Lanes flagged `flag` own a new determinant (slot `slot`) that found no room in the workgroup's list: one atomic for the wave.
// All lanes of the wave must call.
template <int LEN, typename T>
__device__ __forceinline__ void allocate_now(const OnepassOut<T> &o, int sorb, bool flag, uint32_t slot, const uint64_t (&ket)[LEN]) {
  const uint64_t m = __ballot(flag);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int32_t base = 0;
  if (lane == leader) base = atomicAdd(o.counters, (int32_t)__popcll(m));
  base = __shfl(base, leader);
  const int32_t r = base + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
  const bool ok = flag && assign_row<LEN, T>(o, slot, r, ket);
  emit_rows<LEN, T>(o, sorb, ok, ket, r);
}

// End of a phase: the workgroup's new determinants get their rows.  Every thread of the block must call; contains barriers.
template <int LEN, typename T>
__device__ __forceinline__ void flush_winner_list(const OnepassOut<T> &o, const WinnerList &wl, int32_t *wl_base, const SDParams &p,
                                                  const LdsLayout &L, const Walker<LEN> &wk) {
  __syncthreads();
  const uint32_t n = min(*wl.n, wl.cap);
  if (threadIdx.x == 0 && n) *wl_base = atomicAdd(o.counters, (int32_t)n);
  __syncthreads();
  if (n) {
    const int32_t base = *wl_base;
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < n; i0 += blockDim.x) {
      const uint32_t i = i0 + lane;
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = wk.w[w];
      bool ok = false;
      if (i < n) {
        const uint32_t col = wl.col[i];
        if (col) {
          const Excitation x = decode(col - 1, p, L);
          make_ket<LEN>(wk, x, ket);
        }
        ok = assign_row<LEN, T>(o, wl.slot[i], base + (int32_t)i, ket);
      }
      emit_rows<LEN, T>(o, p.sorb, ok, ket, base + (int32_t)i);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *wl.n = 0u;
  __syncthreads();
}

// ---- phase A -------------------------------------------------------------------------------------------------------
// Wave-private LDS: the wave's quarter of the singles staging scratch doubles as the buffer of a doubles tile's kept columns
// (column, value); one word for the running count.

template <int LEN, typename T, bool SAMPLED>
struct KeepSink {
  T eps;
  uint32_t chunk, nchunks, tS;
  int64_t seg_base;
  volatile uint32_t *run;  // kept columns of the current doubles tile (wave-private LDS)
  WinnerList wl;
  uint32_t *bufc;
  T *bufh;
  volatile uint32_t *dstat;
  double *tsum;
  uint32_t *kept_total;
  const SDParams *p;
  const LdsLayout *L;
  const Walker<LEN> *wk;
  OnepassOut<T> o;
  uint32_t tile;
  double sub;

  // returns the de-duplication slot; unlisted: a new determinant whose row the caller has to allocate
  __device__ __forceinline__ int32_t record(int64_t g, uint32_t col, T h, const uint64_t (&ket)[LEN], bool &unlisted) const {
    o.rec_col[g] = (int32_t)col;
    o.rec_w[g] = h;
    if (o.rec_onv) {
#pragma unroll
      for (int i = 0; i < LEN; ++i) o.rec_onv[g * LEN + i] = ket[i];
    }
    const int32_t link = resolve_amplitude<LEN, T>(o, wl, ket, col, unlisted);
    o.rec_link[g] = link;
    return link;
  }

  // a column with a fixed slot (called from divergent code: any set of lanes)
  __device__ __forceinline__ void fixed_slot(uint32_t slot, uint32_t col, T h, const uint64_t (&ket)[LEN]) {
    const T a = fabs(h);
    if (!(a >= eps)) {
      if constexpr (SAMPLED) sub += (double)a;
      return;  // (the slot was pre-filled with -1)
    }
    bool unlisted;
    const int32_t link = record(seg_base + slot, col, h, ket, unlisted);
    if (unlisted) {  // (list full: this lane alone, inside divergent code -- rare)
      const int32_t r = atomicAdd(o.counters, 1);
      if (assign_row<LEN, T>(o, (uint32_t)link, r, ket) && o.uniq_pm1) {
        for (int j = 0; j < p->sorb; ++j) {
          const bool occ = (ket[j >> 6] >> (j & 63)) & 1ull;
          if (o.pm1_f32) reinterpret_cast<float *>(o.uniq_pm1)[(size_t)r * p->sorb + j] = occ ? 1.0f : -1.0f;
          else reinterpret_cast<double *>(o.uniq_pm1)[(size_t)r * p->sorb + j] = occ ? 1.0 : -1.0;
        }
      }
    }
  }

  __device__ __forceinline__ uint32_t advance(uint32_t total) const {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)__ballot(1)) - 1;
    uint32_t before = 0;
    if (lane == leader) { before = *run; *run = before + total; }
    return __shfl(before, leader);
  }

  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&ket)[LEN]) {
    if (tile == 0) { fixed_slot(col == 0 ? 0u : 1u + (threadIdx.x & 63u), col, h, ket); return; }
    if (tile <= tS) {
      const uint32_t r0 = (chunk + (tile - 1) * nchunks) * kSinglesPerTile;
      fixed_slot(kFixedHead + (tile - 1) * kSinglesPerTile + (col - 1 - r0), col, h, ket);
      return;
    }
    const T a = fabs(h);
    const bool k = a >= eps;
    if constexpr (SAMPLED) { if (!k) sub += (double)a; }
    const uint64_t m = __ballot(k);
    if (!m) return;
    const uint32_t before = advance((uint32_t)__popcll(m));
    if (k) {
      const uint32_t at = before + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
      bufc[at] = col;
      bufh[at] = h;
    }
  }
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&)[LEN], uint32_t c1, T h1, const uint64_t (&)[LEN]) {
    const T a0 = fabs(h0), a1 = fabs(h1);
    const bool a = a0 >= eps, b = a1 >= eps;
    if constexpr (SAMPLED) sub += (a ? 0.0 : (double)a0) + (b ? 0.0 : (double)a1);
    const uint64_t ma = __ballot(a), mb = __ballot(b);
    if (!(ma | mb)) return;
    const uint32_t before = advance((uint32_t)(__popcll(ma) + __popcll(mb)));
    const uint64_t below = (1ull << (threadIdx.x & 63)) - 1ull;
    const uint32_t mine = before + __popcll(ma & below) + __popcll(mb & below);
    if (a) { bufc[mine] = c0; bufh[mine] = h0; }
    if (b) { bufc[mine + (a ? 1 : 0)] = c1; bufh[mine + (a ? 1 : 0)] = h1; }
  }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&k0)[LEN], const uint64_t (&k1)[LEN]) {
    two(col, h0, k0, col + 1, h1, k1);
  }

  // Exclusive prefix of the kept counts of the doubles tiles before tile d (decoupled look-back, one window of 64 tiles per step).
  __device__ __forceinline__ uint32_t lookback(uint32_t d, uint32_t c) const {
    const int lane = threadIdx.x & 63;
    if (d == 0 || (o.debug & 4u)) {
      if (lane == 0) dstat[d] = kStatP | c;
      return 0;
    }
    if (lane == 0) dstat[d] = kStatA | c;
    uint32_t excl = 0;
    int32_t top = (int32_t)d - 1;
    for (;;) {
      const int32_t idx = top - lane;
      const uint32_t s = idx >= 0 ? dstat[idx] : kStatP;  // below tile 0: an inclusive prefix of 0
      const uint64_t ready = __ballot(s != 0u);
      const uint64_t isP = __ballot((s & kStatP) != 0u);
      const int fp = isP ? __ffsll((long long)isP) - 1 : 64;  // nearest tile whose inclusive prefix is known
      const uint64_t need = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
      if ((ready & need) != need) {
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      excl += op_wave_sum(lane <= fp ? (s & kStatMask) : 0u);
      if (fp < 64) break;
      top -= 64;
    }
    if (lane == 0) dstat[d] = kStatP | (excl + c);
    return excl;
  }

  __device__ __forceinline__ void flush() {  // wave-uniform
    if (tile == 0xffffffffu) return;
    const int lane = threadIdx.x & 63;
    if constexpr (SAMPLED) {
      const double s = op_wave_sum(sub);
      if (lane == 0) tsum[tile] = s;
    }
    wave_sync();
    if (tile <= tS) return;  // column 0, the singles and the unpaired doubles wrote their fixed slots themselves
    const uint32_t c = *run;
    const uint32_t excl = lookback(tile - 1 - tS, c);
    if (lane == 0 && c) atomicMax(kept_total, excl + c);
    for (uint32_t i0 = 0; i0 < c; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool act = i < c && excl + i < o.cap_d;
      bool unlisted = false;
      int32_t link = -1;
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = 0ull;
      if (act) {
        const uint32_t col = bufc[i];
        const Excitation x = decode(col - 1, *p, *L);
        make_ket<LEN>(*wk, x, ket);
        link = record(seg_base + o.fixed + excl + i, col, bufh[i], ket, unlisted);
      }
      if (__ballot(unlisted)) allocate_now<LEN, T>(o, p->sorb, unlisted, (uint32_t)link, ket);
    }
  }
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush();
    tile = t;
    sub = 0.0;
    if ((threadIdx.x & 63) == 0) *run = 0u;
    __builtin_amdgcn_wave_barrier();
  }
};

// ---- phase C: the draws inside a tile (same scheme as kernels_reduce_sample.hip: SampleSink) ------------------------------
struct DrawLds {
  double *prefix;
  uint32_t *cs;
  uint32_t *hits;
  volatile uint32_t *ncols;
  volatile double *run;
};
constexpr size_t kDrawLdsPerWave = (size_t)kOneTileCols * (8 + 4 + 4) + 16;
// ... of the row-cache form: running sums f64[cols], hit counts u16[cols] (pairs in 32-bit words: LDS atomics are 32-bit).  The column of
// an entry is its position and its sign stays in the register of the lane that loaded it: nothing else is stored.  (16.4 -> 10.2 KB per
// workgroup: more workgroups per CU.)
constexpr size_t kCachedDrawLdsPerWave = (size_t)kOneTileCols * (8 + 2);

template <int LEN, typename T>
struct DrawSink {
  T eps;
  DrawLds S;
  const SDParams *p;
  const LdsLayout *L;
  const Walker<LEN> *wk;
  const uint32_t *dinfo;  // LDS: slot offset << 16 | draws, per tile
  double scale;           // S_walker / N
  uint64_t key;
  int64_t sbase;          // first drawn-record slot of this walker
  OnepassOut<T> o;
  WinnerList wl;
  uint32_t tile;

  __device__ __forceinline__ void entry(uint32_t idx, uint32_t col, T h, double incl) const {
    S.prefix[idx] = incl;
    S.cs[idx] = col | (h < T(0) ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ double width(T h) const {
    const T a = fabs(h);
    return a >= eps ? 0.0 : (double)a;
  }
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(1);
    const double w = width(h);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (lane == b) {
        const uint32_t idx = *S.ncols;
        const double incl = *S.run + w;
        entry(idx, col, h, incl);
        *S.ncols = idx + 1;
        *S.run = incl;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&)[LEN], const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const uint32_t nact = (uint32_t)__popcll(__ballot(1));
    const double w0 = width(h0), w1 = width(h1);
    const double incl = op_scan(w0 + w1, lane);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + 2 * lane, col, h0, run + incl - w1);
    entry(base + 2 * lane + 1, col + 1, h1, run + incl);
    const double total = __shfl(incl, (int)nact - 1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 2 * nact; *S.run = run + total; }
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&)[LEN], uint32_t c1, T h1, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const double w0 = width(h0), w1 = width(h1);
    const double i0 = op_scan(w0, lane), t0 = __shfl(i0, 63);
    const double i1 = op_scan(w1, lane), t1 = __shfl(i1, 63);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + lane, c0, h0, run + i0);
    entry(base + 64 + lane, c1, h1, run + t0 + i1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 128; *S.run = run + t0 + t1; }
    __builtin_amdgcn_wave_barrier();
  }

  __device__ __forceinline__ void flush() {  // wave-uniform
    if (tile == 0xffffffffu) return;
    const int lane = threadIdx.x & 63;
    const uint32_t info = dinfo[tile];
    const uint32_t draws = info & 0xffffu;
    const uint32_t ncols = *S.ncols;
    const double total = *S.run;
    if (draws == 0 || ncols == 0 || !(total > 0.0)) return;
    for (uint32_t k = lane; k < draws; k += 64) {
      const uint64_t r = op_mix64(key ^ op_mix64(((uint64_t)tile << 32) | k));
      const double target = (double)(r >> 11) * 0x1.0p-53 * total;
      uint32_t lo = 0, hi = ncols;  // first idx with prefix[idx] > target
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.prefix[mid] > target) hi = mid; else lo = mid + 1;
      }
      if (lo >= ncols) lo = ncols - 1;
      while (lo > 0 && !(S.prefix[lo] > S.prefix[lo - 1])) --lo;  // rounding at the end: back to a column of positive width
      atomicAdd(&S.hits[lo], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    int64_t pos = sbase + (info >> 16);
    for (uint32_t i0 = 0; i0 < ncols; i0 += 64) {
      const uint32_t idx = i0 + lane;
      const uint32_t hc = idx < ncols ? S.hits[idx] : 0u;
      const uint64_t m = __ballot(hc != 0u);
      bool unlisted = false;
      int32_t link = -1;
      uint64_t ket[LEN];
#pragma unroll
      for (int w = 0; w < LEN; ++w) ket[w] = 0ull;
      if (hc) {
        const uint32_t e = S.cs[idx], col = e & 0x7fffffffu;
        const int64_t at = pos + __popcll(m & ((1ull << lane) - 1ull));
        if (col == 0) {
#pragma unroll
          for (int i = 0; i < LEN; ++i) ket[i] = wk->w[i];
        } else {
          const Excitation x = decode(col - 1, *p, *L);
          make_ket<LEN>(*wk, x, ket);
        }
        o.srec_col[at] = (int32_t)col;
        const double v = scale * (double)hc;
        o.srec_w[at] = (T)((e >> 31) ? -v : v);
        if (o.srec_onv) {
#pragma unroll
          for (int i = 0; i < LEN; ++i) o.srec_onv[at * LEN + i] = ket[i];
        }
        link = resolve_amplitude<LEN, T>(o, wl, ket, col, unlisted);
        o.srec_link[at] = link;
      }
      if (__ballot(unlisted)) allocate_now<LEN, T>(o, p->sorb, unlisted, (uint32_t)link, ket);
      pos += __popcll(m);
    }
  }
  __device__ __forceinline__ bool skip_tile(uint32_t t) const { return (dinfo[t] & 0xffffu) == 0u; }
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush();
    tile = t;
    if ((dinfo[t] & 0xffffu) == 0u) return;
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kOneTileCols; i += 64) S.hits[i] = 0u;
    if (lane == 0) { *S.ncols = 0u; *S.run = 0.0; }
    __builtin_amdgcn_wave_barrier();
  }
};

// LDS after the walker tables and the staging scratch: dstat[max_tiles] | (SAMPLED) tsum[max_tiles] f64, dinfo[max_tiles], draw areas
__host__ __device__ inline size_t onepass_lds(const SDParams &p, size_t esz, uint32_t max_tiles, bool sampled, uint32_t wl_cap) {
  size_t b = (lds_bytes(p, esz) + 15) & ~(size_t)15;
  b += ((size_t)max_tiles * 4 + 15) & ~(size_t)15;
  if (sampled) {
    b += (size_t)max_tiles * 8;
    b += ((size_t)max_tiles * 4 + 15) & ~(size_t)15;
    b += (kBlock / 64) * kDrawLdsPerWave;
  }
  return b + (size_t)wl_cap * 8;
}

// entries of a workgroup's list of new determinants: what one phase can win (all its draws; a few hundred kept columns), within the
// LDS that is left
__host__ inline uint32_t winner_list_cap(int eps_sample) {
  uint32_t c = eps_sample > 256 ? (uint32_t)eps_sample : 256u;
  c = (c + 63u) & ~63u;
  return c > 2048u ? 2048u : c;
}

template <int LEN, typename T, bool SAMPLED>
__global__ __launch_bounds__(kBlock) void reduce_onepass_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks,
                                                                uint32_t chunk_len, uint32_t max_tiles, const T *__restrict__ plan, T eps,
                                                                uint32_t nsample, uint64_t seed, uint32_t wl_cap, OnepassOut<T> o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t next_tile, kept_total, wl_n;
  __shared__ int32_t wl_base;
  __shared__ uint32_t wave_run[kBlock / 64];
  __shared__ double s_part[kBlock / 64 + 1];
  __shared__ uint32_t s_parti[kBlock / 64 + 1];
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, false, walker, chunk);
  o.parent = (int32_t)walker;
  const uint64_t slot = walker * nchunks + chunk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t seg_base = (int64_t)slot * ((int64_t)o.fixed + o.cap_d);
  if (tid == 0) { next_tile = 0; kept_total = 0; wl_n = 0; }
  unsigned char *extra = smem + ((lds_bytes(p, sizeof(T)) + 15) & ~(size_t)15);
  volatile uint32_t *dstat = reinterpret_cast<volatile uint32_t *>(extra);
  extra += ((size_t)max_tiles * 4 + 15) & ~(size_t)15;
  double *tsum = reinterpret_cast<double *>(extra);
  uint32_t *dinfo = reinterpret_cast<uint32_t *>(extra + (SAMPLED ? (size_t)max_tiles * 8 : 0));
  WinnerList wl;
  wl.n = &wl_n;
  wl.cap = wl_cap;
  wl.slot = reinterpret_cast<uint32_t *>(smem + onepass_lds(p, sizeof(T), max_tiles, SAMPLED, 0));
  wl.col = wl.slot + wl_cap;
  for (uint32_t i = tid; i < max_tiles; i += kBlock) {
    dstat[i] = 0u;
    if constexpr (SAMPLED) { tsum[i] = 0.0; dinfo[i] = 0u; }
  }
  for (uint32_t i = tid; i < o.fixed; i += kBlock) o.rec_col[seg_base + i] = -1;
  if constexpr (SAMPLED) {
    for (uint32_t i = tid; i < nsample; i += kBlock) o.srec_col[(int64_t)walker * nsample + i] = -1;
  }
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);  // (ends with a barrier: the pre-fills above are done)

  const uint32_t tS_all = (p.d1 + kSinglesPerTile - 1) / kSinglesPerTile;
  const uint32_t tS = tS_all > chunk ? (tS_all - chunk + nchunks - 1) / nchunks : 0;
  T *quarter = reinterpret_cast<T *>(L.scratch) + wave * (kDiagTile / 4);
  {
    KeepSink<LEN, T, SAMPLED> sink;
    sink.eps = eps; sink.chunk = chunk; sink.nchunks = nchunks; sink.tS = tS; sink.seg_base = seg_base;
    sink.run = wave_run + wave; sink.wl = wl;
    sink.bufh = quarter;
    sink.bufc = reinterpret_cast<uint32_t *>(quarter + kOneTileCols);
    sink.dstat = dstat; sink.tsum = tsum; sink.kept_total = &kept_total;
    sink.p = &p; sink.L = &L; sink.wk = &wk; sink.o = o; sink.tile = 0xffffffffu; sink.sub = 0.0;
    visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
    sink.flush();
  }
  flush_winner_list<LEN, T>(o, wl, &wl_base, p, L, wk);  // (barriers inside: phase A is over for every wave)
  if (tid == 0) {
    o.seg_count[slot] = (int32_t)kept_total;
    if (kept_total > o.cap_d) {  // (what the largest overflowing segment needed; 0 when everything fitted)
      atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 1u);
      atomicMax(o.counters + 2, (int32_t)kept_total);
    }
  }
  if constexpr (SAMPLED) {
    // ---- phase B: inclusive scan of the tile sums (fixed order of additions), the N draws over the tiles, slot offsets ----
    const uint32_t per = (max_tiles + kBlock - 1) / kBlock;
    const uint32_t b0 = min((uint32_t)tid * per, max_tiles), b1 = min(b0 + per, max_tiles);
    double local = 0.0;
    for (uint32_t i = b0; i < b1; ++i) local += tsum[i];
    double incl = op_scan(local, lane);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    double before = 0.0, total = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) {
      if (w < wave) before += s_part[w];
      total += s_part[w];
    }
    double run = before + incl - local;
    for (uint32_t i = b0; i < b1; ++i) { run += tsum[i]; tsum[i] = run; }
    __syncthreads();
    const double Srow = total;
    if (tid == 0 && o.row_sum) o.row_sum[walker] = Srow;
    const uint64_t key = op_mix64((o.seed_dev ? seed + *o.seed_dev : seed) ^ op_mix64(slot));
    if (Srow > 0.0) {
      for (uint32_t k = tid; k < nsample; k += kBlock) {
        const uint64_t r = op_mix64(key ^ op_mix64(0xffffffff00000000ull | k));
        const double target = (double)(r >> 11) * 0x1.0p-53 * Srow;
        uint32_t lo = 0, hi = max_tiles;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (tsum[mid] > target) hi = mid; else lo = mid + 1;
        }
        if (lo >= max_tiles) lo = max_tiles - 1;
        while (lo > 0 && !(tsum[lo] > tsum[lo - 1])) --lo;
        atomicAdd(&dinfo[lo], 1u);
      }
    }
    __syncthreads();
    // exclusive scan of the draw counts -> offset << 16 | count
    uint32_t lsum = 0;
    for (uint32_t i = b0; i < b1; ++i) lsum += dinfo[i];
    uint32_t iscan = lsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t ov = __shfl_up(iscan, d);
      if (lane >= d) iscan += ov;
    }
    if (lane == 63) s_parti[wave] = iscan;
    __syncthreads();
    uint32_t ibefore = 0;
    for (int w = 0; w < wave; ++w) ibefore += s_parti[w];
    uint32_t off = ibefore + iscan - lsum;
    for (uint32_t i = b0; i < b1; ++i) {
      const uint32_t c = dinfo[i];
      dinfo[i] = (off << 16) | c;
      off += c;
    }
    if (tid == 0) next_tile = 0;
    __syncthreads();
    // ---- phase C ----
    unsigned char *mine = reinterpret_cast<unsigned char *>(dinfo) + (((size_t)max_tiles * 4 + 15) & ~(size_t)15) + (size_t)wave * kDrawLdsPerWave;
    DrawLds S;
    S.prefix = reinterpret_cast<double *>(mine);
    S.run = reinterpret_cast<volatile double *>(mine + (size_t)kOneTileCols * 8);
    S.cs = reinterpret_cast<uint32_t *>(mine + (size_t)kOneTileCols * 8 + 8);
    S.hits = S.cs + kOneTileCols;
    S.ncols = reinterpret_cast<volatile uint32_t *>(S.hits + kOneTileCols);
    DrawSink<LEN, T> sink{eps, S, &p, &L, &wk, dinfo, Srow / (double)nsample, key, (int64_t)walker * nsample, o, wl, 0xffffffffu};
    visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
    sink.flush();
    flush_winner_list<LEN, T>(o, wl, &wl_base, p, L, wk);
  }
}


// ====================================================================================================================
// LIST form (the production regime: a few hundred kept columns per segment, e.g. the Fe2S2 example's eps = 1e-2).
// The kept columns of a workgroup do not go through per-wave buffers and a look-back: every lane that keeps a column appends
// (column, value) to ONE LDS list of the workgroup (an LDS atomic; 1 % of the columns).  When the row has been visited the
// workgroup sorts the list by column -- records come out in ASCENDING COLUMN order, like the reference's boolean mask, whatever the
// waves' timing was -- and only then, with all 256 lanes busy, forms the kets, writes the records and asks the wave-function table /
// the de-duplication table for each of them: one round of probe latency per 256 records instead of one per tile.  New determinants
// of a batch take their rows with one global atomic.  The drawn records of phase C are resolved the same way after the draws.
// The look-back form above remains for segments whose kept columns do not fit the LDS list.
// (Tried and dropped, round 3: a ROW form for rows that fit the LDS -- one 1024-thread workgroup per walker keeps the row's sub-eps
// elements in LDS (64 KiB for Fe2S2) and draws from a block-wide prefix sum instead of re-enumerating the drawn tiles.  Correct, but
// 1330 us against 1056 us per 8192 Fe2S2 walkers: with one workgroup per CU nothing overlaps the serial tails (sort, scan, draws,
// resolution), which four 256-thread workgroups per CU hide behind each other's enumeration.)

template <int LEN, typename T>
__device__ __forceinline__ int32_t probe_amplitude(const OnepassOut<T> &o, const uint64_t (&ket)[LEN], bool &won, uint32_t *full_flag = nullptr) {
  won = false;
  if (o.debug & 1u) return -1;
  if (o.lut) {
    const int64_t pos = hash_find<LEN>(o.lut, o.lut_cap, ket);
    if (pos >= 0) return (int32_t)(-2 - pos);
  }
  if (!o.dedup) {  // no de-duplication (io->dedup_table == NULL): every record gets a row of its own
    won = true;
    return 0;
  }
  // a call whose table has overflowed is going to be repeated with a larger one: once a probe of this WORKGROUP has run to its limit
  // (full_flag, in LDS) its further records skip the table (every probe of a full table walks kProbeLimit slots: 0.3 - 0.7 s per launch
  // at sorb 80 with 4096 walkers).  Not the global overflow word: even ONE load of it per workgroup waits behind the row counter's
  // atomics on the same line (Fe2S2 step 0.94 -> 1.18 ms), one per record is 10^7 requests to one L2 channel (2.3 ms).
  if (full_flag && *full_flag) return -1;
  const uint32_t s = dedup_insert<LEN>(o.dedup, o.dedup_mask, ket, won);
  if (s == 0xffffffffu) {
    won = false;
    atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 2u);
    if (full_flag) *full_flag = 1u;
    return -1;
  }
  return (int32_t)s;
}

// The lanes of the WORKGROUP flagged `won` own new determinants (de-duplication slot `slot`): one global atomic for all of them,
// then the slots' rows, the determinants and the +-1 rows.  Every thread of the block must call; contains barriers.
template <int LEN, typename T>
__device__ __forceinline__ int32_t allocate_batch(const OnepassOut<T> &o, int sorb, bool won, uint32_t slot, const uint64_t (&ket)[LEN],
                                                  uint32_t *bw_cnt, int32_t *bw_base) {
  int32_t mine = -1;  // the row this lane's determinant got
  const int lane = threadIdx.x & 63;
  const uint64_t m = __ballot(won);
  uint32_t woff = 0;
  if (m && lane == 0) woff = atomicAdd(bw_cnt, (uint32_t)__popcll(m));
  woff = __shfl(woff, 0);
  __syncthreads();
  const uint32_t total = *bw_cnt;
  if (threadIdx.x == 0 && total) *bw_base = atomicAdd(o.counters, (int32_t)total);
  __syncthreads();
  if (total) {
    const int32_t r = *bw_base + (int32_t)woff + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool ok = won && assign_row<LEN, T>(o, slot, r, ket);
    emit_rows<LEN, T>(o, sorb, ok, ket, r);
    if (ok) mine = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) *bw_cnt = 0u;
  __syncthreads();
  return mine;
}

// The same for K records per thread (flags won[k], slots slot[k], kets ket[k]): ONE global atomic for up to K * blockDim new determinants.
// A walker's 1000 draw slots are resolved in one go: the serial chain per walker (probe latency + the atomic's round trip + barriers) is
// paid once instead of four times (semi-stochastic kernel: -100 us per 8192 Fe2S2 walkers).
template <int LEN, typename T, int K>
__device__ __forceinline__ void allocate_batch_k(const OnepassOut<T> &o, int sorb, const bool (&won)[K], const uint32_t (&slot)[K],
                                                 const uint64_t (&ket)[K][LEN], uint32_t *bw_cnt, int32_t *bw_base, int32_t (&rows)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) rows[k] = -1;
  const int lane = threadIdx.x & 63;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) mine += won[k] ? 1u : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ov = __shfl_up(incl, d);
    if (lane >= d) incl += ov;
  }
  const uint32_t wave_total = __shfl(incl, 63);
  uint32_t woff = 0;
  if (wave_total && lane == 0) woff = atomicAdd(bw_cnt, wave_total);
  woff = __shfl(woff, 0);
  __syncthreads();
  const uint32_t total = *bw_cnt;
  if (threadIdx.x == 0 && total) *bw_base = atomicAdd(o.counters, (int32_t)total);
  __syncthreads();
  if (total) {
    int32_t r = *bw_base + (int32_t)(woff + incl - mine);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool ok = won[k] && assign_row<LEN, T>(o, slot[k], r, ket[k]);
      emit_rows<LEN, T>(o, sorb, ok, ket[k], r);
      if (ok) rows[k] = r;
      r += won[k] ? 1 : 0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *bw_cnt = 0u;
  __syncthreads();
}

// FLUSH (the flushing LIST form, rows whose kept columns exceed the list): the workgroup empties the list whenever it is nearly full
// (pause(): asked by visit_tiles before a wave takes a tile); entries of tile 0 (column 0 and the few unpaired doubles, whose columns lie
// anywhere in the row) keep bit 63 of the key clear and so sort in front of everything else of the first flush, the others carry it:
// the records of a segment are then the same whatever the timing of the flushes.
template <int LEN, typename T, bool SAMPLED, bool CACHED = false, bool FLUSH = false>
struct ListKeepSink {
  T eps;
  uint32_t *list_n;
  unsigned long long *list_key;  // LDS: column << 32 | position of the value in rec_w (unsorted), sorted afterwards
  T *__restrict__ rec_w;         // this segment's record weights (global): the kept values wait there in order of arrival
  uint32_t cap;
  double *tsum;
  uint32_t tile;
  double sub;
  T *__restrict__ hrow;  // CACHED: this walker's row of matrix elements in global memory (the draws read it back instead of a second enumeration)
  uint32_t pause_at = 0;          // FLUSH: the list is emptied once it holds more than this
  unsigned long long tag = 0ull;  // FLUSH: bit 63 for the entries of every tile but tile 0
  template <bool F = FLUSH, typename = std::enable_if_t<F>>
  __device__ __forceinline__ bool pause() const {
    return __builtin_amdgcn_readfirstlane(__atomic_load_n(list_n, __ATOMIC_RELAXED)) > pause_at;
  }
  __device__ __forceinline__ void add(uint32_t col, T h) {
    const T a = fabs(h);
    if constexpr (CACHED) hrow[col] = h;
    if (a >= eps) {
      const uint32_t k = atomicAdd(list_n, 1u);
      if (k < cap) {
        list_key[k] = FLUSH ? (((unsigned long long)col << 32) | k | tag) : (((unsigned long long)col << 32) | k);
        rec_w[k] = h;
      }
    } else if constexpr (SAMPLED && !CACHED) {
      sub += (double)a;
    }
  }
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&)[LEN]) { add(col, h); }
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&)[LEN], uint32_t c1, T h1, const uint64_t (&)[LEN]) { add(c0, h0); add(c1, h1); }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&)[LEN], const uint64_t (&)[LEN]) {
    if constexpr (CACHED && sizeof(T) == 8) {
      // the two neighbouring elements of the cached row in ONE 16-byte store when they are aligned (two 8-byte stores of a wave each
      // touch every other 8 bytes of the same lines: twice the write requests at the L2)
      if ((reinterpret_cast<uintptr_t>(hrow + col) & 15u) == 0) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<d2 *>(hrow + col) = d2{(double)h0, (double)h1};
        add_nocache(col, h0); add_nocache(col + 1, h1);
        return;
      }
    }
    add(col, h0); add(col + 1, h1);
  }
  __device__ __forceinline__ void add_nocache(uint32_t col, T h) {
    const T a = fabs(h);
    if (a >= eps) {
      const uint32_t k = atomicAdd(list_n, 1u);
      if (k < cap) { list_key[k] = ((unsigned long long)col << 32) | k; rec_w[k] = h; }
    } else if constexpr (SAMPLED && !CACHED) {
      sub += (double)a;
    }
  }
  __device__ __forceinline__ void flush() {
    if constexpr (SAMPLED && !CACHED) {
      if (tile == 0xffffffffu) return;
      const double s = op_wave_sum(sub);
      if ((threadIdx.x & 63) == 0) tsum[tile] = s;
    }
  }
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush(); tile = t; sub = 0.0;
    if constexpr (FLUSH) tag = t ? (1ull << 63) : 0ull;
  }
};

// phase C of the LIST form: as DrawSink, but a drawn record only notes its column in the LDS array `pend` (one entry per draw
// slot of the walker); kets, links and rows follow for all of them together
template <int LEN, typename T>
struct ListDrawSink {
  T eps;
  DrawLds S;
  const uint32_t *dinfo;
  double scale;
  uint64_t key;
  int64_t sbase;
  int32_t *__restrict__ srec_col;
  T *__restrict__ srec_w;
  uint32_t *pend;
  uint32_t tile;
  bool nodraw = false;
  uint32_t info_cur = 0;            // dinfo[tile], read once per tile (global memory on long rows: three dependent loads per drawn tile otherwise)
  const uint32_t *tlist = nullptr;  // (optional) the drawn tiles, ascending: visit_tiles then walks this list instead of asking skip_tile per tile
  uint32_t ndrawn = 0;
  __device__ __forceinline__ uint32_t remap(uint32_t k) const { return tlist ? (k < ndrawn ? tlist[k] : 0xffffffffu) : k; }

  __device__ __forceinline__ void entry(uint32_t idx, uint32_t col, T h, double incl) const {
    S.prefix[idx] = incl;
    S.cs[idx] = col | (h < T(0) ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ double width(T h) const {
    const T a = fabs(h);
    return a >= eps ? 0.0 : (double)a;
  }
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(1);
    const double w = width(h);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (lane == b) {
        const uint32_t idx = *S.ncols;
        const double incl = *S.run + w;
        entry(idx, col, h, incl);
        *S.ncols = idx + 1;
        *S.run = incl;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&)[LEN], const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const uint32_t nact = (uint32_t)__popcll(__ballot(1));
    const double w0 = width(h0), w1 = width(h1);
    const double incl = op_scan(w0 + w1, lane);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + 2 * lane, col, h0, run + incl - w1);
    entry(base + 2 * lane + 1, col + 1, h1, run + incl);
    const double total = __shfl(incl, (int)nact - 1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 2 * nact; *S.run = run + total; }
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&)[LEN], uint32_t c1, T h1, const uint64_t (&)[LEN]) const {
    const int lane = threadIdx.x & 63;
    const double w0 = width(h0), w1 = width(h1);
    const double i0 = op_scan(w0, lane), t0 = __shfl(i0, 63);
    const double i1 = op_scan(w1, lane), t1 = __shfl(i1, 63);
    const uint32_t base = *S.ncols;
    const double run = *S.run;
    entry(base + lane, c0, h0, run + i0);
    entry(base + 64 + lane, c1, h1, run + t0 + i1);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { *S.ncols = base + 128; *S.run = run + t0 + t1; }
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void flush() {  // wave-uniform
    if (tile == 0xffffffffu) return;
    const int lane = threadIdx.x & 63;
    const uint32_t info = info_cur;
    const uint32_t draws = info & 0xffffu;
    const uint32_t ncols = *S.ncols;
    const double total = *S.run;
    if (draws == 0 || ncols == 0 || !(total > 0.0) || nodraw) return;
    for (uint32_t k = lane; k < draws; k += 64) {
      const uint64_t r = op_mix64(key ^ op_mix64(((uint64_t)tile << 32) | k));
      const double target = (double)(r >> 11) * 0x1.0p-53 * total;
      uint32_t lo = 0, hi = ncols;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.prefix[mid] > target) hi = mid; else lo = mid + 1;
      }
      if (lo >= ncols) lo = ncols - 1;
      while (lo > 0 && !(S.prefix[lo] > S.prefix[lo - 1])) --lo;
      atomicAdd(&S.hits[lo], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t pos = info >> 16;
    for (uint32_t i0 = 0; i0 < ncols; i0 += 64) {
      const uint32_t idx = i0 + lane;
      const uint32_t hc = idx < ncols ? S.hits[idx] : 0u;
      const uint64_t m = __ballot(hc != 0u);
      if (hc) {
        const uint32_t e = S.cs[idx], col = e & 0x7fffffffu;
        const uint32_t at = pos + __popcll(m & ((1ull << lane) - 1ull));
        srec_col[sbase + at] = (int32_t)col;
        const double v = scale * (double)hc;
        srec_w[sbase + at] = (T)((e >> 31) ? -v : v);
        pend[at] = col;
      }
      pos += __popcll(m);
    }
  }
  __device__ __forceinline__ bool skip_tile(uint32_t) const { return (info_cur & 0xffffu) == 0u; }  // (asked after tile_begin)
  __device__ __forceinline__ void tile_begin(uint32_t t) {
    flush();
    tile = t;
    info_cur = dinfo[t];
    if ((info_cur & 0xffffu) == 0u) return;
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kOneTileCols; i += 64) S.hits[i] = 0u;
    if (lane == 0) { *S.ncols = 0u; *S.run = 0.0; }
    __builtin_amdgcn_wave_barrier();
  }
};

// The LIST form keeps nothing in the staging scratch but the singles' and the diagonal's terms.  The sampled kernel, which is short
// of LDS (draw areas), takes 256 elements per wave instead of 512: two rounds per singles tile, 3 -> 4 workgroups per CU
// (1253 -> 1057 us per 8192 Fe2S2 walkers; 128 elements: 1148 us); the deterministic kernel too (end of round 3: 260 -> 248 us with 256,
// 252 with 384; with 128 it had lost 10 %).
#ifndef PYNQS_LIST_Q_SAMPLED
#define PYNQS_LIST_Q_SAMPLED 256
#endif
#ifndef PYNQS_LIST_Q_DET
#define PYNQS_LIST_Q_DET 256
#endif
__host__ __device__ constexpr int list_quarter(bool sampled) { return sampled ? PYNQS_LIST_Q_SAMPLED : PYNQS_LIST_Q_DET; }
// `cached` (row-cache form): the draws read the row back and never enumerate again, so the waves' draw areas share the memory of the
// staging scratch of phase A (barriers lie between the two uses): 8 KB less per workgroup, 5 instead of 4 workgroups per CU for Fe2S2
__host__ __device__ inline size_t list_scratch_offset(const SDParams &p) { return (lds_fixed_bytes(p) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t list_base_lds(const SDParams &p, size_t esz, bool sampled, bool cached) {
  const size_t scratch = esz * (size_t)(list_quarter(sampled) * (kBlock / 64)), draw = (kBlock / 64) * kCachedDrawLdsPerWave;
  if (cached) return (list_scratch_offset(p) + (scratch > draw ? scratch : draw) + 15) & ~(size_t)15;
  return (lds_fixed_bytes(p) + scratch + 15) & ~(size_t)15;
}

// LDS of the LIST form after the walker tables and the staging scratch:
//   (SAMPLED) tsum[max_tiles] f64 | dinfo[max_tiles] u32 | draw areas ;  then the list: key[P] u64  (P = power of two >= capacity),
//   which the draw slots' columns (pend[N] u32) re-use in phase C
__host__ __device__ inline size_t tile_scratch_stride(uint32_t max_tiles) { return ((size_t)max_tiles * 12 + 15) & ~(size_t)15; }

// gtile: the tile sums and the tiles' draw counts live in global memory (io->tile_scratch) instead of the LDS -- long rows: 4768 tiles at
// sorb 120 are 57 KB, which with the draw areas and the list leaves ONE workgroup per CU
__host__ __device__ inline size_t onepass_list_lds(const SDParams &p, size_t esz, uint32_t max_tiles, bool sampled, uint32_t P, uint32_t nsample,
                                                   bool cached, bool gtile = false) {
  size_t b = list_base_lds(p, esz, sampled, cached);
  if (sampled) {
    if (!gtile) {
      b += (size_t)max_tiles * 8;
      b += ((size_t)max_tiles * 4 + 15) & ~(size_t)15;
    }
    if (!cached) b += (kBlock / 64) * kDrawLdsPerWave;
  }
  const size_t list = (size_t)P * 8, pend = (size_t)nsample * 4 * (gtile ? 2 : 1);  // (gtile: + the list of the drawn tiles)
  return b + ((list > pend ? list : pend) + 15 & ~(size_t)15);
}

template <int LEN, typename T, bool SAMPLED, bool CACHED, bool FLUSH = false, bool GTILE = false>
__device__ __forceinline__ void reduce_onepass_list_body(const uint64_t *__restrict__ bra, const SDParams &p, const PlanLayout &pl, uint32_t nchunks,
                                                         uint32_t chunk_len, uint32_t max_tiles, const T *__restrict__ plan, T eps,
                                                         uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(!FLUSH || !CACHED, "the flushing form is for long rows: no row cache");
  __shared__ uint32_t next_tile, list_n, bw_cnt, s_done, s_full;
  __shared__ int32_t bw_base;
  __shared__ double s_part[kBlock / 64 + 1];
  __shared__ uint32_t s_parti[kBlock / 64 + 1];
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, false, walker, chunk);
  o.parent = (int32_t)walker;
  const uint64_t slot = walker * nchunks + chunk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t cap = o.fixed + o.cap_d;  // records of a segment (<= P unless FLUSH)
  const int64_t seg_base = (int64_t)slot * cap;
  if (tid == 0) { next_tile = 0; list_n = 0; bw_cnt = 0; s_done = 0; s_full = 0; }
  static_assert(!GTILE || (SAMPLED && !CACHED), "tile sums in global memory: the re-enumerating semi-stochastic form only");
  unsigned char *extra = smem + list_base_lds(p, sizeof(T), SAMPLED, CACHED);
  unsigned char *tile_mem = GTILE ? o.tile_scratch + (size_t)walker * o.tile_stride : extra;
  double *tsum = reinterpret_cast<double *>(tile_mem);
  uint32_t *dinfo = reinterpret_cast<uint32_t *>(tile_mem + (SAMPLED ? (size_t)max_tiles * 8 : 0));
  unsigned char *after = (SAMPLED && !GTILE) ? reinterpret_cast<unsigned char *>(dinfo) + (((size_t)max_tiles * 4 + 15) & ~(size_t)15) : extra;
  unsigned char *draw0 = CACHED ? smem + list_scratch_offset(p) : after;  // (CACHED: over the staging scratch, see list_base_lds)
  if (SAMPLED && !CACHED) after += (kBlock / 64) * kDrawLdsPerWave;
  // the kept list: ONE 64-bit key per entry (column << 32 | order of arrival); the values wait in the segment's rec_w, in order of arrival,
  // and are permuted after the sort (12 -> 8 bytes of LDS per entry: with the draw slots' 4000 bytes sharing the memory that is what
  // decides between 7 and 8 workgroups per CU)
  unsigned long long *list_key = reinterpret_cast<unsigned long long *>(after);
  uint32_t *pend = reinterpret_cast<uint32_t *>(after);           // phase C re-uses the list's memory
  if constexpr (SAMPLED) {
    for (uint32_t i = tid; i < max_tiles; i += kBlock) { tsum[i] = 0.0; dinfo[i] = 0u; }
    for (uint32_t i = tid; i < nsample; i += kBlock) o.srec_col[(int64_t)walker * nsample + i] = -1;
  }
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  PYNQS_STAMP(0);
  const int nocc = build_walker_tables<LEN>(wk, p, L);
  PYNQS_STAMP(1);
  // FLUSH: rounds of (enumerate until the list is nearly full, sort, write, resolve) until the tiles are exhausted; the tiles are taken in
  // order and every taken tile is finished before a flush, so the flushes cover consecutive ranges of tiles = ascending columns
  uint32_t flushed = 0;  // records of this segment written by earlier rounds
  uint32_t needed = 0;   // kept columns so far, whether they had room or not
  PYNQS_STAMP_ZERO(10); PYNQS_STAMP_ZERO(11); PYNQS_STAMP_ZERO(12); PYNQS_STAMP_ZERO(13);
  PYNQS_STAMP_MARK();
  for (;;) {
  const uint32_t room = FLUSH ? (cap > flushed ? min(cap - flushed, P) : 0u) : cap;
  {
    ListKeepSink<LEN, T, SAMPLED, CACHED, FLUSH> sink{eps, &list_n, list_key, o.rec_w + seg_base + flushed, room, tsum, 0xffffffffu, 0.0,
                                                      CACHED ? o.row_cache + (size_t)walker * (p.nsd + 1) : nullptr};
    if constexpr (FLUSH) sink.pause_at = P - (kBlock / 64) * kMaxKeptPerTile - 64;  // (every wave may be in the middle of a tile)
    const bool exhausted =
        visit_tiles<LEN, T, decltype(sink), true, list_quarter(SAMPLED)>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
    sink.flush();
    if (FLUSH && exhausted && lane == 0) s_done = 1u;  // (a wave that found no tile left: every tile has been taken, and finished by the barrier)
  }
  __syncthreads();
  if (o.debug & 32u) return;  // (timing ablation: the enumeration alone)
  // ---- the kept columns: sort by column, write, resolve ----
  const bool last = !FLUSH || s_done != 0u;
  const uint32_t ntot = list_n;
  PYNQS_STAMP(2);
  if constexpr (FLUSH) PYNQS_STAMP_ADD(10);  // enumeration (with the wait for the slowest wave)
  const uint32_t n = min(ntot, room);
  needed += ntot;
  if constexpr (!FLUSH) {
    if (tid == 0) {
      o.seg_count[slot] = (int32_t)(ntot > o.fixed ? ntot - o.fixed : 0u);
      if (ntot > cap) {
        atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 1u);
        atomicMax(o.counters + 2, (int32_t)(ntot - o.fixed));
      }
    }
    for (uint32_t i = n + tid; i < o.fixed; i += kBlock) o.rec_col[seg_base + i] = -1;
  }
  for (uint32_t i = n + tid; i < P; i += kBlock) list_key[i] = ~0ull;
  __syncthreads();
  PYNQS_STAMP(3);
  uint32_t Ps = 64;  // sort only as many entries as there are
  while (Ps < n) Ps <<= 1;
  for (uint32_t k = 2; k <= Ps; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < Ps; i += kBlock) {
        const uint32_t ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = list_key[i], b = list_key[ixj];
          if ((a > b) == ((i & k) == 0)) {
            list_key[i] = b; list_key[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  PYNQS_STAMP(4);
  if constexpr (FLUSH) PYNQS_STAMP_ADD(11);  // sort
  // the values, from their order of arrival into the sorted order: every thread fetches its entries' values, then (barrier) stores them
  const int64_t out_base = seg_base + flushed;
  {
    constexpr int kMaxPer = 8;  // n <= 2048 = 8 x 256
    T mine_w[kMaxPer];
#pragma unroll
    for (int r = 0; r < kMaxPer; ++r) {
      const uint32_t i = (uint32_t)r * kBlock + tid;
      mine_w[r] = i < n ? o.rec_w[out_base + (uint32_t)(list_key[i] & 0xffffffffull)] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kMaxPer; ++r) {
      const uint32_t i = (uint32_t)r * kBlock + tid;
      if (i < n) o.rec_w[out_base + i] = mine_w[r];
    }
  }
  if constexpr (FLUSH) PYNQS_STAMP_ADD(12);  // values into sorted order
  for (uint32_t i0 = 0; i0 < n; i0 += kBlock) {
    const uint32_t i = i0 + tid;
    bool won = false;
    int32_t link = -1;
    uint64_t ket[LEN];
#pragma unroll
    for (int w = 0; w < LEN; ++w) ket[w] = wk.w[w];
    if (i < n) {
      const uint32_t col = (uint32_t)(list_key[i] >> 32) & (FLUSH ? 0x7fffffffu : 0xffffffffu);
      if (col) {
        const Excitation x = decode(col - 1, p, L);
        make_ket<LEN>(wk, x, ket);
      }
      const int64_t g = out_base + i;
      o.rec_col[g] = (int32_t)col;
      if (o.rec_onv) {
#pragma unroll
        for (int w = 0; w < LEN; ++w) o.rec_onv[g * LEN + w] = ket[w];
      }
      link = probe_amplitude<LEN, T>(o, ket, won, &s_full);
    }
    const int32_t mine = allocate_batch<LEN, T>(o, p.sorb, won, (uint32_t)link, ket, &bw_cnt, &bw_base);
    if (i < n) o.rec_link[out_base + i] = final_link<LEN, T>(o, link, mine);
  }
  flushed += n;
  if constexpr (FLUSH) PYNQS_STAMP_ADD(13);  // kets, probes, rows, links
  if (last) break;
  __syncthreads();  // (everybody is done with the list)
  if (tid == 0) list_n = 0;
  __syncthreads();
  }
  if constexpr (FLUSH) {
    if (tid == 0) {
      o.seg_count[slot] = (int32_t)(needed > o.fixed ? needed - o.fixed : 0u);
      if (needed > cap) {
        atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 1u);
        atomicMax(o.counters + 2, (int32_t)(needed - o.fixed));
      }
    }
    for (uint32_t i = flushed + tid; i < o.fixed; i += kBlock) o.rec_col[seg_base + i] = -1;
  }
  PYNQS_STAMP(5);
  if constexpr (SAMPLED) {
    const uint32_t ncomb = p.nsd + 1;
    const T *__restrict__ hrow = CACHED ? o.row_cache + (size_t)walker * ncomb : nullptr;
    if constexpr (CACHED) {
      // sums of the sub-eps |H| per COLUMN tile of 256 from the cached row (written by this workgroup, the barriers above make it
      // visible): wave w takes tiles w, w + 4, ...; fixed order of additions
      const uint32_t nct = (ncomb + kOneTileCols - 1) / kOneTileCols;
      for (uint32_t t = wave; t < nct; t += kBlock / 64) {
        double sl = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t c = t * kOneTileCols + lane * 4 + j;
          const T a = c < ncomb ? fabs(hrow[c]) : T(0);
          sl += a >= eps ? 0.0 : (double)a;
        }
        sl = op_wave_sum(sl);
        if (lane == 0) tsum[t] = sl;
      }
      __syncthreads();
    }
    PYNQS_STAMP(6);
    // ---- phase B (as in the look-back form) ----
    const uint32_t per = (max_tiles + kBlock - 1) / kBlock;
    const uint32_t b0 = min((uint32_t)tid * per, max_tiles), b1 = min(b0 + per, max_tiles);
    double local = 0.0;
    for (uint32_t i = b0; i < b1; ++i) local += tsum[i];
    double incl = op_scan(local, lane);
    if (lane == 63) s_part[wave] = incl;
    for (uint32_t i = tid; i < nsample; i += kBlock) pend[i] = 0xffffffffu;  // (the list is done with)
    __syncthreads();
    double before = 0.0, total = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) {
      if (w < wave) before += s_part[w];
      total += s_part[w];
    }
    double run = before + incl - local;
    for (uint32_t i = b0; i < b1; ++i) { run += tsum[i]; tsum[i] = run; }
    __syncthreads();
    const double Srow = total;
    if (tid == 0 && o.row_sum) o.row_sum[walker] = Srow;
    const uint64_t key = op_mix64((o.seed_dev ? seed + *o.seed_dev : seed) ^ op_mix64(slot));
    if (Srow > 0.0) {
      for (uint32_t k = tid; k < nsample; k += kBlock) {
        const uint64_t r = op_mix64(key ^ op_mix64(0xffffffff00000000ull | k));
        const double target = (double)(r >> 11) * 0x1.0p-53 * Srow;
        uint32_t lo = 0, hi = max_tiles;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (tsum[mid] > target) hi = mid; else lo = mid + 1;
        }
        if (lo >= max_tiles) lo = max_tiles - 1;
        while (lo > 0 && !(tsum[lo] > tsum[lo - 1])) --lo;
        atomicAdd(&dinfo[lo], 1u);
      }
    }
    __syncthreads();
    uint32_t lsum = 0;
    for (uint32_t i = b0; i < b1; ++i) lsum += dinfo[i];
    uint32_t iscan = lsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t ov = __shfl_up(iscan, d);
      if (lane >= d) iscan += ov;
    }
    if (lane == 63) s_parti[wave] = iscan;
    __syncthreads();
    uint32_t ibefore = 0;
    for (int w = 0; w < wave; ++w) ibefore += s_parti[w];
    uint32_t off = ibefore + iscan - lsum;
    for (uint32_t i = b0; i < b1; ++i) {
      const uint32_t c = dinfo[i];
      dinfo[i] = (off << 16) | c;
      off += c;
    }
    uint32_t *tlist = pend + nsample;  // GTILE: the drawn tiles in ascending order (LDS, after the draw slots' columns)
    uint32_t ndrawn = 0;
    if constexpr (GTILE) {
      uint32_t nd = 0;
      for (uint32_t i = b0; i < b1; ++i) nd += (dinfo[i] & 0xffffu) ? 1u : 0u;  // (this thread's own stores above)
      uint32_t dscan = nd;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ov = __shfl_up(dscan, d);
        if (lane >= d) dscan += ov;
      }
      __syncthreads();  // (s_parti is free again)
      if (lane == 63) s_parti[wave] = dscan;
      __syncthreads();
      uint32_t dbefore = 0;
      for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wave) dbefore += s_parti[w];
        ndrawn += s_parti[w];
      }
      uint32_t doff = dbefore + dscan - nd;
      for (uint32_t i = b0; i < b1; ++i)
        if (dinfo[i] & 0xffffu) tlist[doff++] = i;
    }
    if (tid == 0) next_tile = 0;
    __syncthreads();
    PYNQS_STAMP(7);
    // ---- phase C: the draws inside the tiles ----
    unsigned char *mine = draw0 + (size_t)wave * (CACHED ? kCachedDrawLdsPerWave : kDrawLdsPerWave);
    DrawLds S;
    S.prefix = reinterpret_cast<double *>(mine);
    S.run = reinterpret_cast<volatile double *>(mine + (size_t)kOneTileCols * 8);
    S.cs = reinterpret_cast<uint32_t *>(mine + (size_t)kOneTileCols * 8 + 8);
    S.hits = S.cs + kOneTileCols;
    S.ncols = reinterpret_cast<volatile uint32_t *>(S.hits + kOneTileCols);
    if constexpr (CACHED) {
      // the draws inside the column tiles, from the cached row: no second enumeration.  A wave pulls a tile, loads its 256 matrix
      // elements (4 per lane), forms the running sums with ONE wave scan and hands them to the same draw / hit-count / emission code
      const uint32_t nct = (ncomb + kOneTileCols - 1) / kOneTileCols;
      const double scale = Srow / (double)nsample;
      const int64_t sbase = (int64_t)walker * nsample;
      for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&next_tile, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nct) break;
        const uint32_t info = dinfo[t], draws = info & 0xffffu;
        if (draws == 0 || (o.debug & 64u)) continue;
        const uint32_t c0 = t * kOneTileCols + lane * 4;
        double w4[4];
        uint32_t neg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T h = c0 + j < ncomb ? hrow[c0 + j] : T(0);
          const T a = fabs(h);
          w4[j] = a >= eps ? 0.0 : (double)a;
          neg |= (h < T(0) ? 1u : 0u) << j;
        }
        const double ls = (w4[0] + w4[1]) + (w4[2] + w4[3]);
        const double incl = op_scan(ls, lane);
        const double total = __shfl(incl, 63);
        double run = incl - ls;
        __builtin_amdgcn_wave_barrier();
        uint32_t *hits2 = reinterpret_cast<uint32_t *>(mine + (size_t)kOneTileCols * 8);   // [cols / 2]: two 16-bit counts per word
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          run += w4[j];
          S.prefix[lane * 4 + j] = run;
        }
        hits2[lane * 2] = 0u; hits2[lane * 2 + 1] = 0u;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ncols = min((uint32_t)kOneTileCols, ncomb - t * kOneTileCols);
        if (!(total > 0.0)) continue;
        for (uint32_t k = lane; k < draws; k += 64) {
          const uint64_t r = op_mix64(key ^ op_mix64(((uint64_t)t << 32) | k));
          const double target = (double)(r >> 11) * 0x1.0p-53 * total;
          uint32_t lo = 0, hi = ncols;
          if (o.debug & 128u) { lo = (uint32_t)(r % ncols); hi = lo; }
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (S.prefix[mid] > target) hi = mid; else lo = mid + 1;
          }
          if (lo >= ncols) lo = ncols - 1;
          while (lo > 0 && !(S.prefix[lo] > S.prefix[lo - 1])) --lo;
          atomicAdd(&hits2[lo >> 1], 1u << (16u * (lo & 1u)));  // (a column is drawn < 2^16 times: nsample < 2^16)
        }
        __builtin_amdgcn_wave_barrier();
        if (o.debug & 256u) continue;
        // emission: lane l looks at the four columns it loaded (4 l .. 4 l + 3: their signs are still in its registers), one scan over
        // the lanes places them -- ascending columns, as a pass of 64 columns at a time with a ballot each produced them (4 passes: 140
        // instead of ~80 instructions per tile)
        const uint32_t h01 = hits2[lane * 2], h23 = hits2[lane * 2 + 1];
        const uint32_t hc[4] = {h01 & 0xffffu, h01 >> 16, h23 & 0xffffu, h23 >> 16};
        const uint32_t mine = (hc[0] ? 1u : 0u) + (hc[1] ? 1u : 0u) + (hc[2] ? 1u : 0u) + (hc[3] ? 1u : 0u);
        uint32_t before = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t ov = __shfl_up(before, d);
          if (lane >= d) before += ov;
        }
        uint32_t at = (info >> 16) + before - mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (hc[j]) {
            const uint32_t col = c0 + j;
            o.srec_col[sbase + at] = (int32_t)col;
            const double v = scale * (double)hc[j];
            o.srec_w[sbase + at] = (T)(((neg >> j) & 1u) ? -v : v);
            pend[at] = col;
            ++at;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else if (!(o.debug & 16u)) {
      ListDrawSink<LEN, T> sink{eps, S, dinfo, Srow / (double)nsample, key, (int64_t)walker * nsample, o.srec_col, o.srec_w, pend, 0xffffffffu};
      sink.nodraw = (o.debug & 8u) != 0;
      if constexpr (GTILE) { sink.tlist = tlist; sink.ndrawn = ndrawn; }
      visit_tiles<LEN, T, decltype(sink), true, list_quarter(SAMPLED)>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
      sink.flush();
    }
    __syncthreads();
    PYNQS_STAMP(8);
    // ---- the drawn records: kets, links, rows -- four draw slots per thread and round, one row allocation per round ----
    constexpr int K = 4;
    for (uint32_t i0 = 0; i0 < nsample; i0 += K * kBlock) {
      constexpr int32_t kNoRecord = -0x7fffffff;
      bool won[K];
      uint32_t slot[K];
      int32_t lk[K];
      uint64_t ket[K][LEN];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t i = i0 + k * kBlock + tid;
        won[k] = false;
        slot[k] = 0;
        lk[k] = kNoRecord;
#pragma unroll
        for (int w = 0; w < LEN; ++w) ket[k][w] = wk.w[w];
        const uint32_t col = i < nsample ? pend[i] : 0xffffffffu;
        if (col != 0xffffffffu) {
          if (col) {
            const Excitation x = decode(col - 1, p, L);
            make_ket<LEN>(wk, x, ket[k]);
          }
          const int64_t at = (int64_t)walker * nsample + i;
          if (o.srec_onv) {
#pragma unroll
            for (int w = 0; w < LEN; ++w) o.srec_onv[at * LEN + w] = ket[k][w];
          }
          lk[k] = probe_amplitude<LEN, T>(o, ket[k], won[k]);
          slot[k] = (uint32_t)lk[k];
        }
      }
      int32_t mine[K];
      allocate_batch_k<LEN, T, K>(o, p.sorb, won, slot, ket, &bw_cnt, &bw_base, mine);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t i = i0 + k * kBlock + tid;
        if (lk[k] != kNoRecord) o.srec_link[(int64_t)walker * nsample + i] = final_link<LEN, T>(o, lk[k], mine[k]);
      }
    }
    PYNQS_STAMP(9);
  }
}

// The kernels.  Eight waves per SIMD (64 VGPRs, 96 SGPRs; the 106 scalar registers the compiler would otherwise take cap the CU at SIX
// workgroups -- measured, tools/onepass_stamps.py -- whatever the LDS allows) for the forms whose LDS fits eight workgroups per CU: 8192
// walkers are 32 workgroups per CU, i.e. exactly four generations of eight.  The form that enumerates the drawn tiles a second time
// (no row cache: ~38 KB of LDS, four workgroups per CU) keeps its registers.
template <int LEN, typename T, bool SAMPLED, bool CACHED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
void reduce_onepass_list_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks, uint32_t chunk_len, uint32_t max_tiles,
                                const T *__restrict__ plan, T eps, uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  static_assert(CACHED || !SAMPLED, "the re-enumerating form has its own kernel");
  reduce_onepass_list_body<LEN, T, SAMPLED, CACHED>(bra, p, pl, nchunks, chunk_len, max_tiles, plan, eps, nsample, seed, P, o);
}

// rows whose kept columns do not fit the list (deterministic): the list is emptied as it fills (FLUSH); ~40 KB of LDS at sorb 120, four
// workgroups per CU: no register squeeze
template <int LEN, typename T, bool SAMPLED = false, bool GTILE = false>
__global__ __launch_bounds__(kBlock) void reduce_onepass_list_flush_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks,
                                                                           uint32_t chunk_len, uint32_t max_tiles, const T *__restrict__ plan, T eps,
                                                                           uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  reduce_onepass_list_body<LEN, T, SAMPLED, false, true, GTILE>(bra, p, pl, nchunks, chunk_len, max_tiles, plan, eps, nsample, seed, P, o);
}

template <int LEN, typename T, bool GTILE = false>
__global__ __launch_bounds__(kBlock) void reduce_onepass_list_redraw_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks,
                                                                            uint32_t chunk_len, uint32_t max_tiles, const T *__restrict__ plan, T eps,
                                                                            uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  reduce_onepass_list_body<LEN, T, true, false, false, GTILE>(bra, p, pl, nchunks, chunk_len, max_tiles, plan, eps, nsample, seed, P, o);
}

// ---- contraction: E_loc(x) = sum_records w psi(x') / psi(x) ------------------------------------------------------------
// One wave per walker.  Slots are visited in their fixed order (fixed slots, compacted doubles, drawn records; chunk by chunk),
// lane l takes slots l, l + 64, ...; the 64 partial sums meet in a butterfly: the result does not depend on anything but the
// records' positions.  psi(x) is the amplitude of the record of column 0 (kept slot 0, else among the drawn ones; 0 if it is
// nowhere -- the reference divides by zero there, too).
template <typename T, bool CPLX>
__global__ __launch_bounds__(kBlock) void reduce_contract_kernel(int64_t nbatch, uint32_t nchunks, uint32_t fixed, uint32_t cap_d,
                                                                 uint32_t nsample, const int32_t *__restrict__ rec_col,
                                                                 const T *__restrict__ rec_w, const int32_t *__restrict__ rec_link,
                                                                 const int32_t *__restrict__ seg_count, const int32_t *__restrict__ srec_col,
                                                                 const T *__restrict__ srec_w, const int32_t *__restrict__ srec_link,
                                                                 const int32_t *__restrict__ dedup_i32, int slot_i32, int row_off, uint32_t ucap,
                                                                 const double *__restrict__ psi_u, const double *__restrict__ psi_t, int divide,
                                                                 double *__restrict__ eloc, double *__restrict__ psi_x) {
  const int lane = threadIdx.x & 63;
  const int64_t walker = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (walker >= nbatch) return;
  double ar = 0.0, ai = 0.0, xr = 0.0, xi = 0.0;
  bool bad = false;
  auto amp = [&](int32_t link, double &re, double &im) {
    const double *src;
    int64_t at;
    if (link >= kDirectLink) {
      src = psi_u; at = link - kDirectLink;
      if ((uint32_t)at >= ucap) { bad = true; re = im = 0.0; return; }
    } else if (link >= 0) {
      if (!dedup_i32) { bad = true; re = im = 0.0; return; }  // (no de-duplication table: every link is direct)
      const int32_t row = dedup_i32[(int64_t)link * slot_i32 + row_off];
      if (row < 0 || (uint32_t)row >= ucap) { bad = true; re = im = 0.0; return; }
      src = psi_u; at = row;
    } else if (link <= -2) {
      src = psi_t; at = -2 - (int64_t)link;
    } else { bad = true; re = im = 0.0; return; }
    if constexpr (CPLX) { re = src[2 * at]; im = src[2 * at + 1]; }
    else { re = src[at]; im = 0.0; }
  };
  auto visit = [&](int32_t col, double w, int32_t link) {
    if (col < 0) return;
    double re, im;
    amp(link, re, im);
    ar += w * re; ai += w * im;
    if (col == 0) { xr = re; xi = im; }
  };
  const int64_t stride = (int64_t)fixed + cap_d;
  for (uint32_t c = 0; c < nchunks; ++c) {
    const int64_t seg = walker * nchunks + c, base = seg * stride;
    const uint32_t kept = (uint32_t)seg_count[seg];
    if (kept > cap_d) bad = true;
    const uint32_t nslots = fixed + min(kept, cap_d);
    for (uint32_t i = lane; i < nslots; i += 64) visit(rec_col[base + i], (double)rec_w[base + i], rec_link[base + i]);
  }
  for (uint32_t i = lane; i < nsample; i += 64) {
    const int64_t at = walker * nsample + i;
    visit(srec_col[at], (double)srec_w[at], srec_link[at]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    ar += __shfl_xor(ar, d); ai += __shfl_xor(ai, d);
    xr += __shfl_xor(xr, d); xi += __shfl_xor(xi, d);  // (exactly one lane holds a non-zero psi(x))
  }
  const bool anybad = __ballot(bad) != 0;
  if (lane == 0) {
    if (anybad) { ar = ai = __builtin_nan(""); }
    if constexpr (CPLX) {
      const double dn = xr * xr + xi * xi;
      eloc[2 * walker] = divide ? (ar * xr + ai * xi) / dn : ar;
      eloc[2 * walker + 1] = divide ? (ai * xr - ar * xi) / dn : ai;
      psi_x[2 * walker] = xr; psi_x[2 * walker + 1] = xi;
    } else {
      eloc[walker] = divide ? ar / xr : ar;
      psi_x[walker] = xr;
    }
  }
}

}  // namespace pynqs

using namespace pynqs;

static int onepass_geometry(int64_t nbatch, const SDParams &p, bool sampled, uint32_t *nchunks, uint32_t *chunk_len, uint32_t *max_tiles,
                            uint32_t *fixed) {
  const uint32_t ncomb = p.nsd + 1;
  if (sampled) {  // the draws need the sums of the WHOLE row in one workgroup's LDS
    *nchunks = 1;
    *chunk_len = (ncomb + 255u) & ~255u;
  } else {
    plan_chunks(nbatch, ncomb, nchunks, chunk_len);
  }
  *max_tiles = max_tiles_per_chunk(p, *nchunks, *chunk_len);
  const uint32_t tS_all = (p.d1 + kSinglesPerTile - 1) / kSinglesPerTile;
  *fixed = kFixedHead + ((tS_all + *nchunks - 1) / *nchunks) * kSinglesPerTile;
  return 0;
}

static size_t onepass_static_lds(int) { return 16 + (kBlock / 64) * 4 + (kBlock / 64 + 1) * 12 + 64; }

extern "C" int pynqs_reduce_onepass_geometry(int64_t nbatch, int sorb, int nele, int noA, int noB, int eps_sample, int64_t *out4) {
  SDParams p;
  PlanLayout pl;
  if (!out4) return set_error(PYNQS_EINVAL, "null pointer");
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB (even sorb in [2, 192])");
  if (nbatch < 0 || nbatch > 0x7fffffffll || eps_sample < 0 || eps_sample > 65535) return set_error(PYNQS_EINVAL, "bad nbatch / eps_sample (at most 65535 draws)");
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, eps_sample > 0, &nchunks, &chunk_len, &max_tiles, &fixed);
  const int len = (sorb - 1) / 64 + 1;
  const int64_t slots = out4[2];
  out4[0] = nbatch * (int64_t)nchunks;
  out4[1] = fixed;
  out4[2] = slots > 0 ? slots * dedup_slot_words(len) * 8 : 0;
  // the look-back form, or (draws on long rows, io->tile_scratch given) the flushing LIST form with its tile sums in global memory
  const bool lookback = onepass_lds(p, 8, max_tiles, eps_sample > 0, winner_list_cap(eps_sample)) + onepass_static_lds(len) <= 160 * 1024;
  const bool flushing = eps_sample > 0 && p.nsd + 1 > 65536 &&
                        onepass_list_lds(p, 8, max_tiles, true, 2048, (uint32_t)eps_sample, false, true) + 256 + onepass_static_lds(len) <= 160 * 1024;
  out4[3] = lookback || flushing ? 1 : 0;
  return PYNQS_OK;
}

// Which of the two forms a call takes.  LIST form when a segment's records fit an LDS list (PYNQS_OP_LIST=0 / 1 overrides; 1 only where it
// fits): up to 1024 slots per segment, up to 2048 on rows of more than kLongRow columns -- there the look-back form is the slow one (sorb 80
// / 120: 2 to 4 times slower than the multi-pass entry points, tools/reduce_big_paths.py; energy.py routes such calls away from it), on short
// rows a list of 2048 costs more than it saves.  With a row cache (io->row_cache: [nbatch][ncomb] elements of the integral dtype) the draws
// read the row back instead of visiting the drawn tiles a second time (PYNQS_OP_CACHE=0 ignores the buffer); LIST form only.
constexpr uint32_t kLongRow = 65536;
struct OnepassForm {
  uint32_t P;
  bool use_list, use_cache, use_flush, use_gtile;
  size_t lds;
};
constexpr uint32_t kFlushList = 2048;  // list slots of the flushing form
static OnepassForm onepass_form(const SDParams &p, size_t esz, uint32_t max_tiles, uint32_t fixed, uint64_t cap_doubles, int eps_sample,
                                bool have_cache, uint32_t chunk_len, bool no_table, bool have_tile_scratch = false) {
  static const int list_env = getenv("PYNQS_OP_LIST") ? atoi(getenv("PYNQS_OP_LIST")) : -1;
  static const int cache_env = getenv("PYNQS_OP_CACHE") ? atoi(getenv("PYNQS_OP_CACHE")) : -1;
  static const int flush_env = getenv("PYNQS_OP_FLUSH") ? atoi(getenv("PYNQS_OP_FLUSH")) : -1;
  const bool sampled = eps_sample > 0;
  const uint64_t seg_cap = (uint64_t)fixed + cap_doubles;
  OnepassForm f;
  f.P = 64;
  while (f.P < seg_cap && f.P < (1u << 20)) f.P <<= 1;
  const bool want_cache = sampled && have_cache && cache_env != 0;
  const bool gtile = sampled && !want_cache && have_tile_scratch;  // tile sums / draw counts in global memory (long rows)
  const size_t lds_list = onepass_list_lds(p, esz, max_tiles, sampled, f.P, (uint32_t)eps_sample, want_cache, gtile);
  const bool list_fits = seg_cap <= 2048 && lds_list + 256 <= 160 * 1024;
  f.use_list = list_env == 0 ? false : (list_fits && (list_env == 1 || seg_cap <= 1024 || p.nsd + 1 > kLongRow));
  f.use_cache = want_cache && f.use_list;
  f.use_gtile = gtile && f.use_list;
  f.lds = f.use_list ? lds_list : onepass_lds(p, esz, max_tiles, sampled, winner_list_cap(eps_sample));
  // the flushing LIST form: deterministic calls whose kept columns do not fit the list -- on long rows, on any row when at most a tenth
  // of a segment's columns can be kept (a flush costs a sort of 2048 entries; the look-back form pays per tile instead: Fe2S2 with 9 % kept
  // 0.81 look-back against 2.45 ms, sorb 56 with 6 % 6.5 against 3.7, sorb 80 with 10 % / 40 % 103 / 292 against 71 / 269), and whenever
  // there is no de-duplication table (the look-back form needs one).  PYNQS_OP_FLUSH=0 / 1: never / wherever possible.
  // With draws the kept list is flushed during the enumeration in the same way, under the same rule; the draws follow as before (sorb 56,
  // 200 draws, 2 % / 6 % kept: 6.2 / 7.1 ms look-back -> 1.6 / 2.2 flushing and table-less; Fe2S2, 9 %: 1.37 stays).  A row cache is not
  // used by this form: when the LIST form with the cache does not fit, flushing without it beats the look-back form (sorb 56, 1000 draws:
  // 6.2 / 3.8 -> 2.6 / 1.9 ms).
  const bool gtile_f = sampled && have_tile_scratch;
  const size_t lds_flush = onepass_list_lds(p, esz, max_tiles, sampled, kFlushList, (uint32_t)eps_sample, false, gtile_f);
  const bool long_row = p.nsd + 1 > kLongRow;
  f.use_flush = !f.use_list && flush_env != 0 && lds_flush + 256 <= 160 * 1024 &&
                (long_row || flush_env == 1 || no_table || cap_doubles * 10 <= (uint64_t)chunk_len);
  if (f.use_flush) { f.use_gtile = gtile_f; f.use_cache = false; }
  if (f.use_flush) { f.P = kFlushList; f.lds = lds_flush; }
  return f;
}

extern "C" int64_t pynqs_reduce_onepass_tile_scratch_bytes(int64_t nbatch, int sorb, int nele, int noA, int noB, int eps_sample) {
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl) || nbatch < 0 || nbatch > 0x7fffffffll || eps_sample < 0 || eps_sample > 65535) {
    set_error(PYNQS_EINVAL, "bad arguments");
    return -1;
  }
  if (eps_sample == 0) return 0;
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, true, &nchunks, &chunk_len, &max_tiles, &fixed);
  return (int64_t)((size_t)nbatch * tile_scratch_stride(max_tiles));
}

extern "C" int pynqs_reduce_onepass_list_capacity(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample,
                                                  int with_row_cache, int without_table, int64_t *cap_doubles) {
  SDParams p;
  PlanLayout pl;
  if (!cap_doubles) return set_error(PYNQS_EINVAL, "null pointer");
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB (even sorb in [2, 192])");
  if (nbatch < 0 || nbatch > 0x7fffffffll || eps_sample < 0 || eps_sample > 65535 || (dtype != PYNQS_F32 && dtype != PYNQS_F64))
    return set_error(PYNQS_EINVAL, "bad nbatch / eps_sample / dtype");
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, eps_sample > 0, &nchunks, &chunk_len, &max_tiles, &fixed);
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  auto listed = [&](uint64_t cap) {
    // (with draws and no row cache the caller is expected to pass io->tile_scratch: pynqs_reduce_onepass_tile_scratch_bytes)
    const OnepassForm f = onepass_form(p, esz, max_tiles, fixed, cap, eps_sample, with_row_cache != 0, chunk_len, without_table != 0,
                                       eps_sample > 0 && with_row_cache == 0);
    return (f.use_list || f.use_flush) && f.lds + onepass_static_lds(0) <= 160 * 1024;
  };
  *cap_doubles = -1;
  // (the forms are monotonic in the capacity: LIST up to a limit, then possibly the flushing form up to another, or without one)
  if (listed((uint64_t)1 << 29)) { *cap_doubles = ((int64_t)1 << 30) - 1; return PYNQS_OK; }
  for (uint32_t seg = 2048; seg >= 128; seg >>= 1) {
    if (seg < fixed) break;
    if (listed(seg - fixed)) { *cap_doubles = (int64_t)(seg - fixed); break; }
  }
  if ((int64_t)(chunk_len / 10) > *cap_doubles && listed(chunk_len / 10)) *cap_doubles = chunk_len / 10;
  return PYNQS_OK;
}

template <typename T>
static OnepassOut<T> make_out(const pynqs_reduce_io *io, int len, uint32_t fixed, uint32_t gtile_max_tiles = 0) {
  OnepassOut<T> o;
  o.rec_col = io->rec_col; o.rec_w = (T *)io->rec_w; o.rec_onv = io->rec_onv; o.rec_link = io->rec_link; o.seg_count = io->seg_count;
  o.srec_col = io->srec_col; o.srec_w = (T *)io->srec_w; o.srec_onv = io->srec_onv; o.srec_link = io->srec_link; o.row_sum = io->row_sum;
  o.dedup = (uint64_t *)io->dedup_table; o.dedup_mask = io->dedup_table ? (uint32_t)(io->dedup_slots - 1) : 0u;
  o.lut = (const uint64_t *)io->lut_table; o.lut_cap = io->lut_table ? hash_capacity(io->lut_nkeys) : 0;
  o.uniq_parent = io->uniq_parent; o.parent = 0;
  o.uniq_onv = io->uniq_onv; o.uniq_pm1 = io->uniq_pm1; o.pm1_f32 = io->pm1_dtype == PYNQS_F32; o.ucap = (uint32_t)io->cap_unique;
  o.counters = io->counters; o.cap_d = (uint32_t)io->cap_doubles; o.fixed = fixed;
  static const uint32_t dbg = getenv("PYNQS_OP_DEBUG") ? (uint32_t)atoi(getenv("PYNQS_OP_DEBUG")) : 0u;
  o.debug = dbg;
  o.seed_dev = io->seed_dev;
  o.row_cache = (T *)io->row_cache;
  if (gtile_max_tiles) { o.tile_scratch = (unsigned char *)io->tile_scratch; o.tile_stride = (uint32_t)tile_scratch_stride(gtile_max_tiles); }
  (void)len;
  return o;
}

extern "C" int pynqs_reduce_onepass(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB, const void *plan, int dtype,
                                    double eps, int eps_sample, uint64_t seed, const pynqs_reduce_io *io, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB (even sorb in [2, 192])");
  if (nbatch < 0 || nbatch > 0x7fffffffll || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return set_error(PYNQS_EINVAL, "bad nbatch/dtype");
  if (eps_sample < 0 || eps_sample > 65535) return set_error(PYNQS_EINVAL, "eps_sample must be in [0, 65535]");
  if (!io) return set_error(PYNQS_EINVAL, "null pointer");
  const bool sampled = eps_sample > 0;
  if (!io->counters || !io->uniq_onv || !io->rec_col || !io->rec_w || !io->rec_link || !io->seg_count ||
      (sampled && (!io->srec_col || !io->srec_w || !io->srec_link)))
    return set_error(PYNQS_EINVAL, "null pointer");
  if (io->cap_unique < 1 || io->cap_unique >= (1ll << 30) || io->cap_doubles < 0 || io->cap_doubles >= (1ll << 30) ||
      (io->dedup_table && (io->dedup_slots < 64 || (io->dedup_slots & (io->dedup_slots - 1)) || io->dedup_slots > (1ll << 30) ||
                           2 * io->cap_unique > io->dedup_slots)))
    return set_error(PYNQS_EINVAL, "bad capacities (dedup_slots: power of two >= 2 * cap_unique)");
  if (io->pm1_dtype != PYNQS_F32 && io->pm1_dtype != PYNQS_F64) return set_error(PYNQS_EINVAL, "bad pm1_dtype");
  if (io->lut_table && io->lut_nkeys < 0) return set_error(PYNQS_EINVAL, "bad lut_nkeys");
  hipStream_t st = (hipStream_t)stream;
  const int len = (sorb - 1) / 64 + 1;
  if (hipMemsetAsync(io->counters, 0, 16, st) != hipSuccess) return check_launch("memset");
  if (io->dedup_table &&
      hipMemsetAsync(io->dedup_table, 0xFF, (size_t)io->dedup_slots * dedup_slot_words(len) * 8, st) != hipSuccess) return check_launch("memset");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, sampled, &nchunks, &chunk_len, &max_tiles, &fixed);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large");
  if (grid * ((uint64_t)fixed + (uint64_t)io->cap_doubles) > 0x7fffffffull * 16ull) return set_error(PYNQS_EINVAL, "record arrays too large");
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  const bool have_tiles = io->tile_scratch != nullptr && io->tile_scratch_bytes >= (int64_t)((size_t)nbatch * tile_scratch_stride(max_tiles));
  const OnepassForm form = onepass_form(p, esz, max_tiles, fixed, (uint64_t)io->cap_doubles, eps_sample, io->row_cache != nullptr, chunk_len,
                                        io->dedup_table == nullptr, have_tiles);
  const uint32_t P = form.P;
  const bool use_list = form.use_list || form.use_flush, use_cache = form.use_cache, use_flush = form.use_flush, use_gtile = form.use_gtile;
  const size_t lds = form.lds;
  if (lds + onepass_static_lds(len) > 160 * 1024) return set_error(PYNQS_EINVAL, "row too long for the fused form (LDS): use the multi-pass entry points");
  if (!io->dedup_table && !use_list)
    return set_error(PYNQS_EINVAL, "no de-duplication table: only the LIST forms run without one (cap_doubles <= pynqs_reduce_onepass_list_capacity)");
  static const bool verbose = getenv("PYNQS_OP_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "pynqs_reduce_onepass: %s form%s, LDS %zu bytes per workgroup (walker tables %zu, max_tiles %u, list P %u), %u chunk(s) per walker\n",
            use_flush ? "flushing LIST" : use_list ? (use_gtile ? "LIST (tile sums in global memory)" : "LIST") : "look-back", use_cache ? " with row cache" : "", lds, (size_t)lds_fixed_bytes(p), max_tiles, P, nchunks);
  // eloc.py:257-264: with draws and eps <= 0 nothing is kept (every column can be drawn); without draws |H| >= eps as it stands
  const double eps_eff = (sampled && !(eps > 0.0)) ? __builtin_inf() : eps;
#define PYNQS_OP_LAUNCH(TT, SM)                                                                                                      \
  do {                                                                                                                               \
    auto kfn = use_flush ? (use_gtile ? reduce_onepass_list_flush_kernel<LEN, TT, SM, SM> : reduce_onepass_list_flush_kernel<LEN, TT, SM, false>) \
             : use_list ? (use_cache ? reduce_onepass_list_kernel<LEN, TT, SM, SM>                                                   \
                                     : (SM ? (use_gtile ? reduce_onepass_list_redraw_kernel<LEN, TT, true> : reduce_onepass_list_redraw_kernel<LEN, TT, false>) \
                                           : reduce_onepass_list_kernel<LEN, TT, false, false>)) \
                        : reduce_onepass_kernel<LEN, TT, SM>;                                                                        \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                               (int)lds) != hipSuccess)                                                              \
      return check_launch("hipFuncSetAttribute");                                                                                   \
    hipLaunchKernelGGL(kfn, dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks, chunk_len, max_tiles, (const TT *)plan, \
                       (TT)eps_eff, (uint32_t)eps_sample, seed, use_list ? P : winner_list_cap(eps_sample), make_out<TT>(io, len, fixed, use_gtile ? max_tiles : 0u));                \
  } while (0)
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64) { if (sampled) PYNQS_OP_LAUNCH(double, true); else PYNQS_OP_LAUNCH(double, false); }
    else { if (sampled) PYNQS_OP_LAUNCH(float, true); else PYNQS_OP_LAUNCH(float, false); }
  });
#undef PYNQS_OP_LAUNCH
  return check_launch("reduce_onepass");
}

extern "C" int pynqs_reduce_contract(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample,
                                     const pynqs_reduce_io *io, const double *psi_unique, const double *psi_table, int psi_is_complex,
                                     int divide, double *eloc, double *psi_x, void *stream) {
  pynqs::DeviceScope device_scope_(eloc);
  SDParams p;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (nbatch < 0 || nbatch > 0x7fffffffll || (dtype != PYNQS_F32 && dtype != PYNQS_F64) || eps_sample < 0 || eps_sample > 65535)
    return set_error(PYNQS_EINVAL, "bad nbatch/dtype/eps_sample");
  if (nbatch == 0) return PYNQS_OK;
  if (!io || !io->rec_col || !io->rec_w || !io->rec_link || !io->seg_count || !psi_unique || !eloc || !psi_x ||
      (eps_sample > 0 && (!io->srec_col || !io->srec_w || !io->srec_link)) || (io->lut_table && !psi_table))
    return set_error(PYNQS_EINVAL, "null pointer");
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, eps_sample > 0, &nchunks, &chunk_len, &max_tiles, &fixed);
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t grid = (uint32_t)((nbatch + kBlock / 64 - 1) / (kBlock / 64));
#define PYNQS_CT_LAUNCH(TT, CX)                                                                                                          \
  hipLaunchKernelGGL((reduce_contract_kernel<TT, CX>), dim3(grid), dim3(kBlock), 0, st, nbatch, nchunks, fixed, (uint32_t)io->cap_doubles, \
                     (uint32_t)eps_sample, io->rec_col, (const TT *)io->rec_w, io->rec_link, io->seg_count, io->srec_col,                \
                     (const TT *)io->srec_w, io->srec_link, (const int32_t *)io->dedup_table, dedup_slot_words(len) * 2,                 \
                     dedup_row_offset(len), (uint32_t)io->cap_unique, psi_unique, psi_table, divide, eloc, psi_x)
  if (dtype == PYNQS_F64) { if (psi_is_complex) PYNQS_CT_LAUNCH(double, true); else PYNQS_CT_LAUNCH(double, false); }
  else { if (psi_is_complex) PYNQS_CT_LAUNCH(float, true); else PYNQS_CT_LAUNCH(float, false); }
#undef PYNQS_CT_LAUNCH
  return check_launch("reduce_contract");
}

#ifdef PYNQS_OP_STAMPS
extern "C" int pynqs_debug_stamps(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pynqs::g_stamps), sizeof(unsigned long long) * 8192 * 16);
}
#endif
