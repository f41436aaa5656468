// reduce_common.h -- what the kernels of the one-launch REDUCE front end share (kernels_reduce_onepass.hip, kernels_reduce_rowlds.hip):
// the output block, the de-duplication table, the hand-out of rows of the distinct list, the +-1 rows.
#pragma once
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "plan_tiles.h"

namespace pynqs {

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
  float *row_f32 = nullptr;  // [nbatch][ncomb] or NULL: the row's sub-eps elements as float32 (two-kernel semi-stochastic form)
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
  for (int i = 0; i < LEN; ++i) __builtin_nontemporal_store(ket[i], o.uniq_onv + (size_t)r * LEN + i);  // (streaming: read by the next kernel only)
  if (o.uniq_parent) __builtin_nontemporal_store(o.parent, o.uniq_parent + r);
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

// bytes per walker of io->tile_scratch (tile sums f64 + draw counts u32 per tile)
__host__ __device__ inline size_t tile_scratch_stride(uint32_t max_tiles) { return ((size_t)max_tiles * 12 + 15) & ~(size_t)15; }

// the output block of a launch from the caller's pynqs_reduce_io (host)
template <typename T>
inline OnepassOut<T> make_out(const pynqs_reduce_io *io, int len, uint32_t fixed, uint32_t gtile_max_tiles = 0) {
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
  o.row_f32 = (float *)io->row_f32;
  if (gtile_max_tiles) { o.tile_scratch = (unsigned char *)io->tile_scratch; o.tile_stride = (uint32_t)tile_scratch_stride(gtile_max_tiles); }
  (void)len;
  return o;
}


// ---- the semi-stochastic LIST kernel for rows of up to 8192 columns (kernels_reduce_rowout.hip; reduce_list.h ROWOUT + reduce_draw.h) ----
bool reduce_draw_supported(const SDParams &p, int eps_sample);
int launch_reduce_rowout(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, uint32_t chunk_len, uint32_t max_tiles,
                         const void *plan, int dtype, double eps_eff, int eps_sample, uint64_t seed, uint32_t P, size_t lds,
                         const pynqs_reduce_io *io, uint32_t fixed, hipStream_t st);
// the flushing semi-stochastic form with the row's float32 copy (kernels_reduce_rowout.hip): as above, for rows of any length
int launch_reduce_flush_row32(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, uint32_t chunk_len, uint32_t max_tiles,
                              const void *plan, int dtype, double eps_eff, int eps_sample, uint64_t seed, uint32_t P, size_t lds,
                              const pynqs_reduce_io *io, uint32_t fixed, bool gtile, hipStream_t st);

}  // namespace pynqs
