// reduce_list.h -- the LIST forms of the one-launch REDUCE front end (kernels_reduce_onepass.hip: the kernels and the C ABI;
// kernels_reduce_rowout.hip: the semi-stochastic kernel for rows whose float32 copy goes through global memory, round 4).
#pragma once
#include "reduce_common.h"

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

}  // namespace pynqs
#include "reduce_draw.h"
namespace pynqs {

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


// FLUSH (the flushing LIST form, rows whose kept columns exceed the list): the workgroup empties the list whenever it is nearly full
// (pause(): asked by visit_tiles before a wave takes a tile); entries of tile 0 (column 0 and the few unpaired doubles, whose columns lie
// anywhere in the row) keep bit 63 of the key clear and so sort in front of everything else of the first flush, the others carry it:
// the records of a segment are then the same whatever the timing of the flushes.
// ROWOUT (round 4, the two-kernel semi-stochastic form): the enumeration leaves every sub-eps matrix element of the row as a float32 in
// global memory (io->row_f32; kept columns: 0, they are never drawn) -- a quarter of round 3's row-cache traffic, written once and read
// once by the draw kernel (kernels_reduce_draw.hip) -- and sums the exact |H| per tile in float64 as the re-enumerating form does.
// ROW32 (end of round 4): the flushing semi-stochastic form on long rows leaves the float32 copy of the row too (no bitmap: the kept records
// go through the list and its flushes as before) -- the draws inside the drawn tiles then read their <= 256 columns back instead of
// enumerating the tile a second time.
template <int LEN, typename T, bool SAMPLED, bool CACHED = false, bool FLUSH = false, bool ROWOUT = false, bool ROW32 = false>
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
  float *__restrict__ frow = nullptr;  // ROWOUT: this walker's row of float32 sub-eps elements (global)
  uint32_t *kbm = nullptr;             // ROWOUT: bitmap of the kept columns (LDS): a kept column's place among the records is the rank of its bit
  template <bool F = FLUSH, typename = std::enable_if_t<F>>
  __device__ __forceinline__ bool pause() const {
    return __builtin_amdgcn_readfirstlane(__atomic_load_n(list_n, __ATOMIC_RELAXED)) > pause_at;
  }
  __device__ __forceinline__ void add(uint32_t col, T h) {
    const T a = fabs(h);
    if constexpr (CACHED) hrow[col] = h;
    // (streaming stores: the row is read back once, 50 us later, and must not push the integral plan out of the XCD's L2 meanwhile: 598 -> 593 us)
    if constexpr (ROWOUT || ROW32) __builtin_nontemporal_store(a >= eps ? 0.0f : (float)h, frow + col);
    if (a >= eps) {
      const uint32_t k = atomicAdd(list_n, 1u);
      if (k < cap) {
        list_key[k] = FLUSH ? (((unsigned long long)col << 32) | k | tag) : (((unsigned long long)col << 32) | k);
        rec_w[k] = h;
      }
      if constexpr (ROWOUT) atomicOr(&kbm[col >> 5], 1u << (col & 31u));
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
    if constexpr (ROWOUT || ROW32) {
      if ((reinterpret_cast<uintptr_t>(frow + col) & 7u) == 0) {  // (neighbouring elements in one 8-byte store)
        typedef float f2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f2{fabs(h0) >= eps ? 0.0f : (float)h0, fabs(h1) >= eps ? 0.0f : (float)h1}, reinterpret_cast<f2 *>(frow + col));
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
      if (k < cap) { list_key[k] = FLUSH ? (((unsigned long long)col << 32) | k | tag) : (((unsigned long long)col << 32) | k); rec_w[k] = h; }
      if constexpr (ROWOUT) atomicOr(&kbm[col >> 5], 1u << (col & 31u));
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
__host__ __device__ inline size_t list_base_lds(const SDParams &p, size_t esz, bool sampled, bool cached, bool rowout = false) {
  if (rowout) {  // the draws' arrays (reduce_draw.h) lie over the staging scratch once the enumeration is over: at least that much (float32 integrals: 4 KB of scratch)
    const size_t scratch = esz * (size_t)(list_quarter(sampled) * (kBlock / 64)), draw = draw_lds_bytes(0);
    return (list_scratch_offset(p) + (scratch > draw ? scratch : draw) + 15) & ~(size_t)15;
  }
  const size_t scratch = esz * (size_t)(list_quarter(sampled) * (kBlock / 64)), draw = (kBlock / 64) * kCachedDrawLdsPerWave;
  if (cached) return (list_scratch_offset(p) + (scratch > draw ? scratch : draw) + 15) & ~(size_t)15;
  return (lds_fixed_bytes(p) + scratch + 15) & ~(size_t)15;
}

// LDS of the LIST form after the walker tables and the staging scratch:
//   (SAMPLED) tsum[max_tiles] f64 | dinfo[max_tiles] u32 | draw areas ;  then the list: key[P] u64  (P = power of two >= capacity),
//   which the draw slots' columns (pend[N] u32) re-use in phase C

// gtile: the tile sums and the tiles' draw counts live in global memory (io->tile_scratch) instead of the LDS -- long rows: 4768 tiles at
// sorb 120 are 57 KB, which with the draw areas and the list leaves ONE workgroup per CU
// row32 (the flushing form with the row's float32 copy): no draw areas of their own -- the waves' 2.5 KB (running sums and hit counts of one
// tile) lie over the staging scratch and, for the waves that do not fit there, behind the draw slots' columns in the list's memory, both
// done with when the draws begin (row32_draw_area; row32_fits: whether they do)
__host__ __device__ inline size_t row32_scratch_bytes(size_t esz) { return esz * (size_t)(list_quarter(true) * (kBlock / 64)); }
__host__ __device__ inline size_t row32_pend_bytes(uint32_t nsample, bool gtile) { return (((size_t)nsample * 4 * (gtile ? 2 : 1)) + 15) & ~(size_t)15; }
__host__ __device__ inline bool row32_fits(size_t esz, uint32_t P, uint32_t nsample, bool gtile) {
  const size_t in_scratch = row32_scratch_bytes(esz) / kCachedDrawLdsPerWave;
  const size_t rest = in_scratch >= (size_t)(kBlock / 64) ? 0 : (size_t)(kBlock / 64) - in_scratch;
  return row32_pend_bytes(nsample, gtile) + rest * kCachedDrawLdsPerWave <= (size_t)P * 8;
}
// offset of wave `wave`'s area from the start of the staging scratch (in_list = false) or of the list (in_list = true)
__host__ __device__ inline size_t row32_draw_area(size_t esz, uint32_t nsample, bool gtile, int wave, bool &in_list) {
  const size_t in_scratch = row32_scratch_bytes(esz) / kCachedDrawLdsPerWave;
  in_list = (size_t)wave >= in_scratch;
  return in_list ? row32_pend_bytes(nsample, gtile) + ((size_t)wave - in_scratch) * kCachedDrawLdsPerWave : (size_t)wave * kCachedDrawLdsPerWave;
}

__host__ __device__ inline size_t onepass_list_lds(const SDParams &p, size_t esz, uint32_t max_tiles, bool sampled, uint32_t P, uint32_t nsample,
                                                   bool cached, bool gtile = false, bool rowout = false, bool row32 = false) {
  size_t b = list_base_lds(p, esz, sampled, cached, rowout);
  if (sampled) {
    if (!gtile) {
      if (rowout) b += ((size_t)max_tiles * 8 + 15) & ~(size_t)15;  // (no draw counts per tile)
      else b += (size_t)max_tiles * 8 + (((size_t)max_tiles * 4 + 15) & ~(size_t)15);
    }
    if (!cached && !rowout && !row32) b += (kBlock / 64) * kDrawLdsPerWave;
  }
  if (rowout) {
    // enumeration: kept list [P] u64 | bitmap of the kept columns [kBlock] u32; afterwards: the kept columns in slot order [P] u32 | the drawn
    // records' slots [nsample] u32 (over the second half of the list and the bitmap, both done with by then); the draws' other arrays lie
    // over the staging scratch (reduce_draw.h)
    const size_t during = (((size_t)P * 8 + 15) & ~(size_t)15) + (size_t)kBlock * 4;
    const size_t after = (((size_t)P * 4 + 15) & ~(size_t)15) + (((size_t)nsample * 4 + 15) & ~(size_t)15);
    return b + (during > after ? during : after);
  }
  const size_t list = (size_t)P * 8, pend = (size_t)nsample * 4 * (gtile ? 2 : 1);  // (gtile: + the list of the drawn tiles)
  return b + ((list > pend ? list : pend) + 15 & ~(size_t)15);
}

// Tile depth of the flushing form.  Three-word determinants (sorb > 128) have 33 KB of walker tables: with the list and the staging scratch two
// workgroups per CU = 2 waves per SIMD, each with 2 PYNQS_U gathers per lane in flight -- 9 MB in flight chip-wide, 4.7 TB/s at ~2 us of
// latency.  The deterministic form keeps nothing per tile, so its tiles are twice as deep there (8 gathers per lane);
// a wave in mid-tile may then hold 512 kept columns back, and the list has 4096 slots for the pause threshold to stay above half of it.
#ifndef PYNQS_U_LONG
#define PYNQS_U_LONG (3 * PYNQS_U)
#endif
__host__ __device__ constexpr int flush_tile_depth(int len, bool sampled, bool flush) { return (flush && !sampled && len >= 3) ? PYNQS_U_LONG : PYNQS_U; }
__host__ __device__ constexpr uint32_t flush_list_slots(int len, bool sampled, bool flush = true) {
  return flush_tile_depth(len, sampled, flush) > PYNQS_U ? 4096u : 2048u;
}
static_assert((kBlock / 64) * 128 * PYNQS_U_LONG + 64 < 4096 && (kBlock / 64) * 128 * PYNQS_U + 64 < 2048, "the pause threshold of the flushing form");

template <int LEN, typename T, bool SAMPLED, bool CACHED, bool FLUSH = false, bool GTILE = false, bool ROWOUT = false, bool ROW32 = false>
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
  unsigned char *extra = smem + list_base_lds(p, sizeof(T), SAMPLED, CACHED, ROWOUT);
  unsigned char *tile_mem = GTILE ? o.tile_scratch + (size_t)walker * o.tile_stride : extra;
  double *tsum = reinterpret_cast<double *>(tile_mem);
  uint32_t *dinfo = reinterpret_cast<uint32_t *>(tile_mem + (SAMPLED ? (size_t)max_tiles * 8 : 0));
  unsigned char *after = (SAMPLED && !GTILE) ? (ROWOUT ? extra + (((size_t)max_tiles * 8 + 15) & ~(size_t)15)
                                                       : reinterpret_cast<unsigned char *>(dinfo) + (((size_t)max_tiles * 4 + 15) & ~(size_t)15)) : extra;
  unsigned char *draw0 = CACHED ? smem + list_scratch_offset(p) : after;  // (CACHED: over the staging scratch, see list_base_lds)
  static_assert(!ROWOUT || (SAMPLED && !CACHED && !FLUSH && !GTILE), "row to global memory as float32, the draws of reduce_draw.h");
  static_assert(!ROW32 || (SAMPLED && !CACHED && FLUSH && !ROWOUT), "float32 copy of a long row: the flushing semi-stochastic form");
  if (SAMPLED && !CACHED && !ROWOUT && !ROW32) after += (kBlock / 64) * kDrawLdsPerWave;
  // the kept list: ONE 64-bit key per entry (column << 32 | order of arrival); the values wait in the segment's rec_w, in order of arrival,
  // and are permuted after the sort (12 -> 8 bytes of LDS per entry: with the draw slots' 4000 bytes sharing the memory that is what
  // decides between 7 and 8 workgroups per CU)
  unsigned long long *list_key = reinterpret_cast<unsigned long long *>(after);
  uint32_t *pend = reinterpret_cast<uint32_t *>(after);           // phase C re-uses the list's memory
  uint32_t *kbm = reinterpret_cast<uint32_t *>(after + (((size_t)P * 8 + 15) & ~(size_t)15));  // ROWOUT: [kBlock] words (rows of up to 8192 columns)
  uint32_t *drec = reinterpret_cast<uint32_t *>(after + (((size_t)P * 4 + 15) & ~(size_t)15));  // ROWOUT: the drawn records' slots [nsample], behind kcol[P]:
                                                                                                // over the list's second half and kbm, dead by then
  if constexpr (SAMPLED) {
    for (uint32_t i = tid; i < max_tiles; i += kBlock) {
      tsum[i] = 0.0;
      if constexpr (!ROWOUT) dinfo[i] = 0u;
    }
    if constexpr (!ROWOUT) {
      for (uint32_t i = tid; i < nsample; i += kBlock) o.srec_col[(int64_t)walker * nsample + i] = -1;
    } else {  // (the zeros behind the row's last column; the bitmap of the kept columns)
      for (size_t i = p.nsd + 1 + tid; i < draw_row_stride(p.nsd + 1); i += kBlock) o.row_f32[(size_t)walker * draw_row_stride(p.nsd + 1) + i] = 0.0f;
      kbm[tid] = 0u;
    }
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
    ListKeepSink<LEN, T, SAMPLED, CACHED, FLUSH, ROWOUT, ROW32> sink{eps, &list_n, list_key, o.rec_w + seg_base + flushed, room, tsum, 0xffffffffu, 0.0,
                                                                     CACHED ? o.row_cache + (size_t)walker * (p.nsd + 1) : nullptr};
    if constexpr (ROWOUT) { sink.frow = o.row_f32 + (size_t)walker * draw_row_stride(p.nsd + 1); sink.kbm = kbm; }
    if constexpr (ROW32) sink.frow = o.row_f32 + (size_t)walker * draw_row_stride(p.nsd + 1);
    constexpr int kUU = flush_tile_depth(LEN, SAMPLED, FLUSH);
    if constexpr (FLUSH) sink.pause_at = P - (kBlock / 64) * (kMaxKeptPerTile * kUU / PYNQS_U) - 64;  // (every wave may be in the middle of a tile)
    const bool exhausted =
        visit_tiles<LEN, T, decltype(sink), true, list_quarter(SAMPLED), kUU>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, 0u, &next_tile, sink);
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
  if constexpr (ROWOUT) {
    // ---- no sort: a kept column's slot is the RANK of its bit in the bitmap (block scan of the words' popcounts) -- ascending columns like
    //      the reference's boolean mask, whatever the waves' timing was; the records are resolved later, together with the drawn ones ----
    uint32_t *kpre = reinterpret_cast<uint32_t *>(smem + list_scratch_offset(p));  // (over the staging scratch: the enumeration is over)
    {
      uint32_t dummy;
      const uint32_t c = (uint32_t)__popc(kbm[tid]);
      const uint32_t inc = draw_block_scan<uint32_t>(c, s_parti, &dummy);
      kpre[tid] = inc - c;
    }
    __syncthreads();
    PYNQS_STAMP(3);
    constexpr int kMaxPer = 4;  // n <= 1024
    T mine_w[kMaxPer];
    uint32_t mine_c[kMaxPer], mine_r[kMaxPer];
#pragma unroll
    for (int r = 0; r < kMaxPer; ++r) {
      const uint32_t i = (uint32_t)r * kBlock + tid;
      mine_r[r] = 0xffffffffu; mine_c[r] = 0; mine_w[r] = T(0);
      if (i < n) {
        const unsigned long long e = list_key[i];
        mine_c[r] = (uint32_t)(e >> 32);
        mine_w[r] = o.rec_w[seg_base + (uint32_t)(e & 0xffffffffull)];
        mine_r[r] = kpre[mine_c[r] >> 5] + (uint32_t)__popc(kbm[mine_c[r] >> 5] & ((1u << (mine_c[r] & 31u)) - 1u));
      }
    }
    __syncthreads();
    uint32_t *kcol = reinterpret_cast<uint32_t *>(list_key);  // the kept columns in slot order (over the list, which has been read)
    for (uint32_t i = tid; i < cap; i += kBlock) kcol[i] = 0xffffffffu;  // (a slot that stays empty -- overflow -- must not be decoded)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kMaxPer; ++r) {
      if (mine_r[r] < cap) {
        __builtin_nontemporal_store(mine_w[r], o.rec_w + seg_base + mine_r[r]);
        __builtin_nontemporal_store((int32_t)mine_c[r], o.rec_col + seg_base + mine_r[r]);
        kcol[mine_r[r]] = mine_c[r];
      }
    }
    PYNQS_STAMP(4);
    PYNQS_STAMP(5);
    // S: the tiles' exact sums in tile order (their own sums were formed lane by lane + butterfly: nothing depends on the waves' timing)
    if (wave == 0) {
      double sl = 0.0;
      for (uint32_t i = lane; i < max_tiles; i += 64) sl += tsum[i];
      sl = op_wave_sum(sl);
      if (lane == 0) { s_part[kBlock / 64] = sl; if (o.row_sum) o.row_sum[walker] = sl; }
    }
    __syncthreads();  // (also: the row in global memory is complete, kcol is complete, the staging scratch / tile sums are done with)
    const double Srow = s_part[kBlock / 64];
    __syncthreads();
    PYNQS_STAMP(6);
    rowout_draws<LEN, T>(smem + list_scratch_offset(p), drec, kcol, n, seg_base, p, L, wk, o.row_f32 + (size_t)walker * draw_row_stride(p.nsd + 1), nsample, seed,
                         walker, Srow, o, &bw_cnt, &bw_base, &s_full, s_part, s_parti);
    PYNQS_STAMP(9);
    return;
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
    constexpr int kMaxPer = (int)(flush_list_slots(LEN, SAMPLED, FLUSH) / kBlock);  // n <= 2048 = 8 x 256 (three-word deterministic flushing form: 4096)
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
    if constexpr (ROW32) {  // (over the staging scratch / behind the draw slots' columns in the list's memory: see row32_draw_area)
      bool in_list;
      const size_t off = row32_draw_area(sizeof(T), nsample, GTILE, wave, in_list);
      mine = (in_list ? after : smem + list_scratch_offset(p)) + off;
    }
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
    } else if constexpr (ROW32) {
      // the draws inside the drawn tiles from the float32 copy of the row (written by this workgroup; the barriers above make it visible):
      // no second enumeration.  A wave pulls a drawn tile, loads its <= 256 columns (4 per lane; tile 0: its <= 7 scattered columns), forms
      // the running sums of the float32 widths with one wave scan and draws / counts / emits as the row-cache form does.  P(column | tile) =
      // w32 / sum of the tile's w32 (|H| rounded to float32, relative 6e-8); P(tile) and S stay the exact float64 sums of the enumeration.
      const TileGeom<LEN, true, PYNQS_U> geom(p, nchunks, chunk, chunk_len, 0u);
      const float *__restrict__ frow = o.row_f32 + (size_t)walker * draw_row_stride(p.nsd + 1);
      const double scale = Srow / (double)nsample;
      const int64_t sbase = (int64_t)walker * nsample;
      uint32_t *hits2 = reinterpret_cast<uint32_t *>(mine + (size_t)kOneTileCols * 8);   // [cols / 2]: two 16-bit counts per word
      for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(&next_tile, 1u);
        q = __builtin_amdgcn_readfirstlane(q);
        uint32_t t;
        if constexpr (GTILE) { if (q >= ndrawn) break; t = tlist[q]; }
        else { if (q >= geom.ntiles) break; t = q; }
        const uint32_t info = dinfo[t], draws = info & 0xffffu;
        if (draws == 0 || (o.debug & 64u)) continue;
        uint32_t c0 = 0, ncols = 0;
        if (t) geom.columns(t, c0, ncols);
        float x4[4];
        uint32_t cl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t e = (uint32_t)lane * 4 + j;
          cl[j] = t ? (e < ncols ? c0 + e : 0xffffffffu) : geom.odd_column((int)e < 7 ? (int)e : 7);
          x4[j] = cl[j] != 0xffffffffu ? frow[cl[j]] : 0.0f;
        }
        const double w4[4] = {(double)fabsf(x4[0]), (double)fabsf(x4[1]), (double)fabsf(x4[2]), (double)fabsf(x4[3])};
        const double ls = (w4[0] + w4[1]) + (w4[2] + w4[3]);
        const double incl = op_scan(ls, lane);
        const double total = __shfl(incl, 63);
        double run = incl - ls;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          run += w4[j];
          S.prefix[lane * 4 + j] = run;
        }
        hits2[lane * 2] = 0u; hits2[lane * 2 + 1] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (!(total > 0.0)) continue;
        for (uint32_t k = lane; k < draws; k += 64) {
          const uint64_t r = op_mix64(key ^ op_mix64(((uint64_t)t << 32) | k));
          const double target = (double)(r >> 11) * 0x1.0p-53 * total;
          uint32_t lo = 0, hi = kOneTileCols;
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (S.prefix[mid] > target) hi = mid; else lo = mid + 1;
          }
          if (lo >= (uint32_t)kOneTileCols) lo = kOneTileCols - 1;
          while (lo > 0 && !(S.prefix[lo] > S.prefix[lo - 1])) --lo;   // (never a column of zero width: a kept one, or padding)
          atomicAdd(&hits2[lo >> 1], 1u << (16u * (lo & 1u)));  // (a column is drawn < 2^16 times: nsample < 2^16)
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t h01 = hits2[lane * 2], h23 = hits2[lane * 2 + 1];
        const uint32_t hc[4] = {h01 & 0xffffu, h01 >> 16, h23 & 0xffffu, h23 >> 16};
        const uint32_t cnt4 = (hc[0] ? 1u : 0u) + (hc[1] ? 1u : 0u) + (hc[2] ? 1u : 0u) + (hc[3] ? 1u : 0u);
        uint32_t before = cnt4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t ov = __shfl_up(before, d);
          if (lane >= d) before += ov;
        }
        uint32_t at = (info >> 16) + before - cnt4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (hc[j]) {
            o.srec_col[sbase + at] = (int32_t)cl[j];
            const double v = scale * (double)hc[j];
            o.srec_w[sbase + at] = (T)(x4[j] < 0.0f ? -v : v);
            pend[at] = cl[j];
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

}  // namespace pynqs
