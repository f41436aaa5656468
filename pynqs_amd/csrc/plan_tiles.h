// plan_tiles.h -- the tile scheduler shared by the plan-based kernels: visits every column of a walker's
// row range exactly once and hands (column, <x|H|x'>, ket) to a sink.  The drop-in kernel's sink stores to
// comb / Hmat (kernels_plan.hip); the fused local-energy kernel's sink looks psi(x') up and accumulates
// (kernels_eloc.hip).
//
// Work distribution inside a workgroup.  After the (workgroup-wide) table build there is NO barrier: every wave
// pulls tiles from an LDS counter until none is left.  Tile 0 holds the odd jobs (the few unpaired columns and
// column 0 with its ordered diagonal sum, ~4k cycles of one lane), then come tiles of kSinglesPerTile singles
// (coalesced gather staged in the wave's private LDS quarter, ordered sum by one lane per single), then tiles of
// 64*U pair slots of the doubles (classes padded to whole tiles; a lane produces two consecutive columns).
// Earlier versions ran singles and diagonal as workgroup-wide, barrier-separated phases: 2 % of the columns
// cost 0.08-0.1 ms of a 0.26 ms kernel (tools/ab.sh ablations).
#pragma once

#include <type_traits>
#include <utility>

#include "detcore.h"
#include "plan.h"
#include "plan_dev.h"

namespace pynqs {

#ifndef PYNQS_U
#define PYNQS_U 2
#endif

// One class of doubles restricted to this workgroup's rank range, cut into pair slots:
// slot m = ranks (r_e + 2m, r_e + 2m + 1).  `odd_base` = parity of the walker's first element index
// (walker * ncomb): r_e is the first rank whose column has an even element index, so that 16-byte stores of a
// slot are aligned; at most one leading and one trailing column have no partner.
struct ClassRange {
  uint32_t a0, a1;   // ranks [a0, a1)
  uint32_t r_e;      // first paired rank
  uint32_t npairs;   // complete pairs
};

template <bool PAIRED>
__device__ __forceinline__ ClassRange class_range(uint32_t b0, uint32_t b1, uint32_t rlo, uint32_t rhi, uint32_t odd_base) {
  ClassRange g;
  g.a0 = max(rlo, b0);
  g.a1 = max(g.a0, min(rhi, b1));
  if constexpr (PAIRED) {
    g.r_e = g.a0 + ((odd_base + g.a0 + 1) & 1u);
    g.npairs = g.a1 > g.r_e ? (g.a1 - g.r_e) / 2 : 0;
  } else {  // slots are single ranks: npairs counts ranks
    g.r_e = g.a0;
    g.npairs = g.a1 - g.a0;
  }
  return g;
}

// Multi-word ONVs (LEN >= 2) do NOT pair neighbouring columns: a ket is already 16 or 24 bytes, and a lane that
// writes two neighbouring kets makes every store instruction cover only half (a third) of each cache line --
// the L2 then sees two (three) partial write requests per line and the comb stream runs at 2.7 TB/s instead of
// ~6 (measured at sorb 120, --no-comb ablation).  There a lane takes columns c + lane and c + 64 + lane, so that
// each instruction writes one dense span.
// Sink concept:
//   void one(uint32_t col, T h, const uint64_t (&ket)[LEN]);                       // a single column
//   void pair(uint32_t col, T h0, T h1, const uint64_t (&k0)[LEN], const uint64_t (&k1)[LEN]);  // col, col+1
//   void two(uint32_t c0, T h0, const uint64_t (&k0)[LEN], uint32_t c1, T h1, const uint64_t (&k1)[LEN]);  // any two
//   void tile_begin(uint32_t tile);   // called by all lanes of the wave that took tile `tile` of this workgroup
// Number of tiles of a workgroup: at most max_tiles_per_chunk() (host and device agree on it).
// `next_tile` is a workgroup-shared counter that must be 0 when the first wave arrives (set it before
// build_walker_tables, whose final barrier publishes it).
// blockIdx -> (walker, chunk).  A chunk of columns reads its own part of the plan (the alpha-beta doubles of a chunk
// use a few [pb][pj] sub-matrices of Vab).  Workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB
// L2 (blocks b and b + 8 share one; observed, not promised -- only speed depends on it): with xcd_map, XCD slot
// b % 8 gets an eighth of the chunk indices and runs through them chunk by chunk over all walkers, so that the
// blocks resident on one XCD re-use the same plan lines instead of all eight L2s streaming the whole plan.
__device__ __forceinline__ void map_workgroup(uint32_t nchunks, bool xcd_map, uint64_t &walker, uint32_t &chunk) {
  const uint32_t wg = blockIdx.x;
  if (xcd_map) {  // host guarantees nchunks % 8 == 0
    const uint32_t nbatch = gridDim.x / nchunks, cpx = nchunks >> 3;
    const uint32_t q = wg >> 3, cl = q / nbatch;
    walker = q - cl * nbatch;
    chunk = (wg & 7u) * cpx + cl;
  } else {
    walker = wg / nchunks;
    chunk = wg - (uint32_t)walker * nchunks;
  }
}

// upper bound of `ntiles` below for any chunk of a walker's row
__host__ __device__ inline uint32_t max_tiles_per_chunk(const SDParams &p, uint32_t nchunks, uint32_t chunk_len) {
  const uint32_t tS_all = (p.d1 + kSinglesPerTile - 1) / kSinglesPerTile;
  return 1 + (tS_all + nchunks - 1) / nchunks + chunk_len / (128u * PYNQS_U) + 4;
}

constexpr int kSinglesPerFastTile = 64;

// EXACT = true : <x|H|x> and the singles are summed in the reference's order (bit-identical values; needs the LDS
//                staging scratch of lds_bytes(p, sizeof(T)), or 4 * QUARTER elements when QUARTER is given).
// EXACT = false: order-free sums (plan_dev.h: fast_diag / fast_single), no scratch (lds_bytes(p, 0)), 64 singles per
//                tile: for sinks that only accumulate a rounded sum over all columns (fused local energies).
// optional member of a sink: bool skip_tile(uint32_t tile) (wave-uniform), asked after tile_begin -- true: none of
// this tile's columns is wanted, do not even enumerate them (the draw pass of the semi-stochastic REDUCE: a tile that
// received no draws)
template <typename S, typename = void>
struct sink_can_skip : std::false_type {};
template <typename S>
struct sink_can_skip<S, std::void_t<decltype(std::declval<S &>().skip_tile(0u))>> : std::true_type {};

// optional member of a sink: bool pause() (wave-uniform), asked before a wave takes its next tile -- true: visit_tiles returns (false)
// with the tile counter untouched, and may be called again later to go on (the flushing LIST form of the REDUCE front end: the
// workgroup's list is nearly full).  visit_tiles returns true when this wave found the tiles exhausted.
template <typename S, typename = void>
struct sink_can_pause : std::false_type {};
template <typename S>
struct sink_can_pause<S, std::void_t<decltype(std::declval<S &>().pause())>> : std::true_type {};

// optional member of a sink: uint32_t remap(uint32_t k) -- the k-th tile a wave takes is tile remap(k) (0xffffffff: none left): a pass over
// a LIST of tiles instead of all of them (the draw pass of the semi-stochastic REDUCE on long rows: ~900 drawn tiles of 4768, and asking
// skip_tile for each of the others is a load from global memory when the draw counts live there)
template <typename S, typename = void>
struct sink_can_remap : std::false_type {};
template <typename S>
struct sink_can_remap<S, std::void_t<decltype(std::declval<S &>().remap(0u))>> : std::true_type {};

// UU: pair slots per lane and tile.  Everything that sizes buffers by the tile (the semi-stochastic forms' draw areas, the look-back buffers,
// max_tiles_per_chunk) assumes PYNQS_U; a sink that keeps nothing per tile may ask for deeper tiles = more gathers in flight per lane.
// The tiles of one workgroup (one chunk of a walker's row), as visit_tiles numbers them: tile 0 (column 0 and the unpaired doubles), tS tiles
// of singles, then tA / tB / tO tiles of same-spin alpha, same-spin beta and opposite-spin doubles.  columns(): the columns a tile covers -- a
// contiguous range for every tile but tile 0 -- for code that comes back to a tile later (the draws of the semi-stochastic REDUCE from the
// row's float32 copy).
template <int LEN, bool EXACT = true, int UU = PYNQS_U>
struct TileGeom {
  static constexpr bool kPaired = LEN == 1;
  static constexpr uint32_t kTile = 64u * UU;                       // pair slots per tile
  static constexpr uint32_t kSlots = kPaired ? kTile : 2 * kTile;   // slots per tile: pair slots, or single ranks
  static constexpr uint32_t kSPT = EXACT ? kSinglesPerTile : kSinglesPerFastTile;
  ClassRange gA, gB, gO;
  uint32_t tS, tA, tB, tO, ntiles, lo, d1, chunk, nchunks;
  __device__ __forceinline__ TileGeom(const SDParams &p, uint32_t nchunks_, uint32_t chunk_, uint32_t chunk_len, uint32_t odd_base) {
    const uint32_t ncomb = p.nsd + 1;
    lo = chunk_ * chunk_len;
    const uint32_t hi = min(lo + chunk_len, ncomb);
    const uint32_t rlo = lo == 0 ? 0 : lo - 1, rhi = hi - 1;  // excitation ranks [rlo, rhi): column = rank + 1
    gA = class_range<kPaired>(p.d1, p.d2, rlo, rhi, odd_base);
    gB = class_range<kPaired>(p.d2, p.d3, rlo, rhi, odd_base);
    gO = class_range<kPaired>(p.d3, p.nsd, rlo, rhi, odd_base);
    tA = (gA.npairs + kSlots - 1) / kSlots; tB = (gB.npairs + kSlots - 1) / kSlots; tO = (gO.npairs + kSlots - 1) / kSlots;
    // The singles tiles of the walker are dealt round-robin to its workgroups (they cost far more per column
    // than doubles; left to the first chunk they would make it the straggler when rows are cut into many chunks).
    const uint32_t tS_all = (p.d1 + kSPT - 1) / kSPT;
    tS = tS_all > chunk_ ? (tS_all - chunk_ + nchunks_ - 1) / nchunks_ : 0;
    ntiles = 1 + tS + tA + tB + tO;
    d1 = p.d1; chunk = chunk_; nchunks = nchunks_;
  }
  // tile >= 1: first column and number of columns of the tile
  __device__ __forceinline__ void columns(uint32_t tile, uint32_t &c0, uint32_t &n) const {
    if (tile <= tS) {
      const uint32_t r0 = (chunk + (tile - 1) * nchunks) * kSPT;
      c0 = r0 + 1; n = min(r0 + kSPT, d1) - r0;
      return;
    }
    tile -= 1 + tS;
    const int k = tile < tA ? 0 : (tile < tA + tB ? 1 : 2);
    const ClassRange &g = k == 0 ? gA : (k == 1 ? gB : gO);
    const uint32_t first = (tile - (k == 0 ? 0u : (k == 1 ? tA : tA + tB))) * kSlots;
    const uint32_t slots = min(kSlots, g.npairs - first);
    if (kPaired) { c0 = g.r_e + 2 * first + 1; n = 2 * slots; }
    else { c0 = g.r_e + first + 1; n = slots; }
  }
  // tile 0: its j-th column (j < 7), or 0xffffffff: column 0 when the chunk starts the row, then the unpaired heads / tails of the three classes
  __device__ __forceinline__ uint32_t odd_column(int j) const {
    if (j == 0) return lo == 0 ? 0u : 0xffffffffu;
    if (!kPaired || j > 6) return 0xffffffffu;
    const int k = (j - 1) >> 1;
    const ClassRange &g = k == 0 ? gA : (k == 1 ? gB : gO);
    const uint32_t tail = g.r_e + 2 * g.npairs;
    const bool head = ((j - 1) & 1) == 0;
    if (head ? (g.r_e > g.a0 && g.a0 < g.a1) : (tail < g.a1)) return (head ? g.a0 : tail) + 1;
    return 0xffffffffu;
  }
};

template <int LEN, typename T, typename Sink, bool EXACT = true, int QUARTER = kDiagTile / 4, int UU = PYNQS_U>
__device__ __forceinline__ bool visit_tiles(const SDParams &p, const PlanLayout &pl, const LdsLayout &L, int nocc,
                                            const T *__restrict__ plan, const Walker<LEN> &wk, uint32_t nchunks, uint32_t chunk,
                                            uint32_t chunk_len, uint32_t odd_base, uint32_t *next_tile, Sink &sink) {
  constexpr int U = UU;                // pair slots per lane and tile (2*U gathers in flight)
  static_assert(UU >= PYNQS_U && UU % PYNQS_U == 0, "max_tiles_per_chunk (sized for PYNQS_U) must stay an upper bound");
  constexpr uint32_t kTile = 64u * U;  // pair slots per tile
  const int lane = threadIdx.x & 63;
  const uint32_t ncomb = p.nsd + 1;
  const uint32_t lo = chunk * chunk_len;
  const uint32_t hi = min(lo + chunk_len, ncomb);
  const uint32_t rlo = lo == 0 ? 0 : lo - 1, rhi = hi - 1;  // excitation ranks [rlo, rhi): column = rank + 1

  constexpr bool kPaired = LEN == 1;
  constexpr uint32_t kSlots = kPaired ? kTile : 2 * kTile;  // slots per tile: pair slots, or single ranks
  const TileGeom<LEN, EXACT, UU> geom(p, nchunks, chunk, chunk_len, odd_base);
  const ClassRange gA = geom.gA, gB = geom.gB, gO = geom.gO;
  const uint32_t tA = geom.tA, tB = geom.tB, tS = geom.tS;
  constexpr uint32_t kSPT = EXACT ? kSinglesPerTile : kSinglesPerFastTile;
  const uint32_t ntiles = geom.ntiles;
  const T *__restrict__ Vss = plan + pl.offVss;
  const T *__restrict__ Vab = plan + pl.offVab;

  auto emit_pair = [&](const PendingDouble<T> &d0, const PendingDouble<T> &d1, uint32_t col, const DoubleClass &c) {
    uint64_t k0[LEN], k1[LEN];
    const T h0 = finish_double<LEN, T>(d0, c, wk, k0);
    const T h1 = finish_double<LEN, T>(d1, c, wk, k1);
    sink.pair(col, h0, h1, k0, k1);
  };

  for (;;) {
    if constexpr (sink_can_pause<Sink>::value) {
      if (sink.pause()) return false;
    }
    uint32_t tile = 0;
    if (lane == 0) tile = atomicAdd(next_tile, 1u);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if constexpr (sink_can_remap<Sink>::value) tile = sink.remap(tile);
    if (tile >= ntiles) break;
    sink.tile_begin(tile);  // wave-uniform; everything until the next call belongs to this tile, in a fixed order
    if constexpr (sink_can_skip<Sink>::value) {
      if (sink.skip_tile(tile)) continue;
    }
    if (tile == 0) {
      // unpaired columns of the three classes: lanes 0..5
      if (kPaired && lane < 6) {
        const int k = lane >> 1;
        const ClassRange g = k == 0 ? gA : (k == 1 ? gB : gO);
        const uint32_t tail = g.r_e + 2 * g.npairs;
        const bool head = (lane & 1) == 0;
        if (head ? (g.r_e > g.a0 && g.a0 < g.a1) : (tail < g.a1)) {
          const DoubleClass c = k == 2 ? make_opp_spin(p, pl) : make_same_spin(p, pl, k);
          const uint32_t r = head ? g.a0 : tail;
          uint64_t ket[LEN];
          const PendingDouble<T> d = fetch_double<LEN, T>(r, c, L, k == 2 ? Vab : Vss + (size_t)k * pl.NP * pl.NP);
          const T h = finish_double<LEN, T>(d, c, wk, ket);
          sink.one(r + 1, h, ket);
        }
      }
      if (lo == 0) {
        if constexpr (EXACT) {
          diag_wave<T, QUARTER>(p, pl, L, plan, [&](T v) {
            uint64_t ket[LEN];
#pragma unroll
            for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
            sink.one(0u, v, ket);
          });
        } else {
          const T v = fast_diag<T>(p, pl, L, plan);
          if (lane == 0) {
            uint64_t ket[LEN];
#pragma unroll
            for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
            sink.one(0u, v, ket);
          }
        }
      }
      continue;
    }
    if (tile <= tS) {
      const uint32_t r0 = (chunk + (tile - 1) * nchunks) * kSPT;
      if constexpr (EXACT) {
        singles_tile<T, QUARTER>(r0, min(r0 + kSPT, p.d1), p, pl, L, nocc, plan, [&](uint32_t r, T v, uint32_t e) {
          uint64_t ket[LEN];
#pragma unroll
          for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
          toggle<LEN>(ket, e & 0xff);
          toggle<LEN>(ket, (e >> 8) & 0xff);
          sink.one(r + 1, v, ket);
        });
      } else {
        const uint32_t r = r0 + lane;
        if (r < p.d1) {
          const T v = fast_single<T>(r, p, pl, L, nocc, plan);
          const uint32_t e = L.tab[p.offSa + r];
          uint64_t ket[LEN];
#pragma unroll
          for (int i = 0; i < LEN; ++i) ket[i] = wk.w[i];
          toggle<LEN>(ket, e & 0xff);
          toggle<LEN>(ket, (e >> 8) & 0xff);
          sink.one(r + 1, v, ket);
        }
      }
      continue;
    }
    tile -= 1 + tS;
    // which class (wave-uniform)
    const int k = tile < tA ? 0 : (tile < tA + tB ? 1 : 2);
    const ClassRange g = k == 0 ? gA : (k == 1 ? gB : gO);
    const uint32_t first = (tile - (k == 0 ? 0u : (k == 1 ? tA : tA + tB))) * kSlots;  // first slot of the tile
    const DoubleClass c = k == 2 ? make_opp_spin(p, pl) : make_same_spin(p, pl, k);
    const T *__restrict__ V = k == 2 ? Vab : Vss + (size_t)k * pl.NP * pl.NP;
    if constexpr (!kPaired) {
      // ranks first + (2u + v) * 64 + lane: every store instruction of the sink covers 64 neighbouring columns
      if (first + kSlots <= g.npairs) {
        PendingDouble<T> d[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) d[u][v] = fetch_double<LEN, T>(g.r_e + first + (2 * u + v) * 64 + lane, c, L, V);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          uint64_t k0[LEN], k1[LEN];
          const T h0 = finish_double<LEN, T>(d[u][0], c, wk, k0);
          const T h1 = finish_double<LEN, T>(d[u][1], c, wk, k1);
          const uint32_t r0 = g.r_e + first + (2 * u) * 64 + lane;
          sink.two(r0 + 1, h0, k0, r0 + 65, h1, k1);
        }
      } else {
        for (uint32_t m = first + lane; m < g.npairs; m += 64) {
          uint64_t ket[LEN];
          const PendingDouble<T> d0 = fetch_double<LEN, T>(g.r_e + m, c, L, V);
          const T h = finish_double<LEN, T>(d0, c, wk, ket);
          sink.one(g.r_e + m + 1, h, ket);
        }
      }
      continue;
    }
    if (first + kTile <= g.npairs) {  // full tile: no guards
      PendingDouble<T> d[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) fetch_double2<LEN, T>(g.r_e + 2 * (first + u * 64 + lane), c, L, V, d[u][0], d[u][1]);
#pragma unroll
      for (int u = 0; u < U; ++u) emit_pair(d[u][0], d[u][1], g.r_e + 2 * (first + u * 64 + lane) + 1, c);
    } else {
      for (uint32_t m = first + lane; m < g.npairs; m += 64) {
        PendingDouble<T> d0, d1;
        fetch_double2<LEN, T>(g.r_e + 2 * m, c, L, V, d0, d1);
        emit_pair(d0, d1, g.r_e + 2 * m + 1, c);
      }
    }
  }
  return true;
}

}  // namespace pynqs
