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
#include "reduce_common.h"
#include "reduce_list.h"

namespace pynqs {



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
  bool use_split = false;  // LIST kernel with the row's float32 copy in global memory and the draws of reduce_draw.h (kernels_reduce_rowout.hip)
  bool use_row32 = false;  // flushing semi-stochastic form whose draws read the drawn tiles back from the row's float32 copy (same file)
};
// (list slots of the flushing form: flush_list_slots(), reduce_list.h)
static OnepassForm onepass_form(const SDParams &p, size_t esz, uint32_t max_tiles, uint32_t fixed, uint64_t cap_doubles, int eps_sample,
                                bool have_cache, uint32_t chunk_len, bool no_table, bool have_tile_scratch = false, bool have_row_f32 = false) {
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
  // the two-kernel form takes over from the row cache / the re-enumerating form when the caller passed io->row_f32 and the kept records fit
  // the list (PYNQS_OP_SPLIT=0 ignores the buffer)
  static const int split_env = getenv("PYNQS_OP_SPLIT") ? atoi(getenv("PYNQS_OP_SPLIT")) : -1;
  if (sampled && have_row_f32 && split_env != 0 && list_env != 0 && seg_cap <= 1024 && reduce_draw_supported(p, eps_sample)) {
    // (no sort in this form: the list needs as many slots as a segment has records, not a power of two)
    const uint32_t slots = ((uint32_t)seg_cap + 1u) & ~1u;
    const size_t lds_a = onepass_list_lds(p, esz, max_tiles, true, slots, (uint32_t)eps_sample, false, false, true);
    if (lds_a + 256 <= 160 * 1024) {
      f.P = slots;
      f.use_list = true; f.use_split = true; f.use_cache = f.use_gtile = f.use_flush = false; f.lds = lds_a;
      return f;
    }
  }
  // the flushing LIST form: deterministic calls whose kept columns do not fit the list -- on long rows, on any row when at most a tenth
  // of a segment's columns can be kept (a flush costs a sort of 2048 entries; the look-back form pays per tile instead: Fe2S2 with 9 % kept
  // 0.81 look-back against 2.45 ms, sorb 56 with 6 % 6.5 against 3.7, sorb 80 with 10 % / 40 % 103 / 292 against 71 / 269), and whenever
  // there is no de-duplication table (the look-back form needs one).  PYNQS_OP_FLUSH=0 / 1: never / wherever possible.
  // With draws the kept list is flushed during the enumeration in the same way, under the same rule; the draws follow as before (sorb 56,
  // 200 draws, 2 % / 6 % kept: 6.2 / 7.1 ms look-back -> 1.6 / 2.2 flushing and table-less; Fe2S2, 9 %: 1.37 stays).  A row cache is not
  // used by this form: when the LIST form with the cache does not fit, flushing without it beats the look-back form (sorb 56, 1000 draws:
  // 6.2 / 3.8 -> 2.6 / 1.9 ms).
  const bool gtile_f = sampled && have_tile_scratch;
  const uint32_t flush_P = flush_list_slots((p.sorb - 1) / 64 + 1, sampled);
  const size_t lds_flush = onepass_list_lds(p, esz, max_tiles, sampled, flush_P, (uint32_t)eps_sample, false, gtile_f);
  const bool long_row = p.nsd + 1 > kLongRow;
  f.use_flush = !f.use_list && flush_env != 0 && lds_flush + 256 <= 160 * 1024 &&
                (long_row || flush_env == 1 || no_table || cap_doubles * 10 <= (uint64_t)chunk_len);
  if (f.use_flush) { f.use_gtile = gtile_f; f.use_cache = false; }
  if (f.use_flush) { f.P = flush_P; f.lds = lds_flush; }
  // (PYNQS_OP_ROW32=0: the drawn tiles are enumerated a second time, as before the end of round 4)
  static const int row32_env = getenv("PYNQS_OP_ROW32") ? atoi(getenv("PYNQS_OP_ROW32")) : -1;
  // The copy costs 4 bytes of stores per column; it pays where it puts more workgroups on a CU (no draw areas of their own: sorb 120 four
  // instead of two, 6.30 -> 4.79 ms per 1024 walkers) or where a good part of the tiles is drawn (sorb 56 / 80, 1000 draws: 1.81 -> 1.34 /
  // 3.41 -> 2.54 ms); at sorb 184 (26146 tiles, two workgroups per CU either way) 1000 draws re-enumerate 4 % of the row and the copy is
  // the dearer of the two (13.1 -> 13.9 ms per 512 walkers): not used there.  PYNQS_OP_ROW32=1: wherever it fits.
  // The LIST form that enumerates the drawn tiles again (kept records within the list, no row cache) gives way to the same kernel under
  // the same rule: a list that never fills is never flushed.
  const bool from_list = f.use_list && !f.use_cache && flush_env != 0;
  if ((f.use_flush || from_list) && sampled && have_row_f32 && row32_env != 0 && row32_fits(esz, flush_P, (uint32_t)eps_sample, gtile_f)) {
    const size_t lds_r = onepass_list_lds(p, esz, max_tiles, true, flush_P, (uint32_t)eps_sample, false, gtile_f, false, true);
    const size_t wg_old = (size_t)160 * 1024 / ((f.use_flush ? lds_flush : f.lds) + 512), wg_new = (size_t)160 * 1024 / (lds_r + 512);
    if (lds_r + 256 <= 160 * 1024 && (row32_env == 1 || wg_new > wg_old || (uint64_t)eps_sample * 8 >= max_tiles)) {
      f.use_row32 = true; f.use_flush = true; f.use_list = false; f.use_cache = false; f.use_gtile = gtile_f; f.P = flush_P; f.lds = lds_r;
    }
  }
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

extern "C" int pynqs_reduce_onepass_wants_row_f32(int64_t nbatch, int sorb, int nele, int noA, int noB, int dtype, int eps_sample, int64_t cap_doubles,
                                                  int with_tile_scratch, int without_table) {
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || !make_plan_layout(sorb, &pl) || nbatch < 0 || nbatch > 0x7fffffffll || eps_sample < 0 ||
      eps_sample > 65535 || cap_doubles < 0 || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) {
    set_error(PYNQS_EINVAL, "bad arguments");
    return -1;
  }
  if (eps_sample == 0) return 0;
  uint32_t nchunks, chunk_len, max_tiles, fixed;
  onepass_geometry(nbatch, p, true, &nchunks, &chunk_len, &max_tiles, &fixed);
  const size_t esz = dtype == PYNQS_F64 ? 8 : 4;
  const OnepassForm f = onepass_form(p, esz, max_tiles, fixed, (uint64_t)cap_doubles, eps_sample, false, chunk_len, without_table != 0, false, true);
  if (f.use_split) return 1;
  // the flushing form with draws: with io->tile_scratch its draw slots need twice the room (the list of the drawn tiles)
  const OnepassForm g = with_tile_scratch ? onepass_form(p, esz, max_tiles, fixed, (uint64_t)cap_doubles, eps_sample, false, chunk_len, without_table != 0, true, true) : f;
  return g.use_row32 ? 2 : 0;
}

extern "C" int64_t pynqs_reduce_onepass_row_f32_elements(int64_t nbatch, int sorb, int nele, int noA, int noB) {
  SDParams p;
  if (!make_sd_params(sorb, nele, noA, noB, &p) || nbatch < 0 || nbatch > 0x7fffffffll) {
    set_error(PYNQS_EINVAL, "bad arguments");
    return -1;
  }
  return (int64_t)((size_t)nbatch * draw_row_stride(p.nsd + 1));
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
                                        io->dedup_table == nullptr, have_tiles, io->row_f32 != nullptr);
  const uint32_t P = form.P;
  const bool use_list = form.use_list || form.use_flush, use_cache = form.use_cache, use_flush = form.use_flush, use_gtile = form.use_gtile;
  const bool use_split = form.use_split, use_row32 = form.use_row32;
  const size_t lds = form.lds;
  if (lds + onepass_static_lds(len) > 160 * 1024) return set_error(PYNQS_EINVAL, "row too long for the fused form (LDS): use the multi-pass entry points");
  if (!io->dedup_table && !use_list)
    return set_error(PYNQS_EINVAL, "no de-duplication table: only the LIST forms run without one (cap_doubles <= pynqs_reduce_onepass_list_capacity)");
  static const bool verbose = getenv("PYNQS_OP_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "pynqs_reduce_onepass: %s form%s, LDS %zu bytes per workgroup (walker tables %zu, max_tiles %u, list P %u), %u chunk(s) per walker\n",
            use_split ? "LIST + float32 row copy + sorted draws" : use_row32 ? (use_gtile ? "flushing LIST + float32 row copy (tile sums in global memory)" : "flushing LIST + float32 row copy") : use_flush ? "flushing LIST" : use_list ? (use_gtile ? "LIST (tile sums in global memory)" : "LIST") : "look-back", use_cache ? " with row cache" : "", lds, (size_t)lds_fixed_bytes(p), max_tiles, P, nchunks);
  // eloc.py:257-264: with draws and eps <= 0 nothing is kept (every column can be drawn); without draws |H| >= eps as it stands
  const double eps_eff = (sampled && !(eps > 0.0)) ? __builtin_inf() : eps;
  if (use_split) return launch_reduce_rowout(bra, nbatch, p, pl, chunk_len, max_tiles, plan, dtype, eps_eff, eps_sample, seed, P, lds, io, fixed, st);
  if (use_row32) return launch_reduce_flush_row32(bra, nbatch, p, pl, chunk_len, max_tiles, plan, dtype, eps_eff, eps_sample, seed, P, lds, io, fixed, use_gtile, st);
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
