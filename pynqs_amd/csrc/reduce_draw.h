// reduce_draw.h -- the draws of the semi-stochastic REDUCE front end for rows of up to 8192 columns (round 4; the form the Fe2S2 example
// runs: vmc/energy/eloc.py:257-298 -- |H| >= eps kept, N draws of the rest proportional to |H| -- and `Func`, vmc/energy/flip.py:29-63).
// Called by the LIST kernel (reduce_list.h, ROWOUT) once a workgroup has enumerated its walker's row, left the sub-eps matrix elements as
// float32 in global memory (io->row_f32; kept columns: 0.0f, they are never drawn), summed the exact |H| per tile and resolved the kept records.
//
// Round 3 cached the row in float64, read it back twice (tile sums, then per-tile scans in LDS) and drew hierarchically: 49 of a workgroup's
// 118 us and 1.5 GB of L2 misses per launch -- and the kernel turned out to be bound by its VECTOR INSTRUCTIONS (305 M per launch at 0.7 of
// the issue rate, profiles/), not by that traffic: what counts is instructions per draw.  Here every draw is one lane's work, start to end:
//   segments      the row in segments of 16 columns; their sums of the float32 values in float64 and a block scan give the segments' starting
//                 sums (LDS, <= 513 doubles)
//   draws         draw k: u S' (S' = the sum of the float32 values; counter-based generator keyed (seed, walker, k)) finds its segment by binary
//                 search, reads the segment's 16 elements back from the row (four 16-byte loads, L2) and walks them: ~150 instructions, no
//                 divergence, no queues
//   hit counts    a bitmap of the drawn columns (LDS atomicOr); the RANK of a column's bit (block scan of the words' popcounts) is its place among
//                 the drawn records -- ascending columns like the reference's unique(sorted=True), whatever the timing -- and every draw adds
//                 one to the count of its rank: no sort, no second look at the row
//   amplitudes    kets from the walker's tables (decode), the de-duplication probes of a thread's four records issued side by side (one-word
//                 determinants: key and row of a slot come together), ONE allocation of rows per 1024 records, direct links
// P(column j) = w32_j / S' -- |H_j| rounded to float32, relative 6e-8 -- the multinomial law of torch.multinomial(prob, N, replacement=True) up
// to that rounding; the weight of a drawn record is exactly the reference's (c / N) sign(H_j) S with the float64 S.
// LDS (over the staging scratch of the enumeration): pfx f64[513] | bitmap u32[256] | kpre u32[256]; the drawn records' slots rec u32[N] and
// the kept records' columns are the caller's (reduce_list.h).  Kept and drawn records are resolved together: one round of probes, one
// allocation of rows per workgroup instead of two.
#pragma once
#include "reduce_common.h"

namespace pynqs {

constexpr int kDrawSeg = 16;                                               // columns per segment of the row
constexpr uint32_t kDrawMaxCols = 2u * kBlock * kDrawSeg, kDrawMaxDraws = 16383;  // (a distinct drawn column waits in LDS as count << 16 | sign << 15 | column)
constexpr uint32_t kNoRec = 0xffffffffu;

__host__ __device__ inline size_t draw_lds_bytes(uint32_t nsample) {
  auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
  (void)nsample;  // (the drawn records' slots are the caller's)
  return al((size_t)(2 * kBlock + 1) * 8) + (size_t)kBlock * 8;
}
// elements between the rows of io->row_f32 (rows start 64-byte aligned and end in zeros)
__host__ __device__ inline size_t draw_row_stride(uint32_t ncomb) { return ((size_t)ncomb + kDrawSeg - 1) & ~(size_t)(kDrawSeg - 1); }

// De-duplication probes of K records per thread, side by side (one-word determinants, no wave-function table): the first loads of all K
// records are in flight together, and the key comes with the slot's row (same 16 bytes), so that a determinant that is already there with
// its row published needs no second look.  lk: slot (>= 0) or -1 (table full); rowhint: the slot's row as seen (or -1).
template <typename T, int K>
__device__ __forceinline__ void probe_k_oneword(const OnepassOut<T> &o, const uint64_t (&ket)[K][1], const bool (&act)[K], int32_t (&lk)[K],
                                                bool (&won)[K], int32_t (&rowhint)[K], uint32_t *full_flag) {
  uint32_t s[K];
  bool open[K];
  const bool full = *full_flag != 0u;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    won[k] = false; rowhint[k] = -1;
    open[k] = act[k] && !full && !(o.debug & 1u);
    if (act[k]) lk[k] = -1;
    s[k] = (uint32_t)(hash_of<1>(ket[k]) >> 17) & o.dedup_mask;
  }
  for (uint32_t probes = 0;; ++probes) {
    unsigned long long cur[K];
    int32_t rw[K];
    if constexpr (K == 4) {
      // key and row of the four slots: four 16-byte loads at agent scope (what a relaxed atomic load compiles to, `sc1`, twice as wide: a
      // slot's words are each written atomically and only ever go from empty to their final value, so a torn view is a valid earlier one),
      // in flight together; a lane without an open record reads the table's first slot.  (`nt` on these loads: 577 -> 732 us per launch.)
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 x0, x1, x2, x3;
      const uint64_t *a0 = o.dedup + (open[0] ? (size_t)s[0] * 2 : 0), *a1 = o.dedup + (open[1] ? (size_t)s[1] * 2 : 0),
                     *a2 = o.dedup + (open[2] ? (size_t)s[2] * 2 : 0), *a3 = o.dedup + (open[3] ? (size_t)s[3] * 2 : 0);
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                   "global_load_dwordx4 %1, %5, off sc1\n\t"
                   "global_load_dwordx4 %2, %6, off sc1\n\t"
                   "global_load_dwordx4 %3, %7, off sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                   : "memory");
      cur[0] = ((unsigned long long)x0[1] << 32) | x0[0]; rw[0] = (int32_t)x0[2];
      cur[1] = ((unsigned long long)x1[1] << 32) | x1[0]; rw[1] = (int32_t)x1[2];
      cur[2] = ((unsigned long long)x2[1] << 32) | x2[0]; rw[2] = (int32_t)x2[2];
      cur[3] = ((unsigned long long)x3[1] << 32) | x3[0]; rw[3] = (int32_t)x3[2];
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        cur[k] = 0ull; rw[k] = -1;
        if (open[k]) {
          unsigned long long *kp = reinterpret_cast<unsigned long long *>(o.dedup + (size_t)s[k] * 2);
          cur[k] = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rw[k] = __hip_atomic_load(reinterpret_cast<int32_t *>(kp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (open[k] && cur[k] == ~0ull && (o.debug & 2048u)) { open[k] = false; continue; }  // (timing ablation: look, never insert)
      if (open[k] && cur[k] == ~0ull) {
        unsigned long long *kp = reinterpret_cast<unsigned long long *>(o.dedup + (size_t)s[k] * 2);
        cur[k] = atomicCAS(kp, ~0ull, (unsigned long long)ket[k][0]);
        if (cur[k] == ~0ull) { won[k] = true; lk[k] = (int32_t)s[k]; open[k] = false; }
        else rw[k] = -1;  // (somebody else's key arrived in between: its row was not in our load)
      }
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (open[k]) {
        if (cur[k] == ket[k][0]) { lk[k] = (int32_t)s[k]; rowhint[k] = rw[k]; open[k] = false; }
        else { s[k] = (s[k] + 1) & o.dedup_mask; any = true; }
      }
    }
    if (!any) break;
    if (probes + 1 >= kProbeLimit) {  // a table this full is an overflow: the call is repeated with a larger one
      atomicOr(reinterpret_cast<unsigned int *>(o.counters + 1), 2u);
      *full_flag = 1u;
      break;
    }
  }
}

// inclusive block scan over the kBlock threads (fixed order of additions); *total = the sum.  Contains barriers.
template <typename V>
__device__ __forceinline__ V draw_block_scan(V v, V *s_part, V *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  V incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const V ov = __shfl_up(incl, d);
    if (lane >= d) incl += ov;
  }
  __syncthreads();  // (s_part free)
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  V before = V(0), tot = V(0);
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) before += s_part[w];
    tot += s_part[w];
  }
  *total = tot;
  return before + incl;
}

// Every thread of the workgroup must call (barriers inside).  lds: draw_lds_bytes(nsample) bytes nobody else uses any more.
// w: this walker's row of float32 elements, kDrawSeg-aligned and padded with zeros to a multiple of kDrawSeg.
// rec: [nsample] words of LDS for the drawn records; kcol / nkept / seg_base: the kept records' columns in slot order (0xffffffff: an empty
// slot), their number and the first slot of the segment -- they are resolved here, together with the drawn ones.
template <int LEN, typename T>
__device__ __forceinline__ void rowout_draws(unsigned char *lds, uint32_t *rec, const uint32_t *kcol, uint32_t nkept, int64_t seg_base, const SDParams &p,
                                             const LdsLayout &L, const Walker<LEN> &wk, const float *__restrict__ w, uint32_t nsample, uint64_t seed,
                                             uint64_t walker, double Srow, const OnepassOut<T> &o, uint32_t *bw_cnt, int32_t *bw_base, uint32_t *s_full,
                                             double *s_part, uint32_t *s_parti) {
  constexpr int NT = kBlock, K = 4, SEG = kDrawSeg;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const uint32_t ncomb = p.nsd + 1;
  const uint32_t nseg = (ncomb + SEG - 1) / SEG, nbw = (ncomb + 31) / 32;  // <= 2 NT segments, <= NT bitmap words
  const int64_t sbase = (int64_t)walker * nsample;
  auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
  double *pfx = reinterpret_cast<double *>(lds);                                      // [2 NT + 1] the segments' starting sums
  uint32_t *bitmap = reinterpret_cast<uint32_t *>(lds + al((size_t)(2 * NT + 1) * 8));  // [NT] the drawn columns
  uint32_t *kpre = bitmap + NT;                                                       // [NT] drawn columns before each word
  // rec[N]: per distinct drawn column (ascending): count << 16 | sign << 15 | column
  // ---- the segments' sums of the float32 values in float64 (thread t: segments 2 t, 2 t + 1), their starting sums ----
  double ls[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t sgm = 2u * (uint32_t)tid + h;
    double a = 0.0;
    if (sgm < nseg) {
#pragma unroll
      for (int q4 = 0; q4 < SEG / 4; ++q4) {
        const f4 x = *reinterpret_cast<const f4 *>(w + (size_t)sgm * SEG + 4 * q4);  // (nontemporal loads here: 598 -> 718 us)
        a += (double)fabsf(x[0]); a += (double)fabsf(x[1]); a += (double)fabsf(x[2]); a += (double)fabsf(x[3]);
      }
    }
    ls[h] = a;
  }
  bitmap[tid] = 0u;
  for (uint32_t i = tid; i < nsample; i += NT) rec[i] = 0u;
  double total;
  const double incl = draw_block_scan<double>(ls[0] + ls[1], s_part, &total);
  pfx[2 * tid] = incl - (ls[0] + ls[1]);
  pfx[2 * tid + 1] = incl - ls[1];
  if (tid == NT - 1) pfx[2 * NT] = total;
  __syncthreads();
#ifdef PYNQS_OP_STAMPS
  if (threadIdx.x == 0 && walker < 8192) g_stamps[walker][7] = wall_clock64();
#endif
  const double scale = Srow / (double)nsample;
  const uint64_t key = op_mix64((o.seed_dev ? seed + *o.seed_dev : seed) ^ op_mix64(walker));
  const bool any_width = total > 0.0;
  if (o.debug & 64u) return;
  // ---- the draws, one per lane and step: segment by binary search, column by a walk over the segment's 16 elements (read back from the row) ----
  // (rounds of K draws per thread: the located columns wait in registers between the two passes)
  for (uint32_t k0 = 0; k0 < nsample; k0 += K * NT) {
    uint32_t dcol[K];
#pragma unroll 1
    for (int jj = 0; jj < K; ++jj) {
      const uint32_t k = k0 + (uint32_t)jj * NT + (uint32_t)tid;
      uint32_t found = kNoRec;
      if (k < nsample && any_width) {
        const uint64_t r = op_mix64(key ^ ((uint64_t)(k + 1u) * 0x9e3779b97f4a7c15ull));
        const double target = (double)(r >> 11) * 0x1.0p-53 * total;
        uint32_t lo = 0, hi = 2 * NT;  // the last segment whose starting sum is <= target
#pragma unroll
        for (int it = 0; it < 9; ++it) {
          const uint32_t mid = (lo + hi) >> 1;
          if (pfx[mid] <= target) lo = mid; else hi = mid;
        }
        if (lo >= nseg) lo = nseg - 1;
        const double res = target - pfx[lo];
        double acc = 0.0;
        int kf = -1, klast = -1;
        uint32_t neg = 0;
#pragma unroll
        for (int q4 = 0; q4 < SEG / 4; ++q4) {
          const f4 x = *reinterpret_cast<const f4 *>(w + (size_t)lo * SEG + 4 * q4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double a = (double)fabsf(x[e]);
            acc += a;
            if (a > 0.0) {
              klast = 4 * q4 + e;
              if (kf < 0 && acc > res) kf = 4 * q4 + e;
            }
            neg |= (x[e] < 0.0f ? 1u : 0u) << (4 * q4 + e);
          }
        }
        if (kf < 0) kf = klast;  // (rounding at the end of the segment: its last column of positive width)
        if (kf >= 0) found = (lo * SEG + (uint32_t)kf) | (((neg >> kf) & 1u) << 15);
        else {  // (a segment without any width, chosen by rounding: the nearest column of positive width after it, else before it; never seen)
          for (uint32_t c = lo * SEG; c < ncomb && found == kNoRec; ++c)
            if (fabsf(w[c]) > 0.0f) found = c | ((w[c] < 0.0f ? 1u : 0u) << 15);
          for (int32_t c = (int32_t)(lo * SEG) - 1; c >= 0 && found == kNoRec; --c)
            if (fabsf(w[c]) > 0.0f) found = (uint32_t)c | ((w[c] < 0.0f ? 1u : 0u) << 15);
        }
        if (found != kNoRec) atomicOr(&bitmap[(found & 0x7fffu) >> 5], 1u << (found & 31u));
      }
      dcol[jj] = found;
    }
    __syncthreads();
    // drawn columns before each word of the bitmap (so far: the rounds before this one only add bits, the counts are redone)
    if (k0 + K * NT >= nsample) {
      uint32_t dummy;
      const uint32_t c = tid < (int)nbw ? (uint32_t)__popc(bitmap[tid]) : 0u;
      const uint32_t inc = draw_block_scan<uint32_t>(c, s_parti, &dummy);
      kpre[tid] = inc - c;
      __syncthreads();
    }
    // (several rounds -- more than 1024 draws: the columns of the earlier rounds wait in the draw slots' link words)
    if (nsample > K * NT) {
#pragma unroll
      for (int jj = 0; jj < K; ++jj) {
        const uint32_t k = k0 + (uint32_t)jj * NT + (uint32_t)tid;
        if (k < nsample) o.srec_link[sbase + k] = (int32_t)dcol[jj];
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < K; ++jj) {
        const uint32_t f = dcol[jj];
        if (f != kNoRec) {
          const uint32_t c = f & 0x7fffu, wd = c >> 5;
          const uint32_t r = kpre[wd] + (uint32_t)__popc(bitmap[wd] & ((1u << (c & 31u)) - 1u));
          atomicAdd(&rec[r], 0x10000u);
          atomicOr(&rec[r], f);
        }
      }
    }
  }
  if (nsample > K * NT) {  // (the counting pass of the many-draws case, from the parked columns)
    __syncthreads();
    for (uint32_t k = tid; k < nsample; k += NT) {
      const uint32_t f = (uint32_t)o.srec_link[sbase + k];
      if (f != kNoRec) {
        const uint32_t c = f & 0x7fffu, wd = c >> 5;
        const uint32_t r = kpre[wd] + (uint32_t)__popc(bitmap[wd] & ((1u << (c & 31u)) - 1u));
        atomicAdd(&rec[r], 0x10000u);
        atomicOr(&rec[r], f);
      }
    }
  }
  __syncthreads();
  const uint32_t nd = nbw ? kpre[nbw - 1] + (uint32_t)__popc(bitmap[nbw - 1]) : 0u;  // distinct drawn columns
#ifdef PYNQS_OP_STAMPS
  if (threadIdx.x == 0 && walker < 8192) g_stamps[walker][8] = wall_clock64();
#endif
  if (o.debug & 128u) return;
  // ---- ALL records of the walker, kept (slot order) and drawn (ascending columns, like the reference's unique), four per thread side by
  //      side: columns, weights, kets, probes, ONE allocation of rows for the new determinants, links ----
  constexpr int32_t kNoRecord = -0x7fffffff;
  // (record stores are streaming stores: the next kernels read them, nothing in this one does, and the XCD's L2 is wanted for the plan and the rows)
  for (uint32_t i = nd + tid; i < nsample; i += NT) __builtin_nontemporal_store((int32_t)-1, o.srec_col + sbase + i);  // (unused draw slots)
  const uint32_t nrec = nkept + nd;
  for (uint32_t i0 = 0; i0 < nrec; i0 += K * NT) {
    bool act[K], won[K];
    uint32_t slot[K];
    int32_t lk[K], rowhint[K];
    uint64_t ket[K][LEN];
    int64_t dst[K];  // >= 0: kept record, rec_link[dst]; < 0: drawn record, srec_link[-1 - dst]
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t i = i0 + k * NT + tid;
      act[k] = false; won[k] = false; slot[k] = 0; lk[k] = kNoRecord; rowhint[k] = -1; dst[k] = 0;
#pragma unroll
      for (int ww = 0; ww < LEN; ++ww) ket[k][ww] = wk.w[ww];
      if (i >= nrec) continue;
      uint32_t col;
      if (i < nkept) {
        col = kcol[i];
        if (col == kNoRec) continue;
        dst[k] = seg_base + i;
      } else {
        const uint32_t e = rec[i - nkept];
        col = e & 0x7fffu;
        const int64_t at = sbase + (i - nkept);
        dst[k] = -1 - at;
        __builtin_nontemporal_store((int32_t)col, o.srec_col + at);
        const double val = scale * (double)(e >> 16);
        __builtin_nontemporal_store((T)(((e >> 15) & 1u) ? -val : val), o.srec_w + at);
      }
      act[k] = true;
      if (col) {
        const Excitation x = decode(col - 1, p, L);
        make_ket<LEN>(wk, x, ket[k]);
      }
      uint64_t *onv = dst[k] >= 0 ? o.rec_onv : o.srec_onv;
      if (onv) {
        const int64_t g = dst[k] >= 0 ? dst[k] : -1 - dst[k];
#pragma unroll
        for (int ww = 0; ww < LEN; ++ww) __builtin_nontemporal_store(ket[k][ww], onv + g * LEN + ww);
      }
    }
    bool fast = false;
    if constexpr (LEN == 1) fast = o.dedup != nullptr && o.lut == nullptr;
    if (fast) {
      if constexpr (LEN == 1) probe_k_oneword<T, K>(o, ket, act, lk, won, rowhint, s_full);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (act[k]) lk[k] = probe_amplitude<LEN, T>(o, ket[k], won[k], s_full);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) slot[k] = (uint32_t)lk[k];
    int32_t rows[K];
    allocate_batch_k<LEN, T, K>(o, p.sorb, won, slot, ket, bw_cnt, bw_base, rows);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!act[k]) continue;
      int32_t link;
      if (lk[k] >= 0 && rows[k] < 0 && rowhint[k] >= 0 && (uint32_t)rowhint[k] < o.ucap) link = rowhint[k] | kDirectLink;
      else if (o.debug & 4096u) link = lk[k];  // (timing ablation: no second look at a slot whose row was not out yet)
      else link = final_link<LEN, T>(o, lk[k], rows[k]);
      __builtin_nontemporal_store(link, dst[k] >= 0 ? o.rec_link + dst[k] : o.srec_link + (-1 - dst[k]));
    }
  }
}

}  // namespace pynqs
