// kernels_plan.hip -- the fast path of libpynqs_amd: enumeration + matrix elements on the integral plan
// (plan.h).  Same results, bit for bit, as kernels_det.hip / the reference CPU path
// (cpp_src/cpu/excitation.cpp:125-169, hamiltonian.cpp:34-50); what changes is where the bytes live:
//   * doubles read one element of the dense spin-blocked tables whose fast index follows the lanes;
//   * singles gather their nele+1 terms with lanes over the occupied orbitals (one contiguous table row
//     per single), stage them in LDS and add them in the reference's order (plan_dev.h);
//   * each excitation class has its own loop, so class parameters are scalar and no lane diverges;
//   * a lane produces TWO consecutive columns and writes them with 16-byte stores: the vector-memory
//     pipeline (TA/TD busy 67-83 % in profiles/r01_fe2s2_dropin_v3_plan_pmc_deep.txt) is paid per
//     instruction, so half as many store instructions move the same bytes.
#include "detcore.h"
#include "launch.h"
#include "plan.h"
#include "plan_dev.h"
#include "plan_tiles.h"

namespace pynqs {

// -------------------------------------------------------------------------------------------------
// Plan construction: one lane per table element; every element is h2e/h1e[...] or its negation.
template <typename T>
__global__ __launch_bounds__(kBlock) void plan_build_kernel(const T *__restrict__ h1e, const T *__restrict__ h2e, PlanLayout pl,
                                                            T *__restrict__ plan) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= pl.total) return;
  const int64_t K = pl.K, NP = pl.NP, sorb = pl.sorb;
  T v = T(0);
  if (e < pl.offVss) {  // Vab[pb][pj][pa][pi]
    int64_t r = e;
    const int pi = (int)(r % K); r /= K;
    const int pa = (int)(r % K); r /= K;
    const int pj = (int)(r % K); r /= K;
    const int pb = (int)r;
    const int hA = 2 * pi, hB = 2 * pj + 1, qA = 2 * pa, qB = 2 * pb + 1;
    const int p0 = max(hA, hB), p1 = min(hA, hB), q0 = max(qA, qB), q1 = min(qA, qB);
    v = two_body<T>(h2e, p0, p1, q0, q1);
  } else if (e < pl.offS2) {  // Vss[spin][ab][ij]
    int64_t r = e - pl.offVss;
    const int spin = (int)(r / (NP * NP)); r -= (int64_t)spin * NP * NP;
    const int ab = (int)(r / NP), ij = (int)(r - (int64_t)ab * NP);
    int m1, m0, n1, n0;
    pair_unrank(ij, m1, m0);
    pair_unrank(ab, n1, n0);
    v = two_body<T>(h2e, 2 * m1 + spin, 2 * m0 + spin, 2 * n1 + spin, 2 * n0 + spin);
  } else if (e < pl.offS1) {  // S2[spin][pm][qm][k]
    int64_t r = e - pl.offS2;
    const int k = (int)(r % sorb); r /= sorb;
    const int qm = (int)(r % K); r /= K;
    const int pm = (int)(r % K); r /= K;
    const int spin = (int)r;
    const int pp = 2 * pm + spin, q = 2 * qm + spin;
    v = two_body<T>(h2e, pp, k, q, k);
  } else if (e < pl.offD2) {  // S1[spin][pm][qm] = h1e_get(p, q) = h1e[q*sorb + p]
    int64_t r = e - pl.offS1;
    const int qm = (int)(r % K); r /= K;
    const int pm = (int)(r % K); r /= K;
    const int spin = (int)r;
    v = h1e[(int64_t)(2 * qm + spin) * sorb + (2 * pm + spin)];
  } else if (e < pl.offD1) {  // D2[p][q] = <pq||pq>
    const int64_t r = e - pl.offD2;
    const int pp = (int)(r / sorb), q = (int)(r % sorb);
    v = two_body<T>(h2e, pp, q, pp, q);
  } else if (e < pl.offD1 + sorb) {
    const int64_t pp = e - pl.offD1;
    v = h1e[pp * sorb + pp];
  }
  plan[e] = v;
}

// -------------------------------------------------------------------------------------------------
// Output stores relative to the walker's (wave-uniform) row pointers, with a 32-bit lane offset:
// a row is < 2^32 bytes (ncomb < 2^24, <= 24 B per ket).
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// NT: non-temporal stores.  The outputs are written once and never read by the kernel.  When the plan does not fit
// the L2 (large systems) keeping the output stream out of the caches leaves them to the gathers: sorb 120 0.637 ->
// 0.517 ms.  When it does fit (Fe2S2, 2 MiB) ordinary stores are faster (0.252 vs 0.285 ms): the host picks.
#define PYNQS_STORE(ptr, val) do { if constexpr (NT) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)

template <bool NT, typename T>
__device__ __forceinline__ void store_h(T *__restrict__ hrow, uint32_t col, T v) {
  PYNQS_STORE(reinterpret_cast<T *>(reinterpret_cast<char *>(hrow) + (size_t)(col * (uint32_t)sizeof(T))), v);
}
template <bool NT, typename T>
__device__ __forceinline__ void store_h2(T *__restrict__ hrow, uint32_t col, T v0, T v1) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  T2 v = {v0, v1};
  PYNQS_STORE(reinterpret_cast<T2 *>(reinterpret_cast<char *>(hrow) + (size_t)(col * (uint32_t)sizeof(T))), v);
}
template <bool NT, int LEN>
__device__ __forceinline__ void store_ket(uint64_t *__restrict__ crow, uint32_t col, const uint64_t (&ket)[LEN]) {
  char *dst = reinterpret_cast<char *>(crow) + (size_t)(col * (uint32_t)(8 * LEN));
  if constexpr (LEN == 2) {  // one 16-byte store (rows of two-word kets are 16-byte aligned)
    u64x2 v = {ket[0], ket[1]};
    PYNQS_STORE(reinterpret_cast<u64x2 *>(dst), v);
  } else {
#pragma unroll
    for (int i = 0; i < LEN; ++i) PYNQS_STORE(reinterpret_cast<uint64_t *>(dst) + i, ket[i]);
  }
}
// two consecutive kets = 2*LEN words = LEN 16-byte stores; (row base + col) is even, so the address is
// 16-byte aligned for every LEN
template <bool NT, int LEN>
__device__ __forceinline__ void store_ket2(uint64_t *__restrict__ crow, uint32_t col, const uint64_t (&k0)[LEN],
                                           const uint64_t (&k1)[LEN]) {
  u64x2 *dst = reinterpret_cast<u64x2 *>(reinterpret_cast<char *>(crow) + (size_t)(col * (uint32_t)(8 * LEN)));
  uint64_t w[2 * LEN];
#pragma unroll
  for (int i = 0; i < LEN; ++i) { w[i] = k0[i]; w[LEN + i] = k1[i]; }
#pragma unroll
  for (int i = 0; i < LEN; ++i) { u64x2 v = {w[2 * i], w[2 * i + 1]}; PYNQS_STORE(dst + i, v); }
}

// The drop-in kernel: every column of the walker's range goes to HBM (comb and Hmat in the reference layout).
typedef __attribute__((address_space(3))) uint64_t lds_u64;

template <int LEN, typename T, bool WRITE_COMB, bool NT>
struct StoreSink {
  T *__restrict__ hrow;
  uint64_t *__restrict__ crow;
  lds_u64 *stage;  // this wave's quarter of the LDS scratch (free while the wave is in a doubles tile)
  __device__ __forceinline__ void tile_begin(uint32_t) const {}
  __device__ __forceinline__ void one(uint32_t col, T h, const uint64_t (&ket)[LEN]) const {
    store_h<NT, T>(hrow, col, h);
    if constexpr (WRITE_COMB) store_ket<NT, LEN>(crow, col, ket);
  }
  __device__ __forceinline__ void pair(uint32_t col, T h0, T h1, const uint64_t (&k0)[LEN], const uint64_t (&k1)[LEN]) const {
    store_h2<NT, T>(hrow, col, h0, h1);
    if constexpr (WRITE_COMB) store_ket2<NT, LEN>(crow, col, k0, k1);
  }
  // columns c0 = b + lane and c1 = b + 64 + lane of every lane of the wave (plan_tiles.h)
  __device__ __forceinline__ void two(uint32_t c0, T h0, const uint64_t (&k0)[LEN], uint32_t c1, T h1, const uint64_t (&k1)[LEN]) const {
    store_h<NT, T>(hrow, c0, h0);
    store_h<NT, T>(hrow, c1, h1);
    if constexpr (WRITE_COMB) {
      if constexpr (LEN == 3) {
        // 24-byte kets: a lane storing its own ket covers a third of each line per instruction.  64 kets of the
        // wave are one span of 192 words: pass them through the wave's LDS quarter and let lane l store words
        // l, l + 64, l + 128: three dense 512-byte stores.  (LDS operations of one wave execute in order.)
        const uint32_t lane = threadIdx.x & 63;
        lds_u64 *st = stage;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int i = 0; i < 3; ++i) st[lane * 3 + i] = half ? k1[i] : k0[i];
          __builtin_amdgcn_wave_barrier();
          uint64_t *dst = crow + (size_t)((half ? c1 : c0) - lane) * 3;
#pragma unroll
          for (int k = 0; k < 3; ++k) PYNQS_STORE(dst + lane + 64 * k, (uint64_t)st[lane + 64 * k]);
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        store_ket<NT, LEN>(crow, c0, k0);
        store_ket<NT, LEN>(crow, c1, k1);
      }
    }
  }
};

template <int LEN, typename T, bool WRITE_COMB, bool NT>
__global__ __launch_bounds__(kBlock) void comb_hij_plan_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl,
                                                               uint32_t nchunks, uint32_t chunk_len, bool xcd_map,
                                                               const T *__restrict__ plan, uint64_t *__restrict__ comb,
                                                               T *__restrict__ hmat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t next_tile;
  uint64_t walker;
  uint32_t chunk;
  map_workgroup(nchunks, xcd_map, walker, chunk);
  if (threadIdx.x == 0) next_tile = 0;
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);  // ends with a workgroup barrier
  const uint32_t ncomb = p.nsd + 1;
  static_assert(LEN != 3 || (kDiagTile / 4) * sizeof(T) >= 192 * 8, "a wave's scratch quarter must hold 64 three-word kets");
  StoreSink<LEN, T, WRITE_COMB, NT> sink{hmat + (size_t)walker * ncomb, comb + (size_t)walker * ncomb * LEN,
                                     (lds_u64 *)(reinterpret_cast<T *>(L.scratch) + (threadIdx.x >> 6) * (kDiagTile / 4))};
  const uint32_t odd_base = (uint32_t)((walker * (uint64_t)ncomb) & 1u);  // 16-byte alignment of the pair stores
  visit_tiles<LEN, T>(p, pl, L, nocc, plan, wk, nchunks, chunk, chunk_len, odd_base, &next_tile, sink);
}

}  // namespace pynqs

// =================================================================================================
using namespace pynqs;

extern "C" int64_t pynqs_plan_bytes(int sorb, int dtype) {
  PlanLayout pl;
  if (!make_plan_layout(sorb, &pl) || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return -1;
  return pl.total * (dtype == PYNQS_F64 ? 8 : 4);
}

extern "C" int pynqs_plan_build(const void *h1e, const void *h2e, int sorb, int dtype, void *plan, void *stream) {
  pynqs::DeviceScope device_scope_(h1e);
  PlanLayout pl;
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (dtype != PYNQS_F32 && dtype != PYNQS_F64) return set_error(PYNQS_EINVAL, "bad dtype");
  if (!h1e || !h2e || !plan) return set_error(PYNQS_EINVAL, "null pointer");
  const uint64_t grid = ((uint64_t)pl.total + kBlock - 1) / kBlock;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYNQS_F64)
    hipLaunchKernelGGL((plan_build_kernel<double>), dim3((uint32_t)grid), dim3(kBlock), 0, st, (const double *)h1e,
                       (const double *)h2e, pl, (double *)plan);
  else
    hipLaunchKernelGGL((plan_build_kernel<float>), dim3((uint32_t)grid), dim3(kBlock), 0, st, (const float *)h1e,
                       (const float *)h2e, pl, (float *)plan);
  return check_launch("plan_build");
}

template <int LEN, typename T>
static int launch_plan(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, const T *plan,
                       uint64_t *comb, T *hmat, hipStream_t st) {
  const uint32_t ncomb = p.nsd + 1;
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, ncomb, &nchunks, &chunk_len);
  const size_t lds = lds_bytes(p, sizeof(T));
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large: nbatch*nchunks > 2^31-1");
  // non-temporal output stores once the plan is larger than one XCD's L2 (4 MiB); PYNQS_NT=0/1 overrides
  static const int nt_env = getenv("PYNQS_NT") ? atoi(getenv("PYNQS_NT")) : -1;
  const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)pl.total * sizeof(T) > ((size_t)4 << 20);
#define PYNQS_PLAN_LAUNCH(WC, NTV)                                                                                          \
  hipLaunchKernelGGL((comb_hij_plan_kernel<LEN, T, WC, NTV>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, pl, nchunks, \
                     chunk_len, xcd_mapping(nchunks), plan, comb, hmat)
  if (comb) { if (nt) PYNQS_PLAN_LAUNCH(true, true); else PYNQS_PLAN_LAUNCH(true, false); }
  else { if (nt) PYNQS_PLAN_LAUNCH(false, true); else PYNQS_PLAN_LAUNCH(false, false); }
#undef PYNQS_PLAN_LAUNCH
  return check_launch("comb_hij_plan");
}

extern "C" int pynqs_comb_hij_fused_plan(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                         const void *plan, int dtype, uint64_t *comb, void *hmat, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  PlanLayout pl;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (!make_plan_layout(sorb, &pl)) return set_error(PYNQS_EINVAL, "plan needs an even sorb in [2, 192]");
  if (nbatch < 0 || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return set_error(PYNQS_EINVAL, "bad nbatch/dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !plan || !hmat) return set_error(PYNQS_EINVAL, "null pointer");
  if (nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "nbatch too large");
  // the paired 16-byte stores need 16-byte aligned output buffers (any allocator gives that)
  if (((uintptr_t)hmat & 15u) || ((uintptr_t)comb & 15u)) return set_error(PYNQS_EINVAL, "comb/hmat must be 16-byte aligned");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  DISPATCH_LEN(len, rc = dtype == PYNQS_F64 ? launch_plan<LEN, double>(bra, nbatch, p, pl, (const double *)plan, comb, (double *)hmat, st)
                                            : launch_plan<LEN, float>(bra, nbatch, p, pl, (const float *)plan, comb, (float *)hmat, st));
  return rc;
}
