// kernels_det.hip -- drop-in determinant kernels of libpynqs_amd (gfx950).
//   comb_hij_kernel : enumerate S+D excitations (+ <x|H|x'>)      [get_comb_hij_fused / get_comb_tensor]
//   hij_pairs_kernel: generic bra/ket pairs, 3-D and 2-D mode      [get_hij_torch]
//   onv_to_pm1 / pm01_to_onv / lut_search                          [onv_to_tensor / tensor_to_onv / wavefunction_lut]
// Reference behaviour: cpp_src/cpu/*.cpp, cpp_src/tensor/cpu_tensor.cpp (cited per kernel).
#include "detcore.h"
#include "launch.h"

namespace pynqs {

// -------------------------------------------------------------------------------------------------
// Fused enumerate + matrix element.  cpu_tensor.cpp:220-272 / excitation.cpp:125-169.
// grid = nbatch * nchunks workgroups of 256; workgroup (walker, chunk) covers columns
// [chunk*chunk_len, (chunk+1)*chunk_len) of that walker's row.  Column 0 (x itself) only gets its comb
// entry from the loop; Hmat[:,0] is produced by diag_phase in the walker's first workgroup.
template <int LEN, typename T, bool WRITE_COMB, bool WITH_H>
__global__ __launch_bounds__(kBlock) void comb_hij_kernel(const uint64_t *__restrict__ bra, SDParams p,
                                                          uint32_t nchunks, uint32_t chunk_len,
                                                          const T *__restrict__ h1e, const T *__restrict__ h2e,
                                                          uint64_t *__restrict__ comb, T *__restrict__ hmat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t wg = blockIdx.x;
  const uint64_t walker = wg / nchunks;
  const uint32_t chunk = (uint32_t)(wg - walker * nchunks);
  Walker<LEN> wk;
  load_walker<LEN>(bra + walker * LEN, wk);
  const LdsLayout L = carve_lds(smem, p);
  const int nocc = build_walker_tables<LEN>(wk, p, L);

  const uint32_t ncomb = p.nsd + 1;
  const uint32_t lo = chunk * chunk_len;
  const uint32_t hi = min(lo + chunk_len, ncomb);
  const size_t row = (size_t)walker * ncomb;
  if constexpr (WITH_H) {
    if (chunk == 0) diag_phase<T>(p, L, h1e, h2e, hmat + row);
  }
  for (uint32_t k = lo + threadIdx.x; k < hi; k += kBlock) {
    uint64_t ket[LEN];
    if (k == 0) {
      if constexpr (WRITE_COMB) {
#pragma unroll
        for (int i = 0; i < LEN; ++i) comb[row * LEN + i] = wk.w[i];
      }
      continue;
    }
    const Excitation x = decode(k - 1, p, L);
    if constexpr (WRITE_COMB) {
      make_ket<LEN>(wk, x, ket);
#pragma unroll
      for (int i = 0; i < LEN; ++i) comb[(row + k) * LEN + i] = ket[i];
    }
    if constexpr (WITH_H) hmat[row + k] = element<T>(x, p, L, nocc, h1e, h2e);
  }
}

// -------------------------------------------------------------------------------------------------
// Generic pairs, hamiltonian.cpp:53-103 + onstate.cpp:10-55.  One lane per (i, j); consecutive lanes
// take consecutive j so ket reads and Hmat writes are coalesced.
template <int LEN>
__device__ __forceinline__ int sign_below(const uint64_t (&w)[LEN], int n) {
  uint32_t par = 0;
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    const int word = n >> 6;
    uint64_t m = i < word ? ~0ull : (i == word ? ((1ull << (n & 63)) - 1ull) : 0ull);
    par ^= (uint32_t)__popcll(w[i] & m);
  }
  return par & 1u;
}

// <x|H|x> of lane `src`'s determinant, by the whole wave: the nele(nele+1)/2 terms (hamiltonian.cpp:34-50: for p ascending h(p,p), then
// <pq||pq> for the occupied q < p ascending) are gathered 512 at a time into the wave's LDS tile, one term per lane and pass, and lane
// `src` adds them in the reference's order (bit-identical).  A lane evaluating its diagonal alone spends ~40 instructions and an L2
// round trip per term while its 63 neighbours wait: 0.28 ms for the 4096 diagonal pairs of a 4096 x 4096 Fe2S2 call (tools/hij_diag_time.py).
// List entries past the determinant's electron count read as orbital 0, like the reference's zero-initialised olst[MAX_NELE].
constexpr int kHijDiagTile = 512;

template <int LEN, typename T>
__device__ __forceinline__ T diag_by_wave(const uint64_t (&mine)[LEN], int src, const T *__restrict__ h1e, const T *__restrict__ h2e, int sorb,
                                          int nele, T *__restrict__ tile, uint8_t *__restrict__ occ) {
  const int lane = threadIdx.x & 63;
  uint64_t x[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) x[w] = __shfl(mine[w], src);
  __builtin_amdgcn_wave_barrier();
  int before = 0;
#pragma unroll
  for (int w = 0; w < LEN; ++w) {
    if ((x[w] >> lane) & 1ull) occ[before + __popcll(x[w] & ((1ull << lane) - 1ull))] = (uint8_t)(64 * w + lane);
    before += __popcll(x[w]);
  }
  for (int k = before + lane; k < nele; k += 64) occ[k] = 0;
  __builtin_amdgcn_wave_barrier();
  const int nterms = nele * (nele + 1) / 2;
  T acc = T(0);
  for (int base = 0; base < nterms; base += kHijDiagTile) {
    const int end = min(base + kHijDiagTile, nterms);
    for (int t = base + lane; t < end; t += 64) {
      int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > t) --a;
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      const int pos = t - a * (a + 1) / 2;
      const int pa = occ[a];
      tile[t - base] = pos == 0 ? h1e[(size_t)pa * sorb + pa] : two_body<T>(h2e, pa, occ[pos - 1], pa, occ[pos - 1]);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == src) {
      const int cnt = end - base;
      for (int t = 0; t < cnt; t += 8) {  // eight LDS reads in flight, added in order
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tile[min(t + u, kHijDiagTile - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (t + u < cnt) acc += v[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  return acc;
}

template <int LEN, typename T>
__global__ __launch_bounds__(kBlock) void hij_pairs_kernel(const uint64_t *__restrict__ bra, uint64_t n,
                                                           const uint64_t *__restrict__ ket, uint64_t m, int ket_is_3d,
                                                           const T *__restrict__ h1e, const T *__restrict__ h2e, int sorb,
                                                           int nele, T *__restrict__ hmat) {
  __shared__ T dtile[kBlock / 64][kHijDiagTile];
  __shared__ uint8_t docc[kBlock / 64][192];
  const uint64_t idx0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = idx0 < n * m;  // (no early exit: the diagonal pairs below are evaluated by whole waves)
  const uint64_t idx = valid ? idx0 : n * m - 1;
  const uint64_t i = idx / m, j = idx - i * m;
  uint64_t b[LEN], k[LEN];
  const uint64_t *kp = ket + ((ket_is_3d ? i * m : 0) + j) * LEN;
  int nc = 0, na = 0;
#pragma unroll
  for (int w = 0; w < LEN; ++w) {
    b[w] = bra[i * LEN + w];
    k[w] = kp[w];
    const uint64_t d = b[w] ^ k[w];
    nc += __popcll(d & b[w]);
    na += __popcll(d & k[w]);
  }
  T val = T(0);
  uint64_t diag_lanes = __ballot(valid && nc == 0 && na == 0);
  while (diag_lanes) {  // wave-uniform
    const int src = __builtin_ctzll(diag_lanes);
    diag_lanes &= diag_lanes - 1;
    const T v = diag_by_wave<LEN, T>(b, src, h1e, h2e, sorb, nele, dtile[threadIdx.x >> 6], docc[threadIdx.x >> 6]);
    if ((int)(threadIdx.x & 63) == src) val = v;
  }
  if (nc == 0 && na == 0) {
  } else if ((nc == 1 && na == 1) || (nc == 2 && na == 2)) {
    int cre[2] = {0, 0}, ann[2] = {0, 0};
    int ic = 0, ia = 0;
#pragma unroll
    for (int w = LEN - 1; w >= 0; --w) {  // highest orbital first (onstate.cpp:34-55)
      const uint64_t d = b[w] ^ k[w];
      uint64_t c = d & b[w], a = d & k[w];
      while (c) { const int bit = 63 - __builtin_clzll(c); if (ic == 0) cre[0] = w * 64 + bit; else cre[1] = w * 64 + bit; ++ic; c &= ~(1ull << bit); }
      while (a) { const int bit = 63 - __builtin_clzll(a); if (ia == 0) ann[0] = w * 64 + bit; else ann[1] = w * 64 + bit; ++ia; a &= ~(1ull << bit); }
    }
    if (nc == 1) {
      const int hp = cre[0], q = ann[0];
      T acc = T(0);
      acc += h1e[(size_t)q * sorb + hp];
#pragma unroll
      for (int w = 0; w < LEN; ++w) {
        uint64_t bits = b[w];
        while (bits) {
          const int bit = 63 - __builtin_clzll(bits);
          const int o = 64 * w + bit;
          acc += two_body<T>(h2e, hp, o, q, o);
          bits &= ~(1ull << bit);
        }
      }
      const uint32_t par = sign_below<LEN>(b, hp) ^ sign_below<LEN>(k, q);
      val = par ? -acc : acc;
    } else {
      const uint32_t par = sign_below<LEN>(b, cre[0]) ^ sign_below<LEN>(b, cre[1]) ^ sign_below<LEN>(k, ann[0]) ^
                           sign_below<LEN>(k, ann[1]);
      const T v = two_body<T>(h2e, cre[0], cre[1], ann[0], ann[1]);
      val = par ? -v : v;
    }
  }
  if (valid) hmat[idx] = val;
}

// -------------------------------------------------------------------------------------------------
// onstate.h:45-63 : +1 occupied / -1 empty.  The output is one flat array; a lane writes 16 consecutive
// bytes of it (2 doubles / 4 floats, possibly across a row boundary) with one store.
template <typename T>
__global__ __launch_bounds__(kBlock) void onv_to_pm1_kernel(const uint64_t *__restrict__ bra, uint64_t n, int sorb, int len,
                                                            T *__restrict__ out) {
  constexpr int V = 16 / sizeof(T);
  typedef T TV __attribute__((ext_vector_type(V)));
  const uint64_t total = n * (uint64_t)sorb;
  const uint64_t e0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * V;
  if (e0 >= total) return;
  uint64_t w = e0 / (uint32_t)sorb;
  int o = (int)(e0 - w * (uint32_t)sorb);
  // one load per determinant word the lane needs (the V elements of a lane lie in one word unless a row or word boundary cuts them)
  T v[V];
  uint64_t word = bra[w * len + (o >> 6)];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    v[i] = ((word >> (o & 63)) & 1ull) ? T(1) : T(-1);
    if (++o == sorb) { o = 0; ++w; }
    if (i + 1 < V && (o & 63) == 0 && e0 + i + 1 < total) word = bra[w * len + (o >> 6)];
  }
  if (e0 + V <= total) {
    TV pack;
#pragma unroll
    for (int i = 0; i < V; ++i) pack[i] = v[i];
    *reinterpret_cast<TV *>(out + e0) = pack;
  } else {
    for (int i = 0; e0 + i < total; ++i) out[e0 + i] = v[i];
  }
}

// cpu_tensor.cpp:8-44 : one wave per output word; lane l tests byte l, ballot packs the word.
__global__ __launch_bounds__(kBlock) void pm01_to_onv_kernel(const uint8_t *__restrict__ occ, uint64_t n, int sorb, int len,
                                                             uint64_t *__restrict__ out) {
  const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n * (uint64_t)len) return;
  const uint64_t w = wave / (uint32_t)len;
  const int word = (int)(wave - w * (uint32_t)len);
  const int o = word * 64 + lane;
  const bool set = o < sorb && occ[w * (uint64_t)sorb + o] == 1;
  const uint64_t bits = __ballot(set);
  if (lane == 0) out[wave] = bits;
}

// Same, for rows that are 8-byte aligned (sorb % 8 == 0, aligned base): one lane per output word, 8 input bytes per
// load, each tested against 1 with the carry-free zero-byte test and packed with one multiplication.  The wave still
// reads one contiguous span and every byte of it is used (3-4x the rate of the byte-per-lane kernel).
__global__ __launch_bounds__(kBlock) void pm01_to_onv_kernel_x8(const uint8_t *__restrict__ occ, uint64_t n, int sorb, int len,
                                                                uint64_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * (uint64_t)len) return;
  const uint64_t w = i / (uint32_t)len;
  const int word = (int)(i - w * (uint32_t)len);
  const int nbytes = min(64, sorb - word * 64);  // multiple of 8
  const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(occ + w * (uint64_t)sorb + (uint64_t)word * 64);
  uint64_t bits = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (8 * k < nbytes) {
      const uint64_t t = src[k] ^ 0x0101010101010101ull;                                  // byte == 1  <=>  zero byte
      // 0x80 in exactly the zero bytes (the carry-free form: the shorter (t - 0x01..) & ~t test lets a borrow flag
      // a 0x01 byte above a zero byte)
      const uint64_t lo7 = 0x7f7f7f7f7f7f7f7full;
      const uint64_t z = ~(((t & lo7) + lo7) | t | lo7);
      bits |= (((z >> 7) * 0x0102040810204080ull) >> 56) << (8 * k);                        // byte j -> bit j
    }
  }
  out[i] = bits;
}

// cpu_tensor.cpp:589-688 : binary search of multi-word keys (most significant word last).
template <int LEN>
__global__ __launch_bounds__(kBlock) void lut_search_kernel(const uint64_t *__restrict__ keys, int64_t nkeys,
                                                            const uint64_t *__restrict__ onv, uint64_t n,
                                                            int64_t *__restrict__ idx, uint8_t *__restrict__ mask) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint64_t q[LEN];
#pragma unroll
  for (int w = 0; w < LEN; ++w) q[w] = onv[i * LEN + w];
  idx[i] = lut_find<LEN>(keys, nkeys, q);
  mask[i] = idx[i] >= 0;
}

}  // namespace pynqs

// =================================================================================================
// C ABI
// =================================================================================================
using namespace pynqs;

template <int LEN, typename T>
static int launch_comb_hij(const uint64_t *bra, int64_t nbatch, const SDParams &p, const T *h1e, const T *h2e,
                           uint64_t *comb, T *hmat, hipStream_t st) {
  const uint32_t ncomb = p.nsd + 1;
  uint32_t nchunks, chunk_len;
  plan_chunks(nbatch, ncomb, &nchunks, &chunk_len);
  const bool with_h = hmat != nullptr;
  const size_t lds = lds_bytes(p, with_h ? sizeof(T) : 0);
  const uint64_t grid = (uint64_t)nbatch * nchunks;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "grid too large: nbatch*nchunks > 2^31-1");
  if (comb && with_h)
    hipLaunchKernelGGL((comb_hij_kernel<LEN, T, true, true>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, nchunks,
                       chunk_len, h1e, h2e, comb, hmat);
  else if (comb)
    hipLaunchKernelGGL((comb_hij_kernel<LEN, T, true, false>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, nchunks,
                       chunk_len, h1e, h2e, comb, hmat);
  else
    hipLaunchKernelGGL((comb_hij_kernel<LEN, T, false, true>), dim3((uint32_t)grid), dim3(kBlock), lds, st, bra, p, nchunks,
                       chunk_len, h1e, h2e, comb, hmat);
  return check_launch("comb_hij");
}

extern "C" int pynqs_comb_hij_fused(const uint64_t *bra, int64_t nbatch, int sorb, int nele, int noA, int noB,
                                    const void *h1e, const void *h2e, int dtype, uint64_t *comb, void *hmat, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  if (!make_sd_params(sorb, nele, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (nbatch < 0 || (dtype != PYNQS_F32 && dtype != PYNQS_F64)) return set_error(PYNQS_EINVAL, "bad nbatch/dtype");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !h1e || !h2e || !hmat) return set_error(PYNQS_EINVAL, "null pointer");
  if (nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "nbatch too large");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  DISPATCH_LEN(len, rc = dtype == PYNQS_F64
                          ? launch_comb_hij<LEN, double>(bra, nbatch, p, (const double *)h1e, (const double *)h2e, comb, (double *)hmat, st)
                          : launch_comb_hij<LEN, float>(bra, nbatch, p, (const float *)h1e, (const float *)h2e, comb, (float *)hmat, st));
  return rc;
}

extern "C" int pynqs_comb(const uint64_t *bra, int64_t nbatch, int sorb, int noA, int noB, uint64_t *comb, double *comb_pm1,
                          void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  if (!make_sd_params(sorb, noA + noB, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (nbatch < 0) return set_error(PYNQS_EINVAL, "bad nbatch");
  if (nbatch == 0) return PYNQS_OK;
  if (!bra || !comb) return set_error(PYNQS_EINVAL, "null pointer");
  if (nbatch > 0x7fffffffll) return set_error(PYNQS_EINVAL, "nbatch too large");
  const int len = (sorb - 1) / 64 + 1;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  DISPATCH_LEN(len, rc = launch_comb_hij<LEN, double>(bra, nbatch, p, nullptr, nullptr, comb, nullptr, st));
  if (rc != PYNQS_OK) return rc;
  if (comb_pm1) return pynqs_onv_to_pm1(comb, nbatch * (int64_t)(p.nsd + 1), sorb, PYNQS_F64, comb_pm1, stream);
  return PYNQS_OK;
}

extern "C" int pynqs_hij(const uint64_t *bra, int64_t n, const uint64_t *ket, int64_t m, int ket_is_3d, const void *h1e,
                         const void *h2e, int dtype, int sorb, int nele, void *hmat, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  if (sorb < 1 || sorb > kMaxSorb || n < 0 || m < 0 || nele < 0 || nele > kMaxSorb || (dtype != PYNQS_F32 && dtype != PYNQS_F64))
    return set_error(PYNQS_EINVAL, "bad sorb/n/m/dtype");
  if (n == 0 || m == 0) return PYNQS_OK;
  if (!bra || !ket || !h1e || !h2e || !hmat) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t total = (uint64_t)n * (uint64_t)m;
  const uint64_t grid = (total + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n*m too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64)
      hipLaunchKernelGGL((hij_pairs_kernel<LEN, double>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, (uint64_t)n, ket,
                         (uint64_t)m, ket_is_3d, (const double *)h1e, (const double *)h2e, sorb, nele, (double *)hmat);
    else
      hipLaunchKernelGGL((hij_pairs_kernel<LEN, float>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, (uint64_t)n, ket,
                         (uint64_t)m, ket_is_3d, (const float *)h1e, (const float *)h2e, sorb, nele, (float *)hmat);
  });
  return check_launch("hij_pairs");
}

extern "C" int pynqs_onv_to_pm1(const uint64_t *bra, int64_t n, int sorb, int dtype, void *out, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  if (sorb < 1 || sorb > kMaxSorb || n < 0 || (dtype != PYNQS_F32 && dtype != PYNQS_F64))
    return set_error(PYNQS_EINVAL, "bad sorb/n/dtype");
  if (n == 0) return PYNQS_OK;
  if (!bra || !out) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t total = (uint64_t)n * (uint64_t)sorb;
  const uint64_t per = dtype == PYNQS_F64 ? 2 : 4;  // elements per lane (16-byte stores)
  const uint64_t grid = ((total + per - 1) / per + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n*sorb too large for one launch");
  if ((uintptr_t)out & 15u) return set_error(PYNQS_EINVAL, "out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYNQS_F64)
    hipLaunchKernelGGL((onv_to_pm1_kernel<double>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, (uint64_t)n, sorb, len, (double *)out);
  else
    hipLaunchKernelGGL((onv_to_pm1_kernel<float>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, (uint64_t)n, sorb, len, (float *)out);
  return check_launch("onv_to_pm1");
}

extern "C" int pynqs_pm01_to_onv(const uint8_t *occ, int64_t n, int sorb, uint64_t *out, void *stream) {
  pynqs::DeviceScope device_scope_(occ);
  if (sorb < 1 || sorb > kMaxSorb || n < 0) return set_error(PYNQS_EINVAL, "bad sorb/n");
  if (n == 0) return PYNQS_OK;
  if (!occ || !out) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t waves = (uint64_t)n * len;
  const uint64_t grid = (waves * 64 + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n too large for one launch");
  if (sorb % 8 == 0 && ((uintptr_t)occ & 7u) == 0) {
    const uint64_t g8 = (waves + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(pm01_to_onv_kernel_x8, dim3((uint32_t)g8), dim3(kBlock), 0, (hipStream_t)stream, occ, (uint64_t)n, sorb, len, out);
    return check_launch("pm01_to_onv");
  }
  hipLaunchKernelGGL(pm01_to_onv_kernel, dim3((uint32_t)grid), dim3(kBlock), 0, (hipStream_t)stream, occ, (uint64_t)n, sorb, len, out);
  return check_launch("pm01_to_onv");
}

extern "C" int pynqs_wavefunction_lut(const uint64_t *keys, int64_t nkeys, const uint64_t *onv, int64_t n, int sorb,
                                      int64_t *idx, uint8_t *mask, void *stream) {
  pynqs::DeviceScope device_scope_(keys);
  if (sorb < 1 || sorb > kMaxSorb || n < 0 || nkeys < 0) return set_error(PYNQS_EINVAL, "bad sorb/n/nkeys");
  if (n == 0) return PYNQS_OK;
  if (!onv || !idx || !mask || (nkeys > 0 && !keys)) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t grid = ((uint64_t)n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_LEN(len, hipLaunchKernelGGL((lut_search_kernel<LEN>), dim3((uint32_t)grid), dim3(kBlock), 0, st, keys, nkeys, onv,
                                       (uint64_t)n, idx, mask));
  return check_launch("lut_search");
}

// -------------------------------------------------------------------------------------------------
// spin_flip_rand (cpu_tensor.cpp:90-137, cuda kernel.cu:691-716): one random single/double move per walker.
// r0 is uniform on [0, nsd] (both ends included, like std::uniform_int_distribution<int>(0, ncomb));
// r0 == 0 leaves the walker unchanged, otherwise excitation rank r0 - 1 is applied.  One lane per walker: the
// slot list is rebuilt serially (sorb <= 192 bit scans), the rank is unpacked with the reference's formulas.
// The random stream is a counter-based hash of (seed, call offset, walker index): reproducible for a given
// seed, but -- like the reference's own CPU (mt19937) and CUDA (XORWOW) paths -- a different stream from either.
namespace pynqs {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

template <int LEN>
__global__ __launch_bounds__(kBlock) void spin_flip_rand_kernel(const uint64_t *__restrict__ bra, uint64_t n, SDParams p,
                                                                uint64_t seed, uint64_t offset, uint64_t *__restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= n) return;
  uint64_t x[LEN];
#pragma unroll
  for (int i = 0; i < LEN; ++i) x[i] = bra[w * LEN + i];
  // unbiased integer in [0, nsd]: 64-bit multiply-high of a 64-bit hash (bias < 2^-40 for nsd < 2^24)
  const uint64_t h = mix64(mix64(seed) ^ mix64(offset + w));
  const uint32_t r0 = (uint32_t)__umul64hi(h, (uint64_t)p.nsd + 1);
  if (r0 != 0) excite_by_rank<LEN>(x, r0 - 1, p);
#pragma unroll
  for (int i = 0; i < LEN; ++i) out[w * LEN + i] = x[i];
}

}  // namespace pynqs

extern "C" int pynqs_spin_flip_rand(const uint64_t *bra, int64_t n, int sorb, int noA, int noB, uint64_t seed, uint64_t offset,
                                    uint64_t *out, void *stream) {
  pynqs::DeviceScope device_scope_(bra);
  SDParams p;
  if (!make_sd_params(sorb, noA + noB, noA, noB, &p)) return set_error(PYNQS_EINVAL, "bad sorb/noA/noB");
  if (n < 0) return set_error(PYNQS_EINVAL, "bad n");
  if (n == 0) return PYNQS_OK;
  if (!bra || !out) return set_error(PYNQS_EINVAL, "null pointer");
  const int len = (sorb - 1) / 64 + 1;
  const uint64_t grid = ((uint64_t)n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffull) return set_error(PYNQS_EINVAL, "n too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_LEN(len, hipLaunchKernelGGL((spin_flip_rand_kernel<LEN>), dim3((uint32_t)grid), dim3(kBlock), 0, st, bra, (uint64_t)n, p,
                                       seed, offset, out));
  return check_launch("spin_flip_rand");
}
