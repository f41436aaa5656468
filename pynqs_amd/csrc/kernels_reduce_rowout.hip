// kernels_reduce_rowout.hip -- the semi-stochastic REDUCE front end for rows of up to 8192 columns (round 4; Fe2S2: 7876), one launch:
// the LIST kernel of reduce_list.h with ROWOUT (enumeration, kept records, the row's sub-eps matrix elements as float32 through global
// memory) followed, in the same workgroup, by the draws of reduce_draw.h.  In a translation unit of its own: kernels_reduce_onepass.hip
// instantiates forty kernels and takes two minutes to compile.
#include "reduce_list.h"

namespace pynqs {

template <int LEN, typename T>
// (at least six waves per SIMD, 80 VGPRs: the four records a thread resolves side by side want registers; measured 8 / 7 / 6 / 5 waves:
// 610 / 583 / 572 / 588 us per 8192 Fe2S2 walkers while 23 KB of LDS allowed six workgroups per CU whatever the registers; with 19.9 KB
// (eight fit) 8 / 7 / 6: 667 / 613 / 598 us in one run -- more workgroups in flight are SLOWER: 7 x 32 rows of 31.5 KB per XCD fall out of
// its 4 MiB L2 before the draws read them back)
#ifndef PYNQS_ROWOUT_WAVES
#define PYNQS_ROWOUT_WAVES 6
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PYNQS_ROWOUT_WAVES, 8)))
void reduce_onepass_list_rowout_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks, uint32_t chunk_len, uint32_t max_tiles,
                                       const T *__restrict__ plan, T eps, uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  reduce_onepass_list_body<LEN, T, true, false, false, false, true>(bra, p, pl, nchunks, chunk_len, max_tiles, plan, eps, nsample, seed, P, o);
}

// The flushing semi-stochastic form (rows of any length; the kept list is emptied whenever it is nearly full) with the row's float32 copy:
// the draws inside the drawn tiles read their columns back instead of enumerating the tile again (reduce_list.h, ROW32).
template <int LEN, typename T, bool GTILE>
__global__ __launch_bounds__(kBlock) void reduce_onepass_list_flush_row32_kernel(const uint64_t *__restrict__ bra, SDParams p, PlanLayout pl, uint32_t nchunks,
                                                                                uint32_t chunk_len, uint32_t max_tiles, const T *__restrict__ plan, T eps,
                                                                                uint32_t nsample, uint64_t seed, uint32_t P, OnepassOut<T> o) {
  reduce_onepass_list_body<LEN, T, true, false, true, GTILE, false, true>(bra, p, pl, nchunks, chunk_len, max_tiles, plan, eps, nsample, seed, P, o);
}

int launch_reduce_flush_row32(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, uint32_t chunk_len, uint32_t max_tiles,
                              const void *plan, int dtype, double eps_eff, int eps_sample, uint64_t seed, uint32_t P, size_t lds,
                              const pynqs_reduce_io *io, uint32_t fixed, bool gtile, hipStream_t st) {
  if (!io->row_f32) return set_error(PYNQS_EINVAL, "io->row_f32 missing");
  const int len = (p.sorb - 1) / 64 + 1;
#define PYNQS_R32_LAUNCH(TT)                                                                                                             \
  do {                                                                                                                                   \
    auto kfn = gtile ? reduce_onepass_list_flush_row32_kernel<LEN, TT, true> : reduce_onepass_list_flush_row32_kernel<LEN, TT, false>;   \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return check_launch("hipFuncSetAttribute");                                                                                       \
    hipLaunchKernelGGL(kfn, dim3((uint32_t)nbatch), dim3(kBlock), lds, st, bra, p, pl, 1u, chunk_len, max_tiles, (const TT *)plan, (TT)eps_eff, \
                       (uint32_t)eps_sample, seed, P, make_out<TT>(io, len, fixed, gtile ? max_tiles : 0u));                            \
  } while (0)
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64) PYNQS_R32_LAUNCH(double); else PYNQS_R32_LAUNCH(float);
  });
#undef PYNQS_R32_LAUNCH
  return check_launch("reduce_onepass (semi-stochastic, flushing, float32 row copy)");
}

bool reduce_draw_supported(const SDParams &p, int eps_sample) {
  return eps_sample > 0 && (uint32_t)eps_sample <= kDrawMaxDraws && p.nsd + 1 <= kDrawMaxCols;
}

int launch_reduce_rowout(const uint64_t *bra, int64_t nbatch, const SDParams &p, const PlanLayout &pl, uint32_t chunk_len, uint32_t max_tiles,
                         const void *plan, int dtype, double eps_eff, int eps_sample, uint64_t seed, uint32_t P, size_t lds,
                         const pynqs_reduce_io *io, uint32_t fixed, hipStream_t st) {
  if (!io->row_f32) return set_error(PYNQS_EINVAL, "io->row_f32 missing");
  const int len = (p.sorb - 1) / 64 + 1;
#define PYNQS_RO_LAUNCH(TT)                                                                                                              \
  do {                                                                                                                                   \
    auto kfn = reduce_onepass_list_rowout_kernel<LEN, TT>;                                                                               \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return check_launch("hipFuncSetAttribute");                                                                                       \
    hipLaunchKernelGGL(kfn, dim3((uint32_t)nbatch), dim3(kBlock), lds, st, bra, p, pl, 1u, chunk_len, max_tiles, (const TT *)plan, (TT)eps_eff, \
                       (uint32_t)eps_sample, seed, P, make_out<TT>(io, len, fixed, 0u));                                                 \
  } while (0)
  DISPATCH_LEN(len, {
    if (dtype == PYNQS_F64) PYNQS_RO_LAUNCH(double); else PYNQS_RO_LAUNCH(float);
  });
#undef PYNQS_RO_LAUNCH
  return check_launch("reduce_onepass (semi-stochastic, sorted draws)");
}

}  // namespace pynqs

#ifdef PYNQS_OP_STAMPS
extern "C" int pynqs_debug_stamps(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pynqs::g_stamps), sizeof(unsigned long long) * 8192 * 16);
}
#endif
