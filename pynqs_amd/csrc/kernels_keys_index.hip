// kernels_keys_index.hip -- block index of a table of determinants, for the INDEXED key-major SAMPLE_SPACE local energy
// (kernels_eloc_keys.hip; the table is the reference's WavefunctionLUT key set, utils/public_function.py:1110-1215).
//
// The sorb bits are cut into kIndexBlocks = 5 blocks with even boundaries (detcore.h: index_block_lo).  For every block b the index holds
// the keys' block values (tagged with b in bits 40-42) in ascending order, svals[b][nkeys] (uint64), and the number of the key each came from, perm[b][nkeys] (uint32;
// equal values keep the order of the key array: the sort is stable, so the index -- and the order of additions in the kernel that reads
// it -- is a function of the key array alone).  60 bytes per key.  A determinant within a double excitation of x agrees with x in at
// least one block, so its number stands in one of the five runs svals[b][..] == block_b(x).
// Built once per table (a VMC iteration's sample space) by ONE radix sort of the 5 nkeys (tag, value) pairs: rocPRIM's device sort is
// set-up, as torch.sort is in the reference's table construction; the look-ups and everything per walker are hand-written.
#include <cstring>

#include "detcore.h"
#include "launch.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace pynqs {

// vals[b][k] = (b << 40) | block b of key k, number[b][k] = k: ONE sort of the 5 nkeys pairs then leaves the blocks one after the other,
// each in ascending order of its values (the tag stays in the sorted values: a look-up adds it to the value it searches for)
template <int LEN>
__global__ __launch_bounds__(kBlock) void index_block_values_kernel(const uint64_t *__restrict__ keys, int64_t nkeys, int sorb,
                                                                    uint64_t *__restrict__ vals, uint32_t *__restrict__ number) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nkeys) return;
  uint64_t x[LEN];
#pragma unroll
  for (int i = 0; i < LEN; ++i) x[i] = keys[k * LEN + i];
#pragma unroll
  for (int b = 0; b < kIndexBlocks; ++b) {
    vals[(size_t)b * nkeys + k] = ((uint64_t)b << kIndexTagShift) | index_block_value<LEN>(x, index_block_lo(sorb, b), index_block_lo(sorb, b + 1));
    number[(size_t)b * nkeys + k] = (uint32_t)k;
  }
}

// sum over the runs of equal values of (run length)^2, all blocks together: a key of the table used as a walker finds that many keys
// through the index, on average sum / nkeys -- against nkeys for the streamed form.  The first element of a run measures it: a few steps
// forward (tables of samples: runs of one or two), a binary search for the end of a long run (CAS-like tables: thousands).
__global__ __launch_bounds__(kBlock) void index_density_kernel(const uint64_t *__restrict__ svals, int64_t total, unsigned long long *__restrict__ sum) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long sq = 0;
  if (i < total) {
    const uint64_t v = svals[i];
    if (i == 0 || svals[i - 1] != v) {
      int64_t end = i + 1;
      while (end < total && end < i + 8 && svals[end] == v) ++end;
      if (end == i + 8 && end < total && svals[end] == v) {  // first position in (end, total] with svals[pos] > v
        int64_t first = end + 1, n = total - first;
        while (n > 0) {
          const int64_t half = n >> 1;
          const bool right = svals[first + half] <= v;
          first = right ? first + half + 1 : first;
          n = right ? n - half - 1 : half;
        }
        end = first;
      }
      const unsigned long long len = (unsigned long long)(end - i);
      sq = len * len;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
  if ((threadIdx.x & 63) == 0 && sq) atomicAdd(sum, sq);
}

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t nkeys) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (size_t)kIndexBlocks * (size_t)nkeys, 0u, (unsigned)kIndexTagShift + 3u, (hipStream_t)0);
  return bytes;
}

}  // namespace pynqs

using namespace pynqs;

extern "C" int64_t pynqs_keys_index_bytes(int64_t nkeys, int sorb) {
  if (nkeys < 0 || nkeys >= (1ll << 27) || sorb < 2 || sorb > kMaxSorb || (sorb & 1)) return -1;
  return (int64_t)((size_t)kIndexBlocks * (size_t)nkeys * 12);
}

extern "C" int64_t pynqs_keys_index_workspace(int64_t nkeys, int sorb) {
  if (nkeys < 0 || nkeys >= (1ll << 27) || sorb < 2 || sorb > kMaxSorb || (sorb & 1)) return -1;
  if (nkeys == 0) return 0;
  return (int64_t)(align256((size_t)kIndexBlocks * nkeys * 8) + align256((size_t)kIndexBlocks * nkeys * 4) + align256(sort_temp_bytes(nkeys)));
}

extern "C" int pynqs_keys_index_build(const uint64_t *keys, int64_t nkeys, int sorb, void *index, void *workspace, void *stream) {
  pynqs::DeviceScope device_scope_(keys);
  if (nkeys < 0 || nkeys >= (1ll << 27) || sorb < 2 || sorb > kMaxSorb || (sorb & 1)) return set_error(PYNQS_EINVAL, "bad nkeys / sorb (even, nkeys < 2^27)");
  if (nkeys == 0) return PYNQS_OK;
  if (!keys || !index || !workspace) return set_error(PYNQS_EINVAL, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int len = (sorb - 1) / 64 + 1;
  uint64_t *vals = (uint64_t *)workspace;
  uint32_t *number = (uint32_t *)((char *)workspace + align256((size_t)kIndexBlocks * nkeys * 8));
  void *temp = (char *)number + align256((size_t)kIndexBlocks * nkeys * 4);
  size_t temp_bytes = sort_temp_bytes(nkeys);
  uint64_t *svals = (uint64_t *)index;
  uint32_t *perm = (uint32_t *)(svals + (size_t)kIndexBlocks * (size_t)nkeys);
  const uint32_t grid = (uint32_t)((nkeys + kBlock - 1) / kBlock);
  DISPATCH_LEN(len, hipLaunchKernelGGL((index_block_values_kernel<LEN>), dim3(grid), dim3(kBlock), 0, st, keys, nkeys, sorb, vals, number));
  if (rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t *)vals, svals, (const uint32_t *)number, perm,
                                (size_t)kIndexBlocks * (size_t)nkeys, 0u, (unsigned)kIndexTagShift + 3u, st) != hipSuccess)
    return set_error(PYNQS_ELAUNCH, "keys_index sort");
  return check_launch("keys_index_build");
}

extern "C" int pynqs_keys_index_density(const void *index, int64_t nkeys, int sorb, uint64_t *sum_sq, void *stream) {
  pynqs::DeviceScope device_scope_(index);
  if (nkeys < 0 || nkeys >= (1ll << 27) || sorb < 2 || sorb > kMaxSorb || (sorb & 1)) return set_error(PYNQS_EINVAL, "bad nkeys / sorb (even, nkeys < 2^27)");
  if (!sum_sq || (nkeys > 0 && !index)) return set_error(PYNQS_EINVAL, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sum_sq, 0, 8, st) != hipSuccess) return check_launch("keys_index_density memset");
  if (nkeys == 0) return PYNQS_OK;
  const int64_t total = (int64_t)kIndexBlocks * nkeys;
  hipLaunchKernelGGL(index_density_kernel, dim3((uint32_t)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (const uint64_t *)index, total,
                     (unsigned long long *)sum_sq);
  return check_launch("keys_index_density");
}
