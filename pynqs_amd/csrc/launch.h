// launch.h -- host-side helpers shared by the C-ABI translation units of libpynqs_amd.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/pynqs_amd.h"

namespace pynqs {

char *error_buffer();  // thread-local, defined in capi_common.hip

inline int set_error(int code, const char *msg) {
  snprintf(error_buffer(), 256, "%s", msg);
  return code;
}

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(error_buffer(), 256, "%s: %s", what, hipGetErrorString(e));
    return PYNQS_ELAUNCH;
  }
  return PYNQS_OK;
}

// Every launching entry point runs on the device that owns its first device pointer, whatever the caller's current
// device is (a caller holding tensors on cuda:k without torch.cuda.set_device(k) would otherwise launch, memset and set
// function attributes on the wrong GPU); the previous device is restored on return.  ~1 us per call.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(const void *p) {
    hipPointerAttribute_t a;
    if (p != nullptr && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeDevice) {
      if (hipGetDevice(&prev) == hipSuccess && prev != a.device && hipSetDevice(a.device) == hipSuccess) switched = true;
    } else {
      (void)hipGetLastError();  // not a device pointer (the argument checks of the entry point report it)
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};

// How a walker's row of ncomb columns is split over workgroups.  One workgroup per walker when there
// are enough walkers to fill the chip (256 CUs x 8 resident workgroups); otherwise rows are cut into
// chunks (multiples of 256 columns, never shorter than 2048 so that the per-workgroup table build
// stays amortised).
inline void plan_chunks(int64_t nbatch, uint32_t ncomb, uint32_t *nchunks, uint32_t *chunk_len) {
  static const int64_t want = getenv("PYNQS_WANT_WG") ? atoll(getenv("PYNQS_WANT_WG")) : 4096;  // workgroups in flight target
  static const int64_t minlen = getenv("PYNQS_MIN_CHUNK") ? atoll(getenv("PYNQS_MIN_CHUNK")) : 2048;
  int64_t c = nbatch >= want ? 1 : (want + nbatch - 1) / nbatch;
  int64_t maxc = (ncomb + minlen - 1) / minlen;
  if (c > maxc) c = maxc;
  if (c < 1) c = 1;
  uint32_t len = (uint32_t)((ncomb + c - 1) / c);
  len = (len + 255u) & ~255u;
  *chunk_len = len;
  *nchunks = (ncomb + len - 1) / len;
}

// whether map_workgroup's XCD-aware order applies (plan_tiles.h): at least 8 chunks per walker, a multiple of 8.
// OFF unless PYNQS_XCD_MAP=1: measured neutral to 8 % slower (sorb 120: 64 walkers 0.575 -> 0.620 ms, 512 walkers
// 4.376 -> 4.366 ms; sorb 184: 1.075 -> 1.072 ms) -- these kernels sit at the HBM ceiling for mixed traffic
// (4.5-5.8 TB/s moved; a device copy reaches 4.55), not on L2 misses of the plan.
inline bool xcd_mapping(uint32_t nchunks) {
  static const bool on = getenv("PYNQS_XCD_MAP") && atoi(getenv("PYNQS_XCD_MAP")) == 1;
  return on && nchunks >= 8 && nchunks % 8 == 0;
}

}  // namespace pynqs

// run-time dispatch of the ONV word count (the reference compiles one build per MAX_SORB_LEN)
#define DISPATCH_LEN(len, ...)                                  \
  switch (len) {                                                \
    case 1: { constexpr int LEN = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int LEN = 2; __VA_ARGS__; } break;      \
    default: { constexpr int LEN = 3; __VA_ARGS__; } break;     \
  }
