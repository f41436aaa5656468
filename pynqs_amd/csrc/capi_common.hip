// capi_common.hip -- host-only entry points of the C ABI (no kernels).
#include "detcore.h"
#include "launch.h"

namespace pynqs {
char *error_buffer() {
  static thread_local char buf[256] = {0};
  return buf;
}
}  // namespace pynqs

extern "C" int pynqs_abi_version(void) { return PYNQS_ABI_VERSION; }

extern "C" const char *pynqs_last_error(void) { return pynqs::error_buffer(); }

// cpp_src/cpu/excitation.cpp:8-16
extern "C" int64_t pynqs_num_sd(int sorb, int noA, int noB) {
  const int64_t k = sorb / 2;
  const int64_t nvA = k - noA, nvB = k - noB;
  return noA * nvA + noB * nvB + noA * (int64_t)(noA - 1) * nvA * (nvA - 1) / 4 +
         noB * (int64_t)(noB - 1) * nvB * (nvB - 1) / 4 + (int64_t)noA * noB * nvA * nvB;
}

// cpp_src/tensor/bind.cpp:282-301.  The reference compares the word count with the compile-time
// MAX_SORB_LEN (one build per length); this library dispatches the length at run time, so only
// lengths beyond PYNQS_MAX_SORB_LEN are a length error.
extern "C" int pynqs_check_sorb(int sorb, int nele) {
  if (sorb < 1 || (sorb - 1) / 64 + 1 > PYNQS_MAX_SORB_LEN) return pynqs::set_error(PYNQS_ELENGTH, "Sorb error");
  if (nele > PYNQS_MAX_NELE) return pynqs::set_error(PYNQS_EOVERFLOW, "electron overflow");
  if (sorb - nele > PYNQS_MAX_NVIR) return pynqs::set_error(PYNQS_EOVERFLOW, "unoccupied orbital error");
  return PYNQS_OK;
}
