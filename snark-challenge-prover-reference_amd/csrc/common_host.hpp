// Shared host-side plumbing of libmnt753_hip.so: error reporting and the device guard.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mnt753_hip.h"

namespace mnt753 {
// records msg for mnt753_last_error() and returns code
int set_error(int code, const char* msg);
int set_hip_error(hipError_t e, const char* what, const char* file, int line);
// 0 if mnt753_init() succeeded on a HIP device, else MNT753_ENODEV (there is no CPU fallback)
int require_device();
// physical HIP ordinal of the calling thread's current logical device (mnt753_set_device), -1 before initialisation
int current_physical_device();
}  // namespace mnt753

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) return ::mnt753::set_hip_error(_e, #expr, __FILE__, __LINE__); \
  } while (0)
