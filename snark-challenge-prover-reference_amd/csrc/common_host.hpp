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
// Every entry point that takes an object living on one device (a base set, an evaluation domain) runs on that device and puts the
// thread back on its own afterwards.  HIP's own notion of the thread's current device, not the library's bookkeeping: a host thread
// that never called mnt753_set_device, or whose device PyTorch changed, still gets its kernels, events and allocations on the object's GPU.
struct OnDevice {
  int back = -1;
  explicit OnDevice(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    if (cur != dev) { (void)hipSetDevice(dev); back = cur; }
  }
  ~OnDevice() { if (back >= 0) (void)hipSetDevice(back); }
  OnDevice(const OnDevice&) = delete;
  OnDevice& operator=(const OnDevice&) = delete;
};
}  // namespace mnt753

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) return ::mnt753::set_hip_error(_e, #expr, __FILE__, __LINE__); \
  } while (0)
