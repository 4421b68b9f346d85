// Lifecycle, error reporting, device-memory helpers and the O(1) host-side group operations of the
// C ABI (include/mnt753_hip.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "common_host.hpp"
#include "host_field.hpp"

namespace mnt753 {
namespace {
thread_local std::string t_last_error;
bool g_ready = false;
int g_device = -1;
// persistent staging for mnt753_load_file_to_device: creating a stream costs ~8 ms and pinning 32 MB a few more, so
// they are made once in mnt753_init (outside any timed region) and reused under a mutex
struct IoStaging {
  static constexpr size_t CHUNK = (size_t)16 << 20;
  std::mutex mu;
  hipStream_t stream = nullptr;
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool ok = false;
} g_io;
}  // namespace

int set_error(int code, const char* msg) {
  t_last_error = msg ? msg : "";
  return code;
}
int set_hip_error(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  t_last_error = buf;
  return e == hipErrorOutOfMemory ? MNT753_ENOMEM : MNT753_EHIP;
}
int require_device() {
  if (!g_ready) return set_error(MNT753_ENODEV, "no HIP device: call mnt753_init() on a machine with an MI355X (there is no CPU fallback)");
  return 0;
}
}  // namespace mnt753

using namespace mnt753;
using namespace mnt753::host;

namespace {
template <class HC>
int point_add_t(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  HPoint<HC>::from_wire(a).add(HPoint<HC>::from_wire(b)).to_wire(out);
  return 0;
}
template <class HC>
int point_scale_t(const uint64_t* scalar, const uint64_t* p, uint64_t* out) {
  uint64_t e[12];
  HFp<HC::FR>::from_words(scalar).to_integer(e);
  HPoint<HC>::from_wire(p).mul_words(e, 12).to_wire(out);
  return 0;
}
template <class HC>
int point_to_affine_t(const uint64_t* p, uint64_t* out) {
  typename HC::F x, y;
  HPoint<HC>::from_wire(p).to_affine(x, y);
  for (int k = 0; k < HC::F::DEG; ++k) {
    memcpy(out + 12 * k, x.comp(k).l, 96);
    memcpy(out + 12 * (HC::F::DEG + k), y.comp(k).l, 96);
  }
  return 0;
}
template <class HC>
int point_from_affine_t(const uint64_t* aff, uint64_t* out) {
  typedef typename HC::F F;
  HPoint<HC> p;
  for (int k = 0; k < F::DEG; ++k) {
    p.X.comp(k) = F::B::from_words(aff + 12 * k);
    p.Y.comp(k) = F::B::from_words(aff + 12 * (F::DEG + k));
  }
  if (p.Y.is_zero()) p = HPoint<HC>::zero();  // serialization.hpp:87-89 / :107-109
  else p.Z = F::one();
  p.to_wire(out);
  return 0;
}
bool bad_cg(int curve, int group) { return curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2); }
}  // namespace

#define DISPATCH_CG(fn, ...)                                                                                   \
  (curve == MNT753_CURVE_MNT4753 ? (group == MNT753_G1 ? fn<HMnt4G1>(__VA_ARGS__) : fn<HMnt4G2>(__VA_ARGS__)) \
                                 : (group == MNT753_G1 ? fn<HMnt6G1>(__VA_ARGS__) : fn<HMnt6G2>(__VA_ARGS__)))

extern "C" {

int mnt753_init(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_ready = false;
    return set_error(MNT753_ENODEV, "mnt753_init: no HIP device visible");
  }
  if (device < 0 || device >= count) return set_error(MNT753_EINVAL, "mnt753_init: device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(nullptr));
  // Keep scratch resident.  The 512-register point-arithmetic kernels spill a few hundred bytes (G1 reduction kernels)
  // to a few KB (Fq2/Fq3) per lane; ROCr sizes a dispatch's scratch for every wave slot of the device, and a dispatch
  // above the queue's scratch threshold falls back to "use-once" scratch -- a host round trip per launch (~90 us
  // measured here) that also keeps independent streams from overlapping.  Raise the threshold to the device maximum.
  {
    size_t smax = 0, scur = 0;
    if (hipDeviceGetLimit(&smax, hipExtLimitScratchMax) == hipSuccess && hipDeviceGetLimit(&scur, hipExtLimitScratchCurrent) == hipSuccess &&
        smax > scur) {
      size_t want = smax;
      if (const char* e = getenv("MNT753_SCRATCH_LIMIT_MB")) want = (size_t)atoll(e) << 20;
      if (want > smax) want = smax;
      if (want > scur) (void)hipDeviceSetLimit(hipExtLimitScratchCurrent, want);
    }
    if (getenv("MNT753_VERBOSE")) {
      size_t now = 0; (void)hipDeviceGetLimit(&now, hipExtLimitScratchCurrent);
      fprintf(stderr, "mnt753: scratch limit max %zu MB, was %zu MB, now %zu MB\n", smax >> 20, scur >> 20, now >> 20);
    }
    (void)hipGetLastError();
  }
  g_device = device;
  if (!g_io.ok) {
    bool ok = hipStreamCreateWithFlags(&g_io.stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
      ok = hipHostMalloc(&g_io.buf[k], IoStaging::CHUNK) == hipSuccess && hipEventCreateWithFlags(&g_io.done[k], hipEventDisableTiming) == hipSuccess;
    g_io.ok = ok;
  }
  g_ready = true;
  return 0;
}

const char* mnt753_last_error(void) { return t_last_error.c_str(); }

size_t mnt753_affine_words(int curve, int group) {
  if (bad_cg(curve, group)) return 0;
  int deg = group == MNT753_G1 ? 1 : (curve == MNT753_CURVE_MNT4753 ? 2 : 3);
  return (size_t)24 * deg;
}
size_t mnt753_projective_words(int curve, int group) {
  if (bad_cg(curve, group)) return 0;
  int deg = group == MNT753_G1 ? 1 : (curve == MNT753_CURVE_MNT4753 ? 2 : 3);
  return (size_t)36 * deg;
}

int mnt753_dev_alloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr) return set_error(MNT753_EINVAL, "dev_alloc: null");
  if (int rc = require_device()) return rc;
  HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 16));
  return 0;
}
int mnt753_dev_free(void* dev_ptr) {
  if (!dev_ptr) return 0;
  if (int rc = require_device()) return rc;
  HIP_TRY(hipFree(dev_ptr));
  return 0;
}
int mnt753_copy_h2d(void* dev_dst, const void* src, size_t bytes) {
  if (int rc = require_device()) return rc;
  if (bytes && (!dev_dst || !src)) return set_error(MNT753_EINVAL, "copy_h2d: null");
  HIP_TRY(hipMemcpy(dev_dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int mnt753_copy_d2h(void* dst, const void* dev_src, size_t bytes) {
  if (int rc = require_device()) return rc;
  if (bytes && (!dst || !dev_src)) return set_error(MNT753_EINVAL, "copy_d2h: null");
  HIP_TRY(hipMemcpy(dst, dev_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int mnt753_copy_d2d(void* dev_dst, const void* dev_src, size_t bytes) {
  if (int rc = require_device()) return rc;
  if (bytes && (!dev_dst || !dev_src)) return set_error(MNT753_EINVAL, "copy_d2d: null");
  HIP_TRY(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, nullptr));
  return 0;
}
int mnt753_dev_memset(void* dev_dst, int value, size_t bytes) {
  if (int rc = require_device()) return rc;
  if (bytes && !dev_dst) return set_error(MNT753_EINVAL, "dev_memset: null");
  HIP_TRY(hipMemsetAsync(dev_dst, value, bytes, nullptr));
  return 0;
}
int mnt753_load_file_to_device(const char* path, size_t file_offset, size_t bytes, void* dev_dst) {
  if (int rc = require_device()) return rc;
  if (!path || (bytes && !dev_dst)) return set_error(MNT753_EINVAL, "load_file_to_device: null argument");
  HIP_TRY(hipSetDevice(g_device));   // may be called from a thread that has not touched the device yet
  FILE* f = fopen(path, "rb");
  if (!f) return set_error(MNT753_EINVAL, "load_file_to_device: cannot open file");
  if (fseeko(f, (off_t)file_offset, SEEK_SET) != 0) { fclose(f); return set_error(MNT753_EINVAL, "load_file_to_device: seek failed"); }
  std::lock_guard<std::mutex> lock(g_io.mu);
  if (!g_io.ok) { fclose(f); return set_error(MNT753_EHIP, "load_file_to_device: staging buffers were not created by mnt753_init"); }
  constexpr size_t CHUNK = IoStaging::CHUNK;
  int rc = 0;
  size_t off = 0;
  for (size_t i = 0; off < bytes; ++i) {
    const int k = (int)(i & 1);
    const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
    if (i >= 2 && hipEventSynchronize(g_io.done[k]) != hipSuccess) { rc = set_error(MNT753_EHIP, "load_file_to_device: event"); break; }
    if (fread(g_io.buf[k], 1, n, f) != n) { rc = set_error(MNT753_EINVAL, "load_file_to_device: short read"); break; }
    if (hipMemcpyAsync((char*)dev_dst + off, g_io.buf[k], n, hipMemcpyHostToDevice, g_io.stream) != hipSuccess ||
        hipEventRecord(g_io.done[k], g_io.stream) != hipSuccess) { rc = set_error(MNT753_EHIP, "load_file_to_device: copy"); break; }
    off += n;
  }
  if (hipStreamSynchronize(g_io.stream) != hipSuccess && rc == 0) rc = set_error(MNT753_EHIP, "load_file_to_device: sync");
  fclose(f);
  return rc;
}
int mnt753_sync(void* stream) {
  if (int rc = require_device()) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int mnt753_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  if (bad_cg(curve, group) || !a || !b || !out) return set_error(MNT753_EINVAL, "point_add: bad argument");
  return DISPATCH_CG(point_add_t, a, b, out);
}
int mnt753_point_scale(int curve, int group, const uint64_t* scalar, const uint64_t* p, uint64_t* out) {
  if (bad_cg(curve, group) || !scalar || !p || !out) return set_error(MNT753_EINVAL, "point_scale: bad argument");
  return DISPATCH_CG(point_scale_t, scalar, p, out);
}
int mnt753_point_to_affine(int curve, int group, const uint64_t* p, uint64_t* out) {
  if (bad_cg(curve, group) || !p || !out) return set_error(MNT753_EINVAL, "point_to_affine: bad argument");
  return DISPATCH_CG(point_to_affine_t, p, out);
}
int mnt753_point_from_affine(int curve, int group, const uint64_t* aff, uint64_t* out) {
  if (bad_cg(curve, group) || !aff || !out) return set_error(MNT753_EINVAL, "point_from_affine: bad argument");
  return DISPATCH_CG(point_from_affine_t, aff, out);
}

}  // extern "C"
