// Lifecycle, error reporting, device-memory helpers and the O(1) host-side group operations of the
// C ABI (include/mnt753_hip.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>

#include "common_host.hpp"
#include "host_field.hpp"

namespace mnt753 {
namespace {
thread_local std::string t_last_error;
// persistent staging for mnt753_load_file_to_device: creating a stream costs ~8 ms and pinning memory a few more, so they are made
// once per device at initialisation (outside any timed region) and reused under a mutex.  IO_LANES independent lanes (a stream, two
// pinned 8 MB buffers and their events each): a large read is cut into IO_LANES contiguous parts that are read (pread) and copied
// concurrently (opt-in, see mnt753_load_file_to_device).
constexpr int IO_LANES = 2;
struct IoLane {
  hipStream_t stream = nullptr;
  void* buf[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
};
struct IoStaging {
  static constexpr size_t CHUNK = (size_t)8 << 20;
  std::mutex mu;
  IoLane lane[IO_LANES];
  bool ok = false;
};
// Logical devices 0 .. g_ndev-1 of this process.  mnt753_init(d) makes physical device d logical device 0 (the single-GPU
// contract of the reference wrapper); mnt753_init_devices(n) maps logical i to physical i -- or, with MNT753_SHARE_DEVICE=1
// (development: exercising the sharded path on a one-GPU box), to physical i % (visible devices).
constexpr int MAX_DEVICES = 16;
struct DevState {
  int phys = -1; bool ready = false; IoStaging io;
  hipEvent_t xfer_ev[MAX_DEVICES] = {};   // [dst]: "everything enqueued on this device's default stream so far" for a copy to dst
};
DevState g_devs[MAX_DEVICES];
int g_peer[MAX_DEVICES][MAX_DEVICES];   // [device][peer]: 0 = not asked yet, else 1 + MNT753_PEER_*
int g_ndev = 0;
thread_local int t_cur_dev = 0;   // logical device the calling thread works on (mnt753_set_device)

int init_one(int logical, int phys) {
  HIP_TRY(hipSetDevice(phys));
  // The host thread SPINS while it waits for an MSM (hipEventSynchronize): a blocking wait costs 0.1-0.25 ms of wake-up latency per
  // MSM on the shared hosts of the GPU boxes (profiles/r06/sync_policy.txt: 22.32-22.53 ms per step spinning, 22.49-22.68 blocking),
  // and which of the two the runtime picks by itself (hipDeviceScheduleAuto) depends on the box.  A prover has a core to spare for it;
  // MNT753_SYNC_SPIN=0 leaves the choice to the runtime.  (Ignored without harm where the device is already active.)
  {
    const char* e = getenv("MNT753_SYNC_SPIN");
    if (!e || atoi(e) != 0) { (void)hipSetDeviceFlags(hipDeviceScheduleSpin); (void)hipGetLastError(); }
  }
  HIP_TRY(hipFree(nullptr));
  // Keep scratch resident.  The 512-register point-arithmetic kernels spill a few hundred bytes (G1 reduction kernels)
  // to a few KB (Fq2/Fq3) per lane; ROCr sizes a dispatch's scratch for every wave slot of the device, and a dispatch
  // above the queue's scratch threshold falls back to "use-once" scratch -- a host round trip per launch (~90 us
  // measured here) that also keeps independent streams from overlapping.  Raise the threshold to the device maximum.
  {
    size_t smax = 0, scur = 0;
    if (hipDeviceGetLimit(&smax, hipExtLimitScratchMax) == hipSuccess && hipDeviceGetLimit(&scur, hipExtLimitScratchCurrent) == hipSuccess &&
        smax > scur) {
      (void)hipDeviceSetLimit(hipExtLimitScratchCurrent, smax);
    }
    if (const char* e = getenv("MNT753_TRACE"); e && atoi(e) > 1) {   // MNT753_TRACE=2: initialisation details as well
      size_t now = 0; (void)hipDeviceGetLimit(&now, hipExtLimitScratchCurrent);
      fprintf(stderr, "mnt753: device %d (physical %d): scratch limit max %zu MB, was %zu MB, now %zu MB\n", logical, phys, smax >> 20, scur >> 20, now >> 20);
    }
    (void)hipGetLastError();
  }
  DevState& d = g_devs[logical];
  d.phys = phys;
  if (!d.io.ok) {
    bool ok = true;
    for (int l = 0; l < IO_LANES && ok; ++l) {
      ok = hipStreamCreateWithFlags(&d.io.lane[l].stream, hipStreamNonBlocking) == hipSuccess;
      for (int k = 0; k < 2 && ok; ++k) {
        ok = hipHostMalloc(&d.io.lane[l].buf[k], IoStaging::CHUNK) == hipSuccess && hipEventCreateWithFlags(&d.io.lane[l].done[k], hipEventDisableTiming) == hipSuccess;
        // touch every page now: the first proof of a process otherwise pays the page faults of 64 MB of pinned memory inside its
        // timed window (measured: first input load 24-32 ms, later ones 9-13 ms)
        if (ok) memset(d.io.lane[l].buf[k], 0, IoStaging::CHUNK);
      }
    }
    // first use of a stream creates its hardware queue, and the first LARGE copy of a stream whatever a copy of that size needs
    // (a 4 KB copy did not warm it: with several lanes the first input load of a process took 25-45 ms against 10-15 later): both
    // buffers of every lane go over once now, from threads of their own as in a load, not inside the first proof
    if (ok) {
      void* scratch = nullptr;
      if (hipMalloc(&scratch, IoStaging::CHUNK) == hipSuccess) {
        std::thread warm[IO_LANES];
        for (int l = 0; l < IO_LANES; ++l)
          warm[l] = std::thread([&d, l, phys, scratch]() {
            if (hipSetDevice(phys) != hipSuccess) return;
            for (int k = 0; k < 2; ++k) {
              (void)hipMemcpyAsync(scratch, d.io.lane[l].buf[k], IoStaging::CHUNK, hipMemcpyHostToDevice, d.io.lane[l].stream);
              (void)hipEventRecord(d.io.lane[l].done[k], d.io.lane[l].stream);
            }
            (void)hipEventSynchronize(d.io.lane[l].done[1]);
            (void)hipStreamSynchronize(d.io.lane[l].stream);
          });
        for (int l = 0; l < IO_LANES; ++l) warm[l].join();
        (void)hipFree(scratch);
      }
      (void)hipGetLastError();
    }
    d.io.ok = ok;
  }
  d.ready = true;
  return 0;
}
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
  if (g_ndev == 0 || !g_devs[t_cur_dev < g_ndev ? t_cur_dev : 0].ready)
    return set_error(MNT753_ENODEV, "no HIP device: call mnt753_init() on a machine with an MI355X (there is no CPU fallback)");
  return 0;
}
int current_physical_device() { return g_ndev ? g_devs[t_cur_dev < g_ndev ? t_cur_dev : 0].phys : -1; }
int physical_device_of(int logical) { return logical >= 0 && logical < g_ndev ? g_devs[logical].phys : -1; }
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
    g_ndev = 0;
    return set_error(MNT753_ENODEV, "mnt753_init: no HIP device visible");
  }
  if (device < 0 || device >= count) return set_error(MNT753_EINVAL, "mnt753_init: device ordinal out of range");
  if (int rc = init_one(0, device)) return rc;
  if (g_ndev < 1) g_ndev = 1;
  t_cur_dev = 0;
  return 0;
}

int mnt753_init_devices(int n_devices) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_ndev = 0;
    return set_error(MNT753_ENODEV, "mnt753_init_devices: no HIP device visible");
  }
  const char* sh = getenv("MNT753_SHARE_DEVICE");
  const bool share = sh && atoi(sh) != 0;
  if (n_devices < 1 || n_devices > MAX_DEVICES || (!share && n_devices > count))
    return set_error(MNT753_EINVAL, "mnt753_init_devices: more devices requested than visible (MNT753_SHARE_DEVICE=1 maps them onto the visible ones)");
  for (int i = 0; i < n_devices; ++i)
    if (int rc = init_one(i, share ? i % count : i)) return rc;
  g_ndev = n_devices;
  // peer access for EVERY ordered pair the sharded prover copies between (cuda_prover_piecewise.cu:24-34 has one device; here the
  // transformed cb / cc travel 1 -> 0 and 2 -> 0, the slices of coefficients_for_H 0 -> g, operands of B:: vector calls any -> any):
  // without it hipMemcpyPeerAsync stages through host memory.  Failing to enable a pair is not an error (the copy still works).
  for (int a = 0; a < n_devices; ++a)
    for (int b = 0; b < n_devices; ++b)
      if (a != b) { int how = 0; (void)mnt753_enable_peer_access(a, b, &how); }
  g_ndev = n_devices;
  t_cur_dev = 0;
  HIP_TRY(hipSetDevice(g_devs[0].phys));
  return 0;
}

int mnt753_device_count(void) { return g_ndev; }

int mnt753_enable_peer_access(int device, int peer, int* how) {
  if (device < 0 || device >= g_ndev || peer < 0 || peer >= g_ndev || !g_devs[device].ready || !g_devs[peer].ready)
    return set_error(MNT753_EINVAL, "enable_peer_access: not an initialised device");
  int& state = g_peer[device][peer];
  if (state == 0) {
    const int pa = g_devs[device].phys, pb = g_devs[peer].phys;
    if (pa == pb) state = 1 + MNT753_PEER_SAME;
    else {
      int cur = -1, can = 0;
      if (hipGetDevice(&cur) != hipSuccess) cur = -1;
      state = 1 + MNT753_PEER_STAGED;
      if (hipDeviceCanAccessPeer(&can, pa, pb) == hipSuccess && can && hipSetDevice(pa) == hipSuccess) {
        const hipError_t e = hipDeviceEnablePeerAccess(pb, 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) state = 1 + MNT753_PEER_DIRECT;
      }
      (void)hipGetLastError();
      if (cur >= 0) (void)hipSetDevice(cur);
    }
  }
  if (how) *how = state - 1;
  return 0;
}

int mnt753_set_device(int logical) {
  if (logical < 0 || logical >= g_ndev || !g_devs[logical].ready) return set_error(MNT753_EINVAL, "mnt753_set_device: not an initialised device");
  t_cur_dev = logical;
  HIP_TRY(hipSetDevice(g_devs[logical].phys));
  return 0;
}

int mnt753_get_device(void) { return t_cur_dev < g_ndev ? t_cur_dev : 0; }

int mnt753_copy_peer(int dst_device, void* dev_dst, int src_device, const void* dev_src, size_t bytes) {
  if (dst_device < 0 || dst_device >= g_ndev || src_device < 0 || src_device >= g_ndev) return set_error(MNT753_EINVAL, "copy_peer: bad device");
  if (bytes && (!dev_dst || !dev_src)) return set_error(MNT753_EINVAL, "copy_peer: null");
  if (g_devs[dst_device].phys == g_devs[src_device].phys) HIP_TRY(hipMemcpy(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice));
  else HIP_TRY(hipMemcpyPeer(dev_dst, g_devs[dst_device].phys, dev_src, g_devs[src_device].phys, bytes));
  return 0;
}

int mnt753_copy_peer_async(int dst_device, void* dev_dst, int src_device, const void* dev_src, size_t bytes) {
  if (dst_device < 0 || dst_device >= g_ndev || src_device < 0 || src_device >= g_ndev) return set_error(MNT753_EINVAL, "copy_peer_async: bad device");
  if (bytes && (!dev_dst || !dev_src)) return set_error(MNT753_EINVAL, "copy_peer_async: null");
  if (bytes == 0) return 0;
  const int pd = g_devs[dst_device].phys, ps = g_devs[src_device].phys;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
  struct Back { int d; ~Back() { if (d >= 0) (void)hipSetDevice(d); } } back{cur};
  if (pd == ps) {   // logical devices sharing one GPU (MNT753_SHARE_DEVICE): the default stream orders it by itself
    HIP_TRY(hipSetDevice(pd));
    HIP_TRY(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, nullptr));
    return 0;
  }
  hipEvent_t& ev = g_devs[src_device].xfer_ev[dst_device];
  HIP_TRY(hipSetDevice(ps));
  if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ev, nullptr));
  HIP_TRY(hipSetDevice(pd));
  HIP_TRY(hipStreamWaitEvent(nullptr, ev, 0));
  HIP_TRY(hipMemcpyPeerAsync(dev_dst, pd, dev_src, ps, bytes, nullptr));
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
int mnt753_dev_mem_info(size_t* free_bytes, size_t* total_bytes) {
  if (int rc = require_device()) return rc;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return 0;
}
int mnt753_load_file_to_device(const char* path, size_t file_offset, size_t bytes, void* dev_dst) {
  if (int rc = require_device()) return rc;
  if (!path || (bytes && !dev_dst)) return set_error(MNT753_EINVAL, "load_file_to_device: null argument");
  IoStaging& g_io = g_devs[t_cur_dev < g_ndev ? t_cur_dev : 0].io;
  const int phys = current_physical_device();
  HIP_TRY(hipSetDevice(phys));   // may be called from a thread that has not touched the device yet
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return set_error(MNT753_EINVAL, "load_file_to_device: cannot open file");
  struct stat fst;
  if (fstat(fd, &fst) != 0 || (unsigned long long)fst.st_size < (unsigned long long)file_offset + bytes) {
    close(fd);
    return set_error(MNT753_EINVAL, "load_file_to_device: short read");
  }
  std::lock_guard<std::mutex> lock(g_io.mu);
  if (!g_io.ok) { close(fd); return set_error(MNT753_EHIP, "load_file_to_device: staging buffers were not created by mnt753_init"); }
  constexpr size_t CHUNK = IoStaging::CHUNK;
  // one lane's share: chunks of CHUNK, the pread of chunk i + 1 overlaps the H2D copy of chunk i; 0 = fine, 1 = read error, 2 = HIP error
  auto run_lane = [&](int l, size_t lo, size_t hi) -> int {
    IoLane& L = g_io.lane[l];
    if (hipSetDevice(phys) != hipSuccess) return 2;
    int rc = 0;
    size_t off = lo;
    for (size_t i = 0; off < hi; ++i) {
      const int k = (int)(i & 1);
      const size_t n = hi - off < CHUNK ? hi - off : CHUNK;
      if (i >= 2 && hipEventSynchronize(L.done[k]) != hipSuccess) { rc = 2; break; }
      size_t got = 0;
      while (got < n) {
        const ssize_t r = pread(fd, (char*)L.buf[k] + got, n - got, (off_t)(file_offset + off + got));
        if (r <= 0) break;
        got += (size_t)r;
      }
      if (got != n) { rc = 1; break; }
      if (hipMemcpyAsync((char*)dev_dst + off, L.buf[k], n, hipMemcpyHostToDevice, L.stream) != hipSuccess || hipEventRecord(L.done[k], L.stream) != hipSuccess) { rc = 2; break; }
      off += n;
    }
    if (hipStreamSynchronize(L.stream) != hipSuccess && rc == 0) rc = 2;
    return rc;
  };
  // small reads stay on one lane; large ones are cut into IO_LANES contiguous parts (multiples of CHUNK)
  // What made the loader twice as fast in round 3 was pread straight into the pinned buffer instead of fread through a stdio buffer
  // (403 MB: 21 -> 12 ms).  More lanes bring the 403 MB of a 2^20 input from 11-17 ms to 10-12; in round 3 the FIRST proof of a
  // process -- the reference's metric -- paid 5-30 ms for them, which was the first large copy of every lane's stream: the
  // initialisation sends both buffers of every lane over once since round 4, and two lanes are the default (first proof 0.1562 ->
  // 0.1557 s median of six alternations, later proofs 0.1552 -> 0.1548: profiles/r04/prove_io_lanes.txt).
  constexpr int max_lanes = 2;
  const int lanes = bytes >= 4 * CHUNK ? max_lanes : 1;
  const size_t per = ((bytes / (size_t)lanes + CHUNK - 1) / CHUNK) * CHUNK;
  int rcs[IO_LANES] = {0, 0};
  std::thread workers[IO_LANES];
  for (int l = 1; l < lanes; ++l) {
    const size_t lo = std::min(bytes, (size_t)l * per), hi = std::min(bytes, (size_t)(l + 1) * per);
    workers[l] = std::thread([&, l, lo, hi]() { rcs[l] = run_lane(l, lo, hi); });
  }
  rcs[0] = run_lane(0, 0, std::min(bytes, per));
  for (int l = 1; l < lanes; ++l) workers[l].join();
  close(fd);
  for (int l = 0; l < lanes; ++l) {
    if (rcs[l] == 1) return set_error(MNT753_EINVAL, "load_file_to_device: short read");
    if (rcs[l] == 2) return set_error(MNT753_EHIP, "load_file_to_device: copy");
  }
  return 0;
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
