// The one exchange of a sharded MSM, over RCCL, INSIDE the boundary (include/mnt753_hip.h: mnt753_exchange_points).
//
// The reference shards an MSM over OpenMP threads and sums the partial results serially (depends/libff/libff/algebra/
// scalar_multiplication/multiexp.tcc:417-440).  Lifted to the GPUs of a node the partial results are one projective point per device
// (36 / 72 / 108 u64) and "sum serially" needs them in one place: an all-gather.  SURVEY.md section 8e asks for that collective over
// xGMI; this file is it for the one-process form (B::use_devices): a single-process RCCL communicator over the devices of
// mnt753_init_devices (ncclCommInitAll), one ncclAllGather per device inside a group call, every device's block staged through a
// persistent device buffer on that device's exchange stream.  The fold stays on the host (EC addition is not a reduction operator).
//
// A partial point of this library ARRIVES on the host (the last 19 doublings and additions of a bucket reduction are a host Horner,
// DESIGN.md 4.3), so the host fold needs no collective at all and is the default; the RCCL path costs a round trip host -> device ->
// all-gather -> host and exists so that the exchange the north star names is in the product, selectable (MNT753_FOLD=rccl, main_hip
// --fold rccl) and measured (mnt753_exchange_points reports its own latency).
//
// librccl (570 MB) is loaded on first use with dlopen, never at library load time: a prover that folds on the host does not pay for it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "common_host.hpp"

namespace mnt753 {
int physical_device_of(int logical);   // mnt753_core.hip
}
using namespace mnt753;

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl lacks an expected symbol"; return false; }
    return true;
  }
};
struct Exchange {
  std::mutex mu;
  Rccl rccl;
  int n = 0;
  std::vector<int> phys;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> stream;
  std::vector<uint64_t*> d_in, d_out;
  uint64_t* h_pin = nullptr;       // pinned: n blocks in, n * n words out (device 0's gathered copy)
  size_t cap_words = 0;
  double last_us = 0;
};
Exchange g_ex;

int fail_nccl(const char* what, ncclResult_t r) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, g_ex.rccl.GetErrorString ? g_ex.rccl.GetErrorString(r) : "RCCL error");
  return set_error(MNT753_EHIP, buf);
}
}  // namespace

extern "C" {

// host_in[g]: `words` u64 of logical device g (its partial points); host_out: n_devices x words u64 in rank order.
int mnt753_exchange_points(const uint64_t* const* host_in, size_t words, uint64_t* host_out) {
  if (!host_in || !host_out || words == 0 || words > (1u << 16)) return set_error(MNT753_EINVAL, "exchange_points: bad argument");
  if (int rc = require_device()) return rc;
  std::lock_guard<std::mutex> lock(g_ex.mu);
  const int n = mnt753_device_count();
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
  struct Back { int d; ~Back() { if (d >= 0) (void)hipSetDevice(d); } } back{cur};
  if (g_ex.n != n) {
    // (re)build the communicator: the logical devices must be distinct GPUs (MNT753_SHARE_DEVICE maps several onto one: no collective)
    std::vector<int> phys((size_t)n);
    for (int g = 0; g < n; ++g) phys[(size_t)g] = physical_device_of(g);
    for (int g = 0; g < n; ++g)
      for (int h = 0; h < g; ++h)
        if (phys[(size_t)g] == phys[(size_t)h]) return set_error(MNT753_ENODEV, "exchange_points: logical devices share a GPU (MNT753_SHARE_DEVICE): no RCCL communicator over them");
    std::string err;
    if (!g_ex.rccl.load(err)) return set_error(MNT753_ENODEV, err.c_str());
    // whatever an earlier device count left behind goes first: communicators, streams, staging buffers (each on its own device)
    for (auto c : g_ex.comm) if (c) (void)g_ex.rccl.CommDestroy(c);
    for (size_t g = 0; g < g_ex.stream.size(); ++g) {
      if (g < g_ex.phys.size()) (void)hipSetDevice(g_ex.phys[g]);
      if (g_ex.stream[g]) (void)hipStreamDestroy(g_ex.stream[g]);
      if (g < g_ex.d_in.size() && g_ex.d_in[g]) (void)hipFree(g_ex.d_in[g]);
      if (g < g_ex.d_out.size() && g_ex.d_out[g]) (void)hipFree(g_ex.d_out[g]);
    }
    (void)hipGetLastError();
    g_ex.stream.clear(); g_ex.d_in.clear(); g_ex.d_out.clear();
    g_ex.cap_words = 0; g_ex.n = 0;
    g_ex.comm.assign((size_t)n, nullptr);
    if (ncclResult_t r = g_ex.rccl.CommInitAll(g_ex.comm.data(), n, phys.data()); r != ncclSuccess) { g_ex.comm.clear(); g_ex.n = 0; return fail_nccl("ncclCommInitAll", r); }
    g_ex.phys = phys;
    g_ex.stream.assign((size_t)n, nullptr);
    g_ex.d_in.assign((size_t)n, nullptr);
    g_ex.d_out.assign((size_t)n, nullptr);
    for (int g = 0; g < n; ++g) {
      HIP_TRY(hipSetDevice(phys[(size_t)g]));
      HIP_TRY(hipStreamCreateWithFlags(&g_ex.stream[(size_t)g], hipStreamNonBlocking));
    }
    g_ex.cap_words = 0;
    g_ex.n = n;
  }
  if (g_ex.cap_words < words) {
    // grow: every pointer is nulled as it is freed and the capacity is zero until all of them are back, so a failing allocation
    // leaves a state the next call rebuilds from instead of freed pointers it would free again or use
    g_ex.cap_words = 0;
    for (int g = 0; g < n; ++g) {
      HIP_TRY(hipSetDevice(g_ex.phys[(size_t)g]));
      if (g_ex.d_in[(size_t)g]) { (void)hipFree(g_ex.d_in[(size_t)g]); g_ex.d_in[(size_t)g] = nullptr; }
      if (g_ex.d_out[(size_t)g]) { (void)hipFree(g_ex.d_out[(size_t)g]); g_ex.d_out[(size_t)g] = nullptr; }
      HIP_TRY(hipMalloc(&g_ex.d_in[(size_t)g], 8 * words));
      HIP_TRY(hipMalloc(&g_ex.d_out[(size_t)g], 8 * words * (size_t)n));
    }
    if (g_ex.h_pin) { (void)hipHostFree(g_ex.h_pin); g_ex.h_pin = nullptr; }
    HIP_TRY(hipHostMalloc(&g_ex.h_pin, 8 * words * (size_t)n * 2));
    g_ex.cap_words = words;
  }
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t* h_in = g_ex.h_pin;
  uint64_t* h_out = g_ex.h_pin + words * (size_t)n;
  for (int g = 0; g < n; ++g) {
    if (!host_in[g]) return set_error(MNT753_EINVAL, "exchange_points: null block");
    memcpy(h_in + words * (size_t)g, host_in[g], 8 * words);
    HIP_TRY(hipSetDevice(g_ex.phys[(size_t)g]));
    HIP_TRY(hipMemcpyAsync(g_ex.d_in[(size_t)g], h_in + words * (size_t)g, 8 * words, hipMemcpyHostToDevice, g_ex.stream[(size_t)g]));
  }
  if (ncclResult_t r = g_ex.rccl.GroupStart(); r != ncclSuccess) return fail_nccl("ncclGroupStart", r);
  for (int g = 0; g < n; ++g)
    if (ncclResult_t r = g_ex.rccl.AllGather(g_ex.d_in[(size_t)g], g_ex.d_out[(size_t)g], words, ncclUint64, g_ex.comm[(size_t)g], g_ex.stream[(size_t)g]); r != ncclSuccess) {
      (void)g_ex.rccl.GroupEnd();
      return fail_nccl("ncclAllGather", r);
    }
  if (ncclResult_t r = g_ex.rccl.GroupEnd(); r != ncclSuccess) return fail_nccl("ncclGroupEnd", r);
  HIP_TRY(hipSetDevice(g_ex.phys[0]));
  HIP_TRY(hipMemcpyAsync(h_out, g_ex.d_out[0], 8 * words * (size_t)n, hipMemcpyDeviceToHost, g_ex.stream[0]));
  for (int g = 0; g < n; ++g) {
    HIP_TRY(hipSetDevice(g_ex.phys[(size_t)g]));
    HIP_TRY(hipStreamSynchronize(g_ex.stream[(size_t)g]));
  }
  memcpy(host_out, h_out, 8 * words * (size_t)n);
  g_ex.last_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// microseconds the last mnt753_exchange_points took (staging, collective, copy back), 0 before the first
double mnt753_exchange_last_us(void) { return g_ex.last_us; }

}  // extern "C"
