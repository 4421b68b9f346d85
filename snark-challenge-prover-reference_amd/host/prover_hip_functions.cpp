// Implementation of include/prover_hip_functions.hpp on top of the C ABI (include/mnt753_hip.h).
// Counterpart of the reference's libsnark/prover_reference_functions.cpp (662 lines of libff/libfqfft calls):
// here every vector lives in HBM and every heavy call is a HIP kernel launch.
#include "../../include/prover_hip_functions.hpp"

#include <sys/stat.h>

#include <condition_variable>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include "../../include/mnt753_hip.h"

namespace mnt753_hip_detail {

[[noreturn]] static void fail(const char* what) {
  throw std::runtime_error(std::string(what) + ": " + mnt753_last_error());
}
static void check(int rc, const char* what) { if (rc != 0) fail(what); }
static bool trace_on() { const char* e = getenv("MNT753_TRACE"); return e && atoi(e) != 0; }
// MNT753_TRACE=1 or MNT753_TRACE_LOAD=1: where the seconds of a parameter load go (file reads, base sets with their window tables,
// domains, warm-up).  MNT753_TRACE_LOAD alone prints nothing inside a proof's timing window (bench.py's timed child uses it).
static bool trace_load_on() { const char* e = getenv("MNT753_TRACE_LOAD"); return trace_on() || (e && atoi(e) != 0); }
struct LoadTrace {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what) {
    const auto t1 = std::chrono::steady_clock::now();
    if (trace_load_on()) fprintf(stderr, "mnt753: load params: %-46s %7.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};

// Per-proof vectors (w, ca, cb, cc, coefficients_for_H, the per-device slices) have the same sizes proof after proof, and both
// hipMalloc and hipFree are expensive where it hurts: four 100 MB hipMallocs open the reference's timing window (1 ms on an idle box,
// 9 ms measured on a loaded one) and every hipFree is a device-wide synchronisation.  Freed blocks are therefore kept, per
// (device, size), and handed to the next buffer of that size; B::read_params pre-allocates the set of a proof so that the first proof
// finds them too; B::delete_groth16_params gives everything back.
struct BufferCache {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> blocks;
  size_t held = 0;
  static constexpr size_t LIMIT = (size_t)4 << 30;   // bytes kept per process
  void* take(int dev, size_t n) {
    std::lock_guard<std::mutex> l(mu);
    auto it = blocks.find({dev, n});
    if (it == blocks.end()) return nullptr;
    void* p = it->second;
    blocks.erase(it);
    held -= n;
    return p;
  }
  bool give(int dev, size_t n, void* p) {
    std::lock_guard<std::mutex> l(mu);
    if (held + n > LIMIT) return false;
    blocks.insert({{dev, n}, p});
    held += n;
    return true;
  }
  void release_all() {
    std::multimap<std::pair<int, size_t>, void*> all;
    { std::lock_guard<std::mutex> l(mu); all.swap(blocks); held = 0; }
    for (auto& kv : all) mnt753_dev_free(kv.second);
  }
};
static BufferCache g_buffers;
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int device = 0;   // logical device the block lives on (the calling thread's current one)
  explicit DeviceBuffer(size_t n) : bytes(n), device(mnt753_get_device()) {
    ptr = g_buffers.take(device, n);
    if (!ptr) check(mnt753_dev_alloc(&ptr, n), "mnt753_dev_alloc");
  }
  ~DeviceBuffer() { if (ptr && !g_buffers.give(device, bytes, ptr)) mnt753_dev_free(ptr); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
};
// Work enqueued through the C ABI's plain vector entry points (copies, vec_muleq / subeq / scale) lands on the calling thread's current
// logical device: this scope puts the thread on `dev` for a few calls and back afterwards.  A no-op with one device.
struct DeviceScope {
  int back = -1;
  explicit DeviceScope(int dev) {
    if (mnt753_device_count() < 2) return;
    const int cur = mnt753_get_device();
    if (cur != dev) { check(mnt753_set_device(dev), "mnt753_set_device"); back = cur; }
  }
  ~DeviceScope() { if (back >= 0) (void)mnt753_set_device(back); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
// one-shot readiness latch: set by the input loader thread, awaited by the first consumer of a vector
struct Ready {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  std::string error;
  void set(const std::string& err = std::string()) { { std::lock_guard<std::mutex> l(mu); done = true; error = err; } cv.notify_all(); }
  void wait() {
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] { return done; });
    if (!error.empty()) throw std::runtime_error(error);
  }
};
// g_gate[g] = the base set of C's MSM on device g until the next MSM there is ordered behind it (start_sharded, below); a base set
// that is going away is nobody's predecessor any more
static std::vector<mnt753_bases*> g_gate;
static void gate_forget(mnt753_bases* b) {
  for (auto& p : g_gate) if (p == b) p = nullptr;
}
struct BaseSetHolder {
  mnt753_bases* h = nullptr;
  ~BaseSetHolder() { if (h) { gate_forget(h); mnt753_bases_free(h); } }
};
static int g_n_devices = 0;   // 0: not chosen yet (init_public_params reads MNT753_GPUS, default 1)
// Ht + Lt + r Bt1 as ONE multi-scalar multiplication over the concatenated base set H | L | B1 (B::groth16_C).  -1: not chosen yet
// (read_params reads MNT753_FUSED_C, default on).
static int g_fused_c = -1;
static bool fused_c() {
  if (g_fused_c < 0) { const char* e = getenv("MNT753_FUSED_C"); g_fused_c = e ? (atoi(e) != 0) : 1; }
  return g_fused_c != 0;
}
// B::one_shot(true): the process will prove once (the reference's CLI, libsnark/main.cpp:274-293): read_params builds no window tables
// and runs no warm-up MSM -- both pay only over the proofs of a resident prover.  MNT753_ONE_SHOT=0 / 1 overrides.
static int g_one_shot = 0;
static bool one_shot_mode() {
  if (const char* e = getenv("MNT753_ONE_SHOT")) return atoi(e) != 0;
  return g_one_shot != 0;
}
// contiguous slice g of n elements over n_dev devices (multiexp.tcc:417-431: one = n / chunks, the last slice takes the remainder)
static void slice_bounds(size_t n, int n_dev, int g, size_t* lo, size_t* hi) {
  const size_t one = n / (size_t)n_dev;
  *lo = (size_t)g * one;
  *hi = g == n_dev - 1 ? n : (size_t)(g + 1) * one;
}
// elements [first, first + count) of a scalar vector, resident on one logical device (streamed there from the input file by that
// device's own loader thread)
struct DevSlice {
  std::shared_ptr<DeviceBuffer> buf;
  size_t first = 0, count = 0;
  std::shared_ptr<Ready> ready;
};
// A parameter vector cut into contiguous slices, slice g resident on logical device g (multiexp.tcc:417-431: one = n / chunks,
// the last slice takes the remainder).  One device = one slice = the single-GPU wrapper.
struct ShardedBases {
  struct Part {
    std::shared_ptr<BaseSetHolder> set;
    size_t lo = 0, hi = 0;
    std::shared_ptr<DeviceBuffer> scalars;   // staging for the scalar slice on devices other than the vector's home (grow-only)
    // the concatenated set H | L | B1 over several devices: part g = H[sub[0]) | L[sub[1]) | B1[sub[2]) -- slice g of EACH of the three
    // vectors (multiexp.tcc:417-431 applied to each of the reference's three multiexps), so that device g needs the same range of w for
    // all of its base sets and its own slice of coefficients_for_H
    size_t sub_lo[3] = {0, 0, 0}, sub_hi[3] = {0, 0, 0};
  };
  std::vector<Part> parts;
  size_t n = 0;
  bool interleaved = false;   // parts carry sub-ranges (the concatenated set on more than one device)
};
// one MSM in flight on every slice of a sharded vector
struct PendingMsm {
  std::vector<std::shared_ptr<BaseSetHolder>> sets;
  std::vector<int> devs;                   // logical device of sets[k] (rank order)
  std::shared_ptr<DeviceBuffer> scalars;   // scalar vector assembled for this MSM alone (groth16_C): lives as long as the MSM does
};
// A G1 value that has not been computed yet.  With the concatenated base set H | L | B1 resident (fused parameters), the three
// multiexps over B1, L and H are not started when the driver asks for them: cuda_prover_piecewise.cu:79-90 only ever uses their
// results in  C = Ht + (Lt + r * Bt1),  and that whole expression is ONE multi-scalar multiplication over the concatenated set.
// B::multiexp_G1 / G1_scale / G1_add therefore build this little expression tree, and the addition that completes the pattern starts
// the single MSM.  Anything else a caller does with such a value (print it, write it, add it to something unrelated) evaluates the
// tree the plain way -- each MSM on its own base set, built on first use.
struct LazyPoint {
  enum Kind { VALUE, MSM, SCALE, ADD } kind = VALUE;
  uint64_t value[36] = {0};                       // VALUE: projective, wire format
  void* owner = nullptr;                          // MSM: the groth16_params the base vector belongs to,
  int which = 0;                                  //      which of its vectors (1 B1, 2 L, 3 H),
  size_t length = 0;                              //      and the scalars (kept alive by `keep`)
  std::function<const uint64_t*()> scalars;
  int home = 0;                                   //      logical device that pointer is valid on
  std::shared_ptr<std::vector<DevSlice>> slices;
  size_t offset = 0;
  std::shared_ptr<void> keep;
  uint64_t k[12] = {0};                           // SCALE: k * a
  std::shared_ptr<LazyPoint> a, b;                // SCALE: a;  ADD: a + b
};
struct DomainHolder {
  mnt753_domain* h = nullptr;
  ~DomainHolder() { if (h) mnt753_domain_free(h); }
};

static void read_exact(FILE* f, void* dst, size_t bytes, const char* path) {
  if (bytes && fread(dst, 1, bytes, f) != bytes) { fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
}
}  // namespace mnt753_hip_detail

using namespace mnt753_hip_detail;

// an evaluation domain is its size; the tables live per device (cached_domain), `data` is device 0's
template <int CURVE> struct mnt753_hip_impl<CURVE>::evaluation_domain { std::shared_ptr<DomainHolder> data; size_t m = 0; };
template <int CURVE> struct mnt753_hip_impl<CURVE>::field { uint64_t data[12]; };
// G1 / G2 returned by multiexp_* are lazy: the MSM is in flight on its base set's stream until the value is first used
template <int CURVE> struct mnt753_hip_impl<CURVE>::G1 { uint64_t data[36]; std::shared_ptr<PendingMsm> pending; std::shared_ptr<LazyPoint> lazy; };
template <int CURVE> struct mnt753_hip_impl<CURVE>::G2 { uint64_t data[108]; std::shared_ptr<PendingMsm> pending; };  // 72 used on MNT4753, 108 on MNT6753
template <int CURVE> struct mnt753_hip_impl<CURVE>::vector_Fr {
  std::shared_ptr<DeviceBuffer> data;
  size_t size;     // elements in the underlying buffer
  size_t offset;   // element offset honoured by multiexp / muleq / subeq (prover_reference_functions.cpp:173,254)
  std::shared_ptr<Ready> ready;   // set for vectors of a groth16_input that is still streaming in from its file
  // several devices: ranges of the same vector that other devices hold (index = logical device, element indices of the underlying
  // buffer); empty for vectors that only exist on device 0 (coefficients_for_H)
  std::shared_ptr<std::vector<DevSlice>> slices;
  std::shared_ptr<void> keep;     // buffers that kernels enqueued for this vector still read (staged copies of operands from other devices)
  // device pointer (on device()); waits (once) until the loader thread has put the vector there
  uint64_t* ptr() const {
    if (ready) ready->wait();
    return reinterpret_cast<uint64_t*>(data->ptr) + 12 * offset;
  }
  int device() const { return data->device; }   // logical device the vector lives on (cb / cc of a sharded prover: devices 1 / 2)
};
// owner / which: set for B1 (1), L (2), H (3) of fused parameters, whose own base sets exist only once something needs them
template <int CURVE> struct mnt753_hip_impl<CURVE>::vector_G1 { std::shared_ptr<ShardedBases> data; groth16_params* owner = nullptr; int which = 0; };
template <int CURVE> struct mnt753_hip_impl<CURVE>::vector_G2 { std::shared_ptr<ShardedBases> data; };

// params file: u64 d, u64 m, A[m+1] G1, B1[m+1] G1, B2[m+1] G2, L[m-1] G1, H[d] G1
// (libsnark/generate_parameters.cpp:60-85, reader prover_reference_functions.cpp:86-116)
template <int CURVE>
class mnt753_hip_impl<CURVE>::groth16_params {
public:
  size_t d = 0, m = 0;
  // A and B2 always; B1, L, H either as three base sets (the reference's five multiexps) or, fused, as the one concatenated set HLB
  // = H[d] | L[m-1] | B1[m+1] that B::groth16_C multiplies in one pass -- the three separate sets are then built on first use only
  // (B::params_B1 / params_L / params_H), from the file.
  std::shared_ptr<ShardedBases> A, B1, L, H, B2, HLB;
  std::string path;
  size_t off_B1 = 0, off_L = 0, off_H = 0;   // byte offsets of the vectors in the params file
  std::mutex mu;
  static std::shared_ptr<ShardedBases> make_set(int group, size_t words, size_t n, const uint64_t* host) {
    const int n_dev = std::max(1, mnt753_device_count());
    auto sb = std::make_shared<ShardedBases>();
    sb->n = n;
    struct BackToDevice0 { int n; ~BackToDevice0() { if (n > 1) (void)mnt753_set_device(0); } } back{n_dev};   // also when a creation throws
    for (int g = 0; g < n_dev; ++g) {
      ShardedBases::Part part;
      slice_bounds(n, n_dev, g, &part.lo, &part.hi);
      part.set = std::make_shared<BaseSetHolder>();
      if (n_dev > 1) check(mnt753_set_device(g), "mnt753_set_device");
      check(mnt753_bases_create(CURVE, group, host + words * part.lo, 0, part.hi - part.lo, &part.set->h), "mnt753_bases_create");
      sb->parts.push_back(part);
    }
    return sb;
  }
  explicit groth16_params(const char* path_) : path(path_) {
    FILE* f = fopen(path_, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open params file ") + path);
    uint64_t dm[2];
    read_exact(f, dm, 16, path_);
    d = dm[0]; m = dm[1];
    const size_t g1w = mnt753_affine_words(CURVE, MNT753_G1), g2w = mnt753_affine_words(CURVE, MNT753_G2);
    // The reference trusts d and m (prover_reference_functions.cpp:86-116); here they size device allocations, so they
    // are checked against the file before anything is allocated: 16 + 8*(g1w*(3m + d + 1) + g2w*(m + 1)) bytes exactly.
    {
      struct stat st;
      if (m < 2 || d < 1 || m > ((size_t)1 << 31) || d > ((size_t)1 << 31) || stat(path_, &st) != 0) {
        fclose(f);
        throw std::runtime_error(std::string("bad params header (d, m) in ") + path);
      }
      const unsigned long long expect = 16ull + 8ull * ((unsigned long long)g1w * (3ull * m + d + 1ull) + (unsigned long long)g2w * (m + 1ull));
      if ((unsigned long long)st.st_size != expect) {
        fclose(f);
        throw std::runtime_error(std::string("params file size does not match its header (d=") + std::to_string(d) + ", m=" + std::to_string(m) +
                                 ", expected " + std::to_string(expect) + " bytes, found " + std::to_string((unsigned long long)st.st_size) + "): " + path);
      }
    }
    off_B1 = 16 + 8 * g1w * (m + 1);
    off_L = off_B1 + 8 * (g1w + g2w) * (m + 1);
    off_H = off_L + 8 * g1w * (m - 1);
    const bool fused = fused_c();
    std::vector<uint64_t> host, cat;
    LoadTrace lt;
    auto load = [&](int group, size_t words, size_t n, size_t cat_at) {
      host.resize(words * n);
      read_exact(f, host.data(), host.size() * 8, path_);
      if (fused && group == MNT753_G1 && cat_at != (size_t)-1) {   // a part of the concatenated set: keep the host copy, no set of its own
        memcpy(cat.data() + g1w * cat_at, host.data(), host.size() * 8);
        lt.lap("read a part of H | L | B1 from the file");
        return std::shared_ptr<ShardedBases>();
      }
      lt.lap(group == MNT753_G1 ? "read A from the file" : "read B2 from the file");
      auto set = make_set(group, words, n, host.data());
      lt.lap(group == MNT753_G1 ? "base set A (upload, window table, workspace)" : "base set B2 (upload, window table, workspace)");
      return set;
    };
    if (fused) cat.resize(g1w * (d + 2 * m));
    try {
      A = load(MNT753_G1, g1w, m + 1, (size_t)-1);
      B1 = load(MNT753_G1, g1w, m + 1, d + m - 1);
      B2 = load(MNT753_G2, g2w, m + 1, (size_t)-1);
      L = load(MNT753_G1, g1w, m - 1, d);
      H = load(MNT753_G1, g1w, d, 0);
      if (fused) { HLB = make_hlb_set(g1w, cat); lt.lap("base set H | L | B1 (upload, window table, workspace)"); }
    } catch (...) { fclose(f); throw; }
    fclose(f);
  }
  // the concatenated set H[d] | L[m-1] | B1[m+1] (host copy in `cat`): one device holds it as it is; over several devices part g is
  // slice g of H, of L and of B1 (ShardedBases::Part::sub_lo / sub_hi)
  std::shared_ptr<ShardedBases> make_hlb_set(size_t g1w, const std::vector<uint64_t>& cat) {
    const int n_dev = std::max(1, mnt753_device_count());
    if (n_dev == 1) return make_set(MNT753_G1, g1w, d + 2 * m, cat.data());
    auto sb = std::make_shared<ShardedBases>();
    sb->n = d + 2 * m;
    sb->interleaved = true;
    struct BackToDevice0 { ~BackToDevice0() { (void)mnt753_set_device(0); } } back;
    const size_t len[3] = {d, m - 1, m + 1}, at[3] = {0, d, d + m - 1};
    std::vector<uint64_t> rows;
    for (int g = 0; g < n_dev; ++g) {
      ShardedBases::Part part;
      rows.clear();
      for (int k = 0; k < 3; ++k) {
        slice_bounds(len[k], n_dev, g, &part.sub_lo[k], &part.sub_hi[k]);
        rows.insert(rows.end(), cat.begin() + g1w * (at[k] + part.sub_lo[k]), cat.begin() + g1w * (at[k] + part.sub_hi[k]));
      }
      part.lo = 0; part.hi = rows.size() / g1w;
      part.set = std::make_shared<BaseSetHolder>();
      check(mnt753_set_device(g), "mnt753_set_device");
      check(mnt753_bases_create(CURVE, MNT753_G1, rows.data(), 0, part.hi, &part.set->h), "mnt753_bases_create");
      sb->parts.push_back(part);
    }
    return sb;
  }
  // B1, L, H as base sets of their own (the reference's call sequence asks for them): built now if the parameters were loaded fused
  void ensure_separate() {
    std::lock_guard<std::mutex> l(mu);
    if (B1 && L && H) return;
    const size_t g1w = mnt753_affine_words(CURVE, MNT753_G1);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error(std::string("cannot open params file ") + path);
    std::vector<uint64_t> host;
    auto load_at = [&](size_t off, size_t n) {
      host.resize(g1w * n);
      if (fseeko(f, (off_t)off, SEEK_SET) != 0) { fclose(f); throw std::runtime_error(std::string("seek failed: ") + path); }
      read_exact(f, host.data(), host.size() * 8, path.c_str());
      return make_set(MNT753_G1, g1w, n, host.data());
    };
    try {
      if (!B1) B1 = load_at(off_B1, m + 1);
      if (!L) L = load_at(off_L, m - 1);
      if (!H) H = load_at(off_H, d);
    } catch (...) { fclose(f); throw; }
    fclose(f);
  }
};

struct R1csHolder {
  mnt753_r1cs* h = nullptr;
  ~R1csHolder() { if (h) mnt753_r1cs_free(h); }
};
template <int CURVE>
class mnt753_hip_impl<CURVE>::r1cs {
public:
  std::shared_ptr<R1csHolder> data;
  uint64_t num_inputs = 0, m = 0, nc = 0;
  explicit r1cs(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open r1cs file ") + path);
    uint64_t hdr[3];
    read_exact(f, hdr, 24, path);   // read_exact closes f before it throws
    num_inputs = hdr[0]; m = hdr[1]; nc = hdr[2];
    struct stat st;
    if (stat(path, &st) != 0 || nc > ((uint64_t)1 << 31) || m > ((uint64_t)1 << 31) || (uint64_t)st.st_size < 24 + 3 * 8 * (nc + 1)) {
      fclose(f);
      throw std::runtime_error(std::string("bad r1cs header in ") + path);
    }
    std::vector<uint64_t> rp[3], cf[3];
    std::vector<uint32_t> col[3];
    for (int k = 0; k < 3; ++k) {
      rp[k].resize(nc + 1);
      read_exact(f, rp[k].data(), 8 * (nc + 1), path);
      const uint64_t nnz = rp[k][nc];
      if (nnz > (uint64_t)st.st_size / 100) { fclose(f); throw std::runtime_error(std::string("r1cs file shorter than its row pointers say: ") + path); }
      col[k].resize(nnz); cf[k].resize(12 * nnz);
      read_exact(f, col[k].data(), 4 * nnz, path);
      read_exact(f, cf[k].data(), 96 * nnz, path);
    }
    fclose(f);
    const uint64_t* rpp[3] = {rp[0].data(), rp[1].data(), rp[2].data()};
    const uint32_t* cp[3] = {col[0].data(), col[1].data(), col[2].data()};
    const uint64_t* fp[3] = {cf[0].data(), cf[1].data(), cf[2].data()};
    data = std::make_shared<R1csHolder>();
    check(mnt753_r1cs_create(CURVE, num_inputs, m, nc, rpp, cp, fp, &data->h), "mnt753_r1cs_create");
  }
};

// input file: w[m+1], ca[d+1], cb[d+1], cc[d+1], r   (generate_parameters.cpp:88-108, reader :48-76)
// The constructor returns at once; loader threads stream the vectors to the devices and release them one by one, so kernels that
// only need w (the G2 MSM and A's) start while ca / cb / cc are still being read.
//
// One device: one loader, file order (w, ca, cb, cc).
// Several devices (B::use_devices): NOTHING of the proof's critical path is funnelled through device 0 (round 4).  Every device g has
// its own loader thread, pinned staging buffers and PCIe link and reads from the file
//   * the range of w that its slices of A / B1 / B2 (w[i]) and L (w[2 + i]) multiply -- 96 (m + 1) / N bytes -- first;
//   * then ONE of the three vectors of compute_H: ca on device 0, cb on device 1, cc on device 2 (device 0 again when there are only
//     two).  Each device runs cosetFFT(iFFT(.)) on its own vector (B::compute_H_fused, or the B::domain_* calls of the reference's
//     compute_H<B>, which run where their vector lives), the transformed cb / cc travel to device 0 over xGMI
//     (mnt753_copy_peer_async) and the pointwise step and the last transform run there: 134 MB instead of 403 MB over device 0's
//     link, and two of the three transform chains off device 0;
//   * device 0, last and needed by nobody on the critical path, the rest of w (so that B::input_w still is a whole vector there).
template <int CURVE>
class mnt753_hip_impl<CURVE>::groth16_input {
public:
  std::shared_ptr<DeviceBuffer> w, ca, cb, cc;
  std::shared_ptr<Ready> w_ready, ca_ready, cb_ready, cc_ready;
  size_t n_w = 0, n_c = 0;
  uint64_t r[12];
  // seconds until the last loader was done (max over the loader threads; written before a thread sets its last latch)
  struct LoadClock { std::mutex mu; double secs = 0; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); };
  std::shared_ptr<LoadClock> clock = std::make_shared<LoadClock>();
  // ranges of w resident per device (index = logical device): device g > 0 a buffer of its own, device 0 a view into w that is
  // released as soon as ITS range is there (the whole of w follows later)
  std::shared_ptr<std::vector<DevSlice>> w_slices;
  std::vector<std::thread> loaders;
  struct Job { void* dst; size_t off, bytes; std::shared_ptr<Ready> ready; };   // ready: released after this job (may be null)

  static void run_jobs(const std::string& path, int device, std::vector<Job> jobs, std::shared_ptr<LoadClock> clk, std::function<std::string()> tail) {
    std::string err;
    if (mnt753_device_count() > 1 && mnt753_set_device(device) != 0) err = std::string("mnt753_set_device: ") + mnt753_last_error();
    for (size_t k = 0; k < jobs.size(); ++k) {
      const Job& job = jobs[k];
      if (err.empty() && job.bytes && mnt753_load_file_to_device(path.c_str(), job.off, job.bytes, job.dst) != 0)
        err = std::string("mnt753_load_file_to_device (device ") + std::to_string(device) + "): " + mnt753_last_error();
      if (k + 1 == jobs.size()) {
        if (err.empty() && tail) err = tail();
        std::lock_guard<std::mutex> l(clk->mu);
        clk->secs = std::max(clk->secs, std::chrono::duration<double>(std::chrono::steady_clock::now() - clk->t0).count());
      }
      if (job.ready) job.ready->set(err);
    }
  }
  // the range of w device g multiplies: A, B1, B2 take w[lo .. hi) of slice g of m + 1 elements, L takes w[2 + lo .. 2 + hi) of m - 1
  static void w_range(size_t m, int n_dev, int g, size_t* first, size_t* count) {
    size_t lo_a, hi_a, lo_l, hi_l;
    slice_bounds(m + 1, n_dev, g, &lo_a, &hi_a);
    slice_bounds(m - 1, n_dev, g, &lo_l, &hi_l);   // vector_Fr_offset(w, primary_input_size + 1)
    *first = std::min(lo_a, lo_l + 2);
    *count = std::max(hi_a, hi_l + 2) - *first;
  }
  void check_size(const char* path, unsigned long long expect, const char* what) {
    struct stat st;
    if (stat(path, &st) != 0 || (unsigned long long)st.st_size != expect)
      throw std::runtime_error(std::string(what) + " file size does not match the parameters (expected " + std::to_string(expect) + " bytes): " + path);
  }
  void read_r(const char* path, size_t r_off) {
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open input file ") + path);
    if (fseeko(f, (off_t)r_off, SEEK_SET) != 0 || fread(r, 1, 96, f) != 96) { fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
    fclose(f);
  }
  // buffers, latches and the per-device ranges of w; vec_home[k] = device of ca / cb / cc
  void allocate(size_t m, const int vec_home[3]) {
    const int n_dev = std::max(1, mnt753_device_count());
    w_ready = std::make_shared<Ready>(); ca_ready = std::make_shared<Ready>(); cb_ready = std::make_shared<Ready>();
    cc_ready = std::make_shared<Ready>();
    struct BackToDevice0 { int n; ~BackToDevice0() { if (n > 1) (void)mnt753_set_device(0); } } back{n_dev};   // also when an allocation throws
    w = std::make_shared<DeviceBuffer>(96 * n_w);
    std::shared_ptr<DeviceBuffer>* vecs[3] = {&ca, &cb, &cc};
    for (int k = 0; k < 3; ++k) {
      if (n_dev > 1) check(mnt753_set_device(vec_home[k]), "mnt753_set_device");
      *vecs[k] = std::make_shared<DeviceBuffer>(96 * n_c);
    }
    if (n_dev < 2) return;
    w_slices = std::make_shared<std::vector<DevSlice>>((size_t)n_dev);
    for (int g = 0; g < n_dev; ++g) {
      DevSlice& sl = (*w_slices)[(size_t)g];
      w_range(m, n_dev, g, &sl.first, &sl.count);
      sl.ready = std::make_shared<Ready>();
      if (g == 0) { sl.buf = w; continue; }   // device 0's range starts at w[0] (slice 0 of every vector): a view into w
      check(mnt753_set_device(g), "mnt753_set_device");
      sl.buf = std::make_shared<DeviceBuffer>(96 * sl.count);
    }
  }
  groth16_input(const char* path, size_t d, size_t m) {
    n_w = m + 1; n_c = d + 1;
    const size_t r_off = 96 * (n_w + 3 * n_c);
    check_size(path, (unsigned long long)r_off + 96ull, "input");
    read_r(path, r_off);
    const int n_dev = std::max(1, mnt753_device_count());
    const int vec_home[3] = {0, n_dev > 1 ? 1 : 0, n_dev > 2 ? 2 : 0};
    allocate(m, vec_home);
    const std::string p(path);
    const size_t off_ca = 96 * n_w, off_cb = 96 * (n_w + n_c), off_cc = 96 * (n_w + 2 * n_c);
    std::vector<std::vector<Job>> jobs((size_t)n_dev);
    if (n_dev == 1) {
      jobs[0] = {{w->ptr, 0, 96 * n_w, w_ready}, {ca->ptr, off_ca, 96 * n_c, ca_ready}, {cb->ptr, off_cb, 96 * n_c, cb_ready}, {cc->ptr, off_cc, 96 * n_c, cc_ready}};
    } else {
      for (int g = 1; g < n_dev; ++g) {
        const DevSlice& sl = (*w_slices)[(size_t)g];
        jobs[(size_t)g].push_back({sl.buf->ptr, 96 * sl.first, 96 * sl.count, sl.ready});
      }
      const DevSlice& s0 = (*w_slices)[0];
      jobs[0].push_back({w->ptr, 0, 96 * s0.count, s0.ready});
      jobs[0].push_back({ca->ptr, off_ca, 96 * n_c, ca_ready});
      jobs[(size_t)vec_home[1]].push_back({cb->ptr, off_cb, 96 * n_c, cb_ready});
      jobs[(size_t)vec_home[2]].push_back({cc->ptr, off_cc, 96 * n_c, cc_ready});
      jobs[0].push_back({(char*)w->ptr + 96 * s0.count, 96 * s0.count, 96 * (n_w - s0.count), w_ready});   // the rest of w, last
    }
    for (int g = 0; g < n_dev; ++g)
      if (!jobs[(size_t)g].empty()) loaders.emplace_back(run_jobs, p, g, jobs[(size_t)g], clock, std::function<std::string()>());
  }
  // witness file: w[m+1], r.  ca / cb / cc are evaluated on device 0 from the constraint system as soon as w is there.
  groth16_input(const char* path, size_t d, size_t m, std::shared_ptr<R1csHolder> cs) {
    n_w = m + 1; n_c = d + 1;
    check_size(path, 96ull * (n_w + 1), "witness");
    if (mnt753_r1cs_domain_size(cs->h) > n_c) throw std::runtime_error("the constraint system does not fit the parameters' evaluation domain");
    // the evaluation kernel gathers w[col] for col <= cs.m and copies w[0 .. num_inputs]: both must stay inside the m + 1 elements of w
    if (mnt753_r1cs_num_variables(cs->h) != m || mnt753_r1cs_num_inputs(cs->h) > m)
      throw std::runtime_error("the constraint system has " + std::to_string(mnt753_r1cs_num_variables(cs->h)) + " variables and " +
                               std::to_string(mnt753_r1cs_num_inputs(cs->h)) + " inputs, the parameters are for m = " + std::to_string(m));
    read_r(path, 96 * n_w);
    const int n_dev = std::max(1, mnt753_device_count());
    const int vec_home[3] = {0, 0, 0};
    allocate(m, vec_home);
    const std::string p(path);
    for (int g = 1; g < n_dev; ++g) {
      const DevSlice& sl = (*w_slices)[(size_t)g];
      loaders.emplace_back(run_jobs, p, g, std::vector<Job>{{sl.buf->ptr, 96 * sl.first, 96 * sl.count, sl.ready}}, clock, std::function<std::string()>());
    }
    auto w_ = w, ca_ = ca, cb_ = cb, cc_ = cc;
    auto ar = ca_ready, br = cb_ready, cr = cc_ready;
    auto s0 = w_slices ? (*w_slices)[0].ready : std::shared_ptr<Ready>();
    const size_t ncc = n_c;
    // after w: evaluate the constraint system (device 0), then release ca / cb / cc together
    auto tail = [w_, ca_, cb_, cc_, ncc, cs]() -> std::string {
      if (mnt753_r1cs_evaluate(cs->h, reinterpret_cast<const uint64_t*>(w_->ptr), reinterpret_cast<uint64_t*>(ca_->ptr), reinterpret_cast<uint64_t*>(cb_->ptr),
                               reinterpret_cast<uint64_t*>(cc_->ptr), ncc, nullptr) != 0 || mnt753_sync(nullptr) != 0)
        return std::string("mnt753_r1cs_evaluate: ") + mnt753_last_error();
      return std::string();
    };
    auto wr = w_ready;
    auto clk = clock;
    const size_t nw = n_w;
    loaders.emplace_back([p, w_, wr, s0, ar, br, cr, clk, tail, nw]() {
      // w first and released at once (the MSMs over w start), then the evaluation
      std::string err;
      if (mnt753_load_file_to_device(p.c_str(), 0, 96 * nw, w_->ptr) != 0) err = std::string("mnt753_load_file_to_device: ") + mnt753_last_error();
      wr->set(err);
      if (s0) s0->set(err);
      if (err.empty()) err = tail();
      {
        std::lock_guard<std::mutex> l(clk->mu);
        clk->secs = std::max(clk->secs, std::chrono::duration<double>(std::chrono::steady_clock::now() - clk->t0).count());
      }
      ar->set(err); br->set(err); cr->set(err);
    });
  }
  ~groth16_input() {
    for (auto& t : loaders) if (t.joinable()) t.join();
  }
};

#define HIP_B mnt753_hip_impl<CURVE>

// Where the partial points of a sharded MSM meet.  Default: on the host -- mnt753_msm_finish hands every partial point to the host
// anyway (the tail of a bucket reduction is a host Horner), so the serial fold of multiexp.tcc:433-438 needs no collective.
// MNT753_FOLD=rccl / B::fold_over_rccl(true) / main_hip --fold rccl: the blocks go through mnt753_exchange_points first -- an RCCL
// all-gather over the devices' communicator (xGMI), the collective SURVEY.md section 8e names -- and are folded from what came back; its
// latency is traced (MNT753_TRACE=1).  Where no communicator can exist (logical devices sharing one GPU, no librccl) the host fold runs.
static int g_fold_rccl = -1;
static bool fold_rccl() {
  if (g_fold_rccl < 0) { const char* e = getenv("MNT753_FOLD"); g_fold_rccl = e && !strcmp(e, "rccl") ? 1 : 0; }
  return g_fold_rccl != 0;
}
// collect the partial results of the slices and fold them in rank order (multiexp.tcc:433-438: final = final + partial[i])
template <int CURVE, int GROUP, class P> static void resolve_t(P* p) {
  if (!p->pending) return;
  const size_t pw = mnt753_projective_words(CURVE, GROUP);
  std::vector<std::vector<uint64_t>> parts;
  for (auto& set : p->pending->sets) {
    parts.emplace_back(108, 0);
    check(mnt753_msm_finish(set->h, parts.back().data()), "mnt753_msm_finish");
  }
  const int n_dev = std::max(1, mnt753_device_count());
  if (fold_rccl() && !parts.empty()) {
    // one block per logical device (the identity where a device had no points), through the collective, back in rank order
    uint64_t zero_aff[72] = {0}, ident[108];
    check(mnt753_point_from_affine(CURVE, GROUP, zero_aff, ident), "mnt753_point_from_affine");
    std::vector<const uint64_t*> in((size_t)n_dev, ident);
    for (size_t k = 0; k < parts.size(); ++k) in[(size_t)p->pending->devs[k]] = parts[k].data();
    std::vector<uint64_t> out((size_t)n_dev * pw);
    const int rc = mnt753_exchange_points(in.data(), pw, out.data());
    const bool trace = getenv("MNT753_TRACE") && atoi(getenv("MNT753_TRACE"));
    if (rc == 0) {
      if (trace) fprintf(stderr, "mnt753: partial points over RCCL: all-gather of %zu u64 x %d devices in %.1f us\n", pw, n_dev, mnt753_exchange_last_us());
      parts.clear();
      for (int g = 0; g < n_dev; ++g) parts.emplace_back(out.begin() + (size_t)g * pw, out.begin() + (size_t)(g + 1) * pw);
    } else if (rc == MNT753_ENODEV) {
      if (trace) fprintf(stderr, "mnt753: partial points folded on the host (%s)\n", mnt753_last_error());
    } else {
      fail("mnt753_exchange_points");
    }
  }
  bool have = false;
  for (auto& part : parts) {
    if (!have) { memcpy(p->data, part.data(), sizeof(uint64_t) * pw); have = true; }
    else check(mnt753_point_add(CURVE, GROUP, p->data, part.data(), p->data), "mnt753_point_add");
  }
  if (!have) {   // an empty MSM: the identity (0 : 1 : 0)
    uint64_t zero_aff[72] = {0};
    check(mnt753_point_from_affine(CURVE, GROUP, zero_aff, p->data), "mnt753_point_from_affine");
  }
  p->pending.reset();
}
template <int CURVE> static void evaluate_lazy(typename mnt753_hip_impl<CURVE>::G1* p);   // below, behind start_sharded
template <int CURVE> static void resolve(typename mnt753_hip_impl<CURVE>::G1* p) {
  if (p->lazy) evaluate_lazy<CURVE>(p);
  resolve_t<CURVE, MNT753_G1>(p);
}
template <int CURVE> static void resolve(typename mnt753_hip_impl<CURVE>::G2* p) { resolve_t<CURVE, MNT753_G2>(p); }

template <int CURVE> void HIP_B::use_devices(int n) { g_n_devices = n < 1 ? 1 : n; }
template <int CURVE> void HIP_B::init_public_params() {
  if (g_n_devices == 0) { const char* e = getenv("MNT753_GPUS"); g_n_devices = e && atoi(e) > 0 ? atoi(e) : 1; }
  if (g_n_devices > 1) {
    check(mnt753_init_devices(g_n_devices), "mnt753_init_devices");
    // every ordered pair the sharded prover copies between: cb / cc 1 -> 0 and 2 -> 0, slices of coefficients_for_H 0 -> g, operands
    // of the B:: vector calls between any two (operand_on).  The destination's copy engine reads the source's memory; without peer
    // access the copy stages through the host.  One line per pair under MNT753_TRACE=1.
    const char* e = getenv("MNT753_TRACE");
    const bool trace = e && atoi(e);
    static const char* const names[] = {"same GPU (logical devices share it)", "direct (peer access enabled)", "staged through host memory (no peer access)"};
    for (int a = 0; a < g_n_devices; ++a)
      for (int b = 0; b < g_n_devices; ++b) {
        if (a == b) continue;
        int how = MNT753_PEER_STAGED;
        check(mnt753_enable_peer_access(a, b, &how), "mnt753_enable_peer_access");
        if (trace) fprintf(stderr, "mnt753: device %d reads device %d: %s\n", a, b, names[how >= 0 && how <= 2 ? how : 2]);
      }
  } else check(mnt753_init(0), "mnt753_init");
  // Known answers of THIS build, once per process, before anything is proved with it (mnt753_self_test, include/mnt753_hip.h; the
  // reference's check of the same purpose: libsnark/main.cpp:295-343).  ~75 ms for this class's curve, inside the parameter-load phase, outside the timing
  // window (main.cpp:201-203).  A mismatch is fatal: a prover must not write proofs with arithmetic that fails its known answers.
  static bool self_tested[2] = {false, false};
  if (!self_tested[CURVE]) {
    int level = 1;
    if (const char* e = getenv("MNT753_SELFTEST")) level = atoi(e);
    if (level > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      check(mnt753_self_test_curve(CURVE, level > 2 ? 2 : level), "mnt753_self_test");
      if (trace_load_on()) fprintf(stderr, "mnt753: load params: %-46s %7.3f s\n", "known-answer self-test of this build",
                                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    self_tested[CURVE] = true;
  }
}

template <int CURVE> void HIP_B::print_G1(G1* a) {
  uint64_t aff[24];
  resolve<CURVE>(a);
  check(mnt753_point_to_affine(CURVE, MNT753_G1, a->data, aff), "mnt753_point_to_affine");
  printf("G1 affine (Montgomery limbs, little-endian):\n x =");
  for (int i = 11; i >= 0; --i) printf(" %016llx", (unsigned long long)aff[i]);
  printf("\n y =");
  for (int i = 11; i >= 0; --i) printf(" %016llx", (unsigned long long)aff[12 + i]);
  printf("\n");
}
template <int CURVE> void HIP_B::print_G2(G2* a) {
  const size_t w = mnt753_affine_words(CURVE, MNT753_G2);
  uint64_t aff[72];
  resolve<CURVE>(a);
  check(mnt753_point_to_affine(CURVE, MNT753_G2, a->data, aff), "mnt753_point_to_affine");
  printf("G2 affine (Montgomery limbs, little-endian), %zu coefficients:\n", w / 12);
  for (size_t k = 0; k < w / 12; ++k) {
    printf(" c%zu =", k);
    for (int i = 11; i >= 0; --i) printf(" %016llx", (unsigned long long)aff[12 * k + i]);
    printf("\n");
  }
}

// Domains (twiddle and coset tables, ~0.5 GB of HBM at 2^20) are cached per (curve, size): creating one allocates and
// frees device memory, which synchronises the whole device and would stall behind MSMs already in flight.  read_params
// creates the domain for d + 1 ahead of time -- it depends on the parameters only, like the MSM window tables.
template <int CURVE> static std::shared_ptr<DomainHolder> cached_domain(size_t d, int device = 0) {
  static std::mutex mu;
  static std::map<std::pair<int, size_t>, std::shared_ptr<DomainHolder>> cache;
  std::lock_guard<std::mutex> l(mu);
  auto it = cache.find({device, d});
  if (it != cache.end()) return it->second;
  auto h = std::make_shared<DomainHolder>();
  {
    DeviceScope on(device);   // a domain lives on the device that is current when it is created
    check(mnt753_domain_create(CURVE, d, &h->h), "mnt753_domain_create");
  }
  cache[{device, d}] = h;
  return h;
}
template <int CURVE> typename HIP_B::evaluation_domain* HIP_B::get_evaluation_domain(size_t d) {
  return new evaluation_domain{cached_domain<CURVE>(d), d};
}
// the tables of `domain` on the device a vector lives on (device 0's are in the handle; the others are built on first use --
// B::read_params builds the ones a sharded proof needs)
template <int CURVE> static mnt753_domain* domain_on(typename HIP_B::evaluation_domain* domain, int device) {
  return device == 0 ? domain->data->h : cached_domain<CURVE>(domain->m, device)->h;
}
// staged blocks stay alive as long as the vector whose pending kernels read them
static void keep_alive(std::shared_ptr<void>& slot, std::vector<std::shared_ptr<DeviceBuffer>>& bufs) {
  if (bufs.empty()) return;
  struct Chain { std::shared_ptr<void> prev; std::vector<std::shared_ptr<DeviceBuffer>> bufs; };
  auto c = std::make_shared<Chain>();
  c->prev = slot; c->bufs.swap(bufs);
  slot = c;
}
// `v` (n elements) where device `dev` can read it: the vector itself, or a staged copy brought over by an asynchronous peer copy
// (ordered behind everything enqueued on the source device's default stream); the staging block is appended to `keep`
template <class V> static const uint64_t* operand_on(V* v, int dev, size_t n, std::vector<std::shared_ptr<DeviceBuffer>>& keep) {
  if (v->device() == dev) return v->ptr();
  std::shared_ptr<DeviceBuffer> tmp;
  { DeviceScope on(dev); tmp = std::make_shared<DeviceBuffer>(96 * n); }
  check(mnt753_copy_peer_async(dev, tmp->ptr, v->device(), v->ptr(), 96 * n), "mnt753_copy_peer_async");
  keep.push_back(tmp);
  return reinterpret_cast<const uint64_t*>(tmp->ptr);
}

template <int CURVE> static std::shared_ptr<PendingMsm> try_fuse(const std::shared_ptr<LazyPoint>& root);   // below
// the expression node of a G1 operand: its tree if it has one, else its value
template <int CURVE> static std::shared_ptr<LazyPoint> node_of(typename mnt753_hip_impl<CURVE>::G1* a) {
  if (a->lazy) return a->lazy;
  resolve<CURVE>(a);
  auto n = std::make_shared<LazyPoint>();
  memcpy(n->value, a->data, sizeof(n->value));
  return n;
}
template <int CURVE> typename HIP_B::G1* HIP_B::G1_add(G1* a, G1* b) {
  G1* r = new G1();
  if (a->lazy || b->lazy) {
    auto n = std::make_shared<LazyPoint>();
    n->kind = LazyPoint::ADD;
    n->a = node_of<CURVE>(a); n->b = node_of<CURVE>(b);
    // Ht + (Lt + r Bt1) complete: the one MSM over H | L | B1 starts here
    if (auto pend = try_fuse<CURVE>(n)) r->pending = pend; else r->lazy = n;
    return r;
  }
  resolve<CURVE>(a); resolve<CURVE>(b);
  check(mnt753_point_add(CURVE, MNT753_G1, a->data, b->data, r->data), "mnt753_point_add");
  return r;
}
template <int CURVE> typename HIP_B::G1* HIP_B::G1_scale(field* a, G1* b) {
  G1* r = new G1();
  if (b->lazy) {
    auto n = std::make_shared<LazyPoint>();
    n->kind = LazyPoint::SCALE;
    memcpy(n->k, a->data, sizeof(n->k));
    n->a = b->lazy;
    r->lazy = n;
    return r;
  }
  resolve<CURVE>(b);
  check(mnt753_point_scale(CURVE, MNT753_G1, a->data, b->data, r->data), "mnt753_point_scale");
  return r;
}

// Element-wise operations run where their FIRST operand lives (the vector they overwrite); an operand on another device is brought
// over first (operand_on).  With one device this is the plain call.
template <int CURVE> void HIP_B::vector_Fr_muleq(vector_Fr* a, vector_Fr* b, size_t size) {
  std::vector<std::shared_ptr<DeviceBuffer>> keep;
  const uint64_t* bp = operand_on(b, a->device(), size, keep);
  DeviceScope on(a->device());
  check(mnt753_vec_muleq(CURVE, a->ptr(), bp, size, nullptr), "mnt753_vec_muleq");
  keep_alive(a->keep, keep);
}
template <int CURVE> void HIP_B::vector_Fr_subeq(vector_Fr* a, vector_Fr* b, size_t size) {
  std::vector<std::shared_ptr<DeviceBuffer>> keep;
  const uint64_t* bp = operand_on(b, a->device(), size, keep);
  DeviceScope on(a->device());
  check(mnt753_vec_subeq(CURVE, a->ptr(), bp, size, nullptr), "mnt753_vec_subeq");
  keep_alive(a->keep, keep);
}
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::vector_Fr_offset(vector_Fr* a, size_t offset) {
  return new vector_Fr{a->data, a->size, offset, a->ready, a->slices, a->keep};
}
template <int CURVE> void HIP_B::vector_Fr_copy_into(vector_Fr* src, vector_Fr* dst, size_t length) {
  // MNT4753: dst[i] = src[i] ignoring offsets (prover_reference_functions.cpp:209-212);
  // MNT6753: dst[i] = src[i + src->offset]            (:515-520)
  if (src->ready) src->ready->wait();
  if (dst->ready) dst->ready->wait();
  const uint64_t* s = reinterpret_cast<const uint64_t*>(src->data->ptr) + (CURVE == 1 ? 12 * src->offset : 0);
  if (src->device() == dst->device()) {
    DeviceScope on(dst->device());
    check(mnt753_copy_d2d(dst->data->ptr, s, 96 * length), "mnt753_copy_d2d");
  } else {
    check(mnt753_copy_peer_async(dst->device(), dst->data->ptr, src->device(), s, 96 * length), "mnt753_copy_peer_async");
  }
}
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::vector_Fr_zeros(size_t length) {
  auto b = std::make_shared<DeviceBuffer>(96 * length);
  DeviceScope on(b->device);
  check(mnt753_dev_memset(b->ptr, 0, 96 * length), "mnt753_dev_memset");
  return new vector_Fr{b, length, 0, nullptr, nullptr, nullptr};
}

// the transforms run on the device their vector lives on, with that device's tables
template <int CURVE> void HIP_B::domain_iFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain_on<CURVE>(domain, a->device()), MNT753_IFFT, a->ptr(), nullptr), "mnt753_fft(iFFT)");
}
template <int CURVE> void HIP_B::domain_cosetFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain_on<CURVE>(domain, a->device()), MNT753_COSET_FFT, a->ptr(), nullptr), "mnt753_fft(cosetFFT)");
}
template <int CURVE> void HIP_B::domain_icosetFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain_on<CURVE>(domain, a->device()), MNT753_ICOSET_FFT, a->ptr(), nullptr), "mnt753_fft(icosetFFT)");
}
template <int CURVE> void HIP_B::domain_divide_by_Z_on_coset(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_divide_by_z_on_coset(domain_on<CURVE>(domain, a->device()), a->ptr(), nullptr), "mnt753_divide_by_z_on_coset");
}
template <int CURVE> size_t HIP_B::domain_get_m(evaluation_domain* domain) { return mnt753_domain_size(domain->data->h); }

// sum_{i < length} scalars[i] * bases[i] over the slices of a sharded vector: slice g covers [lo_g, hi_g) of the bases and runs on
// device g, on that base set's own stream.  Where the scalars of a slice come from, in this order:
//   * a range of the vector that is resident on device g (w: DevSlice, streamed from the file by device g's own loader -- device 0's
//     range is released before the whole of w is there);
//   * the vector itself when device g is its home (waits for the input loader if the vector is still streaming in);
//   * otherwise (coefficients_for_H, computed on device 0): an asynchronous peer copy into a grow-only staging buffer on device g,
//     ordered behind the home device's default stream (compute_H) by an event and ahead of the MSM by the destination's default
//     stream -- the host never blocks (mnt753_copy_peer_async).
// The slices of the other devices are enqueued first: their inputs are ready first, and device 0 is the one that waits for the file.
struct ScalarSource {
  std::function<const uint64_t*()> home_ptr;   // element `offset` of the vector on its home device
  int home;                                    // logical device of that pointer
  const std::vector<DevSlice>* slices;         // ranges of the underlying buffer resident per device (index = device), or null
  size_t offset;                               // element offset of the logical vector inside the underlying buffer
};
// elements [lo, hi) of the source where device g can read them without a copy, or null
static const uint64_t* resident_on(const ScalarSource& src, int g, size_t lo, size_t hi) {
  const DevSlice* sl = src.slices && (size_t)g < src.slices->size() && (*src.slices)[(size_t)g].buf ? &(*src.slices)[(size_t)g] : nullptr;
  if (sl && src.offset + lo >= sl->first && src.offset + hi <= sl->first + sl->count) {
    sl->ready->wait();
    return reinterpret_cast<const uint64_t*>(sl->buf->ptr) + 12 * (src.offset + lo - sl->first);
  }
  if (g == src.home) return src.home_ptr() + 12 * lo;
  return nullptr;
}
// elements [lo, hi) of the source into `dst` on device g (the calling thread's current device): a local copy on g's default stream,
// or a peer copy ordered behind the home device's default stream
static void fetch_into(const ScalarSource& src, int g, size_t lo, size_t hi, uint64_t* dst) {
  if (hi <= lo) return;
  if (const uint64_t* p = resident_on(src, g, lo, hi)) check(mnt753_copy_d2d(dst, p, 96 * (hi - lo)), "mnt753_copy_d2d");
  else check(mnt753_copy_peer_async(g, dst, src.home, src.home_ptr() + 12 * lo, 96 * (hi - lo)), "mnt753_copy_peer_async");
}
// sets[g] was filled per device (null where a device got no points): keep the non-empty ones in rank order with their device numbers
static void compact_in_rank_order(PendingMsm& pm) {
  std::vector<std::shared_ptr<BaseSetHolder>> sets;
  pm.devs.clear();
  for (size_t g = 0; g < pm.sets.size(); ++g)
    if (pm.sets[g]) { sets.push_back(pm.sets[g]); pm.devs.push_back((int)g); }
  pm.sets.swap(sets);
}
// The MSM for C (three times the points of A's) is started first and has the longer latency-bound tail; the next MSM started on a
// device is ordered behind C's point kernels there (mnt753_msm_order_after): C's tail then runs under that MSM's point kernels instead
// of ending the prove beside it (g_gate, above).
// Measured (profiles/r04/prove_msm_order.txt, alternating on one box): MNT6753 2^15 15.7 -> 15.2 ms in one series, 15.3 -> 15.2 in a
// second; MNT4753 2^20 157.2 -> 158.4 ms
// -- there the two MSMs interleaved fill each other's kernel ends, which is worth more than a hidden 3 ms tail -- so only the small
// sets are ordered (C over at most 2^18 points on the device).
static void gate_set(int g, mnt753_bases* c) {
  if (mnt753_bases_size(c) > ((size_t)1 << 18)) return;
  if (g_gate.size() <= (size_t)g) g_gate.resize((size_t)g + 1, nullptr);
  g_gate[(size_t)g] = c;
}
static void gate_apply(int g, mnt753_bases* next) {
  if ((size_t)g >= g_gate.size() || !g_gate[(size_t)g]) return;
  if (g_gate[(size_t)g] != next) check(mnt753_msm_order_after(next, g_gate[(size_t)g]), "mnt753_msm_order_after");
  g_gate[(size_t)g] = nullptr;
}
static std::shared_ptr<PendingMsm> start_sharded(ShardedBases& sb, const ScalarSource& src, size_t length, const char* what, bool is_c = false) {
  auto pend = std::make_shared<PendingMsm>();
  const int n_dev = (int)sb.parts.size();
  pend->sets.resize((size_t)n_dev);
  for (int k = 0; k < n_dev; ++k) {
    const int g = k + 1 < n_dev ? k + 1 : 0;   // 1, 2, ..., n_dev - 1, 0
    ShardedBases::Part& part = sb.parts[(size_t)g];
    const size_t lo = part.lo, hi = std::min(part.hi, length);
    if (hi <= lo) continue;
    const uint64_t* sc = resident_on(src, g, lo, hi);
    if (!sc) {
      DeviceScope on(g);
      if (!part.scalars || part.scalars->bytes < 96 * (hi - lo)) part.scalars = std::make_shared<DeviceBuffer>(96 * (part.hi - part.lo));
      check(mnt753_copy_peer_async(g, part.scalars->ptr, src.home, src.home_ptr() + 12 * lo, 96 * (hi - lo)), "mnt753_copy_peer_async");
      sc = reinterpret_cast<const uint64_t*>(part.scalars->ptr);
    }
    if (!is_c) gate_apply(g, part.set->h);
    check(mnt753_msm_start(part.set->h, 0, sc, 1, hi - lo, nullptr), what);
    if (is_c) gate_set(g, part.set->h);
    pend->sets[(size_t)g] = part.set;
  }
  // rank order for the fold (multiexp.tcc:433-438), whatever the order of enqueueing was
  compact_in_rank_order(*pend);
  return pend;
}
template <class V> static ScalarSource source_of(V* v) {
  return ScalarSource{[v]() { return (const uint64_t*)v->ptr(); }, v->device(), v->slices.get(), v->offset};
}
// C = Ht + Lt + r Bt1 as ONE multi-scalar multiplication per device over the concatenated base set.
//   one device : the set is H | L | B1 and the scalars h | w_L | r w are assembled on the default stream (two copies and one scaling
//                pass, ~0.2 ms: behind compute_H, ahead of the MSM);
//   N devices  : part g of the set is H_g | L_g | B1_g (slice g of each vector).  Device g assembles ITS scalars on its own default
//                stream: h_g arrives from the device that ran compute_H (asynchronous peer copy behind it), w_L and w are read from the
//                range of w that device g streamed from the input file itself, r w_g is scaled there.  The partial points are folded in
//                rank order like those of any sharded MSM.
template <int CURVE>
static std::shared_ptr<PendingMsm> start_fused_c(typename HIP_B::groth16_params* p, const ScalarSource& h, const ScalarSource& w_L, const ScalarSource& w, const uint64_t* r) {
  const size_t d = p->d, m = p->m, n = d + 2 * m;
  ShardedBases& sb = *p->HLB;
  if (const char* e = getenv("MNT753_TRACE")) { if (atoi(e)) fprintf(stderr, "mnt753: C = Ht + Lt + r Bt1 as one MSM over H | L | B1 (%zu points, %zu device%s)\n", n, sb.parts.size(), sb.parts.size() > 1 ? "s" : ""); }
  if (!sb.interleaved) {
    auto sc = std::make_shared<DeviceBuffer>(96 * n);
    uint64_t* s = reinterpret_cast<uint64_t*>(sc->ptr);
    fetch_into(h, 0, 0, d, s);
    fetch_into(w_L, 0, 0, m - 1, s + 12 * d);
    const uint64_t* wp = resident_on(w, 0, 0, m + 1);
    if (!wp) { fetch_into(w, 0, 0, m + 1, s + 12 * (d + m - 1)); wp = s + 12 * (d + m - 1); }
    check(mnt753_vec_scale(CURVE, s + 12 * (d + m - 1), wp, r, m + 1, nullptr), "mnt753_vec_scale");
    auto pend = start_sharded(sb, ScalarSource{[s]() { return (const uint64_t*)s; }, 0, nullptr, 0}, n, "mnt753_msm_start(C)", true);
    pend->scalars = sc;
    return pend;
  }
  auto pend = std::make_shared<PendingMsm>();
  const int n_dev = (int)sb.parts.size();
  pend->sets.resize((size_t)n_dev);
  const ScalarSource* srcs[3] = {&h, &w_L, &w};
  for (int k = 0; k < n_dev; ++k) {
    const int g = k + 1 < n_dev ? k + 1 : 0;   // device 0 (the one compute_H ran on) last
    ShardedBases::Part& part = sb.parts[(size_t)g];
    const size_t total = part.hi - part.lo;
    if (total == 0) continue;
    DeviceScope on(g);
    if (!part.scalars || part.scalars->bytes < 96 * total) part.scalars = std::make_shared<DeviceBuffer>(96 * total);
    uint64_t* s = reinterpret_cast<uint64_t*>(part.scalars->ptr);
    size_t at = 0;
    for (int v = 0; v < 3; ++v) {
      const size_t lo = part.sub_lo[v], hi = part.sub_hi[v];
      if (v < 2) fetch_into(*srcs[v], g, lo, hi, s + 12 * at);
      else if (hi > lo) {
        const uint64_t* wp = resident_on(w, g, lo, hi);
        if (!wp) { fetch_into(w, g, lo, hi, s + 12 * at); wp = s + 12 * at; }
        check(mnt753_vec_scale(CURVE, s + 12 * at, wp, r, hi - lo, nullptr), "mnt753_vec_scale");
      }
      at += hi - lo;
    }
    check(mnt753_msm_start(part.set->h, 0, s, 1, total, nullptr), "mnt753_msm_start(C)");
    gate_set(g, part.set->h);
    pend->sets[(size_t)g] = part.set;
  }
  compact_in_rank_order(*pend);
  return pend;
}
// Does the tree say Ht + Lt + r Bt1 -- in any association and order: exactly one unstarted MSM over each of H and L (unscaled) and B1
// (scaled once), all of the same fused parameters and over the whole vectors?  Then start the one MSM; else null.
template <int CURVE> static std::shared_ptr<PendingMsm> try_fuse(const std::shared_ptr<LazyPoint>& root) {
  struct Term { const LazyPoint* msm; const uint64_t* k; };
  std::vector<Term> terms;
  std::function<bool(const LazyPoint*, const uint64_t*)> walk = [&](const LazyPoint* n, const uint64_t* k) -> bool {
    switch (n->kind) {
      case LazyPoint::ADD: return walk(n->a.get(), k) && walk(n->b.get(), k);
      case LazyPoint::SCALE: return k == nullptr && walk(n->a.get(), n->k);
      case LazyPoint::MSM: terms.push_back({n, k}); return terms.size() <= 3;
      default: return false;
    }
  };
  if (!walk(root.get(), nullptr) || terms.size() != 3) return nullptr;
  const LazyPoint* t[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint64_t* r = nullptr;
  for (const Term& term : terms) {
    const int which = term.msm->which;
    if (which < 1 || which > 3 || t[which] || term.msm->owner != terms[0].msm->owner) return nullptr;
    if ((which == 1) != (term.k != nullptr)) return nullptr;
    if (term.k) r = term.k;
    t[which] = term.msm;
  }
  auto* p = static_cast<typename HIP_B::groth16_params*>(terms[0].msm->owner);
  if (!p->HLB || t[1]->length != p->m + 1 || t[2]->length != p->m - 1 || t[3]->length != p->d) return nullptr;
  auto src = [](const LazyPoint* n) { return ScalarSource{n->scalars, n->home, n->slices.get(), n->offset}; };
  return start_fused_c<CURVE>(p, src(t[3]), src(t[2]), src(t[1]), r);
}
// the plain way: every MSM on its own base set (built now if the parameters were loaded fused), one after the other
template <int CURVE> static void lazy_value(LazyPoint& n, uint64_t* out) {
  switch (n.kind) {
    case LazyPoint::VALUE: memcpy(out, n.value, sizeof(n.value)); return;
    case LazyPoint::MSM: {
      auto* p = static_cast<typename HIP_B::groth16_params*>(n.owner);
      p->ensure_separate();
      ShardedBases& sb = n.which == 1 ? *p->B1 : (n.which == 2 ? *p->L : *p->H);
      typename HIP_B::G1 tmp;
      tmp.pending = start_sharded(sb, ScalarSource{n.scalars, n.home, n.slices.get(), n.offset}, n.length, "mnt753_msm_start(G1)");
      resolve_t<CURVE, MNT753_G1>(&tmp);
      memcpy(out, tmp.data, sizeof(tmp.data));
      return;
    }
    case LazyPoint::SCALE: {
      uint64_t a[36];
      lazy_value<CURVE>(*n.a, a);
      check(mnt753_point_scale(CURVE, MNT753_G1, n.k, a, out), "mnt753_point_scale");
      return;
    }
    case LazyPoint::ADD: {
      uint64_t a[36], b[36];
      lazy_value<CURVE>(*n.a, a); lazy_value<CURVE>(*n.b, b);
      check(mnt753_point_add(CURVE, MNT753_G1, a, b, out), "mnt753_point_add");
      return;
    }
  }
}
template <int CURVE> static void evaluate_lazy(typename mnt753_hip_impl<CURVE>::G1* p) {
  auto node = p->lazy;
  p->lazy.reset();
  if (auto pend = try_fuse<CURVE>(node)) { p->pending = pend; return; }
  lazy_value<CURVE>(*node, p->data);
}
template <int CURVE> typename HIP_B::G1* HIP_B::multiexp_G1(vector_Fr* scalar_start, vector_G1* g_start, size_t length) {
  G1* r = new G1();
  if (g_start->which && !g_start->data) {
    // B1, L or H of fused parameters: nothing runs yet (see LazyPoint) -- unless it is a partial vector, which only the set itself can do
    groth16_params* p = g_start->owner;
    const size_t full = g_start->which == 1 ? p->m + 1 : (g_start->which == 2 ? p->m - 1 : p->d);
    if (p->HLB && length == full && scalar_start->size - scalar_start->offset >= length) {
      auto n = std::make_shared<LazyPoint>();
      n->kind = LazyPoint::MSM;
      n->owner = p; n->which = g_start->which; n->length = length;
      auto v = std::make_shared<vector_Fr>(*scalar_start);   // shares the buffer and its readiness latch
      n->keep = v;
      vector_Fr* vp = v.get();
      n->scalars = [vp]() { return (const uint64_t*)vp->ptr(); };
      n->home = scalar_start->device();
      n->slices = scalar_start->slices;
      n->offset = scalar_start->offset;
      r->lazy = n;
      return r;
    }
    p->ensure_separate();
    g_start->data = g_start->which == 1 ? p->B1 : (g_start->which == 2 ? p->L : p->H);
  }
  r->pending = start_sharded(*g_start->data, source_of(scalar_start), length, "mnt753_msm_start(G1)");
  return r;
}
template <int CURVE> typename HIP_B::G2* HIP_B::multiexp_G2(vector_Fr* scalar_start, vector_G2* g_start, size_t length) {
  G2* r = new G2();
  r->pending = start_sharded(*g_start->data, source_of(scalar_start), length, "mnt753_msm_start(G2)");
  return r;
}

// C = Ht + Lt + r Bt1 (cuda_prover_piecewise.cu:79-90) = sum_i h[i] H[i] + sum_i w_L[i] L[i] + sum_i (r w[i]) B1[i]: one multi-scalar
// multiplication of d + 2m points over the concatenated set instead of three of ~m points each plus a scalar multiplication and two
// additions.  Same group element, so the same bytes in the proof file; what it saves is everything an MSM pays per call and per
// bucket rather than per point (sort, edge merge, bucket reduction: 3.2 of 25.8 ms at 2^20 points) -- one 3 * 2^20-point MSM
// instead of three 2^20-point ones.  The scalar vector is assembled on device 0's default stream (two copies and one scaling pass,
// ~0.2 ms), behind compute_H and ahead of the MSM.
template <int CURVE> typename HIP_B::G1* HIP_B::groth16_C(groth16_params* p, vector_Fr* coefficients_for_H, vector_Fr* w_L, vector_Fr* w, field* r) {
  if (!p->HLB) throw std::runtime_error("groth16_C: the parameters were loaded without the concatenated base set (B::fuse_C(false) / MNT753_FUSED_C=0)");
  const size_t d = p->d, m = p->m;
  if (coefficients_for_H->size - coefficients_for_H->offset < d || w_L->size - w_L->offset < m - 1 || w->size - w->offset < m + 1)
    throw std::runtime_error("groth16_C: a scalar vector is shorter than its base vector");
  G1* out = new G1();
  out->pending = start_fused_c<CURVE>(p, source_of(coefficients_for_H), source_of(w_L), source_of(w), r->data);
  return out;
}
template <int CURVE> void HIP_B::fuse_C(bool on) { g_fused_c = on ? 1 : 0; }
template <int CURVE> void HIP_B::fold_over_rccl(bool on) { g_fold_rccl = on ? 1 : 0; }
template <int CURVE> void HIP_B::one_shot(bool on) { g_one_shot = on ? 1 : 0; }

template <int CURVE> typename HIP_B::groth16_input* HIP_B::read_input(const char* path, groth16_params* params) {
  return new groth16_input(path, params->d, params->m);
}
template <int CURVE> typename HIP_B::r1cs* HIP_B::read_r1cs(const char* path) { return new r1cs(path); }
template <int CURVE> typename HIP_B::groth16_input* HIP_B::read_witness(const char* path, groth16_params* params, r1cs* cs) {
  return new groth16_input(path, params->d, params->m, cs->data);
}
template <int CURVE> void HIP_B::delete_r1cs(r1cs* a) { delete a; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_w(groth16_input* in) { return new vector_Fr{in->w, in->n_w, 0, in->w_ready, in->w_slices, nullptr}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_ca(groth16_input* in) { return new vector_Fr{in->ca, in->n_c, 0, in->ca_ready, nullptr, nullptr}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_cb(groth16_input* in) { return new vector_Fr{in->cb, in->n_c, 0, in->cb_ready, nullptr, nullptr}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_cc(groth16_input* in) { return new vector_Fr{in->cc, in->n_c, 0, in->cc_ready, nullptr, nullptr}; }
template <int CURVE> typename HIP_B::field* HIP_B::input_r(groth16_input* in) {
  field* f = new field();
  memcpy(f->data, in->r, 96);
  return f;
}

// One MSM per base set over uniform scalars, at parameter-load time (the reference's timing window opens after the parameters
// are loaded, libsnark/main.cpp:201-203).  It touches every workspace page, loads every kernel and settles the allocator, so
// that the first proof costs what every later one does: without it the first proof of a process was measured at 0.23 s on a
// fresh device but 0.7-1.7 s when the device memory had just been used by another process (bench.py's parent, a previous
// prover), the later ones always at 0.22 s.  MNT753_NO_WARMUP=1 turns it off.
template <int CURVE> static void warm_up(typename mnt753_hip_impl<CURVE>::groth16_params* p) {
  if (const char* e = getenv("MNT753_NO_WARMUP")) { if (atoi(e) != 0) return; }
  if (one_shot_mode()) return;    // one proof: what the warm-up would pay at load time the proof pays itself, once
  std::vector<ShardedBases*> sets;
  for (auto* sb : {p->B2.get(), p->HLB.get(), p->A.get(), p->B1.get(), p->L.get(), p->H.get()}) if (sb) sets.push_back(sb);
  size_t n = 1;
  for (auto* sb : sets) n = std::max(n, sb->n);
  std::vector<uint64_t> host(12 * n);
  check(mnt753_synth_scalars(CURVE, 0x7761726dull, n, host.data()), "mnt753_synth_scalars");
  DeviceBuffer dev(96 * n);
  check(mnt753_copy_h2d(dev.ptr, host.data(), 96 * n), "mnt753_copy_h2d");
  std::vector<std::shared_ptr<PendingMsm>> pend;
  for (auto* sb : sets)
    pend.push_back(start_sharded(*sb, ScalarSource{[&dev]() { return reinterpret_cast<const uint64_t*>(dev.ptr); }, 0, nullptr, 0}, sb->n, "mnt753_msm_start(warm-up)"));
  uint64_t sink[108];
  for (auto& pm : pend)
    for (auto& set : pm->sets) check(mnt753_msm_finish(set->h, sink), "mnt753_msm_finish(warm-up)");
}
template <int CURVE> typename HIP_B::groth16_params* HIP_B::read_params(const char* path) {
  size_t mem_free0 = 0, mem_total = 0;
  if (trace_load_on()) (void)mnt753_dev_mem_info(&mem_free0, &mem_total);
  // a one-proof process builds its base sets without window tables (include/mnt753_hip.h, mnt753_msm_set_window_table)
  struct TableMode {
    int old;
    explicit TableMode(bool none) : old(mnt753_msm_set_window_table(none ? 0 : 1)) { if (trace_load_on() && none) fprintf(stderr, "mnt753: one-shot prover: no window tables, no warm-up MSM\n"); }
    ~TableMode() { (void)mnt753_msm_set_window_table(old); }
  } table_mode(one_shot_mode());
  groth16_params* p = new groth16_params(path);
  const int n_dev = std::max(1, mnt753_device_count());
  LoadTrace lt;
  auto dom = cached_domain<CURVE>(p->d + 1);
  // a sharded prover transforms cb on device 1 and cc on device 2 (groth16_input): their tables are built now, like device 0's
  for (int g = 1; g < std::min(n_dev, 3); ++g) (void)cached_domain<CURVE>(p->d + 1, g);
  lt.lap("evaluation domain(s): twiddle and coset tables");
  warm_up<CURVE>(p);
  lt.lap("warm-up MSM per base set");
  if (trace_load_on()) {
    size_t mem_free1 = 0;
    if (mnt753_dev_mem_info(&mem_free1, &mem_total) == 0 && mem_free0 >= mem_free1)
      fprintf(stderr, "mnt753: load params: device memory of this parameter set (device 0): %.1f GB (tables, workspaces, pooled level buffers)\n", (mem_free0 - mem_free1) / 1e9);
  }
  // the buffers of one proof, allocated now and parked in the cache: w, ca, cb, cc, coefficients_for_H, the scalars of groth16_C, and
  // with several devices each device's range of w, its vector of compute_H and the staging of the transformed cb / cc on device 0
  {
    const size_t m_dom = mnt753_domain_size(dom->h);
    std::vector<std::unique_ptr<DeviceBuffer>> pre;
    struct BackToDevice0 { int n; ~BackToDevice0() { if (n > 1) (void)mnt753_set_device(0); } } back{n_dev};
    pre.emplace_back(new DeviceBuffer(96 * (p->m + 1)));
    pre.emplace_back(new DeviceBuffer(96 * (m_dom + 1)));
    if (n_dev == 1) {
      for (int k = 0; k < 3; ++k) pre.emplace_back(new DeviceBuffer(96 * (p->d + 1)));
      if (p->HLB) pre.emplace_back(new DeviceBuffer(96 * (p->d + 2 * p->m)));   // the scalars of groth16_C
    } else {
      pre.emplace_back(new DeviceBuffer(96 * (p->d + 1)));                                  // ca
      for (int k = 0; k < 2; ++k) pre.emplace_back(new DeviceBuffer(96 * m_dom));           // staged cb, cc
      if (n_dev == 2) pre.emplace_back(new DeviceBuffer(96 * (p->d + 1)));                  // cc stays on device 0
      for (int g = 1; g < n_dev; ++g) {
        check(mnt753_set_device(g), "mnt753_set_device");
        size_t first, count;
        groth16_input::w_range(p->m, n_dev, g, &first, &count);
        pre.emplace_back(new DeviceBuffer(96 * count));
        if (g <= 2) pre.emplace_back(new DeviceBuffer(96 * (p->d + 1)));                    // cb / cc
      }
    }
  }
  return p;
}
template <int CURVE> size_t HIP_B::params_d(groth16_params* p) { return p->d; }
template <int CURVE> size_t HIP_B::params_m(groth16_params* p) { return p->m; }
template <int CURVE> typename HIP_B::vector_G1* HIP_B::params_A(groth16_params* p) { return new vector_G1{p->A}; }
// B1, L, H of fused parameters: a handle only; the base set is built when a multiexp really needs it (multiexp_G1)
template <int CURVE> typename HIP_B::vector_G1* HIP_B::params_B1(groth16_params* p) { return new vector_G1{p->B1, p, 1}; }
template <int CURVE> typename HIP_B::vector_G1* HIP_B::params_L(groth16_params* p) { return new vector_G1{p->L, p, 2}; }
template <int CURVE> typename HIP_B::vector_G1* HIP_B::params_H(groth16_params* p) { return new vector_G1{p->H, p, 3}; }
template <int CURVE> typename HIP_B::vector_G2* HIP_B::params_B2(groth16_params* p) { return new vector_G2{p->B2}; }

template <int CURVE> void HIP_B::delete_G1(G1* a) { if (a && !a->lazy) resolve<CURVE>(a); delete a; }   // an MSM in flight is waited for; one never started is dropped
template <int CURVE> void HIP_B::delete_G2(G2* a) { if (a) resolve<CURVE>(a); delete a; }
template <int CURVE> void HIP_B::delete_field(field* a) { delete a; }
template <int CURVE> void HIP_B::delete_vector_Fr(vector_Fr* a) { delete a; }
template <int CURVE> void HIP_B::delete_vector_G1(vector_G1* a) { delete a; }
template <int CURVE> void HIP_B::delete_vector_G2(vector_G2* a) { delete a; }
template <int CURVE> void HIP_B::delete_groth16_input(groth16_input* a) { delete a; }
template <int CURVE> void HIP_B::delete_groth16_params(groth16_params* a) { delete a; g_buffers.release_all(); }
template <int CURVE> void HIP_B::delete_evaluation_domain(evaluation_domain* a) { delete a; }

// write_g1(A) write_g2(B) write_g1(C)   (prover_reference_functions.cpp:347-356, serialization.hpp:44-67)
template <int CURVE> void HIP_B::groth16_output_write(G1* A, G2* B, G1* C, const char* output_path) {
  const size_t g2w = mnt753_affine_words(CURVE, MNT753_G2);
  uint64_t a[24], b[72], c[24];
  resolve<CURVE>(A); resolve<CURVE>(B); resolve<CURVE>(C);
  check(mnt753_point_to_affine(CURVE, MNT753_G1, A->data, a), "mnt753_point_to_affine(A)");
  check(mnt753_point_to_affine(CURVE, MNT753_G2, B->data, b), "mnt753_point_to_affine(B)");
  check(mnt753_point_to_affine(CURVE, MNT753_G1, C->data, c), "mnt753_point_to_affine(C)");
  FILE* out = fopen(output_path, "wb");
  if (!out) throw std::runtime_error(std::string("cannot open output file ") + output_path);
  fwrite(a, 8, 24, out);
  fwrite(b, 8, g2w, out);
  fwrite(c, 8, 24, out);
  fclose(out);
}

// compute_H<B> (cuda_prover_piecewise.cu:18-53) in device-resident calls.  All three vectors on one device: one call.  A sharded
// prover keeps ca, cb, cc on three devices (groth16_input): each runs x <- cosetFFT(iFFT(x)) where its vector lives
// (mnt753_compute_h_chain), the transformed cb and cc travel to ca's device (asynchronous peer copies, ordered behind the chains by
// events), and the pointwise step, the last transform and coefficients_for_H happen there (mnt753_compute_h_finish).
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::compute_H_fused(evaluation_domain* domain, vector_Fr* ca, vector_Fr* cb, vector_Fr* cc) {
  const size_t m = mnt753_domain_size(domain->data->h);
  const int da = ca->device(), db = cb->device(), dc = cc->device();
  std::shared_ptr<DeviceBuffer> h;
  { DeviceScope on(da); h = std::make_shared<DeviceBuffer>(96 * (m + 1)); }
  vector_Fr* out = new vector_Fr{h, m + 1, 0, nullptr, nullptr, nullptr};
  try {
    if (da == db && da == dc) {
      check(mnt753_compute_h(domain_on<CURVE>(domain, da), ca->ptr(), cb->ptr(), cc->ptr(), reinterpret_cast<uint64_t*>(h->ptr), nullptr), "mnt753_compute_h");
      return out;
    }
    // the chains of the other devices first: device da is the one the rest of the proof waits for
    if (db != da) check(mnt753_compute_h_chain(domain_on<CURVE>(domain, db), cb->ptr(), nullptr), "mnt753_compute_h_chain(cb)");
    if (dc != da) check(mnt753_compute_h_chain(domain_on<CURVE>(domain, dc), cc->ptr(), nullptr), "mnt753_compute_h_chain(cc)");
    check(mnt753_compute_h_chain(domain_on<CURVE>(domain, da), ca->ptr(), nullptr), "mnt753_compute_h_chain(ca)");
    if (db == da) check(mnt753_compute_h_chain(domain_on<CURVE>(domain, da), cb->ptr(), nullptr), "mnt753_compute_h_chain(cb)");
    if (dc == da) check(mnt753_compute_h_chain(domain_on<CURVE>(domain, da), cc->ptr(), nullptr), "mnt753_compute_h_chain(cc)");
    std::vector<std::shared_ptr<DeviceBuffer>> keep;
    const uint64_t* bp = operand_on(cb, da, m, keep);
    const uint64_t* cp = operand_on(cc, da, m, keep);
    check(mnt753_compute_h_finish(domain_on<CURVE>(domain, da), ca->ptr(), bp, cp, reinterpret_cast<uint64_t*>(h->ptr), nullptr), "mnt753_compute_h_finish");
    keep_alive(out->keep, keep);
    if (const char* e = getenv("MNT753_TRACE")) { if (atoi(e)) fprintf(stderr, "mnt753: compute_H over devices %d / %d / %d (chains where ca / cb / cc live, finish on %d)\n", da, db, dc, da); }
  } catch (...) { delete out; throw; }
  return out;
}
template <int CURVE> double HIP_B::input_load_seconds(groth16_input* in) {
  in->w_ready->wait(); in->ca_ready->wait(); in->cb_ready->wait(); in->cc_ready->wait();
  if (in->w_slices) for (auto& sl : *in->w_slices) if (sl.ready) sl.ready->wait();
  std::lock_guard<std::mutex> l(in->clock->mu);
  return in->clock->secs;
}
template <int CURVE> const uint64_t* HIP_B::G1_words(const G1* a) { resolve<CURVE>(const_cast<G1*>(a)); return a->data; }
template <int CURVE> const uint64_t* HIP_B::G2_words(const G2* a) { resolve<CURVE>(const_cast<G2*>(a)); return a->data; }

template class mnt753_hip_impl<0>;
template class mnt753_hip_impl<1>;
