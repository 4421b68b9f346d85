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
struct BaseSetHolder {
  mnt753_bases* h = nullptr;
  ~BaseSetHolder() { if (h) mnt753_bases_free(h); }
};
static int g_n_devices = 0;   // 0: not chosen yet (init_public_params reads MNT753_GPUS, default 1)
// Ht + Lt + r Bt1 as ONE multi-scalar multiplication over the concatenated base set H | L | B1 (B::groth16_C).  -1: not chosen yet
// (read_params reads MNT753_FUSED_C, default on).
static int g_fused_c = -1;
static bool fused_c() {
  if (g_fused_c < 0) { const char* e = getenv("MNT753_FUSED_C"); g_fused_c = e ? (atoi(e) != 0) : 1; }
  return g_fused_c != 0;
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
    std::shared_ptr<DeviceBuffer> scalars;   // staging for the scalar slice on devices other than 0 (grow-only)
  };
  std::vector<Part> parts;
  size_t n = 0;
};
// one MSM in flight on every slice of a sharded vector
struct PendingMsm {
  std::vector<std::shared_ptr<BaseSetHolder>> sets;
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

template <int CURVE> struct mnt753_hip_impl<CURVE>::evaluation_domain { std::shared_ptr<DomainHolder> data; };
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
  // device pointer; waits (once) until the loader thread has put the vector on the device
  uint64_t* ptr() const {
    if (ready) ready->wait();
    return reinterpret_cast<uint64_t*>(data->ptr) + 12 * offset;
  }
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
    auto load = [&](int group, size_t words, size_t n, size_t cat_at) {
      host.resize(words * n);
      read_exact(f, host.data(), host.size() * 8, path_);
      if (fused && group == MNT753_G1 && cat_at != (size_t)-1) {   // a part of the concatenated set: keep the host copy, no set of its own
        memcpy(cat.data() + g1w * cat_at, host.data(), host.size() * 8);
        return std::shared_ptr<ShardedBases>();
      }
      return make_set(group, words, n, host.data());
    };
    if (fused) cat.resize(g1w * (d + 2 * m));
    try {
      A = load(MNT753_G1, g1w, m + 1, (size_t)-1);
      B1 = load(MNT753_G1, g1w, m + 1, d + m - 1);
      B2 = load(MNT753_G2, g2w, m + 1, (size_t)-1);
      L = load(MNT753_G1, g1w, m - 1, d);
      H = load(MNT753_G1, g1w, d, 0);
      if (fused) HLB = make_set(MNT753_G1, g1w, d + 2 * m, cat.data());
    } catch (...) { fclose(f); throw; }
    fclose(f);
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
// The constructor returns at once; a loader thread streams the four vectors to the device in file order and releases
// them one by one, so kernels that only need w (the G2 MSM and A's) start while ca / cb / cc are still being read.
template <int CURVE>
class mnt753_hip_impl<CURVE>::groth16_input {
public:
  std::shared_ptr<DeviceBuffer> w, ca, cb, cc;
  std::shared_ptr<Ready> w_ready, ca_ready, cb_ready, cc_ready, r_ready;
  size_t n_w = 0, n_c = 0;
  uint64_t r[12];
  std::thread loader;
  std::shared_ptr<double> load_seconds = std::make_shared<double>(0.0);
  // several devices: device g > 0 streams the part of w its slices of A / B1 / B2 (w[i]) and L (w[2 + i]) multiply, from the file,
  // on its own staging buffers and PCIe link, while device 0 reads w, ca, cb, cc -- nothing is funnelled through device 0
  std::shared_ptr<std::vector<DevSlice>> w_slices;
  std::vector<std::thread> slice_loaders;
  void start_slice_loaders(const std::string& path, size_t m) {
    const int n_dev = std::max(1, mnt753_device_count());
    if (n_dev < 2) return;
    w_slices = std::make_shared<std::vector<DevSlice>>((size_t)n_dev);
    struct BackToDevice0 { ~BackToDevice0() { (void)mnt753_set_device(0); } } back;   // also when an allocation throws
    for (int g = 1; g < n_dev; ++g) {
      size_t lo_a, hi_a, lo_l, hi_l;
      slice_bounds(m + 1, n_dev, g, &lo_a, &hi_a);   // A, B1, B2: scalars w[lo .. hi)
      slice_bounds(m - 1, n_dev, g, &lo_l, &hi_l);   // L: scalars w[2 + lo .. 2 + hi)  (vector_Fr_offset(w, primary_input_size + 1))
      DevSlice& sl = (*w_slices)[(size_t)g];
      sl.first = std::min(lo_a, lo_l + 2);
      sl.count = std::max(hi_a, hi_l + 2) - sl.first;
      sl.ready = std::make_shared<Ready>();
      check(mnt753_set_device(g), "mnt753_set_device");
      sl.buf = std::make_shared<DeviceBuffer>(96 * sl.count);
    }
    check(mnt753_set_device(0), "mnt753_set_device");
    for (int g = 1; g < n_dev; ++g) {
      DevSlice sl = (*w_slices)[(size_t)g];
      slice_loaders.emplace_back([path, sl, g]() {
        std::string err;
        if (mnt753_set_device(g) != 0 || mnt753_load_file_to_device(path.c_str(), 96 * sl.first, 96 * sl.count, sl.buf->ptr) != 0)
          err = std::string("mnt753_load_file_to_device (device ") + std::to_string(g) + "): " + mnt753_last_error();
        sl.ready->set(err);
      });
    }
  }
  groth16_input(const char* path, size_t d, size_t m) {
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open input file ") + path);
    n_w = m + 1; n_c = d + 1;
    const size_t r_off = 96 * (n_w + 3 * n_c);
    {
      struct stat st;
      if (stat(path, &st) != 0 || (unsigned long long)st.st_size != (unsigned long long)r_off + 96ull) {
        fclose(f);
        throw std::runtime_error(std::string("input file size does not match the parameters (expected ") + std::to_string(r_off + 96) + " bytes): " + path);
      }
    }
    if (fseeko(f, (off_t)r_off, SEEK_SET) != 0 || fread(r, 1, 96, f) != 96) { fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
    fclose(f);
    w = std::make_shared<DeviceBuffer>(96 * n_w);
    ca = std::make_shared<DeviceBuffer>(96 * n_c);
    cb = std::make_shared<DeviceBuffer>(96 * n_c);
    cc = std::make_shared<DeviceBuffer>(96 * n_c);
    w_ready = std::make_shared<Ready>(); ca_ready = std::make_shared<Ready>(); cb_ready = std::make_shared<Ready>();
    cc_ready = std::make_shared<Ready>();
    const std::string p(path);
    start_slice_loaders(p, m);
    struct Part { void* dst; size_t off, bytes; std::shared_ptr<Ready> ready; };
    std::vector<Part> parts = {{w->ptr, 0, 96 * n_w, w_ready}, {ca->ptr, 96 * n_w, 96 * n_c, ca_ready},
                               {cb->ptr, 96 * (n_w + n_c), 96 * n_c, cb_ready}, {cc->ptr, 96 * (n_w + 2 * n_c), 96 * n_c, cc_ready}};
    auto secs_out = load_seconds;
    loader = std::thread([p, parts, secs_out]() {
      const auto t0 = std::chrono::steady_clock::now();
      std::string err;
      for (size_t k = 0; k < parts.size(); ++k) {
        const Part& part = parts[k];
        if (err.empty() && mnt753_load_file_to_device(p.c_str(), part.off, part.bytes, part.dst) != 0)
          err = std::string("mnt753_load_file_to_device: ") + mnt753_last_error();
        if (k + 1 == parts.size()) *secs_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        part.ready->set(err);
      }
    });
  }
  // witness file: w[m+1], r.  ca / cb / cc are evaluated on the device from the constraint system as soon as w is there.
  groth16_input(const char* path, size_t d, size_t m, std::shared_ptr<R1csHolder> cs) {
    n_w = m + 1; n_c = d + 1;
    struct stat st;
    if (stat(path, &st) != 0 || (unsigned long long)st.st_size != 96ull * (n_w + 1))
      throw std::runtime_error(std::string("witness file size does not match the parameters (expected ") + std::to_string(96 * (n_w + 1)) + " bytes): " + path);
    if (mnt753_r1cs_domain_size(cs->h) > n_c) throw std::runtime_error("the constraint system does not fit the parameters' evaluation domain");
    // the evaluation kernel gathers w[col] for col <= cs.m and copies w[0 .. num_inputs]: both must stay inside the m + 1 elements of w
    if (mnt753_r1cs_num_variables(cs->h) != m || mnt753_r1cs_num_inputs(cs->h) > m)
      throw std::runtime_error("the constraint system has " + std::to_string(mnt753_r1cs_num_variables(cs->h)) + " variables and " +
                               std::to_string(mnt753_r1cs_num_inputs(cs->h)) + " inputs, the parameters are for m = " + std::to_string(m));
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open witness file ") + path);
    if (fseeko(f, (off_t)(96 * n_w), SEEK_SET) != 0 || fread(r, 1, 96, f) != 96) { fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
    fclose(f);
    w = std::make_shared<DeviceBuffer>(96 * n_w);
    ca = std::make_shared<DeviceBuffer>(96 * n_c);
    cb = std::make_shared<DeviceBuffer>(96 * n_c);
    cc = std::make_shared<DeviceBuffer>(96 * n_c);
    w_ready = std::make_shared<Ready>(); ca_ready = std::make_shared<Ready>(); cb_ready = std::make_shared<Ready>();
    cc_ready = std::make_shared<Ready>();
    const std::string p(path);
    start_slice_loaders(p, m);
    auto secs_out = load_seconds;
    auto w_ = w, ca_ = ca, cb_ = cb, cc_ = cc;
    auto wr = w_ready, ar = ca_ready, br = cb_ready, cr = cc_ready;
    const size_t nw = n_w, ncc = n_c;
    loader = std::thread([p, secs_out, w_, ca_, cb_, cc_, wr, ar, br, cr, nw, ncc, cs]() {
      const auto t0 = std::chrono::steady_clock::now();
      std::string err;
      if (mnt753_load_file_to_device(p.c_str(), 0, 96 * nw, w_->ptr) != 0) err = std::string("mnt753_load_file_to_device: ") + mnt753_last_error();
      wr->set(err);
      if (err.empty() && (mnt753_r1cs_evaluate(cs->h, reinterpret_cast<const uint64_t*>(w_->ptr), reinterpret_cast<uint64_t*>(ca_->ptr),
                                               reinterpret_cast<uint64_t*>(cb_->ptr), reinterpret_cast<uint64_t*>(cc_->ptr), ncc, nullptr) != 0 ||
                          mnt753_sync(nullptr) != 0))
        err = std::string("mnt753_r1cs_evaluate: ") + mnt753_last_error();
      *secs_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      ar->set(err); br->set(err); cr->set(err);
    });
  }
  ~groth16_input() {
    if (loader.joinable()) loader.join();
    for (auto& t : slice_loaders) if (t.joinable()) t.join();
  }
};

#define HIP_B mnt753_hip_impl<CURVE>

// collect the partial results of the slices and fold them in rank order (multiexp.tcc:433-438: final = final + partial[i])
template <int CURVE, int GROUP, class P> static void resolve_t(P* p) {
  if (!p->pending) return;
  bool have = false;
  for (auto& set : p->pending->sets) {
    uint64_t part[108];
    check(mnt753_msm_finish(set->h, part), "mnt753_msm_finish");
    if (!have) { memcpy(p->data, part, sizeof(uint64_t) * mnt753_projective_words(CURVE, GROUP)); have = true; }
    else check(mnt753_point_add(CURVE, GROUP, p->data, part, p->data), "mnt753_point_add");
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
  if (g_n_devices > 1) check(mnt753_init_devices(g_n_devices), "mnt753_init_devices");
  else check(mnt753_init(0), "mnt753_init");
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
template <int CURVE> static std::shared_ptr<DomainHolder> cached_domain(size_t d) {
  static std::mutex mu;
  static std::map<size_t, std::shared_ptr<DomainHolder>> cache;
  std::lock_guard<std::mutex> l(mu);
  auto it = cache.find(d);
  if (it != cache.end()) return it->second;
  auto h = std::make_shared<DomainHolder>();
  check(mnt753_domain_create(CURVE, d, &h->h), "mnt753_domain_create");
  cache[d] = h;
  return h;
}
template <int CURVE> typename HIP_B::evaluation_domain* HIP_B::get_evaluation_domain(size_t d) {
  return new evaluation_domain{cached_domain<CURVE>(d)};
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

template <int CURVE> void HIP_B::vector_Fr_muleq(vector_Fr* a, vector_Fr* b, size_t size) {
  check(mnt753_vec_muleq(CURVE, a->ptr(), b->ptr(), size, nullptr), "mnt753_vec_muleq");
}
template <int CURVE> void HIP_B::vector_Fr_subeq(vector_Fr* a, vector_Fr* b, size_t size) {
  check(mnt753_vec_subeq(CURVE, a->ptr(), b->ptr(), size, nullptr), "mnt753_vec_subeq");
}
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::vector_Fr_offset(vector_Fr* a, size_t offset) {
  return new vector_Fr{a->data, a->size, offset, a->ready, a->slices};
}
template <int CURVE> void HIP_B::vector_Fr_copy_into(vector_Fr* src, vector_Fr* dst, size_t length) {
  // MNT4753: dst[i] = src[i] ignoring offsets (prover_reference_functions.cpp:209-212);
  // MNT6753: dst[i] = src[i + src->offset]            (:515-520)
  if (src->ready) src->ready->wait();
  if (dst->ready) dst->ready->wait();
  const uint64_t* s = reinterpret_cast<const uint64_t*>(src->data->ptr) + (CURVE == 1 ? 12 * src->offset : 0);
  check(mnt753_copy_d2d(dst->data->ptr, s, 96 * length), "mnt753_copy_d2d");
}
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::vector_Fr_zeros(size_t length) {
  auto b = std::make_shared<DeviceBuffer>(96 * length);
  check(mnt753_dev_memset(b->ptr, 0, 96 * length), "mnt753_dev_memset");
  return new vector_Fr{b, length, 0, nullptr, nullptr};
}

template <int CURVE> void HIP_B::domain_iFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain->data->h, MNT753_IFFT, a->ptr(), nullptr), "mnt753_fft(iFFT)");
}
template <int CURVE> void HIP_B::domain_cosetFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain->data->h, MNT753_COSET_FFT, a->ptr(), nullptr), "mnt753_fft(cosetFFT)");
}
template <int CURVE> void HIP_B::domain_icosetFFT(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_fft(domain->data->h, MNT753_ICOSET_FFT, a->ptr(), nullptr), "mnt753_fft(icosetFFT)");
}
template <int CURVE> void HIP_B::domain_divide_by_Z_on_coset(evaluation_domain* domain, vector_Fr* a) {
  check(mnt753_divide_by_z_on_coset(domain->data->h, a->ptr(), nullptr), "mnt753_divide_by_z_on_coset");
}
template <int CURVE> size_t HIP_B::domain_get_m(evaluation_domain* domain) { return mnt753_domain_size(domain->data->h); }

// sum_{i < length} scalars[i] * bases[i] over the slices of a sharded vector: slice g covers [lo_g, hi_g) of the bases and runs on
// device g, on that base set's own stream.  Where the scalars come from, per device:
//   * device 0: the vector itself (dev0(), which waits for the input loader if the vector is still streaming in);
//   * device g > 0, the vector has a resident range there (w: DevSlice, loaded from the file by device g's own loader): that;
//   * otherwise (coefficients_for_H, computed on device 0): an asynchronous peer copy into a grow-only staging buffer on device g,
//     ordered behind device 0's default stream (compute_H) by an event and ahead of the MSM by the destination's default stream --
//     the host never blocks (mnt753_copy_peer_async).
// The slices of the other devices are enqueued first: their inputs are ready first, and device 0 is the one that waits for the file.
struct ScalarSource {
  std::function<const uint64_t*()> dev0;       // element `offset` of the vector on device 0
  const std::vector<DevSlice>* slices;         // ranges of the underlying buffer on the other devices, or null
  size_t offset;                               // element offset of the logical vector inside the underlying buffer
};
static std::shared_ptr<PendingMsm> start_sharded(ShardedBases& sb, const ScalarSource& src, size_t length, const char* what) {
  auto pend = std::make_shared<PendingMsm>();
  const int n_dev = (int)sb.parts.size();
  pend->sets.resize((size_t)n_dev);
  for (int k = 0; k < n_dev; ++k) {
    const int g = k + 1 < n_dev ? k + 1 : 0;   // 1, 2, ..., n_dev - 1, 0
    ShardedBases::Part& part = sb.parts[(size_t)g];
    const size_t lo = part.lo, hi = std::min(part.hi, length);
    if (hi <= lo) continue;
    const uint64_t* sc;
    if (g == 0) {
      sc = src.dev0() + 12 * lo;
    } else {
      const DevSlice* sl = src.slices && (size_t)g < src.slices->size() && (*src.slices)[(size_t)g].buf ? &(*src.slices)[(size_t)g] : nullptr;
      if (sl && src.offset + lo >= sl->first && src.offset + hi <= sl->first + sl->count) {
        sl->ready->wait();
        sc = reinterpret_cast<const uint64_t*>(sl->buf->ptr) + 12 * (src.offset + lo - sl->first);
      } else {
        if (!part.scalars || part.scalars->bytes < 96 * (hi - lo)) {
          struct BackToDevice0 { ~BackToDevice0() { (void)mnt753_set_device(0); } } back;   // also when the allocation throws
          check(mnt753_set_device(g), "mnt753_set_device");
          part.scalars = std::make_shared<DeviceBuffer>(96 * (part.hi - part.lo));
        }
        check(mnt753_copy_peer_async(g, part.scalars->ptr, 0, src.dev0() + 12 * lo, 96 * (hi - lo)), "mnt753_copy_peer_async");
        sc = reinterpret_cast<const uint64_t*>(part.scalars->ptr);
      }
    }
    check(mnt753_msm_start(part.set->h, 0, sc, 1, hi - lo, nullptr), what);
    pend->sets[(size_t)g] = part.set;
  }
  // rank order for the fold (multiexp.tcc:433-438), whatever the order of enqueueing was
  pend->sets.erase(std::remove(pend->sets.begin(), pend->sets.end(), nullptr), pend->sets.end());
  return pend;
}
template <class V> static ScalarSource source_of(V* v) {
  return ScalarSource{[v]() { return v->ptr(); }, v->slices.get(), v->offset};
}
// the scalars h | w_L | r w of the concatenated sum, assembled on device 0's default stream (two copies and one scaling pass, ~0.2 ms:
// behind compute_H, ahead of the MSM), and the MSM over H | L | B1
template <int CURVE>
static std::shared_ptr<PendingMsm> start_fused_c(typename HIP_B::groth16_params* p, const uint64_t* h, const uint64_t* w_L, const uint64_t* w, const uint64_t* r) {
  const size_t d = p->d, m = p->m, n = d + 2 * m;
  auto sc = std::make_shared<DeviceBuffer>(96 * n);
  uint64_t* s = reinterpret_cast<uint64_t*>(sc->ptr);
  check(mnt753_copy_d2d(s, h, 96 * d), "mnt753_copy_d2d");
  check(mnt753_copy_d2d(s + 12 * d, w_L, 96 * (m - 1)), "mnt753_copy_d2d");
  check(mnt753_vec_scale(CURVE, s + 12 * (d + m - 1), w, r, m + 1, nullptr), "mnt753_vec_scale");
  auto pend = start_sharded(*p->HLB, ScalarSource{[s]() { return (const uint64_t*)s; }, nullptr, 0}, n, "mnt753_msm_start(C)");
  pend->scalars = sc;
  if (const char* e = getenv("MNT753_TRACE")) { if (atoi(e)) fprintf(stderr, "mnt753: C = Ht + Lt + r Bt1 as one MSM over H | L | B1 (%zu points)\n", n); }
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
  return start_fused_c<CURVE>(p, t[3]->scalars(), t[2]->scalars(), t[1]->scalars(), r);
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
      tmp.pending = start_sharded(sb, ScalarSource{n.scalars, n.slices.get(), n.offset}, n.length, "mnt753_msm_start(G1)");
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
  out->pending = start_fused_c<CURVE>(p, coefficients_for_H->ptr(), w_L->ptr(), w->ptr(), r->data);
  return out;
}
template <int CURVE> void HIP_B::fuse_C(bool on) { g_fused_c = on ? 1 : 0; }

template <int CURVE> typename HIP_B::groth16_input* HIP_B::read_input(const char* path, groth16_params* params) {
  return new groth16_input(path, params->d, params->m);
}
template <int CURVE> typename HIP_B::r1cs* HIP_B::read_r1cs(const char* path) { return new r1cs(path); }
template <int CURVE> typename HIP_B::groth16_input* HIP_B::read_witness(const char* path, groth16_params* params, r1cs* cs) {
  return new groth16_input(path, params->d, params->m, cs->data);
}
template <int CURVE> void HIP_B::delete_r1cs(r1cs* a) { delete a; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_w(groth16_input* in) { return new vector_Fr{in->w, in->n_w, 0, in->w_ready, in->w_slices}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_ca(groth16_input* in) { return new vector_Fr{in->ca, in->n_c, 0, in->ca_ready, nullptr}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_cb(groth16_input* in) { return new vector_Fr{in->cb, in->n_c, 0, in->cb_ready, nullptr}; }
template <int CURVE> typename HIP_B::vector_Fr* HIP_B::input_cc(groth16_input* in) { return new vector_Fr{in->cc, in->n_c, 0, in->cc_ready, nullptr}; }
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
    pend.push_back(start_sharded(*sb, ScalarSource{[&dev]() { return reinterpret_cast<const uint64_t*>(dev.ptr); }, nullptr, 0}, sb->n, "mnt753_msm_start(warm-up)"));
  uint64_t sink[108];
  for (auto& pm : pend)
    for (auto& set : pm->sets) check(mnt753_msm_finish(set->h, sink), "mnt753_msm_finish(warm-up)");
}
template <int CURVE> typename HIP_B::groth16_params* HIP_B::read_params(const char* path) {
  groth16_params* p = new groth16_params(path);
  auto dom = cached_domain<CURVE>(p->d + 1);
  warm_up<CURVE>(p);
  // the buffers of one proof, allocated now and parked in the cache: w, ca, cb, cc, coefficients_for_H (device 0)
  {
    const size_t m_dom = mnt753_domain_size(dom->h);
    std::vector<std::unique_ptr<DeviceBuffer>> pre;
    pre.emplace_back(new DeviceBuffer(96 * (p->m + 1)));
    for (int k = 0; k < 3; ++k) pre.emplace_back(new DeviceBuffer(96 * (p->d + 1)));
    pre.emplace_back(new DeviceBuffer(96 * (m_dom + 1)));
    if (p->HLB) pre.emplace_back(new DeviceBuffer(96 * (p->d + 2 * p->m)));   // the scalars of groth16_C
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

template <int CURVE> typename HIP_B::vector_Fr* HIP_B::compute_H_fused(evaluation_domain* domain, vector_Fr* ca, vector_Fr* cb, vector_Fr* cc) {
  const size_t m = mnt753_domain_size(domain->data->h);
  auto h = std::make_shared<DeviceBuffer>(96 * (m + 1));
  check(mnt753_compute_h(domain->data->h, ca->ptr(), cb->ptr(), cc->ptr(), reinterpret_cast<uint64_t*>(h->ptr), nullptr), "mnt753_compute_h");
  return new vector_Fr{h, m + 1, 0, nullptr, nullptr};
}
template <int CURVE> double HIP_B::input_load_seconds(groth16_input* in) {
  in->cc_ready->wait();
  return *in->load_seconds;
}
template <int CURVE> const uint64_t* HIP_B::G1_words(const G1* a) { resolve<CURVE>(const_cast<G1*>(a)); return a->data; }
template <int CURVE> const uint64_t* HIP_B::G2_words(const G2* a) { resolve<CURVE>(const_cast<G2*>(a)); return a->data; }

template class mnt753_hip_impl<0>;
template class mnt753_hip_impl<1>;
