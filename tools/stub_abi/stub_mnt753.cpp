// TEST STUB of the C ABI (include/mnt753_hip.h) for the host-only sanitizer builds (`make asan`, `make tsan`): "device" memory is
// host memory, copies are memcpy, the file loader is fread -- and every compute entry point (MSM, FFT, compute_H, constraint
// evaluation) does NO arithmetic: an MSM returns the identity, a transform leaves its vector alone.  It exists so that the host
// side of the product (host/main.cpp, host/prover_hip_functions.cpp: loader threads, readiness latches, per-device slices,
// sharded folds, error paths) can run under AddressSanitizer / UBSan / ThreadSanitizer on a machine without a GPU (GPU sanitizers
// are not available on the pool).  It is never linked into the product, the tests of results, or the bench; a binary built on it
// writes the proof of the all-identity MSMs, which nothing compares with anything.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/mnt753_hip.h"
#include "../../snark-challenge-prover-reference_amd/csrc/host_field.hpp"

using namespace mnt753;
using namespace mnt753::host;

namespace {
thread_local std::string t_err;
thread_local int t_dev = 0;
int g_ndev = 0;
std::mutex g_io[16];
int fail(int code, const char* msg) { t_err = msg; return code; }
bool bad_cg(int curve, int group) { return curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2); }
int deg_of(int curve, int group) { return group == MNT753_G1 ? 1 : (curve == 0 ? 2 : 3); }
template <class HC> int add_t(const uint64_t* a, const uint64_t* b, uint64_t* o) { HPoint<HC>::from_wire(a).add(HPoint<HC>::from_wire(b)).to_wire(o); return 0; }
template <class HC> int scale_t(const uint64_t* s, const uint64_t* p, uint64_t* o) {
  uint64_t e[12]; HFp<HC::FR>::from_words(s).to_integer(e); HPoint<HC>::from_wire(p).mul_words(e, 12).to_wire(o); return 0;
}
template <class HC> int aff_t(const uint64_t* p, uint64_t* o) {
  typename HC::F x, y; HPoint<HC>::from_wire(p).to_affine(x, y);
  for (int k = 0; k < HC::F::DEG; ++k) { memcpy(o + 12 * k, x.comp(k).l, 96); memcpy(o + 12 * (HC::F::DEG + k), y.comp(k).l, 96); }
  return 0;
}
template <class HC> int from_aff_t(const uint64_t* a, uint64_t* o) {
  typedef typename HC::F F; HPoint<HC> p;
  for (int k = 0; k < F::DEG; ++k) { p.X.comp(k) = F::B::from_words(a + 12 * k); p.Y.comp(k) = F::B::from_words(a + 12 * (F::DEG + k)); }
  if (p.Y.is_zero()) p = HPoint<HC>::zero(); else p.Z = F::one();
  p.to_wire(o); return 0;
}
template <class HC> int zero_t(uint64_t* o) { HPoint<HC>::zero().to_wire(o); return 0; }
}  // namespace
#define CG(fn, ...) (curve == 0 ? (group == MNT753_G1 ? fn<HMnt4G1>(__VA_ARGS__) : fn<HMnt4G2>(__VA_ARGS__)) : (group == MNT753_G1 ? fn<HMnt6G1>(__VA_ARGS__) : fn<HMnt6G2>(__VA_ARGS__)))

struct mnt753_bases { int curve, group, device; size_t n; void* copy; int pending; size_t pending_n; };
struct mnt753_domain { int curve; size_t m; int device; };
struct mnt753_r1cs { uint64_t num_inputs, m, nc; };

extern "C" {
int mnt753_init(int device) { if (device != 0) return fail(MNT753_EINVAL, "stub: one device"); if (g_ndev < 1) g_ndev = 1; t_dev = 0; return 0; }
int mnt753_init_devices(int n) { if (n < 1 || n > 16) return fail(MNT753_EINVAL, "mnt753_init_devices: more devices requested than visible"); g_ndev = n; t_dev = 0; return 0; }
int mnt753_device_count(void) { return g_ndev; }
int mnt753_set_device(int d) { if (d < 0 || d >= g_ndev) return fail(MNT753_EINVAL, "mnt753_set_device: not an initialised device"); t_dev = d; return 0; }
int mnt753_get_device(void) { return t_dev; }
// peer access: the stub's devices are host memory; every request is recorded on stderr under MNT753_TRACE=1 so that the sanitizer
// suite can assert that the wrapper asks for every ordered pair
int mnt753_enable_peer_access(int a, int b, int* how) {
  if (a < 0 || a >= g_ndev || b < 0 || b >= g_ndev) return fail(MNT753_EINVAL, "enable_peer_access: not an initialised device");
  if (const char* e = getenv("MNT753_TRACE")) { if (atoi(e)) fprintf(stderr, "stub: peer access requested %d -> %d\n", a, b); }
  if (how) *how = MNT753_PEER_SAME;
  return 0;
}
int mnt753_copy_peer(int, void* d, int, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int mnt753_copy_peer_async(int dd, void* d, int sd, const void* s, size_t n) { if (dd < 0 || dd >= g_ndev || sd < 0 || sd >= g_ndev) return fail(MNT753_EINVAL, "copy_peer_async: bad device"); if (n) memcpy(d, s, n); return 0; }
const char* mnt753_last_error(void) { return t_err.c_str(); }
// the exchange: a plain copy (the stub's "devices" are host memory); MNT753_STUB_NO_RCCL=1 plays a box without librccl
int mnt753_exchange_points(const uint64_t* const* in, size_t words, uint64_t* out) {
  if (!in || !out || !words) return fail(MNT753_EINVAL, "exchange_points: bad argument");
  if (getenv("MNT753_STUB_NO_RCCL")) return fail(MNT753_ENODEV, "exchange_points: cannot load librccl (stub)");
  for (int g = 0; g < g_ndev; ++g) { if (!in[g]) return fail(MNT753_EINVAL, "exchange_points: null block"); memcpy(out + words * (size_t)g, in[g], 8 * words); }
  return 0;
}
double mnt753_exchange_last_us(void) { return 1.0; }
size_t mnt753_affine_words(int curve, int group) { return bad_cg(curve, group) ? 0 : (size_t)24 * deg_of(curve, group); }
size_t mnt753_projective_words(int curve, int group) { return bad_cg(curve, group) ? 0 : (size_t)36 * deg_of(curve, group); }
int mnt753_dev_alloc(void** p, size_t n) { if (!g_ndev) return fail(MNT753_ENODEV, "no device"); *p = malloc(n ? n : 16); return *p ? 0 : fail(MNT753_ENOMEM, "alloc"); }
int mnt753_dev_free(void* p) { free(p); return 0; }
int mnt753_copy_h2d(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int mnt753_copy_d2h(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int mnt753_copy_d2d(void* d, const void* s, size_t n) { if (n) memmove(d, s, n); return 0; }
int mnt753_dev_memset(void* d, int v, size_t n) { if (n) memset(d, v, n); return 0; }
int mnt753_sync(void*) { return 0; }
int mnt753_load_file_to_device(const char* path, size_t off, size_t bytes, void* dst) {
  if (!g_ndev) return fail(MNT753_ENODEV, "no device");
  std::lock_guard<std::mutex> l(g_io[t_dev & 15]);   // the product serialises a device's staging buffers the same way
  FILE* f = fopen(path, "rb");
  if (!f) return fail(MNT753_EINVAL, "load_file_to_device: cannot open file");
  int rc = 0;
  if (fseeko(f, (off_t)off, SEEK_SET) != 0 || (bytes && fread(dst, 1, bytes, f) != bytes)) rc = fail(MNT753_EINVAL, "load_file_to_device: short read");
  fclose(f);
  return rc;
}
int mnt753_bases_create(int curve, int group, const uint64_t* aff, int, size_t n, mnt753_bases** out) {
  if (!out || (n && !aff) || bad_cg(curve, group)) return fail(MNT753_EINVAL, "bases_create: bad argument");
  if (!g_ndev) return fail(MNT753_ENODEV, "no device");
  mnt753_bases* b = new mnt753_bases{curve, group, t_dev, n, nullptr, 0, 0};
  const size_t bytes = n * mnt753_affine_words(curve, group) * 8;
  b->copy = malloc(bytes ? bytes : 16);
  if (bytes) memcpy(b->copy, aff, bytes);       // reads every byte the caller promised: ASan checks the caller's buffer
  *out = b; return 0;
}
int mnt753_bases_free(mnt753_bases* b) { if (b) { free(b->copy); delete b; } return 0; }
size_t mnt753_bases_size(const mnt753_bases* b) { return b ? b->n : 0; }
int mnt753_msm_start(mnt753_bases* b, size_t off, const uint64_t* sc, int, size_t n, void*) {
  if (!b || (n && !sc)) return fail(MNT753_EINVAL, "msm_start: null argument");
  if (off + n > b->n) return fail(MNT753_EINVAL, "msm_start: base_offset + n exceeds the base set");
  if (b->pending) return fail(MNT753_EINVAL, "msm_start: this base set already has an MSM in flight (finish it first)");
  volatile uint64_t sink = 0;
  for (size_t i = 0; i < 12 * n; i += 12) sink += sc[i];   // touch the scalar range: ASan / TSan see the access the kernels would make
  (void)sink;
  b->pending = 1; b->pending_n = n; return 0;
}
int mnt753_msm_finish(mnt753_bases* b, uint64_t* out) {
  if (!b || !out) return fail(MNT753_EINVAL, "msm_finish: null argument");
  if (!b->pending) return fail(MNT753_EINVAL, "msm_finish: no MSM in flight on this base set");
  b->pending = 0;
  const int curve = b->curve, group = b->group;
  return CG(zero_t, out);
}
int mnt753_msm(mnt753_bases* b, size_t off, const uint64_t* sc, int od, size_t n, uint64_t* out, void* st) {
  if (int rc = mnt753_msm_start(b, off, sc, od, n, st)) return rc;
  return mnt753_msm_finish(b, out);
}
int mnt753_msm_set_window_bits(int) { return 0; }
int mnt753_dev_mem_info(size_t* f, size_t* t) { if (f) *f = (size_t)200 << 30; if (t) *t = (size_t)288 << 30; return 0; }
int mnt753_self_test_curve(int curve, int level) { if (getenv("MNT753_STUB_LOG")) fprintf(stderr, "stub: self-test level %d curve %d\n", level, curve); const char* e = getenv("MNT753_STUB_SELFTEST_FAILS"); return e && atoi(e) ? fail(MNT753_ESELFTEST, "stub: self-test check 3 failed (as asked)") : 0; }
int mnt753_self_test(int level) { if (getenv("MNT753_STUB_LOG")) fprintf(stderr, "stub: self-test level %d\n", level); const char* e = getenv("MNT753_STUB_SELFTEST_FAILS"); return e && atoi(e) ? fail(MNT753_ESELFTEST, "stub: self-test check 3 failed (as asked)") : 0; }
int mnt753_msm_set_window_table(int mode) { static int m = 1; const int old = m; m = mode != 0; if (getenv("MNT753_STUB_LOG")) fprintf(stderr, "stub: window table mode %d\n", m); return old; }
int mnt753_msm_order_after(mnt753_bases* b, const mnt753_bases* first) { return (b && first && b != first) ? 0 : 22; }
int mnt753_msm_last_timing(float o[5]) { for (int i = 0; i < 5; ++i) o[i] = 0; return 0; }
int mnt753_msm_last_plan(int o[4]) { for (int i = 0; i < 4; ++i) o[i] = 0; return 0; }
int mnt753_msm_last_pair_levels(void) { return 0; }
int mnt753_msm_last_irr_levels(void) { return 0; }
int mnt753_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* o) { if (bad_cg(curve, group)) return fail(MNT753_EINVAL, "point_add"); return CG(add_t, a, b, o); }
int mnt753_point_scale(int curve, int group, const uint64_t* s, const uint64_t* p, uint64_t* o) { if (bad_cg(curve, group)) return fail(MNT753_EINVAL, "point_scale"); return CG(scale_t, s, p, o); }
int mnt753_point_to_affine(int curve, int group, const uint64_t* p, uint64_t* o) { if (bad_cg(curve, group)) return fail(MNT753_EINVAL, "point_to_affine"); return CG(aff_t, p, o); }
int mnt753_point_from_affine(int curve, int group, const uint64_t* a, uint64_t* o) { if (bad_cg(curve, group)) return fail(MNT753_EINVAL, "point_from_affine"); return CG(from_aff_t, a, o); }
int mnt753_domain_create(int curve, size_t m, mnt753_domain** out) {
  if (!out || curve < 0 || curve > 1 || m == 0 || (m & (m - 1))) return fail(MNT753_EDOMAIN, "domain_create: not a power of two");
  *out = new mnt753_domain{curve, m, t_dev}; return 0;
}
int mnt753_domain_free(mnt753_domain* d) { delete d; return 0; }
size_t mnt753_domain_size(const mnt753_domain* d) { return d ? d->m : 0; }
static void touch(uint64_t* v, size_t n) { volatile uint64_t s = 0; for (size_t i = 0; i < 12 * n; i += 12) { s += v[i]; v[i] = v[i]; } (void)s; }
int mnt753_fft(mnt753_domain* d, int, uint64_t* v, void*) { if (!d || !v) return fail(MNT753_EINVAL, "fft: null"); touch(v, d->m); return 0; }
int mnt753_divide_by_z_on_coset(mnt753_domain* d, uint64_t* v, void*) { if (!d || !v) return fail(MNT753_EINVAL, "divide_by_z: null"); touch(v, d->m); return 0; }
int mnt753_vec_muleq(int, uint64_t* a, const uint64_t* b, size_t n, void*) { touch(a, n); touch(const_cast<uint64_t*>(b), n); return 0; }
int mnt753_vec_scale(int, uint64_t* d, const uint64_t* s, const uint64_t* k, size_t n, void*) { touch(const_cast<uint64_t*>(s), n); touch(d, n); return k ? 0 : -2; }
int mnt753_vec_subeq(int, uint64_t* a, const uint64_t* b, size_t n, void*) { touch(a, n); touch(const_cast<uint64_t*>(b), n); return 0; }
int mnt753_compute_h(mnt753_domain* d, uint64_t* a, uint64_t* b, uint64_t* c, uint64_t* h, void*) {
  if (!d || !a || !b || !c || !h) return fail(MNT753_EINVAL, "compute_h: null");
  touch(a, d->m); touch(b, d->m); touch(c, d->m); memset(h, 0, 96 * (d->m + 1)); return 0;
}
int mnt753_compute_h_chain(mnt753_domain* d, uint64_t* v, void*) { if (!d || !v) return fail(MNT753_EINVAL, "compute_h_chain: null"); touch(v, d->m); return 0; }
int mnt753_compute_h_finish(mnt753_domain* d, uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* h, void*) {
  if (!d || !a || !b || !c || !h) return fail(MNT753_EINVAL, "compute_h_finish: null");
  touch(a, d->m); touch(const_cast<uint64_t*>(b), d->m); touch(const_cast<uint64_t*>(c), d->m); memset(h, 0, 96 * (d->m + 1)); return 0;
}
int mnt753_domain_device(const mnt753_domain* d) { return d ? d->device : -1; }
int mnt753_r1cs_create(int curve, uint64_t ni, uint64_t m, uint64_t nc, const uint64_t* const rp[3], const uint32_t* const col[3], const uint64_t* const cf[3], mnt753_r1cs** out) {
  if (curve < 0 || curve > 1 || !out || !rp || !col || !cf || ni > m) return fail(MNT753_EINVAL, "r1cs_create: bad argument");
  for (int k = 0; k < 3; ++k) for (uint64_t i = 0; i < rp[k][nc]; ++i) if (col[k][i] > m) return fail(MNT753_EINVAL, "r1cs_create: variable index out of range");
  *out = new mnt753_r1cs{ni, m, nc}; return 0;
}
int mnt753_r1cs_free(mnt753_r1cs* r) { delete r; return 0; }
size_t mnt753_r1cs_domain_size(const mnt753_r1cs* r) { return r ? (size_t)(r->nc + r->num_inputs + 1) : 0; }
size_t mnt753_r1cs_num_variables(const mnt753_r1cs* r) { return r ? (size_t)r->m : 0; }
size_t mnt753_r1cs_num_inputs(const mnt753_r1cs* r) { return r ? (size_t)r->num_inputs : 0; }
int mnt753_r1cs_evaluate(mnt753_r1cs* r, const uint64_t* w, uint64_t* a, uint64_t* b, uint64_t* c, size_t n, void*) {
  if (!r || !w || !a || !b || !c) return fail(MNT753_EINVAL, "r1cs_evaluate: null");
  touch(const_cast<uint64_t*>(w), r->m + 1); memset(a, 0, 96 * n); memset(b, 0, 96 * n); memset(c, 0, 96 * n); return 0;
}
int mnt753_synth_scalars(int, uint64_t seed, size_t n, uint64_t* out) { for (size_t i = 0; i < 12 * n; ++i) out[i] = seed + i; return 0; }
}
