// FFT part of the C ABI (include/mnt753_hip.h): evaluation domains, the four transform kinds,
// divide_by_Z_on_coset, the element-wise vector ops and the device-resident compute_H.
#include <hip/hip_runtime.h>
#include <cstring>
#include <new>
#include <vector>

#include "common_host.hpp"
#include "host_field.hpp"
#include "ntt_kernels.hip.h"

using namespace mnt753;
using namespace mnt753::host;

struct mnt753_domain {
  int curve = 0, frm = 0;
  int device = 0, logical_device = 0;                  // physical HIP ordinal / logical device the tables live on (the creating thread's current device)
  size_t m = 0;
  int logm = 0;
  uint32_t *tw_fwd = nullptr, *tw_inv = nullptr;       // omega^i, omega^-i            (m/2 each)
  uint32_t *cos_fwd = nullptr;                         // g^i                         (m)
  uint32_t *cos_fwd_s = nullptr;                       // g^i / m                     (m)  iFFT scale fused with the coset shift
  uint32_t *cos_inv_s = nullptr;                       // g^-i / m                    (m)
  uint32_t *consts = nullptr;                          // [0] 1/m  [1] 2^12 (k1)  [2] Z^-1 * 2^-12 (k2)  [3] Z^-1
  uint32_t *work = nullptr;                            // m wire elements
  hipEvent_t work_free = nullptr;                      // recorded after every transform: the next user of `work` (any stream) waits for it
  uint32_t *stage = nullptr;                           // the seeds the tables were generated from (freed with the domain:
                                                       // hipFree is a device-wide sync and would wait for MSMs in flight)
};

namespace {

template <int M>
int build_tables(mnt753_domain* d) {
  typedef HFp<M> Fr;
  const size_t m = d->m;
  const int logm = d->logm;
  // omega: primitive m-th root = (2^s-th root)^(2^(s-logm))   (libff get_root_of_unity, field_utils.tcc:40-89)
  Fr omega = Fr::from_words(FRD[M].root_of_unity);
  for (int i = FRD[M].two_adicity; i > logm; --i) omega = omega.squared();
  Fr omega_inv = omega.inverse();
  Fr g = Fr::from_words(FRD[M].mult_gen), g_inv = g.inverse();
  Fr minv = Fr::from_uint((uint64_t)m).inverse();
  Fr z = g.pow_u64((uint64_t)m) - Fr::one();     // vanishing polynomial on the coset, basic_radix2_domain.tcc:113-116
  Fr zinv = z.inverse();
  Fr two12 = Fr::from_uint(4096), two12_inv = two12.inverse();

  // staging buffer (host): 4 power tables of 32 entries + 4 scales + 4 constants, all wire form
  std::vector<uint64_t> stage((4 * 32 + 4 + 4) * 12);
  auto put = [&](size_t slot, const Fr& v) { memcpy(&stage[slot * 12], v.l, 96); };
  const Fr bases[4] = {omega, omega_inv, g, g_inv};
  for (int t = 0; t < 4; ++t) {
    Fr p = bases[t];
    for (int k = 0; k < 32; ++k) { put(t * 32 + k, p); p = p.squared(); }
  }
  put(128, Fr::one()); put(129, minv);                   // scales
  put(132, minv); put(133, two12); put(134, zinv * two12_inv); put(135, zinv);
  HIP_TRY(hipMalloc(&d->stage, stage.size() * 8));
  uint32_t* d_stage = d->stage;
  HIP_TRY(hipMemcpy(d_stage, stage.data(), stage.size() * 8, hipMemcpyHostToDevice));
  const size_t half = m / 2 ? m / 2 : 1;
  HIP_TRY(hipMalloc(&d->tw_fwd, half * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->tw_inv, half * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->cos_fwd, m * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->cos_fwd_s, m * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->cos_inv_s, m * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->consts, 4 * FPS_WORDS * 4));
  HIP_TRY(hipMalloc(&d->work, m * 96));
  auto table = [&](uint32_t* out, int base_slot, int scale_slot, size_t n) {
    hipLaunchKernelGGL((k_pow_table<M>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, out, d_stage + (size_t)base_slot * 32 * 24,
                       d_stage + (size_t)scale_slot * 24, n, logm);
  };
  table(d->tw_fwd, 0, 128, half);
  table(d->tw_inv, 1, 128, half);
  table(d->cos_fwd, 2, 128, m);
  table(d->cos_fwd_s, 2, 129, m);
  table(d->cos_inv_s, 3, 129, m);
  hipLaunchKernelGGL((k_consts_to_internal<M>), dim3(1), dim3(64), 0, 0, d->consts, d_stage + (size_t)132 * 24, 4);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(nullptr));   // the tables are built on the default stream; MSM streams are not waited for
  return 0;
}

// bit-reversal + log2 m butterfly stages, in place on `vec` (via the domain's work buffer)
// in_scale / out_scale: tables the elements are multiplied by on the way into the first group / out of the last (k_ntt_group)
template <int M>
int run_stages(mnt753_domain* d, uint32_t* vec, const uint32_t* tw, hipStream_t st, const uint32_t* in_scale = nullptr, const uint32_t* out_scale = nullptr) {
  const int logm = d->logm;
  if (logm == 0) return 0;
  int n_groups = (logm + NTT_MAX_NS - 1) / NTT_MAX_NS;
  int s0 = 0;
  for (int gi = 0; gi < n_groups; ++gi) {
    int ns = (logm - s0 + (n_groups - gi) - 1) / (n_groups - gi);   // balanced split
    const uint32_t* src = gi == 0 ? vec : d->work;
    uint32_t* dst = (gi == n_groups - 1 && n_groups > 1) ? vec : d->work;
    const size_t n_tiles = (size_t)1 << (logm - ns);
    const int tiles_per_block = NTT_BLOCK / (1 << (ns - 1));
    const unsigned blocks = (unsigned)((n_tiles + tiles_per_block - 1) / tiles_per_block);
    const uint32_t* is = gi == 0 ? in_scale : nullptr;
    const uint32_t* os = gi == n_groups - 1 ? out_scale : nullptr;
    const int bitrev = gi == 0 ? 1 : 0;
    // (no transform of the path scales on both sides: callers pass in_scale or out_scale, never both)
    if (is && os) return set_error(MNT753_EINVAL, "ntt: a transform scales on the way in or on the way out, not both");
    else if (is) hipLaunchKernelGGL((k_ntt_group<M, true, false>), dim3(blocks), dim3(NTT_BLOCK), 0, st, src, dst, tw, logm, s0, ns, bitrev, is, os);
    else if (os) hipLaunchKernelGGL((k_ntt_group<M, false, true>), dim3(blocks), dim3(NTT_BLOCK), 0, st, src, dst, tw, logm, s0, ns, bitrev, is, os);
    else hipLaunchKernelGGL((k_ntt_group<M, false, false>), dim3(blocks), dim3(NTT_BLOCK), 0, st, src, dst, tw, logm, s0, ns, bitrev, is, os);
    s0 += ns;
  }
  if (n_groups == 1) HIP_TRY(hipMemcpyAsync(vec, d->work, d->m * 96, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int M>
int fft_t(mnt753_domain* d, int kind, uint32_t* vec, hipStream_t st) {
  const size_t m = d->m;
  const unsigned gb = (unsigned)((m + 255) / 256);
  switch (kind) {
    case MNT753_FFT:
      return run_stages<M>(d, vec, d->tw_fwd, st);
    case MNT753_IFFT:
      if (int rc = run_stages<M>(d, vec, d->tw_inv, st)) return rc;
      hipLaunchKernelGGL((k_vec_mul_const<M>), dim3(gb), dim3(256), 0, st, vec, d->consts, m);
      break;
    case MNT753_COSET_FFT:
      return run_stages<M>(d, vec, d->tw_fwd, st, d->cos_fwd, nullptr);
    case MNT753_ICOSET_FFT:
      return run_stages<M>(d, vec, d->tw_inv, st, nullptr, d->cos_inv_s);
    default:
      return set_error(MNT753_EINVAL, "fft: unknown kind");
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// compute_H (cuda_prover_piecewise.cu:18-53), all on the device:
//   x -> cosetFFT(iFFT(x)) for x in {a, b, c} = stages(inv), *(g^i/m), stages(fwd)
//   a = (a*b - c)/Z ; a = icosetFFT(a) ; h = a | 0
// x -> cosetFFT(iFFT(x)) = stages(inv), *(g^i/m), stages(fwd)
template <int M>
int h_chain_t(mnt753_domain* d, uint32_t* vec, hipStream_t st) {
  if (int rc = run_stages<M>(d, vec, d->tw_inv, st)) return rc;
  return run_stages<M>(d, vec, d->tw_fwd, st, d->cos_fwd_s, nullptr);      // (g^i / m) rides on the forward transform's first pass
}
// a = (a*b - c)/Z ; a = icosetFFT(a) ; h = a | 0
template <int M>
int h_finish_t(mnt753_domain* d, uint32_t* ca, const uint32_t* cb, const uint32_t* cc, uint32_t* h, hipStream_t st) {
  const size_t m = d->m;
  const unsigned gb = (unsigned)((m + 255) / 256);
  hipLaunchKernelGGL((k_h_pointwise<M>), dim3(gb), dim3(256), 0, st, ca, cb, cc, d->consts + 1 * FPS_WORDS, d->consts + 2 * FPS_WORDS, m);
  if (int rc = run_stages<M>(d, ca, d->tw_inv, st, nullptr, d->cos_inv_s)) return rc;
  const size_t quads = m * 6 + 6;
  hipLaunchKernelGGL(k_copy_h, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, h, ca, m);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int M>
int compute_h_t(mnt753_domain* d, uint32_t* ca, uint32_t* cb, uint32_t* cc, uint32_t* h, hipStream_t st) {
  uint32_t* vecs[3] = {ca, cb, cc};
  for (int v = 0; v < 3; ++v)
    if (int rc = h_chain_t<M>(d, vecs[v], st)) return rc;
  return h_finish_t<M>(d, ca, cb, cc, h, st);
}

}  // namespace

extern "C" {

int mnt753_domain_create(int curve, size_t m, mnt753_domain** out) {
  if (!out || curve < 0 || curve > 1) return set_error(MNT753_EINVAL, "domain_create: bad argument");
  if (int rc = require_device()) return rc;
  const int frm = curve == MNT753_CURVE_MNT4753 ? MOD_A : MOD_B;
  // basic_radix2_domain constructor (basic_radix2_domain.tcc:25-60): m > 1, a power of two, log2 m <= s
  int logm = 0;
  while (((size_t)1 << logm) < m) ++logm;
  if (m <= 1 || ((size_t)1 << logm) != m || logm > FRD[frm].two_adicity)
    return set_error(MNT753_EDOMAIN, "domain_create: size must be a power of two in (1, 2^s], s = 30 (MNT4753) / 15 (MNT6753)");
  mnt753_domain* d = new (std::nothrow) mnt753_domain();
  if (!d) return set_error(MNT753_ENOMEM, "domain_create: host allocation failed");
  d->curve = curve; d->frm = frm; d->m = m; d->logm = logm;
  d->device = current_physical_device(); d->logical_device = mnt753_get_device();
  OnDevice on(d->device);
  int rc = frm == MOD_A ? build_tables<MOD_A>(d) : build_tables<MOD_B>(d);
  if (rc) { mnt753_domain_free(d); return rc; }
  *out = d;
  return 0;
}

int mnt753_domain_free(mnt753_domain* d) {
  if (!d) return 0;
  OnDevice on(d->device);
  void* ptrs[] = {d->tw_fwd, d->tw_inv, d->cos_fwd, d->cos_fwd_s, d->cos_inv_s, d->consts, d->work, d->stage};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (d->work_free) (void)hipEventDestroy(d->work_free);
  delete d;
  return 0;
}

size_t mnt753_domain_size(const mnt753_domain* d) { return d ? d->m : 0; }
int mnt753_domain_device(const mnt753_domain* d) { return d ? d->logical_device : -1; }

// Every transform of a domain ping-pongs through the domain's one work buffer.  Callers on different streams are serialised on
// the device: a transform first waits for the event the previous one recorded (host threads must still not call into the
// same domain concurrently -- the threading contract of the reference wrapper).
static int work_acquire(mnt753_domain* d, hipStream_t st) {
  if (!d->work_free) HIP_TRY(hipEventCreateWithFlags(&d->work_free, hipEventDisableTiming));
  else HIP_TRY(hipStreamWaitEvent(st, d->work_free, 0));
  return 0;
}
static int work_release(mnt753_domain* d, hipStream_t st, int rc) {
  if (rc) return rc;
  HIP_TRY(hipEventRecord(d->work_free, st));
  return 0;
}

int mnt753_fft(mnt753_domain* d, int kind, uint64_t* dev_vec, void* stream) {
  if (!d || !dev_vec) return set_error(MNT753_EINVAL, "fft: null argument");
  if (int rc = require_device()) return rc;
  uint32_t* v = reinterpret_cast<uint32_t*>(dev_vec);
  OnDevice on(d->device);
  if (int rc = work_acquire(d, (hipStream_t)stream)) return rc;
  const int rc = d->frm == MOD_A ? fft_t<MOD_A>(d, kind, v, (hipStream_t)stream) : fft_t<MOD_B>(d, kind, v, (hipStream_t)stream);
  return work_release(d, (hipStream_t)stream, rc);
}

int mnt753_divide_by_z_on_coset(mnt753_domain* d, uint64_t* dev_vec, void* stream) {
  if (!d || !dev_vec) return set_error(MNT753_EINVAL, "divide_by_z_on_coset: null argument");
  if (int rc = require_device()) return rc;
  const unsigned gb = (unsigned)((d->m + 255) / 256);
  uint32_t* v = reinterpret_cast<uint32_t*>(dev_vec);
  OnDevice on(d->device);
  if (d->frm == MOD_A) hipLaunchKernelGGL((k_vec_mul_const<MOD_A>), dim3(gb), dim3(256), 0, (hipStream_t)stream, v, d->consts + 3 * FPS_WORDS, d->m);
  else hipLaunchKernelGGL((k_vec_mul_const<MOD_B>), dim3(gb), dim3(256), 0, (hipStream_t)stream, v, d->consts + 3 * FPS_WORDS, d->m);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mnt753_vec_muleq(int curve, uint64_t* dev_a, const uint64_t* dev_b, size_t n, void* stream) {
  if (curve < 0 || curve > 1 || (n && (!dev_a || !dev_b))) return set_error(MNT753_EINVAL, "vec_muleq: bad argument");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  const unsigned gb = (unsigned)((n + 255) / 256);
  uint32_t* a = reinterpret_cast<uint32_t*>(dev_a);
  const uint32_t* b = reinterpret_cast<const uint32_t*>(dev_b);
  if (curve == MNT753_CURVE_MNT4753) hipLaunchKernelGGL((k_vec_muleq<MOD_A>), dim3(gb), dim3(256), 0, (hipStream_t)stream, a, b, n);
  else hipLaunchKernelGGL((k_vec_muleq<MOD_B>), dim3(gb), dim3(256), 0, (hipStream_t)stream, a, b, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mnt753_vec_scale(int curve, uint64_t* dev_dst, const uint64_t* dev_src, const uint64_t* host_scalar, size_t n, void* stream) {
  if (curve < 0 || curve > 1 || !host_scalar || (n && (!dev_dst || !dev_src))) return set_error(MNT753_EINVAL, "vec_scale: bad argument");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  const unsigned gb = (unsigned)((n + 255) / 256);
  WireElem k;
  memcpy(k.w, host_scalar, 96);
  uint32_t* d = reinterpret_cast<uint32_t*>(dev_dst);
  const uint32_t* s = reinterpret_cast<const uint32_t*>(dev_src);
  if (curve == MNT753_CURVE_MNT4753) hipLaunchKernelGGL((k_vec_scale<MOD_A>), dim3(gb), dim3(256), 0, (hipStream_t)stream, d, s, k, n);
  else hipLaunchKernelGGL((k_vec_scale<MOD_B>), dim3(gb), dim3(256), 0, (hipStream_t)stream, d, s, k, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mnt753_vec_subeq(int curve, uint64_t* dev_a, const uint64_t* dev_b, size_t n, void* stream) {
  if (curve < 0 || curve > 1 || (n && (!dev_a || !dev_b))) return set_error(MNT753_EINVAL, "vec_subeq: bad argument");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  const unsigned gb = (unsigned)((n + 255) / 256);
  uint32_t* a = reinterpret_cast<uint32_t*>(dev_a);
  const uint32_t* b = reinterpret_cast<const uint32_t*>(dev_b);
  if (curve == MNT753_CURVE_MNT4753) hipLaunchKernelGGL((k_vec_subeq<MOD_A>), dim3(gb), dim3(256), 0, (hipStream_t)stream, a, b, n);
  else hipLaunchKernelGGL((k_vec_subeq<MOD_B>), dim3(gb), dim3(256), 0, (hipStream_t)stream, a, b, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mnt753_compute_h(mnt753_domain* d, uint64_t* dev_ca, uint64_t* dev_cb, uint64_t* dev_cc, uint64_t* dev_h, void* stream) {
  if (!d || !dev_ca || !dev_cb || !dev_cc || !dev_h) return set_error(MNT753_EINVAL, "compute_h: null argument");
  if (int rc = require_device()) return rc;
  uint32_t *a = reinterpret_cast<uint32_t*>(dev_ca), *b = reinterpret_cast<uint32_t*>(dev_cb), *c = reinterpret_cast<uint32_t*>(dev_cc),
           *h = reinterpret_cast<uint32_t*>(dev_h);
  OnDevice on(d->device);
  if (int rc = work_acquire(d, (hipStream_t)stream)) return rc;
  const int rc = d->frm == MOD_A ? compute_h_t<MOD_A>(d, a, b, c, h, (hipStream_t)stream) : compute_h_t<MOD_B>(d, a, b, c, h, (hipStream_t)stream);
  return work_release(d, (hipStream_t)stream, rc);
}

int mnt753_compute_h_chain(mnt753_domain* d, uint64_t* dev_vec, void* stream) {
  if (!d || !dev_vec) return set_error(MNT753_EINVAL, "compute_h_chain: null argument");
  if (int rc = require_device()) return rc;
  OnDevice on(d->device);
  if (int rc = work_acquire(d, (hipStream_t)stream)) return rc;
  uint32_t* v = reinterpret_cast<uint32_t*>(dev_vec);
  const int rc = d->frm == MOD_A ? h_chain_t<MOD_A>(d, v, (hipStream_t)stream) : h_chain_t<MOD_B>(d, v, (hipStream_t)stream);
  return work_release(d, (hipStream_t)stream, rc);
}

int mnt753_compute_h_finish(mnt753_domain* d, uint64_t* dev_a, const uint64_t* dev_b, const uint64_t* dev_c, uint64_t* dev_h, void* stream) {
  if (!d || !dev_a || !dev_b || !dev_c || !dev_h) return set_error(MNT753_EINVAL, "compute_h_finish: null argument");
  if (int rc = require_device()) return rc;
  OnDevice on(d->device);
  if (int rc = work_acquire(d, (hipStream_t)stream)) return rc;
  uint32_t* a = reinterpret_cast<uint32_t*>(dev_a);
  const uint32_t *b = reinterpret_cast<const uint32_t*>(dev_b), *c = reinterpret_cast<const uint32_t*>(dev_c);
  uint32_t* h = reinterpret_cast<uint32_t*>(dev_h);
  const int rc = d->frm == MOD_A ? h_finish_t<MOD_A>(d, a, b, c, h, (hipStream_t)stream) : h_finish_t<MOD_B>(d, a, b, c, h, (hipStream_t)stream);
  return work_release(d, (hipStream_t)stream, rc);
}

}  // extern "C"
