// The MSM part of the C ABI (include/mnt753_hip.h): dispatch to the per-group instantiations.
#include <hip/hip_runtime.h>
#include <cstring>
#include <new>
#include "common_host.hpp"
#include "msm_api.hpp"
#include "msm_types.hpp"

namespace mnt753 {
int g_window_bits_override = 0;
float g_last_timing[5] = {0, 0, 0, 0, 0};
int g_last_plan[4] = {0, 0, 0, 0};
}
using namespace mnt753;

extern "C" {


int mnt753_bases_create(int curve, int group, const uint64_t* affine, int on_device, size_t n, mnt753_bases** out) {
  if (!out || (n && !affine) || curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2)) return set_error(MNT753_EINVAL, "bases_create: bad argument");
  if (int rc = require_device()) return rc;
  mnt753_bases* b = new (std::nothrow) mnt753_bases();
  if (!b) return set_error(MNT753_ENOMEM, "bases_create: host allocation failed");
  b->curve = curve; b->group = group; b->n = n;
  int rc;
  if (curve == MNT753_CURVE_MNT4753) rc = group == MNT753_G1 ? bases_create_mnt4g1(b, affine, on_device, n) : bases_create_mnt4g2(b, affine, on_device, n);
  else rc = group == MNT753_G1 ? bases_create_mnt6g1(b, affine, on_device, n) : bases_create_mnt6g2(b, affine, on_device, n);
  if (rc) { mnt753_bases_free(b); return rc; }
  *out = b;
  return 0;
}

int mnt753_bases_free(mnt753_bases* b) {
  if (!b) return 0;
  msm_free_workspace(b);
  if (b->d_aff) (void)hipFree(b->d_aff);
  if (b->d_inf) (void)hipFree(b->d_inf);
  for (int i = 0; i < 5; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  delete b;
  return 0;
}

size_t mnt753_bases_size(const mnt753_bases* b) { return b ? b->n : 0; }

int mnt753_msm(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n,
               uint64_t* out_projective, void* stream) {
  if (!b || !out_projective || (n && !scalars)) return set_error(MNT753_EINVAL, "msm: null argument");
  if (base_offset + n > b->n) return set_error(MNT753_EINVAL, "msm: base_offset + n exceeds the base set");
  if (int rc = require_device()) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (b->curve == MNT753_CURVE_MNT4753)
    return b->group == MNT753_G1 ? msm_mnt4g1(b, base_offset, scalars, scalars_on_device, n, out_projective, st)
                                 : msm_mnt4g2(b, base_offset, scalars, scalars_on_device, n, out_projective, st);
  return b->group == MNT753_G1 ? msm_mnt6g1(b, base_offset, scalars, scalars_on_device, n, out_projective, st)
                               : msm_mnt6g2(b, base_offset, scalars, scalars_on_device, n, out_projective, st);
}

int mnt753_msm_set_window_bits(int c) {
  int old = g_window_bits_override;
  g_window_bits_override = (c >= 2 && c <= 22) ? c : 0;
  return old;
}

int mnt753_msm_last_timing(float out_ms[5]) {
  if (!out_ms) return set_error(MNT753_EINVAL, "msm_last_timing: null");
  memcpy(out_ms, g_last_timing, sizeof(g_last_timing));
  return 0;
}

int mnt753_msm_last_plan(int out[4]) {
  if (!out) return set_error(MNT753_EINVAL, "msm_last_plan: null");
  memcpy(out, g_last_plan, sizeof(g_last_plan));
  return 0;
}

}  // extern "C"
