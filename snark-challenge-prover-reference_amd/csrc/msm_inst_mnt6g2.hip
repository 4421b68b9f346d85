// MSM kernels + host orchestration instantiated for Mnt6G2.
#include "msm_host.hpp"
#include "msm_api.hpp"
namespace mnt753 {
int bases_create_mnt6g2(mnt753_bases* b, const uint64_t* affine, int on_device, size_t n) { return bases_create_t<Mnt6G2>(b, affine, on_device, n); }
int msm_mnt6g2(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, uint64_t* out, hipStream_t st) {
  return msm_t<Mnt6G2, host::HMnt6G2>(b, base_offset, scalars, scalars_on_device, n, out, st);
}
int msm_start_mnt6g2(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, hipStream_t st) {
  return msm_start_t<Mnt6G2, host::HMnt6G2>(b, base_offset, scalars, scalars_on_device, n, st);
}
int msm_finish_mnt6g2(mnt753_bases* b, uint64_t* out) { return msm_finish_t<Mnt6G2, host::HMnt6G2>(b, out); }
}  // namespace mnt753
