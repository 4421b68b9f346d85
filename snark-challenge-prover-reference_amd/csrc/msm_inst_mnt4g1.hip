// MSM kernels + host orchestration instantiated for Mnt4G1.
#include "msm_host.hpp"
#include "msm_api.hpp"
namespace mnt753 {
void msm_free_workspace(mnt753_bases* b) { free_ws(b); }
int bases_create_mnt4g1(mnt753_bases* b, const uint64_t* affine, int on_device, size_t n) { return bases_create_t<Mnt4G1>(b, affine, on_device, n); }
int msm_mnt4g1(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, uint64_t* out, hipStream_t st) {
  return msm_t<Mnt4G1, host::HMnt4G1>(b, base_offset, scalars, scalars_on_device, n, out, st);
}
int msm_start_mnt4g1(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, hipStream_t st) {
  return msm_start_t<Mnt4G1, host::HMnt4G1>(b, base_offset, scalars, scalars_on_device, n, st);
}
int msm_finish_mnt4g1(mnt753_bases* b, uint64_t* out) { return msm_finish_t<Mnt4G1, host::HMnt4G1>(b, out); }
}  // namespace mnt753
