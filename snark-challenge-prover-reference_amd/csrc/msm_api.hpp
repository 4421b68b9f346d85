// Per-group entry points of the MSM host code (each defined in its own translation unit so the four
// instantiations of the kernels compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
struct mnt753_bases;
namespace mnt753 {
void msm_free_workspace(mnt753_bases* b);
#define MNT753_DECL_GROUP(tag)                                                                                 \
  int bases_create_##tag(mnt753_bases* b, const uint64_t* affine, int on_device, size_t n);                   \
  int msm_##tag(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, \
                uint64_t* out, hipStream_t st);                                                                \
  int msm_start_##tag(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device,     \
                      size_t n, hipStream_t st);                                                               \
  int msm_finish_##tag(mnt753_bases* b, uint64_t* out);
MNT753_DECL_GROUP(mnt4g1)
MNT753_DECL_GROUP(mnt4g2)
MNT753_DECL_GROUP(mnt6g1)
MNT753_DECL_GROUP(mnt6g2)
#undef MNT753_DECL_GROUP
}  // namespace mnt753
