// The device-wide sort stage of an MSM (msm_sort.hip): group-independent, shared by the four instantiation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "msm_types.hpp"

namespace mnt753 {
// hand-written two-level counting sort (round 3): d_offsets, padded d_sorted; d_hist ends as the bucket counts
int msm_sort_partition(int frm, const uint32_t* d_scal, const uint8_t* d_inf, size_t n, const MsmPlan& p, uint32_t entry_stride, uint32_t entry_base,
                       uint32_t* keys_out, uint32_t* vals_out, uint32_t* part_ws, uint32_t* d_hist, uint32_t* d_offsets, uint32_t* d_cursor,
                       uint32_t* d_blocksums, uint32_t* d_total, uint32_t* d_sorted, hipStream_t st);
size_t msm_sort_partition_ws_words();
bool msm_sort_partition_fits(uint32_t n_buckets, int W);   // the plan fits the LDS staging of the placing pass
}  // namespace mnt753
