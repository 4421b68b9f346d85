// Sort stage of an MSM for large inputs: (bucket, table row) pairs ordered by bucket with rocPRIM's radix sort instead of
// the histogram-atomic counting sort of msm_kernels.hip.h.
//
// The counting sort issues one returning atomic per non-zero digit -- 38 * 2^20 = 40 M of them into a 2 MB histogram that
// every XCD updates, so they are executed at the memory side at ~20 G atomics/s: 1.9 ms, plus 1.3 ms for the scatter that
// re-reads digits and ranks.  Here: k_scalar_keys writes (key = bucket, value = row | sign) pairs without any atomic (zero
// digits get the sentinel key n_buckets and sort to the end), rocprim::radix_sort_pairs orders them (39.8 M pairs, 20 key
// bits: 1.0 ms measured, tools/experiments/sort_bench.hip), k_bucket_bounds finds every bucket's start in the sorted keys by
// bisection, the usual scan turns the counts into (padded) offsets and k_expand copies the values to their padded positions.
// The order of the entries inside a bucket differs from the counting sort's; the MSM result is a sum and does not depend on it.
// Group-independent, hence its own translation unit (rocPRIM is not pulled into the four point-arithmetic units).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common_host.hpp"
#include "msm_kernels.hip.h"
#include "msm_types.hpp"
#include "msm_sort.hpp"

using namespace mnt753;

namespace {
// same digit extraction as k_scalar_digits; writes one (key, value) pair per (window, scalar)
template <int FRM>
__global__ void __launch_bounds__(256) k_scalar_keys(const uint32_t* __restrict__ scal_wire, const uint8_t* __restrict__ inf,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, size_t n, int c, int W,
                                                    uint32_t hist_stride, uint32_t entry_stride, uint32_t entry_base, uint32_t sentinel) {
  __shared__ uint32_t sw[24 * 256];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int tid = threadIdx.x;
  if (i >= n) return;
  {
    uint32_t w[24], s[24];
    load_wire24(w, scal_wire + i * 24);
    fp_wire_to_integer<FRM>(s, w);
    if (inf[i]) {
#pragma unroll
      for (int j = 0; j < 24; ++j) s[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 24; ++j) sw[j * 256 + tid] = s[j];   // each lane only reads back its own column: no barrier needed
  }
  for (int w = 0; w < W; ++w) {
    const int pos = w * c;
    const uint32_t win = lds_bits(sw + tid, 256, pos, c);
    const uint32_t blo = pos ? lds_bits(sw + tid, 256, pos - 1, 1) : 0u;
    const uint32_t top = (win >> (c - 1)) & 1u;
    const int32_t d = (int32_t)win + (int32_t)blo - (int32_t)(top << c);
    const uint32_t b = d ? (uint32_t)(d < 0 ? -d : d) - 1u : 0u;
    keys[(size_t)w * n + i] = d ? (uint32_t)w * hist_stride + b : sentinel;
    vals[(size_t)w * n + i] = ((uint32_t)w * entry_stride + entry_base + (uint32_t)i) | (d < 0 ? 0x80000000u : 0u);
  }
}
// dense[b] = first index of the sorted keys with key >= b, for b = 0 .. n_buckets (dense[n_buckets] = number of real entries)
__global__ void __launch_bounds__(256) k_bucket_bounds(const uint32_t* __restrict__ keys, size_t total, uint32_t* __restrict__ dense, uint32_t n_buckets) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_buckets) return;
  size_t lo = 0, hi = total;
  while (lo < hi) {
    const size_t mid = (lo + hi) >> 1;
    if (keys[mid] < b) lo = mid + 1; else hi = mid;
  }
  dense[b] = (uint32_t)lo;
}
__global__ void __launch_bounds__(256) k_bucket_counts(const uint32_t* __restrict__ dense, uint32_t* __restrict__ hist, uint32_t n_buckets) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_buckets) hist[b] = dense[b + 1] - dense[b];
}
// sorted entry i of bucket k goes to (offsets[k] << shift) + (i - dense[k])
__global__ void __launch_bounds__(256) k_expand(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ dense,
                                               const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, uint32_t shift, uint32_t n_buckets) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dense[n_buckets]) return;
  const uint32_t k = keys[i];
  sorted[((size_t)offsets[k] << shift) + (i - dense[k])] = vals[i];
}
unsigned key_bits(uint32_t sentinel) { unsigned b = 1; while ((1ull << b) <= sentinel) ++b; return b; }
}  // namespace

namespace mnt753 {
size_t msm_sort_temp_bytes(size_t total) {
  size_t bytes = 0;
  uint32_t* nul = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, total, 0, 32) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return bytes;
}

// keys_in / vals_in / keys_out / vals_out: W * n u32 each; dense: n_buckets + 2 u32; tmp: msm_sort_temp_bytes(W * n)
int msm_sort_radix(int frm, const uint32_t* d_scal, const uint8_t* d_inf, size_t n, const MsmPlan& p, uint32_t entry_stride, uint32_t entry_base,
                   uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, void* tmp, size_t tmp_bytes, uint32_t* dense,
                   uint32_t* d_hist, uint32_t* d_offsets, uint32_t* d_cursor, uint32_t* d_blocksums, uint32_t* d_total, uint32_t* d_sorted,
                   hipStream_t st) {
  const size_t total = (size_t)p.W * n;
  const uint32_t sentinel = p.n_buckets;
  const unsigned gb = (unsigned)((n + 255) / 256);
  const uint32_t hs = p.pre ? 0u : p.nb;
  if (frm == MOD_A)
    hipLaunchKernelGGL((k_scalar_keys<MOD_A>), dim3(gb), dim3(256), 0, st, d_scal, d_inf, keys_in, vals_in, n, p.c, p.W, hs, entry_stride, entry_base, sentinel);
  else
    hipLaunchKernelGGL((k_scalar_keys<MOD_B>), dim3(gb), dim3(256), 0, st, d_scal, d_inf, keys_in, vals_in, n, p.c, p.W, hs, entry_stride, entry_base, sentinel);
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, total, 0, key_bits(sentinel), st));
  hipLaunchKernelGGL(k_bucket_bounds, dim3((p.n_buckets + 256) / 256), dim3(256), 0, st, keys_out, total, dense, p.n_buckets);
  hipLaunchKernelGGL(k_bucket_counts, dim3((p.n_buckets + 255) / 256), dim3(256), 0, st, dense, d_hist, p.n_buckets);
  const unsigned nsb = (unsigned)(((size_t)p.n_buckets + SCAN_BLOCK - 1) / SCAN_BLOCK);
  const uint32_t pshift = (uint32_t)p.pair_levels;
  if (pshift) HIP_TRY(hipMemsetAsync(d_sorted, 0xff, sizeof(uint32_t) * (total + (size_t)p.n_buckets * (((size_t)1 << pshift) - 1)), st));
  hipLaunchKernelGGL(k_scan_blocks, dim3(nsb), dim3(SCAN_THREADS), 0, st, d_hist, d_offsets, d_blocksums, (size_t)p.n_buckets, pshift);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, d_blocksums, (size_t)nsb, d_total);
  hipLaunchKernelGGL(k_scan_finish, dim3(nsb), dim3(SCAN_THREADS), 0, st, d_offsets, d_cursor, d_blocksums, d_total, (size_t)p.n_buckets);
  hipLaunchKernelGGL(k_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, keys_out, vals_out, dense, d_offsets, d_sorted, pshift, p.n_buckets);
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace mnt753
