// Sort stage of an MSM for large inputs (from 2^22 list entries): (bucket, table row) pairs ordered by bucket, instead of the
// histogram-atomic counting sort of msm_kernels.hip.h.
//
// That counting sort issues one returning atomic per non-zero digit -- 38 * 2^20 = 40 M of them into a 2 MB histogram that
// every XCD updates, executed at the memory side at ~20 G atomics/s: 1.9 ms, plus 1.3 ms for the scatter that re-reads digits and
// ranks.  msm_sort_partition (round 3) is a hand-written two-level counting sort with the Booth-digit extraction and the slot-tree
// padding fused in, without an atomic per entry reaching memory -- 0.92 ms at 2^20 G1 points, see the comment above its kernels.
// (Round 2's stage -- rocprim::radix_sort_pairs between four small kernels, 1.33 ms -- left the product in round 5 and took the
// rocPRIM dependency with it; profiles/r03/ab_sort_stage.txt keeps the A/B.)
// The order of the entries inside a bucket is whatever the passes produce; the MSM result is a sum and does not depend on it.
// Group-independent, hence its own translation unit.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include "common_host.hpp"
#include "msm_kernels.hip.h"
#include "msm_types.hpp"
#include "msm_sort.hpp"

using namespace mnt753;

namespace {
// ---- hand-written sort stage (round 3): two-level counting sort ----------------------------------------------------------------
// The keys are bucket numbers below 2^19..2^21 and nothing needs the order inside a bucket, so a comparison-free two-level counting
// sort does with ~1.2 GB of traffic what three onesweep passes do with ~2.2 GB, and the Booth-digit extraction and the padding of the
// slot tree fuse into its passes:
//   level 1  partitions of PART_BUCKETS = 1024 consecutive buckets.  k_part_count: a block of 256 scalars extracts its 38 x 256
//            digits, histograms them over the partitions in LDS and adds its counts to the global totals (one atomic per block and
//            non-empty partition).  k_part_scan: exclusive scan of the totals (one block).  k_part_place: the same block extracts the
//            digits again (cheaper than keeping them), reserves a range in every partition it feeds (one atomic each) and writes its
//            (bucket, entry) pairs there, ranked inside the block by LDS atomics: no atomic per entry ever reaches memory.
//   level 2  chunks of CHUNK = 8192 pairs of the partition-ordered list, whatever partitions they span (work is split by ENTRIES, so
//            a skewed digit distribution -- all scalars equal, half of them one -- costs no more than a uniform one).
//            k_bucket_count: LDS histogram over the 1024 buckets of a spanned partition, flushed to the global bucket histogram (one
//            atomic per chunk and non-empty bucket); the usual scan turns it into padded offsets (k_scan_blocks / _sums / _finish);
//            k_bucket_place_staged: counts again, reserves per bucket, orders the chunk's entries by bucket in LDS and writes them to their
//            padded positions; k_bucket_pad writes the ENTRY_EMPTY
//            padding of every bucket's last group (no 160 MB memset).
constexpr uint32_t PART_BITS = 10, PART_BUCKETS = 1u << PART_BITS, PART_MAX = 4096, SORT_CHUNK = 8192;

template <int FRM>
__device__ __forceinline__ void scalar_to_lds(const uint32_t* __restrict__ scal_wire, const uint8_t* __restrict__ inf, size_t i, size_t n, uint32_t* sw, int tid) {
  uint32_t w[24], s[24];
  if (i < n) {
    load_wire24(w, scal_wire + i * 24);
    fp_wire_to_integer<FRM>(s, w);
  }
  if (i >= n || inf[i]) {
#pragma unroll
    for (int j = 0; j < 24; ++j) s[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < 24; ++j) sw[j * 256 + tid] = s[j];   // each lane only reads back its own column: no barrier needed
}
__device__ __forceinline__ int32_t booth_digit(const uint32_t* sw, int tid, int w, int c) {
  const int pos = w * c;
  const uint32_t win = lds_bits(sw + tid, 256, pos, c);
  const uint32_t blo = pos ? lds_bits(sw + tid, 256, pos - 1, 1) : 0u;
  const uint32_t top = (win >> (c - 1)) & 1u;
  return (int32_t)win + (int32_t)blo - (int32_t)(top << c);
}
// exclusive scan of v over the 256 threads of a block (LDS scratch of 256 words); returns the prefix of this thread
__device__ __forceinline__ uint32_t block256_exclusive_scan(uint32_t v, uint32_t* scratch) {
  const uint32_t t = threadIdx.x;
  scratch[t] = v;
  __syncthreads();
  for (uint32_t o = 1; o < 256; o <<= 1) {
    const uint32_t y = t >= o ? scratch[t - o] : 0u;
    __syncthreads();
    scratch[t] += y;
    __syncthreads();
  }
  return scratch[t] - v;
}
// the same over the four waves of a block with wave shuffles: one barrier instead of sixteen (scratch: 4 words); *total = the block's sum
__device__ __forceinline__ uint32_t block256_exclusive_scan_shfl(uint32_t v, uint32_t* scratch, uint32_t* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) scratch[wid] = x;
  __syncthreads();
  const uint32_t s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
  *total = s0 + s1 + s2 + s3;
  const uint32_t before = wid == 0 ? 0u : wid == 1 ? s0 : wid == 2 ? s0 + s1 : s0 + s1 + s2;
  return before + x - v;
}
// dynamic LDS of the placing pass: scalar words, partition histogram / local starts / global bases, scan scratch, the block's
// (key, value) pairs (W per scalar); of the counting pass: scalar words and the histogram
inline size_t part_place_lds(uint32_t n_parts, int W) { return sizeof(uint32_t) * (24u * 256u + 3u * (size_t)n_parts + 256u + 2u * 256u * (size_t)W); }
constexpr size_t PART_LDS_LIMIT = 160u * 1024u;
template <int FRM, bool PLACE>
__global__ void __launch_bounds__(256) k_part_pass(const uint32_t* __restrict__ scal_wire, const uint8_t* __restrict__ inf, size_t n, int c, int W,
                                                  uint32_t hist_stride, uint32_t entry_stride, uint32_t entry_base, uint32_t n_parts,
                                                  uint32_t* __restrict__ part_total, uint32_t* __restrict__ part_cursor,
                                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  extern __shared__ uint32_t part_lds[];
  uint32_t* sw = part_lds;                       // [24][256]
  uint32_t* hist = sw + 24 * 256;                // [n_parts]
  const int tid = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
  scalar_to_lds<FRM>(scal_wire, inf, i, n, sw, tid);
  for (uint32_t p = tid; p < n_parts; p += 256) hist[p] = 0;
  __syncthreads();
  for (int w = 0; w < W; ++w) {
    const int32_t d = booth_digit(sw, tid, w, c);
    if (d) atomicAdd(&hist[((uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1u) >> PART_BITS], 1u);
  }
  __syncthreads();
  if constexpr (!PLACE) {
    for (uint32_t p = tid; p < n_parts; p += 256) { const uint32_t cnt = hist[p]; if (cnt) atomicAdd(&part_total[p], cnt); }
  } else {
    // The block's pairs are first ordered by partition in LDS and then written out: consecutive threads write consecutive entries of
    // one partition's range.  Written straight from the digit loop every lane of a store went to a different partition (64 partial
    // sectors per instruction): 0.55 ms for 318 MB.
    uint32_t* lstart = hist + n_parts;           // [n_parts] start of the partition inside the block's ordered list
    uint32_t* base = lstart + n_parts;           // [n_parts] start of the block's range inside the partition (global)
    uint32_t* scratch = base + n_parts;          // [256]
    uint32_t* st_k = scratch + 256;              // [256 * W]
    uint32_t* st_v = st_k + 256u * (uint32_t)W;
    // local exclusive scan of the partition counts: thread t owns partitions [t * per, (t + 1) * per)
    const uint32_t per = (n_parts + 255u) / 256u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t p = tid * per + k; if (p < n_parts) mine += hist[p]; }
    uint32_t run = block256_exclusive_scan(mine, scratch);
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t p = tid * per + k;
      if (p < n_parts) {
        const uint32_t cnt = hist[p];
        lstart[p] = run; run += cnt;
        base[p] = cnt ? atomicAdd(&part_cursor[p], cnt) : 0u;
        hist[p] = 0;
      }
    }
    __syncthreads();
    const uint32_t block_total = scratch[255];
    for (int w = 0; w < W; ++w) {
      const int32_t d = booth_digit(sw, tid, w, c);
      if (!d) continue;
      const uint32_t key = (uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1u;
      const uint32_t p = key >> PART_BITS;
      const uint32_t j = lstart[p] + atomicAdd(&hist[p], 1u);
      st_k[j] = key;
      st_v[j] = ((uint32_t)w * entry_stride + entry_base + (uint32_t)i) | (d < 0 ? 0x80000000u : 0u);
    }
    __syncthreads();
    for (uint32_t j = tid; j < block_total; j += 256) {
      const uint32_t key = st_k[j], p = key >> PART_BITS;
      const uint32_t pos = base[p] + (j - lstart[p]);
      keys[pos] = key;
      vals[pos] = st_v[j];
    }
  }
}
// ---- the same pass with the window width a template parameter (round 6) ----------------------------------------------------------
// k_part_pass reads its digits out of LDS (lds_bits: the position is a run-time value, registers cannot be indexed by one) and
// extracts every digit twice, once to count and once to place: ~25 of its ~60 instructions per digit, and with the scalar words
// (24.5 KB) and a (key, value) pair per digit (82 KB at W = 40) one workgroup per CU -- one wave per SIMD, every LDS atomic and
// global access paid in full latency: 0.41 ms for 100 MB read + 335 MB written (1 TB/s) at 2^20 scalars.  With C known at compile
// time the loop over the windows unrolls: the scalar stays in 24 registers, a digit is two shifts and a mask and is kept (W
// registers) from the counting loop to the placing loop, and the staged pair shrinks to one word -- low 10 bits of the key,
// window, owner thread, sign -- plus a 16-bit partition number: 65 KB at W = 40, two workgroups per CU.
template <int C>
__device__ __forceinline__ uint32_t reg_bits(const uint32_t (&s)[24], int pos, int nbits) {   // pos, nbits: constants after unrolling
  if (pos >= 768) return 0u;
  const int wi = pos >> 5, sh = pos & 31;
  uint32_t v = s[wi] >> sh;
  if (sh + nbits > 32 && wi + 1 < 24) v |= s[wi + 1] << (32 - sh);
  return v & ((1u << nbits) - 1u);
}
inline size_t part_place_lds_c(uint32_t n_parts, int W) { return sizeof(uint32_t) * (3u * (size_t)n_parts + 256u + 256u * (size_t)W) + sizeof(uint16_t) * 256u * (size_t)W; }
template <int FRM, int C, bool PLACE>
__global__ void __launch_bounds__(256) k_part_pass_c(const uint32_t* __restrict__ scal_wire, const uint8_t* __restrict__ inf, size_t n,
                                                    uint32_t hist_stride, uint32_t entry_stride, uint32_t entry_base, uint32_t n_parts,
                                                    uint32_t* __restrict__ part_total, uint32_t* __restrict__ part_cursor,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ scal_int) {
  constexpr int W = (754 + C - 1) / C;
  static_assert(W < 64, "the staged word holds the window in six bits");
  static_assert(W >= 24, "the integer scalars of the counting pass are parked in the sorted list's buffer: 24 words per scalar");
  extern __shared__ uint32_t part_lds[];
  uint32_t* hist = part_lds;                     // [n_parts]
  const int tid = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
  // The counting pass converts the scalar (wire Montgomery form -> integer: one product by a constant, ~2000 of its ~2800 instructions)
  // and parks the integer, zeroed for an identity base, where the sorted list will be written later; the placing pass reads it back.
  uint32_t s[24];
  if constexpr (!PLACE) {
    uint32_t w[24];
    if (i < n) {
      load_wire24(w, scal_wire + i * 24);
      fp_wire_to_integer<FRM>(s, w);
    }
    if (i >= n || inf[i]) {
#pragma unroll
      for (int j = 0; j < 24; ++j) s[j] = 0;
    }
    if (i < n) store_wire24(scal_int + i * 24, s);
  } else {
    if (i < n) load_wire24(s, scal_int + i * 24);
    else {
#pragma unroll
      for (int j = 0; j < 24; ++j) s[j] = 0;
    }
  }
  for (uint32_t p = tid; p < n_parts; p += 256) hist[p] = 0;
  __syncthreads();
  int32_t d[W];
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const uint32_t win = reg_bits<C>(s, w * C, C);
    const uint32_t blo = w ? reg_bits<C>(s, w * C - 1, 1) : 0u;
    const uint32_t top = (win >> (C - 1)) & 1u;
    d[w] = (int32_t)win + (int32_t)blo - (int32_t)(top << C);
    if (d[w]) atomicAdd(&hist[((uint32_t)w * hist_stride + (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1u) >> PART_BITS], 1u);
  }
  __syncthreads();
  if constexpr (!PLACE) {
    for (uint32_t p = tid; p < n_parts; p += 256) { const uint32_t cnt = hist[p]; if (cnt) atomicAdd(&part_total[p], cnt); }
  } else {
    uint32_t* lstart = hist + n_parts;           // [n_parts] start of the partition inside the block's ordered list
    uint32_t* base = lstart + n_parts;           // [n_parts] start of the block's range inside the partition (global)
    uint32_t* scratch = base + n_parts;          // [256]
    uint32_t* st = scratch + 256;                // [256 * W] key & 1023 | window << 10 | owner thread << 16 | sign << 24
    uint16_t* st_p = reinterpret_cast<uint16_t*>(st + 256u * (uint32_t)W);   // [256 * W] partition
    const uint32_t per = (n_parts + 255u) / 256u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t p = tid * per + k; if (p < n_parts) mine += hist[p]; }
    uint32_t block_total;
    uint32_t run = block256_exclusive_scan_shfl(mine, scratch, &block_total);
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t p = tid * per + k;
      if (p < n_parts) {
        const uint32_t cnt = hist[p];
        lstart[p] = run; run += cnt;
        base[p] = cnt ? atomicAdd(&part_cursor[p], cnt) : 0u;
        hist[p] = 0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (!d[w]) continue;
      const uint32_t key = (uint32_t)w * hist_stride + (uint32_t)(d[w] < 0 ? -d[w] : d[w]) - 1u;
      const uint32_t p = key >> PART_BITS;
      const uint32_t j = lstart[p] + atomicAdd(&hist[p], 1u);
      st[j] = (key & (PART_BUCKETS - 1u)) | ((uint32_t)w << 10) | ((uint32_t)tid << 16) | (d[w] < 0 ? 1u << 24 : 0u);
      st_p[j] = (uint16_t)p;
    }
    __syncthreads();
    const uint32_t i0 = entry_base + blockIdx.x * 256u;
    for (uint32_t j = tid; j < block_total; j += 256) {
      const uint32_t v = st[j], p = st_p[j];
      const uint32_t pos = base[p] + (j - lstart[p]);
      keys[pos] = (p << PART_BITS) | (v & (PART_BUCKETS - 1u));
      vals[pos] = (((v >> 10) & 63u) * entry_stride + i0 + ((v >> 16) & 255u)) | ((v >> 24) << 31);
    }
  }
}
// exclusive scan of the partition totals (n_parts <= PART_MAX, one block): part_start[0 .. n_parts], cursor = copy of the starts
__global__ void __launch_bounds__(1024) k_part_scan(const uint32_t* __restrict__ part_total, uint32_t* __restrict__ part_start, uint32_t* __restrict__ part_cursor,
                                                   uint32_t n_parts) {
  __shared__ uint32_t tmp[1024];
  const uint32_t t = threadIdx.x;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint32_t p = 4 * t + k; v[k] = p < n_parts ? part_total[p] : 0u; s += v[k]; }
  tmp[t] = s;
  __syncthreads();
  for (uint32_t o = 1; o < 1024; o <<= 1) {
    const uint32_t y = t >= o ? tmp[t - o] : 0u;
    __syncthreads();
    tmp[t] += y;
    __syncthreads();
  }
  uint32_t ex = tmp[t] - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t p = 4 * t + k;
    if (p < n_parts) { part_start[p] = ex; part_cursor[p] = ex; }
    ex += v[k];
  }
  if (t == 1023) part_start[n_parts] = tmp[1023];
}
// one chunk of the partition-ordered (key, value) list; PLACE = false: bucket histogram, PLACE = true: entries to their padded places
template <bool PLACE>
__global__ void __launch_bounds__(1024) k_bucket_pass(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ part_start,
                                                     uint32_t n_parts, uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ sorted, uint32_t shift) {
  __shared__ uint32_t cnt[PART_BUCKETS];
  __shared__ uint32_t base[PLACE ? PART_BUCKETS : 1];
  const uint32_t total = part_start[n_parts];
  const uint32_t lo = blockIdx.x * SORT_CHUNK;
  if (lo >= total) return;
  const uint32_t hi = min(lo + SORT_CHUNK, total);
  // first partition that reaches into the chunk: largest p with part_start[p] <= lo
  uint32_t a = 0, b = n_parts - 1u;
  while (a < b) { const uint32_t mid = (a + b + 1u) >> 1; if (part_start[mid] <= lo) a = mid; else b = mid - 1u; }
  for (uint32_t p = a; p < n_parts; ++p) {
    const uint32_t ps = part_start[p], pe = part_start[p + 1];
    if (ps >= hi) break;
    const uint32_t s = max(ps, lo), e = min(pe, hi);
    if (s >= e) continue;
    const uint32_t b0 = p << PART_BITS;
    for (uint32_t k = threadIdx.x; k < PART_BUCKETS; k += 1024) cnt[k] = 0;
    __syncthreads();
    for (uint32_t k = s + threadIdx.x; k < e; k += 1024) atomicAdd(&cnt[keys[k] - b0], 1u);
    __syncthreads();
    if constexpr (!PLACE) {
      for (uint32_t k = threadIdx.x; k < PART_BUCKETS; k += 1024) { const uint32_t v = cnt[k]; if (v) atomicAdd(&hist[b0 + k], v); }
    } else {
      for (uint32_t k = threadIdx.x; k < PART_BUCKETS; k += 1024) {
        const uint32_t v = cnt[k];
        base[k] = v ? (offsets[b0 + k] << shift) + atomicAdd(&hist[b0 + k], v) : 0u;   // hist was zeroed again after the scan: entries placed so far
        cnt[k] = 0;
      }
      __syncthreads();
      for (uint32_t k = s + threadIdx.x; k < e; k += 1024) {
        const uint32_t kb = keys[k] - b0;
        sorted[base[kb] + atomicAdd(&cnt[kb], 1u)] = vals[k];
      }
    }
    __syncthreads();
  }
}
// The placing pass of level 2 with its entries ordered by bucket in LDS first, so that a bucket receives a run of consecutive entries
// from one chunk instead of single 4-byte writes (k_bucket_pass<true>: 0.54 ms for 159 MB).  Chunks of 8192 pairs (runs of ~8 entries,
// 60 KB of LDS: two workgroups per CU) since round 6: 0.18 ms at 2^20 G1 points against 0.23 with 16384 (one workgroup per CU; runs of
// ~16 entries did not pay for the lost overlap, profiles/r06/sort_stage_by_width.txt).
#ifndef MNT753_PLACE_CHUNK
#define MNT753_PLACE_CHUNK 8192
#endif
constexpr uint32_t PLACE_CHUNK = MNT753_PLACE_CHUNK;
constexpr size_t PLACE_LDS = sizeof(uint32_t) * (3u * PART_BUCKETS + PLACE_CHUNK) + sizeof(uint16_t) * PLACE_CHUNK;
__global__ void __launch_bounds__(1024) k_bucket_place_staged(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ part_start,
                                                             uint32_t n_parts, uint32_t* __restrict__ placed, const uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ sorted, uint32_t shift) {
  extern __shared__ uint32_t place_lds[];
  uint32_t* cnt = place_lds;                     // [1024]
  uint32_t* lstart = cnt + PART_BUCKETS;         // [1024]
  uint32_t* base = lstart + PART_BUCKETS;        // [1024]
  uint32_t* st_v = base + PART_BUCKETS;          // [PLACE_CHUNK]
  uint16_t* st_b = reinterpret_cast<uint16_t*>(st_v + PLACE_CHUNK);
  const uint32_t total = part_start[n_parts];
  const uint32_t lo = blockIdx.x * PLACE_CHUNK;
  if (lo >= total) return;
  const uint32_t hi = min(lo + PLACE_CHUNK, total);
  uint32_t a = 0, b = n_parts - 1u;
  while (a < b) { const uint32_t mid = (a + b + 1u) >> 1; if (part_start[mid] <= lo) a = mid; else b = mid - 1u; }
  const uint32_t t = threadIdx.x;               // 1024 threads = one per bucket of a partition
  for (uint32_t p = a; p < n_parts; ++p) {
    const uint32_t ps = part_start[p], pe = part_start[p + 1];
    if (ps >= hi) break;
    const uint32_t s = max(ps, lo), e = min(pe, hi);
    if (s >= e) continue;
    const uint32_t b0 = p << PART_BITS;
    // this thread's (at most PLACE_CHUNK / 1024) pairs, all loads in flight at once and each pair read once (round 6; before: keys
    // read in the counting loop and again, with the values, in the ranking loop -- three exposed latencies per segment instead of one)
    uint32_t kb[PLACE_CHUNK / 1024], vv[PLACE_CHUNK / 1024];
#pragma unroll
    for (uint32_t q = 0; q < PLACE_CHUNK / 1024; ++q) {
      const uint32_t k = s + t + q * 1024u;
      const bool in = k < e;
      kb[q] = in ? keys[k] - b0 : 0xffffffffu;
      vv[q] = in ? vals[k] : 0u;
    }
    cnt[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < PLACE_CHUNK / 1024; ++q) if (kb[q] != 0xffffffffu) atomicAdd(&cnt[kb[q]], 1u);
    __syncthreads();
    const uint32_t v = cnt[t];
    // the bucket's range is reserved (a global atomic whose value is needed) while the scan runs
    const uint32_t reserved = v ? (offsets[b0 + t] << shift) + atomicAdd(&placed[b0 + t], v) : 0u;
    uint32_t seg_total;
    const uint32_t ex = block_exclusive_scan(v, &seg_total);   // (1024 threads; ends with a barrier)
    lstart[t] = ex;
    base[t] = reserved;
    cnt[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < PLACE_CHUNK / 1024; ++q) {
      if (kb[q] == 0xffffffffu) continue;
      const uint32_t j = lstart[kb[q]] + atomicAdd(&cnt[kb[q]], 1u);
      st_v[j] = vv[q];
      st_b[j] = (uint16_t)kb[q];
    }
    __syncthreads();
    for (uint32_t j = t; j < e - s; j += 1024) {
      const uint32_t kb = st_b[j];
      sorted[base[kb] + (j - lstart[kb])] = st_v[j];
    }
    __syncthreads();
  }
}
// the unused tail of every bucket's last group of 2^shift entries
__global__ void __launch_bounds__(256) k_bucket_pad(const uint32_t* __restrict__ placed, const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, uint32_t shift,
                                                   uint32_t n_buckets) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_buckets) return;
  const uint32_t c = placed[b], start = offsets[b] << shift, end = offsets[b + 1] << shift;
  for (uint32_t k = start + c; k < end; ++k) sorted[k] = 0xffffffffu;   // ENTRY_EMPTY
}
}  // namespace

namespace mnt753 {
// part_ws: 3 * (PART_MAX + 1) u32 (totals, starts, cursors)
int msm_sort_partition(int frm, const uint32_t* d_scal, const uint8_t* d_inf, size_t n, const MsmPlan& p, uint32_t entry_stride, uint32_t entry_base,
                       uint32_t* keys_out, uint32_t* vals_out, uint32_t* part_ws, uint32_t* d_hist, uint32_t* d_offsets, uint32_t* d_cursor,
                       uint32_t* d_blocksums, uint32_t* d_total, uint32_t* d_sorted, hipStream_t st) {
  const size_t total = (size_t)p.W * n;
  const uint32_t n_parts = (p.n_buckets + PART_BUCKETS - 1u) >> PART_BITS;
  if (n_parts > PART_MAX || total >= 0xffffffffull) return set_error(MNT753_EINVAL, "msm_sort_partition: too many buckets or entries");
  uint32_t *part_total = part_ws, *part_start = part_ws + (PART_MAX + 1), *part_cursor = part_ws + 2 * (PART_MAX + 1);
  const unsigned gb = (unsigned)((n + 255) / 256);
  const uint32_t hs = p.pre ? 0u : p.nb;
  const uint32_t pshift = (uint32_t)p.pair_levels;
  HIP_TRY(hipMemsetAsync(part_total, 0, sizeof(uint32_t) * (PART_MAX + 1), st));
  HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * (size_t)p.n_buckets, st));
  // window width known to a template (14 .. 22: every width pick_precomp_bits / pick_window_bits hands out for sets this large):
  // digits from registers, one staged word per entry (k_part_pass_c); any other width, or MNT753_MSM_SORT=generic: k_part_pass
  const char* sort_env = getenv("MNT753_MSM_SORT");
  const bool generic_only = sort_env && !strcmp(sort_env, "generic");
  const bool by_width = !generic_only && p.c >= 14 && p.c <= 22 && part_place_lds_c(n_parts, p.W) <= PART_LDS_LIMIT;
  if (by_width) {
    const size_t lds_place = part_place_lds_c(n_parts, p.W), lds_count = sizeof(uint32_t) * (size_t)n_parts;
#define MNT753_PART_PASS_C(FRM, C)                                                                                                              \
  {                                                                                                                                             \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_pass_c<FRM, C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_LIMIT)); \
    hipLaunchKernelGGL((k_part_pass_c<FRM, C, false>), dim3(gb), dim3(256), lds_count, st, d_scal, d_inf, n, hs, entry_stride, entry_base, n_parts, part_total, part_cursor, keys_out, vals_out, d_sorted); \
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, st, part_total, part_start, part_cursor, n_parts);                                   \
    hipLaunchKernelGGL((k_part_pass_c<FRM, C, true>), dim3(gb), dim3(256), lds_place, st, d_scal, d_inf, n, hs, entry_stride, entry_base, n_parts, part_total, part_cursor, keys_out, vals_out, d_sorted); \
  }
#define MNT753_PART_PASS_W(C) case C: if (frm == MOD_A) MNT753_PART_PASS_C(MOD_A, C) else MNT753_PART_PASS_C(MOD_B, C) break;
    switch (p.c) {
      MNT753_PART_PASS_W(14) MNT753_PART_PASS_W(15) MNT753_PART_PASS_W(16) MNT753_PART_PASS_W(17) MNT753_PART_PASS_W(18)
      MNT753_PART_PASS_W(19) MNT753_PART_PASS_W(20) MNT753_PART_PASS_W(21) MNT753_PART_PASS_W(22)
      default: return set_error(MNT753_EINVAL, "msm_sort_partition: window width outside the instantiated range");
    }
#undef MNT753_PART_PASS_W
#undef MNT753_PART_PASS_C
  } else {
    const size_t lds_place = part_place_lds(n_parts, p.W), lds_count = sizeof(uint32_t) * (24u * 256u + (size_t)n_parts);
    if (lds_place > PART_LDS_LIMIT) return set_error(MNT753_EINVAL, "msm_sort_partition: plan does not fit the LDS staging");
    // the placing passes stage their pairs in up to 160 KB of dynamic LDS: the opt-in is per kernel AND device, the call costs
    // microseconds, so it is simply repeated on whatever device this MSM runs on
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(frm == MOD_A ? (const void*)&k_part_pass<MOD_A, true> : (const void*)&k_part_pass<MOD_B, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_LIMIT));
#define MNT753_PART_PASS(FRM, PLACE) hipLaunchKernelGGL((k_part_pass<FRM, PLACE>), dim3(gb), dim3(256), (PLACE) ? lds_place : lds_count, st, d_scal, d_inf, n, p.c, p.W, hs, entry_stride, entry_base, \
                                                        n_parts, part_total, part_cursor, keys_out, vals_out)
    if (frm == MOD_A) MNT753_PART_PASS(MOD_A, false); else MNT753_PART_PASS(MOD_B, false);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, st, part_total, part_start, part_cursor, n_parts);
    if (frm == MOD_A) MNT753_PART_PASS(MOD_A, true); else MNT753_PART_PASS(MOD_B, true);
#undef MNT753_PART_PASS
  }
  const unsigned gc = (unsigned)((total + SORT_CHUNK - 1) / SORT_CHUNK);   // worst case: every digit non-zero
  hipLaunchKernelGGL((k_bucket_pass<false>), dim3(gc), dim3(1024), 0, st, keys_out, vals_out, part_start, n_parts, d_hist, d_offsets, d_sorted, pshift);
  const unsigned nsb = (unsigned)(((size_t)p.n_buckets + SCAN_BLOCK - 1) / SCAN_BLOCK);
  hipLaunchKernelGGL(k_scan_blocks, dim3(nsb), dim3(SCAN_THREADS), 0, st, d_hist, d_offsets, d_blocksums, (size_t)p.n_buckets, pshift);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, d_blocksums, (size_t)nsb, d_total);
  hipLaunchKernelGGL(k_scan_finish, dim3(nsb), dim3(SCAN_THREADS), 0, st, d_offsets, d_cursor, d_blocksums, d_total, (size_t)p.n_buckets);
  HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * (size_t)p.n_buckets, st));   // now: entries placed per bucket
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_place_staged), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PLACE_LDS));
  hipLaunchKernelGGL(k_bucket_place_staged, dim3((unsigned)((total + PLACE_CHUNK - 1) / PLACE_CHUNK)), dim3(1024), PLACE_LDS, st, keys_out, vals_out, part_start, n_parts,
                     d_hist, d_offsets, d_sorted, pshift);
  if (pshift) hipLaunchKernelGGL(k_bucket_pad, dim3((p.n_buckets + 255) / 256), dim3(256), 0, st, d_hist, d_offsets, d_sorted, pshift, p.n_buckets);
  HIP_TRY(hipGetLastError());
  return 0;
}
size_t msm_sort_partition_ws_words() { return 3 * (PART_MAX + 1); }
bool msm_sort_partition_fits(uint32_t n_buckets, int W) {
  const uint32_t n_parts = (n_buckets + PART_BUCKETS - 1u) >> PART_BITS;
  return n_parts <= PART_MAX && part_place_lds(n_parts, W) <= PART_LDS_LIMIT;
}
}  // namespace mnt753
