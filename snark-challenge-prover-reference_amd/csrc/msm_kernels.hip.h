// Pippenger multi-scalar multiplication kernels for gfx950.
//
// Replaces libff::multi_exp_with_mixed_addition<G, Fr, multi_exp_method_BDLO12>
// (reference: depends/libff/libff/algebra/scalar_multiplication/multiexp.tcc:165-282, :402-496),
// called from B::multiexp_G1 / B::multiexp_G2 (libsnark/prover_reference_functions.cpp:247-266).
// The result is the same group element; the schedule is GPU-native:
//
//   k_bases_to_internal   wire affine bases -> device form (27x28-bit limbs, R'=2^756), once per base set
//   k_precompute_windows  window table 2^(cw) P_i (affine), once per base set: every window then indexes ONE bucket set
//   sort stage            Montgomery scalar -> integer (as_bigint), signed radix-2^c Booth digits, (bucket, entry) pairs; from 2^22
//                         entries the two-level counting sort of msm_sort.hip, below that k_scalar_digits / scan / k_scatter
//                         (histogram with atomics + counting sort); buckets padded to a multiple of 2^levels entries either way
//   k_pair_level (x 3)    batched-affine additions of adjacent entries inside every bucket on a regular slot tree, operands staged
//                         through LDS by row-cooperative LDS-DMA, one divstep inversion per lane batch (large sets only);
//                         k_pair_fix undoes the stand-in of cancelled pairs
//   k_bucket_accumulate   each lane sums exactly T consecutive entries (perfect SIMD balance for ANY digit
//                         distribution); whole buckets go straight to the bucket array, the first / last partial run of
//                         each lane goes to an edge array.  Straight-line mixed additions for base fields and the two-lane
//                         Fq2, the point VM (curve753.hip.h) otherwise
//   k_edge_tree_*         the edge pieces that belong to one bucket, summed by an in-place binary tree over the bucket's lane range
//                         (round 4; the pointer-jumping merge of rounds 1-3 left the product in round 5)
//   k_reduce_step*        bucket reduction by halving: c - 1 launches, each one group addition deep (T and the G_l of
//                         sum_b (b+1) B[b] = T + sum_l 2^l G_l): _line = straight-line additions (wide steps, base fields),
//                         _pair = two lanes per addition (middle steps), plain = the VM; k_reduce_collect gathers the c
//                         points the host combines
//   msm_flow.hip.h        one addition on a GROUP of 8 / 16 lanes, four products deep, for the launches that are one addition
//                         deep and narrow: k_reduce_step_flow (the narrowest steps), k_edge_nodes + k_edge_tree_level_list
//                         (the later levels of the edge merge, from a node list)
//   k_points_to_wire      -> wire form (projective, Montgomery R=2^768)
//   host                  only without the window table (small sets): Horner over the window sums
// The G2 instantiations of the point kernels run on lane-split extension fields (curve753.hip.h): 2 or 3 lanes per point.
#pragma once
#include <hip/hip_runtime.h>
#include "curve753.hip.h"

namespace mnt753 {

constexpr int FPS_WORDS = 28;            // storage words per base-field element (27 limbs + pad), 112 B
constexpr uint32_t EDGE_NONE = 0xffffffffu;

// one wave per SIMD for every point-arithmetic kernel: they keep their state in the 512-register file (two waves per SIMD at 256
// registers were measured 3x slower for the extension fields, round 1)
template <class C>
constexpr int vm_waves() { return 1; }

// ---- storage helpers ---------------------------------------------------------------------
template <int M>
__device__ __forceinline__ void fp_load(Fp<M>& r, const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v = q[i];
    r.l[4 * i] = v.x;
    r.l[4 * i + 1] = v.y;
    r.l[4 * i + 2] = v.z;
    if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w;
  }
}
template <int M>
__device__ __forceinline__ void fp_store(uint32_t* p, const Fp<M>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v;
    v.x = a.l[4 * i];
    v.y = a.l[4 * i + 1];
    v.z = a.l[4 * i + 2];
    v.w = (4 * i + 3 < NL) ? a.l[4 * i + 3] : 0u;
    q[i] = v;
  }
}
// The 28th storage word of an element is padding; the pairing levels keep two flags of a ROW (affine point) in the pad
// word of every component of its x coordinate: PF_EMPTY (the slot holds no point) and PF_NEG (the point is (x, -y)).
constexpr uint32_t PF_EMPTY = 1u, PF_NEG = 2u;
template <int M>
__device__ __forceinline__ uint32_t fp_load_flag(Fp<M>& r, const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint32_t flag = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v = q[i];
    r.l[4 * i] = v.x;
    r.l[4 * i + 1] = v.y;
    r.l[4 * i + 2] = v.z;
    if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w; else flag = v.w;
  }
  return flag;
}
template <int M>
__device__ __forceinline__ void fp_store_flag(uint32_t* p, const Fp<M>& a, uint32_t flag) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v;
    v.x = a.l[4 * i];
    v.y = a.l[4 * i + 1];
    v.z = a.l[4 * i + 2];
    v.w = (4 * i + 3 < NL) ? a.l[4 * i + 3] : flag;
    q[i] = v;
  }
}
// ---- thread -> (logical lane, component) ------------------------------------------------------------------
// One-lane fields: thread = logical lane.  Lane-split fields (FieldFp2S / FieldFp3S, curve753.hip.h): LANES adjacent
// threads form one logical lane and each of them loads / stores only its own component of every element, so the
// memory layout is identical and split and one-lane kernels can be mixed freely in one pipeline.
//   LANES = 2: threads (2j, 2j+1);   LANES = 3: lanes (3g, 3g+1, 3g+2) of a wave, lane 63 idles (21 triples per wave).
// All point kernels are launched with 256-thread blocks.
template <class F>
__device__ __forceinline__ uint32_t logical_lane() {
  if constexpr (F::LANES == 1) return blockIdx.x * blockDim.x + threadIdx.x;
  else if constexpr (F::LANES == 2) return (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  else {
    const uint32_t lane = threadIdx.x & 63u;
    if (lane == 63u) return 0xffffffffu;
    return (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 21u + lane / 3u;
  }
}
template <class F>
__device__ __forceinline__ uint32_t lane_comp() {
  if constexpr (F::LANES == 1) return 0u;
  else if constexpr (F::LANES == 2) return threadIdx.x & 1u;
  else return (threadIdx.x & 63u) % 3u;
}
// 256-thread blocks needed for n logical lanes (host side)
template <class F>
constexpr unsigned blocks_for(uint64_t n) {
  return F::LANES == 3 ? (unsigned)((n + 83) / 84) : (unsigned)((n * F::LANES + 255) / 256);
}

template <class F>
__device__ __forceinline__ void e_load(typename F::E& r, const uint32_t* p) {
  if constexpr (F::LANES == 1) {
#pragma unroll
    for (int k = 0; k < F::DEG; ++k) fp_load(F::comp(r, k), p + k * FPS_WORDS);
  } else {
    fp_load(r, p + lane_comp<F>() * FPS_WORDS);
  }
}
template <class F>
__device__ __forceinline__ void e_store(uint32_t* p, const typename F::E& a) {
  if constexpr (F::LANES == 1) {
#pragma unroll
    for (int k = 0; k < F::DEG; ++k) fp_store(p + k * FPS_WORDS, F::comp(a, k));
  } else {
    fp_store(p + lane_comp<F>() * FPS_WORDS, a);
  }
}
template <class C>
constexpr int aff_words() { return 2 * C::F::DEG * FPS_WORDS; }
template <class C>
constexpr int proj_words() { return 3 * C::F::DEG * FPS_WORDS; }
template <class C>
constexpr int wire_coord_words() { return 24 * C::F::DEG; }

template <class C>
__device__ __forceinline__ void proj_load(Proj<C>& P, const uint32_t* p) {
  e_load<typename C::F>(P.X, p);
  e_load<typename C::F>(P.Y, p + C::F::DEG * FPS_WORDS);
  e_load<typename C::F>(P.Z, p + 2 * C::F::DEG * FPS_WORDS);
}
template <class C>
__device__ __forceinline__ void proj_store(uint32_t* p, const Proj<C>& P) {
  e_store<typename C::F>(p, P.X);
  e_store<typename C::F>(p + C::F::DEG * FPS_WORDS, P.Y);
  e_store<typename C::F>(p + 2 * C::F::DEG * FPS_WORDS, P.Z);
}

__device__ __forceinline__ void load_wire24(uint32_t w[24], const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
}
__device__ __forceinline__ void store_wire24(uint32_t* p, const uint32_t w[24]) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 6; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ---- base conversion ------------------------------------------------------------------------
// wire affine (x then y, each DEG x 12 u64 little-endian Montgomery R=2^768; y == 0 encodes the
// identity, libsnark/serialization.hpp:84-111) -> device affine + identity flag
template <class C>
__global__ void __launch_bounds__(256) k_bases_to_internal(const uint32_t* __restrict__ wire, uint32_t* __restrict__ out,
                                                          uint8_t* __restrict__ inf, size_t n) {
  using F = typename C::F;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* src = wire + i * 2 * wire_coord_words<C>();
  uint32_t* dst = out + i * aff_words<C>();
  uint32_t yor = 0;
#pragma unroll 1
  for (int k = 0; k < 2 * F::DEG; ++k) {
    uint32_t w[24];
    load_wire24(w, src + 24 * k);
    if (k >= F::DEG) {
#pragma unroll
      for (int j = 0; j < 24; ++j) yor |= w[j];
    }
    Fp<F::MOD> v;
    fp_from_wire(v, w);
    fp_store(dst + k * FPS_WORDS, v);
  }
  inf[i] = (yor == 0) ? 1 : 0;
}

// ---- wave-aggregated atomics ---------------------------------------------------------------------------
// Digit distributions are skewed where it matters: the top window of a 753-bit scalar only takes the values
// {0,1,2}, and real witnesses repeat scalars.  Up to AGG_ROUNDS times, the lanes that share the bucket of
// the first pending lane issue ONE atomic for the whole group; whatever is left falls back to per-lane
// atomics (the uniform-digit case, where almost every lane has its own bucket).
constexpr int AGG_ROUNDS = 3;
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// returns the value of counter[key] before this lane's increment (lanes with !active return 0)
__device__ __forceinline__ uint32_t wave_atomic_inc(uint32_t* counter, uint32_t key, bool active) {
  uint32_t result = 0;
  unsigned long long pending = __ballot(active);
  const uint32_t lane = lane_id();
  // Probe before aggregating (round 6): every aggregation round is an atomic whose return value the next round waits for -- with far
  // more buckets than lanes (the usual window: 2^13 .. 2^17 buckets) three rounds served three lanes and cost three round trips in
  // front of the per-lane atomics, four dependent latencies per window (k_scalar_digits: 0.18 ms for 2^15 scalars).  If neither the
  // first nor the last pending lane shares its bucket with at least three others the wave goes straight to per-lane atomics: one
  // latency.  Skewed digits (the top window, repeated scalars, zeros and ones) still aggregate.
  int rounds = AGG_ROUNDS;
  if (pending) {
    const int first = __ffsll((long long)pending) - 1, last = 63 - __clzll((long long)pending);
    const uint32_t k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, first), k2 = (uint32_t)__builtin_amdgcn_readlane((int)key, last);
    if (__popcll(__ballot(active && key == k1)) < 4 && __popcll(__ballot(active && key == k2)) < 4) rounds = 0;
  }
#pragma unroll 1
  for (int r = 0; r < rounds && pending; ++r) {
    const int leader = __ffsll((long long)pending) - 1;
    const uint32_t lkey = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    const bool mine = active && key == lkey && ((pending >> lane) & 1ull);
    const unsigned long long grp = __ballot(mine);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&counter[lkey], (uint32_t)__popcll(grp));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (mine) result = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
    pending &= ~grp;
  }
  if (active && ((pending >> lane) & 1ull)) result = atomicAdd(&counter[key], 1u);
  return result;
}

// ---- scalars -> signed digits + histogram -----------------------------------------------------
// digit w of integer s (radix 2^c, Booth): d = s[cw .. cw+c) + s[cw-1] - 2^c * s[cw+c-1], |d| <= 2^(c-1)
__device__ __forceinline__ uint32_t lds_bits(const uint32_t* sw, int stride, int pos, int n) {
  // bits [pos, pos+n) of the 768-bit integer whose word j is sw[j*stride]; n <= 25
  if (pos >= 768) return 0;
  int wi = pos >> 5, sh = pos & 31;
  uint64_t lo = sw[wi * stride];
  uint64_t hi = (wi + 1 < 24) ? sw[(wi + 1) * stride] : 0u;
  uint64_t v = (lo | (hi << 32)) >> sh;
  return (uint32_t)v & ((1u << n) - 1u);
}

template <int FRM>
__global__ void __launch_bounds__(256) k_scalar_digits(const uint32_t* __restrict__ scal_wire, const uint8_t* __restrict__ inf,
                                                      int32_t* __restrict__ digits, uint32_t* __restrict__ rank,
                                                      uint32_t* __restrict__ hist, size_t n, int c, int W, uint32_t hist_stride) {
  __shared__ uint32_t sw[24 * 256];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int tid = threadIdx.x;
  bool live = i < n;
  if (live) {
    uint32_t w[24], s[24];
    load_wire24(w, scal_wire + i * 24);
    fp_wire_to_integer<FRM>(s, w);
    if (inf[i]) {
#pragma unroll
      for (int j = 0; j < 24; ++j) s[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 24; ++j) sw[j * 256 + tid] = s[j];
  }
  // each lane only reads back its own column: no barrier needed; dead lanes keep taking part in the ballots
  const uint32_t nb = 1u << (c - 1);
  // The windows of a scalar are walked one atomic (whose return value is needed) after the other: ~2 us each, 54 of them for an
  // Fq3 set -- 0.11 ms of a 1.1 ms MSM over 4096 points, on 16 workgroups.  Small inputs split the walk over gridDim.y workgroups
  // per 256 scalars (each converts its scalars again: one product); the ranks an entry gets differ, the buckets' contents do not.
  const int per = (W + (int)gridDim.y - 1) / (int)gridDim.y, w_begin = (int)blockIdx.y * per, w_end = min(W, w_begin + per);
  for (int w = w_begin; w < w_end; ++w) {
    int32_t d = 0;
    if (live) {
      int pos = w * c;
      uint32_t win = lds_bits(sw + tid, 256, pos, c);
      uint32_t blo = pos ? lds_bits(sw + tid, 256, pos - 1, 1) : 0u;
      uint32_t top = (win >> (c - 1)) & 1u;
      d = (int32_t)win + (int32_t)blo - (int32_t)(top << c);
      digits[(size_t)w * n + i] = d;
    }
    uint32_t b = d ? (uint32_t)(d < 0 ? -d : d) - 1u : 0u;
    // the value the histogram atomic returns is the entry's rank inside its bucket: kept, so that the scatter needs no
    // second round of 38 * N atomics
    const uint32_t rk = wave_atomic_inc(hist + (size_t)w * hist_stride, b, d != 0);
    if (d != 0) rank[(size_t)w * n + i] = rk;
  }
}

// ---- exclusive scan (three small kernels) ------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  if (wid == 0) {
    uint32_t s = lane < SCAN_THREADS / 64 ? wsum[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      uint32_t y = __shfl_up(s, o, 64);
      if (lane >= o) s += y;
    }
    if (lane < SCAN_THREADS / 64) wsum[lane] = s;
  }
  __syncthreads();
  uint32_t base = wid ? wsum[wid - 1] : 0;
  *total = wsum[SCAN_THREADS / 64 - 1];
  __syncthreads();
  return base + x - v;
}

// scans ceil(in[i] / 2^shift): with shift = L the offsets count groups of 2^L entries -- the slot layout of L pairing levels
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                             uint32_t* __restrict__ block_sums, size_t n, uint32_t shift) {
  size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
  const uint32_t rnd = (1u << shift) - 1u;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? ((in[base + k] + rnd) >> shift) : 0u;
    s += v[k];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(s, &total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one block: exclusive scan of up to SCAN_BLOCK block sums, in place; writes the grand total to *total_out
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(uint32_t* __restrict__ sums, size_t nblocks, uint32_t* __restrict__ total_out) {
  size_t base = (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < nblocks) ? sums[base + k] : 0u;
    s += v[k];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(s, &total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < nblocks) sums[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) *total_out = total;
}

// offsets[i] += block base; also duplicates into cursor[]; offsets[n] = total
static __global__ void __launch_bounds__(SCAN_THREADS) k_scan_finish(uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                             const uint32_t* __restrict__ block_sums,
                                                             const uint32_t* __restrict__ total, size_t n) {
  size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t add = block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) {
      uint32_t o = offsets[base + k] + add;
      offsets[base + k] = o;
      cursor[base + k] = o;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = *total;
}

// ---- scatter: counting sort of (point, sign) by flattened bucket id ----------------------------
// hist_stride = 2^(c-1) (one bucket set per window) or 0 (all windows share one bucket set: precomputed tables);
// the sorted entry is the row index  w * entry_stride + entry_base + i  of the base table, plus the sign bit.
// shift = L > 0: `offsets` counts groups of 2^L entries (k_scan_blocks) and every bucket's entries start at a multiple of 2^L in
// `sorted` (the unused tail of a bucket's last group keeps the ENTRY_EMPTY the buffer was filled with).
static __global__ void __launch_bounds__(256) k_scatter(const int32_t* __restrict__ digits, const uint32_t* __restrict__ rank,
                                                const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, size_t n, int c, int W,
                                                uint32_t hist_stride, uint32_t entry_stride, uint32_t entry_base, uint32_t shift) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // small inputs: the walk over the windows split over gridDim.y workgroups, as in k_scalar_digits
  const int per = (W + (int)gridDim.y - 1) / (int)gridDim.y, w_begin = (int)blockIdx.y * per, w_end = min(W, w_begin + per);
  for (int w = w_begin; w < w_end; ++w) {
    const int32_t d = digits[(size_t)w * n + i];
    if (d == 0) continue;
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
    const uint32_t pos = (offsets[(size_t)w * hist_stride + b] << shift) + rank[(size_t)w * n + i];   // no atomics: k_scalar_digits kept the rank
    sorted[pos] = ((uint32_t)w * entry_stride + entry_base + (uint32_t)i) | (d < 0 ? 0x80000000u : 0u);
  }
}

// ---- bucket accumulation ------------------------------------------------------------------------
// Lane t owns sorted entries [t*T, (t+1)*T).  Runs of equal bucket id inside the segment are summed
// with mixed additions.  A run that is neither the first nor the last of the segment is a whole
// bucket -> written to buckets[].  The first and the last run may continue in a neighbouring lane
// -> written to edges[2t], edges[2t+1] with their bucket ids.
// BLOCKED = false: `bases` holds row-major rows (the window table, or the base set itself).
// BLOCKED = true:  `bases` holds the four blocked planes the last pairing level wrote ((x | y) x (even | odd slot), plane
//                  stride `plane_stride` uint4s, see k_pair_level): consecutive entries of a lane then sit in adjacent 16-byte
//                  pieces of the same sectors, and the level's stores are contiguous KiB instead of 64 partial sectors each.
__host__ __device__ __forceinline__ size_t blk_index(uint32_t j);
// (An LDS-DMA prefetch of the next entry's row was built in round 3 and measured neutral -- the kernel's clock follows its
// utilisation, profiles/r03/ab_acc_prefetch.txt -- and left the source in round 5.)
// acc += Q (Q affine) in a straight line: the same eleven products as the VM's program 0..10 (mixed_add, mnt4753_g1.cpp:265-313),
// without its switch machine.  pc == PC_MADD: add; anything else (PC_END): the lane keeps its value (it only took over Q) but runs the
// same instructions.  Equal points fall back to the VM's doubling.  Base fields and the two-lane Fq2 (k_bucket_accumulate).
// A macro, expanded in place in the kernel: the accumulate kernel's register allocation is fragile (finding 4) and the
// same statements behind a function call boundary -- even a force-inlined one -- came out with other spills (k_bucket_accumulate<Mnt6G1>
// 68 -> 168 B of scratch).  pt_madd_line() wraps the same text for the test hook (mnt753_test_point_op).
#define MNT753_MADD_LINE(C_, F_, acc_, Q_, pc_)                                                                        \
  do {                                                                                                                 \
    using E = typename F_::E;                                                                                          \
    E u, v, uu, vv, vvv, R, A, w, X3, Y3;                                                                              \
    F_::mul(w, acc_.Z, Q_.X); F_::sub(v, w, acc_.X);                                                                   \
    F_::mul(w, acc_.Z, Q_.Y); F_::sub(u, w, acc_.Y);                                                                   \
    const bool add = pc_ == PC_MADD;                                                                                   \
    const bool same = add && F_::is_zero(u) && F_::is_zero(v);                                                         \
    if constexpr (has_sqr<F_>::value) { F_::sqr(uu, u); F_::sqr(vv, v); } else { F_::mul(uu, u, u); F_::mul(vv, v, v); } \
    F_::mul(vvv, v, vv);                                                                                               \
    F_::mul(R, vv, acc_.X);                                                                                            \
    F_::mul(w, uu, acc_.Z);                                                                                            \
    F_::sub(A, w, vvv); F_::sub(A, A, R); F_::sub(A, A, R);      /* A = uu Z1 - vvv - 2R */                            \
    F_::mul(X3, v, A);                                                                                                 \
    F_::sub(w, R, A);                                                                                                  \
    F_::mul(w, u, w);                                                                                                  \
    F_::mul(R, vvv, acc_.Y);                                                                                           \
    F_::sub(Y3, w, R);                                                                                                 \
    F_::mul(w, vvv, acc_.Z);                                                                                           \
    if (add && !same) { acc_.X = X3; acc_.Y = Y3; acc_.Z = w; }                                                        \
    if (same) pt_vm<C_, false>(acc_, Q_, PC_DBL);                                                                      \
  } while (0)
template <class C>
__device__ __forceinline__ void pt_madd_line(Proj<C>& acc, const Proj<C>& Q, int pc) {
  using F = typename C::F;
  MNT753_MADD_LINE(C, F, acc, Q, pc);
}
template <class C, bool BLOCKED = false>
__global__ void __launch_bounds__(256, vm_waves<C>()) k_bucket_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                             const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                             uint32_t* __restrict__ buckets, uint32_t* __restrict__ edges,
                                                             uint32_t* __restrict__ edge_bucket, uint32_t T, uint32_t n_lanes,
                                                             size_t plane_stride = 0) {
  using F = typename C::F;
  const uint32_t t = logical_lane<typename C::F>();
  if (t >= n_lanes) return;
  const uint32_t total = offsets[n_buckets];
  // BLOCKED (behind the pairing levels): the host only knows the worst case of the list the levels leave; every lane walks its
  // entries one mixed addition after the other, so the kernel's time is the length of a lane's share -- taken from the actual total
  if constexpr (BLOCKED) T = max((total + n_lanes - 1u) / n_lanes, min(T, 8u));
  uint64_t e0 = (uint64_t)t * T;
  if (e0 >= total) {
    edge_bucket[2 * t] = EDGE_NONE;
    edge_bucket[2 * t + 1] = EDGE_NONE;
    return;
  }
  uint32_t e = (uint32_t)e0;
  uint32_t end = (e0 + T < total) ? (uint32_t)(e0 + T) : total;
  // largest b with offsets[b] <= e  (first x with offsets[x] > e, minus one)
  uint32_t lo = 0, hi = n_buckets;  // offsets[n_buckets] = total > e
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid + 1] > e) hi = mid; else lo = mid + 1;
  }
  uint32_t b = lo;
  uint32_t next = offsets[b + 1];
  bool first_run = true;
  bool acc_zero = true;
  Proj<C> acc, Q;
  pt_set_zero(acc);
  F::one(Q.Z);
  for (; e < end; ++e) {
    if (e == next) {
      // bucket b is finished inside this segment
      if (first_run) {
        proj_store<C>(edges + (size_t)(2 * t) * proj_words<C>(), acc);
        edge_bucket[2 * t] = b;
        first_run = false;
      } else {
        proj_store<C>(buckets + (size_t)b * proj_words<C>(), acc);
      }
      acc_zero = true;
      do { ++b; next = offsets[b + 1]; } while (next == e);
    }
    uint32_t s;
    if constexpr (BLOCKED) {
      s = sorted[e];
      static_assert(F::DEG == 1 || F::LANES > 1, "blocked rows hold one component per thread");
      // entries of the blocked list are (row << 1) | sign: with the sign in bit 31 hipcc 7.2 dropped the mask from
      // (s & 0x7fffffff) >> 7 in the address computation below (s >> 7 fed the 64-bit multiply-add; entries of negated points
      // faulted 120 GB past the planes)
      const uint32_t row = s >> 1;
      s <<= 31;
      const uint4* px = reinterpret_cast<const uint4*>(bases) + (size_t)(row & 1u) * plane_stride + blk_index((row >> 1) * F::LANES + lane_comp<F>());
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const uint4 vx = px[(size_t)i * 64], vy = px[2 * plane_stride + (size_t)i * 64];
        Q.X.l[4 * i] = vx.x; Q.X.l[4 * i + 1] = vx.y; Q.X.l[4 * i + 2] = vx.z;
        Q.Y.l[4 * i] = vy.x; Q.Y.l[4 * i + 1] = vy.y; Q.Y.l[4 * i + 2] = vy.z;
        if (4 * i + 3 < NL) { Q.X.l[4 * i + 3] = vx.w; Q.Y.l[4 * i + 3] = vy.w; }
      }
    } else {
      s = sorted[e];
      const uint32_t* src = bases + (size_t)(s & 0x7fffffffu) * aff_words<C>();
      e_load<F>(Q.X, src);
      e_load<F>(Q.Y, src + F::DEG * FPS_WORDS);
    }
    if (s & 0x80000000u) F::neg(Q.Y, Q.Y);
    int pc = PC_MADD;
    if (acc_zero) {
      acc.X = Q.X; acc.Y = Q.Y; F::one(acc.Z);
      acc_zero = false;
      pc = PC_END;
    } else if (pt_is_zero(acc)) {  // a run summed to the identity (P + -P): restart from Q
      acc.X = Q.X; acc.Y = Q.Y; F::one(acc.Z);
      pc = PC_END;
    }
    if constexpr ((F::LANES == 1 && F::DEG == 1) || F::LANES == 2) {
      // base fields and the two-lane Fq2 (-1.0 ms of 72; the three-lane Fq3 measured neutral): the mixed addition in a straight line (the same eleven products as the VM's program 0..10, without its
      // switch machine); lanes that only took over Q keep their value, equal points fall back to the VM's doubling
      MNT753_MADD_LINE(C, F, acc, Q, pc);
    } else {
      pt_vm<C, false>(acc, Q, pc);
    }
  }
  // last run of the segment
  if (first_run) {
    proj_store<C>(edges + (size_t)(2 * t) * proj_words<C>(), acc);
    edge_bucket[2 * t] = b;
    // single-run lane: the second slot is an identity piece of the same bucket (keeps the slot list gap-free)
    pt_set_zero(acc);
    proj_store<C>(edges + (size_t)(2 * t + 1) * proj_words<C>(), acc);
    edge_bucket[2 * t + 1] = b;
  } else {
    proj_store<C>(edges + (size_t)(2 * t + 1) * proj_words<C>(), acc);
    edge_bucket[2 * t + 1] = b;
  }
}

// ---- pairing levels: batched affine additions ahead of the accumulate ---------------------------------------
// L levels halve the sorted entry list L times before the projective accumulate.  An affine addition costs 3 products
// (lambda, lambda^2 -- a dedicated squaring on the base field --, y3) plus 3 for Montgomery's simultaneous-inversion trick
// plus a 1/B share of one divstep inversion (fp_inv.hip.h, ~94 product-equivalents) instead of the 11 of a mixed addition.
//
// Slot layout: k_scan_blocks / k_scatter place every bucket's entries at a multiple of 2^L (offsG counts groups of 2^L; the
// tail of a bucket's last group holds ENTRY_EMPTY), so the pairing is a REGULAR binary tree over slot indices: level-l slot o
// adds level-(l-1) slots 2o and 2o+1 -- no bucket walk, no offset lookups, and slot o of level L is final slot o of the bucket
// list the accumulate kernel reads with offsG.  Emptiness and the sign of y travel with the rows (PF_EMPTY / PF_NEG in the pad
// word of x), so no y is ever negated: for P1 = (x1, s1 y1), P2 = (x2, s2 y2)
//      s1 == s2:  lambda' = (y2 - y1) / (x2 - x1),  y3' = lambda' (x1 - x3) - y1,  result (x3,  s1 y3')
//      s1 != s2:  lambda' = (y2 + y1) / (x2 - x1),  y3' = lambda' (x1 - x3) + y1,  result (x3,  s2 y3')
// with x3 = lambda'^2 - x1 - x2 in both cases.
//
// Schedule: lane t handles slots t, t + NL, t + 2 NL, ... so adjacent lanes hold adjacent slots.  Forward sweep: x only,
// running product of the denominators, prefix products and the slot's kind to HBM; one inversion per lane; backward sweep:
// the sums.  One wave per SIMD: two were measured no faster (the stores below were the bottleneck, not latency) and halve
// the batch length per inversion.
//
// Memory layout -- what the first version of this pass got wrong.  A 16-byte access per lane with a 112- or 224-byte lane
// stride is 64 separate partial-sector requests per wave instruction; the 21 such stores per slot (prefix product + output
// row) cost ~16k cycles of a ~85k-cycle slot, for every occupancy and every amount of prefetching, because the CU's one
// address path serialises them.  Everything this pass writes for itself is therefore BLOCKED: element j (one Fp, 7 quads)
// keeps quad q at uint4 index ((j / 64) * 7 + q) * 64 + j % 64, so that the 64 lanes of a wave (consecutive j) move one
// contiguous KiB per instruction:
//   * prefix products: j = slot * LANES + component;
//   * rows handed to the NEXT pairing level: four planes (x | y) x (even | odd slot), j = (slot / 2) * LANES + component --
//     the reader's slot o' takes x1, y1 from the even planes and x2, y2 from the odd planes at j = o' * LANES + component,
//     the writer's lanes alternate between the two planes and still fill whole 64-byte sectors.
// Only level 1's input stays row-major (224-byte rows of the window table, gathered row-cooperatively); the accumulate kernel
// reads the last level's planes (a row-major last level cost 1.2 ms at 2^20 in 64-partial-sector stores).
// Side paths: an odd leftover is copied with its sign flag; equal points are doubled (denominator 2y); opposite points
// cancel: the slot gets the fixed point D (`gen`, the group generator) and fix_count[bucket] is incremented -- k_pair_fix
// subtracts fix_count * D from those buckets after the edge merge, so the accumulate kernel never sees an empty slot.
constexpr uint32_t ENTRY_EMPTY = 0xffffffffu;
// timing experiment only (results wrong by design): the first level's table rows folded into a span of 2^k rows, to measure what
// the width of the gathered range costs (tools/experiments/README.md, profiles/r05/level1_row_span.txt)
#ifdef MNT753_EXP_ROW_MASK
#define PAIR_ROW(r) ((r) & (uint32_t)(MNT753_EXP_ROW_MASK))
#else
#define PAIR_ROW(r) (r)
#endif
#ifndef MNT753_PAIR_WAVES
#define MNT753_PAIR_WAVES 1
#endif
#ifndef MNT753_PAIR_DIET
#define MNT753_PAIR_DIET 1
#endif
enum : uint32_t { PK_ADD = 0, PK_DBL = 1, PK_CANCEL = 2, PK_SINGLE = 3, PK_EMPTY = 4 };

// b + (negate ? -y : y) without a separate negation: the subtrahend / addend is chosen limb-wise
template <int M>
__device__ __forceinline__ void fp_addsub(Fp<M>& r, const Fp<M>& a, const Fp<M>& y, bool subtract) {
  uint32_t s[NL];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int32_t yi = subtract ? (int32_t)FPC[M].p2[i] - (int32_t)y.l[i] : (int32_t)y.l[i];
    const int32_t t = (int32_t)a.l[i] + yi + c;
    s[i] = (uint32_t)t & LMASK;
    c = t >> LB;
  }
  fp_reduce2p<M>(r, s);
}

// blocked element arrays (see above): uint4 index of quad 0 of element j; quad q is 64 uint4 further on
__host__ __device__ __forceinline__ size_t blk_index(uint32_t j) { return (size_t)(j >> 6) * (7 * 64) + (j & 63u); }
__host__ __device__ __forceinline__ size_t blk_quads(uint64_t n_elems) { return (size_t)((n_elems + 63) / 64) * (7 * 64); }   // uint4s of an array of n elements
template <int M>
__device__ __forceinline__ uint32_t fp_load_blk(Fp<M>& r, const uint4* __restrict__ base, uint32_t j) {
  const uint4* q = base + blk_index(j);
  uint32_t flag = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const uint4 v = q[(size_t)i * 64];
    r.l[4 * i] = v.x;
    r.l[4 * i + 1] = v.y;
    r.l[4 * i + 2] = v.z;
    if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w; else flag = v.w;
  }
  return flag;
}
template <int M>
__device__ __forceinline__ void fp_store_blk(uint4* __restrict__ base, uint32_t j, const Fp<M>& a, uint32_t flag) {
  uint4* q = base + blk_index(j);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v;
    v.x = a.l[4 * i];
    v.y = a.l[4 * i + 1];
    v.z = a.l[4 * i + 2];
    v.w = (4 * i + 3 < NL) ? a.l[4 * i + 3] : flag;
    q[(size_t)i * 64] = v;
  }
}

// ---- LDS staging of everything the sweeps read ----------------------------------------------------------------
// Gathering table rows one lane per row (7 x global_load_dwordx4 with a different row in every lane) reads 6 G rows/s on
// MI355X -- 64 address translations and 64 partial-sector requests per instruction -- and made level 1 gather-bound (the
// forward sweep took the same 25 k cycles per slot with or without its product).  Fetched ROW-COOPERATIVELY by LDS-DMA
// (global_load_lds_dwordx4: consecutive lanes fetch consecutive 16-byte pieces of a row, 4-9 rows per instruction, straight
// into LDS with no VGPR in between) the same rows arrive at 27 G rows/s (tools/experiments/gather_bench.hip).  The sweeps
// therefore read ALL their inputs through a per-wave LDS image that is filled one slot ahead of the arithmetic:
//     top of a slot:  s_waitcnt vmcnt(0)  ->  ds_read the slot's operands  ->  issue the LDS-DMA of the NEXT slot  ->  products
// One wave per SIMD, 36 KB of LDS per wave.  Image of a wave (uint4 units):
//     rows   first level: [point 0 | 1][slot of the wave][quads of the row (x | y) or of x only]   (packed, lane-linear fill)
//            later levels: [plane x-even | x-odd | y-even | y-odd][quad][thread]                    (own data, already coalesced)
//     pre    [quad][thread] (the slot's kind rides in its pad word)     entries [4 buffers][2 * slots of the wave]  (first level: fetched three slots ahead)
// The forward sweep only needs x: its half-size images alternate between the two halves of `rows`.  (Fetching them two slots ahead,
// s_waitcnt vmcnt(14) leaving the younger image in flight, was measured in round 2: the wait at the top of a slot is ~130 cycles
// either way; the variant left the source in round 5.)
constexpr uint32_t PAIR_IMG_QUADS = 1792;                       // 2 points x 64 lanes x 14 quads (every field: NS * RQ * 2 <= 1792)
constexpr uint32_t PAIR_LDS_WAVE_QUADS = PAIR_IMG_QUADS + 448 + 128;  // rows, prefix product, entries (4 x 128 u32)
constexpr uint32_t PAIR_LDS_BYTES = 4 * PAIR_LDS_WAVE_QUADS * 16;
// (Round 5 measured the other cut of the first level's LDS-DMA pieces: every instruction fetching WHOLE rows -- 4 rows x 14 quads on 56
// lanes of a full-row image, 9 rows x 7 quads on 63 lanes of an x image -- instead of 64 consecutive quads of the packed image, which
// cross rows (4.6 rows per instruction, every lane dividing by 14 to find its row).  Simpler addresses (121 fewer VALU instructions in
// the kernel, 162 fewer register moves), 32 + 15 instructions per slot instead of 28 + 14, some lanes idle: level 1 of the 2^20 G1 MSM
// 11.08 / 11.15 / 11.30 ms packed against 11.69 / 11.70 / 11.67 ms whole rows, alternating on one box
// (profiles/r05/level1_whole_row_pieces.txt) -- the number of DMA instructions is what the wave pays for, not their address arithmetic.)
// portions the LDS-DMA of the next slot's image is issued in, one ahead of each of the first products of a slot: gathered rows of a
// base field in four, of the lane-split fields in five, own planes in three (profiles/r03/ab_first_level_dma_portions.txt)
// (round 6, with the shorter step loop of the base fields, same box, per 2^20-point G1 MSM -- profiles/r06/g1_dma_portions_ab.txt: the
// regular later levels want TWO portions, 6.86 ms against 7.47 with three; the irregular ones three, 1.32 + 0.84 against 1.49 + 0.93
// with two; reading the prefix product where it is multiplied instead of with the other operands: 6.94 / 1.39 + 0.92)
#ifndef MNT753_PAIR_DMA_FIRST
#define MNT753_PAIR_DMA_FIRST 4
#endif
#ifndef MNT753_PAIR_DMA_LATER
#define MNT753_PAIR_DMA_LATER 2
#endif
#ifndef MNT753_PAIR_DMA_IRR
#define MNT753_PAIR_DMA_IRR 3
#endif
// (the lane-split fields keep three portions in their later levels: with two, MNT4753 G2 2^20 66.6-67.0 -> 68.9-69.6 ms, level 2 / 3
// 10.59 -> 11.6 ms each; MNT6753 G2 2^15 unchanged -- profiles/r06/g2_dma_portions_ab.txt)
constexpr uint32_t PAIR_DMA_STEPS_FIRST = MNT753_PAIR_DMA_FIRST, PAIR_DMA_STEPS_LATER = MNT753_PAIR_DMA_LATER, PAIR_DMA_STEPS_IRR = MNT753_PAIR_DMA_IRR;

#ifdef MNT753_PAIR_TIMING
// development: cycle totals of k_pair_level per wave (s_memtime): [0] forward, [1] inversion, [2] backward, [3] waves, [4..] ad hoc
__device__ unsigned long long g_pair_cycles[8];
__device__ unsigned int g_pair_slots;
#define PAIR_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PAIR_ACC(i, a, b) do { if ((threadIdx.x & 63u) == 0) atomicAdd(&g_pair_cycles[i], (unsigned long long)((b) - (a))); } while (0)
// time spent inside one statement, summed in a register: PAIR_TW(total, statement)
#define PAIR_TW_DECL(var) unsigned long long var = 0
#define PAIR_TW(var, ...) do { const unsigned long long t_a = __builtin_readcyclecounter(); __VA_ARGS__; var += __builtin_readcyclecounter() - t_a; } while (0)
#else
#define PAIR_T(var)
#define PAIR_ACC(i, a, b)
#define PAIR_TW_DECL(var)
#define PAIR_TW(var, ...) do { __VA_ARGS__; } while (0)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void glds16(const void* gsrc, const uint4* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds(gsrc, (lds_ptr_t)lds_dst_wave_uniform, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* gsrc, const uint32_t* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds(gsrc, (lds_ptr_t)lds_dst_wave_uniform, 4, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::: "memory"); }     // vmcnt(0)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); asm volatile("" ::: "memory"); }   // lgkmcnt(0)
template <int M>
__device__ __forceinline__ uint32_t fp_from_lds(Fp<M>& r, const uint4* p, uint32_t stride) {   // 7 quads at p, p + stride, ...
  uint32_t flag = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const uint4 v = p[(size_t)i * stride];
    r.l[4 * i] = v.x;
    r.l[4 * i + 1] = v.y;
    r.l[4 * i + 2] = v.z;
    if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w; else flag = v.w;
  }
  return flag;
}

// ---- irregular levels ---------------------------------------------------------------------------------------------------------
// The regular tree pads every bucket to 2^L entries, which is why it stops at L = 3 (76 entries per bucket at 2^20 points: 4.6 % of
// the level-1 slots are padding with 8-entry groups, 10 % with 16, 20 % with 32) and leaves ~10 slots per bucket to the projective
// accumulate -- 11 products an addition against 6.  An IRREGULAR level halves what a previous level left WITHOUT padding: a bucket
// with g slots gets ceil(g / 2) output slots, output slot o' of bucket b adds the input slots
//       s1 = offs_in[b] + 2 (o' - offs_out[b])   and   s1 + 1   (an odd leftover is copied),
// and the only irregularity the level kernel sees is one word per output slot, src[o'] = s1 | (leftover << 31), prepared by the
// three small kernels below (count / scan / fill).  The planes keep their layout (slot s in plane s & 1 at element s >> 1), so a
// wave's 64 consecutive output slots still read two runs of consecutive elements, broken only where a bucket with an odd count ends.
constexpr uint32_t IRR_BLOCK = 256;   // buckets per workgroup of the count / fill kernels
static __global__ void __launch_bounds__(IRR_BLOCK) k_irr_count(const uint32_t* __restrict__ offs_in, uint32_t n_buckets, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t part[IRR_BLOCK / 64];
  const uint32_t b = blockIdx.x * IRR_BLOCK + threadIdx.x;
  uint32_t c = b < n_buckets ? (offs_in[b + 1] - offs_in[b] + 1u) >> 1 : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// exclusive scan of the block sums in place (one workgroup; n_blocks <= a few thousand), total behind them
static __global__ void __launch_bounds__(1024) k_irr_scan(uint32_t* __restrict__ block_sums, uint32_t n_blocks) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((threadIdx.x & 63u) >= (uint32_t)d) x += y; }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    if (i < n_blocks) block_sums[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[n_blocks] = carry;
}
static __global__ void __launch_bounds__(IRR_BLOCK) k_irr_fill(const uint32_t* __restrict__ offs_in, uint32_t n_buckets, const uint32_t* __restrict__ block_sums,
                                                              uint32_t n_blocks, uint32_t* __restrict__ offs_out, uint32_t* __restrict__ src) {
  __shared__ uint32_t out0[IRR_BLOCK + 1], in0[IRR_BLOCK + 1], wsum[IRR_BLOCK / 64];
  const uint32_t b = blockIdx.x * IRR_BLOCK + threadIdx.x;
  const uint32_t lo = b < n_buckets ? offs_in[b] : offs_in[n_buckets];
  const uint32_t hi = b < n_buckets ? offs_in[b + 1] : lo;
  const uint32_t c = (hi - lo + 1u) >> 1;
  uint32_t x = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((threadIdx.x & 63u) >= (uint32_t)d) x += y; }
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t before = block_sums[blockIdx.x];
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
  const uint32_t o0 = before + x - c;
  out0[threadIdx.x] = o0;
  in0[threadIdx.x] = lo;
  if (threadIdx.x == IRR_BLOCK - 1) { out0[IRR_BLOCK] = o0 + c; in0[IRR_BLOCK] = hi; }
  if (b < n_buckets) offs_out[b] = o0;
  if (b == n_buckets - 1u) offs_out[n_buckets] = block_sums[n_blocks];
  __syncthreads();
  // the output slots of the workgroup's buckets, all threads together (a bucket of any length costs its share, no more)
  const uint32_t O0 = out0[0], O1 = out0[IRR_BLOCK];
  for (uint32_t o = O0 + threadIdx.x; o < O1; o += IRR_BLOCK) {
    uint32_t l = 0, h = IRR_BLOCK - 1u;              // largest i with out0[i] <= o (empty buckets share their successor's value)
    while (l < h) { const uint32_t mid = (l + h + 1u) >> 1; if (out0[mid] <= o) l = mid; else h = mid - 1u; }
    const uint32_t s1 = in0[l] + 2u * (o - out0[l]);
    src[o] = s1 | ((s1 + 1u == in0[l + 1u]) ? 0x80000000u : 0u);
  }
}

// first: sources are rows of `src_rows` (the window table, row-major) named by the padded entry list; otherwise the four
//        planes of the previous level at src_planes (plane stride src_stride uint4s).
// IRR:   (not first) an irregular level: offsG holds the level's OUTPUT offsets (shift = 0) and irr_src one word per output slot.
// last:  besides the four planes at out_planes (stride out_stride) the level writes the entry list out_sorted (slot o -> row o
//        of the planes as (o << 1) | sign) that the accumulate kernel (BLOCKED instantiation) reads.
template <class C, bool first, bool last, bool IRR = false>
__global__ void __launch_bounds__(256, MNT753_PAIR_WAVES) k_pair_level(const uint32_t* __restrict__ src_rows,
                                                      const uint32_t* __restrict__ entries, const uint4* __restrict__ src_planes,
                                                      size_t src_stride, const uint32_t* __restrict__ offsG, uint32_t n_buckets,
                                                      uint32_t shift, uint32_t* __restrict__ out_rows, uint32_t* __restrict__ out_sorted,
                                                      uint4* __restrict__ out_planes, size_t out_stride, uint4* __restrict__ prefix_ws,
                                                      uint32_t min_B, uint32_t n_lanes,
                                                      const uint32_t* __restrict__ gen, uint32_t* __restrict__ fix_count,
                                                      const uint32_t* __restrict__ irr_src = nullptr) {
  static_assert(!(IRR && first), "the first level reads the table through the entry list");
  using F = typename C::F;
  using E = typename F::E;                 // a single Fp: base field, or one component per lane of a lane-split field
  constexpr int M = F::MOD;
  constexpr int EW = F::DEG * FPS_WORDS;   // storage words of one element (row-major rows)
  constexpr int AW = aff_words<C>();
  constexpr uint32_t LN = F::LANES;
  constexpr uint32_t NS = 64u / LN;                       // slots of a wave (21 for three lanes per point)
  constexpr uint32_t RQ = 14u * F::DEG;                   // quads of a table row (x | y)
  constexpr uint32_t XQ = 7u * F::DEG;                    // quads of its x coordinate
  static_assert(2 * NS * RQ <= PAIR_IMG_QUADS, "row image");
  // first level of a base field: the table offsets of the next slot's pieces are read out of the entry image ahead of the
  // loads (per portion).  The lane-split fields keep a ds_read in front of every piece: their multiplier leaves no registers.
  constexpr bool PRELOAD = first && LN == 1;
  // backward sweep: the table offsets of the next slot read per portion (seven registers).  (For the lane-split fields the same
  // was measured WORSE in round 3 -- scratch appears, profiles/r03/ab_g2_split_preload.txt -- so they keep the per-piece read.)
  constexpr bool PRELOAD_BWD = first && LN == 1;
  // base fields: differences limb-wise without carries, signed-product multiplier, two normalisations per addition instead of
  // seven carry-propagating subtractions (fp753.hip.h, "lazy arithmetic")
  constexpr bool LAZY = has_lazy<F>::value;
  // round 6: the backward sweep keeps its operands where the multiplier reads them (below, "operands in place")
  constexpr bool DIET = LAZY && MNT753_PAIR_DIET;
  // (The same loop shape for the lane-split fields -- B <- A * B through their fused multiplier, one instance, lambda parked around the
  // squaring -- was built and measured in round 6: MNT4753 G2 2^20 67.4-67.8 ms against 67.2-67.4 with the loop below, MNT6753 G2 2^15
  // 8.5-8.7 against 8.65-8.8 (gpurun_out r6c, profiles/r06/g2_loop_shape_ab.txt): their products are 2187 / 2916 multiply-adds plus the
  // partner exchange, the routing moves a smaller share, and the first level lost what the later ones gained.  Not kept.)
  extern __shared__ uint4 pair_lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint4* img = pair_lds + (size_t)wave * PAIR_LDS_WAVE_QUADS;
  uint4* pre_img = img + PAIR_IMG_QUADS;
  uint32_t* ent_img = reinterpret_cast<uint32_t*>(pre_img + 448);   // [4][128]

  const uint32_t S = offsG[n_buckets] << shift;          // slots of this level (shift = levels still to come)
  // batch length from the ACTUAL number of slots (the host only knows the worst case): witness vectors full of zero and one
  // scalars leave a fraction of it, and a fixed B would leave most lanes idle behind a few long batches
  const uint32_t B = max(min_B, (S + n_lanes - 1u) / n_lanes);
  const uint32_t NLe = (S + B - 1u) / B;                  // lanes in use; slot = it * NLe + t
  // wave-uniform bookkeeping: every lane of a wave takes part in the cooperative loads, also lanes without a slot of their own
  const uint32_t t0w = (blockIdx.x * (blockDim.x >> 6) + wave) * NS;   // first logical lane of the wave
  if (t0w >= NLe || S == 0) return;
  const uint32_t t = logical_lane<F>();                   // 0xffffffff for the idle 64th lane of three-lane fields
  const uint32_t comp = lane_comp<F>();
  const uint32_t sl = LN == 3 ? min(lane / 3u, NS - 1u) : lane / LN;   // this thread's slot inside the wave
  const bool lane_on = t < NLe;
  const uint32_t n_it = (S - t0w + NLe - 1u) / NLe;       // iterations of the wave (its first lane has the most slots)
  const uint32_t cw = comp * FPS_WORDS;
  const uint4* table = reinterpret_cast<const uint4*>(src_rows);
  E run, x1, y1, x2, y2, den, tmp;
  F::one(run);

  // LDS-DMA of the entry pairs of the wave's slots at iteration `it` -> ent_img[buf]
  auto issue_entries = [=](uint32_t it, uint32_t buf) __attribute__((always_inline)) {
    const uint32_t ob = it * NLe + t0w;                   // first slot of the wave
    // 2 * NS u32, clamped to the padded list (slots >= S belong to no lane)
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      const uint32_t i = 64u * k + lane;
      const uint32_t slot = min(ob + (i >> 1), S - 1u);
      glds4(entries + 2 * (size_t)slot + (i & 1u), ent_img + buf * 128u + 64u * k);
    }
  };
  // LDS-DMA instruction k of the rows of iteration `it` (entries already in ent_img[buf]); xonly = x coordinates only (forward
  // sweep).  first: 2 * NS * rq quads in image order, 64 per instruction; later levels: one instruction per (plane, quad).
  // IRR: the per-lane part (uint4 units, plane of the slot's parity included) of the addresses of the two input slots a word of
  // irr_src names; an odd leftover names its one slot twice
  auto irr_bases = [=](uint32_t sw, uint32_t& va, uint32_t& vb) __attribute__((always_inline)) {
    const uint32_t s1 = sw & 0x7fffffffu, s2 = s1 + ((sw >> 31) ^ 1u);
    va = (s1 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index((s1 >> 1) * LN + comp);
    vb = (s2 & 1u) * (uint32_t)src_stride + (uint32_t)blk_index((s2 >> 1) * LN + comp);
  };
  // IRR: the word of the slot this thread (or, for idle lanes, the wave's first lane) has at iteration `it`
  auto irr_word = [=](uint32_t it) __attribute__((always_inline)) {
    return irr_src[min(it * NLe + (lane_on ? t : t0w), S - 1u)];
  };
  auto issue_row_piece = [=](uint32_t it, uint32_t buf, uint32_t k, auto xonly_c, uint4* im, uint32_t sw = 0u) __attribute__((always_inline)) {
    constexpr bool xonly = decltype(xonly_c)::value;
    if constexpr (IRR) {
      uint32_t va, vb;
      irr_bases(sw, va, vb);
      const uint32_t pl = k / 7u, q = k - pl * 7u;        // image planes: x1 | x2 | y1 | y2
      glds16(src_planes + (size_t)(pl & 2u) * src_stride + (size_t)q * 64 + ((pl & 1u) ? vb : va), im + k * 64u);
    } else if constexpr (first) {
      constexpr uint32_t rq = xonly ? XQ : RQ;
      constexpr uint32_t total = 2u * NS * rq;
      const uint32_t i = min(64u * k + lane, total - 1u);
      const uint32_t rs = i / rq, q = i - rs * rq;       // row slot = point * NS + slot
      const uint32_t p = rs >= NS ? 1u : 0u, s = rs - p * NS;
      const uint32_t e = ent_img[buf * 128u + 2u * s + p];
      const uint32_t r = e == ENTRY_EMPTY ? 0u : PAIR_ROW(e & 0x7fffffffu);
      glds16(table + (size_t)r * RQ + q, im + 64u * k);
    } else {
      const uint32_t o = min(it * NLe + (lane_on ? t : t0w), S - 1u);
      const uint32_t j = o * LN + comp;
      const uint32_t pl = k / 7u, q = k - pl * 7u;
      glds16(src_planes + blk_index(j) + (size_t)pl * src_stride + (size_t)q * 64, im + k * 64u);
    }
  };
  constexpr uint32_t ROW_PIECES_X = first ? (2u * NS * XQ + 63u) / 64u : 14u;
  constexpr uint32_t ROW_PIECES = first ? (2u * NS * RQ + 63u) / 64u : 28u;
  auto issue_rows = [=](uint32_t it, uint32_t buf, auto xonly_c, uint4* im, uint32_t sw = 0u) __attribute__((always_inline)) {
    constexpr uint32_t n = decltype(xonly_c)::value ? ROW_PIECES_X : ROW_PIECES;
#pragma unroll
    for (uint32_t k = 0; k < n; ++k) issue_row_piece(it, buf, k, xonly_c, im, sw);
  };
  // first level: the table offsets (uint4 units) of this lane's pieces of one slot, all read out of ent_img[buf] in one go.
  // A ds_read + s_waitcnt lgkmcnt(0) in front of EVERY LDS-DMA instruction cost a quarter of the level (the LDS queue is
  // busy with the DMA's own writes); the offsets of the next slot now sit in registers before the first piece is issued.
  auto row_offset = [=](uint32_t buf, uint32_t k) __attribute__((always_inline)) {      // piece k of a full-row image
    const uint32_t i = min(64u * k + lane, 2u * NS * RQ - 1u);
    const uint32_t rs = i / RQ, q = i - rs * RQ;
    const uint32_t p = rs >= NS ? 1u : 0u, s = rs - p * NS;
    const uint32_t e = ent_img[buf * 128u + 2u * s + p];
    return (e == ENTRY_EMPTY ? 0u : PAIR_ROW(e & 0x7fffffffu)) * RQ + q;
  };
  auto load_row_offsets = [=](uint32_t buf, auto xonly_c, uint32_t (&off)[ROW_PIECES]) __attribute__((always_inline)) {
    constexpr bool xonly = decltype(xonly_c)::value;
    constexpr uint32_t rq = xonly ? XQ : RQ;
    constexpr uint32_t total = 2u * NS * rq;
    constexpr uint32_t n = xonly ? ROW_PIECES_X : ROW_PIECES;
#pragma unroll
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t i = min(64u * k + lane, total - 1u);
      const uint32_t rs = i / rq, q = i - rs * rq;
      const uint32_t p = rs >= NS ? 1u : 0u, s = rs - p * NS;
      const uint32_t e = ent_img[buf * 128u + 2u * s + p];
      const uint32_t r = e == ENTRY_EMPTY ? 0u : PAIR_ROW(e & 0x7fffffffu);
      off[k] = r * RQ + q;
    }
  };
  // this thread's operands of the current slot, out of the image
  auto read_rows = [=](uint32_t buf, auto xonly_c, const uint4* im, uint32_t& f0, uint32_t& f1, E& x1, E& y1, E& x2, E& y2) __attribute__((always_inline)) {
    constexpr bool xonly = decltype(xonly_c)::value;
    if constexpr (first) {
      constexpr uint32_t rq = xonly ? XQ : RQ;
      const uint32_t e0 = ent_img[buf * 128u + 2u * sl], e1 = ent_img[buf * 128u + 2u * sl + 1u];
      f0 = e0 == ENTRY_EMPTY ? PF_EMPTY : (e0 >> 31) * PF_NEG;
      f1 = e1 == ENTRY_EMPTY ? PF_EMPTY : (e1 >> 31) * PF_NEG;
      const uint4* p0 = im + (size_t)sl * rq + comp * 7u;
      const uint4* p1 = im + (size_t)(NS + sl) * rq + comp * 7u;
      (void)fp_from_lds(x1, p0, 1u);
      (void)fp_from_lds(x2, p1, 1u);
      if constexpr (!xonly) { (void)fp_from_lds(y1, p0 + XQ, 1u); (void)fp_from_lds(y2, p1 + XQ, 1u); }
    } else {
      f0 = fp_from_lds(x1, im + lane, 64u);
      f1 = fp_from_lds(x2, im + 7u * 64u + lane, 64u);
      if constexpr (!xonly) { (void)fp_from_lds(y1, im + 14u * 64u + lane, 64u); (void)fp_from_lds(y2, im + 21u * 64u + lane, 64u); }
    }
  };

  // ---- forward: prefix products of the denominators, kinds
  PAIR_T(tc0);
  PAIR_TW_DECL(tw_a); PAIR_TW_DECL(tw_b); PAIR_TW_DECL(tw_c); PAIR_TW_DECL(tw_d);
  constexpr uint32_t XIMG = PAIR_IMG_QUADS / 2u;          // one x-only image (14 pieces of 64 quads)
  static_assert(ROW_PIECES_X * 64u <= XIMG, "x image");
  // x image of slot `it` -> half (it & 1) of the row image; first level: offsets from the entries in ent_img[it & 3]
  auto issue_x = [=](uint32_t it, uint32_t sw = 0u) __attribute__((always_inline)) {
    uint4* im = img + (it & 1u) * XIMG;
    issue_rows(it, it & 3u, std::true_type{}, im, sw);
  };
  constexpr uint32_t AH = 1u;                             // slots between the issue of an x image and its use
  // IRR: the source words of this slot and the next (the one after that is loaded while this slot's product runs)
  uint32_t sw_cur = 0u, sw_nxt = 0u;
  if constexpr (IRR) { sw_cur = irr_word(0); sw_nxt = irr_word(min(1u, n_it - 1u)); }
  if constexpr (first) {
    issue_entries(0, 0);
    issue_entries(min(1u, n_it - 1u), 1);
    wait_vm0();
  }
  issue_x(0, sw_cur);
  for (uint32_t it = 0; it < n_it; ++it) {
    const uint32_t o = it * NLe + t;
    const bool on = lane_on && o < S;
    uint32_t f0, f1;
    uint32_t sw_nn = 0u;
    wait_vm0();                                            // the image of this slot is complete
    read_rows(it & 3u, std::true_type{}, img + (it & 1u) * XIMG, f0, f1, x1, y1, x2, y2);
    if constexpr (IRR) { if (sw_cur >> 31) f1 = PF_EMPTY; }   // odd leftover: the second operand is the first one again
    const bool ahead = it + AH < n_it;
    uint32_t off[ROW_PIECES];
    if constexpr (PRELOAD) { if (ahead) load_row_offsets((it + AH) & 3u, std::true_type{}, off); }
    wait_lgkm0();
    // image of slot it + AH into the half it will be read from (entries first, the 14 row loads last): at once, the half is free
    auto issue_ahead = [=](const uint32_t (&off)[ROW_PIECES]) __attribute__((always_inline)) {
      uint4* im = img + ((it + AH) & 1u) * XIMG;
      if constexpr (first) issue_entries(min(it + AH + 1u, n_it - 1u), (it + AH + 1u) & 3u);
      if constexpr (PRELOAD) {
#pragma unroll
        for (uint32_t k = 0; k < ROW_PIECES_X; ++k) glds16(table + off[k], im + 64u * k);
      } else {
        issue_rows(it + AH, (it + AH) & 3u, std::true_type{}, im, sw_nxt);
      }
    };
    if (ahead) issue_ahead(off);
    if constexpr (IRR) { if (it + 2u < n_it) sw_nn = irr_word(it + 2u); }   // in flight during this slot's product
    uint32_t kind;
    if (!on || (f0 & PF_EMPTY)) kind = PK_EMPTY;
    else if (f1 & PF_EMPTY) kind = PK_SINGLE;
    else {
      kind = PK_ADD;
      bool same_x;
      if constexpr (LAZY) {
        F::sub_raw(den, x2, x1);
        same_x = F::raw_maybe_zero(den);       // three low-limb patterns in 2^28; settled exactly below
        if (same_x) { E du; F::sub(du, x2, x1); same_x = F::is_zero(du); }
      } else {
        F::sub(den, x2, x1);
        same_x = F::is_zero(den);
      }
      if (same_x) {
        // same x: equal points (doubling, denominator 2y) or opposite points (cancellation, take 1).  Rare: plain loads.
        if constexpr (first) {
          const uint2 e = reinterpret_cast<const uint2*>(entries)[o];
          fp_load(y1, src_rows + (size_t)PAIR_ROW(e.x & 0x7fffffffu) * AW + EW + cw);
          fp_load(y2, src_rows + (size_t)PAIR_ROW(e.y & 0x7fffffffu) * AW + EW + cw);
        } else if constexpr (IRR) {
          const uint32_t s1 = sw_cur & 0x7fffffffu, s2 = s1 + 1u;
          (void)fp_load_blk(y1, src_planes + (2 + (s1 & 1u)) * src_stride, (s1 >> 1) * LN + comp);
          (void)fp_load_blk(y2, src_planes + (2 + (s2 & 1u)) * src_stride, (s2 >> 1) * LN + comp);
        } else {
          (void)fp_load_blk(y1, src_planes + 2 * src_stride, o * LN + comp);
          (void)fp_load_blk(y2, src_planes + 3 * src_stride, o * LN + comp);
        }
        fp_addsub<M>(den, y1, y2, ((f0 ^ f1) & PF_NEG) != 0);      // s1 y1 + s2 y2 up to the sign s1
        if (F::is_zero(den)) { F::one(den); kind = PK_CANCEL; } else kind = PK_DBL;
      }
    }
    // the prefix product travels with the slot's kind in its pad word (read back by the thread that wrote it)
    PAIR_TW(tw_b, if (on) fp_store_blk(prefix_ws, o * LN + comp, run, kind));
    PAIR_T(tfa0);
    asm volatile("" ::: "memory");
#ifdef MNT753_PAIR_TIMING
    tw_a += __builtin_readcyclecounter() - tfa0;
#endif
    if (kind <= PK_CANCEL) {
      if constexpr (LAZY) F::mul_s(tmp, run, den); else F::mul(tmp, run, den);
      run = tmp;
    }
    if constexpr (IRR) { sw_cur = sw_nxt; sw_nxt = sw_nn; }
  }
  E inv;
  PAIR_T(tc1);
  if constexpr (LAZY) { F::norm(tmp, run); run = tmp; }   // signed top limb, (-0.3p, 1.3p) -> [0, 2p) for the inversion
  F::inv(inv, run);
  PAIR_T(tc2);
  // ---- backward: individual inverses and the sums (highest slot of the wave first)
  // piece `idx` of the LDS-DMA of iteration `it`: ROW_PIECES row pieces, then the 7 quads of the prefix product
  constexpr uint32_t BWD_PIECES = ROW_PIECES + 7u;
  auto issue_bwd_piece = [=](uint32_t it, uint32_t buf, uint32_t idx, uint32_t sw = 0u) __attribute__((always_inline)) {
    if (idx < ROW_PIECES) issue_row_piece(it, buf, idx, std::false_type{}, img, sw);
    else if (idx < BWD_PIECES) {
      const uint32_t o = min(it * NLe + (lane_on ? t : t0w), S - 1u);
      const uint32_t q = idx - ROW_PIECES;
      glds16(prefix_ws + blk_index(o * LN + comp) + (size_t)q * 64, pre_img + q * 64u);
    }
  };
  auto issue_bwd = [=](uint32_t it, uint32_t buf, uint32_t sw = 0u) __attribute__((always_inline)) {
    for (uint32_t idx = 0; idx < BWD_PIECES; ++idx) issue_bwd_piece(it, buf, idx, sw);
  };
  wait_vm0();                                            // the forward sweep's own stores (prefix products, kinds) are complete
  if constexpr (first) {
    issue_entries(n_it - 1u, 0);
    wait_vm0();
    issue_entries(n_it > 1u ? n_it - 2u : 0u, 1);
  }
  // IRR: the source words of this slot and of the next one down (the one after that is loaded during this slot's products)
  if constexpr (IRR) { sw_cur = irr_word(n_it - 1u); sw_nxt = irr_word(n_it > 1u ? n_it - 2u : 0u); }
  issue_bwd(n_it - 1u, 0, sw_cur);
  wait_vm0();
  for (uint32_t n = 0; n < n_it; ++n) {
    const uint32_t it = n_it - 1u - n;
    const uint32_t o = it * NLe + t;
    const bool on = lane_on && o < S;
    uint32_t f0, f1;
    E pre;
    uint32_t sw_nn = 0u;
    if constexpr (IRR) { if (it >= 2u) sw_nn = irr_word(it - 2u); }
    read_rows(n & 1u, std::false_type{}, img, f0, f1, x1, y1, x2, y2);
    if constexpr (IRR) { if (sw_cur >> 31) f1 = PF_EMPTY; }
    uint32_t kflag;
    kflag = fp_from_lds(pre, pre_img + lane, 64u);
    const uint32_t kind = on ? kflag : (uint32_t)PK_EMPTY;
    const bool more = n + 1u < n_it;
    // blocked index (uint4 units) of the next slot's element: the per-lane part of the addresses of its planes and prefix product
    const uint32_t vb_next = more ? (uint32_t)blk_index(min((it - 1u) * NLe + (lane_on ? t : t0w), S - 1u) * LN + comp) : 0u;
    uint32_t va_irr = 0u, vb_irr = 0u;        // IRR: per-lane parts of the next slot's two input slots
    if constexpr (IRR) { if (more) irr_bases(sw_nxt, va_irr, vb_irr); }
    wait_lgkm0();
    if constexpr (first) { if (more) issue_entries(it >= 2u ? it - 2u : 0u, n & 1u); }
    const bool flip = ((f0 ^ f1) & PF_NEG) != 0;
    uint32_t out_flag = PF_EMPTY;
    E num, ysave, yo;
    if constexpr (!DIET) {
      ysave = y1;
      if (kind == PK_ADD) {
        out_flag = f1 & PF_NEG;
        if constexpr (LAZY) {
          F::sub_raw(den, x2, x1);
          F::addsub_raw(num, y2, y1, !flip);
        } else {
          F::sub(den, x2, x1);
          fp_addsub<M>(num, y2, y1, !flip);         // y2 - y1  or  y2 + y1
        }
      } else if (kind == PK_DBL) {
        // P1 == P2 as signed points, s1 y1 = s2 y2: lambda = (3 x^2 + a) / (2 s1 y1) = s1 lambda' with the denominator 2 y1 formed
        // exactly as in the forward sweep (y1 + y2, or y1 - y2 when the flags differ); result (x3, s1 (lambda' (x1 - x3) - y1))
        E a;
        fp_addsub<M>(den, y1, y2, flip);
        F::mul(tmp, x1, x1);
        F::add(num, tmp, tmp); F::add(num, num, tmp);
        C::coeff_a(a);
        F::add(num, num, a);
        out_flag = f0 & PF_NEG;
      } else {
        // cancellation, odd leftover, empty slot: denominator 1 -- the lane runs the same products and its inversion chain
        // keeps its value (inv * 1)
        F::one(den);
        num = den;
      }
    }
    // The five products of a slot run through ONE inlined multiplier (and one squarer) in a wave-uniform step loop; every
    // result replaces an operand that is dead by then, which keeps the loop's live state at seven elements:
    //   0: pre <- inv * pre  (= 1 / den)      1: inv <- inv * den        2: den <- num * pre  (= lambda)
    //   3: x2 <- lambda^2 - x1 - x2 (= x3), num <- x1 - x3             4: y1 <- lambda * num -+ y1  (= y3')
    // Every lane runs every step (lanes without a pair carry denominator 1 and drop the results).  The LDS-DMA of the next
    // slot is issued in five portions, one ahead of every product: 35 DMA instructions in a row stall the wave for as long as
    // the address path needs to take them (~300 cycles each for gathered rows).
    {
      E opa, opb, res;
      // the image of the next slot is issued during the first DMA_STEPS products: gathered rows in five small portions (each
      // instruction holds the wave while the address path takes it), own planes in three (12.2 -> 11.1 ms and 5.36 -> 5.21 ms)
      // (first level, round 3: four portions for the base fields -- the last one then has two products to land in instead of one:
      // G1 2^20 25.04 / 24.93 -> 24.54 / 24.12 ms, three portions 24.45 / 24.35; the lane-split Fq2 wants five: 66.9 / 67.3 ms against
      // 68.5 / 68.5 with four; profiles/r03/ab_first_level_dma_portions.txt)
      constexpr uint32_t DMA_STEPS = first ? (LN == 1 ? PAIR_DMA_STEPS_FIRST : 5u) : (LN == 1 ? (IRR ? PAIR_DMA_STEPS_IRR : PAIR_DMA_STEPS_LATER) : 3u);
      constexpr uint32_t PER_STEP = (BWD_PIECES + DMA_STEPS - 1u) / DMA_STEPS;
      auto dma_step = [=](int step) __attribute__((always_inline)) {
        if (more && (uint32_t)step < DMA_STEPS) {
          // portion `step` of the next slot's image.  The pieces are named by constants (so that plane / quad offsets are scalar
          // constants), and the per-lane part of every address passes through an opaque move:
          // the 64-bit addresses are formed here, one VALU instruction each, not hoisted out of the step loop (35 live
          // addresses would not fit the register file) and not recomputed from the slot number either.
          auto portion = [=](auto step_c, uint32_t vb) __attribute__((always_inline)) {
            constexpr uint32_t base = decltype(step_c)::value * PER_STEP;
            uint32_t offp[PER_STEP];
            if constexpr (PRELOAD_BWD) {
              // the table offsets of this portion out of the entry image, one wait for all of them
#pragma unroll
              for (uint32_t u = 0; u < PER_STEP; ++u) offp[u] = base + u < ROW_PIECES ? row_offset((n + 1u) & 1u, base + u) : 0u;
              wait_lgkm0();
            }
#pragma unroll
            for (uint32_t u = 0; u < PER_STEP; ++u) {
              const uint32_t idx = base + u;
              if (idx < ROW_PIECES) {
                if constexpr (PRELOAD_BWD) {
                  uint32_t o = offp[u];
                  asm volatile("" : "+v"(o));
                  glds16(table + o, img + 64u * idx);
                } else if constexpr (first) {
                  uint32_t k = idx;
                  asm volatile("" : "+s"(k));
                  issue_row_piece(it - 1u, (n + 1u) & 1u, k, std::false_type{}, img);
                } else if constexpr (IRR) {
                  const uint32_t pl = idx / 7u, q = idx - pl * 7u;
                  uint32_t o = (pl & 1u) ? vb_irr : va_irr;
                  asm volatile("" : "+v"(o));
                  glds16(src_planes + (size_t)(pl & 2u) * src_stride + (size_t)q * 64 + o, img + 64u * idx);
                } else {
                  const uint32_t pl = idx / 7u, q = idx - pl * 7u;
                  uint32_t o = vb;
                  asm volatile("" : "+v"(o));
                  glds16(src_planes + (size_t)pl * src_stride + (size_t)q * 64 + o, img + 64u * idx);
                }
              } else if (idx < BWD_PIECES) {
                const uint32_t q = idx - ROW_PIECES;
                uint32_t o = vb;
                asm volatile("" : "+v"(o));
                glds16(prefix_ws + (size_t)q * 64 + o, pre_img + q * 64u);
              }
            }
          };
          switch (step) {
            case 0: portion(std::integral_constant<uint32_t, 0>{}, vb_next); break;
            case 1: portion(std::integral_constant<uint32_t, 1>{}, vb_next); break;
            case 2: portion(std::integral_constant<uint32_t, 2>{}, vb_next); break;
            case 3: portion(std::integral_constant<uint32_t, 3>{}, vb_next); break;
            default: portion(std::integral_constant<uint32_t, 4>{}, vb_next); break;
          }
        }
      };
      if constexpr (DIET) {
        // Operands in place (round 6).  The loop below (round 5's, still the lane-split fields') routes two operands into the multiplier
        // and one result out of it per product through a switch on either side: tools/isa_walk.py counts 1165 register moves per slot in
        // it (v_mov and AGPR traffic: the compiler copies both operand blocks at every merge point) of 9632 VALU instructions.  Here the
        // multiplier works on two blocks A, B with B <- A * B (fp_mul_s_ip), and every value is PRODUCED where its product reads it:
        //   0: A = inv, B = den = x2 - x1       B <- inv * den, kept as the next slot's inv (the one copy of the slot)
        //   1: B = pre                           B <- inv * pre = 1 / den
        //   2: A = num = y2 -+ y1               B <- num / den = lambda
        //   3: A <- B^2 (the squarer keeps B), x2 <- norm(A - x1 - x2) = x3, A <- x1 - x3;   B <- lambda * (x1 - x3)
        //   behind the loop: yo <- norm(B -+ y1) = y3'
        // 9029 instructions per slot by the walk; on the GPU 5.4 % fewer VALU instructions, level 1 11.2 -> 10.35 ms (profiles/r06/).
        // Measured and NOT kept: a second switch behind the multiplier (first form: -1.8 % instructions only), the prefix product read
        // from its image where it is multiplied (the ds_read queues behind the DMA's own LDS writes: irregular levels +6 %), results
        // pinned to their operand's registers by tied asm operands (+1 move per limb), the same shape for the lane-split fields.
        static_assert(DMA_STEPS <= 4u, "the loop has four iterations");
        E A = inv, B;
        if (kind == PK_ADD) {
          out_flag = f1 & PF_NEG;
          F::sub_raw(B, x2, x1);
        } else if (kind == PK_DBL) {
          // P1 == P2 as signed points, s1 y1 = s2 y2: lambda = (3 x^2 + a) / (2 s1 y1) = s1 lambda' with the denominator 2 y1 formed
          // exactly as in the forward sweep; result (x3, s1 (lambda' (x1 - x3) - y1)).  The numerator is handed to step 2 INSIDE y2
          // (y2 is dead for this lane): y2 <- N +- y1 limb-wise, so that the y2 -+ y1 of step 2 gives back the limbs of N
          E a, t, nn;
          fp_addsub<M>(B, y1, y2, flip);
          F::mul(t, x1, x1);
          F::add(nn, t, t); F::add(nn, nn, t);
          C::coeff_a(a);
          F::add(nn, nn, a);
          F::addsub_raw(y2, nn, y1, flip);
          out_flag = f0 & PF_NEG;
        } else {
          F::one(B);                              // cancellation, odd leftover, empty slot: the chain keeps its value
        }
        // ONE merge point per iteration: what a product leaves to do is done in front of the next one (the squaring, which has its own
        // instance, sits in the last iteration's preparation), the multiplier is the last thing in the body, and the loop's back edge
        // follows it directly -- with a second switch behind the multiplier the compiler copied both operand blocks twice per product.
#pragma nounroll
        for (int step = 0; step < 4; ++step) {
          dma_step(step);
          switch (step) {
            case 0: break;
            case 1: inv = B; B = pre; break;
            case 2: F::addsub_raw(A, y2, y1, !flip); break;    // y2 - y1  or  y2 + y1
            default:
              F::sqr_s_keep(A, B);
              F::sub_raw(A, A, x1);
              F::sub_raw(A, A, x2);
              F::norm(x2, A);
              F::sub_raw(A, x1, x2);
              break;
          }
          F::mul_s_ip(B, A);
        }
        F::addsub_raw(B, B, y1, !(kind == PK_ADD && flip));
        F::norm(B, B);
        yo = B;
      } else {
#pragma nounroll
        for (int step = 0; step < 5; ++step) {
          PAIR_T(tdm0);
          dma_step(step);
#ifdef MNT753_PAIR_TIMING
          tw_c += __builtin_readcyclecounter() - tdm0;
#endif
          switch (step) {
            case 0: opa = inv; opb = pre; break;
            case 1: opa = inv; opb = den; break;
            case 2: opa = num; opb = pre; break;
            case 3: opa = den; opb = den; break;
            default: opa = den; opb = num; break;
          }
          if constexpr (LAZY) {
            if (step == 3) F::sqr_s(res, opa); else F::mul_s(res, opa, opb);
          } else if constexpr (has_sqr<F>::value) {
            if (step == 3) F::sqr(res, opa); else F::mul(res, opa, opb);
          } else {
            F::mul(res, opa, opb);
          }
          switch (step) {
            case 0: pre = res; break;
            case 1: inv = res; break;
            case 2: den = res; break;
            case 3:
              if constexpr (LAZY) {
                // x3 = lambda^2 - x1 - x2 limb-wise, ONE normalisation; x1 - x3 stays raw (it only feeds the last product)
                F::sub_raw(res, res, x1);
                F::sub_raw(res, res, x2);
                F::norm(x2, res);
                F::sub_raw(num, x1, x2);
              } else {
                F::sub(res, res, x1);
                F::sub(x2, res, x2);
                F::sub(num, x1, x2);
              }
              break;
            default:
              if constexpr (LAZY) {
                F::addsub_raw(res, res, y1, !(kind == PK_ADD && flip));
                F::norm(y1, res);
              } else {
                fp_addsub<M>(y1, res, y1, !(kind == PK_ADD && flip));
              }
              break;
          }
        }
        yo = y1;
      }
    }
    if (kind == PK_SINGLE) {                    // odd leftover: copy, the sign travels in the flag
      out_flag = f0 & PF_NEG;
      x2 = x1;
      if constexpr (DIET) yo = y1; else yo = ysave;
    }
    {
      if (kind == PK_CANCEL) {                  // P + (-P): emit D, remember to take it out of the bucket again
        fp_load(x2, gen + cw);
        fp_load(yo, gen + EW + cw);
        out_flag = 0;
        if (comp == 0) {
          const uint32_t f = o >> shift;        // final slot -> bucket: largest b with offsG[b] <= f
          uint32_t lo = 0, hi = n_buckets - 1u;
          while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (offsG[mid] <= f) lo = mid; else hi = mid - 1u;
          }
          atomicAdd(&fix_count[lo], 1u);
        }
      }
    }
    // result (x2, y1) with out_flag (an empty slot only needs its flag; the coordinates written with it are never used).
    // The wait for the next slot's image stands BEFORE the stores: vmcnt counts loads and stores alike on gfx9, and a wait at
    // the top of the next slot would sit out the write latency of the fourteen stores issued just ahead of it.  Here it
    // covers the DMA (last portion one product old) and the previous slot's stores (a whole slot old).
    wait_vm0();
    PAIR_T(tst0);
    if (on) {
      const uint32_t j = (o >> 1) * LN + comp;
      uint4* px = out_planes + (size_t)(o & 1u) * out_stride;
      fp_store_blk(px, j, x2, out_flag);
      fp_store_blk(px + 2 * out_stride, j, yo, 0u);
      if constexpr (last) {   // the entry list of the accumulate kernel: slot o is row o of the planes
        if (comp == 0) out_sorted[o] = (out_flag & PF_EMPTY) ? ENTRY_EMPTY : ((o << 1) | ((out_flag & PF_NEG) ? 1u : 0u));
      }
    }
#ifdef MNT753_PAIR_TIMING
    tw_d += __builtin_readcyclecounter() - tst0;
#endif
    if constexpr (IRR) { sw_cur = sw_nxt; sw_nxt = sw_nn; }
  }
  PAIR_T(tc3);
  PAIR_ACC(0, tc0, tc1); PAIR_ACC(1, tc1, tc2); PAIR_ACC(2, tc2, tc3); PAIR_ACC(3, 0ull, 1ull);
#ifdef MNT753_PAIR_TIMING
  PAIR_ACC(4, 0ull, tw_a); PAIR_ACC(5, 0ull, tw_b); PAIR_ACC(6, 0ull, tw_c); PAIR_ACC(7, 0ull, tw_d);
  if ((threadIdx.x & 63u) == 0 && blockIdx.x == 0 && wave == 0) g_pair_slots = n_it;
#endif
}

template <class C>
__device__ __forceinline__ int add_pc(Proj<C>& P, const Proj<C>& Q);

// buckets[b] -= fix_count[b] * D for the buckets in which the pairing pass replaced a cancelled pair by D.
// k * D by double-and-add: an adversarial input (every base next to its negative) can put hundreds of thousands of
// cancellations into one bucket, and k sequential additions in one lane would take seconds.
template <class C>
__global__ void __launch_bounds__(256, 1) k_pair_fix(uint32_t* __restrict__ buckets, const uint32_t* __restrict__ fix_count,
                                                    const uint32_t* __restrict__ gen, uint32_t n_buckets) {
  using F = typename C::F;
  const uint32_t b = logical_lane<F>();
  if (b >= n_buckets) return;
  const uint32_t k = fix_count[b];
  if (k == 0) return;
  Proj<C> R, Q;
  e_load<F>(Q.X, gen);
  e_load<F>(Q.Y, gen + F::DEG * FPS_WORDS);
  F::neg(Q.Y, Q.Y);                          // -D
  F::one(Q.Z);
  R = Q;                                     // top bit of k
  for (int bit = 30 - __clz((int)k); bit >= 0; --bit) {
    pt_vm<C, true>(R, Q, PC_DBL);
    if ((k >> bit) & 1u) pt_vm<C, true>(R, Q, pt_is_zero(R) ? PC_END : PC_MADD);
  }
  Proj<C> acc;
  proj_load<C>(acc, buckets + (size_t)b * proj_words<C>());
  const int pc = add_pc<C>(acc, R);
  pt_vm<C, true>(acc, R, pc);
  proj_store<C>(buckets + (size_t)b * proj_words<C>(), acc);
}

// full projective addition with identity handling, through the VM
template <class C>
__device__ __forceinline__ int add_pc(Proj<C>& P, const Proj<C>& Q) {
  // returns the VM entry point for P += Q after resolving identities in place
  if (pt_is_zero(Q)) return PC_END;
  if (pt_is_zero(P)) { P = Q; return PC_END; }
  return PC_ADD;
}

// the VM's addition as a real call: its registers are then not part of the caller's allocation (the rare equal-points case of the
// two-lane addition: inlined, the three-lane kernels spilled 500 registers and the step ran 2.4x slower than through the VM; the tree
// level of the edge merge: see there)
template <class C>
__device__ __attribute__((noinline)) void pt_vm_add_outlined(Proj<C>& P, const Proj<C>& Q, int pc) { pt_vm<C, true>(P, Q, pc); }

// ---- edge merge as a tree over the lanes of a bucket (round 4) -----------------------------------------------------------------------
// The pointer-jumping merge of rounds 1-3 (every slot adds the slot 2^s further on if it belongs to the same bucket; out of the
// product since round 5) was general but paid for it: 2 log2(slots) dependent launches (34 for 65536 lanes: a sum into a
// temporary and a copy back per level), every one of them over ALL slots, and half the partners are the identity pieces that keep the
// slot list gap-free.  But where the pieces of a bucket are is no secret: bucket b holds the entries [o0, o1) of the list, lane t walks
// the entries [t T, (t + 1) T), so b has one piece in each of the lanes t_lo = o0 / T .. t_hi = (o1 - 1) / T -- the last run of lane t_lo
// (its first run if the bucket starts exactly with the lane) and the first run of every later lane.  Piece i of the bucket = lane
// t_lo + i.  A tree over i, IN PLACE: at level l (stride S = K^l) the piece with i % (K S) == 0 adds the pieces i + S, i + 2 S, .. into
// its own slot; after ceil(log_K(pieces)) levels piece 0 holds the bucket.  No temporary, no copy, every piece read once per level it
// takes part in, a bucket inside one lane costs nothing, the common bucket (two lanes) one addition; the levels after the last useful
// one leave at once.  K = 2: one addition deep per level, log2(lanes) launches (half the old count).  K = 8
// (six launches) was measured first and lost wherever buckets span several lanes -- its K - 1 additions per level run one after the
// other in one lane: MNT6753 G1 2^13 points with c = 14 (54 entries per bucket over ~7 lanes) 1.16 ms with pointer jumping, 3.9 ms with
// K = 8 (profiles/r04/small_msm_window_width_k8_tree.txt) -- and a witness full of ones puts half the list into one bucket.
constexpr uint32_t EDGE_TREE_K = 2;
// entries per lane the accumulate kernel really used (its BLOCKED form derives them from the actual length of the list)
__device__ __forceinline__ uint32_t acc_entries_per_lane(uint32_t T, uint32_t n_lanes, uint32_t total, bool blocked) {
  return blocked ? max((total + n_lanes - 1u) / n_lanes, min(T, 8u)) : T;
}
// slot of the piece lane t holds of the bucket that starts at entry o0 in lane t_lo
__device__ __forceinline__ uint32_t edge_piece_slot(uint32_t t, uint32_t t_lo, uint32_t o0, uint32_t T) {
  return (t == t_lo && (uint64_t)o0 > (uint64_t)t_lo * T) ? 2u * t + 1u : 2u * t;
}
// piece 0 of every bucket that left pieces in the edge slots now holds its sum
template <class C>
__global__ void __launch_bounds__(256) k_edge_tree_finish(const uint32_t* __restrict__ edges, const uint32_t* __restrict__ edge_bucket,
                                                         const uint32_t* __restrict__ offsets, uint32_t n_buckets, uint32_t T_arg, uint32_t n_lanes,
                                                         uint32_t blocked, uint32_t* __restrict__ buckets) {
  const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= 2u * n_lanes) return;
  const uint32_t b = edge_bucket[sidx];
  if (b == EDGE_NONE) return;
  const uint32_t T = acc_entries_per_lane(T_arg, n_lanes, offsets[n_buckets], blocked != 0);
  const uint32_t o0 = offsets[b];
  const uint32_t t = sidx >> 1, t_lo = o0 / T;
  if (t != t_lo || edge_piece_slot(t, t_lo, o0, T) != sidx) return;
  const uint4* src = reinterpret_cast<const uint4*>(edges + (size_t)sidx * proj_words<C>());
  uint4* dst = reinterpret_cast<uint4*>(buckets + (size_t)b * proj_words<C>());
#pragma unroll
  for (int q = 0; q < proj_words<C>() / 4; ++q) dst[q] = src[q];
}

// ---- bucket reduction ------------------------------------------------------------------------------
// sum_b (b + 1) B[b] over nb = 2^k buckets by HALVING, every step one group addition deep:
//     A_0 = B,   A_{l+1}[j] = A_l[2j] + A_l[2j+1],   G_l = sum of the odd-indexed elements of A_l        (l = 0 .. k-1)
// A_k[0] = T is the plain sum and  sum_b b B[b] = sum_l 2^l G_l  (bit l of b is set exactly for the elements that sit at an
// odd index of A_l), so the result is T + sum_l 2^l G_l -- a 19-step Horner over single points, done on the host by
// msm_finish (k doublings + k additions, < 0.1 ms; one GPU lane would need milliseconds).  G_l is a tree sum of 2^(k-l-1)
// elements; its first step reads A_l directly (A_l[4j+1] + A_l[4j+3]) in the SAME step that halves A_l, so tree l runs in
// steps l .. k-2 and ALL the trees finish together: k launches, about 2 * nb additions in total, every launch a flat list of
// independent additions (step s: 2^(k-s-1) halvings + (s+1) 2^(k-s-2) tree additions).  It replaces the chunked
// running-sum reduction (55 dependent additions per lane over 65536 lanes = 3.6 M additions and 17 launches).
// Storage per bucket set (points): A_1 .. A_k packed in nb slots (A_l at 2^k - 2^(k-l+1)); the trees ping-pong in another nb
// slots (tree l at 2^k - 2^(k-l), two halves of 2^(k-l-2)).
__host__ __device__ __forceinline__ uint32_t red_off_a(uint32_t k, uint32_t l) { return (1u << k) - (1u << (k - l + 1)); }   // l >= 1
__host__ __device__ __forceinline__ uint32_t red_off_g(uint32_t k, uint32_t l) { return (1u << k) - (1u << (k - l)); }
__host__ __device__ __forceinline__ uint32_t red_items(uint32_t k, uint32_t s) {   // additions of step s in one bucket set
  const uint32_t nh = 1u << (k - s - 1);
  return k >= s + 2 ? nh + (s + 1) * (1u << (k - s - 2)) : nh;
}
template <class C>
__global__ void __launch_bounds__(256, vm_waves<C>()) k_reduce_step(const uint32_t* __restrict__ buckets, const uint32_t* __restrict__ offsets,
                                                       uint32_t* __restrict__ A, uint32_t* __restrict__ G, uint32_t n_sets, uint32_t k, uint32_t s) {
  const uint32_t t = logical_lane<typename C::F>();
  const uint32_t per_set = red_items(k, s);
  if (t >= n_sets * per_set) return;
  constexpr int PW = proj_words<C>();
  const uint32_t set = t / per_set, r0 = t - set * per_set;
  const uint32_t nb = 1u << k, nh = 1u << (k - s - 1);
  const uint32_t* a_src = s == 0 ? buckets + (size_t)set * nb * PW : A + ((size_t)set * nb + red_off_a(k, s)) * PW;
  uint32_t i0, i1;
  const uint32_t* src;
  uint32_t* dst;
  if (r0 < nh) {                                   // halving: A_{s+1}[j] = A_s[2j] + A_s[2j+1]
    i0 = 2u * r0; i1 = i0 + 1u; src = a_src;
    dst = A + ((size_t)set * nb + red_off_a(k, s + 1) + r0) * PW;
  } else {
    const uint32_t sh = k - s - 2, r = r0 - nh, l = r >> sh, j = r & ((1u << sh) - 1u);
    uint32_t* g = G + ((size_t)set * nb + red_off_g(k, l)) * PW;
    const uint32_t half = 1u << (k - l - 2);
    if (l == s) {                                  // first step of tree s: odd elements of A_s
      i0 = 4u * j + 1u; i1 = i0 + 2u; src = a_src;
      dst = g + (size_t)j * PW;
    } else {                                       // tree l < s: one more level, ping-pong
      i0 = 2u * j; i1 = i0 + 1u; src = g + (size_t)(((s - l - 1u) & 1u) * half) * PW;
      dst = g + (size_t)(((s - l) & 1u) * half + j) * PW;
    }
  }
  Proj<C> acc, Q;
  // buckets no entry was sorted into hold stale data: they count as the identity (only A_0 = the bucket array has them)
  const bool from_buckets = s == 0 && src == a_src;
  const bool e0 = from_buckets && offsets[(size_t)set * nb + i0 + 1] == offsets[(size_t)set * nb + i0];
  const bool e1 = from_buckets && offsets[(size_t)set * nb + i1 + 1] == offsets[(size_t)set * nb + i1];
  if (e0) pt_set_zero(acc); else proj_load<C>(acc, src + (size_t)i0 * PW);
  if (e1) pt_set_zero(Q); else proj_load<C>(Q, src + (size_t)i1 * PW);
  const int pc = add_pc<C>(acc, Q);
  pt_vm<C, true>(acc, Q, pc);
  proj_store<C>(dst, acc);
}
// The same step with TWO lanes per addition, for the narrow steps (one addition deep whatever their width: 15 of the 19 steps
// of a 2^19-bucket reduction, 78 us each).  The 12 products + 2 squarings of a projective addition are five dependency
// levels deep; two lanes run them as 3 + 1 + 2 + 1 + 1 = 8 sequential products: the odd lane holds the operands swapped, so
// the first level is the same code in both lanes, later levels pick their operands by parity and exchange one element each
// (ds_bpermute with the neighbour lane).  Same group element as pt_vm<PC_ADD> (mnt4753_g1.cpp:134-207: add-1998-cmo-2);
// identities are resolved before, equal points (a doubling) fall back to the VM in the even half.  Base fields, and the
// two-lane Fq2 with four lanes per addition.
// the value the partner lane holds (src4 = partner lane * 4)
template <int M>
__device__ __forceinline__ void fp_pair_xchg(Fp<M>& r, const Fp<M>& a, int src4) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)a.l[i]);
}
template <int M>
__device__ __forceinline__ void fp_pick(Fp<M>& r, bool c, const Fp<M>& a, const Fp<M>& b) {   // c ? a : b
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
}
// Lanes of one addition: LN lanes hold the first point, the next LN lanes the second (LN = lanes per point: 1, 2, or 3 with 21
// triples per wave -- ten pairs of triples, lanes 60..63 idle).  pair_geometry gives this thread's half, its partner lane and the
// index of its addition inside the wave.
template <class F>
struct PairGeom {
  bool odd, valid;          // second half of the pair; the wave has a pair for this thread at all
  int partner4;             // partner lane * 4 (ds_bpermute address)
  uint32_t pair_in_wave;    // 0 .. PAIRS_PER_WAVE - 1
  static constexpr uint32_t PAIRS_PER_WAVE = F::LANES == 3 ? 10u : 32u / (uint32_t)F::LANES;
};
template <class F>
__device__ __forceinline__ PairGeom<F> pair_geometry() {
  PairGeom<F> g;
  const uint32_t lane = threadIdx.x & 63u;
  if constexpr (F::LANES == 3) {
    const uint32_t triple = lane / 3u;                 // 0 .. 20, 21 for lane 63
    g.odd = (triple & 1u) != 0;
    g.valid = triple < 20u;
    g.pair_in_wave = g.valid ? triple >> 1 : 9u;
    g.partner4 = (int)((g.valid ? (g.odd ? lane - 3u : lane + 3u) : lane) << 2);
  } else {
    constexpr uint32_t LN = F::LANES;
    g.odd = ((lane / LN) & 1u) != 0;
    g.valid = true;
    g.pair_in_wave = lane / (2u * LN);
    g.partner4 = (int)((lane ^ LN) << 2);
  }
  return g;
}
// S + T with TWO point-lanes per addition.  The 12 products + 2 squarings of a projective addition are five dependency levels
// deep; two halves run them as 3 + 1 + 2 + 1 + 1 = 8 sequential products: the odd half holds the operands swapped (S = its first
// operand, T = its second), so the first level is the same code in both halves, later levels pick their operands by parity and
// exchange one element each (ds_bpermute with the partner lane).  Same group element as pt_vm<PC_ADD> (mnt4753_g1.cpp:134-207:
// add-1998-cmo-2); identities are resolved by the caller's flags, equal points (a doubling) fall back to the VM in the even half.
// Every lane of the wave must call this (the exchanges need the partner); the sum is valid in the EVEN half only.
template <class C>
__device__ __forceinline__ void pt_add_pairlanes(Proj<C>& out, const Proj<C>& S, const Proj<C>& T, bool odd, int partner4) {
  using F = typename C::F;
  using E = typename F::E;
  constexpr int M = F::MOD;
  const bool zS = pt_is_zero(S), zT = pt_is_zero(T);
  // level 1: X1Z2, Y1Z2, Z1Z2 (even) | X2Z1, Y2Z1, Z1Z2 (odd)
  E t1, t2, t3, o, x1z2, x2z1, y1z2, y2z1, u, v;
  F::mul(t1, S.X, T.Z);
  F::mul(t2, S.Y, T.Z);
  F::mul(t3, S.Z, T.Z);
  fp_pair_xchg<M>(o, t1, partner4); fp_pick<M>(x1z2, odd, o, t1); fp_pick<M>(x2z1, odd, t1, o);
  fp_pair_xchg<M>(o, t2, partner4); fp_pick<M>(y1z2, odd, o, t2); fp_pick<M>(y2z1, odd, t2, o);
  F::sub(v, x2z1, x1z2);
  F::sub(u, y2z1, y1z2);
  const bool same = F::is_zero(u) && F::is_zero(v);
  // level 2: vv (even) | uu (odd)
  E sq, vv, uu, vvv, m, R, uuZ, Aq, RA;
  fp_pick<M>(m, odd, u, v);
  if constexpr (has_sqr<F>::value) F::sqr(sq, m); else F::mul(sq, m, m);
  fp_pair_xchg<M>(o, sq, partner4); fp_pick<M>(vv, odd, o, sq); fp_pick<M>(uu, odd, sq, o);
  // level 3: vvv (both), R = vv X1Z2 (even) | uu Z1Z2 (odd)
  F::mul(vvv, v, vv);
  {
    E a, b;
    fp_pick<M>(a, odd, uu, vv); fp_pick<M>(b, odd, t3, x1z2);
    F::mul(m, a, b);
  }
  fp_pair_xchg<M>(o, m, partner4); fp_pick<M>(R, odd, o, m); fp_pick<M>(uuZ, odd, m, o);
  F::sub(Aq, uuZ, vvv); F::sub(Aq, Aq, R); F::sub(Aq, Aq, R);      // A = uu Z1Z2 - vvv - 2R
  F::sub(RA, R, Aq);
  // level 4: X3 = v A (even) | vvv Y1Z2 (odd)
  E m3, m4, o3, o4;
  {
    E a, b;
    fp_pick<M>(a, odd, vvv, v); fp_pick<M>(b, odd, y1z2, Aq);
    F::mul(m3, a, b);
  }
  fp_pair_xchg<M>(o3, m3, partner4);
  // level 5: u (R - A) (even) | Z3 = vvv Z1Z2 (odd)
  {
    E a, b;
    fp_pick<M>(a, odd, vvv, u); fp_pick<M>(b, odd, t3, RA);
    F::mul(m4, a, b);
  }
  fp_pair_xchg<M>(o4, m4, partner4);
  if (zS || zT) {                       // identities: S + 0 = S, 0 + T = T
    out = zT ? S : T;
  } else if (same) {                    // equal points: the VM's addition turns into its doubling
    out = S;
    // (a real call only where inlining the VM made the kernel spill: for base fields and Fq2 the inlined form measured 0.14 ms faster
    // over the 14 narrow steps of a 2^20 G1 MSM)
    if constexpr (F::LANES == 3) pt_vm_add_outlined<C>(out, T, odd ? PC_END : PC_ADD);
    else pt_vm<C, true>(out, T, odd ? PC_END : PC_ADD);
  } else {
    out.X = m3;
    F::sub(out.Y, m4, o3);
    out.Z = o4;
  }
}
// The halving step with two point-lanes per addition, for the narrow steps (one addition deep whatever their width: 15 of the 19
// steps of a 2^19-bucket reduction, 78 us each with one lane per addition).  Base fields (2 lanes per addition), the two-lane Fq2
// (4) and, since round 3, the three-lane Fq3 (6: its steps took 177 us through the VM).
template <class C>
__global__ void __launch_bounds__(256, 1) k_reduce_step_pair(const uint32_t* __restrict__ buckets, const uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ A, uint32_t* __restrict__ G, uint32_t n_sets, uint32_t k, uint32_t s) {
  using F = typename C::F;
  static_assert((F::LANES == 1 && F::DEG == 1) || F::LANES == 2 || F::LANES == 3, "base fields and the lane-split extension fields");
  const PairGeom<F> g = pair_geometry<F>();
  const bool odd = g.odd;
  const uint32_t per_set = red_items(k, s);
  // both halves of a pair always run together (the exchanges need the partner): pairs beyond the list repeat the last item
  const uint32_t n_items = n_sets * per_set;
  const uint32_t item = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PairGeom<F>::PAIRS_PER_WAVE + g.pair_in_wave;
  const bool live = g.valid && item < n_items;
  const uint32_t t = item < n_items ? item : n_items - 1u;
  constexpr int PW = proj_words<C>();
  const uint32_t set = t / per_set, r0 = t - set * per_set;
  const uint32_t nb = 1u << k, nh = 1u << (k - s - 1);
  const uint32_t* a_src = s == 0 ? buckets + (size_t)set * nb * PW : A + ((size_t)set * nb + red_off_a(k, s)) * PW;
  uint32_t i0, i1;
  const uint32_t* src;
  uint32_t* dst;
  if (r0 < nh) {
    i0 = 2u * r0; i1 = i0 + 1u; src = a_src;
    dst = A + ((size_t)set * nb + red_off_a(k, s + 1) + r0) * PW;
  } else {
    const uint32_t sh = k - s - 2, r = r0 - nh, l = r >> sh, j = r & ((1u << sh) - 1u);
    uint32_t* gt = G + ((size_t)set * nb + red_off_g(k, l)) * PW;
    const uint32_t half = 1u << (k - l - 2);
    if (l == s) {
      i0 = 4u * j + 1u; i1 = i0 + 2u; src = a_src;
      dst = gt + (size_t)j * PW;
    } else {
      i0 = 2u * j; i1 = i0 + 1u; src = gt + (size_t)(((s - l - 1u) & 1u) * half) * PW;
      dst = gt + (size_t)(((s - l) & 1u) * half + j) * PW;
    }
  }
  // S = this lane's first operand, T = its second: the odd half holds them swapped
  const uint32_t iS = odd ? i1 : i0, iT = odd ? i0 : i1;
  const bool from_buckets = s == 0 && src == a_src;
  const bool eS = from_buckets && offsets[(size_t)set * nb + iS + 1] == offsets[(size_t)set * nb + iS];
  const bool eT = from_buckets && offsets[(size_t)set * nb + iT + 1] == offsets[(size_t)set * nb + iT];
  Proj<C> S, T, out;
  if (eS) pt_set_zero(S); else proj_load<C>(S, src + (size_t)iS * PW);
  if (eT) pt_set_zero(T); else proj_load<C>(T, src + (size_t)iT * PW);
  pt_add_pairlanes<C>(out, S, T, odd, g.partner4);
  if (odd || !live) return;
  proj_store<C>(dst, out);
}
// out = P + Q, one lane per addition without the VM: operator+ of the reference (mnt4753_g1.cpp:134-207: add-1998-cmo-2, 12 products
// + 2 squarings) in a straight line; identities pass through, equal points fall back to the VM (whose addition turns into its doubling).
// A macro for the same reason as MNT753_MADD_LINE (k_reduce_step_line<Mnt4G1>: 4 spilled registers in place, 88 behind a force-inlined
// function); declares `out`.  pt_add_line() wraps the same text for the test hook.
#define MNT753_ADD_LINE(C_, F_, out_, P_, Q_, RELOAD_)                                                                        \
  const bool zP = pt_is_zero(P_), zQ = pt_is_zero(Q_);                                                                 \
  typename F_::E x1z2, y1z2, z1z2, u, v, uu, vv, vvv, R, Aq, w;                                                        \
  F_::mul(x1z2, P_.X, Q_.Z);                                                                                           \
  F_::mul(y1z2, P_.Y, Q_.Z);                                                                                           \
  F_::mul(z1z2, P_.Z, Q_.Z);                                                                                           \
  F_::mul(w, Q_.X, P_.Z); F_::sub(v, w, x1z2);                                                                         \
  F_::mul(w, Q_.Y, P_.Z); F_::sub(u, w, y1z2);                                                                         \
  const bool same = F_::is_zero(u) && F_::is_zero(v);                                                                  \
  Proj<C_> out_;                                                                                                       \
  if constexpr (has_sqr<F_>::value) { F_::sqr(uu, u); F_::sqr(vv, v); } else { F_::mul(uu, u, u); F_::mul(vv, v, v); } \
  F_::mul(vvv, v, vv);                                                                                                 \
  F_::mul(R, vv, x1z2);                                                                                                \
  F_::mul(w, uu, z1z2);                                                                                                \
  F_::sub(Aq, w, vvv); F_::sub(Aq, Aq, R); F_::sub(Aq, Aq, R);                                                         \
  F_::mul(out_.X, v, Aq);                                                                                              \
  F_::sub(w, R, Aq);                                                                                                   \
  F_::mul(w, u, w);                                                                                                    \
  F_::mul(R, vvv, y1z2);                                                                                               \
  F_::sub(out_.Y, w, R);                                                                                               \
  F_::mul(out_.Z, vvv, z1z2);                                                                                          \
  if (zP || zQ || same) { RELOAD_; if (zP || zQ) out_ = zQ ? P_ : Q_; else { out_ = P_; pt_vm<C_, true>(out_, Q_, PC_ADD); } }
// RELOAD_: statements that bring P_ and Q_ back (the kernel reads them from memory again, so that the two operands need not stay in
// registers across the fourteen products for the sake of the rare identity / equal-points cases: with them alive the kernel spilled)
template <class C>
__device__ __forceinline__ void pt_add_line(Proj<C>& res, const Proj<C>& P, const Proj<C>& Q) {
  using F = typename C::F;
  MNT753_ADD_LINE(C, F, out, P, Q, (void)0)
  res = out;
}
// One lane per addition without the VM: the same formulas in a straight line (14 products), for the wide steps of a base field
// (a round of the VM's addition takes ~90 us, its switch machine and operand routing included).
template <class C>
__global__ void __launch_bounds__(256, 1) k_reduce_step_line(const uint32_t* __restrict__ buckets, const uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ A, uint32_t* __restrict__ G, uint32_t n_sets, uint32_t k, uint32_t s) {
  using F = typename C::F;
  using E = typename F::E;
  static_assert(F::LANES == 1 && F::DEG == 1, "base fields (the two-lane Fq2 spills in this form and measured slower than the VM)");
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_set = red_items(k, s);
  if (t >= n_sets * per_set) return;
  constexpr int PW = proj_words<C>();
  const uint32_t set = t / per_set, r0 = t - set * per_set;
  const uint32_t nb = 1u << k, nh = 1u << (k - s - 1);
  const uint32_t* a_src = s == 0 ? buckets + (size_t)set * nb * PW : A + ((size_t)set * nb + red_off_a(k, s)) * PW;
  uint32_t i0, i1;
  const uint32_t* src;
  uint32_t* dst;
  if (r0 < nh) {
    i0 = 2u * r0; i1 = i0 + 1u; src = a_src;
    dst = A + ((size_t)set * nb + red_off_a(k, s + 1) + r0) * PW;
  } else {
    const uint32_t sh = k - s - 2, r = r0 - nh, l = r >> sh, j = r & ((1u << sh) - 1u);
    uint32_t* g = G + ((size_t)set * nb + red_off_g(k, l)) * PW;
    const uint32_t half = 1u << (k - l - 2);
    if (l == s) {
      i0 = 4u * j + 1u; i1 = i0 + 2u; src = a_src;
      dst = g + (size_t)j * PW;
    } else {
      i0 = 2u * j; i1 = i0 + 1u; src = g + (size_t)(((s - l - 1u) & 1u) * half) * PW;
      dst = g + (size_t)(((s - l) & 1u) * half + j) * PW;
    }
  }
  const bool from_buckets = s == 0 && src == a_src;
  const bool e0 = from_buckets && offsets[(size_t)set * nb + i0 + 1] == offsets[(size_t)set * nb + i0];
  const bool e1 = from_buckets && offsets[(size_t)set * nb + i1 + 1] == offsets[(size_t)set * nb + i1];
  Proj<C> P, Q;
  if (e0) pt_set_zero(P); else proj_load<C>(P, src + (size_t)i0 * PW);
  if (e1) pt_set_zero(Q); else proj_load<C>(Q, src + (size_t)i1 * PW);
  MNT753_ADD_LINE(C, F, out, P, Q, (e0 ? pt_set_zero(P) : proj_load<C>(P, src + (size_t)i0 * PW), e1 ? pt_set_zero(Q) : proj_load<C>(Q, src + (size_t)i1 * PW)))
  proj_store<C>(dst, out);
}
// the k + 1 points the host combines, per bucket set: out[set][0] = T = A_k[0], out[set][1 + l] = G_l
template <class C>
__global__ void __launch_bounds__(64) k_reduce_collect(const uint32_t* __restrict__ buckets, const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ A, const uint32_t* __restrict__ G, uint32_t* __restrict__ out,
                                                      uint32_t n_sets, uint32_t k) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sets * (k + 1u)) return;
  constexpr int PW = proj_words<C>();
  const uint32_t set = t / (k + 1u), which = t - set * (k + 1u), nb = 1u << k;
  const uint32_t* src;
  bool empty = false;
  if (which == 0) src = A + ((size_t)set * nb + red_off_a(k, k)) * PW;
  else {
    const uint32_t l = which - 1u;
    if (l == k - 1u) {                              // G_{k-1} = A_{k-1}[1]; for k = 1 that is bucket 1 itself
      if (k == 1u) { src = buckets + ((size_t)set * nb + 1u) * PW; empty = offsets[(size_t)set * nb + 2] == offsets[(size_t)set * nb + 1]; }
      else src = A + ((size_t)set * nb + red_off_a(k, k - 1u) + 1u) * PW;
    } else {
      const uint32_t half = 1u << (k - l - 2);
      src = G + ((size_t)set * nb + red_off_g(k, l) + ((k - 2u - l) & 1u) * half) * PW;
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)t * PW);
  if (empty) {
    // identity (0 : 1 : 0) in device form: written through the point type of the one-lane configuration
    Proj<C> z;
    pt_set_zero(z);
    if constexpr (C::F::LANES == 1) proj_store<C>(out + (size_t)t * PW, z);
    return;
  }
  const uint4* q = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int i = 0; i < PW / 4; ++i) dst[i] = q[i];
}

// ---- device form -> wire form (projective; identity is written as (0, 1, 0)) ----------------------------
template <class C>
__global__ void __launch_bounds__(64) k_points_to_wire(const uint32_t* __restrict__ in, uint32_t* __restrict__ out_wire, uint32_t n) {
  using F = typename C::F;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Proj<C> P;
  proj_load<C>(P, in + (size_t)t * proj_words<C>());
  if (pt_is_zero(P)) pt_set_zero(P);
  uint32_t* dst = out_wire + (size_t)t * 3 * wire_coord_words<C>();
#pragma unroll 1
  for (int k = 0; k < 3 * F::DEG; ++k) {
    const typename F::E& coord = (k / F::DEG == 0) ? P.X : ((k / F::DEG == 1) ? P.Y : P.Z);
    uint32_t w[24];
    fp_to_wire(w, F::comp(coord, k % F::DEG));
    store_wire24(dst + 24 * k, w);
  }
}


// ---- precomputed window multiples ------------------------------------------------------------------
// table[w][i] = 2^(c*w) * P_i in affine device form, w < W.  With it every window's digit of a scalar indexes the
// SAME bucket set (sum_w d_w * (2^(cw) P) = s * P), so one MSM needs W*N bucket additions into 2^(c-1) buckets,
// ONE bucket reduction instead of W, and no Horner pass.  Built once per base set, at parameter-load time
// (the reference's timing window opens after the parameters are loaded, libsnark/main.cpp:201-203).
template <int M>
__device__ void e_inv(Fp<M>& r, const Fp<M>& a, FieldFp<M>*) { fp_inv(r, a); }
template <int M, unsigned NR>
__device__ void e_inv(Fp2E<M>& r, const Fp2E<M>& x, FieldFp2<M, NR>*) {   // fp2.tcc:129-142
  Fp<M> t0, t1, t2, t3;
  fp_mul(t0, x.c0, x.c0);
  fp_mul(t1, x.c1, x.c1);
  fp_mul_small(t1, t1, NR);
  fp_sub(t2, t0, t1);
  fp_inv(t3, t2);
  fp_mul(r.c0, x.c0, t3);
  fp_mul(t0, x.c1, t3);
  fp_neg(r.c1, t0);
}
template <int M, unsigned NR>
__device__ void e_inv(Fp3E<M>& r, const Fp3E<M>& x, FieldFp3<M, NR>*) {   // fp3.tcc:126-143
  Fp<M> t0, t1, t2, t3, t4, t5, c0, c1, c2, u, v, t6;
  fp_mul(t0, x.c0, x.c0); fp_mul(t1, x.c1, x.c1); fp_mul(t2, x.c2, x.c2);
  fp_mul(t3, x.c0, x.c1); fp_mul(t4, x.c0, x.c2); fp_mul(t5, x.c1, x.c2);
  fp_mul_small(u, t5, NR); fp_sub(c0, t0, u);
  fp_mul_small(u, t2, NR); fp_sub(c1, u, t3);
  fp_sub(c2, t1, t4);
  fp_mul(u, x.c2, c1); fp_mul(v, x.c1, c2); fp_add(u, u, v); fp_mul_small(u, u, NR);
  fp_mul(v, x.c0, c0); fp_add(u, u, v);
  fp_inv(t6, u);
  fp_mul(r.c0, t6, c0); fp_mul(r.c1, t6, c1); fp_mul(r.c2, t6, c2);
}

// (the doubling chain of the table runs in modified Jacobian coordinates: jac_dbl, curve753.hip.h)
// One logical lane (1, 2 or 3 threads, msm_kernels' lane-split convention) per point of a tile [i0, i0 + count): table rows for
// w >= 1; row 0 is the base itself (d_aff).   ztmp: [W][count] E  (Z_w),  ptmp: [W][count] E (prefix products of the Z_w)
template <class C>
__global__ void __launch_bounds__(256, 1) k_precompute_windows(uint32_t* __restrict__ table, const uint8_t* __restrict__ inf,
                                                              uint32_t* __restrict__ ztmp, uint32_t* __restrict__ ptmp, size_t n_total,
                                                              size_t i0, size_t count, int c, int W) {
  using F = typename C::F;
  using E = typename F::E;
  constexpr int EW = F::DEG * FPS_WORDS;
  const uint32_t tl = logical_lane<F>();
  if (tl == 0xffffffffu || (size_t)tl >= count) return;   // (all threads of a logical lane leave together)
  const size_t t = tl;
  const size_t i = i0 + t;
  if (inf[i]) return;   // identity bases never enter a bucket (their digits are forced to 0)
  Jac<F> R;
  e_load<F>(R.X, table + i * aff_words<C>());
  e_load<F>(R.Y, table + i * aff_words<C>() + EW);
  F::one(R.Z);
  C::coeff_a(R.W);
  // pass 1: R_w = 2^c R_{w-1}; keep X, Y in the table row and Z in ztmp
#pragma unroll 1
  for (int w = 1; w < W; ++w) {
#pragma unroll 1
    for (int k = 0; k < c; ++k) jac_dbl<C>(R);
    uint32_t* row = table + ((size_t)w * n_total + i) * aff_words<C>();
    e_store<F>(row, R.X);
    e_store<F>(row + EW, R.Y);
    e_store<F>(ztmp + ((size_t)w * count + t) * EW, R.Z);
  }
  // pass 2: batch inversion of Z_1..Z_{W-1} (Montgomery's trick) and normalisation to affine: x = X / Z^2, y = Y / Z^3
  E pre, z, inv, zi, x, y, tmp;
  F::one(pre);
#pragma unroll 1
  for (int w = 1; w < W; ++w) {
    e_store<F>(ptmp + ((size_t)w * count + t) * EW, pre);       // product of Z_1..Z_{w-1}
    e_load<F>(z, ztmp + ((size_t)w * count + t) * EW);
    F::mul(tmp, pre, z);
    pre = tmp;
  }
  if constexpr (has_inv<F>::value) F::inv(inv, pre); else e_inv(inv, pre, (F*)nullptr);
#pragma unroll 1
  for (int w = W - 1; w >= 1; --w) {
    e_load<F>(tmp, ptmp + ((size_t)w * count + t) * EW);
    e_load<F>(z, ztmp + ((size_t)w * count + t) * EW);
    uint32_t* row = table + ((size_t)w * n_total + i) * aff_words<C>();
    e_load<F>(x, row);
    e_load<F>(y, row + EW);
    // five products through one multiplier instance: 1 / Z_w, the running inverse, 1 / Z^2, x, 1 / Z^3, y
    E zi2;
#pragma nounroll
    for (int step = 0; step < 6; ++step) {
      E a, b, r;
      switch (step) {
        case 0: a = inv; b = tmp; break;    // 1 / Z_w
        case 1: a = inv; b = z; break;      // inverse of the shorter prefix
        case 2: a = zi; b = zi; break;
        case 3: a = x; b = zi2; break;
        case 4: a = zi2; b = zi; break;
        default: a = y; b = zi; break;      // zi holds 1 / Z^3 by then
      }
      F::mul(r, a, b);
      switch (step) {
        case 0: zi = r; break;
        case 1: inv = r; break;
        case 2: zi2 = r; break;
        case 3: e_store<F>(row, r); break;
        case 4: zi = r; break;
        default: e_store<F>(row + EW, r); break;
      }
    }
  }
}

}  // namespace mnt753

#include "msm_flow.hip.h"
