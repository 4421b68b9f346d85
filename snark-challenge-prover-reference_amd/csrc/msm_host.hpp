// Host orchestration of the MSM kernels (templates; instantiated once per curve/group in msm_inst_*.hip).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mnt753_hip.h"
#include "common_host.hpp"
#include "host_field.hpp"
#include "msm_kernels.hip.h"
#include "msm_types.hpp"
#include "msm_sort.hpp"
#include "mnt753_generators.h"

using namespace mnt753;

namespace mnt753 {
extern int g_window_bits_override;
extern int g_window_table_mode;
// (PairPool, the device's pooled buffers of the batched-affine levels: msm_types.hpp)
extern int g_force_pair_levels, g_force_irr_levels;   // >= 0: mnt753_self_test puts the level kernels onto its small sets (as MNT753_MSM_PAIR / _IRR do for the tests)
extern float g_last_timing[5];
extern int g_last_plan[4];
extern int g_last_pair_levels;
extern int g_last_irr_levels;
}
namespace {


// window size: minimise  N*W (bucket adds)  +  ~3 * nb * W (reduction adds, incl. the k0*run tail)
int pick_window_bits(size_t n) {
  if (g_window_bits_override > 0) return g_window_bits_override;   // mnt753_msm_set_window_bits (tests)
  int best = 2; double best_cost = 1e300;
  for (int c = 2; c <= 20; ++c) {
    double W = (754 + c - 1) / c;
    double cost = W * ((double)n + 3.0 * (double)(1u << (c - 1)) * 14.0 / 11.0);
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// window size when all windows share one bucket set: N*W bucket adds + one reduction of 2^(c-1) buckets.  A bucket of the reduction
// is priced at eight additions of the accumulation (with the irregular levels an addition got cheaper, the reduction did not), and
// the base fields never go below c = 18: with fewer buckets than lanes a bucket's entries spread over several lanes and the edge
// merge, one general addition per level, becomes the longest phase of a small MSM.  profiles/r03/window_width_sweep.txt, G1:
//   2^20 points 25.0 / 24.3 / 25.0 / 26.2 ms at c = 18 / 19 / 20 / 21;  3 * 2^20: 66.0 / 64.1 / 62.3 / 62.0;  2^17: 5.3 ms at 18 against 6.2
//   at 17;  MNT6753 2^15: 2.6 ms at 18 against 3.7 at 16.   Fq2 G2 2^20: 66.9 / 65.6 / 67.3 / 71.6 ms.
// Round 6, behind the cheaper levels (profiles/r06/window_width_sweep.txt, MNT753_MSM_TABLE_BITS): the same widths win -- 2^20 G1
// 23.2 / 22.7 / 23.5 / 24.6 ms at 18 / 19 / 20 / 21, H | L | B1 57.7 / 57.9 / 63.9 at 20 / 21 / 22, Fq2 G2 68.5 / 67.1 / 68.8 at 18 / 19 / 20.  A
// level's time follows its padded slot pairs (0.49 ns each), and the padding of every bucket's last group of eight grows with the
// bucket count as fast as the entries shrink (kernels_by_window_width.txt).
int pick_precomp_bits(size_t n, double bucket_weight = 8.0, int min_c = 2) {
  int best = min_c; double best_cost = 1e300;
  for (int c = min_c; c <= 22; ++c) {
    double W = (754 + c - 1) / c;
    double cost = W * (double)n + bucket_weight * (double)(1u << (c - 1)) * 14.0 / 11.0;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Compute units the long-running point kernels (pairing levels, accumulate) are sized for: one 256-thread workgroup per CU, every
// workgroup alive for the whole kernel, the whole chip.  (A smaller budget, leaving CUs to the latency-bound phases of the other MSMs
// of a prove, was measured in rounds 3 and 4 at 240 / 224 / 208 CUs and gains nothing -- the prove is the energy of its kernels,
// profiles/r04/prove_point_cus_final.txt -- and left the product in round 5.)
constexpr uint32_t POINT_CUS = 256;
inline uint32_t point_cus() { return POINT_CUS; }
// logical lanes one round of those CUs holds: 256 threads per CU, 2 or 3 of them per point for the lane-split fields (21 triples
// per wave, 84 per CU)
inline uint32_t machine_lanes(int lanes_per_point) {
  return lanes_per_point == 3 ? point_cus() * 84u : point_cus() * 256u / (uint32_t)lanes_per_point;
}
// lanes_per_point: threads that share one point in the point kernels (1, or 2 / 3 with the lane-split G2 fields); the
// machine holds 65536 threads at one wave per SIMD, i.e. 65536 / lanes_per_point points at a time
MsmPlan make_plan(size_t n, int pre_c, int lanes_per_point = 1) {
  MsmPlan p;
  p.pre = pre_c > 0 && g_window_bits_override == 0;
  p.c = p.pre ? pre_c : pick_window_bits(n);
  p.W = (754 + p.c - 1) / p.c;
  p.nb = 1u << (p.c - 1);
  p.n_sets = p.pre ? 1u : (uint32_t)p.W;
  p.n_buckets = p.n_sets * p.nb;
  // lanes: aim at `rounds` full rounds of the machine (256 CUs x 256 lanes, one wave per SIMD)
  const uint64_t entries = (uint64_t)p.W * n;
  const int rounds = entries < ((uint64_t)1 << 23) ? 1 : 2;   // small sets: fewer lanes = fewer edge pieces to combine (measured)
  const uint64_t machine = machine_lanes(lanes_per_point);
  uint64_t lanes_target = machine * rounds;
  uint64_t T = (entries + lanes_target - 1) / lanes_target;
  // floor of entries per lane: a lane's walk is sequential (one mixed addition after the other, ~45 us each), so a small MSM is as
  // long as its T, while more lanes mean more edge pieces to merge (one addition per bucket that straddles a lane boundary).  Round 4
  // sweep with the tree merge (profiles/r04/small_msm_sweep.txt, MNT6753 G1): 2^12 points 1.96 / 1.62 / 1.61 ms at T = 16 / 8 / 4,
  // 2^13 points 1.99 / 1.80 / 1.81; from 2^14 points on the natural T is above the floor.  MNT753_MSM_TMIN overrides: the tests use it
  // to make buckets span hundreds of lanes (trees ten levels deep in the edge merge).
  uint64_t t_min = 8;
  if (const char* e = getenv("MNT753_MSM_TMIN")) { int v = atoi(e); if (v >= 1 && v <= 4096) t_min = (uint64_t)v; }
  if (T < t_min) T = t_min;
  p.T = (uint32_t)T;
  p.n_lanes = (uint32_t)((entries + T - 1) / T);
  if (p.n_lanes == 0) p.n_lanes = 1;
  // reduce chunk: ~one round of lanes
  uint32_t L = 1;
  while ((uint64_t)p.n_buckets / L > machine && L < p.nb) L <<= 1;
  p.L = L;
  p.n_chunks = p.n_buckets / L;
  p.pair_levels = 0;   // filled in by the caller (plan_for): depends on the group and on the workspace the base set could get
  p.irr_levels = 0;
  return p;
}

}  // namespace


namespace {

template <class C>
int proj_w() { return proj_words<C>(); }

// Lane-split point kernels (FieldFp2S / FieldFp3S instantiations) for the groups that have a split configuration (G2 of both
// curves).  (The one-lane-per-point G2 kernels -- Karatsuba through one multiplier, 1.7-3.8 KB of scratch per lane, 1.8x slower -- left
// the product in round 5; the test library still runs the one-lane forms of the group law against the reference's vectors.)
template <class C>
constexpr bool use_split_acc() { return !std::is_void<typename SplitOf<C>::type>::value; }
// threads per point in the point kernels of group C under the current settings
template <class C>
int point_lanes() {
  using CS = typename SplitOf<C>::type;
  if constexpr (std::is_void<CS>::value) return 1;
  else return CS::F::LANES;
}
void free_pair_ws(mnt753_bases* b) {
  // (d_pairpts, d_sorted2, d_pair_ws belong to the device's PairPool: only forgotten here)
  void* ptrs[] = {b->d_fix, b->d_gen, b->d_irr_offs[0], b->d_irr_offs[1], b->d_irr_src, b->d_irr_blocks};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  b->d_pair_ws = b->d_fix = b->d_gen = b->d_sorted2 = nullptr;
  b->d_irr_offs[0] = b->d_irr_offs[1] = b->d_irr_src = b->d_irr_blocks = nullptr;
  b->d_pairpts[0] = b->d_pairpts[1] = nullptr;
  b->pair_cap = 0;
  b->pair_buckets = 0;
}

void free_ws(mnt753_bases* b) {
  void* ptrs[] = {b->d_rank, b->d_digits, b->d_hist, b->d_offsets, b->d_cursor, b->d_blocksums, b->d_total, b->d_sorted, b->d_buckets,
                  b->d_edges, b->d_edge_bucket, b->d_edge_tmp, b->d_edge_flags, b->d_part_a, b->d_part_b, b->d_tmp, b->d_wire_out, b->d_scalars_stage,
                  b->d_keys_out, b->d_vals_out, b->d_part_ws};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  free_pair_ws(b);
  if (b->h_wire_out) (void)hipHostFree(b->h_wire_out);
  b->d_rank = nullptr; b->d_digits = nullptr; b->d_hist = b->d_offsets = b->d_cursor = b->d_blocksums = b->d_total = nullptr;
  b->d_edge_tmp = b->d_edge_flags = nullptr;
  b->d_sorted = b->d_buckets = b->d_edges = b->d_edge_bucket = b->d_part_a = b->d_part_b = b->d_tmp = b->d_wire_out = nullptr;
  b->h_wire_out = nullptr; b->d_scalars_stage = nullptr;
  b->d_keys_out = b->d_vals_out = nullptr; b->d_part_ws = nullptr;
  b->ws_n = 0;
  b->sorted_cap = 0;
}

// Sort stage from 2^20 entries on: the hand-written two-level counting sort of msm_sort.hip instead of the histogram-atomic counting
// sort (below that its handful of launches cost more than the atomics it saves: profiles/r06/sort_stage_by_width.txt -- 2^14 points
// 0.088 ms atomic / 0.111 partitioned, 2^15 0.134 / 0.106, 98 302 points 0.317 / 0.132, 2^16 0.231 / 0.112; the threshold was 2^22 until
// the partition passes took the window width as a template parameter).  MNT753_MSM_SORT=atomic / part overrides (the tests run both
// stages on small and large inputs), =generic also keeps the partition passes on the kernels that read the width at run time.
// (rocPRIM's radix sort, the stage of round 2 -- 1.33 ms against 0.92 -- left the product in round 5.)
enum SortMode { SORT_ATOMIC = 0, SORT_PART = 2 };
inline SortMode sort_mode(uint64_t entries) {
  if (const char* e = getenv("MNT753_MSM_SORT")) return !strcmp(e, "atomic") ? SORT_ATOMIC : SORT_PART;
  return entries >= ((uint64_t)1 << 20) ? SORT_PART : SORT_ATOMIC;
}
template <class C>
int ensure_ws(mnt753_bases* b, size_t n, const MsmPlan& p) {
  // entries of the sorted list: every bucket padded to a multiple of 2^pair_levels
  const size_t sorted_need = (size_t)p.W * n + (size_t)p.n_buckets * (((size_t)1 << p.pair_levels) - 1);
  if (b->ws_n >= n && b->ws_plan.c == p.c && b->ws_plan.pre == p.pre && b->ws_plan.T == p.T && b->ws_plan.L == p.L && b->ws_plan.n_lanes >= p.n_lanes &&
      b->sorted_cap >= sorted_need) return 0;
  free_ws(b);
  const size_t PW = proj_words<C>();
  const size_t nscan_blocks = ((size_t)p.n_buckets + SCAN_BLOCK - 1) / SCAN_BLOCK;
  HIP_TRY(hipMalloc(&b->d_digits, sizeof(int32_t) * (size_t)p.W * n));
  HIP_TRY(hipMalloc(&b->d_rank, sizeof(uint32_t) * (size_t)p.W * n));
  HIP_TRY(hipMalloc(&b->d_hist, sizeof(uint32_t) * (size_t)p.n_buckets));
  HIP_TRY(hipMalloc(&b->d_offsets, sizeof(uint32_t) * ((size_t)p.n_buckets + 1)));
  HIP_TRY(hipMalloc(&b->d_cursor, sizeof(uint32_t) * (size_t)p.n_buckets));
  HIP_TRY(hipMalloc(&b->d_blocksums, sizeof(uint32_t) * (nscan_blocks + 1)));
  HIP_TRY(hipMalloc(&b->d_total, sizeof(uint32_t) * 4));
  HIP_TRY(hipMalloc(&b->d_sorted, sizeof(uint32_t) * sorted_need));
  b->sorted_cap = sorted_need;
  if (sort_mode((uint64_t)p.W * n) == SORT_PART) {
    // (key, value) buffers of the two-level counting sort: failing to get them only falls back to the histogram-atomic sort
    if (hipMalloc(&b->d_keys_out, sizeof(uint32_t) * (size_t)p.W * n) != hipSuccess || hipMalloc(&b->d_vals_out, sizeof(uint32_t) * (size_t)p.W * n) != hipSuccess ||
        hipMalloc(&b->d_part_ws, sizeof(uint32_t) * msm_sort_partition_ws_words()) != hipSuccess) {
      (void)hipGetLastError();
      for (void* q : {(void*)b->d_keys_out, (void*)b->d_vals_out, (void*)b->d_part_ws}) if (q) (void)hipFree(q);
      b->d_keys_out = b->d_vals_out = nullptr; b->d_part_ws = nullptr;
    }
  }
  HIP_TRY(hipMalloc(&b->d_buckets, sizeof(uint32_t) * PW * (size_t)p.n_buckets));
  HIP_TRY(hipMalloc(&b->d_edges, sizeof(uint32_t) * PW * 2 * (size_t)p.n_lanes));
  HIP_TRY(hipMalloc(&b->d_edge_bucket, sizeof(uint32_t) * 2 * (size_t)p.n_lanes));
  HIP_TRY(hipMalloc(&b->d_edge_tmp, sizeof(uint32_t) * PW * 2 * (size_t)p.n_lanes));
  HIP_TRY(hipMalloc(&b->d_edge_flags, sizeof(uint32_t) * 128));   // 40 level flags, the node counts of the list-driven levels behind them
  // bucket reduction by halving (k_reduce_step): A_1 .. A_k and the ping-pong halves of the trees, nb points each per bucket
  // set; then the c points per set (T, G_0 .. G_{c-2}) the host combines
  HIP_TRY(hipMalloc(&b->d_part_a, sizeof(uint32_t) * PW * (size_t)p.n_buckets));
  HIP_TRY(hipMalloc(&b->d_part_b, sizeof(uint32_t) * PW * (size_t)p.n_buckets));
  HIP_TRY(hipMalloc(&b->d_tmp, sizeof(uint32_t) * PW * (size_t)p.n_sets * (size_t)p.c));
  HIP_TRY(hipMalloc(&b->d_wire_out, sizeof(uint32_t) * 3 * wire_coord_words<C>() * (size_t)p.n_sets * (size_t)p.c));
  HIP_TRY(hipHostMalloc(&b->h_wire_out, sizeof(uint32_t) * 3 * wire_coord_words<C>() * (size_t)p.n_sets * (size_t)p.c));
  HIP_TRY(hipMalloc(&b->d_scalars_stage, 96 * n));
  b->ws_n = n;
  b->ws_plan = p;
  return 0;
}

template <class C> MsmPlan plan_for(const mnt753_bases* b, size_t n);
template <class C> int ensure_pair_ws_for(mnt753_bases* b, const MsmPlan& p, size_t n);

template <class C>
int bases_create_t(mnt753_bases* b, const uint64_t* affine, int on_device, size_t n) {
  const size_t wire_bytes = n * 2 * wire_coord_words<C>() * 4;
  // window table: on by default for base sets large enough to amortise it over the proofs of a resident prover; a one-proof process
  // turns it off for the sets it creates (mnt753_msm_set_window_table(0): a table costs 0.33 s per 2^20 G1 points to build and saves
  // 20 ms per MSM); MNT753_MSM_PRECOMP=0 / 1 overrides both
  bool want_table = (n >= 4096 && g_window_table_mode != 0) || (n > 0 && g_window_table_mode == 2);   // 2: mnt753_self_test, level 2
  if (const char* e = getenv("MNT753_MSM_PRECOMP")) want_table = atoi(e) != 0 && n > 0;
  int pc = 0, pW = 1;
  if (want_table) {
    // floors (and, for Fq3, the one width its sizes ever want) from the sweep: the bucket count at which one round of lanes holds
    // the whole reduction -- 2^17 buckets for the base fields and the two-lane Fq2, 2^13 for the three-lane Fq3 (MNT6753 G2 2^15:
    // 11.9 ms at c = 14 against 12.3-12.4 at 15 / 16; two-adicity 15 caps its size)
    // (round 4, with the narrow halving steps on lane groups at ~30 us each: 4096 base-field points 1.43 ms at 18 bits, 1.26 at 14 --
    // thirteen steps instead of seventeen --; from 8192 points on 18 is still best: profiles/r04/flow_window_sweep.txt)
    if (C::F::DEG == 1) pc = pick_precomp_bits(n, 8.0, n <= 4096 ? 14 : 18);
    else if (C::F::DEG == 2) pc = pick_precomp_bits(n, 8.0, n >= ((size_t)1 << 16) ? 18 : 2);
    else pc = n <= ((size_t)1 << 15) ? 14 : pick_precomp_bits(n, 8.0, 2);
    if (const char* e = getenv("MNT753_MSM_TABLE_BITS")) { int v = atoi(e); if (v >= 8 && v <= 22) pc = v; }   // tools/experiments/window_sweep.sh
    pW = (754 + pc - 1) / pc;
    if ((uint64_t)pW * n >= 0x7fffffffull) { want_table = false; pc = 0; pW = 1; }   // row index must fit 31 bits
  }
  if (hipMalloc(&b->d_aff, sizeof(uint32_t) * aff_words<C>() * std::max<size_t>(n, 1) * (size_t)pW) != hipSuccess) {
    // no room for the window table (8.9 GB per 2^20 G1 points, 26 GB for Fq3): keep the set usable with one bucket set per window
    (void)hipGetLastError();
    if (!want_table) return set_error(MNT753_ENOMEM, "bases_create: device allocation failed");
    want_table = false; pc = 0; pW = 1;
    HIP_TRY(hipMalloc(&b->d_aff, sizeof(uint32_t) * aff_words<C>() * std::max<size_t>(n, 1)));
  }
  HIP_TRY(hipMalloc(&b->d_inf, std::max<size_t>(n, 1)));
  if (n == 0) return 0;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(affine);
  uint32_t* staged = nullptr;
  if (!on_device) {
    HIP_TRY(hipMalloc(&staged, wire_bytes));
    HIP_TRY(hipMemcpy(staged, affine, wire_bytes, hipMemcpyHostToDevice));
    src = staged;
  }
  hipLaunchKernelGGL((k_bases_to_internal<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, b->d_aff, b->d_inf, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  if (staged) HIP_TRY(hipFree(staged));
  if (want_table) {
    const size_t tile = std::min<size_t>(n, (size_t)1 << 17);
    const size_t EW = (size_t)C::F::DEG * FPS_WORDS;
    uint32_t *ztmp = nullptr, *ptmp = nullptr;
    HIP_TRY(hipMalloc(&ztmp, sizeof(uint32_t) * EW * tile * (size_t)pW));
    HIP_TRY(hipMalloc(&ptmp, sizeof(uint32_t) * EW * tile * (size_t)pW));
    // the doubling chains run on the configuration the point kernels use: two / three lanes per point for G2
    using CS = typename SplitOf<C>::type;
    for (size_t i0 = 0; i0 < n; i0 += tile) {
      const size_t cnt = std::min(tile, n - i0);
      if constexpr (!std::is_void<CS>::value)
        hipLaunchKernelGGL((k_precompute_windows<CS>), dim3(blocks_for<typename CS::F>(cnt)), dim3(256), 0, 0, b->d_aff, b->d_inf, ztmp, ptmp, n, i0, cnt, pc, pW);
      else
        hipLaunchKernelGGL((k_precompute_windows<C>), dim3(blocks_for<typename C::F>(cnt)), dim3(256), 0, 0, b->d_aff, b->d_inf, ztmp, ptmp, n, i0, cnt, pc, pW);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(ztmp));
    HIP_TRY(hipFree(ptmp));
    b->pre_c = pc;
    b->pre_W = pW;
  }
  // Workspace and events of a full-size MSM over this set are allocated here, at parameter-load time, so that the
  // first mnt753_msm* call on the set does not start with ~20 hipMallocs.
  // A set of a one-proof process (mode 0, no table) also runs WITHOUT the batched-affine levels: their buffers are 14 GB per 2^20 G1
  // points without a table (43 GB for the 3 x 2^20 points of H | L | B1), which the driver hands over at ~30 GB/s -- 1.5 s of a
  // parameter load measured in round 6, and as much again inside the first proof when the memory had just been another process's --
  // to save ~0.05 s on the one proof that will ever run over the set.
  if (g_window_table_mode == 0 && !want_table) b->no_pair = 1;
  {
    MsmPlan p = plan_for<C>(b, n);
    if (p.pair_levels > 0 && ensure_pair_ws_for<C>(b, p, n) != 0) {
      // not enough HBM for the pairing buffers (~7 GB per 2^20 G1 points): keep the set usable without the pairing levels
      free_pair_ws(b);
      b->no_pair = 1;
      (void)hipGetLastError();
      p = plan_for<C>(b, n);
    }
    if (int rc = ensure_ws<C>(b, n, p)) return rc;
    for (int i = 0; i < 5; ++i)
      if (!b->ev[i]) HIP_TRY(hipEventCreate(&b->ev[i]));
  }
  return 0;
}

// Host tail of an MSM.  Per bucket set the device hands over c points: T (the plain sum of the buckets) and G_0 .. G_{c-2}
// (k_reduce_step); the set's value sum_b (b + 1) B[b] is  T + sum_l 2^l G_l  -- Horner from the top, c - 1 doublings and
// additions.  Without the window table the W sets are the windows and a second Horner (c doublings per window) combines them;
// with it there is one set and nothing left to do.
template <class HC>
void horner_host(const uint64_t* wire_pts, int n_sets, int c, uint64_t* out) {
  using P = host::HPoint<HC>;
  const int PWW = 36 * HC::F::DEG;
  P acc = P::zero();
  for (int w = n_sets - 1; w >= 0; --w) {
    const uint64_t* pts = wire_pts + (size_t)w * c * PWW;
    P g = P::zero();
    for (int l = c - 2; l >= 0; --l) {
      if (!g.is_zero()) g = g.dbl();
      g = g.add(P::from_wire(pts + (size_t)(1 + l) * PWW));
    }
    g = g.add(P::from_wire(pts));
    if (!acc.is_zero())
      for (int k = 0; k < c; ++k) acc = acc.dbl();
    acc = acc.add(g);
  }
  acc.to_wire(out);
}

// Pairing levels (k_pair_level, MNT753_MSM_PAIR = number of levels) + accumulate over the shortened list.
// Lanes per level: one round of the machine at one wave per SIMD, every lane in use down to batches of PAIR_MIN_B additions per
// inversion.
constexpr uint32_t PAIR_MAX_LANES = 65536u, PAIR_MIN_B = 8u;
// Levels of the pairing pass for an MSM with `entries` sorted entries, from the one-GPU slice sweep of round 3
// (tools/slice_sweep.py, profiles/r03/slice_sweep.json: every size an 8-way split of the benchmark configurations produces, levels
// 0..4).  What a level costs besides its products is one inversion per lane (0.3 ms of wave time whatever the batch length), so
// the floor on the batch length that round 2 used (48 additions per inversion) was the wrong economy for slices: it left most of
// the machine idle behind a few lanes -- at 2^18 G1 points three levels take 8.8 ms with batches of 19..76 on every lane and
// 9.4 ms with the floor, against 10.2 ms without levels.  MNT753_MSM_PAIR=<levels> overrides, 0 turns the pass off.
template <class V>
int pair_levels(uint64_t entries) {
  if constexpr (V::F::DEG != 1 && V::F::LANES == 1) return 0;   // (one-lane Fq2 / Fq3: not instantiated by the product)
  else {
    if (g_force_pair_levels >= 0) return g_force_pair_levels;
    if (const char* e = getenv("MNT753_MSM_PAIR")) { int v = atoi(e); return v < 0 ? 0 : (v > 6 ? 6 : v); }
    if constexpr (V::F::LANES == 1) {          // G1: 2^18 points and up (2^17: 5.3 ms plain, 5.5 with two levels)
      if (entries >= ((uint64_t)1 << 23)) return 3;
    } else if constexpr (V::F::LANES == 2) {   // two-lane Fq2: more arithmetic per gathered byte, crossover ~2^17 points
      if (entries >= ((uint64_t)1 << 23)) return 3;
      if (entries >= ((uint64_t)1 << 22)) return 2;
    } else {                                   // three-lane Fq3: its accumulate runs the point VM, levels pay from 2^12 points on
      if (entries >= ((uint64_t)1 << 20)) return 3;   // MNT6753 2^15: 12.8 ms against 15.0 without
      if (entries >= ((uint64_t)1 << 19)) return 2;   // 2^14: 9.2 against 10.2
      if (entries >= ((uint64_t)1 << 17)) return 1;   // 2^13, 2^12: 6.5 / 6.1 against 6.9 / 6.3
    }
    return 0;
  }
}
// Irregular levels behind the regular ones (k_pair_level<.., IRR>, msm_kernels.hip.h): MNT753_MSM_IRR=<levels> overrides, 0 turns
// them off.  A level is worth its inversion (one per lane, ~0.3 ms of wave time whatever the batch) while it still has a batch per
// lane, and worth anything only while buckets hold more than a couple of slots; from the one-GPU sweep of round 3
// (profiles/r03/irregular_levels_sweep.txt): at least 16 output slots per lane for the base fields, 10 for the lane-split ones (their
// additions cost three times as much against the same inversion), at most three levels.
//   2^20 points:  G1 25.7 -> 25.2 ms with two levels, Fq2 G2 70.0 -> 66.5 ms with three;  3 * 2^20 G1 points (H | L | B1): 66.3 -> 62.8.
template <class C>
int irr_levels_for(uint64_t entries, int regular_levels, uint32_t n_buckets) {
  if (g_force_irr_levels >= 0) return g_force_irr_levels;
  if (const char* e = getenv("MNT753_MSM_IRR")) { int v = atoi(e); return v < 0 ? 0 : (v > 8 ? 8 : v); }
  const double lanes = (double)std::min<uint32_t>(machine_lanes(C::F::LANES), C::F::LANES == 3 ? PAIR_MAX_LANES / 3u : PAIR_MAX_LANES / (uint32_t)C::F::LANES);
  const double min_batch = C::F::LANES == 1 ? 16.0 : 10.0;
  double slots = (double)entries / (double)(1u << regular_levels) + 0.5 * n_buckets;
  int k = 0;
  while (k < 3) {
    if (slots <= 2.2 * n_buckets) break;                 // hardly a pair left per bucket
    const double out = 0.5 * slots + 0.25 * n_buckets;   // half the buckets keep an odd leftover
    if (out < min_batch * lanes) break;
    slots = out;
    ++k;
  }
  return k;
}
// level-1 slots of the worst case of plan p: (W n + n_buckets (2^L - 1)) / 2
inline uint64_t pair_cap1(const MsmPlan& p, size_t n) {
  return ((uint64_t)p.W * n + (uint64_t)p.n_buckets * (((uint64_t)1 << p.pair_levels) - 1)) / 2;
}
// buffers of the pairing levels for an MSM over n points with plan p; grow only
template <class V, class C>
int ensure_pair_ws(mnt753_bases* b, const MsmPlan& p, size_t n) {
  if constexpr (V::F::DEG == 1 || V::F::LANES > 1) {
    const uint64_t cap1 = pair_cap1(p, n);
    const uint64_t capA = std::max<uint64_t>(cap1, b->pair_cap);
    const size_t nbA = std::max<size_t>(p.n_buckets, b->pair_buckets);
    // rows of a level: row-major (last level, 224 B x DEG per slot) or four blocked planes (same bytes + rounding per plane)
    const size_t slack = 4 * 64 * 7 * 16 * 2;
    // the device's pooled buffers: grown to what THIS set needs (bytes: the sets of a device differ in row width), then bound
    {
      // (one set per device for EVERY base set: letting the small sets keep buffers of their own, so that A's MSM still interleaves with
      // C's, was measured -- 0.1547-0.1566 s per prove against 0.1545-0.1560 pooled and 0.1543-0.1584 unpooled, 100 / 90 / 120 GB:
      // profiles/r06/level_buffers_pooling_ab.txt -- and bought nothing)
      PairPool& pool = pair_pool_of(b);
      const size_t need[4] = {sizeof(uint32_t) * aff_words<V>() * capA + slack * V::F::DEG,                      // levels 1, 3, 5
                              // (an irregular level writes at most half its input plus one slot per bucket: the buckets' worth of room covers it at any depth)
                              sizeof(uint32_t) * aff_words<V>() * (capA / 2 + nbA + 1) + slack * V::F::DEG,       // levels 2, 4, 6
                              sizeof(uint32_t) * capA,                                                            // entry list of the last level
                              16 * blk_quads((uint64_t)capA * V::F::LANES + 64)};                                 // one prefix product per slot (blocked)
      bool grow = false;
      for (int i = 0; i < 4; ++i) grow = grow || need[i] > pool.cap[i];
      if (grow) {
        if (pool.used) HIP_TRY(hipEventSynchronize(pool.last_acc));   // nobody reads the old buffers behind its accumulate kernel
        for (int i = 0; i < 4; ++i) {
          if (need[i] <= pool.cap[i]) continue;
          if (pool.buf[i]) { (void)hipFree(pool.buf[i]); pool.buf[i] = nullptr; pool.cap[i] = 0; }
          HIP_TRY(hipMalloc(&pool.buf[i], need[i]));
          pool.cap[i] = need[i];
        }
      }
      b->d_pairpts[0] = static_cast<uint32_t*>(pool.buf[0]);
      b->d_pairpts[1] = static_cast<uint32_t*>(pool.buf[1]);
      b->d_sorted2 = static_cast<uint32_t*>(pool.buf[2]);
      b->d_pair_ws = static_cast<uint32_t*>(pool.buf[3]);
    }
    if (b->pair_cap >= cap1 && b->pair_buckets >= p.n_buckets) return 0;
    // the set's own small buffers
    free_pair_ws(b);
    b->d_pairpts[0] = static_cast<uint32_t*>(pair_pool_of(b).buf[0]);
    b->d_pairpts[1] = static_cast<uint32_t*>(pair_pool_of(b).buf[1]);
    b->d_sorted2 = static_cast<uint32_t*>(pair_pool_of(b).buf[2]);
    b->d_pair_ws = static_cast<uint32_t*>(pair_pool_of(b).buf[3]);
    HIP_TRY(hipMalloc(&b->d_fix, sizeof(uint32_t) * nbA));
    HIP_TRY(hipMalloc(&b->d_gen, sizeof(uint32_t) * aff_words<V>()));
    HIP_TRY(hipMalloc(&b->d_irr_offs[0], sizeof(uint32_t) * (nbA + 1)));
    HIP_TRY(hipMalloc(&b->d_irr_offs[1], sizeof(uint32_t) * (nbA + 1)));
    HIP_TRY(hipMalloc(&b->d_irr_src, sizeof(uint32_t) * (capA / 2 + nbA + 64)));
    HIP_TRY(hipMalloc(&b->d_irr_blocks, sizeof(uint32_t) * (nbA / IRR_BLOCK + 4)));
    // D: the group generator in device form (wire constant -> k_bases_to_internal)
    uint32_t* wire = nullptr; uint8_t* inf = nullptr;
    const size_t gen_bytes = 192 * (size_t)C::F::DEG;
    const uint64_t* gen_wire = b->curve == MNT753_CURVE_MNT4753 ? (C::F::DEG == 1 ? GEN_MNT4_G1 : GEN_MNT4_G2) : (C::F::DEG == 1 ? GEN_MNT6_G1 : GEN_MNT6_G2);
    HIP_TRY(hipMalloc(&wire, gen_bytes)); HIP_TRY(hipMalloc(&inf, 16));
    HIP_TRY(hipMemcpy(wire, gen_wire, gen_bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_bases_to_internal<C>), dim3(1), dim3(256), 0, 0, wire, b->d_gen, inf, (size_t)1);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(wire)); HIP_TRY(hipFree(inf));
    b->pair_cap = capA;
    b->pair_buckets = nbA;
    return 0;
  } else {
    (void)b; (void)p; (void)n;
    return 0;
  }
}
// workspace for the configuration the point kernels of group C currently run with
template <class C>
int ensure_pair_ws_for(mnt753_bases* b, const MsmPlan& p, size_t n) {
  using CS = typename SplitOf<C>::type;
  if constexpr (!std::is_void<CS>::value) return ensure_pair_ws<CS, C>(b, p, n);
  else return ensure_pair_ws<C, C>(b, p, n);
}
template <class C>
int pair_levels_for(uint64_t entries) {
  using CS = typename SplitOf<C>::type;
  if constexpr (!std::is_void<CS>::value) return pair_levels<CS>(entries);
  else return pair_levels<C>(entries);
}
template <class C>
int irr_levels_for_group(uint64_t entries, int regular_levels, uint32_t n_buckets) {
  using CS = typename SplitOf<C>::type;
  if constexpr (!std::is_void<CS>::value) return irr_levels_for<CS>(entries, regular_levels, n_buckets);
  else return irr_levels_for<C>(entries, regular_levels, n_buckets);
}
// plan of an MSM over n points of base set b: make_plan + the pairing levels this group / base set runs with
template <class C>
MsmPlan plan_for(const mnt753_bases* b, size_t n) {
  MsmPlan p = make_plan(n, b->pre_c, point_lanes<C>());
  p.pair_levels = b->no_pair ? 0 : pair_levels_for<C>((uint64_t)p.W * n);
  // the padded list must keep 32-bit slot indices
  while (p.pair_levels > 0 && (uint64_t)p.W * n + (uint64_t)p.n_buckets * (((uint64_t)1 << p.pair_levels) - 1) >= 0xfffffff0ull) --p.pair_levels;
  // base fields: the first level keeps table offsets in 32-bit registers (uint4 units, 14 per row) -- a table of 64 GiB or more
  // (2^23 G1 points with 38 windows) goes through the accumulate kernel alone
  if (point_lanes<C>() == 1 && C::F::DEG == 1 && (uint64_t)p.W * std::max<uint64_t>(b->n, n) * 14u >= 0xffffffffull) p.pair_levels = 0;
  // every field: blocked indices of level-1 slots (7 uint4 per element and lane) are 32-bit in the level kernels
  if (p.pair_levels > 0 && pair_cap1(p, n) * (uint64_t)point_lanes<C>() * 7u >= 0xffffff00ull) p.pair_levels = 0;
  p.irr_levels = p.pair_levels > 0 ? irr_levels_for_group<C>((uint64_t)p.W * n, p.pair_levels, p.n_buckets) : 0;
  return p;
}
template <class V, class C>
int pair_and_accumulate(const MsmPlan& p, size_t n, hipStream_t st, const uint32_t* d_aff, mnt753_bases* b, uint32_t* acc_lanes, uint32_t* acc_T, const uint32_t** acc_offs) {
  if constexpr (V::F::DEG == 1 || V::F::LANES > 1) {
    const int levels = p.pair_levels;
    // logical lanes: one workgroup per CU (the level kernels take 145 KB of LDS), 21 triples per wave for three-lane fields
    const uint32_t max_lanes = std::min<uint32_t>(machine_lanes(V::F::LANES), V::F::LANES == 3 ? PAIR_MAX_LANES / 3u : PAIR_MAX_LANES / (uint32_t)V::F::LANES);
    const uint32_t min_B = PAIR_MIN_B;
    if (int rc = ensure_pair_ws<V, C>(b, p, n)) return rc;
    // the device's pooled level buffers: ours once the accumulate kernel of the MSM that used them last has ended
    PairPool& pool = pair_pool_of(b);
    if (pool.used) HIP_TRY(hipStreamWaitEvent(st, pool.last_acc, 0));
    HIP_TRY(hipMemsetAsync(b->d_fix, 0, sizeof(uint32_t) * (size_t)p.n_buckets, st));
    uint64_t cap = 2 * pair_cap1(p, n);        // worst-case entries of the padded list
    const uint4* src_planes = nullptr;
    size_t src_stride = 0;
    const uint32_t* last_rows = nullptr;
    // irregular levels behind the regular ones, as many as the ping-pong row buffers hold in the worst case (each writes at most half
    // its input plus one slot per bucket)
    int irr = p.irr_levels;
    {
      uint64_t c = cap >> levels;
      for (int k = 1; k <= irr; ++k) {
        c = c / 2 + p.n_buckets;
        const uint64_t room = ((levels + k - 1) & 1) ? b->pair_cap / 2 + b->pair_buckets : b->pair_cap;
        if (c > room || c > b->pair_cap / 2 + b->pair_buckets + 64) { irr = k - 1; break; }   // rows of the level, source words
      }
    }
    for (int l = 1; l <= levels; ++l) {
      cap /= 2;                                 // worst-case slots of this level (the kernel reads the actual count from offsG)
      const uint32_t lanes = (uint32_t)std::min<uint64_t>(max_lanes, (cap + min_B - 1) / min_B);
      uint32_t* out = b->d_pairpts[(l - 1) & 1];
      const int first = l == 1, last = l == levels && irr == 0;
      // planes of a level that feeds another one: (x | y) x (even | odd slot), each holding cap / 2 slots, blocked
      const size_t out_stride = blk_quads((cap / 2 + 1) * (uint64_t)V::F::LANES + 64);
#define MNT753_PAIR_LAUNCH(FST, LST)                                                                                                         \
  do {                                                                                                                                      \
    /* the level kernels stage their operands in 145 KB of dynamic LDS per workgroup: opt in once per kernel and DEVICE */                  \
    static std::atomic<uint32_t> lds_set{0};                                                                                                \
    const uint32_t dev_bit = 1u << (b->device & 31);                                                                                        \
    if (!(lds_set.load(std::memory_order_acquire) & dev_bit)) {                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_level<V, FST, LST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PAIR_LDS_BYTES)); \
      lds_set.fetch_or(dev_bit, std::memory_order_release);                                                                                 \
    }                                                                                                                                       \
    hipLaunchKernelGGL((k_pair_level<V, FST, LST>), dim3(blocks_for<typename V::F>(lanes)), dim3(256), PAIR_LDS_BYTES, st, d_aff, b->d_sorted, src_planes, \
                       src_stride, b->d_offsets, p.n_buckets, (uint32_t)(levels - l), out, b->d_sorted2, reinterpret_cast<uint4*>(out), out_stride,  \
                       reinterpret_cast<uint4*>(b->d_pair_ws), min_B, lanes, b->d_gen, b->d_fix);                              \
  } while (0)
      if (first && last) MNT753_PAIR_LAUNCH(true, true);
      else if (first) MNT753_PAIR_LAUNCH(true, false);
      else if (last) MNT753_PAIR_LAUNCH(false, true);
      else MNT753_PAIR_LAUNCH(false, false);
#undef MNT753_PAIR_LAUNCH
#ifdef MNT753_PAIR_TIMING
      {
        unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pair_cycles), sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_cycles), z, sizeof(z));
        unsigned int slots = 1;
        (void)hipMemcpyFromSymbol(&slots, HIP_SYMBOL(g_pair_slots), sizeof(slots));
        const double per = (double)h[3] * (slots ? slots : 1);
        if (h[3]) fprintf(stderr, "pair level %d: waves %llu slots/wave %u  ticks/slot: forward %.0f (dma issue %.0f, store %.0f)  backward %.0f (dma issue %.0f, store %.0f)  inversion/wave %.0f\n", l, h[3], slots,
                          (double)h[0] / per, (double)h[4] / per, (double)h[5] / per, (double)h[2] / per, (double)h[6] / per, (double)h[7] / per, (double)h[1] / h[3]);
      }
#endif
      src_planes = reinterpret_cast<const uint4*>(out);
      src_stride = out_stride;
      last_rows = out;
    }
    // irregular levels: bucket offsets of their own (the regular levels' final slots are the groups offsG counts)
    const uint32_t* offs_acc = b->d_offsets;
    for (int k = 1; k <= irr; ++k) {
      const int l = levels + k;
      const uint32_t* offs_in = offs_acc;
      uint32_t* offs_out = b->d_irr_offs[k & 1];
      const uint32_t n_blocks = (p.n_buckets + IRR_BLOCK - 1) / IRR_BLOCK;
      hipLaunchKernelGGL(k_irr_count, dim3(n_blocks), dim3(IRR_BLOCK), 0, st, offs_in, p.n_buckets, b->d_irr_blocks);
      hipLaunchKernelGGL(k_irr_scan, dim3(1), dim3(1024), 0, st, b->d_irr_blocks, n_blocks);
      hipLaunchKernelGGL(k_irr_fill, dim3(n_blocks), dim3(IRR_BLOCK), 0, st, offs_in, p.n_buckets, b->d_irr_blocks, n_blocks, offs_out, b->d_irr_src);
      cap = cap / 2 + p.n_buckets;               // worst case: every bucket keeps an odd leftover
      const uint32_t lanes = (uint32_t)std::min<uint64_t>(max_lanes, (cap + min_B - 1) / min_B);
      uint32_t* out = b->d_pairpts[(l - 1) & 1];
      const size_t out_stride = blk_quads((cap / 2 + 1) * (uint64_t)V::F::LANES + 64);
#define MNT753_IRR_LAUNCH(LST)                                                                                                               \
  do {                                                                                                                                      \
    static std::atomic<uint32_t> lds_set{0};                                                                                                \
    const uint32_t dev_bit = 1u << (b->device & 31);                                                                                        \
    if (!(lds_set.load(std::memory_order_acquire) & dev_bit)) {                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_level<V, false, LST, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PAIR_LDS_BYTES)); \
      lds_set.fetch_or(dev_bit, std::memory_order_release);                                                                                 \
    }                                                                                                                                       \
    hipLaunchKernelGGL((k_pair_level<V, false, LST, true>), dim3(blocks_for<typename V::F>(lanes)), dim3(256), PAIR_LDS_BYTES, st, d_aff, b->d_sorted, src_planes, \
                       src_stride, offs_out, p.n_buckets, 0u, out, b->d_sorted2, reinterpret_cast<uint4*>(out), out_stride,                  \
                       reinterpret_cast<uint4*>(b->d_pair_ws), min_B, lanes, b->d_gen, b->d_fix, b->d_irr_src);                              \
  } while (0)
      if (k == irr) MNT753_IRR_LAUNCH(true); else MNT753_IRR_LAUNCH(false);
#undef MNT753_IRR_LAUNCH
      src_planes = reinterpret_cast<const uint4*>(out);
      src_stride = out_stride;
      last_rows = out;
      offs_acc = offs_out;
    }
    const uint32_t* src = last_rows;
    // accumulate over at most `cap` entries: one round of the machine
    const uint32_t lanes_acc = std::min<uint32_t>(p.n_lanes, machine_lanes(V::F::LANES));
    const uint32_t T2 = (uint32_t)std::max<uint64_t>((cap + lanes_acc - 1) / lanes_acc, 8);
    hipLaunchKernelGGL((k_bucket_accumulate<V, true>), dim3(blocks_for<typename V::F>(lanes_acc)), dim3(256), 0, st, src, b->d_sorted2,
                       offs_acc, p.n_buckets, b->d_buckets, b->d_edges, b->d_edge_bucket, T2, lanes_acc, src_stride);
    if (!pool.last_acc) HIP_TRY(hipEventCreateWithFlags(&pool.last_acc, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(pool.last_acc, st));
    pool.used = true;
    *acc_lanes = lanes_acc;
    *acc_T = T2;
    *acc_offs = offs_acc;
    return 0;
  } else {
    (void)p; (void)n; (void)st; (void)d_aff; (void)b; (void)acc_lanes; (void)acc_T; (void)acc_offs;
    return 0;
  }
}

// The stages that run point arithmetic.  V = the configuration the point-operation VM is instantiated with (C itself,
// or its lane-split counterpart); kernels that only move points are layout-agnostic and use C.
template <class V, class C>
int point_stages(const MsmPlan& p, size_t n, hipStream_t st, const uint32_t* d_aff, mnt753_bases* b, uint32_t** result) {
  const int n_pair_levels = p.pair_levels;
  uint32_t acc_lanes = p.n_lanes;   // lanes the accumulate kernel ran with (= edge slots / 2)
  uint32_t acc_T = p.T;             // entries per lane it was given, and the bucket offsets it walked (the last level's, behind pairing levels)
  const uint32_t* acc_offs = b->d_offsets;
  g_last_pair_levels = n_pair_levels;
  g_last_irr_levels = n_pair_levels > 0 ? p.irr_levels : 0;
  if (n_pair_levels > 0) {
    if (int rc = pair_and_accumulate<V, C>(p, n, st, d_aff, b, &acc_lanes, &acc_T, &acc_offs)) return rc;
  } else {
    hipLaunchKernelGGL((k_bucket_accumulate<V>), dim3(blocks_for<typename V::F>(p.n_lanes)), dim3(256), 0, st, d_aff, b->d_sorted, b->d_offsets,
                       p.n_buckets, b->d_buckets, b->d_edges, b->d_edge_bucket, p.T, p.n_lanes);
  }
  HIP_TRY(hipEventRecord(b->ev[2], st));
  // the edge pieces of the buckets that span several lanes: a binary tree over the lanes of every bucket, in place (round 4)
  {
    const uint32_t n_slots = 2 * acc_lanes;
    const unsigned gs = (n_slots + 255) / 256;
    const uint32_t blocked = n_pair_levels > 0 ? 1u : 0u;
    // (T, lanes) the accumulate kernel ran with: its BLOCKED form takes T2 and derives the real share from the list's actual length
    const uint32_t t_arg = n_pair_levels > 0 ? acc_T : p.T;
    const uint32_t* offs = acc_offs;
    HIP_TRY(hipMemsetAsync(b->d_edge_flags, 0, sizeof(uint32_t) * 128, st));
    uint32_t level = 0;
    // A level is one addition deep, and its nodes are a LIST (msm_flow.hip.h): k_edge_nodes writes the first one, every level appends
    // the nodes of the next as its last act, launches are sized for the most nodes the level can have.  A level that can hold at most
    // EDGE_FLOW_NODES additions (estimated as lanes / 2^l: one node per lane boundary at the first level, half as many per level after
    // it) spreads each addition over a group of lanes, the levels before it run one VM addition per lane over the list.
    // MNT753_EDGE_FLOW_NODES moves the boundary: the tests use it to run EVERY level of deep trees through either form (the inlined VM
    // addition of round 4's first tree kernel was miscompiled exactly there).
    const uint64_t EDGE_FLOW_NODES = getenv("MNT753_EDGE_FLOW_NODES") ? strtoull(getenv("MNT753_EDGE_FLOW_NODES"), nullptr, 10)
                                     : (C::F::DEG == 1 ? 32768 : (Flow<C>::K3 ? 8192 : 16384));   // (K3: two additions per wave, 2048 per round)
    uint32_t* counts = b->d_edge_flags + 40;               // nodes of level l, behind the 40 flags
    uint4* lists[2] = {reinterpret_cast<uint4*>(b->d_edge_tmp), reinterpret_cast<uint4*>(b->d_edge_tmp) + acc_lanes};
    uint32_t parity = 0;
    if (acc_lanes > 1)
      hipLaunchKernelGGL((k_edge_nodes<C>), dim3(gs), dim3(256), 0, st, b->d_edge_bucket, offs, p.n_buckets, t_arg, acc_lanes, blocked, 1u, b->d_edge_flags, 0u, lists[0], counts);
    for (uint64_t stride = 1; stride < acc_lanes; stride *= EDGE_TREE_K, ++level) {
      // most nodes the level can have: two pieces per lane, a node takes two pieces `stride` apart
      const uint64_t most = std::min<uint64_t>(acc_lanes, 2 * ((uint64_t)acc_lanes / stride) + 1);
      if (acc_lanes / stride <= EDGE_FLOW_NODES)
        hipLaunchKernelGGL((k_edge_tree_level_list<C>), dim3((unsigned)((most + Flow<C>::PER_WAVE - 1) / Flow<C>::PER_WAVE)), dim3(64), 0, st, b->d_edges, offs, p.n_buckets,
                           t_arg, acc_lanes, blocked, (uint32_t)stride, lists[parity], counts + level, lists[parity ^ 1u], counts + level + 1);
      else
        hipLaunchKernelGGL((k_edge_tree_level_vmlist<V>), dim3(blocks_for<typename V::F>(most)), dim3(256), 0, st, b->d_edges, offs, p.n_buckets, t_arg, acc_lanes, blocked,
                           (uint32_t)stride, lists[parity], counts + level, lists[parity ^ 1u], counts + level + 1);
      parity ^= 1u;
    }
    hipLaunchKernelGGL((k_edge_tree_finish<C>), dim3(gs), dim3(256), 0, st, b->d_edges, b->d_edge_bucket, offs, p.n_buckets, t_arg, acc_lanes, blocked, b->d_buckets);
    if constexpr (V::F::DEG == 1 || V::F::LANES > 1) {
      if (n_pair_levels > 0)
        hipLaunchKernelGGL((k_pair_fix<V>), dim3(blocks_for<typename V::F>(p.n_buckets)), dim3(256), 0, st, b->d_buckets, b->d_fix, b->d_gen, p.n_buckets);
    }
  }
  // bucket reduction: k = c - 1 halving steps (each one group addition deep), then the c points per bucket set for the host
  {
    const uint32_t k = (uint32_t)p.c - 1u, NS = p.n_sets;
    // the narrowest steps: one group of lanes per addition (four products deep instead of fourteen / eight), up to the number of
    // additions at which the kernels below, with more additions per wave, catch up
    constexpr uint64_t FLOW_MAX = C::F::DEG == 3 ? (Flow<C>::K3 ? 8192 : 16384) : (C::F::DEG == 2 ? 4096 : 8192);
    constexpr uint64_t PAIR_MAX = 65536;   // lanes: up to here the two-lanes-per-addition step of the base fields and the two-lane Fq2
    for (uint32_t step = 0; step < k; ++step) {
      const uint64_t items = (uint64_t)NS * red_items(k, step);
      if (items <= FLOW_MAX) {
        hipLaunchKernelGGL((k_reduce_step_flow<C>), dim3((unsigned)((items + Flow<C>::PER_WAVE - 1) / Flow<C>::PER_WAVE)), dim3(64), 0, st, b->d_buckets, b->d_offsets,
                           b->d_part_a, b->d_part_b, NS, k, step);
        continue;
      }
      if constexpr (C::F::LANES == 1 && C::F::DEG == 1) {
        // wide steps: straight-line additions instead of the VM's; middle steps: two lanes per addition (8 sequential products instead of 14)
        if (2 * items > PAIR_MAX)
          hipLaunchKernelGGL((k_reduce_step_line<C>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, b->d_buckets, b->d_offsets, b->d_part_a, b->d_part_b, NS, k, step);
        else
          hipLaunchKernelGGL((k_reduce_step_pair<C>), dim3((unsigned)((2 * items + 255) / 256)), dim3(256), 0, st, b->d_buckets, b->d_offsets, b->d_part_a, b->d_part_b, NS, k, step);
      } else {
        // lane-split fields: the VM in the wide steps; the two-lane Fq2 with four lanes per addition in the middle ones (three-lane
        // Fq3: six lanes per addition were measured 2.4x slower than the VM here in round 3 and are not used)
        bool done = false;
        if constexpr (V::F::LANES == 2) {
          if (4 * items <= PAIR_MAX) {
            hipLaunchKernelGGL((k_reduce_step_pair<V>), dim3((unsigned)((4 * items + 255) / 256)), dim3(256), 0, st, b->d_buckets, b->d_offsets, b->d_part_a, b->d_part_b, NS, k, step);
            done = true;
          }
        }
        if (!done)
          hipLaunchKernelGGL((k_reduce_step<V>), dim3(blocks_for<typename V::F>(items)), dim3(256), 0, st, b->d_buckets, b->d_offsets, b->d_part_a, b->d_part_b, NS, k, step);
      }
    }
    hipLaunchKernelGGL((k_reduce_collect<C>), dim3((NS * (k + 1u) + 63) / 64), dim3(64), 0, st, b->d_buckets, b->d_offsets, b->d_part_a, b->d_part_b, b->d_tmp, NS, k);
  }
  uint32_t* cur = b->d_tmp;
  *result = cur;
  return 0;
}

// Enqueue one MSM on stream `st` (no host synchronisation); msm_finish_t collects the result.
template <class C, class HC>
int msm_start_t(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, hipStream_t st) {
  const int PWW = 36 * HC::F::DEG;  // projective words (u64) on the wire
  if (b->pending) return set_error(MNT753_EINVAL, "msm_start: this base set already has an MSM in flight (finish it first)");
  b->pending = 1; b->pending_n = n; b->pending_stream = st;
  if (n == 0) return 0;
  MsmPlan p = plan_for<C>(b, n);
  if (p.pair_levels > 0 && ensure_pair_ws_for<C>(b, p, n) != 0) {
    // the pairing buffers do not fit (a larger MSM than the set was created for, on a full device): plain accumulate
    free_pair_ws(b);
    b->no_pair = 1;
    (void)hipGetLastError();
    p = plan_for<C>(b, n);
  }
  if (int rc = ensure_ws<C>(b, n, p)) return rc;
  g_last_plan[0] = p.c; g_last_plan[1] = p.W; g_last_plan[2] = p.pre; g_last_plan[3] = (int)p.T;
  for (int i = 0; i < 5; ++i)
    if (!b->ev[i]) HIP_TRY(hipEventCreate(&b->ev[i]));
  const uint32_t* d_scal;
  if (scalars_on_device) {
    d_scal = reinterpret_cast<const uint32_t*>(scalars);
  } else {
    HIP_TRY(hipMemcpyAsync(b->d_scalars_stage, scalars, 96 * n, hipMemcpyHostToDevice, st));
    d_scal = reinterpret_cast<const uint32_t*>(b->d_scalars_stage);
  }
  // with the window table the sorted entries carry absolute table rows (w * n_total + base_offset + i)
  const uint32_t* d_aff = p.pre ? b->d_aff : b->d_aff + base_offset * aff_words<C>();
  const uint8_t* d_inf = b->d_inf + base_offset;
  HIP_TRY(hipEventRecord(b->ev[0], st));
  if (b->d_part_ws && sort_mode((uint64_t)p.W * n) == SORT_PART && msm_sort_partition_fits(p.n_buckets, p.W)) {
    if (int rc = msm_sort_partition(C::FR, d_scal, d_inf, n, p, p.pre ? (uint32_t)b->n : 0u, p.pre ? (uint32_t)base_offset : 0u, b->d_keys_out, b->d_vals_out,
                                    b->d_part_ws, b->d_hist, b->d_offsets, b->d_cursor, b->d_blocksums, b->d_total, b->d_sorted, st))
      return rc;
  } else {
    HIP_TRY(hipMemsetAsync(b->d_hist, 0, sizeof(uint32_t) * (size_t)p.n_buckets, st));
    const unsigned gb = (unsigned)((n + 255) / 256);
    // fewer than ~1024 workgroups: the walk over the windows is split (k_scalar_digits), up to eight ways
    const unsigned wsplit = gb >= 1024 ? 1u : std::min<unsigned>(8u, std::min<unsigned>((unsigned)p.W, (1024u + gb - 1) / gb));
    hipLaunchKernelGGL((k_scalar_digits<C::FR>), dim3(gb, wsplit), dim3(256), 0, st, d_scal, d_inf, b->d_digits, b->d_rank, b->d_hist, n, p.c, p.W, p.pre ? 0u : p.nb);
    const unsigned nsb = (unsigned)(((size_t)p.n_buckets + SCAN_BLOCK - 1) / SCAN_BLOCK);
    // with pairing levels the offsets count groups of 2^levels entries and the padding of the sorted list is ENTRY_EMPTY
    const uint32_t pshift = (uint32_t)p.pair_levels;
    if (pshift) HIP_TRY(hipMemsetAsync(b->d_sorted, 0xff, sizeof(uint32_t) * ((size_t)p.W * n + (size_t)p.n_buckets * (((size_t)1 << pshift) - 1)), st));
    hipLaunchKernelGGL(k_scan_blocks, dim3(nsb), dim3(SCAN_THREADS), 0, st, b->d_hist, b->d_offsets, b->d_blocksums, (size_t)p.n_buckets, pshift);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, b->d_blocksums, (size_t)nsb, b->d_total);
    hipLaunchKernelGGL(k_scan_finish, dim3(nsb), dim3(SCAN_THREADS), 0, st, b->d_offsets, b->d_cursor, b->d_blocksums, b->d_total,
                       (size_t)p.n_buckets);
    hipLaunchKernelGGL(k_scatter, dim3(gb, wsplit), dim3(256), 0, st, b->d_digits, b->d_rank, b->d_offsets, b->d_sorted, n, p.c, p.W, p.pre ? 0u : p.nb,
                       p.pre ? (uint32_t)b->n : 0u, p.pre ? (uint32_t)base_offset : 0u, pshift);
  }
  HIP_TRY(hipEventRecord(b->ev[1], st));
  // mnt753_msm_order_after: the sort above ran whenever it could; the kernels that fill the chip start once the other set's have ended
  if (b->after_ev) { HIP_TRY(hipStreamWaitEvent(st, b->after_ev, 0)); b->after_ev = nullptr; b->after_owner = nullptr; }
  // point stages: with the lane-split configuration of the group (Fq2: two lanes per point, Fq3: three) when it has
  // one, otherwise one lane per point
  uint32_t* cur = nullptr;
  {
    int rc;
    using CS = typename SplitOf<C>::type;
    if constexpr (!std::is_void<CS>::value) rc = point_stages<CS, C>(p, n, st, d_aff, b, &cur);
    else rc = point_stages<C, C>(p, n, st, d_aff, b, &cur);
    if (rc) return rc;
  }
  const uint32_t NS = p.n_sets, NP = NS * (uint32_t)p.c;   // c points per bucket set: T, G_0 .. G_{c-2}
  hipLaunchKernelGGL((k_points_to_wire<C>), dim3((NP + 63) / 64), dim3(64), 0, st, cur, b->d_wire_out, NP);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev[3], st));
  HIP_TRY(hipMemcpyAsync(b->h_wire_out, b->d_wire_out, sizeof(uint64_t) * PWW * (size_t)NP, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(b->ev[4], st));
  b->pending_sets = (int)NS; b->pending_c = p.c;
  return 0;
}

template <class C, class HC>
int msm_finish_t(mnt753_bases* b, uint64_t* out) {
  if (!b->pending) return set_error(MNT753_EINVAL, "msm_finish: no MSM in flight on this base set");
  b->pending = 0;
  if (b->pending_n == 0) {
    host::HPoint<HC>::zero().to_wire(out);
    return 0;
  }
  HIP_TRY(hipEventSynchronize(b->ev[4]));
  auto t0 = std::chrono::steady_clock::now();
  horner_host<HC>(b->h_wire_out, b->pending_sets, b->pending_c, out);   // one point with the window table: a copy
  auto t1 = std::chrono::steady_clock::now();
  float ms;
  HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1])); g_last_timing[1] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, b->ev[1], b->ev[2])); g_last_timing[2] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, b->ev[2], b->ev[3])); g_last_timing[3] = ms;
  g_last_timing[4] = std::chrono::duration<float, std::milli>(t1 - t0).count();
  HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[3])); g_last_timing[0] = ms + g_last_timing[4];
  return 0;
}

template <class C, class HC>
int msm_t(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, uint64_t* out,
          hipStream_t st) {
  if (int rc = msm_start_t<C, HC>(b, base_offset, scalars, scalars_on_device, n, st)) { b->pending = 0; return rc; }
  return msm_finish_t<C, HC>(b, out);
}

}  // namespace

