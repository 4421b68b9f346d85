// Plain host-side types shared by the MSM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mnt753 {
struct MsmPlan {
  int c, W;
  uint32_t nb;        // buckets per window = 2^(c-1)
  uint32_t n_buckets; // W * nb
  uint32_t T;         // sorted entries per accumulate lane
  uint32_t n_lanes;   // accumulate lanes
  uint32_t L;         // buckets per reduce lane
  uint32_t n_chunks;  // n_sets * nb / L
  int pre;            // 1: all windows share one bucket set (precomputed window multiples)
  uint32_t n_sets;    // bucket sets: W, or 1 with the table
  int pair_levels;    // batched-affine pairing levels ahead of the accumulate (0 = none): buckets are padded to 2^pair_levels entries
  int irr_levels;     // irregular levels behind them (no padding: ceil(g / 2) slots for a bucket with g), 0 = none
};
}  // namespace mnt753

struct mnt753_bases {
  int curve, group;
  int device = 0;              // physical HIP device the set lives on (the creating thread's current device)
  size_t n;
  uint32_t* d_aff = nullptr;   // device affine, internal form: pre_W * n rows when the window table is built (row w*n + i = 2^(c w) P_i)
  int pre_c = 0, pre_W = 0;    // window bits / windows of the precomputed table (0 = no table)
  uint8_t* d_inf = nullptr;    // identity flags
  // workspace (sized for an MSM over all n bases; reused by every call)
  size_t ws_n = 0;
  mnt753::MsmPlan ws_plan{};
  int32_t* d_digits = nullptr;
  uint32_t* d_rank = nullptr;   // rank of every (window, scalar) entry inside its bucket (returned by the histogram atomics)
  uint32_t *d_hist = nullptr, *d_offsets = nullptr, *d_cursor = nullptr, *d_blocksums = nullptr, *d_total = nullptr;
  uint32_t* d_sorted = nullptr;
  uint32_t *d_keys_out = nullptr, *d_vals_out = nullptr;   // (key, value) pairs of the two-level counting sort, partition order
  uint32_t* d_part_ws = nullptr;   // partition totals / starts / cursors of the two-level counting sort (msm_sort_partition)
  uint32_t *d_buckets = nullptr, *d_edges = nullptr, *d_edge_bucket = nullptr, *d_edge_tmp = nullptr, *d_edge_flags = nullptr;
  uint32_t *d_part_a = nullptr, *d_part_b = nullptr, *d_tmp = nullptr;
  // pairing levels (batched affine additions ahead of the accumulate): grow-only buffers
  uint32_t *d_pair_ws = nullptr, *d_fix = nullptr, *d_gen = nullptr;   // prefix products, cancellation counts per bucket, the stand-in point D
  uint32_t *d_pairpts[2] = {nullptr, nullptr}, *d_sorted2 = nullptr;    // rows of the levels (ping-pong), entry list of the last level: the DEVICE's pooled buffers (PairPool, msm_host.hpp), bound per MSM
  uint32_t *d_irr_offs[2] = {nullptr, nullptr}, *d_irr_src = nullptr, *d_irr_blocks = nullptr;   // irregular levels: bucket offsets (ping-pong), source words, block sums
  size_t pair_cap = 0;   // level-1 slots the pairing buffers hold
  size_t pair_buckets = 0;   // buckets d_fix holds
  size_t sorted_cap = 0;     // entries d_sorted holds (padded layout: W*n + n_buckets*(2^levels - 1))
  int no_pair = 0;       // the pairing workspace could not be allocated: this set runs the plain accumulate
  int registered = 0;    // counted in its device's PairPool (mnt753_bases_create got as far as registering the set)
  uint32_t* d_wire_out = nullptr;
  uint64_t* h_wire_out = nullptr;   // pinned
  uint64_t* d_scalars_stage = nullptr;
  int pending = 0, pending_sets = 0, pending_c = 0;   // an MSM enqueued by msm_start, not yet collected
  size_t pending_n = 0;
  hipStream_t pending_stream = nullptr, own_stream = nullptr;
  hipEvent_t ev_dep = nullptr;
  hipEvent_t after_ev = nullptr;   // mnt753_msm_order_after: the point kernels of the next MSM wait for this event (another set's accumulate)
  const mnt753_bases* after_owner = nullptr;   // the set that owns after_ev: mnt753_bases_free of that set clears both
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace mnt753 {
// The big buffers of the batched-affine levels -- the rows of the levels (ping-pong), the entry list of the last level, one prefix
// product per slot: 9.4 GB of a 2^20-point G1 set's 10.5 GB of workspace, 28 GB for H | L | B1, 15.6 GB for the G2 set -- are used
// between the sort and the accumulate kernel only, and the level kernels of two MSMs never overlap anyway (each fills the chip).
// Round 6: ONE set of them per DEVICE, sized for the largest request and handed from MSM to MSM by an event behind the accumulate
// kernel (what mnt753_msm_order_after does for the small sets): 53 -> 28 GB per MNT4753 parameter set, 120 -> 90 GB in all, the
// resident prove unchanged (0.1545-0.1560 s against 0.1543-0.1584, profiles/r06/level_buffers_pooling_ab.txt).  Sorts, edge merges and bucket
// reductions keep their own buffers and still run under another MSM's levels.
struct PairPool {
  void* buf[4] = {nullptr, nullptr, nullptr, nullptr};   // rows of levels 1, 3, 5 | rows of levels 2, 4, 6 | entry list of the last level | prefix products
  size_t cap[4] = {0, 0, 0, 0};
  hipEvent_t last_acc = nullptr;   // behind the accumulate kernel of the MSM that used the buffers last
  bool used = false;
  int refs = 0;                    // base sets alive on the device
};
constexpr int PAIR_POOL_DEVICES = 32;
extern PairPool g_pair_pool[PAIR_POOL_DEVICES];
inline PairPool& pair_pool_of(const mnt753_bases* b) { return g_pair_pool[(unsigned)b->device % PAIR_POOL_DEVICES]; }
}  // namespace mnt753
