// The MSM part of the C ABI (include/mnt753_hip.h): dispatch to the per-group instantiations.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_set>
#include "common_host.hpp"
#include "msm_api.hpp"
#include "msm_types.hpp"

namespace mnt753 {
int g_window_bits_override = 0;
int g_force_pair_levels = -1, g_force_irr_levels = -1;
PairPool g_pair_pool[PAIR_POOL_DEVICES];
int g_window_table_mode = 1;   // mnt753_msm_set_window_table: 1 = tables for base sets of 4096 points and more, 0 = none
float g_last_timing[5] = {0, 0, 0, 0, 0};
int g_last_plan[4] = {0, 0, 0, 0};
int g_last_pair_levels = 0;
int g_last_irr_levels = 0;
}
using namespace mnt753;

namespace {
// MSM streams run at the lowest priority the device offers: an MSM is hundreds of milliseconds of throughput work,
// and short kernels on the default stream (the NTTs of compute_H, launched while MSMs are in flight) should be
// scheduled ahead of its remaining workgroups.
// (The G2 sets one level above the G1 ones -- so that the longest MSM of a prove gets the wave slots first -- was measured in rounds 3
// and 4 and gains nothing: workgroups are not pre-empted, profiles/r04/prove_point_cus_final.txt.)
// live base sets: mnt753_msm_order_after lets one set wait for an event another set owns, so freeing a set must take its event out
// of every set that still refers to it
std::mutex g_sets_mu;
std::unordered_set<mnt753_bases*> g_sets;
hipError_t create_msm_stream(hipStream_t* s, int) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; greatest = 0; }
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
}
}  // namespace

extern "C" {


int mnt753_bases_create(int curve, int group, const uint64_t* affine, int on_device, size_t n, mnt753_bases** out) {
  if (!out || (n && !affine) || curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2)) return set_error(MNT753_EINVAL, "bases_create: bad argument");
  if (int rc = require_device()) return rc;
  mnt753_bases* b = new (std::nothrow) mnt753_bases();
  if (!b) return set_error(MNT753_ENOMEM, "bases_create: host allocation failed");
  b->curve = curve; b->group = group; b->n = n;
  b->device = current_physical_device();
  int rc;
  if (curve == MNT753_CURVE_MNT4753) rc = group == MNT753_G1 ? bases_create_mnt4g1(b, affine, on_device, n) : bases_create_mnt4g2(b, affine, on_device, n);
  else rc = group == MNT753_G1 ? bases_create_mnt6g1(b, affine, on_device, n) : bases_create_mnt6g2(b, affine, on_device, n);
  if (rc) { mnt753_bases_free(b); return rc; }
  // the base set's own stream for mnt753_msm_start (creating a stream costs ~8 ms: do it here, at parameter-load time)
  if (create_msm_stream(&b->own_stream, group) != hipSuccess) b->own_stream = nullptr;
  if (hipEventCreateWithFlags(&b->ev_dep, hipEventDisableTiming) != hipSuccess) b->ev_dep = nullptr;
  (void)hipGetLastError();
  { std::lock_guard<std::mutex> l(g_sets_mu); g_sets.insert(b); ++pair_pool_of(b).refs; b->registered = 1; }
  *out = b;
  return 0;
}

int mnt753_bases_free(mnt753_bases* b) {
  if (!b) return 0;
  OnDevice on(b->device);
  if (b->pending && b->pending_n) (void)hipStreamSynchronize(b->pending_stream);   // never free buffers under a running MSM
  {
    std::lock_guard<std::mutex> l(g_sets_mu);
    g_sets.erase(b);
    for (mnt753_bases* o : g_sets)
      if (o->after_owner == b) { o->after_ev = nullptr; o->after_owner = nullptr; }   // its event is about to be destroyed
  }
  msm_free_workspace(b);
  {
    // the last base set of a device gives the pooled level buffers back
    std::lock_guard<std::mutex> l(g_sets_mu);
    PairPool& pool = pair_pool_of(b);
    if (pool.refs > 0 && b->registered) --pool.refs;
    if (pool.refs == 0) {
      for (int i = 0; i < 4; ++i) { if (pool.buf[i]) (void)hipFree(pool.buf[i]); pool.buf[i] = nullptr; pool.cap[i] = 0; }
      if (pool.last_acc) { (void)hipEventDestroy(pool.last_acc); pool.last_acc = nullptr; }
      pool.used = false;
    }
  }
  if (b->d_aff) (void)hipFree(b->d_aff);
  if (b->d_inf) (void)hipFree(b->d_inf);
  for (int i = 0; i < 5; ++i) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->ev_dep) (void)hipEventDestroy(b->ev_dep);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
  return 0;
}

size_t mnt753_bases_size(const mnt753_bases* b) { return b ? b->n : 0; }

int mnt753_msm_order_after(mnt753_bases* b, const mnt753_bases* first) {
  if (!b || !first || b == first) return set_error(MNT753_EINVAL, "msm_order_after: two different base sets");
  if (b->device != first->device) return set_error(MNT753_EINVAL, "msm_order_after: the two base sets live on different devices");
  std::lock_guard<std::mutex> l(g_sets_mu);
  if (!g_sets.count(const_cast<mnt753_bases*>(first)) || !g_sets.count(b)) return set_error(MNT753_EINVAL, "msm_order_after: not a live base set");
  // first's event, recorded behind the accumulate kernel of its LATEST mnt753_msm_start at the time b starts (null before first's
  // first start: no wait).  It stays first's: mnt753_bases_free(first) clears the reference, so b never waits on a destroyed event.
  b->after_ev = first->ev[2];
  b->after_owner = first;
  return 0;
}

int mnt753_msm(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n,
               uint64_t* out_projective, void* stream) {
  if (!b || !out_projective || (n && !scalars)) return set_error(MNT753_EINVAL, "msm: null argument");
  if (base_offset + n > b->n) return set_error(MNT753_EINVAL, "msm: base_offset + n exceeds the base set");
  if (int rc = require_device()) return rc;
  OnDevice on(b->device);
  hipStream_t st = (hipStream_t)stream;
  if (b->curve == MNT753_CURVE_MNT4753)
    return b->group == MNT753_G1 ? msm_mnt4g1(b, base_offset, scalars, scalars_on_device, n, out_projective, st)
                                 : msm_mnt4g2(b, base_offset, scalars, scalars_on_device, n, out_projective, st);
  return b->group == MNT753_G1 ? msm_mnt6g1(b, base_offset, scalars, scalars_on_device, n, out_projective, st)
                               : msm_mnt6g2(b, base_offset, scalars, scalars_on_device, n, out_projective, st);
}

// asynchronous pair: start enqueues the whole MSM and returns; finish waits for it and writes the result
int mnt753_msm_start(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, void* stream) {
  if (!b || (n && !scalars)) return set_error(MNT753_EINVAL, "msm_start: null argument");
  if (base_offset + n > b->n) return set_error(MNT753_EINVAL, "msm_start: base_offset + n exceeds the base set");
  if (int rc = require_device()) return rc;
  OnDevice on(b->device);
  hipStream_t st = (hipStream_t)stream;
  if (!st) {
    // the base set's own non-blocking stream, ordered after everything already enqueued on the default stream
    if (!b->own_stream) HIP_TRY(create_msm_stream(&b->own_stream, b->group));
    if (!b->ev_dep) HIP_TRY(hipEventCreateWithFlags(&b->ev_dep, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->ev_dep, nullptr));
    HIP_TRY(hipStreamWaitEvent(b->own_stream, b->ev_dep, 0));
    st = b->own_stream;
  }
  const bool was_pending = b->pending != 0;
  int rc;
  if (b->curve == MNT753_CURVE_MNT4753)
    rc = b->group == MNT753_G1 ? msm_start_mnt4g1(b, base_offset, scalars, scalars_on_device, n, st)
                               : msm_start_mnt4g2(b, base_offset, scalars, scalars_on_device, n, st);
  else
    rc = b->group == MNT753_G1 ? msm_start_mnt6g1(b, base_offset, scalars, scalars_on_device, n, st)
                               : msm_start_mnt6g2(b, base_offset, scalars, scalars_on_device, n, st);
  if (rc && !was_pending) b->pending = 0;   // a failed start leaves nothing in flight
  return rc;
}

int mnt753_msm_finish(mnt753_bases* b, uint64_t* out_projective) {
  if (!b || !out_projective) return set_error(MNT753_EINVAL, "msm_finish: null argument");
  if (int rc = require_device()) return rc;
  OnDevice on(b->device);
  if (b->curve == MNT753_CURVE_MNT4753)
    return b->group == MNT753_G1 ? msm_finish_mnt4g1(b, out_projective) : msm_finish_mnt4g2(b, out_projective);
  return b->group == MNT753_G1 ? msm_finish_mnt6g1(b, out_projective) : msm_finish_mnt6g2(b, out_projective);
}

int mnt753_msm_set_window_table(int mode) {
  const int old = g_window_table_mode;
  g_window_table_mode = mode != 0 ? 1 : 0;
  return old;
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

int mnt753_msm_last_pair_levels(void) { return g_last_pair_levels; }
int mnt753_msm_last_irr_levels(void) { return g_last_irr_levels; }

int mnt753_msm_last_plan(int out[4]) {
  if (!out) return set_error(MNT753_EINVAL, "msm_last_plan: null");
  memcpy(out, g_last_plan, sizeof(g_last_plan));
  return 0;
}

}  // extern "C"
