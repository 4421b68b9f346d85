// mnt753_self_test (include/mnt753_hip.h): known-answer checks INSIDE the product, run once per process at parameter-load time.
//
// Why it exists: the proof a prover writes is only as sound as the build that computed it, and hipcc has miscompiled kernels of this
// library three times (tools/compiler_repro/): wrong sums on the GPU from source that is right on the CPU.  The test
// suite catches that for THIS toolchain; a user's box with another ROCm would run unverified code.  The reference carries a hook of the
// same purpose around its prover (libsnark/main.cpp:295-343: the debug check of the proof against the keys).
//
// What runs (every expected word is data of the reference's own code, mnt753_selftest_data.h, minted by tools/gen_selftest_data.py):
//   level 0  host tails: read_g1 / read_g2 decoding, B::G1_add as addition and as doubling, affine output, on one libff record per group
//   level 1  + per (curve, group) a 256-point MSM -- zero and one scalars, an identity base, every base seventeen times over (equal
//            points meet inside buckets) -- against libff's multi_exp_with_mixed_addition, twice: as a set of this size runs, and with
//            one regular and one irregular batched-affine level in front of the accumulation (the kernels of the large sets);
//            per curve compute_H on a 2^8 domain against libfqfft's call sequence
//   level 2  + the same MSMs over a window table (the doubling chains of k_precompute_windows: ~20 ms per group, latency of one chain)
// B::init_public_params runs level 1 for ITS curve (mnt753_self_test_curve: ~75 ms, most of it the hipMallocs of four small base sets);
// `main_hip <curve> self-test` runs level 2 for both.  MNT753_SELFTEST=0 skips it.
// MNT753_SELFTEST_CORRUPT=<k> (tests) flips one bit in the expected words of check k (both curves: 8 host checks, then per curve 2 x 2 MSM checks --
// 4 x 2 at level 2 -- and compute_H: 18 checks at level 1, 26 at level 2; one curve: 4 + 5 = 9 at level 1), which must then be reported.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common_host.hpp"
#include "mnt753_selftest_data.h"

namespace mnt753 {
extern int g_force_pair_levels, g_force_irr_levels, g_window_table_mode;
}
using namespace mnt753;

namespace {
struct GroupData { const uint64_t *bases, *msm, *record; };
const GroupData GROUPS[2][2] = {{{ST_BASES_0_1, ST_MSM_0_1, ST_GROUP_0_1}, {ST_BASES_0_2, ST_MSM_0_2, ST_GROUP_0_2}},
                                {{ST_BASES_1_1, ST_MSM_1_1, ST_GROUP_1_1}, {ST_BASES_1_2, ST_MSM_1_2, ST_GROUP_1_2}}};
const uint64_t* const ONE[2] = {ST_ONE_0, ST_ONE_1};
const uint64_t* const H_EXPECT[2] = {ST_H_0, ST_H_1};
const char* const CURVE_NAME[2] = {"MNT4753", "MNT6753"};

struct Ctx {
  int corrupt = -1, check = 0;
  std::string failed;
  // compare `n` words with the expected ones; check number `check` (counted in call order) has one bit flipped under the test switch
  bool same(const uint64_t* got, const uint64_t* want, size_t n, const char* what, int curve, int group) {
    std::vector<uint64_t> w(want, want + n);
    if (check == corrupt) w[n / 2] ^= 0x10000ull;
    const bool ok = memcmp(got, w.data(), 8 * n) == 0;
    if (!ok && failed.empty()) {
      char buf[200];
      snprintf(buf, sizeof(buf), "self-test check %d failed: %s (%s%s%s)", check, what, CURVE_NAME[curve], group ? (group == 1 ? " G1" : " G2") : "", "");
      failed = buf;
    }
    ++check;
    return ok;
  }
};

int group_record(Ctx& c, int curve, int group) {
  const size_t aw = mnt753_affine_words(curve, group), pw = mnt753_projective_words(curve, group);
  const uint64_t* r = GROUPS[curve][group - 1].record;   // P, Q, P + Q, 2 P (affine)
  std::vector<uint64_t> P(pw), Q(pw), S(pw), out(aw);
  if (int rc = mnt753_point_from_affine(curve, group, r, P.data())) return rc;
  if (int rc = mnt753_point_from_affine(curve, group, r + aw, Q.data())) return rc;
  if (int rc = mnt753_point_add(curve, group, P.data(), Q.data(), S.data())) return rc;
  if (int rc = mnt753_point_to_affine(curve, group, S.data(), out.data())) return rc;
  c.same(out.data(), r + 2 * aw, aw, "B::G1_add / G2 addition on the host against libff's P + Q", curve, group);
  if (int rc = mnt753_point_add(curve, group, P.data(), P.data(), S.data())) return rc;
  if (int rc = mnt753_point_to_affine(curve, group, S.data(), out.data())) return rc;
  c.same(out.data(), r + 3 * aw, aw, "P + P on the host against libff's doubling", curve, group);
  return 0;
}

int msm_checks(Ctx& c, int curve, int group, int level) {
  const size_t aw = mnt753_affine_words(curve, group), pw = mnt753_projective_words(curve, group);
  const size_t n = MNT753_SELFTEST_N_MSM;
  const GroupData& g = GROUPS[curve][group - 1];
  std::vector<uint64_t> pts(n * aw), sc(12 * n), proj(pw), aff(aw);
  for (size_t i = 0; i < n; ++i) memcpy(&pts[i * aw], g.bases + (i % MNT753_SELFTEST_N_BASES) * aw, 8 * aw);
  memset(&pts[MNT753_SELFTEST_IDENTITY_AT * aw], 0, 8 * aw);
  if (int rc = mnt753_synth_scalars(curve, MNT753_SELFTEST_SEED_MSM + 16 * curve + group, n, sc.data())) return rc;
  memset(&sc[12 * MNT753_SELFTEST_ZERO_AT], 0, 96);
  memcpy(&sc[12 * MNT753_SELFTEST_ONE_AT], ONE[curve], 96);
  struct Restore {
    int pair = g_force_pair_levels, irr = g_force_irr_levels, table = g_window_table_mode;
    ~Restore() { g_force_pair_levels = pair; g_force_irr_levels = irr; g_window_table_mode = table; }
  } restore;
  for (int pass = 0; pass < (level >= 2 ? 2 : 1); ++pass) {
    g_window_table_mode = pass == 0 ? 1 : 2;          // 1: no table for a set this small; 2: a table whatever the size (this file only)
    mnt753_bases* bs = nullptr;
    if (int rc = mnt753_bases_create(curve, group, pts.data(), 0, n, &bs)) return rc;
    int rc = 0;
    for (int forced = 0; forced < 2 && rc == 0; ++forced) {
      g_force_pair_levels = forced ? 1 : -1;
      g_force_irr_levels = forced ? 1 : -1;
      rc = mnt753_msm(bs, 0, sc.data(), 0, n, proj.data(), nullptr);
      if (rc == 0) rc = mnt753_point_to_affine(curve, group, proj.data(), aff.data());
      if (rc == 0)
        c.same(aff.data(), g.msm, aw, pass ? (forced ? "256-point MSM over a window table, batched-affine levels" : "256-point MSM over a window table")
                                           : (forced ? "256-point MSM with batched-affine levels against libff's multi_exp" : "256-point MSM against libff's multi_exp"), curve, group);
    }
    g_force_pair_levels = g_force_irr_levels = -1;
    (void)mnt753_bases_free(bs);
    if (rc) return rc;
  }
  return 0;
}

int h_check(Ctx& c, int curve) {
  const size_t m = (size_t)1 << MNT753_SELFTEST_LOG_H;
  mnt753_domain* d = nullptr;
  if (int rc = mnt753_domain_create(curve, m, &d)) return rc;
  void* dev = nullptr;
  int rc = mnt753_dev_alloc(&dev, 96 * (4 * m + 1));
  std::vector<uint64_t> host(12 * (m + 1));
  uint64_t* v[4];
  for (int k = 0; k < 4 && rc == 0; ++k) {
    v[k] = reinterpret_cast<uint64_t*>(dev) + 12 * m * k;
    if (k < 3) {
      rc = mnt753_synth_scalars(curve, MNT753_SELFTEST_SEED_H + 16 * curve + k, m, host.data());
      if (rc == 0) rc = mnt753_copy_h2d(v[k], host.data(), 96 * m);
    }
  }
  if (rc == 0) rc = mnt753_compute_h(d, v[0], v[1], v[2], v[3], nullptr);
  if (rc == 0) rc = mnt753_copy_d2h(host.data(), v[3], 96 * (m + 1));
  if (dev) (void)mnt753_dev_free(dev);
  (void)mnt753_domain_free(d);
  if (rc) return rc;
  uint64_t got[36];
  memset(got, 0, sizeof(got));
  for (size_t i = 0; i <= m; ++i)
    for (int k = 0; k < 12; ++k) got[k] = got[k] * MNT753_SELFTEST_CK_MUL + host[12 * i + k];
  memcpy(got + 12, &host[0], 96);
  memcpy(got + 24, &host[12 * (m - 1)], 96);
  c.same(got, H_EXPECT[curve], 36, "compute_H on a 2^8 domain against libfqfft's call sequence", curve, 0);
  return 0;
}
}  // namespace

// curve: MNT753_CURVE_MNT4753 / _MNT6753, or -1 for both.  Check numbers (MNT753_SELFTEST_CORRUPT) count in call order over the
// curves that run: per curve 4 host checks, then 2 x 2 MSM checks (4 x 2 at level 2) and compute_H.
static int self_test_curves(int curve_lo, int curve_hi, int level) {
  Ctx c;
  if (const char* e = getenv("MNT753_SELFTEST_CORRUPT")) c.corrupt = atoi(e);
  for (int curve = curve_lo; curve <= curve_hi; ++curve)
    for (int group = 1; group <= 2; ++group)
      if (int rc = group_record(c, curve, group)) return rc;
  if (level >= 1) {
    if (int rc = require_device()) return rc;
    for (int curve = curve_lo; curve <= curve_hi; ++curve) {
      for (int group = 1; group <= 2; ++group)
        if (int rc = msm_checks(c, curve, group, level)) return rc;
      if (int rc = h_check(c, curve)) return rc;
    }
  }
  if (!c.failed.empty()) return set_error(MNT753_ESELFTEST, c.failed.c_str());
  return 0;
}

extern "C" int mnt753_self_test(int level) {
  if (level < 0 || level > 2) return set_error(MNT753_EINVAL, "self_test: level 0, 1 or 2");
  return self_test_curves(0, 1, level);
}
extern "C" int mnt753_self_test_curve(int curve, int level) {
  if (level < 0 || level > 2 || curve < 0 || curve > 1) return set_error(MNT753_EINVAL, "self_test_curve: curve 0 or 1, level 0, 1 or 2");
  return self_test_curves(curve, curve, level);
}
