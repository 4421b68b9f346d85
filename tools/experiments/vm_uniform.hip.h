// Wave-uniform point-operation programs.
//
// Same idea as pt_vm (curve753.hip.h) -- a group operation is a micro-program with ONE inlined instance of the
// multiplier, the subtractor and the adder -- but the program counter is a plain loop counter, identical in every
// lane, and per-lane differences are expressed as COMMIT MASKS: every lane executes every step, a lane that does not
// take part simply does not write its accumulator.  Consequences on gfx950:
//   * `switch (pc)` compiles to scalar branches (s_cbranch_scc) -- no EXEC-mask bookkeeping around 30 giant cases.
//     (hipcc 7.2 miscompiled a nested switch under a lane-divergent pc, and the divergent VMs are brittle.)
//   * field additions / subtractions are steps of the program too, so each kernel has exactly one copy of each:
//     the hot loop is ~35 KB and stays inside the 64 KB instruction cache (the divergent VM with inlined
//     subtractions was 66-87 KB for G1 and 150-320 KB for G2).
//   * rare per-lane exceptions (the two operands are the same point -> doubling) are handled after the common
//     program with a wave-level vote: if any lane needs it, the whole wave runs the doubling program with that
//     lane's commit mask.
//
// Bucket accumulator: XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, identity ZZ = 0), mixed addition madd-2008-s =
// 8M + 2S (10 products; the homogeneous projective form of the reference needs 11), doubling of an affine point
// mdbl-2008-s-1.  The result is the same group element; buckets are converted to homogeneous projective
// (X*ZZZ : Y*ZZ : ZZ*ZZZ) by a separate pass.
#pragma once
#include "../../snark-challenge-prover-reference_amd/csrc/curve753.hip.h"

namespace mnt753 {

HD bool wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any(p) != 0;
#else
  return p;
#endif
}

template <class C>
struct XyzzAcc {
  typename C::F::E X, Y, ZZ, ZZZ;
};

enum : int { UPOST_NONE = 0, UPOST_R_MINUS_C = 1, UPOST_C_MINUS_R = 2, UPOST_R_PLUS_C = 3 };

// A += Q for the lanes in `act` (A not the identity, Q affine); lanes whose A equals Q are returned in need_dbl
// and left untouched.  13 steps: 10 products, 7 subtractions.
template <class C>
HD void xyzz_madd_uniform(XyzzAcc<C>& A, const typename C::F::E& qx, const typename C::F::E& qy, bool act, bool& need_dbl) {
  using F = typename C::F;
  using E = typename F::E;
  E u, v, w, t3, t5, a, b, d, r;   // u = R, v = P -> Q' -> Q'-X3 -> R(Q'-X3), w = X3, t3 = PP, t5 = PPP
  need_dbl = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
  for (int pc = 0; pc < 13; ++pc) {
    bool do_mul = true;
    switch (pc) {
      case 0: a = qx; b = A.ZZ; break;                 // U2 = x2 ZZ1
      case 1: a = qy; b = A.ZZZ; break;                // S2 = y2 ZZZ1
      case 2: a = v; b = v; break;                     // PP
      case 3: a = v; b = t3; break;                    // PPP
      case 4: a = A.X; b = t3; break;                  // Q' = X1 PP
      case 5: a = u; b = u; break;                     // R^2
      case 6: case 7: a = w; do_mul = false; break;    // X3 = R^2 - PPP - 2Q'
      case 8: a = v; do_mul = false; break;            // Q' - X3
      case 9: a = u; b = v; break;                     // R (Q' - X3)
      case 10: a = A.Y; b = t5; break;                 // Y1 PPP
      case 11: a = A.ZZ; b = t3; break;                // ZZ3 = ZZ1 PP
      default: a = A.ZZZ; b = t5; break;               // 12: ZZZ3 = ZZZ1 PPP
    }
    if (do_mul) F::mul(r, a, b); else r = a;
    int post = UPOST_NONE;
    switch (pc) {
      case 0: d = A.X; post = UPOST_R_MINUS_C; break;  // P = U2 - X1
      case 1: d = A.Y; post = UPOST_R_MINUS_C; break;  // R = S2 - Y1
      case 5: d = t5; post = UPOST_R_MINUS_C; break;
      case 6: case 7: d = v; post = UPOST_R_MINUS_C; break;
      case 8: d = w; post = UPOST_R_MINUS_C; break;
      case 10: d = v; post = UPOST_C_MINUS_R; break;   // Y3 = R (Q' - X3) - Y1 PPP
      default: break;
    }
    if (post == UPOST_C_MINUS_R) { E t = r; r = d; d = t; }
    if (post != UPOST_NONE) F::sub(r, r, d);
    switch (pc) {
      case 0: v = r; break;
      case 1:
        u = r;
        need_dbl = act && F::is_zero(u) && F::is_zero(v);
        act = act && !need_dbl;
        break;
      case 2: t3 = r; break;
      case 3: t5 = r; break;
      case 4: v = r; break;
      case 5: case 6: case 7: w = r; break;
      case 8: case 9: v = r; break;
      case 10: if (act) { A.Y = r; A.X = w; } break;   // X1 was last read at step 4, Y1 at step 10
      case 11: if (act) A.ZZ = r; break;
      default: if (act) A.ZZZ = r; break;
    }
  }
}

// A = 2Q for the lanes in `act` (Q affine).  13 steps: 7 products, 4 additions, 4 subtractions (mdbl-2008-s-1).
template <class C>
HD void xyzz_mdbl_uniform(XyzzAcc<C>& A, const typename C::F::E& qx, const typename C::F::E& qy, bool act) {
  using F = typename C::F;
  using E = typename F::E;
  E u, v, w, t3, t5, a, b, d, r;   // u = M, v = U -> S -> S-X3 -> M(S-X3), w = x2^2 -> X3, t3 = V, t5 = W
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
  for (int pc = 0; pc < 13; ++pc) {
    bool do_mul = true;
    switch (pc) {
      case 0: a = qy; do_mul = false; break;           // U = 2 y2
      case 1: a = v; b = v; break;                     // V = U^2
      case 2: a = v; b = t3; break;                    // W = U V
      case 3: a = qx; b = t3; break;                   // S = x2 V
      case 4: a = qx; b = qx; break;                   // x2^2
      case 5: a = w; do_mul = false; break;            // 2 x2^2
      case 6: case 7: a = u; do_mul = false; break;    // 3 x2^2 ; M = 3 x2^2 + a
      case 8: a = u; b = u; break;                     // M^2
      case 9: a = w; do_mul = false; break;            // X3 = M^2 - 2S
      case 10: a = v; do_mul = false; break;           // S - X3
      case 11: a = u; b = v; break;                    // M (S - X3)
      default: a = t5; b = qy; break;                  // 12: W y2
    }
    if (do_mul) F::mul(r, a, b); else r = a;
    int post = UPOST_NONE;
    switch (pc) {
      case 0: d = qy; post = UPOST_R_PLUS_C; break;
      case 5: case 6: d = w; post = UPOST_R_PLUS_C; break;
      case 7: C::coeff_a(d); post = UPOST_R_PLUS_C; break;
      case 8: case 9: d = v; post = UPOST_R_MINUS_C; break;
      case 10: d = w; post = UPOST_R_MINUS_C; break;
      case 12: d = v; post = UPOST_C_MINUS_R; break;   // Y3 = M (S - X3) - W y2
      default: break;
    }
    if (post == UPOST_C_MINUS_R) { E t = r; r = d; d = t; }
    if (post == UPOST_R_PLUS_C) F::add(r, r, d);
    else if (post != UPOST_NONE) F::sub(r, r, d);
    switch (pc) {
      case 0: v = r; break;
      case 1: t3 = r; break;
      case 2: t5 = r; break;
      case 3: v = r; break;
      case 4: w = r; break;
      case 5: case 6: case 7: u = r; break;
      case 8: case 9: w = r; break;
      case 10: case 11: v = r; break;
      default: if (act) { A.Y = r; A.X = w; A.ZZ = t3; A.ZZZ = t5; } break;
    }
  }
}

// XYZZ -> homogeneous projective (X*ZZZ : Y*ZZ : ZZ*ZZZ); the identity (ZZ = 0) maps to Z = 0
template <class C>
HD void xyzz_to_proj_uniform(Proj<C>& P, const XyzzAcc<C>& A) {
  using F = typename C::F;
  using E = typename F::E;
  E a, b, r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
  for (int pc = 0; pc < 3; ++pc) {
    switch (pc) {
      case 0: a = A.X; b = A.ZZZ; break;
      case 1: a = A.Y; b = A.ZZ; break;
      default: a = A.ZZ; b = A.ZZZ; break;
    }
    F::mul(r, a, b);
    switch (pc) {
      case 0: P.X = r; break;
      case 1: P.Y = r; break;
      default: P.Z = r; break;
    }
  }
}


// ---- kernels of the experiment (moved out of csrc/msm_kernels.hip.h in round 2; include AFTER msm_kernels.hip.h) ----
// ---- bucket accumulation, wave-uniform version (vm_uniform.hip.h) ------------------------------------------
// Same lane schedule as k_bucket_accumulate (lane t sums sorted entries [t*T, (t+1)*T), whole buckets go to the
// bucket array, the first / last partial run to the edge slots), but:
//   * every lane consumes exactly one entry per iteration of a loop whose trip count is T for the whole wave; the
//     mixed addition is the uniform 13-step XYZZ program with a per-lane commit mask;
//   * a finished run is written RAW (X, Y, ZZ, ZZZ) with plain stores -- no field arithmetic on the flush path;
//     k_xyzz_to_proj converts the written slots to homogeneous projective afterwards (3 products per bucket).
template <class C>
constexpr int xyzz_words() { return 4 * C::F::DEG * FPS_WORDS; }

template <class C>
__device__ __forceinline__ void xyzz_store(uint32_t* p, const XyzzAcc<C>& A) {
  constexpr int EW = C::F::DEG * FPS_WORDS;
  e_store<typename C::F>(p, A.X);
  e_store<typename C::F>(p + EW, A.Y);
  e_store<typename C::F>(p + 2 * EW, A.ZZ);
  e_store<typename C::F>(p + 3 * EW, A.ZZZ);
}

template <class C>
__global__ void __launch_bounds__(256, vm_waves<C>()) k_bucket_accumulate_u(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                               uint32_t* __restrict__ raw_buckets, uint32_t* __restrict__ raw_edges,
                                                               uint32_t* __restrict__ edge_bucket, uint8_t* __restrict__ bucket_state,
                                                               uint32_t T, uint32_t n_lanes) {
  using F = typename C::F;
  using E = typename F::E;
  constexpr int EW = F::DEG * FPS_WORDS;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_lanes) return;
  const uint32_t total = offsets[n_buckets];
  uint64_t e0 = (uint64_t)t * T;
  if (e0 >= total) {
    edge_bucket[2 * t] = EDGE_NONE;
    edge_bucket[2 * t + 1] = EDGE_NONE;
    return;
  }
  uint32_t e = (uint32_t)e0;
  const uint32_t end = (e0 + T < total) ? (uint32_t)(e0 + T) : total;
  uint32_t lo = 0, hi = n_buckets;   // first x with offsets[x + 1] > e
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid + 1] > e) hi = mid; else lo = mid + 1;
  }
  uint32_t b = lo;
  uint32_t next = offsets[b + 1];
  bool first_run = true, acc_zero = true;
  XyzzAcc<C> acc;
  E qx, qy;
  F::zero(acc.X); F::zero(acc.Y); F::zero(acc.ZZ); F::zero(acc.ZZZ);
  F::zero(qx); F::zero(qy);
#pragma nounroll
  for (uint32_t it = 0; it < T; ++it) {
    const bool live = e < end;
    if (live && e == next) {
      // bucket b is complete inside this segment: write it out raw and start the next one
      if (first_run) {
        xyzz_store<C>(raw_edges + (size_t)(2 * t) * xyzz_words<C>(), acc);
        edge_bucket[2 * t] = b;
        first_run = false;
      } else {
        xyzz_store<C>(raw_buckets + (size_t)b * xyzz_words<C>(), acc);
        bucket_state[b] = 1;
      }
      acc_zero = true;
      do { ++b; next = offsets[b + 1]; } while (next == e);
    }
    if (live) {
      const uint32_t s = sorted[e];
      const uint32_t* src = bases + (size_t)(s & 0x7fffffffu) * aff_words<C>();
      e_load<F>(qx, src);
      e_load<F>(qy, src + EW);
      if (s & 0x80000000u) F::neg(qy, qy);
    }
    const bool start = live && (acc_zero || F::is_zero(acc.ZZ));   // empty accumulator, or a run that summed to the identity
    if (start) {
      acc.X = qx; acc.Y = qy; F::one(acc.ZZ); F::one(acc.ZZZ);
      acc_zero = false;
    }
    bool need_dbl;
    xyzz_madd_uniform<C>(acc, qx, qy, live && !start, need_dbl);
    if (wave_any(need_dbl)) xyzz_mdbl_uniform<C>(acc, qx, qy, need_dbl);
    if (live) ++e;
  }
  // the last run of the segment is an edge piece
  if (first_run) {
    xyzz_store<C>(raw_edges + (size_t)(2 * t) * xyzz_words<C>(), acc);
    edge_bucket[2 * t] = b;
    // single-run lane: the second slot is an identity piece (ZZ = 0) of the same bucket: keeps the slot list gap-free
    F::zero(acc.X); F::zero(acc.Y); F::zero(acc.ZZ); F::zero(acc.ZZZ);
    xyzz_store<C>(raw_edges + (size_t)(2 * t + 1) * xyzz_words<C>(), acc);
    edge_bucket[2 * t + 1] = b;
  } else {
    xyzz_store<C>(raw_edges + (size_t)(2 * t + 1) * xyzz_words<C>(), acc);
    edge_bucket[2 * t + 1] = b;
  }
}

// raw XYZZ slot j -> homogeneous projective slot j, for the slots marked by state8 (buckets) or ids32 != EDGE_NONE (edges)
template <class C>
__global__ void __launch_bounds__(256, 1) k_xyzz_to_proj(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out,
                                                        const uint8_t* __restrict__ state8, const uint32_t* __restrict__ ids32, uint32_t n) {
  using F = typename C::F;
  constexpr int EW = F::DEG * FPS_WORDS;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const bool on = state8 ? (state8[j] != 0) : (ids32[j] != EDGE_NONE);
  if (!wave_any(on)) return;
  XyzzAcc<C> A;
  F::zero(A.X); F::zero(A.Y); F::zero(A.ZZ); F::zero(A.ZZZ);
  if (on) {
    const uint32_t* p = raw + (size_t)j * xyzz_words<C>();
    e_load<F>(A.X, p); e_load<F>(A.Y, p + EW); e_load<F>(A.ZZ, p + 2 * EW); e_load<F>(A.ZZZ, p + 3 * EW);
  }
  Proj<C> P;
  xyzz_to_proj_uniform<C>(P, A);
  if (on) proj_store<C>(out + (size_t)j * proj_words<C>(), P);
}


}  // namespace mnt753
