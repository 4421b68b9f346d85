// Wave-uniform point-operation programs.
//
// Same idea as pt_vm (curve753.cuh) -- a group operation is a micro-program with ONE inlined instance of the
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
#include "curve753.cuh"

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

}  // namespace mnt753
