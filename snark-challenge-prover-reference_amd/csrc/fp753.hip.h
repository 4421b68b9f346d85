// 753-bit prime-field arithmetic for gfx950 (MI355X) -- device representation.
//
// Replaces, on the device, libff's Fp_model<12, modulus> (reference:
// depends/libff/libff/algebra/fields/fp.tcc:161-186 mul_reduce, :405-417 +=, :491-508 -=).
//
// Design (measured, profiles/r01/valu_rates_mi355x.txt + mulbench_mi355x.txt):
//   * v_mad_u64_u32 (32x32+64 -> 64) issues in ~5 cycles per SIMD, only ~1.25x a v_mul_lo_u32
//     and ~1.7x a plain v_add_u32, while a carry-propagating v_add_co/v_addc_co pair costs
//     ~1.5x a MAD.  Carries, not multiplies, are the expensive thing on this chip.
//   * so an element is 27 limbs x 28 bits (756 bits) instead of 24 x 32: a 28x28-bit product is
//     < 2^56 and 54 of them fit a 64-bit column accumulator, which makes the whole Montgomery
//     product (finely integrated product scanning) a pure chain of 1458 v_mad_u64_u32 with one
//     shift+mask per column and NO carry instructions.
//   * Montgomery radix R' = 2^756; values are kept lazily in [0, 2p) (2p < 2^754):
//     inputs < 2p give outputs < p(4p/R' + 1) < 1.45p, so a multiply needs no final subtraction.
//   * the wire format (12 x u64, Montgomery R = 2^768, canonical) is converted at kernel
//     boundaries with one multiply by a constant (k_in / k_out below).
//
// Everything is __host__ __device__ so the same source can be exercised by a CPU harness during
// development; the product only ever runs it on the GPU.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "mnt753_constants.h"

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace mnt753 {

template <int M>
struct Fp {
  uint32_t l[NL];
};

// (One level of subtractive Karatsuba on the product half -- 561 + 729 multiply-adds instead of 729 + 729 -- was built and measured in
// round 3: +14 % in the microbenchmark at one wave per SIMD, nothing in the product's kernels, which have no registers for its 14-entry
// window of 64-bit sums (profiles/r03/ab_karatsuba.txt, mul_variants_mi355x.txt).  It left the product in round 5; the formulation is
// kept in tools/experiments/mul_variants.hip.)
// ---- Montgomery product, radix 2^756, two interleaved column accumulators ------------------
// r = a*b*2^-756 mod p, r < 2p provided a*b < 4p^2 (e.g. a,b < 2p; or a < 4p, b < p).
// Limbs of a may be up to 2^29 (one un-normalised addition) -- the column bound still holds.
template <int M>
HD void fp_mul(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint64_t acc = 0, acc2 = 0;
  uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (uint64_t)m[k] * FPC[M].p[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}

// r = a*a*2^-756 mod p.  Cross products a_i*a_j (i < j) are taken once against the pre-doubled operand 2a (limbs < 2^29),
// the diagonal once: 351 + 27 product MADs instead of 729; the reduction half is unchanged (729).  1107 MADs vs 1458.
template <int M>
HD void fp_sqr(Fp<M>& r, const Fp<M>& a) {
  uint64_t acc = 0, acc2 = 0;
  uint32_t m[NL], d[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) d[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (uint64_t)m[k] * FPC[M].p[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}

// r = (a1*b1 + a2*b2) * 2^-756 mod p with ONE Montgomery reduction: 2*729 + 729 = 2187 multiply-adds instead of the
// 2916 of two separate products.  All four inputs in [0, 2p): the sum is < 8p^2, so r < p(8p/R' + 1) < 2p because
// 8p < 0.89 R' for both moduli (p ~ 1.77 * 2^752).  A column holds at most 54 + 27 products < 2^57 and one carry, well
// inside 64 bits.  This is the multiplier of the lane-split extension fields (curve753.hip.h): every component of a
// product in Fq2 is a sum of two base-field products.
template <int M>
HD void fp_mul2(Fp<M>& r, const Fp<M>& a1, const Fp<M>& b1, const Fp<M>& a2, const Fp<M>& b2) {
  uint64_t acc = 0, acc2 = 0;
  uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
    for (int i = 0; i <= k; ++i) acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
    acc += acc2;
    acc2 = 0;
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (uint64_t)m[k] * FPC[M].p[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
    acc += acc2;
    acc2 = 0;
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}

// s (normalised limbs, value < 4p) -> r = s mod 2p, in [0, 2p)
template <int M>
HD void fp_reduce2p(Fp<M>& r, const uint32_t s[NL]) {
  uint32_t d[NL];
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int32_t t = (int32_t)s[i] - (int32_t)FPC[M].p2[i] + bw;
    d[i] = (uint32_t)t & LMASK;
    bw = t >> LB;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = bw < 0 ? s[i] : d[i];
}

// r = (a1*b1 + a2*b2 + a3*b3) * 2^-756 mod p, one reduction: 2916 multiply-adds (three separate products: 4374).
// Inputs in [0, 2p): the sum is < 12p^2 and the reduced value < p(12p/R' + 1) < 2.33p, so one conditional subtraction
// of 2p (fp_reduce2p, below) brings it back to [0, 2p).  A column holds at most 81 + 27 products < 2^56.
template <int M>
HD void fp_mul3(Fp<M>& r, const Fp<M>& a1, const Fp<M>& b1, const Fp<M>& a2, const Fp<M>& b2, const Fp<M>& a3, const Fp<M>& b3) {
  uint64_t acc = 0, acc2 = 0;
  uint32_t m[NL], s[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
    for (int i = 0; i <= k; ++i) acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
    acc += acc2;
    acc2 = 0;
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a3.l[i] * b3.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (uint64_t)m[k] * FPC[M].p[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a1.l[i] * b1.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
    acc += acc2;
    acc2 = 0;
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a3.l[i] * b3.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += acc2;
    acc2 = 0;
    s[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  s[NL - 1] = (uint32_t)acc;
  fp_reduce2p<M>(r, s);
}

template <int M>
HD void fp_add(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint32_t s[NL], c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t t = a.l[i] + b.l[i] + c;
    s[i] = t & LMASK;
    c = t >> LB;
  }
  fp_reduce2p<M>(r, s);
}

// r = a - b (mod p), computed as a - b + 2p then reduced into [0, 2p)
template <int M>
HD void fp_sub(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint32_t s[NL];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)FPC[M].p2[i] + c;
    s[i] = (uint32_t)t & LMASK;
    c = t >> LB;
  }
  fp_reduce2p<M>(r, s);
}

// r = a / 2 (mod p).  a: non-negative limbs below 2^30 (normalised or a limb-wise sum of up to four normalised elements), value
// below 8p; r: normalised limbs, value (a + p) / 2 at most.  The value's parity is the parity of limb 0 (every higher limb weighs a
// multiple of 2^28); an odd value takes p first (p is odd), then one pass carries and one shifts.  ~190 instructions.
template <int M>
HD void fp_half(Fp<M>& r, const Fp<M>& a) {
  const uint32_t odd = 0u - (a.l[0] & 1u);
  uint32_t s[NL], c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const uint32_t t = a.l[i] + (FPC[M].p[i] & odd) + c;
    if (i < NL - 1) { s[i] = t & LMASK; c = t >> LB; } else s[i] = t;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (s[i] >> 1) | (i + 1 < NL ? (s[i + 1] & 1u) << (LB - 1) : 0u);
}

// ---- lazy arithmetic for the batched-affine pairing levels (msm_kernels.hip.h, k_pair_level) ------------------------------
// A subtraction with carries and a conditional correction (fp_sub) is 270 VALU instructions, a sixth of a product, and an affine
// addition needs seven of them.  The 28-bit limbs leave four spare bits per word, so differences are taken LIMB-WISE with no
// carry at all (27 instructions; limbs become signed, |limb| < 2^29 after one addition / subtraction of 28-bit operands) and fed
// straight into a multiplier whose product half uses the signed multiply-add (v_mad_i64_i32, same rate as v_mad_u64_u32):
//   * fp_mul_s / fp_sqr_s: operands with int32 limbs, 27 |a_i| |b_j| + 27 2^56 < 2^63 (e.g. 2^29 x 2^28 or 2^30 x 2^28); the
//     Montgomery reduction is unchanged (m_k from the low 28 bits of the two's-complement column, arithmetic shifts).  Result:
//     limbs 0..25 in [0, 2^28), limb 26 SIGNED, value in (-|a b| / R', |a b| / R' + p).
//   * fp_norm: signed un-normalised limbs (|limb| < 2^30, |value| < 5p) -> limbs in [0, 2^28), value in [0.49p, 1.51p), a subset
//     of the lazy range [0, 2p) of everything else in this file: the quotient comes from the top limb alone (the lower limbs
//     move it by < 2^-20 p), one pass subtracts q p and carries -- 4 instructions per limb, and only the two coordinates an
//     addition hands on need it.
// Values mod p are unchanged by all of it, so results are bit-exact against the eager formulas (tools/host_fp_check.cpp).
// A limb as an int32 the compiler knows nothing about.  Where LLVM can prove one operand non-negative (a masked limb) it turns
// sext x sext into sext x zext and expands THAT into two unsigned multiply-adds and two moves per product (hipcc 7.2: the signed
// multiplier came out at 3700 instructions instead of 1650); behind an empty asm every product is one v_mad_i64_i32.
HD int32_t fp_opaque_limb(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(v));
#endif
  return (int32_t)v;
}
template <int M>
HD void fp_mul_s(Fp<M>& r, const Fp<M>& a_in, const Fp<M>& b_in) {
  int64_t acc = 0;
  uint64_t acc2 = 0;
  uint32_t m[NL];
  struct { int32_t l[NL]; } a, b;
#pragma unroll
  for (int i = 0; i < NL; ++i) { a.l[i] = fp_opaque_limb(a_in.l[i]); b.l[i] = fp_opaque_limb(b_in.l[i]); }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}
// a with int32 limbs, |a_i| < 2^29: cross terms once against 2a.  The value is >= 0 whatever the sign of a.
template <int M>
HD void fp_sqr_s(Fp<M>& r, const Fp<M>& a_in) {
  int64_t acc = 0;
  uint64_t acc2 = 0;
  uint32_t m[NL];
  int32_t d[NL];
  struct { int32_t l[NL]; } a;
#pragma unroll
  for (int i = 0; i < NL; ++i) { a.l[i] = fp_opaque_limb(a_in.l[i]); d[i] = fp_opaque_limb(a_in.l[i] << 1); }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (int64_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; ++i) acc += (int64_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}
// ---- the same two products for a step loop that keeps its operands where the multiplier reads them (round 6) -------------------
// fp_mul_s / fp_sqr_s take copies of their operands (fp_opaque_limb returns a NEW value: where the operand stays live the compiler
// has to copy it first) and hand back a fresh result that the caller then moves to wherever the next product wants it: in the step
// loop of k_pair_level that routing was ~80 register moves per product (msm_kernels.hip.h).  Here the operands pass through the empty
// asm IN PLACE (the variable itself is the opaque value from then on, no copy), and the product overwrites b: limb j of the result is
// produced in column 27 + j, which reads a_i, b_i only for i > j, so r_j can take the register of b_j.  (Whether it DOES is the register
// allocator's decision: pinning the destinations with tied asm operands -- "v_and_b32 %0, 0xfffffff, %1" : "=v"(r) : "v"(x), "0"(b_j) --
// made it copy b_j to a fresh register first, one more move per limb instead of one fewer; measured with tools/isa_walk.py, not kept.)
template <int M>
HD void fp_opaque_inplace(Fp<M>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < NL; ++i) asm("" : "+v"(a.l[i]));
#else
  (void)a;
#endif
}
// b <- a * b * 2^-756 (signed limbs, as fp_mul_s); a keeps its value
template <int M>
HD void fp_mul_s_ip(Fp<M>& b, Fp<M>& a) {
  int64_t acc = 0;
  uint64_t acc2 = 0;
  uint32_t m[NL];
  fp_opaque_inplace(a);
  fp_opaque_inplace(b);
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    b.l[k - NL] = (uint32_t)acc & LMASK;      // b_j is dead behind column 26 + j
    acc >>= LB;
  }
  b.l[NL - 1] = (uint32_t)acc;
}
// r <- a * a * 2^-756 (signed limbs, |a_i| < 2^29, as fp_sqr_s); a keeps its value
template <int M>
HD void fp_sqr_s_keep(Fp<M>& r, Fp<M>& a) {
  int64_t acc = 0;
  uint64_t acc2 = 0;
  uint32_t m[NL];
  int32_t d[NL];
  fp_opaque_inplace(a);
#pragma unroll
  for (int i = 0; i < NL; ++i) d[i] = fp_opaque_limb(a.l[i] << 1);
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (int64_t)(int32_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (int64_t)(int32_t)a.l[k / 2] * (int32_t)a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    m[k] = ((uint32_t)acc * FPC[M].inv) & LMASK;
    acc += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; ++i) acc += (int64_t)(int32_t)a.l[i] * d[k - i];
    if ((k & 1) == 0) acc += (int64_t)(int32_t)a.l[k / 2] * (int32_t)a.l[k / 2];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FPC[M].p[k - i];
    acc += (int64_t)acc2;
    acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}
// limb-wise a - b and a +- b (the sign chosen per lane): no carries, signed limbs
template <int M>
HD void fp_sub_raw(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] - b.l[i];
}
template <int M>
HD void fp_addsub_raw(Fp<M>& r, const Fp<M>& a, const Fp<M>& y, bool subtract) {
  const uint32_t mk = subtract ? 0xffffffffu : 0u, one = subtract ? 1u : 0u;   // a + (y ^ mk) + one = a - y
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + (y.l[i] ^ mk) + one;
}
// signed un-normalised limbs (|limb| < 2^30 - 2^28 |q|... see above: |value| < 5p) -> [0.49p, 1.51p), limbs in [0, 2^28)
template <int M>
HD void fp_norm(Fp<M>& r, const Fp<M>& a) {
  const float t = (float)(int32_t)a.l[NL - 1] * (1.0f / (float)FPC[M].p[NL - 1]) - 0.5f;
  int32_t nq = (int32_t)t;              // truncation towards zero ...
  nq -= (float)nq > t ? 1 : 0;          // ... made a floor: q = floor(top / p_top - 1/2)
  nq = -nq;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int32_t v = (int32_t)a.l[i] + nq * (int32_t)FPC[M].p[i] + c;
    if (i < NL - 1) { r.l[i] = (uint32_t)v & LMASK; c = v >> LB; } else r.l[i] = (uint32_t)v;
  }
}
// d = a - b limb-wise with a, b in [0, 2p): d = 0 (mod p) only for d in {-p, 0, p}, whose low 28 bits are 0, p_0 or -p_0 --
// a test that passes for three values in 2^28; the caller settles the rare hit exactly (fp_sub + fp_is_zero)
template <int M>
HD bool fp_raw_maybe_zero(const Fp<M>& d) {
  const uint32_t t = d.l[0] & LMASK;
  return t == 0u || t == FPC[M].p[0] || t == ((0u - FPC[M].p[0]) & LMASK);
}

template <int M>
HD void fp_zero(Fp<M>& r) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
}

template <int M>
HD void fp_one(Fp<M>& r) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = FPC[M].one[i];
}

template <int M>
HD void fp_neg(Fp<M>& r, const Fp<M>& a) {
  Fp<M> z;
  fp_zero(z);
  fp_sub(r, z, a);
}

// a in [0, 2p): a == 0 (mod p)  <=>  a is 0 or p
template <int M>
HD bool fp_is_zero(const Fp<M>& a) {
  uint32_t o0 = 0, o1 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    o0 |= a.l[i];
    o1 |= a.l[i] ^ FPC[M].p[i];
  }
  return o0 == 0 || o1 == 0;
}

// [0,2p) -> canonical [0,p)
template <int M>
HD void fp_canon(Fp<M>& r, const Fp<M>& a) {
  uint32_t d[NL];
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int32_t t = (int32_t)a.l[i] - (int32_t)FPC[M].p[i] + bw;
    d[i] = (uint32_t)t & LMASK;
    bw = t >> LB;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = bw < 0 ? a.l[i] : d[i];
}

// r = k*a mod p (lazily, r in [0, 2p)) for a small integer k < 256 and a in [0, 2p): curve / twist coefficients and
// extension-field non-residues (2, 11, 13, 26, 121).
//   V = k*a is formed limb-wise (no reduction needed: V < 2^762), the quotient q = floor(V / p) is estimated from the
//   top limb:  q' = floor( floor(V / 2^728) / (floor(p / 2^728) + 1) )  which satisfies  q - 1 <= q' <= q  (the relative
//   error of both truncations is < 2^-16), so V - q' p lies in [0, 2p) and needs no conditional subtraction.
// ~260 instructions; the previous double-and-add over lazy additions took ~1250 for k = 13.
template <int M>
HD void fp_mul_small(Fp<M>& r, const Fp<M>& a, unsigned k) {
  uint32_t t[NL + 1];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    c += (uint64_t)a.l[i] * k;
    t[i] = (uint32_t)c & LMASK;
    c >>= LB;
  }
  t[NL] = (uint32_t)c;
  const uint64_t vh = ((uint64_t)t[NL] << LB) | t[NL - 1];
  const uint32_t q = (uint32_t)(vh / (uint64_t)(FPC[M].p[NL - 1] + 1u));   // division by a compile-time constant
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    s += (int64_t)t[i] - (int64_t)((uint64_t)q * FPC[M].p[i]);
    r.l[i] = (uint32_t)s & LMASK;
    s >>= LB;   // arithmetic shift: s carries the (signed) borrow
  }
}

// ---- wire-format conversion -------------------------------------------------------------
// 24 little-endian u32 words (768 bits) -> 27 raw 28-bit limbs (no Montgomery change)
template <int M>
HD void fp_unpack(Fp<M>& r, const uint32_t w[24]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = LB * i, wi = bit >> 5, sh = bit & 31;
    uint32_t lo = w[wi];
    uint32_t hi = (wi + 1 < 24) ? w[wi + 1] : 0u;
    uint32_t v = sh == 0 ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
    r.l[i] = v & LMASK;
  }
}

// 27 normalised limbs (value < 2^756; caller guarantees < 2^768 trivially) -> 24 u32 words
template <int M>
HD void fp_pack(uint32_t w[24], const Fp<M>& a) {
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    // word j covers bits [32j, 32j+32): limbs floor(32j/28) and the next one or two
    const int bit = 32 * j, li = bit / LB, sh = bit - li * LB;  // sh in [0,28)
    uint64_t v = (uint64_t)a.l[li] >> sh;
    if (li + 1 < NL) v |= (uint64_t)a.l[li + 1] << (LB - sh);
    if (li + 2 < NL) v |= (uint64_t)a.l[li + 2] << (2 * LB - sh);
    w[j] = (uint32_t)v;
  }
}

template <int M>
HD void fp_const_limbs(Fp<M>& r, const uint32_t (&c)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c[i];
}

// wire (Montgomery R=2^768, canonical) -> internal (Montgomery R'=2^756, [0,2p))
template <int M>
HD void fp_from_wire(Fp<M>& r, const uint32_t w[24]) {
  Fp<M> raw, k;
  fp_unpack(raw, w);
  fp_const_limbs(k, FPC[M].k_in);
  fp_mul(r, raw, k);
}

// internal -> wire (canonical)
template <int M>
HD void fp_to_wire(uint32_t w[24], const Fp<M>& a) {
  Fp<M> k, t, c;
  fp_const_limbs(k, FPC[M].k_out);
  fp_mul(t, a, k);
  fp_canon(c, t);
  fp_pack(w, c);
}

// wire Montgomery -> plain integer (libff as_bigint, fp.tcc:227-238), packed 24 x u32
template <int M>
HD void fp_wire_to_integer(uint32_t w_out[24], const uint32_t w_in[24]) {
  Fp<M> raw, k, t, c;
  fp_unpack(raw, w_in);
  fp_const_limbs(k, FPC[M].k_std);
  fp_mul(t, raw, k);
  fp_canon(c, t);
  fp_pack(w_out, c);
}

}  // namespace mnt753
