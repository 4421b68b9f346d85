// Modular inversion without exponentiation: Bernstein-Yang "safegcd" division steps, 28 at a time.
//
// Replaces libff's Fp_model::invert (fields/fp.tcc:641-685, mpn_gcdext) on the device.  The Fermat chain a^(p-2) costs
// ~950 Montgomery products (about 3 ms of wave time for ONE inversion on gfx950); this costs 78 batches of
//     28 division steps on the low words of (f, g)              -- 32-bit ALU only, builds a 2x2 transition matrix
//     (f, g) <- M (f, g) / 2^28 ,  (d, e) <- M (d, e) / 2^28 mod p -- 27 x 6 multiply-adds (v_mad_i64_i32)
// i.e. ~90 k simple instructions = ~55 product-equivalents.  Every lane runs the same instruction sequence (the division
// steps are the constant-time formulation: masks, no branches), so a wave inverts 64 different elements at full rate.
//
// Number of steps: the original divstep (delta starts at 1) needs at most floor((49 d + 57) / 17) steps for inputs below
// 2^d (Bernstein-Yang 2019, Theorem 11.2); d = 754 gives 2176 <= 78 * 28 = 2184.
// Measured and NOT adopted (round 3; the build switch left the source in round 5): the half-delta variant (delta starts at 1/2, as in libsecp256k1's
// modinv) run UNTIL g = 0 IN EVERY LANE OF THE WAVE -- once g is 0 a batch is the identity (transition matrix 2^28 I), so lanes
// that are done early are unharmed and correctness rests on no step bound.  Random 753-bit inputs need 1520 steps on average and
// 1536 as the maximum over the 64 lanes of a wave (simulation over 2560 inputs: at most 1544): 55-56 batches instead of 78.  Same
// registers, bit-identical results (131 GPU tests) -- and no gain: same box, alternating, G1 2^20 24.96 / 25.25 ms fixed against
// 25.14 / 25.05 early, G2 2^20 65.98 / 65.79 against 69.36 / 69.76 (profiles/r03/ab_inversion_early_exit.txt).  The inversion is
// the low-power phase of a pairing level (32-bit ALU work, no multiplies); shortening it shortens the time the chip spends below
// its power limit, and the multiplier phases around it pay that back in clock.
//
// Representation inside the routine: 27 limbs of 28 bits, limbs 0..25 in [0, 2^28), limb 26 signed -- the same limb width
// as fp753.hip.h, so a 28-step batch divides by exactly one limb.  d and e stay in (-2p, p), f and g in [-p, p].
#pragma once
#include "fp753.hip.h"

namespace mnt753 {

struct InvMat { int32_t u, v, q, r; };

// 28 division steps on the low limbs; returns the new eta = -(delta + 1/2) (delta > 0 <=> eta < 0) and the transition matrix t with
//   (f', g') = t (f, g) / 2^28
HD int32_t inv_divsteps28(int32_t eta, uint32_t f0, uint32_t g0, InvMat& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
  for (int i = 0; i < 28; ++i) {
    uint32_t c1 = (uint32_t)(eta >> 31);          // delta > 0
    const uint32_t c2 = 0u - (g & 1u);            // g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // -f, -u, -v when delta > 0
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;                                     // swap case: delta > 0 and g odd
    eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));   // eta = -delta: delta <- 1 - delta (swap) or 1 + delta
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return eta;
}

// (f, g) <- t (f, g) / 2^28  (exact: the low 28 bits vanish by construction)
HD void inv_update_fg(int32_t f[NL], int32_t g[NL], const InvMat& t) {
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
  int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= LB; cg >>= LB;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)((uint32_t)cf & LMASK); cf >>= LB;
    g[i - 1] = (int32_t)((uint32_t)cg & LMASK); cg >>= LB;
  }
  f[NL - 1] = (int32_t)cf;
  g[NL - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^28 mod p, keeping both in (-2p, p): a multiple of p is added that clears the low limb
template <int M>
HD void inv_update_de(int32_t d[NL], int32_t e[NL], const InvMat& t) {
  const uint32_t pinv = (0u - FPC[M].inv) & LMASK;        // p^-1 mod 2^28 (the table holds -p^-1)
  const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;   // sign masks
  int32_t md = (t.u & sd) + (t.v & se);                   // start from +p for every negative input
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
  int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  md -= (int32_t)((pinv * (uint32_t)cd + (uint32_t)md) & LMASK);
  me -= (int32_t)((pinv * (uint32_t)ce + (uint32_t)me) & LMASK);
  cd += (int64_t)FPC[M].p[0] * md;
  ce += (int64_t)FPC[M].p[0] * me;
  cd >>= LB; ce >>= LB;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)FPC[M].p[i] * md;
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)FPC[M].p[i] * me;
    d[i - 1] = (int32_t)((uint32_t)cd & LMASK); cd >>= LB;
    e[i - 1] = (int32_t)((uint32_t)ce & LMASK); ce >>= LB;
  }
  d[NL - 1] = (int32_t)cd;
  e[NL - 1] = (int32_t)ce;
}

// r = a^-1 as plain integers mod p: a canonical in [0, p) (limbs as in Fp), r canonical; a = 0 gives r = 0.
template <int M>
HD void fp_inv_integer(uint32_t r[NL], const uint32_t a[NL]) {
  int32_t d[NL], e[NL], f[NL], g[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) { d[i] = 0; e[i] = 0; f[i] = (int32_t)FPC[M].p[i]; g[i] = (int32_t)a[i]; }
  e[0] = 1;
  int32_t eta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int it = 0; it < 78; ++it) {
    InvMat t;
    eta = inv_divsteps28(eta, (uint32_t)f[0], (uint32_t)g[0], t);
    inv_update_de<M>(d, e, t);
    inv_update_fg(f, g, t);
  }
  // g = 0 and f = +-1 now (f = +-p when a = 0); d * a = f (mod p).  Negate d when f < 0, then bring it into [0, p).
  const int32_t sf = f[NL - 1] >> 31;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {            // d <- (d ^ sf) - sf, limb-wise with borrow
    int32_t v = (d[i] ^ sf) - sf + c;
    if (i < NL - 1) { c = v >> LB; v &= (int32_t)LMASK; }
    d[i] = v;
  }
  // limbs 0..25 of the negated value: (x ^ -1) - (-1) = -x per limb needs the carry chain above because inner limbs are
  // unsigned; after it d is again "inner limbs in [0, 2^28), top limb signed", value in (-2p, 2p)
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {     // add p while negative (at most twice)
    const int32_t neg = d[NL - 1] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int32_t v = d[i] + ((int32_t)FPC[M].p[i] & neg) + c;
      if (i < NL - 1) { c = v >> LB; v &= (int32_t)LMASK; }
      d[i] = v;
    }
  }
  {                                          // subtract p once if d >= p
    int32_t s[NL];
    c = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int32_t v = d[i] - (int32_t)FPC[M].p[i] + c;
      if (i < NL - 1) { c = v >> LB; v &= (int32_t)LMASK; }
      s[i] = v;
    }
    const int32_t ge = ~(s[NL - 1] >> 31);   // all ones when d - p >= 0
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (s[i] & ge) | (d[i] & ~ge);
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) r[i] = (uint32_t)d[i];
}

// Montgomery-form inverse in the device representation (radix R' = 2^756, values lazily in [0, 2p)):
// x = a R'  ->  r = a^-1 R'.  The integer inverse of x is a^-1 R'^-1; two products by R'^2 restore the radix:
// mul(mul(y, R'^2), R'^2) = y R'^2.  Zero maps to zero.
template <int M>
HD void fp_inv(Fp<M>& r, const Fp<M>& x) {
  Fp<M> c, y, r2, t;
  fp_canon(c, x);
  fp_inv_integer<M>(y.l, c.l);
  fp_const_limbs(r2, FPC[M].r2p);
  fp_mul(t, y, r2);
  fp_mul(r, t, r2);
}

}  // namespace mnt753
