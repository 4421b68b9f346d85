// MNT4753 / MNT6753 group law on the device: G1 over Fq, G2 over Fq2 (MNT4) / Fq3 (MNT6).
//
// Reference formulas (homogeneous projective, identity (0:1:0)):
//   depends/libff/libff/algebra/curves/mnt753/mnt4753/mnt4753_g1.cpp:134-207 (operator+),
//   :265-313 (mixed_add), :315-346 (dbl); mnt4753_g2.cpp:31-34 (mul_by_a), mnt6753_g2.cpp:38-41;
//   extension fields depends/libff/libff/algebra/fields/fp2.tcc:79-90, fp3.tcc:83-96.
//
// Structure: a point operation is a *micro-program* (one field multiplication per step)
// executed by pt_vm().  Each lane carries its own program counter, so a wave can mix lanes that
// add, lanes that must double (P == Q) and idle lanes, while the kernel contains exactly ONE
// inlined instance of the 1458-MAD multiplier (~14 KB of code): the hot loop stays inside the
// 64 KB instruction cache instead of streaming ~160 KB of straight-line code per addition.
#pragma once
#include <type_traits>
#include <utility>
#include "fp753.hip.h"
#include "fp_inv.hip.h"

namespace mnt753 {

// ------------------------------------------------------------------------------------------
// Coordinate fields.  Every field class provides: E, DEG, mul, add, sub, neg, is_zero, zero, one.
// ------------------------------------------------------------------------------------------
template <int M>
struct FieldFp {
  using E = Fp<M>;
  static constexpr int DEG = 1;
  static constexpr int LANES = 1;
  static constexpr int MOD = M;
  static constexpr bool HAS_SQR = true;
  static HD void mul(E& r, const E& a, const E& b) { fp_mul(r, a, b); }
  static HD void sqr(E& r, const E& a) { fp_sqr(r, a); }
  // lazy arithmetic of the pairing levels (fp753.hip.h): signed limb-wise differences into a signed-product multiplier
  static constexpr bool HAS_LAZY = true;
  static HD void mul_s(E& r, const E& a, const E& b) { fp_mul_s(r, a, b); }
  static HD void sqr_s(E& r, const E& a) { fp_sqr_s(r, a); }
  static HD void mul_s_ip(E& b, E& a) { fp_mul_s_ip(b, a); }         // b <- a b in place, operands opaque in place (fp753.hip.h)
  static HD void sqr_s_keep(E& r, E& a) { fp_sqr_s_keep(r, a); }
  static HD void sub_raw(E& r, const E& a, const E& b) { fp_sub_raw(r, a, b); }
  static HD void addsub_raw(E& r, const E& a, const E& y, bool subtract) { fp_addsub_raw(r, a, y, subtract); }
  static HD void norm(E& r, const E& a) { fp_norm(r, a); }
  static HD bool raw_maybe_zero(const E& d) { return fp_raw_maybe_zero(d); }
  static HD void inv(E& r, const E& a) { fp_inv(r, a); }
  static HD void add(E& r, const E& a, const E& b) { fp_add(r, a, b); }
  static HD void sub(E& r, const E& a, const E& b) { fp_sub(r, a, b); }
  static HD void neg(E& r, const E& a) { fp_neg(r, a); }
  static HD void half(E& r, const E& a) { fp_half(r, a); }
  static HD bool is_zero(const E& a) { return fp_is_zero(a); }
  static HD void zero(E& r) { fp_zero(r); }
  static HD void one(E& r) { fp_one(r); }
  static HD Fp<M>& comp(E& a, int) { return a; }
  static HD const Fp<M>& comp(const E& a, int) { return a; }
};

template <int M>
struct Fp2E {
  Fp<M> c0, c1;
};
template <int M>
struct Fp3E {
  Fp<M> c0, c1, c2;
};

// Fq2 = Fq[u]/(u^2 - NR): Karatsuba, 3 base multiplications through ONE multiplier instance.
template <int M, unsigned NR>
struct FieldFp2 {
  using E = Fp2E<M>;
  static constexpr int DEG = 2;
  static constexpr int LANES = 1;
  static constexpr int MOD = M;
  static constexpr unsigned NONRES = NR;
  static HD void mul(E& r, const E& x, const E& y) {
    Fp<M> a, b, t, aA, bB, sx, sy;
    fp_add(sx, x.c0, x.c1);
    fp_add(sy, y.c0, y.c1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int k = 0; k < 3; ++k) {
      switch (k) {
        case 0: a = x.c0; b = y.c0; break;
        case 1: a = x.c1; b = y.c1; break;
        default: a = sx; b = sy; break;
      }
      fp_mul(t, a, b);
      switch (k) {
        case 0: aA = t; break;
        case 1: bB = t; break;
        default: break;
      }
    }
    // c0 = aA + NR*bB ; c1 = (a+b)(A+B) - aA - bB
    Fp<M> nb;
    fp_mul_small(nb, bB, NR);
    fp_sub(t, t, aA);
    fp_sub(r.c1, t, bB);
    fp_add(r.c0, aA, nb);
  }
  static HD void add(E& r, const E& a, const E& b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); }
  static HD void sub(E& r, const E& a, const E& b) { fp_sub(r.c0, a.c0, b.c0); fp_sub(r.c1, a.c1, b.c1); }
  static HD void neg(E& r, const E& a) { fp_neg(r.c0, a.c0); fp_neg(r.c1, a.c1); }
  static HD void half(E& r, const E& a) { fp_half(r.c0, a.c0); fp_half(r.c1, a.c1); }
  static HD bool is_zero(const E& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
  static HD void zero(E& r) { fp_zero(r.c0); fp_zero(r.c1); }
  static HD void one(E& r) { fp_one(r.c0); fp_zero(r.c1); }
  static HD Fp<M>& comp(E& a, int i) { return i == 0 ? a.c0 : a.c1; }
  static HD const Fp<M>& comp(const E& a, int i) { return i == 0 ? a.c0 : a.c1; }
};

// Fq3 = Fq[u]/(u^3 - NR): Karatsuba, 6 base multiplications through ONE multiplier instance.
template <int M, unsigned NR>
struct FieldFp3 {
  using E = Fp3E<M>;
  static constexpr int DEG = 3;
  static constexpr int LANES = 1;
  static constexpr int MOD = M;
  static constexpr unsigned NONRES = NR;
  static HD void mul(E& r, const E& x, const E& y) {
    Fp<M> a, b, t, aA, bB, cC, t_bc, t_ab, t_ac;
    Fp<M> x_bc, x_ab, x_ac, y_bc, y_ab, y_ac;
    fp_add(x_bc, x.c1, x.c2); fp_add(y_bc, y.c1, y.c2);
    fp_add(x_ab, x.c0, x.c1); fp_add(y_ab, y.c0, y.c1);
    fp_add(x_ac, x.c0, x.c2); fp_add(y_ac, y.c0, y.c2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int k = 0; k < 6; ++k) {
      switch (k) {
        case 0: a = x.c0; b = y.c0; break;
        case 1: a = x.c1; b = y.c1; break;
        case 2: a = x.c2; b = y.c2; break;
        case 3: a = x_bc; b = y_bc; break;
        case 4: a = x_ab; b = y_ab; break;
        default: a = x_ac; b = y_ac; break;
      }
      fp_mul(t, a, b);
      switch (k) {
        case 0: aA = t; break;
        case 1: bB = t; break;
        case 2: cC = t; break;
        case 3: t_bc = t; break;
        case 4: t_ab = t; break;
        default: t_ac = t; break;
      }
    }
    // c0 = aA + NR*((b+c)(B+C) - bB - cC)
    // c1 = (a+b)(A+B) - aA - bB + NR*cC
    // c2 = (a+c)(A+C) - aA + bB - cC
    Fp<M> u, v;
    fp_sub(u, t_bc, bB); fp_sub(u, u, cC); fp_mul_small(v, u, NR); fp_add(r.c0, aA, v);
    fp_sub(u, t_ab, aA); fp_sub(u, u, bB); fp_mul_small(v, cC, NR); fp_add(r.c1, u, v);
    fp_sub(u, t_ac, aA); fp_add(u, u, bB); fp_sub(r.c2, u, cC);
  }
  static HD void add(E& r, const E& a, const E& b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); fp_add(r.c2, a.c2, b.c2); }
  static HD void sub(E& r, const E& a, const E& b) { fp_sub(r.c0, a.c0, b.c0); fp_sub(r.c1, a.c1, b.c1); fp_sub(r.c2, a.c2, b.c2); }
  static HD void neg(E& r, const E& a) { fp_neg(r.c0, a.c0); fp_neg(r.c1, a.c1); fp_neg(r.c2, a.c2); }
  static HD void half(E& r, const E& a) { fp_half(r.c0, a.c0); fp_half(r.c1, a.c1); fp_half(r.c2, a.c2); }
  static HD bool is_zero(const E& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1) && fp_is_zero(a.c2); }
  static HD void zero(E& r) { fp_zero(r.c0); fp_zero(r.c1); fp_zero(r.c2); }
  static HD void one(E& r) { fp_one(r.c0); fp_zero(r.c1); fp_zero(r.c2); }
  static HD Fp<M>& comp(E& a, int i) { return i == 0 ? a.c0 : (i == 1 ? a.c1 : a.c2); }
  static HD const Fp<M>& comp(const E& a, int i) { return i == 0 ? a.c0 : (i == 1 ? a.c1 : a.c2); }
};

// ------------------------------------------------------------------------------------------
// Lane-split Fq2: TWO adjacent lanes (2j, 2j+1) hold one element, lane parity = component index.  Per-lane state is
// that of a base-field element, so a G2 point operation needs the registers of a G1 one (the one-lane Fq2 VM keeps
// ~950 dwords live and spills 1.7 KB per lane to scratch).  A product is
//      even lane:  c0 = x0*y0 + (NR*x1)*y1        odd lane:  c1 = x1*y0 + x0*y1
// i.e. ONE fp_mul2 per lane; the partner's operands arrive through ds_bpermute (the LDS crossbar, no LDS memory).
// Both lanes of a pair always follow the same control flow (same sorted entries, same program counter).
// Additions, subtractions and negations are component-wise and need no exchange.
// ------------------------------------------------------------------------------------------
HD uint32_t pair_swap_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (a DPP quad_perm [1,0,3,2] move was 4 % faster in the accumulate kernel and made k_bucket_reduce<Mnt4G2S> fault or miscompute at
  // 2^16..2^19 points -- DESIGN.md 4.2, finding 3; the exchange is ds_bpermute everywhere, the build switch left the source in round 5)
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ 1u) << 2), (int)v);
#else
  return v;   // host builds only need this to compile
#endif
}
HD bool lane_is_odd() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (threadIdx.x & 1u) != 0;
#else
  return false;
#endif
}
template <int M, unsigned NR>
struct FieldFp2S {
  using E = Fp<M>;
  static constexpr int DEG = 2;      // components per element in memory (same layout as FieldFp2)
  static constexpr int LANES = 2;    // lanes that share one element
  static constexpr int MOD = M;
  static HD void mul(E& r, const E& x, const E& y) {
    E xo, b1, b2;
    const bool odd = lane_is_odd();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      xo.l[i] = pair_swap_u32(x.l[i]);
      const uint32_t yo = pair_swap_u32(y.l[i]);
      b1.l[i] = odd ? yo : y.l[i];
      b2.l[i] = odd ? y.l[i] : yo;
    }
    E nx;
    fp_mul_small(nx, xo, NR);
#pragma unroll
    for (int i = 0; i < NL; ++i) xo.l[i] = odd ? xo.l[i] : nx.l[i];
    fp_mul2(r, x, b1, xo, b2);
  }
  // (x0 - x1 u) / (x0^2 - NR x1^2)   (fp2.tcc:129-142): each lane squares its component, the pair shares the norm,
  // both lanes run the same base-field inversion
  static HD void inv(E& r, const E& x) {
    E sq, a0, a1, t, n, ni;
    const bool odd = lane_is_odd();
    fp_sqr(sq, x);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint32_t o = pair_swap_u32(sq.l[i]);
      a0.l[i] = odd ? o : sq.l[i];
      a1.l[i] = odd ? sq.l[i] : o;
    }
    fp_mul_small(t, a1, NR);
    fp_sub(n, a0, t);
    fp_inv(ni, n);
    fp_mul(t, x, ni);
    if (odd) fp_neg(r, t); else r = t;
  }
  static HD void add(E& r, const E& a, const E& b) { fp_add(r, a, b); }
  static HD void sub(E& r, const E& a, const E& b) { fp_sub(r, a, b); }
  static HD void neg(E& r, const E& a) { fp_neg(r, a); }
  static HD void half(E& r, const E& a) { fp_half(r, a); }   // component-wise: halving needs no exchange
  static HD bool is_zero(const E& a) {
    const uint32_t z = fp_is_zero(a) ? 1u : 0u;
    return (z & pair_swap_u32(z)) != 0;
  }
  static HD void zero(E& r) { fp_zero(r); }
  static HD void one(E& r) {
    fp_one(r);
    if (lane_is_odd()) fp_zero(r);
  }
};

// Lane-split Fq3: THREE adjacent lanes (3g, 3g+1, 3g+2 of a wave; lane 63 idles) hold one element, component =
// lane % 3.  Component k of a product is   x_k*y_0 + f1*x_{k+1}*y_2 + f2*x_{k+2}*y_1   (indices mod 3; f1 = NR unless
// k = 2, f2 = NR only for k = 0): ONE fp_mul3 per lane.  Operands of the other two lanes are fetched with
// ds_bpermute (the wave's LDS crossbar; no LDS memory is used).
HD uint32_t lane_fetch_u32(uint32_t v, int src_lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
#else
  (void)src_lane;
  return v;
#endif
}
HD int wave_lane() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(threadIdx.x & 63u);
#else
  return 0;
#endif
}
template <int M, unsigned NR>
struct FieldFp3S {
  using E = Fp<M>;
  static constexpr int DEG = 3;
  static constexpr int LANES = 3;
  static constexpr int MOD = M;
  static HD void mul(E& r, const E& x, const E& y) {
    const int lane = wave_lane(), k = lane % 3, g = lane - k;
    const int l1 = g + (k + 1) % 3, l2 = g + (k + 2) % 3;
    E x1, x2, y0, y1, y2;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      x1.l[i] = lane_fetch_u32(x.l[i], l1);
      x2.l[i] = lane_fetch_u32(x.l[i], l2);
      y0.l[i] = lane_fetch_u32(y.l[i], g);
      y1.l[i] = lane_fetch_u32(y.l[i], g + 1);
      y2.l[i] = lane_fetch_u32(y.l[i], g + 2);
    }
    E n1, n2;
    fp_mul_small(n1, x1, NR);
    fp_mul_small(n2, x2, NR);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      x1.l[i] = (k != 2) ? n1.l[i] : x1.l[i];
      x2.l[i] = (k == 0) ? n2.l[i] : x2.l[i];
    }
    fp_mul3(r, x, y0, x1, y2, x2, y1);
  }
  // fp3.tcc:126-143: c = (x0^2 - NR x1 x2, NR x2^2 - x0 x1, x1^2 - x0 x2), t = x0 c0 + NR (x2 c1 + x1 c2) = the
  // constant coefficient of x * c, result c / t.  Lane k forms c_k as one fused two-product step.
  static HD void inv(E& r, const E& x) {
    const int lane = wave_lane(), k = lane % 3, g = lane - k;
    E x0, x1, x2, a1, b1, a2, b2, t, c, p, ti;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      x0.l[i] = lane_fetch_u32(x.l[i], g);
      x1.l[i] = lane_fetch_u32(x.l[i], g + 1);
      x2.l[i] = lane_fetch_u32(x.l[i], g + 2);
    }
    E nx1, nx2, neg0, negn1;
    fp_mul_small(nx1, x1, NR);
    fp_mul_small(nx2, x2, NR);
    fp_neg(neg0, x0);
    fp_neg(negn1, nx1);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      a1.l[i] = k == 0 ? x0.l[i] : (k == 1 ? nx2.l[i] : x1.l[i]);
      b1.l[i] = k == 0 ? x0.l[i] : (k == 1 ? x2.l[i] : x1.l[i]);
      a2.l[i] = k == 0 ? negn1.l[i] : neg0.l[i];
      b2.l[i] = k == 1 ? x1.l[i] : x2.l[i];
    }
    fp_mul2(c, a1, b1, a2, b2);
    mul(p, x, c);                       // lane g holds t, the other two hold 0
#pragma unroll
    for (int i = 0; i < NL; ++i) t.l[i] = lane_fetch_u32(p.l[i], g);
    fp_inv(ti, t);
    fp_mul(r, c, ti);
  }
  static HD void add(E& r, const E& a, const E& b) { fp_add(r, a, b); }
  static HD void sub(E& r, const E& a, const E& b) { fp_sub(r, a, b); }
  static HD void neg(E& r, const E& a) { fp_neg(r, a); }
  static HD void half(E& r, const E& a) { fp_half(r, a); }   // component-wise: halving needs no exchange
  static HD bool is_zero(const E& a) {
    const int lane = wave_lane(), g = lane - lane % 3;
    const uint32_t z = fp_is_zero(a) ? 1u : 0u;
    return (lane_fetch_u32(z, g) & lane_fetch_u32(z, g + 1) & lane_fetch_u32(z, g + 2)) != 0;
  }
  static HD void zero(E& r) { fp_zero(r); }
  static HD void one(E& r) {
    fp_one(r);
    if (wave_lane() % 3 != 0) fp_zero(r);
  }
};

// ------------------------------------------------------------------------------------------
// Curve configurations: coordinate field F, scalar-field modulus FR, and mul_by_a.
// ------------------------------------------------------------------------------------------
struct Mnt4G1 {  // y^2 = x^3 + 2x + b over Fq = B          (mnt4753_init.cpp:119)
  using F = FieldFp<MOD_B>;
  static constexpr int FR = MOD_A;
  static HD void mul_by_a(F::E& r, const F::E& x) { fp_add(r, x, x); }
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }   // the curve coefficient a as a field element
};
struct Mnt6G1 {  // y^2 = x^3 + 11x + b over Fq = A         (mnt6753_init.cpp:130)
  using F = FieldFp<MOD_A>;
  static constexpr int FR = MOD_B;
  static HD void mul_by_a(F::E& r, const F::E& x) { fp_mul_small(r, x, 11u); }
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }
};
struct Mnt4G2 {  // twist over Fq2, a' = (2*13, 0): mul_by_a(c0,c1) = (26 c0, 26 c1)   (mnt4753_g2.cpp:31-34)
  using F = FieldFp2<MOD_B, 13u>;
  static constexpr int FR = MOD_A;
  static HD void mul_by_a(F::E& r, const F::E& x) { fp_mul_small(r.c0, x.c0, 26u); fp_mul_small(r.c1, x.c1, 26u); }
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }   // a' = (26, 0)
};
struct Mnt6G2 {  // twist over Fq3, a' = (0,0,11): mul_by_a(c0,c1,c2) = (121 c1, 121 c2, 11 c0)  (mnt6753_g2.cpp:38-41)
  using F = FieldFp3<MOD_A, 11u>;
  static constexpr int FR = MOD_B;
  static HD void mul_by_a(F::E& r, const F::E& x) {
    F::E t;
    fp_mul_small(t.c0, x.c1, 121u);
    fp_mul_small(t.c1, x.c2, 121u);
    fp_mul_small(t.c2, x.c0, 11u);
    r = t;
  }
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }   // a' = (0, 0, 11)
};

// lane-split counterpart of Mnt4G2 (same memory layout, two lanes per point)
struct Mnt4G2S {
  using F = FieldFp2S<MOD_B, 13u>;
  static constexpr int FR = MOD_A;
  static HD void mul_by_a(F::E& r, const F::E& x) { fp_mul_small(r, x, 26u); }   // (26 c0, 26 c1), component-wise
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }
};
// lane-split counterpart of Mnt6G2 (three lanes per point): mul_by_a(c0,c1,c2) = (121 c1, 121 c2, 11 c0)
struct Mnt6G2S {
  using F = FieldFp3S<MOD_A, 11u>;
  static constexpr int FR = MOD_B;
  static HD void mul_by_a(F::E& r, const F::E& x) {
    const int lane = wave_lane(), k = lane % 3, g = lane - k;
    F::E nx;
#pragma unroll
    for (int i = 0; i < NL; ++i) nx.l[i] = lane_fetch_u32(x.l[i], g + (k + 1) % 3);
    fp_mul_small(r, nx, k == 2 ? 11u : 121u);
  }
  static HD void coeff_a(F::E& r) { F::E o; F::one(o); mul_by_a(r, o); }
};
// C -> its lane-split configuration (void: none, the group runs one lane per point)
template <class C> struct SplitOf { using type = void; };
template <> struct SplitOf<Mnt4G2> { using type = Mnt4G2S; };
template <> struct SplitOf<Mnt6G2> { using type = Mnt6G2S; };

template <class F, class = void> struct has_sqr : std::false_type {};
template <class F> struct has_sqr<F, std::enable_if_t<F::HAS_SQR>> : std::true_type {};
// F::inv exists (base fields, lane-split fields); the one-lane extension fields go through the e_inv overloads of msm_kernels.hip.h
template <class F, class = void> struct has_inv : std::false_type {};
template <class F> struct has_inv<F, std::void_t<decltype(F::inv(std::declval<typename F::E&>(), std::declval<const typename F::E&>()))>> : std::true_type {};
template <class F, class = void> struct has_lazy : std::false_type {};
template <class F> struct has_lazy<F, std::enable_if_t<F::HAS_LAZY>> : std::true_type {};

template <class C>
struct Proj {
  typename C::F::E X, Y, Z;
};
template <class C>
struct Aff {
  typename C::F::E x, y;
};

template <class C>
HD bool pt_is_zero(const Proj<C>& P) { return C::F::is_zero(P.Z); }

template <class C>
HD void pt_set_zero(Proj<C>& P) { C::F::zero(P.X); C::F::one(P.Y); C::F::zero(P.Z); }

// ------------------------------------------------------------------------------------------
// The point-operation VM.  Program counters:
//    0..10  mixed/projective addition tail (P += Q, Q with Z2 == 1 for pc 0,1)
//   16..26  doubling  P = 2P
//   32..36  projective addition prologue (then falls into pc 2)
//   PC_END  idle
// ------------------------------------------------------------------------------------------
constexpr int PC_MADD = 0;
constexpr int PC_DBL = 16;
constexpr int PC_ADD = 32;
constexpr int PC_END = 63;

// P, Q: P is updated in place.  Preconditions: for PC_MADD / PC_ADD both P and Q are non-zero
// (the callers handle the identity); PC_DBL needs P non-zero.  Q.Z is only read by PC_ADD.
template <class C, bool WITH_ADD>
HD void pt_vm(Proj<C>& P, const Proj<C>& Q, int pc) {
  using F = typename C::F;
  using E = typename F::E;
  E u, v, t3, t4, t5, a, b, r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
  while (pc != PC_END) {
    switch (pc) {
      case 0: a = P.Z; b = Q.X; break;
      case 1: a = P.Z; b = Q.Y; break;
      case 2: a = u; b = u; break;
      case 3: a = v; b = v; break;
      case 4: a = v; b = t4; break;
      case 5: a = t4; b = P.X; break;
      case 6: a = t3; b = P.Z; break;
      case 7: a = v; b = t3; break;
      case 8: a = u; b = t4; break;
      case 9: a = t5; b = P.Y; break;
      case 10: a = t5; b = P.Z; break;
      case 16: a = P.X; b = P.X; break;
      case 17: a = P.Z; b = P.Z; break;
      case 18: a = P.Y; b = P.Z; break;
      case 19: a = v; b = v; break;
      case 20: a = v; b = t4; break;
      case 21: a = P.Y; b = v; break;
      case 22: a = t4; b = t4; break;
      case 23: F::add(a, P.X, t4); b = a; break;
      case 24: a = u; b = u; break;
      case 25: a = t3; b = v; break;
      case 26: F::sub(a, t4, t3); b = u; break;
      // (flat on purpose: a switch nested in `default:` was miscompiled for a lane-divergent pc by hipcc 7.2)
      case 32: if (WITH_ADD) { a = P.X; b = Q.Z; } break;
      case 33: if (WITH_ADD) { a = P.Y; b = Q.Z; } break;
      case 34: if (WITH_ADD) { a = Q.X; b = P.Z; } break;
      case 35: if (WITH_ADD) { a = Q.Y; b = P.Z; } break;
      case 36: if (WITH_ADD) { a = P.Z; b = Q.Z; } break;
      default: break;
    }
    if constexpr (has_sqr<F>::value) {
      // squaring steps (u^2, v^2 of an addition; six steps of a doubling) take the cheaper squaring; in the accumulate
      // kernel all lanes of a wave sit at the same step, so the two multipliers are not run back to back
      const bool sq = pc == 2 || pc == 3 || pc == 16 || pc == 17 || pc == 19 || pc == 22 || pc == 23 || pc == 24;
      if (sq) F::sqr(r, a); else F::mul(r, a, b);
    } else {
      F::mul(r, a, b);
    }
    switch (pc) {
      case 0: F::sub(v, r, P.X); pc = 1; break;
      case 1:
        F::sub(u, r, P.Y);
        pc = (F::is_zero(u) && F::is_zero(v)) ? PC_DBL : 2;
        break;
      case 2: t3 = r; pc = 3; break;
      case 3: t4 = r; pc = 4; break;
      case 4: t5 = r; pc = 5; break;
      case 5: t4 = r; pc = 6; break;
      case 6:
        F::sub(r, r, t5); F::sub(r, r, t4); F::sub(t3, r, t4);  // A = uu*Z - vvv - 2R
        F::sub(t4, t4, t3);                                      // R - A
        pc = 7;
        break;
      case 7: P.X = r; pc = 8; break;
      case 8: t4 = r; pc = 9; break;
      case 9: F::sub(P.Y, t4, r); pc = 10; break;
      case 10: P.Z = r; pc = PC_END; break;
      case 16: t3 = r; pc = 17; break;                           // XX
      case 17: {                                                 // w = a*ZZ + 3*XX
        E az;
        C::mul_by_a(az, r);
        F::add(u, t3, t3); F::add(u, u, t3); F::add(u, u, az);
        pc = 18;
      } break;
      case 18: F::add(v, r, r); pc = 19; break;                  // s = 2*Y1*Z1
      case 19: t4 = r; pc = 20; break;                           // ss
      case 20: P.Z = r; pc = 21; break;                          // Z3 = sss
      case 21: t4 = r; pc = 22; break;                           // R = Y1*s
      case 22: t5 = r; pc = 23; break;                           // RR
      case 23: F::sub(r, r, t3); F::sub(t4, r, t5); pc = 24; break;   // B = (X1+R)^2 - XX - RR
      case 24: F::sub(r, r, t4); F::sub(t3, r, t4); pc = 25; break;   // h = w^2 - 2B
      case 25: P.X = r; pc = 26; break;                          // X3 = h*s
      case 26: F::sub(r, r, t5); F::sub(P.Y, r, t5); pc = PC_END; break;  // Y3 = w*(B-h) - 2RR
      case 32: P.X = r; pc = 33; break;                          // X1Z2
      case 33: P.Y = r; pc = 34; break;                          // Y1Z2
      case 34: F::sub(v, r, P.X); pc = 35; break;                // v = X2Z1 - X1Z2
      case 35:
        F::sub(u, r, P.Y);                                       // u = Y2Z1 - Y1Z2
        if (F::is_zero(u) && F::is_zero(v)) { P.X = Q.X; P.Y = Q.Y; P.Z = Q.Z; pc = PC_DBL; }
        else pc = 36;
        break;
      case 36: P.Z = r; pc = 2; break;                           // Z1Z2, continue with the shared tail
      default: pc = PC_END; break;
    }
  }
}

// ---- the doubling chain of the table: modified Jacobian coordinates (round 5) ------------------------------------------------------
// Row w of a point is 2^c times row w - 1: c doublings, (W - 1) c = 741 per point at c = 19 -- the whole cost of the table (8.5 G
// products per 2^20 G1 points through the VM's projective doubling: eleven products, thirteen carried additions and the VM's
// operand routing per doubling; 6.1 of the 9.0 s of a parameter load in round 4).  A chain of doublings wants the coordinates whose
// doubling is cheapest, not the reference's: (X, Y, Z, W) with x = X / Z^2, y = Y / Z^3, W = a Z^4 (Cohen, Miyaji, Ono 1998), taken
// up to the scaling (X, Y, Z, W) ~ (X / 4, Y / 8, Z / 2, W / 16), which removes the factors 4 and 8 of the textbook formulas:
//     XX = X^2, YY = Y^2, S = X YY, Y4 = YY^2, H = (3 XX + W) / 2,
//     X' = H^2 - 2 S,   Y' = H (S - X') - Y4,   Z' = Y Z,   W' = Y4 W                               4 products + 4 squarings,
// one halving (fp_half) and five additions.  Only the AFFINE rows leave the kernel (x = X / Z^2, y = Y / Z^3 with one inversion per
// point over all its windows), and an affine point has one representation: the table holds the same group elements as before.
// Base fields: straight-line, the dedicated squarer, and the additions limb-wise without carries into the signed multiplier
// (fp753.hip.h, "lazy arithmetic"; tools/host_jac_check.cpp runs chains of it on the CPU against the host field, limb ranges asserted) -- ranges, with p / R' = 0.1106 and inputs X, Y, Z, W in [0, 2p) on normalised limbs:
//     XX, YY < 1.45p;  S < 1.32p;  Y4 < 1.24p;  3 XX + W < 6.35p on limbs < 2^30  ->  H < 3.68p, normalised limbs;
//     H^2 < 2.51p;  H^2 - 2 S in (-2.64p, 2.51p), limbs in (-2^29, 2^28)  ->  X' = fp_norm(..) in [0.49p, 1.51p);
//     S - X' in (-1.51p, 0.83p), |limbs| < 2^28;  H (S - X') in (-0.62p, 1.62p);  .. - Y4 in (-1.86p, 1.62p)  ->  Y' = fp_norm(..);
//     Z' < 1.45p;  W' < 1.28p.
// Every other field runs the same formulas through ONE instance of its multiplier in a step loop (eight fused lane-split products
// written out would be 160 KB of code) with its own carried additions.  G2 runs on its lane-split configuration: two / three lanes
// per point, as in the MSM's point kernels (the one-lane Karatsuba form spills kilobytes per lane).
template <class F>
struct Jac {
  typename F::E X, Y, Z, W;
};
template <class C>
HD void jac_dbl(Jac<typename C::F>& P) {
  using F = typename C::F;
  using E = typename F::E;
  if constexpr (has_lazy<F>::value) {
    constexpr int M = F::MOD;
    E XX, YY, S, Y4, H, t, u;
    fp_sqr(XX, P.X);
    fp_sqr(YY, P.Y);
    fp_mul(S, P.X, YY);
    fp_sqr(Y4, YY);
#pragma unroll
    for (int i = 0; i < NL; ++i) t.l[i] = 3u * XX.l[i] + P.W.l[i];
    fp_half(H, t);
    fp_sqr(t, H);
#pragma unroll
    for (int i = 0; i < NL; ++i) u.l[i] = t.l[i] - 2u * S.l[i];
    fp_mul(t, P.Y, P.Z);          // Z' (before Y is overwritten; P.X is dead from here on)
    fp_norm(P.X, u);
    P.Z = t;
    fp_sub_raw(u, S, P.X);
    fp_mul_s(t, H, u);
    fp_sub_raw(u, t, Y4);
    fp_norm(P.Y, u);
    fp_mul(t, Y4, P.W);
    P.W = t;
    (void)M;
  } else {
    E XX, YY, S, Y4, H, a, b, r;   // (XX is dead after step 3 and keeps Y' from step 5 on)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int step = 0; step < 8; ++step) {
      switch (step) {
        case 0: a = P.X; b = P.X; break;
        case 1: a = P.Y; b = P.Y; break;
        case 2: a = P.X; b = YY; break;
        case 3: a = YY; b = YY; break;
        case 4: a = H; b = H; break;
        case 5: a = H; b = S; break;     // S holds S - X' by then
        case 6: a = P.Y; b = P.Z; break;
        default: a = Y4; b = P.W; break;
      }
      F::mul(r, a, b);
      switch (step) {
        case 0: XX = r; break;
        case 1: YY = r; break;
        case 2: S = r; break;
        case 3:
          Y4 = r;
          F::add(H, XX, XX); F::add(H, H, XX); F::add(H, H, P.W);
          F::half(H, H);
          break;
        case 4:
          F::sub(r, r, S); F::sub(P.X, r, S);
          F::sub(S, S, P.X);
          break;
        case 5: F::sub(XX, r, Y4); break;    // Y', kept aside: step 6 still reads the old Y
        case 6: P.Z = r; P.Y = XX; break;
        default: P.W = r; break;
      }
    }
  }
}

}  // namespace mnt753
