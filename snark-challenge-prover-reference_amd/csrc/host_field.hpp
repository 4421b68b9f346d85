// Host-side 753-bit field / group arithmetic in WIRE form (12 x u64 limbs, Montgomery R = 2^768,
// canonical) -- the representation libff keeps in memory and libsnark/serialization.hpp writes.
//
// Used by the product for the O(1)-sized tails that do not belong on a GPU:
//   * Horner combination of the per-window MSM sums (a serial chain of ~750 group operations),
//   * summing per-GPU partial results, G1_add / G1_scale of run_prover (cuda_prover_piecewise.cu:83-85),
//   * to_affine_coordinates for groth16_output_write (mnt4753_g1.cpp:68-83, serialization.hpp:44-67),
//   * twiddle / coset constants of an evaluation domain (a handful of field ops per domain).
// It is NOT a fallback for the MSM / FFT: those entry points fail if the HIP device is missing.
#pragma once
#include <stdint.h>
#include <string.h>
#include "mnt753_constants.h"

namespace mnt753 {
namespace host {

typedef unsigned __int128 u128;

template <int M>
struct HFp {
  uint64_t l[12];
  static HFp zero() { HFp r; memset(r.l, 0, sizeof(r.l)); return r; }
  static HFp one() { HFp r; memcpy(r.l, FPC[M].one64, sizeof(r.l)); return r; }
  static HFp from_words(const uint64_t* w) { HFp r; memcpy(r.l, w, sizeof(r.l)); return r; }
  // small integer -> Montgomery form
  static HFp from_uint(uint64_t v) {
    HFp t = zero(); t.l[0] = v;
    HFp r2; memcpy(r2.l, FPC[M].r2_64, sizeof(r2.l));
    return t * r2;
  }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < 12; ++i) o |= l[i]; return o == 0; }
  bool operator==(const HFp& b) const { return memcmp(l, b.l, sizeof(l)) == 0; }
  bool operator!=(const HFp& b) const { return !(*this == b); }

  static bool geq_p(const uint64_t* a) {
    for (int i = 11; i >= 0; --i) {
      if (a[i] > FPC[M].p64[i]) return true;
      if (a[i] < FPC[M].p64[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    u128 bw = 0;
    for (int i = 0; i < 12; ++i) {
      u128 d = (u128)a[i] - FPC[M].p64[i] - bw;
      a[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }
  HFp operator+(const HFp& b) const {
    HFp r; u128 c = 0;
    for (int i = 0; i < 12; ++i) { c += (u128)l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  HFp operator-(const HFp& b) const {
    HFp r; u128 bw = 0;
    for (int i = 0; i < 12; ++i) {
      u128 d = (u128)l[i] - b.l[i] - bw;
      r.l[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    if (bw) {
      u128 c = 0;
      for (int i = 0; i < 12; ++i) { c += (u128)r.l[i] + FPC[M].p64[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
  }
  HFp operator-() const { return zero() - *this; }
  // CIOS Montgomery product, R = 2^768
  HFp operator*(const HFp& b) const {
    uint64_t t[14];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < 12; ++i) {
      u128 c = 0;
      for (int j = 0; j < 12; ++j) {
        c += (u128)l[j] * b.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[12];
      t[12] = (uint64_t)c;
      t[13] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * FPC[M].inv64;
      c = (u128)m * FPC[M].p64[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 12; ++j) {
        c += (u128)m * FPC[M].p64[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[12];
      t[11] = (uint64_t)c;
      t[12] = t[13] + (uint64_t)(c >> 64);
    }
    HFp r; memcpy(r.l, t, sizeof(r.l));
    if (t[12] || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  // k * x for a small constant (the non-residues 11 / 13, the curve coefficients 2 / 11 / 26 / 121): additions, not a conversion of k to
  // Montgomery form plus a product -- the Horner tail of a G2 MSM spent 40 % of its products there
  HFp mul_small(unsigned k) const {
    HFp r = zero();
    bool started = false;
    for (int i = 31; i >= 0; --i) {
      if (started) r = r + r;
      if ((k >> i) & 1u) { r = started ? r + *this : *this; started = true; }
    }
    return r;
  }
  HFp squared() const { return (*this) * (*this); }
  HFp dbl() const { return *this + *this; }
  // Montgomery form -> plain integer words (libff as_bigint)
  void to_integer(uint64_t out[12]) const {
    HFp o = zero(); o.l[0] = 1;
    HFp r = (*this) * o;
    memcpy(out, r.l, sizeof(r.l));
  }
  // x^e for a plain 768-bit exponent (little-endian words)
  HFp pow_words(const uint64_t* e, int nwords) const {
    HFp r = one();
    bool started = false;
    for (int i = nwords * 64 - 1; i >= 0; --i) {
      if (started) r = r.squared();
      if ((e[i >> 6] >> (i & 63)) & 1) { r = started ? r * (*this) : *this; started = true; }
    }
    return r;
  }
  HFp pow_u64(uint64_t e) const { return pow_words(&e, 1); }
  // inverse by Fermat: x^(p-2) -- ~1130 products, 0.3 ms; kept as the cross-check of inverse() (tools/host_inv_check.cpp)
  HFp inverse_fermat() const {
    uint64_t e[12];
    memcpy(e, FPC[M].p64, sizeof(e));
    e[0] -= 2;  // p is odd and p64[0] >= 2
    return pow_words(e, 12);
  }
  // inverse by the binary extended Euclid on the stored words (replaces Fp_model::invert, fp.tcc:641-685, which calls mpn_gcdext):
  // u = x, v = p, x1 u' = x1' x, ... ends with u or v = 1 after at most 2 * 753 halvings; ~20 us instead of Fermat's 0.3 ms -- the
  // three affine results of a proof are normalised on the host, which was 6 % of an MNT6753 prove.  The stored word is a R (Montgomery
  // form), its integer inverse is a^-1 R^-1: two products with R^2 give a^-1 R.  0 -> 0 like the Fermat form.  Not constant time
  // (the prover's outputs are public).  from_words() is a raw copy of caller / device words: a non-canonical stored word (l >= p) is
  // reduced first -- l = p or a multiple of it would otherwise pass is_zero(), drive u to 0 and never leave the halving loop -- and the
  // main loop is bounded (each round halves u or v at least once: 2 * 768 rounds at most) with the Fermat form behind it.
  HFp inverse() const {
    uint64_t u[12], v[12], x1[12], x2[12];
    memcpy(u, l, sizeof(u)); memcpy(v, FPC[M].p64, sizeof(v));
    for (int k = 0; k < 40000 && geq_p(u); ++k) sub_p(u);           // R / p < 2^15.2: canonical after at most ~37 055 subtractions (one for any word a kernel wrote)
    { uint64_t o = 0; for (int i = 0; i < 12; ++i) o |= u[i]; if (o == 0) return zero(); }
    memset(x1, 0, sizeof(x1)); x1[0] = 1; memset(x2, 0, sizeof(x2));
    auto is_one = [](const uint64_t* a) { uint64_t o = a[0] ^ 1u; for (int i = 1; i < 12; ++i) o |= a[i]; return o == 0; };
    auto shr1 = [](uint64_t* a) { for (int i = 0; i < 11; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[11] >>= 1; };
    auto add_p = [](uint64_t* a) { u128 c = 0; for (int i = 0; i < 12; ++i) { c += (u128)a[i] + FPC[M].p64[i]; a[i] = (uint64_t)c; c >>= 64; } };
    auto halve_mod = [&](uint64_t* a) { if (a[0] & 1u) add_p(a); shr1(a); };          // a < p odd: a + p < 2^754 fits the twelve words
    auto geq = [](const uint64_t* a, const uint64_t* b) { for (int i = 11; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i]; return true; };
    auto sub = [](uint64_t* a, const uint64_t* b) { u128 bw = 0; for (int i = 0; i < 12; ++i) { u128 d = (u128)a[i] - b[i] - bw; a[i] = (uint64_t)d; bw = (d >> 64) & 1; } return (bool)bw; };
    auto sub_mod = [&](uint64_t* a, const uint64_t* b) { if (sub(a, b)) add_p(a); };
    int rounds = 0;
    while (!is_one(u) && !is_one(v)) {
      if (++rounds > 2 * 768 + 2) return inverse_fermat();          // cannot happen for u in [1, p), p prime; never loop on a surprise
      while (!(u[0] & 1u)) { shr1(u); halve_mod(x1); }
      while (!(v[0] & 1u)) { shr1(v); halve_mod(x2); }
      if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); } else { sub(v, u); sub_mod(x2, x1); }
    }
    HFp y = from_words(is_one(u) ? x1 : x2), r2;
    memcpy(r2.l, FPC[M].r2_64, sizeof(r2.l));
    return (y * r2) * r2;
  }
};

// ---- extension fields (host) ---------------------------------------------------------------
template <int M>
struct HF1 {  // degree-1 wrapper so G1 and G2 share the group code
  typedef HFp<M> B;
  static constexpr int DEG = 1;
  B c0;
  static HF1 zero() { return HF1{B::zero()}; }
  static HF1 one() { return HF1{B::one()}; }
  bool is_zero() const { return c0.is_zero(); }
  bool operator==(const HF1& o) const { return c0 == o.c0; }
  HF1 operator+(const HF1& o) const { return HF1{c0 + o.c0}; }
  HF1 operator-(const HF1& o) const { return HF1{c0 - o.c0}; }
  HF1 operator-() const { return HF1{-c0}; }
  HF1 operator*(const HF1& o) const { return HF1{c0 * o.c0}; }
  HF1 inverse() const { return HF1{c0.inverse()}; }
  B& comp(int) { return c0; }
  const B& comp(int) const { return c0; }
};

template <int M, unsigned NR>
struct HF2 {
  typedef HFp<M> B;
  static constexpr int DEG = 2;
  B c0, c1;
  static B nr(const B& x) { return x.mul_small(NR); }
  static HF2 zero() { return HF2{B::zero(), B::zero()}; }
  static HF2 one() { return HF2{B::one(), B::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const HF2& o) const { return c0 == o.c0 && c1 == o.c1; }
  HF2 operator+(const HF2& o) const { return HF2{c0 + o.c0, c1 + o.c1}; }
  HF2 operator-(const HF2& o) const { return HF2{c0 - o.c0, c1 - o.c1}; }
  HF2 operator-() const { return HF2{-c0, -c1}; }
  HF2 operator*(const HF2& o) const {
    B aA = c0 * o.c0, bB = c1 * o.c1;
    return HF2{aA + nr(bB), (c0 + c1) * (o.c0 + o.c1) - aA - bB};
  }
  HF2 inverse() const {  // (a - b u) / (a^2 - NR b^2)
    B t = (c0 * c0 - nr(c1 * c1)).inverse();
    return HF2{c0 * t, -(c1 * t)};
  }
  B& comp(int i) { return i == 0 ? c0 : c1; }
  const B& comp(int i) const { return i == 0 ? c0 : c1; }
};

template <int M, unsigned NR>
struct HF3 {
  typedef HFp<M> B;
  static constexpr int DEG = 3;
  B c0, c1, c2;
  static B nr(const B& x) { return x.mul_small(NR); }
  static HF3 zero() { return HF3{B::zero(), B::zero(), B::zero()}; }
  static HF3 one() { return HF3{B::one(), B::zero(), B::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool operator==(const HF3& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
  HF3 operator+(const HF3& o) const { return HF3{c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
  HF3 operator-(const HF3& o) const { return HF3{c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
  HF3 operator-() const { return HF3{-c0, -c1, -c2}; }
  HF3 operator*(const HF3& o) const {
    B aA = c0 * o.c0, bB = c1 * o.c1, cC = c2 * o.c2;
    return HF3{aA + nr((c1 + c2) * (o.c1 + o.c2) - bB - cC),
               (c0 + c1) * (o.c0 + o.c1) - aA - bB + nr(cC),
               (c0 + c2) * (o.c0 + o.c2) - aA + bB - cC};
  }
  HF3 inverse() const {
    B t0 = c0 * c0, t1 = c1 * c1, t2 = c2 * c2, t3 = c0 * c1, t4 = c0 * c2, t5 = c1 * c2;
    B d0 = t0 - nr(t5), d1 = nr(t2) - t3, d2 = t1 - t4;
    B t6 = (c0 * d0 + nr(c2 * d1 + c1 * d2)).inverse();
    return HF3{t6 * d0, t6 * d1, t6 * d2};
  }
  B& comp(int i) { return i == 0 ? c0 : (i == 1 ? c1 : c2); }
  const B& comp(int i) const { return i == 0 ? c0 : (i == 1 ? c1 : c2); }
};

// ---- curves (host) ------------------------------------------------------------------------------
struct HMnt4G1 { typedef HF1<MOD_B> F; static constexpr int FR = MOD_A;
  static F mul_by_a(const F& x) { return x + x; } };
struct HMnt6G1 { typedef HF1<MOD_A> F; static constexpr int FR = MOD_B;
  static F mul_by_a(const F& x) { return F{x.c0.mul_small(11)}; } };
struct HMnt4G2 { typedef HF2<MOD_B, 13u> F; static constexpr int FR = MOD_A;
  static F mul_by_a(const F& x) { return F{x.c0.mul_small(26), x.c1.mul_small(26)}; } };
struct HMnt6G2 { typedef HF3<MOD_A, 11u> F; static constexpr int FR = MOD_B;
  static F mul_by_a(const F& x) { return F{x.c1.mul_small(121), x.c2.mul_small(121), x.c0.mul_small(11)}; } };

template <class C>
struct HPoint {
  typedef typename C::F F;
  F X, Y, Z;
  static HPoint zero() { return HPoint{F::zero(), F::one(), F::zero()}; }
  bool is_zero() const { return X.is_zero() && Z.is_zero(); }
  HPoint dbl() const {
    if (is_zero()) return *this;
    F XX = X * X, ZZ = Z * Z;
    F w = C::mul_by_a(ZZ) + (XX + XX + XX);
    F Y1Z1 = Y * Z, s = Y1Z1 + Y1Z1, ss = s * s, sss = s * ss;
    F R = Y * s, RR = R * R;
    F t = X + R;
    F B = t * t - XX - RR;
    F h = w * w - (B + B);
    return HPoint{h * s, w * (B - h) - (RR + RR), sss};
  }
  HPoint add(const HPoint& o) const {
    if (is_zero()) return o;
    if (o.is_zero()) return *this;
    F X1Z2 = X * o.Z, X2Z1 = Z * o.X, Y1Z2 = Y * o.Z, Y2Z1 = Z * o.Y;
    if (X1Z2 == X2Z1 && Y1Z2 == Y2Z1) return dbl();
    F Z1Z2 = Z * o.Z, u = Y2Z1 - Y1Z2, uu = u * u, v = X2Z1 - X1Z2, vv = v * v, vvv = v * vv;
    F R = vv * X1Z2, A = uu * Z1Z2 - (vvv + R + R);
    return HPoint{v * A, u * (R - A) - vvv * Y1Z2, vvv * Z1Z2};
  }
  // scalar given as plain integer words
  HPoint mul_words(const uint64_t* e, int nwords) const {
    HPoint r = zero();
    bool started = false;
    for (int i = nwords * 64 - 1; i >= 0; --i) {
      if (started) r = r.dbl();
      if ((e[i >> 6] >> (i & 63)) & 1) { r = r.add(*this); started = true; }
    }
    return r;
  }
  // affine x, y (identity -> all-zero, serialization.hpp:45-49)
  void to_affine(F& x, F& y) const {
    if (is_zero()) { x = F::zero(); y = F::zero(); return; }
    F zi = Z.inverse();
    x = X * zi;
    y = Y * zi;
  }
  // wire layout: X | Y | Z, each DEG x 12 u64
  static HPoint from_wire(const uint64_t* w) {
    HPoint p;
    for (int k = 0; k < F::DEG; ++k) {
      p.X.comp(k) = F::B::from_words(w + 12 * k);
      p.Y.comp(k) = F::B::from_words(w + 12 * (F::DEG + k));
      p.Z.comp(k) = F::B::from_words(w + 12 * (2 * F::DEG + k));
    }
    return p;
  }
  void to_wire(uint64_t* w) const {
    for (int k = 0; k < F::DEG; ++k) {
      memcpy(w + 12 * k, X.comp(k).l, 96);
      memcpy(w + 12 * (F::DEG + k), Y.comp(k).l, 96);
      memcpy(w + 12 * (2 * F::DEG + k), Z.comp(k).l, 96);
    }
  }
};

}  // namespace host
}  // namespace mnt753
