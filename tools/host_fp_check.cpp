#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../snark-challenge-prover-reference_amd/csrc/fp_inv.hip.h"   // host build: g++ -O1 -std=c++17 tools/host_fp_check.cpp
using namespace mnt753;
static uint64_t st = 88172645463325252ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
template <int M> void rand_fp(Fp<M>& a) {   // random value in [0, 2p): random limbs then canonical-ish via mul by one
  Fp<M> t, one; for (int i = 0; i < NL; ++i) t.l[i] = (uint32_t)rnd() & LMASK; t.l[NL - 1] &= 0x1fff;  // < 2^741
  fp_one(one); fp_mul(a, t, one);
  if (rnd() & 1) { Fp<M> z; fp_zero(z); fp_sub(a, z, a); }   // exercise the upper half of [0, 2p)
}
template <int M> bool eq(const Fp<M>& a, const Fp<M>& b) { Fp<M> x, y; fp_canon(x, a); fp_canon(y, b); return memcmp(x.l, y.l, sizeof(x.l)) == 0; }
template <int M> bool in_range(const Fp<M>& a) {  // a < 2p
  int64_t bw = 0; for (int i = 0; i < NL; ++i) { int64_t t = (int64_t)a.l[i] - FPC[M].p2[i] + bw; bw = t >> LB; if (a.l[i] > LMASK) return false; } return bw < 0;
}
template <int M> int run() {
  int bad = 0;
  for (int it = 0; it < 2000; ++it) {
    Fp<M> a, b, c, d, e, f, r1, r2, t1, t2, t3;
    rand_fp(a); rand_fp(b); rand_fp(c); rand_fp(d); rand_fp(e); rand_fp(f);
    fp_mul(r1, a, a); fp_sqr(r2, a); if (!eq(r1, r2) || !in_range(r2)) { ++bad; if (bad < 4) printf("sqr mismatch M=%d\n", M); }
    fp_mul(t1, a, b); fp_mul(t2, c, d); fp_add(r1, t1, t2); fp_mul2(r2, a, b, c, d); if (!eq(r1, r2) || !in_range(r2)) { ++bad; if (bad < 4) printf("mul2 mismatch M=%d\n", M); }
    fp_mul(t3, e, f); fp_add(r1, r1, t3); fp_mul3(r2, a, b, c, d, e, f); if (!eq(r1, r2) || !in_range(r2)) { ++bad; if (bad < 4) printf("mul3 mismatch M=%d\n", M); }
  }
  // divstep inversion: x * x^-1 = 1 for random x in [0, 2p), plus 0 -> 0, 1 -> 1, p - 1 (= -1) -> itself
  Fp<M> one; fp_one(one);
  for (int it = 0; it < 300; ++it) {
    Fp<M> x, ix, prod;
    rand_fp(x);
    if (it == 0) x = one;
    if (it == 1) { Fp<M> z; fp_zero(z); fp_sub(x, z, one); }
    fp_inv(ix, x);
    fp_mul(prod, x, ix);
    if (!eq(prod, one) || !in_range(ix)) { ++bad; if (bad < 6) printf("inv mismatch M=%d it=%d\n", M, it); }
  }
  { Fp<M> z, iz; fp_zero(z); fp_inv(iz, z); if (!fp_is_zero(iz)) { ++bad; printf("inv(0) != 0, M=%d\n", M); } }
  return bad;
}
// The lazy arithmetic of the pairing levels (fp_mul_s, fp_sqr_s, fp_sub_raw, fp_addsub_raw, fp_norm) against the eager formulas
// on whole affine additions, chained over several "levels" so that normalised outputs feed the next round: same residues, outputs in
// [0, 2p) with 28-bit limbs, and the cheap zero test never misses a zero difference.
template <int M> bool norm_ok(const Fp<M>& a) {   // limbs in [0, 2^28) and p/2 - eps <= a < 3p/2 + eps (loosely: < 2p)
  for (int i = 0; i < NL; ++i) if (a.l[i] > LMASK) return false;
  return in_range(a);
}
template <int M> int run_lazy() {
  int bad = 0;
  for (int it = 0; it < 1500; ++it) {
    Fp<M> x1, y1, x2, y2, inv, pre;
    rand_fp(x1); rand_fp(y1); rand_fp(x2); rand_fp(y2); rand_fp(inv); rand_fp(pre);
    if (it % 7 == 0) { fp_zero(x1); }                       // extreme operands: 0 and values just below 2p
    if (it % 11 == 0) { Fp<M> z, o; fp_zero(z); fp_one(o); fp_sub(x2, z, o); }
    Fp<M> ex1 = x1, ey1 = y1, ex2 = x2, ey2 = y2, einv = inv, epre = pre;   // eager twin
    for (int level = 0; level < 4; ++level) {
      const bool flip = ((it + level) & 1) != 0;
      // eager (what k_pair_level computed before round 3)
      Fp<M> eden, enum_, t, el, ex3, enum2, ey3, ninv, npre;
      fp_sub(eden, ex2, ex1);
      if (flip) fp_add(enum_, ey2, ey1); else fp_sub(enum_, ey2, ey1);
      fp_mul(npre, einv, epre); fp_mul(ninv, einv, eden); fp_mul(el, enum_, npre);
      fp_sqr(t, el); fp_sub(t, t, ex1); fp_sub(ex3, t, ex2); fp_sub(enum2, ex1, ex3);
      fp_mul(t, el, enum2); if (flip) fp_add(ey3, t, ey1); else fp_sub(ey3, t, ey1);
      // lazy
      Fp<M> den, num, l, res, x3, num2, y3, linv, lpre;
      fp_sub_raw(den, x2, x1);
      if (fp_is_zero(eden) && !fp_raw_maybe_zero(den)) { ++bad; printf("zero test missed M=%d\n", M); }
      fp_addsub_raw(num, y2, y1, !flip);
      fp_mul_s(lpre, inv, pre); fp_mul_s(linv, inv, den); fp_mul_s(l, num, lpre);
      fp_sqr_s(res, l); fp_sub_raw(res, res, x1); fp_sub_raw(res, res, x2); fp_norm(x3, res); fp_sub_raw(num2, x1, x3);
      fp_mul_s(res, l, num2); fp_addsub_raw(res, res, y1, !flip); fp_norm(y3, res);
      Fp<M> linv_n, lpre_n;
      fp_norm(linv_n, linv); fp_norm(lpre_n, lpre);
      if (!eq(x3, ex3) || !eq(y3, ey3) || !eq(linv_n, ninv) || !eq(lpre_n, npre) || !norm_ok(x3) || !norm_ok(y3) || !norm_ok(linv_n)) {
        ++bad; if (bad < 6) printf("lazy slot mismatch M=%d it=%d level=%d\n", M, it, level);
      }
      // next level: the outputs meet a fresh point; the running inverse stays SIGNED (as in the kernel's backward sweep)
      x1 = x3; y1 = y3; ex1 = ex3; ey1 = ey3;
      rand_fp(x2); rand_fp(y2); ex2 = x2; ey2 = y2;
      inv = linv; einv = ninv; pre = lpre; epre = npre;
    }
  }
  return bad;
}
// The carry-free butterflies of the NTT (ntt_kernels.hip.h, k_ntt_group): two stages (the second on the raw outputs of the first) and
// ONE normalisation per element, against eager fp_mul / fp_add / fp_sub; inputs at the edges of what the kernel can hand a stage
// pair -- [0, 1.51p) after fp_norm, [0, p) after fp_unpack, below 1.44p behind an in_scale product --, twiddles anywhere in [0, 2p).
// Checked: the same values mod p, every raw limb below 2^30 in magnitude, the normalised outputs inside [0, 2p) with 28-bit limbs.
template <int M> bool limbs_below(const Fp<M>& a, int bits) {
  for (int i = 0; i < NL; ++i) { const int64_t v = (int32_t)a.l[i]; if (v >= ((int64_t)1 << bits) || v <= -((int64_t)1 << bits)) return false; }
  return true;
}
template <int M> int run_ntt_lazy() {
  int bad = 0;
  Fp<M> pm1, z, one; fp_zero(z); fp_one(one); fp_sub(pm1, z, one);   // p - 1 (in [0, 2p))
  for (int it = 0; it < 4000; ++it) {
    // four elements of one 4-point sub-transform, two twiddle stages
    Fp<M> x[4], e[4], w1, w2, w3;
    for (int i = 0; i < 4; ++i) { rand_fp(x[i]); if ((it & 7) == 1) x[i] = pm1; fp_norm(x[i], x[i]); e[i] = x[i]; }   // normalised: [0.49p, 1.51p)
    if ((it & 7) == 2) for (int i = 0; i < 4; ++i) { Fp<M> c; fp_canon(c, x[i]); x[i] = c; e[i] = c; }              // canonical: [0, p)
    rand_fp(w1); rand_fp(w2); rand_fp(w3);
    if ((it & 3) == 3) { Fp<M> two_p_minus; fp_sub(two_p_minus, z, one); fp_add(w1, two_p_minus, pm1); }              // a twiddle near the top of [0, 2p)
    // eager
    Fp<M> t, a0, a1, a2, a3;
    fp_mul(t, w1, e[1]); fp_add(a0, e[0], t); fp_sub(a1, e[0], t);
    fp_mul(t, w1, e[3]); fp_add(a2, e[2], t); fp_sub(a3, e[2], t);
    Fp<M> b0, b1, b2, b3;
    fp_mul(t, w2, a2); fp_add(b0, a0, t); fp_sub(b2, a0, t);
    fp_mul(t, w3, a3); fp_add(b1, a1, t); fp_sub(b3, a1, t);
    // carry-free: stage A raw, stage B raw on those, one normalisation
    Fp<M> r0, r1, r2, r3, s0, s1, s2, s3;
    fp_mul_s(t, w1, x[1]); fp_sub_raw(r1, x[0], t); fp_addsub_raw(r0, x[0], t, false);
    fp_mul_s(t, w1, x[3]); fp_sub_raw(r3, x[2], t); fp_addsub_raw(r2, x[2], t, false);
    if (!limbs_below(r0, 30) || !limbs_below(r1, 30) || !limbs_below(r2, 30) || !limbs_below(r3, 30)) { ++bad; if (bad < 6) printf("ntt lazy: stage A limb out of range M=%d\n", M); }
    fp_mul_s(t, w2, r2); fp_sub_raw(s2, r0, t); fp_addsub_raw(s0, r0, t, false);
    fp_mul_s(t, w3, r3); fp_sub_raw(s3, r1, t); fp_addsub_raw(s1, r1, t, false);
    if (!limbs_below(s0, 30) || !limbs_below(s1, 30) || !limbs_below(s2, 30) || !limbs_below(s3, 30)) { ++bad; if (bad < 6) printf("ntt lazy: stage B limb out of range M=%d\n", M); }
    fp_norm(s0, s0); fp_norm(s1, s1); fp_norm(s2, s2); fp_norm(s3, s3);
    if (!eq(s0, b0) || !eq(s1, b1) || !eq(s2, b2) || !eq(s3, b3) || !in_range(s0) || !in_range(s1) || !in_range(s2) || !in_range(s3)) {
      ++bad; if (bad < 6) printf("ntt lazy butterfly mismatch M=%d it=%d\n", M, it);
    }
  }
  return bad;
}
int main() { int b = run<0>() + run<1>() + run_lazy<0>() + run_lazy<1>() + run_ntt_lazy<0>() + run_ntt_lazy<1>(); printf(b ? "FAIL %d\n" : "fp_sqr / fp_mul2 / fp_mul3 / fp_inv / lazy arithmetic / carry-free NTT butterflies OK\n", b); return b != 0; }
