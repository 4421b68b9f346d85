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
int main() { int b = run<0>() + run<1>(); printf(b ? "FAIL %d\n" : "fp_sqr / fp_mul2 / fp_mul3 / fp_inv OK\n", b); return b != 0; }
