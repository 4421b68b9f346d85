// Development harness: runs the device point-operation VMs (curve753.hip.h is __host__ __device__) on the CPU
// against the reference's golden group vectors.  Build: hipcc -O1 -std=c++17 tools/host_vm_check.cpp -o build/host_vm_check
#include <cstdio>
#include <cstring>
#include <vector>
#include "experiments/vm_uniform.hip.h"
#include "../snark-challenge-prover-reference_amd/csrc/host_field.hpp"
using namespace mnt753;

template <class C> void load_aff(Aff<C>& q, const uint64_t* w) {
  using F = typename C::F;
  for (int k = 0; k < F::DEG; ++k) {
    fp_from_wire(F::comp(q.x, k), (const uint32_t*)(w + 12 * k));
    fp_from_wire(F::comp(q.y, k), (const uint32_t*)(w + 12 * (F::DEG + k)));
  }
}
template <class C, class HC> bool check(const char* name, const char* path) {
  using F = typename C::F;
  const int aw = 24 * F::DEG;
  FILE* f = fopen(path, "rb"); if (!f) { printf("%s: no file\n", name); return false; }
  std::vector<uint64_t> rec(6 * aw + 12);
  bool all = true;
  for (int i = 0; i < 8; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    const uint64_t *P = rec.data(), *Q = P + aw, *sum = Q + aw + 12, *dbl = sum + aw;
    bool pinf = true, qinf = true;
    for (int k = 0; k < 12 * F::DEG; ++k) { if (P[12 * F::DEG + k]) pinf = false; if (Q[12 * F::DEG + k]) qinf = false; }
    if (pinf || qinf) continue;
    Aff<C> p, q; load_aff<C>(p, P); load_aff<C>(q, Q);
    for (int mode = 0; mode < 2; ++mode) {   // 0: P + Q (madd, may switch to mdbl), 1: P + P
      XyzzAcc<C> A; A.X = p.x; A.Y = p.y; F::one(A.ZZ); F::one(A.ZZZ);
      const Aff<C>& qq = mode == 0 ? q : p;
      bool need = false;
      xyzz_madd_uniform<C>(A, qq.x, qq.y, true, need);
      if (need) xyzz_mdbl_uniform<C>(A, qq.x, qq.y, true);
      uint64_t got[72]; memset(got, 0, sizeof(got));
      if (!F::is_zero(A.ZZ)) {
        Proj<C> T;
        xyzz_to_proj_uniform<C>(T, A);
        uint64_t proj[108];
        for (int k = 0; k < F::DEG; ++k) {
          fp_to_wire((uint32_t*)(proj + 12 * k), F::comp(T.X, k));
          fp_to_wire((uint32_t*)(proj + 12 * (F::DEG + k)), F::comp(T.Y, k));
          fp_to_wire((uint32_t*)(proj + 12 * (2 * F::DEG + k)), F::comp(T.Z, k));
        }
        typename HC::F x, y;
        host::HPoint<HC>::from_wire(proj).to_affine(x, y);
        for (int k = 0; k < F::DEG; ++k) { memcpy(got + 12 * k, x.comp(k).l, 96); memcpy(got + 12 * (F::DEG + k), y.comp(k).l, 96); }
      }
      bool ok = memcmp(got, mode == 0 ? sum : dbl, 8 * aw) == 0;
      printf("%s rec %d %s: %s\n", name, i, mode == 0 ? "P+Q" : "P+P", ok ? "OK" : "MISMATCH");
      all &= ok;
    }
  }
  fclose(f);
  return all;
}
// projective VM: P + Q through PC_ADD (with the doubling switch), mixed add through PC_MADD, 2P through PC_DBL
template <class C, class HC> bool check_proj(const char* name, const char* path) {
  using F = typename C::F;
  const int aw = 24 * F::DEG;
  FILE* f = fopen(path, "rb"); if (!f) return false;
  std::vector<uint64_t> rec(6 * aw + 12);
  bool all = true;
  auto to_aff = [&](const Proj<C>& A, uint64_t* got) {
    memset(got, 0, 8 * 72);
    if (F::is_zero(A.Z)) return;
    uint64_t proj[108];
    for (int k = 0; k < F::DEG; ++k) {
      fp_to_wire((uint32_t*)(proj + 12 * k), F::comp(A.X, k));
      fp_to_wire((uint32_t*)(proj + 12 * (F::DEG + k)), F::comp(A.Y, k));
      fp_to_wire((uint32_t*)(proj + 12 * (2 * F::DEG + k)), F::comp(A.Z, k));
    }
    typename HC::F x, y;
    host::HPoint<HC>::from_wire(proj).to_affine(x, y);
    for (int k = 0; k < F::DEG; ++k) { memcpy(got + 12 * k, x.comp(k).l, 96); memcpy(got + 12 * (F::DEG + k), y.comp(k).l, 96); }
  };
  for (int i = 0; i < 8; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    const uint64_t *P = rec.data(), *Q = P + aw, *sum = Q + aw + 12, *dbl = sum + aw;
    bool pinf = true, qinf = true;
    for (int k = 0; k < 12 * F::DEG; ++k) { if (P[12 * F::DEG + k]) pinf = false; if (Q[12 * F::DEG + k]) qinf = false; }
    if (pinf || qinf) continue;
    Aff<C> p, q; load_aff<C>(p, P); load_aff<C>(q, Q);
    uint64_t got[72];
    // make P projective with Z != 1: scale by 2P's Z via one doubling-free trick: use (X*z, Y*z, z) with z = q.x
    Proj<C> A, B;
    F::mul(A.X, p.x, q.x); F::mul(A.Y, p.y, q.x); A.Z = q.x;
    F::mul(B.X, q.x, p.y); F::mul(B.Y, q.y, p.y); B.Z = p.y;
    { Proj<C> T = A; pt_vm<C, true>(T, B, PC_ADD); to_aff(T, got); bool ok = memcmp(got, sum, 8 * aw) == 0; printf("%s rec %d ADD: %s\n", name, i, ok ? "OK" : "MISMATCH"); all &= ok; }
    { Proj<C> T = A; Proj<C> A2 = A; pt_vm<C, true>(T, A2, PC_ADD); to_aff(T, got); bool ok = memcmp(got, dbl, 8 * aw) == 0; printf("%s rec %d ADD(P,P): %s\n", name, i, ok ? "OK" : "MISMATCH"); all &= ok; }
    { Proj<C> T = A; pt_vm<C, true>(T, T, PC_DBL); to_aff(T, got); bool ok = memcmp(got, dbl, 8 * aw) == 0; printf("%s rec %d DBL: %s\n", name, i, ok ? "OK" : "MISMATCH"); all &= ok; }
    { Proj<C> T = A; Proj<C> Qp; Qp.X = q.x; Qp.Y = q.y; F::one(Qp.Z); pt_vm<C, false>(T, Qp, PC_MADD); to_aff(T, got); bool ok = memcmp(got, sum, 8 * aw) == 0; printf("%s rec %d MADD: %s\n", name, i, ok ? "OK" : "MISMATCH"); all &= ok; }
  }
  fclose(f);
  return all;
}
// chained mixed additions: ((P0 + Q0) + P5) + Q5 + ... against host_field's projective sum
template <class C, class HC> bool chain(const char* name, const char* path) {
  using F = typename C::F;
  const int aw = 24 * F::DEG;
  FILE* f = fopen(path, "rb"); if (!f) return false;
  std::vector<uint64_t> rec(6 * aw + 12);
  std::vector<std::vector<uint64_t>> pts;
  for (int i = 0; i < 8; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    if (i == 0 || i >= 5) { pts.emplace_back(rec.begin(), rec.begin() + aw); pts.emplace_back(rec.begin() + aw, rec.begin() + 2 * aw); }
  }
  fclose(f);
  XyzzAcc<C> A; Aff<C> q;
  load_aff<C>(q, pts[0].data());
  A.X = q.x; A.Y = q.y; F::one(A.ZZ); F::one(A.ZZZ);
  host::HPoint<HC> ref = host::HPoint<HC>::zero();
  bool all = true;
  for (size_t i = 0; i < pts.size(); ++i) {
    uint64_t pw[108];
    {
      // host reference: affine -> projective wire
      memcpy(pw, pts[i].data(), 8 * aw);
      host::HPoint<HC> hp;
      for (int k = 0; k < F::DEG; ++k) { hp.X.comp(k) = HC::F::B::from_words(pw + 12 * k); hp.Y.comp(k) = HC::F::B::from_words(pw + 12 * (F::DEG + k)); }
      hp.Z = HC::F::one();
      ref = ref.add(hp);
    }
    if (i > 0) { load_aff<C>(q, pts[i].data()); bool need = false; xyzz_madd_uniform<C>(A, q.x, q.y, true, need); if (need) xyzz_mdbl_uniform<C>(A, q.x, q.y, true); }
    { // an inactive lane must leave the accumulator untouched
      XyzzAcc<C> B = A; bool nd = false; xyzz_madd_uniform<C>(B, q.x, q.y, false, nd);
      if (nd || memcmp(&B, &A, sizeof(A)) != 0) { printf("%s: inactive lane modified state\n", name); all = false; }
    }
    Proj<C> T;
    xyzz_to_proj_uniform<C>(T, A);
    uint64_t proj[108];
    for (int k = 0; k < F::DEG; ++k) {
      fp_to_wire((uint32_t*)(proj + 12 * k), F::comp(T.X, k));
      fp_to_wire((uint32_t*)(proj + 12 * (F::DEG + k)), F::comp(T.Y, k));
      fp_to_wire((uint32_t*)(proj + 12 * (2 * F::DEG + k)), F::comp(T.Z, k));
    }
    typename HC::F x, y, rx, ry;
    host::HPoint<HC>::from_wire(proj).to_affine(x, y);
    ref.to_affine(rx, ry);
    bool ok = x == rx && y == ry;
    printf("%s chain len %zu: %s\n", name, i + 1, ok ? "OK" : "MISMATCH");
    all &= ok;
  }
  return all;
}
// fp_mul_small against repeated lazy addition, on random and extreme inputs
template <int M> bool check_mul_small() {
  bool all = true;
  uint64_t st = 12345 + M;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
  for (int it = 0; it < 4000; ++it) {
    Fp<M> a;
    for (int i = 0; i < NL; ++i) a.l[i] = rnd() & LMASK;
    a.l[NL - 1] &= 0x1ffffff;                       // < 2^753
    if (it % 5 == 1) for (int i = 0; i < NL; ++i) a.l[i] = FPC[M].p2[i] - (i == 0 ? 1 : 0);   // 2p - 1
    if (it % 5 == 2) fp_zero(a);
    if (it % 5 == 3) for (int i = 0; i < NL; ++i) a.l[i] = FPC[M].p[i];                         // p
    Fp<M> chk; fp_reduce2p(chk, a.l); a = chk;      // into [0, 2p)
    unsigned k = (it % 7 == 0) ? 121u : (it % 7 == 1 ? 13u : (it % 7 == 2 ? 255u : 1u + rnd() % 200u));
    Fp<M> got, ref, c1, c2;
    fp_mul_small(got, a, k);
    fp_zero(ref);
    for (unsigned j = 0; j < k; ++j) fp_add(ref, ref, a);
    fp_canon(c1, got); fp_canon(c2, ref);
    bool in_range = true;   // got < 2p
    { Fp<M> tmp; uint32_t d[NL]; int32_t bw = 0; for (int i = 0; i < NL; ++i) { int32_t t = (int32_t)got.l[i] - (int32_t)FPC[M].p2[i] + bw; d[i] = t & LMASK; bw = t >> LB; } in_range = bw < 0; (void)tmp; (void)d; }
    bool limbs_ok = true; for (int i = 0; i < NL; ++i) limbs_ok &= got.l[i] <= LMASK;
    if (memcmp(c1.l, c2.l, sizeof(c1.l)) != 0 || !in_range || !limbs_ok) { printf("mul_small M=%d k=%u it=%d MISMATCH\n", M, k, it); all = false; }
  }
  printf("mul_small M=%d: %s\n", M, all ? "OK" : "MISMATCH");
  return all;
}
int main() {
  bool ok = true;
  ok &= check_mul_small<0>();
  ok &= check_mul_small<1>();
  ok &= check<Mnt4G1, host::HMnt4G1>("mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  ok &= check<Mnt6G1, host::HMnt6G1>("mnt6 g1", "tests/golden/group_mnt6_g1.bin");
  ok &= check<Mnt4G2, host::HMnt4G2>("mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  ok &= check<Mnt6G2, host::HMnt6G2>("mnt6 g2", "tests/golden/group_mnt6_g2.bin");
  ok &= check_proj<Mnt4G1, host::HMnt4G1>("proj mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  ok &= check_proj<Mnt6G1, host::HMnt6G1>("proj mnt6 g1", "tests/golden/group_mnt6_g1.bin");
  ok &= check_proj<Mnt4G2, host::HMnt4G2>("proj mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  ok &= check_proj<Mnt6G2, host::HMnt6G2>("proj mnt6 g2", "tests/golden/group_mnt6_g2.bin");
  ok &= chain<Mnt4G1, host::HMnt4G1>("mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  ok &= chain<Mnt4G2, host::HMnt4G2>("mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  ok &= chain<Mnt6G2, host::HMnt6G2>("mnt6 g2", "tests/golden/group_mnt6_g2.bin");
  printf("%s\n", ok ? "ALL OK" : "FAILURES");
  return ok ? 0 : 1;
}
