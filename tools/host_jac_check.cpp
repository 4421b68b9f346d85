// Development / CPU-suite harness: the doubling of the window table (jac_dbl, curve753.hip.h: modified Jacobian coordinates,
// __host__ __device__) on the CPU against (a) the reference's golden vectors -- 2P of libff's dbl(), mnt4753_g1.cpp:315-346,
// mnt4753_g2.cpp:331-362, mnt6753_g2.cpp:337-368 -- and (b) chains of 45 doublings against the host field's projective doubling
// (host_field.hpp, u64 limbs, no code shared with the device layer).  Base fields run the lazy straight-line form, whose stated
// ranges are asserted after every doubling (limbs normalised, values below 2p); the one-lane extension fields run the generic
// step-loop form the lane-split G2 kernels instantiate with their own multiplier.
//   g++ -O1 -std=c++17 tools/host_jac_check.cpp -o build/host_jac_check && build/host_jac_check      (from the repo root)
#include <cstdio>
#include <cstring>
#include <vector>
#include "../snark-challenge-prover-reference_amd/csrc/curve753.hip.h"
#include "../snark-challenge-prover-reference_amd/csrc/host_field.hpp"
using namespace mnt753;

template <int M> bool below_2p_normalised(const Fp<M>& a) {
  for (int i = 0; i < NL; ++i) if (a.l[i] > LMASK) return false;
  int32_t bw = 0;
  for (int i = 0; i < NL; ++i) { int32_t t = (int32_t)a.l[i] - (int32_t)FPC[M].p2[i] + bw; bw = t >> LB; }
  return bw < 0;   // a - 2p < 0
}
template <class F> bool elem_ok(const typename F::E& a) {
  bool ok = true;
  for (int k = 0; k < F::DEG; ++k) ok &= below_2p_normalised(F::comp(a, k));
  return ok;
}
// (X, Y, Z) Jacobian -> affine wire words through the HOST field: x = X / Z^2, y = Y / Z^3
template <class C, class HC> void jac_to_affine(const Jac<typename C::F>& P, uint64_t* got) {
  using F = typename C::F;
  typename HC::F X, Y, Z;
  for (int k = 0; k < F::DEG; ++k) {
    uint64_t w[12];
    fp_to_wire((uint32_t*)w, F::comp(P.X, k)); X.comp(k) = HC::F::B::from_words(w);
    fp_to_wire((uint32_t*)w, F::comp(P.Y, k)); Y.comp(k) = HC::F::B::from_words(w);
    fp_to_wire((uint32_t*)w, F::comp(P.Z, k)); Z.comp(k) = HC::F::B::from_words(w);
  }
  typename HC::F zi = Z.inverse(), zi2 = zi * zi, x = X * zi2, y = Y * (zi2 * zi);
  for (int k = 0; k < F::DEG; ++k) { memcpy(got + 12 * k, x.comp(k).l, 96); memcpy(got + 12 * (F::DEG + k), y.comp(k).l, 96); }
}
template <class C, class HC> bool check(const char* name, const char* path) {
  using F = typename C::F;
  const int aw = 24 * F::DEG;
  FILE* f = fopen(path, "rb");
  if (!f) { printf("%s: no file %s\n", name, path); return false; }
  std::vector<uint64_t> rec(6 * aw + 12);
  bool all = true;
  int seen = 0;
  for (int i = 0; i < 8; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    const uint64_t *P = rec.data(), *dbl = P + 3 * aw + 12;
    bool pinf = true;
    for (int k = 0; k < 12 * F::DEG; ++k) if (P[12 * F::DEG + k]) pinf = false;
    if (pinf) continue;
    ++seen;
    Jac<F> J;
    for (int k = 0; k < F::DEG; ++k) {
      fp_from_wire(F::comp(J.X, k), (const uint32_t*)(P + 12 * k));
      fp_from_wire(F::comp(J.Y, k), (const uint32_t*)(P + 12 * (F::DEG + k)));
    }
    F::one(J.Z);
    C::coeff_a(J.W);
    host::HPoint<HC> ref;
    for (int k = 0; k < F::DEG; ++k) { ref.X.comp(k) = HC::F::B::from_words(P + 12 * k); ref.Y.comp(k) = HC::F::B::from_words(P + 12 * (F::DEG + k)); }
    ref.Z = HC::F::one();
    for (int n = 1; n <= 45; ++n) {
      jac_dbl<C>(J);
      ref = ref.dbl();
      uint64_t got[72], want[72];
      memset(got, 0, sizeof(got)); memset(want, 0, sizeof(want));
      jac_to_affine<C, HC>(J, got);
      typename HC::F rx, ry;
      ref.to_affine(rx, ry);
      for (int k = 0; k < F::DEG; ++k) { memcpy(want + 12 * k, rx.comp(k).l, 96); memcpy(want + 12 * (F::DEG + k), ry.comp(k).l, 96); }
      bool ok = memcmp(got, want, 8 * aw) == 0;
      if (n == 1) ok = ok && memcmp(got, dbl, 8 * aw) == 0;          // the reference's own 2P
      const bool ranges = elem_ok<F>(J.X) && elem_ok<F>(J.Y) && elem_ok<F>(J.Z) && elem_ok<F>(J.W);
      if (!ok || !ranges) { printf("%s rec %d after %d doublings: %s%s\n", name, i, n, ok ? "" : "MISMATCH ", ranges ? "" : "RANGE"); all = false; break; }
    }
  }
  fclose(f);
  printf("%s: %d points x 45 doublings: %s\n", name, seen, all && seen ? "OK" : "FAILED");
  return all && seen > 0;
}
int main() {
  bool ok = true;
  ok &= check<Mnt4G1, host::HMnt4G1>("jac mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  ok &= check<Mnt6G1, host::HMnt6G1>("jac mnt6 g1", "tests/golden/group_mnt6_g1.bin");
  ok &= check<Mnt4G2, host::HMnt4G2>("jac mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  ok &= check<Mnt6G2, host::HMnt6G2>("jac mnt6 g2", "tests/golden/group_mnt6_g2.bin");
  printf("%s\n", ok ? "ALL OK" : "FAILURES");
  return ok ? 0 : 1;
}
