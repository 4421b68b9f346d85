// CPU-suite harness: the host field's inversion (host_field.hpp, binary extended Euclid on the stored words) against
//   (a) the reference's own Fp_model::invert (fp.tcc:641-685) through the libff-minted vectors tests/golden/field_{A,B}.bin (a, a^-1),
//       and Fp2_model / Fp3_model::inverse through tests/golden/extfield_mnt{4,6}.bin,
//   (b) Fermat's a^(p-2) -- the form it replaced -- on 4000 values per modulus: the golden inputs, 0, 1, 2, p - 1, p - 2, (p +- 1) / 2,
//       powers of two, and a multiplicative walk; a * a^-1 = 1 asserted for each.
// Prints the time of both forms.
//   g++ -O1 -std=c++17 tools/host_inv_check.cpp -o build/host_inv_check && build/host_inv_check      (from the repo root)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../snark-challenge-prover-reference_amd/csrc/curve753.hip.h"
#include "../snark-challenge-prover-reference_amd/csrc/host_field.hpp"
using namespace mnt753;
using clk = std::chrono::steady_clock;

static std::vector<uint64_t> slurp(const char* path) {
  std::vector<uint64_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  uint64_t buf[4096];
  size_t n;
  while ((n = fread(buf, 8, 4096, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}
template <int M> bool base(const char* name, const char* golden) {
  using H = host::HFp<M>;
  std::vector<uint64_t> g = slurp(golden);
  if (g.size() != 24 * 8 * 12) { printf("%s: cannot read %s\n", name, golden); return false; }
  bool ok = true;
  std::vector<H> xs;
  for (int r = 0; r < 24; ++r) {            // record: a, b, a*b, a+b, a-b, a^-1, -a, as_bigint(a)
    H a = H::from_words(&g[(size_t)r * 96]), want = H::from_words(&g[(size_t)r * 96 + 5 * 12]);
    if (!(a.inverse() == want)) { printf("%s: golden record %d differs\n", name, r); ok = false; }
    xs.push_back(a); xs.push_back(H::from_words(&g[(size_t)r * 96 + 12]));
  }
  H one = H::one(), two = one + one, pm1 = H::zero() - one;
  xs.push_back(H::zero()); xs.push_back(one); xs.push_back(two); xs.push_back(pm1); xs.push_back(pm1 - one);
  xs.push_back(two.inverse_fermat()); xs.push_back(H::zero() - two.inverse_fermat());
  H pw = one;
  for (int i = 0; i < 760; ++i) { pw = pw + pw; xs.push_back(pw); }
  for (int i = 0; i < 12; ++i) { H t = H::zero(); t.l[i] = 1; xs.push_back(t); }                 // single stored words (R^-1 2^(64 i))
  H walk = xs[0];
  while (xs.size() < 4000) { walk = walk * xs[1] + xs[2]; xs.push_back(walk); }
  for (size_t i = 0; i < xs.size(); ++i) {
    H inv = xs[i].inverse();
    if (!(inv == xs[i].inverse_fermat())) { printf("%s: value %zu differs from Fermat\n", name, i); ok = false; break; }
    if (!xs[i].is_zero() && !(inv * xs[i] == one)) { printf("%s: value %zu: a * a^-1 != 1\n", name, i); ok = false; break; }
  }
  // non-canonical stored words (from_words is a raw copy): l = p, 2p, p + 1 must behave as 0, 0, 1 and must terminate
  {
    uint64_t w[12];
    memcpy(w, mnt753::FPC[M].p64, sizeof(w));
    if (!H::from_words(w).inverse().is_zero()) { printf("%s: inverse of the stored word p is not 0\n", name); ok = false; }
    { unsigned __int128 c = 0; for (int i = 0; i < 12; ++i) { c += (unsigned __int128)mnt753::FPC[M].p64[i] * 2; w[i] = (uint64_t)c; c >>= 64; } }
    if (!H::from_words(w).inverse().is_zero()) { printf("%s: inverse of the stored word 2p is not 0\n", name); ok = false; }
    memcpy(w, mnt753::FPC[M].p64, sizeof(w)); w[0] += 1;               // p is odd: no carry
    H one_raw = H::zero(); one_raw.l[0] = 1;
    if (!(H::from_words(w).inverse() == one_raw.inverse())) { printf("%s: inverse of the stored word p + 1 differs from that of 1\n", name); ok = false; }
  }
  auto t0 = clk::now();
  H acc = H::zero();
  for (int i = 0; i < 200; ++i) acc = acc + xs[(size_t)i].inverse();
  auto t1 = clk::now();
  for (int i = 0; i < 200; ++i) acc = acc + xs[(size_t)i].inverse_fermat();
  auto t2 = clk::now();
  printf("%s: %zu values: %s   (binary Euclid %.1f us, Fermat %.1f us per inversion)%s\n", name, xs.size(), ok ? "OK" : "FAILED",
         std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, std::chrono::duration<double, std::micro>(t2 - t1).count() / 200, acc.is_zero() ? "." : "");
  return ok;
}
template <class F> bool ext(const char* name, const char* golden) {
  std::vector<uint64_t> g = slurp(golden);
  const size_t w = 12 * F::DEG;
  if (g.size() != 24 * 7 * w) { printf("%s: cannot read %s\n", name, golden); return false; }
  bool ok = true;
  for (int r = 0; r < 24; ++r) {            // record: a, b, a*b, a^2, a^-1, a+b, a-b
    F a, want;
    for (int k = 0; k < F::DEG; ++k) { a.comp(k) = F::B::from_words(&g[(size_t)r * 7 * w + 12 * k]); want.comp(k) = F::B::from_words(&g[(size_t)r * 7 * w + 4 * w + 12 * k]); }
    if (!(a.inverse() == want)) { printf("%s: golden record %d differs\n", name, r); ok = false; }
  }
  printf("%s: 24 golden records: %s\n", name, ok ? "OK" : "FAILED");
  return ok;
}
int main() {
  bool ok = true;
  ok &= base<MOD_A>("inverse, modulus A (Fr of MNT4753, Fq of MNT6753)", "tests/golden/field_A.bin");
  ok &= base<MOD_B>("inverse, modulus B (Fq of MNT4753, Fr of MNT6753)", "tests/golden/field_B.bin");
  ok &= ext<host::HMnt4G2::F>("inverse Fq2(MNT4753)", "tests/golden/extfield_mnt4.bin");
  ok &= ext<host::HMnt6G2::F>("inverse Fq3(MNT6753)", "tests/golden/extfield_mnt6.bin");
  printf("%s\n", ok ? "ALL OK" : "FAILURES");
  return ok ? 0 : 1;
}
