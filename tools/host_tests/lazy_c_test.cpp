// GPU box: the expression fusion of the wrapper (LazyPoint, host/prover_hip_functions.cpp) from the caller's side.
//   lazy_c_test MNT4753|MNT6753 <params> <input>
// Over fused parameters, B::multiexp_G1 on B1 / L / H starts nothing; C = Ht + Lt + r Bt1 built through G1_scale / G1_add in any
// association and order must come out as ONE MSM over H | L | B1 (MNT753_TRACE=1 prints a line per such MSM on stderr), every other
// expression over the same values must be evaluated the plain way, and all of it must agree with the three values computed
// separately.  Prints "ok <fused evaluations>" or the first mismatch; exit code 0 / 1.  tests/test_prover_gpu.py runs it.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/mnt753_hip.h"
#include "../../include/prover_hip_functions.hpp"

template <typename B>
static bool same_point(int curve, typename B::G1* a, typename B::G1* b) {
  uint64_t x[24], y[24];
  if (mnt753_point_to_affine(curve, MNT753_G1, B::G1_words(a), x) || mnt753_point_to_affine(curve, MNT753_G1, B::G1_words(b), y))
    throw std::runtime_error(mnt753_last_error());
  return memcmp(x, y, sizeof(x)) == 0;
}

template <typename B>
static int run(int curve, const char* params_path, const char* input_path) {
  B::init_public_params();
  auto params = B::read_params(params_path);
  auto input = B::read_input(input_path, params);
  const size_t d = B::params_d(params), m = B::params_m(params);
  auto w = B::input_w(input);
  auto w_off = B::vector_Fr_offset(w, 2);
  auto r = B::input_r(input);
  auto domain = B::get_evaluation_domain(d + 1);
  auto h = B::compute_H_fused(domain, B::input_ca(input), B::input_cb(input), B::input_cc(input));
  auto pB1 = B::params_B1(params); auto pL = B::params_L(params); auto pH = B::params_H(params);
  auto Bt1 = [&]() { return B::multiexp_G1(w, pB1, m + 1); };
  auto Lt = [&]() { return B::multiexp_G1(w_off, pL, m - 1); };
  auto Ht = [&]() { return B::multiexp_G1(h, pH, d); };
  int failures = 0;
  auto expect = [&](const char* what, typename B::G1* got, typename B::G1* want) {
    if (!same_point<B>(curve, got, want)) { printf("MISMATCH: %s\n", what); ++failures; }
  };
  // the three values on their own (touching an unstarted multiexp evaluates it on its own base set, built at that moment)
  auto vB = Bt1(); auto vL = Lt(); auto vH = Ht();
  (void)B::G1_words(vB); (void)B::G1_words(vL); (void)B::G1_words(vH);
  auto want = B::G1_add(vH, B::G1_add(vL, B::G1_scale(r, vB)));
  // 1: the reference's association (cuda_prover_piecewise.cu:85-90)
  expect("Ht + (Lt + r Bt1)", B::G1_add(Ht(), B::G1_add(Lt(), B::G1_scale(r, Bt1()))), want);
  // 2, 3: other associations and orders of the same sum
  expect("(Ht + Lt) + r Bt1", B::G1_add(B::G1_add(Ht(), Lt()), B::G1_scale(r, Bt1())), want);
  expect("r Bt1 + (Lt + Ht)", B::G1_add(B::G1_scale(r, Bt1()), B::G1_add(Lt(), Ht())), want);
  // 4: the explicit entry point
  expect("groth16_C", B::groth16_C(params, h, w_off, w, r), want);
  // 5: NOT the pattern (the factor sits on Lt): evaluated the plain way, and equal to the same expression over the values
  expect("Ht + (r Lt + Bt1)", B::G1_add(Ht(), B::G1_add(B::G1_scale(r, Lt()), Bt1())), B::G1_add(vH, B::G1_add(B::G1_scale(r, vL), vB)));
  // 6: a partial sum of two terms, 7: one term scaled twice, 8: a value mixed with an unstarted multiexp
  expect("Lt + r Bt1", B::G1_add(Lt(), B::G1_scale(r, Bt1())), B::G1_add(vL, B::G1_scale(r, vB)));
  expect("Ht + (Lt + r (r Bt1))", B::G1_add(Ht(), B::G1_add(Lt(), B::G1_scale(r, B::G1_scale(r, Bt1())))),
         B::G1_add(vH, B::G1_add(vL, B::G1_scale(r, B::G1_scale(r, vB)))));
  expect("value Ht + (Lt + r Bt1)", B::G1_add(vH, B::G1_add(Lt(), B::G1_scale(r, Bt1()))), want);
  // 9: a multiexp over a PART of a fused vector is not deferred at all
  {
    auto part = B::multiexp_G1(w, pB1, m / 2);
    auto rest = B::multiexp_G1(B::vector_Fr_offset(w, m / 2), pB1, 0);   // empty MSM: the identity
    expect("half of Bt1 + nothing", B::G1_add(part, rest), part);
  }
  // 10: an unstarted value can be dropped without ever being computed
  B::delete_G1(Bt1());
  printf(failures ? "FAILED %d\n" : "ok\n", failures);
  return failures ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s MNT4753|MNT6753 <params> <input>\n", argv[0]); return 2; }
  try {
    if (!strcmp(argv[1], "MNT4753")) return run<mnt4753_hip>(0, argv[2], argv[3]);
    if (!strcmp(argv[1], "MNT6753")) return run<mnt6753_hip>(1, argv[2], argv[3]);
    fprintf(stderr, "unknown curve %s\n", argv[1]);
    return 2;
  } catch (const std::exception& e) {
    fprintf(stderr, "lazy_c_test: %s\n", e.what());
    return 1;
  }
}
