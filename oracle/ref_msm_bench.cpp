// ref_msm_bench.cpp -- CPU baseline driver (TEST / BENCH INFRASTRUCTURE).
//
// OUR program, compiled against the REFERENCE's own libff where it lies (oracle/build_ref.sh, output oracle/_ref/):
// times exactly what B::multiexp_G1 of the reference wrapper runs (libsnark/prover_reference_functions.cpp:247-256):
//     libff::multi_exp_with_mixed_addition<G1, Fr, multi_exp_method_BDLO12>(bases, scalars, chunks = omp_get_max_threads())
// on MNT4753 G1 over a file of  n affine bases (192 B each, wire format)  followed by  n scalars (96 B each), and prints
// one JSON line with the points per second and the affine result, so that bench.py can use the true reference (kind
// "reference") as its cpu_baseline on the GPU box's host cores and compare the result with the GPU's on the same sample.
//
// With two more arguments the same call on the other curve / group, as B::multiexp_G2 runs it (prover_reference_functions.cpp:257-265,
// 554-571: the same template, multiexp<G, Fr> at lines 27-40) -- the direct libff check of the G2 and MNT6753 MSMs at size
// (tests/test_msm_gpu.py).  A G2 base is x then y, each extension-degree base-field elements (serialization.hpp:94-113).
//
//   ref_msm_bench <file> <n> [MNT4753|MNT6753 [G1|G2]]
//
// Fourth argument H instead of a group (round 6): the file holds ca[m], cb[m], cc[m] (Fr, wire format) and the program runs compute_H
// exactly as libsnark/main.cpp:104-163 spells it with libfqfft calls, printing coefficients_for_H (m + 1 elements) as hex -- how
// tools/gen_selftest_data.py mints the expected words of the product's self-test (mnt753_self_test) for ITS inputs.
//   ref_msm_bench <file> <m> MNT4753|MNT6753 H
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

#include <libff/algebra/curves/mnt753/mnt4753/mnt4753_pp.hpp>
#include <libff/algebra/curves/mnt753/mnt6753/mnt6753_pp.hpp>
#include <libff/algebra/scalar_multiplication/multiexp.hpp>
#include <libsnark/serialization.hpp>
#include <libfqfft/evaluation_domain/get_evaluation_domain.hpp>

using namespace libff;

static void hex_fq(const mp_limb_t* d) {
  for (int i = 0; i < 12; ++i) printf("%016llx", (unsigned long long)d[i]);
}
void hex_elem(const mnt4753_Fq& a) { hex_fq(a.mont_repr.data); }
void hex_elem(const mnt6753_Fq& a) { hex_fq(a.mont_repr.data); }
void hex_elem(const mnt4753_Fq2& a) { hex_fq(a.c0.mont_repr.data); hex_fq(a.c1.mont_repr.data); }
void hex_elem(const mnt6753_Fq3& a) { hex_fq(a.c0.mont_repr.data); hex_fq(a.c1.mont_repr.data); hex_fq(a.c2.mont_repr.data); }

template <class ppT, class G, G (*read_point)(FILE*)> int run(FILE* f, size_t n) {
  std::vector<G> bases;
  std::vector<Fr<ppT>> scalars;
  bases.reserve(n); scalars.reserve(n);
  for (size_t i = 0; i < n; ++i) bases.push_back(read_point(f));
  for (size_t i = 0; i < n; ++i) scalars.push_back(read_fr<ppT>(f));
  fclose(f);
  const size_t chunks = (size_t)omp_get_max_threads();
  libff::inhibit_profiling_info = true;
  auto t0 = std::chrono::steady_clock::now();
  G res = multi_exp_with_mixed_addition<G, Fr<ppT>, multi_exp_method_BDLO12>(bases.begin(), bases.end(), scalars.begin(), scalars.end(), chunks);
  auto t1 = std::chrono::steady_clock::now();
  const double dt = std::chrono::duration<double>(t1 - t0).count();
  // affine result, wire format, as hex words (x then y; an extension-field coordinate component by component)
  res.to_affine_coordinates();
  printf("{\"n\": %zu, \"threads\": %zu, \"seconds\": %.6f, \"points_per_s\": %.3f, \"result_affine_hex\": \"", n, chunks, dt, (double)n / dt);
  if (res.is_zero()) { decltype(res.X()) z = res.X() - res.X(); hex_elem(z); hex_elem(z); }
  else { hex_elem(res.X()); hex_elem(res.Y()); }
  printf("\"}\n");
  return 0;
}

template <class ppT> int run_h(FILE* f, size_t m) {
  std::vector<Fr<ppT>> ca(m), cb(m), cc(m);
  for (auto* v : {&ca, &cb, &cc}) for (size_t i = 0; i < m; ++i) (*v)[i] = read_fr<ppT>(f);
  fclose(f);
  libff::inhibit_profiling_info = true;
  auto domain = libfqfft::get_evaluation_domain<Fr<ppT>>(m);
  const Fr<ppT> g = Fr<ppT>::multiplicative_generator;
  domain->iFFT(ca); domain->iFFT(cb);
  domain->cosetFFT(ca, g); domain->cosetFFT(cb, g);
  for (size_t i = 0; i < m; ++i) ca[i] = ca[i] * cb[i];
  domain->iFFT(cc); domain->cosetFFT(cc, g);
  for (size_t i = 0; i < m; ++i) ca[i] = ca[i] - cc[i];
  domain->divide_by_Z_on_coset(ca);
  domain->icosetFFT(ca, g);
  printf("{\"m\": %zu, \"result_hex\": \"", m);
  for (size_t i = 0; i < m; ++i) hex_fq(ca[i].mont_repr.data);
  const Fr<ppT> z = Fr<ppT>::zero();
  hex_fq(z.mont_repr.data);
  printf("\"}\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <bases+scalars file> <n> [MNT4753|MNT6753 [G1|G2]]\n", argv[0]); return 2; }
  const size_t n = strtoull(argv[2], nullptr, 10);
  const bool mnt6 = argc > 3 && !strcmp(argv[3], "MNT6753");
  const bool g2 = argc > 4 && !strcmp(argv[4], "G2");
  const bool h = argc > 4 && !strcmp(argv[4], "H");
  if (argc > 3 && !mnt6 && strcmp(argv[3], "MNT4753")) { fprintf(stderr, "curve: MNT4753 or MNT6753\n"); return 2; }
  if (argc > 4 && !g2 && !h && strcmp(argv[4], "G1")) { fprintf(stderr, "group: G1 or G2 (or H: compute_H)\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  if (h) {
    if (mnt6) { mnt6753_pp::init_public_params(); return run_h<mnt6753_pp>(f, n); }
    mnt4753_pp::init_public_params();
    return run_h<mnt4753_pp>(f, n);
  }
  if (mnt6) {
    mnt6753_pp::init_public_params();
    return g2 ? run<mnt6753_pp, G2<mnt6753_pp>, read_g2<mnt6753_pp>>(f, n) : run<mnt6753_pp, G1<mnt6753_pp>, read_g1<mnt6753_pp>>(f, n);
  }
  mnt4753_pp::init_public_params();
  return g2 ? run<mnt4753_pp, G2<mnt4753_pp>, read_g2<mnt4753_pp>>(f, n) : run<mnt4753_pp, G1<mnt4753_pp>, read_g1<mnt4753_pp>>(f, n);
}
