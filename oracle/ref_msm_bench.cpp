// ref_msm_bench.cpp -- CPU baseline driver (TEST / BENCH INFRASTRUCTURE).
//
// OUR program, compiled against the REFERENCE's own libff where it lies (oracle/build_ref.sh, output oracle/_ref/):
// times exactly what B::multiexp_G1 of the reference wrapper runs (libsnark/prover_reference_functions.cpp:247-256):
//     libff::multi_exp_with_mixed_addition<G1, Fr, multi_exp_method_BDLO12>(bases, scalars, chunks = omp_get_max_threads())
// on MNT4753 G1 over a file of  n affine bases (192 B each, wire format)  followed by  n scalars (96 B each), and prints
// one JSON line with the points per second and the affine result, so that bench.py can use the true reference (kind
// "reference") as its cpu_baseline on the GPU box's host cores and compare the result with the GPU's on the same sample.
//
//   ref_msm_bench <file> <n>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <omp.h>

#include <libff/algebra/curves/mnt753/mnt4753/mnt4753_pp.hpp>
#include <libff/algebra/scalar_multiplication/multiexp.hpp>
#include <libsnark/serialization.hpp>

using namespace libff;

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <bases+scalars file> <n>\n", argv[0]); return 2; }
  const size_t n = strtoull(argv[2], nullptr, 10);
  mnt4753_pp::init_public_params();
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  std::vector<G1<mnt4753_pp>> bases;
  std::vector<Fr<mnt4753_pp>> scalars;
  bases.reserve(n); scalars.reserve(n);
  for (size_t i = 0; i < n; ++i) bases.push_back(read_g1<mnt4753_pp>(f));
  for (size_t i = 0; i < n; ++i) scalars.push_back(read_fr<mnt4753_pp>(f));
  fclose(f);
  const size_t chunks = (size_t)omp_get_max_threads();
  libff::inhibit_profiling_info = true;
  auto t0 = std::chrono::steady_clock::now();
  G1<mnt4753_pp> res = multi_exp_with_mixed_addition<G1<mnt4753_pp>, Fr<mnt4753_pp>, multi_exp_method_BDLO12>(
      bases.begin(), bases.end(), scalars.begin(), scalars.end(), chunks);
  auto t1 = std::chrono::steady_clock::now();
  const double dt = std::chrono::duration<double>(t1 - t0).count();
  // affine result, wire format, as hex words (x then y)
  res.to_affine_coordinates();
  printf("{\"n\": %zu, \"threads\": %zu, \"seconds\": %.6f, \"points_per_s\": %.3f, \"result_affine_hex\": \"", n, chunks, dt, (double)n / dt);
  const Fq<mnt4753_pp> xy[2] = {res.is_zero() ? Fq<mnt4753_pp>::zero() : res.X(), res.is_zero() ? Fq<mnt4753_pp>::zero() : res.Y()};
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 12; ++i) printf("%016llx", (unsigned long long)xy[k].mont_repr.data[i]);
  printf("\"}\n");
  return 0;
}
