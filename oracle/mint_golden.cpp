// mint_golden.cpp -- golden-vector minting tool (TEST INFRASTRUCTURE).
//
// OUR program, compiled against the REFERENCE's own sources where they lie (oracle/build_ref.sh, output in
// oracle/_ref/): every expected value below is produced by libff / libfqfft / libsnark code of
// MinaProtocol/snark-challenge-prover-reference, never by this repository's arithmetic.  The vectors are written
// in the reference's wire format (libsnark/serialization.hpp) to tests/golden/ and committed; the reference
// itself does not travel.  Inputs come from /dev/urandom (libff random_element), so the files are captured
// once; regenerate with `oracle/_ref/mint_golden tests/golden` followed by tools/mint_e2e.sh.
//
// File layouts (all elements 12 x u64 Montgomery limbs; points affine, (0,0) = identity):
//   field_<A|B>.bin        N x [a, b, a*b, a+b, a-b, a^-1, -a, as_bigint(a)]
//   group_<curve>_g<k>.bin N x [P, Q, s, P+Q, 2P, P-Q, s*P]            (s an Fr element)
//   msm_<curve>_g<k>_<n>.bin   bases[n], scalars[n], result              (multi_exp_with_mixed_addition, BDLO12)
//   fft_<curve>_<logm>.bin     v[m], FFT(v), iFFT(v), cosetFFT(v), icosetFFT(v)
//   h_<curve>_<logm>.bin       ca[m], cb[m], cc[m], H[m+1]                (compute_H of libsnark/main.cpp:104-163)
//   e2e_<curve>_{params,input}.bin   generate_parameters output at a small log2_d
// Second capture (round 4, `mint_golden <dir> --kat`: writes ONLY these, the files above are never regenerated):
//   extfield_<curve>.bin       N x [a, b, a*b, a^2, a^-1, a+b, a-b]       elements of Fqe = the coordinate field of G2
//                              (Fq2 on MNT4753: fp2.tcc:79-142, Fq3 on MNT6753: fp3.tcc:83-143), write_fqe order c0 | c1 [| c2]
//   groupkat_<curve>_g<k>.bin  N x [P, Q, P+Q, 2P, 2P+Q, 2P+3Q, P-Q]      operator+ / dbl / mixed_add of the reference's group
//                              classes (mnt4753_g1.cpp:134-346, mnt4753_g2.cpp:150-362, mnt6753_g1.cpp, mnt6753_g2.cpp:156-368);
//                              2P+Q is  P.dbl().mixed_add(Q)  (projective + affine),  2P+3Q is  P.dbl() + (Q.dbl() + Q)
//   e2e_mnt6_2p10_{params,input}.bin   the reference generator's `fast` size for MNT6753 (generate_parameters.cpp:127-133)
#include <cstdio>
#include <string>
#include <vector>

#define main reference_generate_parameters_main
#include <libsnark/generate_parameters.cpp>   // brings serialization.hpp, both curves, generate_paramaters<ppT>()
#undef main

#include <libfqfft/evaluation_domain/get_evaluation_domain.hpp>

using namespace libff;

static std::string g_dir;
static FILE* open_out(const std::string& name) {
  std::string p = g_dir + "/" + name;
  FILE* f = fopen(p.c_str(), "wb");
  if (!f) { perror(p.c_str()); exit(1); }
  return f;
}

template <typename ppT>
void mint_field(const char* tag_r, const char* tag_q) {
  {
    FILE* f = open_out(std::string("field_") + tag_r + ".bin");
    for (int i = 0; i < 24; ++i) {
      Fr<ppT> a = Fr<ppT>::random_element(), b = Fr<ppT>::random_element();
      if (i == 0) a = Fr<ppT>::one();
      if (i == 1) b = Fr<ppT>::zero();
      if (i == 2) a = -Fr<ppT>::one();
      if (i == 3) { a = -Fr<ppT>::one(); b = a; }
      write_fr<ppT>(f, a); write_fr<ppT>(f, b); write_fr<ppT>(f, a * b); write_fr<ppT>(f, a + b); write_fr<ppT>(f, a - b);
      write_fr<ppT>(f, a.inverse()); write_fr<ppT>(f, -a);
      auto bi = a.as_bigint();
      fwrite((void*)bi.data, 8, 12, f);
    }
    fclose(f);
  }
  {
    FILE* f = open_out(std::string("field_") + tag_q + ".bin");
    for (int i = 0; i < 24; ++i) {
      Fq<ppT> a = Fq<ppT>::random_element(), b = Fq<ppT>::random_element();
      if (i == 0) a = Fq<ppT>::one();
      if (i == 1) b = Fq<ppT>::zero();
      if (i == 2) a = -Fq<ppT>::one();
      if (i == 3) { a = -Fq<ppT>::one(); b = a; }
      write_fq<ppT>(f, a); write_fq<ppT>(f, b); write_fq<ppT>(f, a * b); write_fq<ppT>(f, a + b); write_fq<ppT>(f, a - b);
      write_fq<ppT>(f, a.inverse()); write_fq<ppT>(f, -a);
      auto bi = a.as_bigint();
      fwrite((void*)bi.data, 8, 12, f);
    }
    fclose(f);
  }
}

template <typename ppT, typename G>
void write_g(FILE* f, const G& g);
template <> void write_g<mnt4753_pp, G1<mnt4753_pp>>(FILE* f, const G1<mnt4753_pp>& g) { write_g1<mnt4753_pp>(f, g); }
template <> void write_g<mnt4753_pp, G2<mnt4753_pp>>(FILE* f, const G2<mnt4753_pp>& g) { write_g2<mnt4753_pp>(f, g); }
template <> void write_g<mnt6753_pp, G1<mnt6753_pp>>(FILE* f, const G1<mnt6753_pp>& g) { write_g1<mnt6753_pp>(f, g); }
template <> void write_g<mnt6753_pp, G2<mnt6753_pp>>(FILE* f, const G2<mnt6753_pp>& g) { write_g2<mnt6753_pp>(f, g); }

template <typename ppT, typename G>
void mint_group(const std::string& name) {
  FILE* f = open_out(name);
  for (int i = 0; i < 8; ++i) {
    G P = Fr<ppT>::random_element() * G::one(), Q = Fr<ppT>::random_element() * G::one();
    Fr<ppT> s = Fr<ppT>::random_element();
    if (i == 1) Q = P;             // doubling branch of operator+
    if (i == 2) Q = -P;            // sum is the identity
    if (i == 3) Q = G::zero();
    if (i == 4) P = G::zero();
    if (i == 5) s = Fr<ppT>::zero();
    if (i == 6) s = Fr<ppT>::one();
    write_g<ppT, G>(f, P); write_g<ppT, G>(f, Q); write_fr<ppT>(f, s);
    write_g<ppT, G>(f, P + Q); write_g<ppT, G>(f, P.dbl()); write_g<ppT, G>(f, P - Q); write_g<ppT, G>(f, s * P);
  }
  fclose(f);
}

template <typename ppT, typename G>
void mint_msm(const std::string& name, size_t n, size_t chunks) {
  std::vector<G> bases;
  std::vector<Fr<ppT>> scalars;
  G cur = Fr<ppT>::random_element() * G::one();
  const G step = Fr<ppT>::random_element() * G::one();
  for (size_t i = 0; i < n; ++i) {
    G b = cur;
    b.to_affine_coordinates();     // params hold affine points (read_g1 gives Z = 1)
    bases.push_back(b);
    scalars.push_back(Fr<ppT>::random_element());
    cur = cur + step;
  }
  if (n >= 17) {
    scalars[0] = Fr<ppT>::zero();
    scalars[1] = Fr<ppT>::one();
    scalars[2] = -Fr<ppT>::one();
    bases[3] = G::zero();                                   // identity base, as in real parameter files
    bases[5] = bases[4]; scalars[5] = scalars[4];           // duplicate pair -> equal points meet in a bucket
    bases[7] = -bases[6]; scalars[7] = scalars[6];          // P and -P with the same scalar
    scalars[8] = Fr<ppT>(12345);                            // short scalar (upper windows empty)
    bases[n - 1] = G::zero();
  }
  G res = multi_exp_with_mixed_addition<G, Fr<ppT>, multi_exp_method_BDLO12>(bases.begin(), bases.end(), scalars.begin(),
                                                                               scalars.end(), chunks);
  FILE* f = open_out(name);
  for (auto& b : bases) write_g<ppT, G>(f, b);
  for (auto& s : scalars) write_fr<ppT>(f, s);
  write_g<ppT, G>(f, res);
  fclose(f);
}

template <typename ppT>
void mint_fft(const std::string& tag, size_t logm) {
  const size_t m = (size_t)1 << logm;
  bool err = false;
  auto domain = libfqfft::get_evaluation_domain<Fr<ppT>>(m); (void)err;

  std::vector<Fr<ppT>> v(m);
  for (auto& x : v) x = Fr<ppT>::random_element();
  if (m >= 4) { v[0] = Fr<ppT>::zero(); v[1] = Fr<ppT>::one(); v[2] = -Fr<ppT>::one(); }
  FILE* f = open_out("fft_" + tag + "_" + std::to_string(logm) + ".bin");
  for (auto& x : v) write_fr<ppT>(f, x);
  const Fr<ppT> g = Fr<ppT>::multiplicative_generator;
  for (int kind = 0; kind < 4; ++kind) {
    std::vector<Fr<ppT>> a = v;
    if (kind == 0) domain->FFT(a);
    if (kind == 1) domain->iFFT(a);
    if (kind == 2) domain->cosetFFT(a, g);
    if (kind == 3) domain->icosetFFT(a, g);
    for (auto& x : a) write_fr<ppT>(f, x);
  }
  fclose(f);
}

// compute_H exactly as libsnark/main.cpp:104-163 spells it with libfqfft calls
template <typename ppT>
void mint_h(const std::string& tag, size_t logm) {
  const size_t m = (size_t)1 << logm;
  bool err = false;
  auto domain = libfqfft::get_evaluation_domain<Fr<ppT>>(m); (void)err;
  std::vector<Fr<ppT>> ca(m), cb(m), cc(m);
  for (size_t i = 0; i < m; ++i) { ca[i] = Fr<ppT>::random_element(); cb[i] = Fr<ppT>::random_element(); cc[i] = Fr<ppT>::random_element(); }
  FILE* f = open_out("h_" + tag + "_" + std::to_string(logm) + ".bin");
  for (auto& x : ca) write_fr<ppT>(f, x);
  for (auto& x : cb) write_fr<ppT>(f, x);
  for (auto& x : cc) write_fr<ppT>(f, x);
  const Fr<ppT> g = Fr<ppT>::multiplicative_generator;
  domain->iFFT(ca); domain->iFFT(cb);
  domain->cosetFFT(ca, g); domain->cosetFFT(cb, g);
  for (size_t i = 0; i < m; ++i) ca[i] = ca[i] * cb[i];
  domain->iFFT(cc); domain->cosetFFT(cc, g);
  for (size_t i = 0; i < m; ++i) ca[i] = ca[i] - cc[i];
  domain->divide_by_Z_on_coset(ca);
  domain->icosetFFT(ca, g);
  std::vector<Fr<ppT>> h(m + 1, Fr<ppT>::zero());
  for (size_t i = 0; i < m; ++i) h[i] = ca[i];
  for (auto& x : h) write_fr<ppT>(f, x);
  fclose(f);
}

static void only_c0(mnt4753_Fq2& a) { a.c1 = mnt4753_Fq::zero(); }
static void only_c0(mnt6753_Fq3& a) { a.c1 = mnt6753_Fq::zero(); a.c2 = mnt6753_Fq::zero(); }
template <typename ppT>
void mint_extfield(const std::string& name) {
  typedef Fqe<ppT> E;
  FILE* f = open_out(name);
  for (int i = 0; i < 24; ++i) {
    E a = E::random_element(), b = E::random_element();
    if (i == 0) a = E::one();
    if (i == 1) b = E::zero();
    if (i == 2) a = -E::one();
    if (i == 3) b = a;
    if (i == 4) { a = E::one() + E::one(); b = a.inverse(); }             // a * b = 1
    if (i == 5) only_c0(a);                                              // only the constant coefficient is non-zero
    if (i == 6) { a.c0 = Fq<ppT>::zero(); b.c0 = Fq<ppT>::zero(); }      // constant coefficients zero
    if (i == 7) b = -a;
    write_fqe<ppT>(f, a); write_fqe<ppT>(f, b); write_fqe<ppT>(f, a * b); write_fqe<ppT>(f, a.squared());
    write_fqe<ppT>(f, a.inverse()); write_fqe<ppT>(f, a + b); write_fqe<ppT>(f, a - b);
  }
  fclose(f);
}

template <typename ppT, typename G>
void mint_groupkat(const std::string& name) {
  FILE* f = open_out(name);
  for (int i = 0; i < 16; ++i) {
    G P = Fr<ppT>::random_element() * G::one(), Q = Fr<ppT>::random_element() * G::one();
    if (i == 1) Q = P;                    // operator+ : doubling branch
    if (i == 2) Q = -P;                   // P + Q = O
    if (i == 3) Q = G::zero();
    if (i == 4) P = G::zero();
    if (i == 5) Q = P.dbl();              // mixed_add meets an equal point: 2P + Q doubles
    if (i == 6) Q = -(P.dbl());           // 2P + Q = O
    if (i == 7) { P = G::zero(); Q = G::zero(); }
    if (i == 8) Q = P + P + P;
    P.to_affine_coordinates(); Q.to_affine_coordinates();
    const G P2 = P.dbl();
    G Q3 = Q.dbl() + Q;
    write_g<ppT, G>(f, P); write_g<ppT, G>(f, Q);
    write_g<ppT, G>(f, P + Q); write_g<ppT, G>(f, P2);
    write_g<ppT, G>(f, Q.is_zero() ? P2 : P2.mixed_add(Q));   // (mixed_add reads other as affine: the identity has no affine form)
    write_g<ppT, G>(f, P2 + Q3); write_g<ppT, G>(f, P - Q);
  }
  fclose(f);
}

static void mint_kat() {
  mint_extfield<mnt4753_pp>("extfield_mnt4.bin");
  mint_extfield<mnt6753_pp>("extfield_mnt6.bin");
  mint_groupkat<mnt4753_pp, G1<mnt4753_pp>>("groupkat_mnt4_g1.bin");
  mint_groupkat<mnt4753_pp, G2<mnt4753_pp>>("groupkat_mnt4_g2.bin");
  mint_groupkat<mnt6753_pp, G1<mnt6753_pp>>("groupkat_mnt6_g1.bin");
  mint_groupkat<mnt6753_pp, G2<mnt6753_pp>>("groupkat_mnt6_g2.bin");
  // the reference generator's own `fast` size for MNT6753 (generate_parameters.cpp:127-133: log2_d = 10)
  std::string p6 = g_dir + "/e2e_mnt6_2p10_params.bin", i6 = g_dir + "/e2e_mnt6_2p10_input.bin";
  generate_paramaters<mnt6753_pp>(10, (char*)p6.c_str(), (char*)i6.c_str());
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <output dir> [--kat]\n", argv[0]); return 2; }
  g_dir = argv[1];
  mnt4753_pp::init_public_params();
  mnt6753_pp::init_public_params();
  libff::inhibit_profiling_info = true;
  libff::inhibit_profiling_counters = true;
  if (argc > 2 && std::string(argv[2]) == "--kat") { mint_kat(); return 0; }

  mint_field<mnt4753_pp>("A", "B");   // Fr(MNT4753) = modulus A, Fq(MNT4753) = modulus B

  mint_group<mnt4753_pp, G1<mnt4753_pp>>("group_mnt4_g1.bin");
  mint_group<mnt4753_pp, G2<mnt4753_pp>>("group_mnt4_g2.bin");
  mint_group<mnt6753_pp, G1<mnt6753_pp>>("group_mnt6_g1.bin");
  mint_group<mnt6753_pp, G2<mnt6753_pp>>("group_mnt6_g2.bin");

  for (size_t n : {1, 2, 3, 17, 256, 1000}) {
    mint_msm<mnt4753_pp, G1<mnt4753_pp>>("msm_mnt4_g1_" + std::to_string(n) + ".bin", n, n >= 256 ? 3 : 1);
    mint_msm<mnt6753_pp, G1<mnt6753_pp>>("msm_mnt6_g1_" + std::to_string(n) + ".bin", n, n >= 256 ? 3 : 1);
  }
  for (size_t n : {1, 2, 17, 128}) {
    mint_msm<mnt4753_pp, G2<mnt4753_pp>>("msm_mnt4_g2_" + std::to_string(n) + ".bin", n, 1);
    mint_msm<mnt6753_pp, G2<mnt6753_pp>>("msm_mnt6_g2_" + std::to_string(n) + ".bin", n, 1);
  }
  for (size_t logm : {1, 2, 3, 6, 10}) {
    mint_fft<mnt4753_pp>("mnt4", logm);
    mint_fft<mnt6753_pp>("mnt6", logm);
  }
  mint_h<mnt4753_pp>("mnt4", 3); mint_h<mnt4753_pp>("mnt4", 8);
  mint_h<mnt6753_pp>("mnt6", 3); mint_h<mnt6753_pp>("mnt6", 8);

  // tiny end-to-end parameter / input sets from the reference's own generator
  std::string p4 = g_dir + "/e2e_mnt4_params.bin", i4 = g_dir + "/e2e_mnt4_input.bin";
  std::string p6 = g_dir + "/e2e_mnt6_params.bin", i6 = g_dir + "/e2e_mnt6_input.bin";
  generate_paramaters<mnt4753_pp>(6, (char*)p4.c_str(), (char*)i4.c_str());
  generate_paramaters<mnt6753_pp>(5, (char*)p6.c_str(), (char*)i6.c_str());
  return 0;
}
