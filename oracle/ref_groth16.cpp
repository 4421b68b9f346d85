// ref_groth16.cpp -- fixtures and checker for the steps either side of the hot path (TEST INFRASTRUCTURE).
//
// OUR program, compiled against the REFERENCE's own libsnark / libff where they lie (oracle/build_ref.sh, output
// oracle/_ref/).  Nothing in it is this repository's arithmetic: the constraint system, the keys, the witness map and the
// verdict all come from MinaProtocol/snark-challenge-prover-reference code.
//
//   ref_groth16 mint <MNT4753|MNT6753> <log2_d> <dir>
//       generate_r1cs_example_with_field_input + r1cs_gg_ppzksnark_generator (exactly what libsnark/generate_parameters.cpp:38-39
//       runs), then writes into <dir>
//         params.bin, input.bin   the challenge files, layout of generate_parameters.cpp:60-108
//         keys.bin                alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2 in the wire format (the proving-key elements the
//                                 challenge files leave out; main.cpp:312-319 needs them to complete a proof)
//         vk.txt                  the verification key, libsnark's own text serialisation (operator<<)
//         r1cs.bin                the constraint system: u64 num_inputs, m (variables), nc (constraints); then for each of the
//                                 matrices a, b, c: u64 row_ptr[nc + 1], u32 col[nnz] (0 = the constant one = w[0] of the input
//                                 file), Fr coeff[nnz] (wire format)
//         witness.bin             w[m + 1] and r only (the input of the witness-map front end: no ca / cb / cc)
//   ref_groth16 verify <MNT4753|MNT6753> <dir> <full_proof>
//       <full_proof> = A (G1) | B (G2) | C (G1) affine wire format, a COMPLETE Groth16 proof (alpha, beta, delta and s terms in);
//       runs r1cs_gg_ppzksnark_verifier_strong_IC (zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.tcc:515-567)
//       with the primary input taken from <dir>/input.bin and prints VERIFIED or REJECTED (exit code 0 / 3).
//   ref_groth16 complete <MNT4753|MNT6753> <dir> <challenge_proof> <s_file> <out>
//       the reference's own completion (main.cpp:312-319) of a challenge proof with the given s (one Fr, wire format): the
//       expected value of main_hip's `complete` mode.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <libff/common/profiling.hpp>
#include <libff/algebra/curves/mnt753/mnt4753/mnt4753_pp.hpp>
#include <libff/algebra/curves/mnt753/mnt6753/mnt6753_pp.hpp>
#include <libsnark/serialization.hpp>
#include <libsnark/relations/constraint_satisfaction_problems/r1cs/examples/r1cs_examples.hpp>
#include <libsnark/zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.hpp>

using namespace libsnark;
using namespace libff;

static FILE* open_or_die(const std::string& p, const char* mode) {
  FILE* f = fopen(p.c_str(), mode);
  if (!f) { perror(p.c_str()); exit(1); }
  return f;
}
static void put_u64(FILE* f, uint64_t v) { fwrite(&v, 8, 1, f); }

template <typename ppT>
int mint(int log2_d, const std::string& dir) {
  ppT::init_public_params();
  libff::inhibit_profiling_info = true;
  libff::inhibit_profiling_counters = true;
  const size_t primary_input_size = 1;
  const size_t d_plus_1 = (size_t)1 << log2_d, d = d_plus_1 - 1;
  r1cs_example<Fr<ppT>> example = generate_r1cs_example_with_field_input<Fr<ppT>>(d - 1, primary_input_size);
  r1cs_gg_ppzksnark_keypair<ppT> keypair = r1cs_gg_ppzksnark_generator<ppT>(example.constraint_system);
  const auto& cs = keypair.pk.constraint_system;
  r1cs_variable_assignment<Fr<ppT>> full = example.primary_input;
  full.insert(full.end(), example.auxiliary_input.begin(), example.auxiliary_input.end());
  const size_t m = example.constraint_system.num_variables(), nc = cs.num_constraints();

  // evaluations of the constraint system on the assignment, as the challenge's input file carries them (generate_parameters.cpp:44-57)
  std::vector<Fr<ppT>> ca(d_plus_1, Fr<ppT>::zero()), cb(d_plus_1, Fr<ppT>::zero()), cc(d_plus_1, Fr<ppT>::zero());
  for (size_t i = 0; i <= primary_input_size; ++i) ca[i + nc] = (i > 0 ? full[i - 1] : Fr<ppT>::one());
  for (size_t i = 0; i < nc; ++i) {
    ca[i] += cs.constraints[i].a.evaluate(full);
    cb[i] += cs.constraints[i].b.evaluate(full);
    cc[i] += cs.constraints[i].c.evaluate(full);
  }
  {
    FILE* f = open_or_die(dir + "/params.bin", "wb");
    write_size_t(f, d); write_size_t(f, m);
    for (size_t i = 0; i <= m; ++i) write_g1<ppT>(f, keypair.pk.A_query[i]);
    for (size_t i = 0; i <= m; ++i) write_g1<ppT>(f, keypair.pk.B_query[i].h);
    for (size_t i = 0; i <= m; ++i) write_g2<ppT>(f, keypair.pk.B_query[i].g);
    for (size_t i = 0; i < m - 1; ++i) write_g1<ppT>(f, keypair.pk.L_query[i]);
    for (size_t i = 0; i < d; ++i) write_g1<ppT>(f, keypair.pk.H_query[i]);
    fclose(f);
  }
  const Fr<ppT> r = Fr<ppT>::random_element();
  {
    FILE* f = open_or_die(dir + "/input.bin", "wb");
    FILE* g = open_or_die(dir + "/witness.bin", "wb");
    write_fr<ppT>(f, Fr<ppT>::one()); write_fr<ppT>(g, Fr<ppT>::one());
    for (size_t i = 0; i < m; ++i) { write_fr<ppT>(f, full[i]); write_fr<ppT>(g, full[i]); }
    for (auto& v : ca) write_fr<ppT>(f, v);
    for (auto& v : cb) write_fr<ppT>(f, v);
    for (auto& v : cc) write_fr<ppT>(f, v);
    write_fr<ppT>(f, r); write_fr<ppT>(g, r);
    fclose(f); fclose(g);
  }
  {
    FILE* f = open_or_die(dir + "/keys.bin", "wb");
    write_g1<ppT>(f, keypair.pk.alpha_g1); write_g1<ppT>(f, keypair.pk.beta_g1); write_g2<ppT>(f, keypair.pk.beta_g2);
    write_g1<ppT>(f, keypair.pk.delta_g1); write_g2<ppT>(f, keypair.pk.delta_g2);
    fclose(f);
  }
  {
    std::ofstream vk((dir + "/vk.txt").c_str());
    vk << keypair.vk;
  }
  {
    FILE* f = open_or_die(dir + "/r1cs.bin", "wb");
    put_u64(f, cs.num_inputs()); put_u64(f, m); put_u64(f, nc);
    for (int which = 0; which < 3; ++which) {
      auto lc = [&](size_t i) -> const linear_combination<Fr<ppT>>& {
        return which == 0 ? cs.constraints[i].a : (which == 1 ? cs.constraints[i].b : cs.constraints[i].c);
      };
      uint64_t nnz = 0;
      put_u64(f, 0);
      for (size_t i = 0; i < nc; ++i) { nnz += lc(i).terms.size(); put_u64(f, nnz); }
      for (size_t i = 0; i < nc; ++i)
        for (auto& t : lc(i).terms) { uint32_t c = (uint32_t)t.index; fwrite(&c, 4, 1, f); }
      for (size_t i = 0; i < nc; ++i)
        for (auto& t : lc(i).terms) write_fr<ppT>(f, t.coeff);
    }
    fclose(f);
  }
  printf("minted d=%zu m=%zu constraints=%zu into %s\n", d, m, nc, dir.c_str());
  return 0;
}

template <typename ppT>
r1cs_gg_ppzksnark_verification_key<ppT> load_vk(const std::string& dir) {
  r1cs_gg_ppzksnark_verification_key<ppT> vk;
  std::ifstream in((dir + "/vk.txt").c_str());
  if (!in) { fprintf(stderr, "cannot open %s/vk.txt\n", dir.c_str()); exit(1); }
  in >> vk;
  return vk;
}

template <typename ppT>
int verify(const std::string& dir, const std::string& proof_path) {
  ppT::init_public_params();
  libff::inhibit_profiling_info = true;
  libff::inhibit_profiling_counters = true;
  auto vk = load_vk<ppT>(dir);
  FILE* in = open_or_die(dir + "/input.bin", "rb");
  (void)read_fr<ppT>(in);                       // w[0] = 1
  std::vector<Fr<ppT>> primary(1, read_fr<ppT>(in));   // primary_input_size = 1 (main.cpp:301-304)
  fclose(in);
  FILE* pf = open_or_die(proof_path, "rb");
  G1<ppT> A = read_g1<ppT>(pf);
  G2<ppT> B = read_g2<ppT>(pf);
  G1<ppT> C = read_g1<ppT>(pf);
  fclose(pf);
  r1cs_gg_ppzksnark_proof<ppT> proof(std::move(A), std::move(B), std::move(C));
  const bool ok = r1cs_gg_ppzksnark_verifier_strong_IC<ppT>(vk, primary, proof);
  printf("%s\n", ok ? "VERIFIED" : "REJECTED");
  return ok ? 0 : 3;
}

template <typename ppT>
int complete(const std::string& dir, const std::string& challenge_path, const std::string& s_path, const std::string& out_path) {
  ppT::init_public_params();
  FILE* kf = open_or_die(dir + "/keys.bin", "rb");
  G1<ppT> alpha_g1 = read_g1<ppT>(kf), beta_g1 = read_g1<ppT>(kf);
  G2<ppT> beta_g2 = read_g2<ppT>(kf);
  G1<ppT> delta_g1 = read_g1<ppT>(kf);
  G2<ppT> delta_g2 = read_g2<ppT>(kf);
  fclose(kf);
  FILE* in = open_or_die(dir + "/witness.bin", "rb");
  fseek(in, -96, SEEK_END);
  Fr<ppT> r = read_fr<ppT>(in);
  fclose(in);
  FILE* sf = open_or_die(s_path, "rb");
  Fr<ppT> s = read_fr<ppT>(sf);
  fclose(sf);
  FILE* pf = open_or_die(challenge_path, "rb");
  G1<ppT> A = read_g1<ppT>(pf);
  G2<ppT> B = read_g2<ppT>(pf);
  G1<ppT> C = read_g1<ppT>(pf);
  fclose(pf);
  // main.cpp:312-319
  G1<ppT> g1_A = alpha_g1 + A + r * delta_g1;
  G2<ppT> g2_B = beta_g2 + B + s * delta_g2;
  G1<ppT> g1_C = C + s * g1_A + r * beta_g1;
  FILE* of = open_or_die(out_path, "wb");
  write_g1<ppT>(of, g1_A); write_g2<ppT>(of, g2_B); write_g1<ppT>(of, g1_C);
  fclose(of);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s mint|verify|complete <curve> ...\n", argv[0]); return 2; }
  const std::string mode(argv[1]), curve(argv[2]);
  const bool c4 = curve == "MNT4753";
  if (!c4 && curve != "MNT6753") return 2;
  if (mode == "mint" && argc >= 5) return c4 ? mint<mnt4753_pp>(atoi(argv[3]), argv[4]) : mint<mnt6753_pp>(atoi(argv[3]), argv[4]);
  if (mode == "verify" && argc >= 5) return c4 ? verify<mnt4753_pp>(argv[3], argv[4]) : verify<mnt6753_pp>(argv[3], argv[4]);
  if (mode == "complete" && argc >= 7)
    return c4 ? complete<mnt4753_pp>(argv[3], argv[4], argv[5], argv[6]) : complete<mnt6753_pp>(argv[3], argv[4], argv[5], argv[6]);
  return 2;
}
