// prover_hip_functions.hpp -- the MI355X drop-in for the reference's prover wrapper.
//
// Mirrors libsnark/prover_reference_include/prover_reference_functions.hpp:5-162 of
// MinaProtocol/snark-challenge-prover-reference member for member: the same nested opaque types and the
// same static methods with the same argument meaning, so that the reference's own drivers
//     template <typename B> compute_H(...)   (cuda_prover_piecewise.cu:18-53)
//     template <typename B> run_prover(...)  (cuda_prover_piecewise.cu:55-98)
// compile unchanged with B = mnt4753_hip / mnt6753_hip.  Everything lands in libmnt753_hip.so through the
// C ABI of include/mnt753_hip.h; vectors live in HBM, points that the driver handles one at a time
// (G1 / G2 / field) live on the host in the reference's in-memory (wire) format.
//
// Differences a maintainer should know:
//   * errors: the reference has none on this path (unchecked fopen/fread).  Here a failed HIP call, a missing
//     device or a short file throws std::runtime_error with the C ABI's message.
//   * ownership is the reference's: every accessor returns a new wrapper that shares the underlying storage;
//     delete_* frees the wrapper (delete_G2 takes a G2*, fixing the reference's `delete_G2(G1*)` typo, hpp:73).
//   * vector_Fr_copy_into of the MNT4753 class ignores the source offset exactly like the reference
//     (prover_reference_functions.cpp:209-212); the MNT6753 class honours it (:515-520).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>

namespace mnt753_hip_detail {
struct DeviceBuffer;   // RAII hipMalloc'd block
struct BaseSetHolder;  // RAII mnt753_bases*
struct DomainHolder;   // RAII mnt753_domain*
}  // namespace mnt753_hip_detail

template <int CURVE>
class mnt753_hip_impl {
public:
  class groth16_input;
  class groth16_params;
  struct evaluation_domain;
  struct field;
  struct G1;
  struct G2;
  struct vector_Fr;
  struct vector_G1;
  struct vector_G2;
  class r1cs;   // extension: a constraint system resident on the device (witness-map front end)

  static void init_public_params();

  static void print_G1(G1 *a);
  static void print_G2(G2 *a);

  static evaluation_domain *get_evaluation_domain(size_t d);

  static G1 *G1_add(G1 *a, G1 *b);
  static G1 *G1_scale(field *a, G1 *b);

  static void vector_Fr_muleq(vector_Fr *a, vector_Fr *b, size_t size);
  static void vector_Fr_subeq(vector_Fr *a, vector_Fr *b, size_t size);
  static vector_Fr *vector_Fr_offset(vector_Fr *a, size_t offset);
  static void vector_Fr_copy_into(vector_Fr *src, vector_Fr *dst, size_t length);
  static vector_Fr *vector_Fr_zeros(size_t length);

  static void domain_iFFT(evaluation_domain *domain, vector_Fr *a);
  static void domain_cosetFFT(evaluation_domain *domain, vector_Fr *a);
  static void domain_icosetFFT(evaluation_domain *domain, vector_Fr *a);
  static void domain_divide_by_Z_on_coset(evaluation_domain *domain, vector_Fr *a);
  static size_t domain_get_m(evaluation_domain *domain);

  static G1 *multiexp_G1(vector_Fr *scalar_start, vector_G1 *g_start, size_t length);
  static G2 *multiexp_G2(vector_Fr *scalar_start, vector_G2 *g_start, size_t length);

  static groth16_input *read_input(const char *path, groth16_params *params);

  static vector_Fr *input_w(groth16_input *input);
  static vector_Fr *input_ca(groth16_input *input);
  static vector_Fr *input_cb(groth16_input *input);
  static vector_Fr *input_cc(groth16_input *input);
  static field *input_r(groth16_input *input);

  static groth16_params *read_params(const char *path);

  static size_t params_d(groth16_params *params);
  static size_t params_m(groth16_params *params);
  static vector_G1 *params_A(groth16_params *params);
  static vector_G1 *params_B1(groth16_params *params);
  static vector_G1 *params_L(groth16_params *params);
  static vector_G1 *params_H(groth16_params *params);
  static vector_G2 *params_B2(groth16_params *params);

  static void delete_G1(G1 *a);
  static void delete_G2(G2 *a);
  static void delete_field(field *a);   // extension: the reference never frees the B::field of input_r (no such member there)
  static void delete_vector_Fr(vector_Fr *a);
  static void delete_vector_G1(vector_G1 *a);
  static void delete_vector_G2(vector_G2 *a);
  static void delete_groth16_input(groth16_input *a);
  static void delete_groth16_params(groth16_params *a);
  static void delete_evaluation_domain(evaluation_domain *a);

  static void groth16_output_write(G1 *A, G2 *B, G1 *C, const char *output_path);

  // ---- extensions (not in the reference wrapper) -------------------------------------------------
  // the whole of compute_H<B> in one device-resident call (overwrites ca, cb, cc like the reference)
  static vector_Fr *compute_H_fused(evaluation_domain *domain, vector_Fr *ca, vector_Fr *cb, vector_Fr *cc);
  // C = Ht + Lt + r Bt1 (cuda_prover_piecewise.cu:79-90: three multiexps, B::G1_scale, two B::G1_add) as ONE multi-scalar
  // multiplication over the concatenated base set H | L | B1 with the scalars coefficients_for_H | w_L | r w.  The same group
  // element, hence the same proof bytes.  read_params builds the concatenated set (and B1 / L / H as sets of their own only when
  // params_B1 / params_L / params_H are first asked for) unless fuse_C(false) was called before it or MNT753_FUSED_C=0 is set.
  static G1 *groth16_C(groth16_params *params, vector_Fr *coefficients_for_H, vector_Fr *w_L, vector_Fr *w, field *r);
  static void fuse_C(bool on);
  // Number of GPUs of this node the parameters are sharded over (call before init_public_params; default 1, or the value of
  // the environment variable MNT753_GPUS).  Every vector_G1 / vector_G2 is cut into contiguous slices, one per device, exactly
  // as libff cuts an MSM over OpenMP threads (multiexp.tcc:417-431); multiexp_G1 / multiexp_G2 run the slices concurrently
  // and fold the partial results in rank order (multiexp.tcc:433-438).  The FFTs stay on device 0.
  static void use_devices(int n);
  // Where the partial points of a sharded multiexp meet: false (default) -- on the host, where mnt753_msm_finish leaves them anyway;
  // true -- through an RCCL all-gather over the devices (mnt753_exchange_points, xGMI) before the fold.  Also MNT753_FOLD=rccl.
  static void fold_over_rccl(bool on);
  // true: this process proves ONCE -- what the reference's `./main <curve> compute ...` is (libsnark/main.cpp:196-203 loads the
  // parameters per invocation, :274-293).  read_params then builds the base sets without their window tables and runs no warm-up MSM:
  // a table costs 0.33 s per 2^20 G1 points (1.3 s for the G2 set) and saves 20 ms per MSM, which pays from the second proof of a
  // resident prover (--repeat / --serve / several jobs), never in one.  Same proof bytes.  main_hip sets it for a single job on one
  // device; MNT753_ONE_SHOT=0 / 1 and MNT753_MSM_PRECOMP=0 / 1 override.  Call before read_params.
  static void one_shot(bool on);
  // The step before the hot path: instead of reading ca / cb / cc from the input file (where the
  // reference's generator put them, generate_parameters.cpp:44-57), evaluate the constraint system on the assignment on the
  // device -- the first loop of r1cs_to_qap_witness_map (reductions/r1cs_to_qap/r1cs_to_qap.tcc:223-237).
  // read_r1cs: the file written by oracle/ref_groth16.cpp (u64 num_inputs, m, nc; per matrix a, b, c: u64 row_ptr[nc + 1],
  // u32 col[nnz], Fr coeff[nnz]).  read_witness: a file holding w[m + 1] and r only; the returned groth16_input behaves like
  // one from read_input.
  static r1cs *read_r1cs(const char *path);
  static groth16_input *read_witness(const char *path, groth16_params *params, r1cs *cs);
  static void delete_r1cs(r1cs *a);
  // seconds the background loader of read_input needed until the whole input file was on the device (waits for it)
  static double input_load_seconds(groth16_input *input);
  // raw access for tests / tools
  static const uint64_t *G1_words(const G1 *a);
  static const uint64_t *G2_words(const G2 *a);
};

using mnt4753_hip = mnt753_hip_impl<0>;
using mnt6753_hip = mnt753_hip_impl<1>;

extern template class mnt753_hip_impl<0>;
extern template class mnt753_hip_impl<1>;
