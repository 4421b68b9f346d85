// ./main_hip <curve> compute <params> <input> <output> [<input2> <output2> ...] [--repeat N] [--gpus N] [--fold rccl|host] [--unfused-h] [--unfused-c] [--ref-order] [--quiet]
//            --serve: keep the parameters resident and prove further "<input> <output>" pairs read from stdin, one per line
// ./main_hip <curve> compute-r1cs <params> <r1cs> <witness> <output> ...      (ca / cb / cc evaluated on the device from the constraint system)
// ./main_hip <curve> complete <keys> <input|witness> <challenge_proof> <full_proof> [--s-file <Fr> | --s-seed N]
//
// The prover driver, same command line as the reference binaries (libsnark/main.cpp:274-293,
// cuda_prover_piecewise.cu:100-120).  compute_H<B> and run_prover<B> keep the reference's shape -- they are
// written against the wrapper type B only -- and are instantiated with the MI355X classes.
// Timing prints follow libsnark/main.cpp:201-270: the window "Total time from input to output" opens after
// the parameters are loaded and contains input load, compute and output write.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mnt753_hip.h"
#include "../../include/prover_hip_functions.hpp"

// Defaults = the fastest schedule on one MI355X: the G2 MSM is enqueued as soon as w is on the device, compute_H (one fused
// call) follows as soon as ca / cb / cc are, then the G1 MSMs.  The point kernels occupy every SIMD with long-lived
// workgroups, so the ~60 short NTT kernels of compute_H must not be enqueued behind the G1 MSMs (measured: 0.24 s with this
// order, 0.44-1.1 s with the reference's source order), and C = Ht + Lt + r Bt1 is one MSM over the concatenated base set
// (B::groth16_C).  --ref-order keeps the call sequence of cuda_prover_piecewise.cu:64-90 (five multiexps, compute_H after the
// first three, G1_scale, two G1_add -- the wrapper still runs the last three multiexps and that tail as one MSM, see LazyPoint in
// prover_hip_functions.cpp), --unfused-c loads the parameters as five base sets and runs the five multiexps, --unfused-h the
// reference's sequence of B:: calls inside compute_H (cuda_prover_piecewise.cu:24-47); every combination writes the same bytes.
static bool g_fused_h = true;
static bool g_quiet = false;
static bool g_h_first = true;
static bool g_c_first = true;    // fused C: its MSM (3x the points of A's) is enqueued ahead of A's; --c-last the other way round
static bool g_explicit_c = true;   // call B::groth16_C; off (--ref-order): the reference's calls, which the wrapper fuses by itself when the parameters are fused
static bool g_touch_all = false;   // --touch-all (with --ref-order): read Bt1, Lt, Ht before they are combined -- with fused parameters that forces the three separate MSMs
static bool g_fused_c = true;   // C = Ht + Lt + r Bt1 as one MSM over H | L | B1 (B::groth16_C); --unfused-c / --ref-order: the reference's five multiexps
static int g_gpus = 0;   // --gpus N: parameter vectors sharded over N devices of this node (0: MNT753_GPUS or 1)
static bool g_serve = false;    // --serve: after the listed jobs, read further "<input> <output>" lines from stdin until EOF
static int g_one_shot = -1;     // -1: decided from the job list (one job, no --serve, one device: a one-proof process); --one-shot / --tables force it
static bool g_peer_bench = false;   // --peer-bench (with --gpus N): time the peer copies the sharded prover makes, 100 MB each, before proving
static int g_fold_rccl = -1;    // --fold rccl | host: where the partial points of a sharded multiexp meet (default: MNT753_FOLD, else host)

typedef std::chrono::steady_clock clk;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
static void ck(int rc, const char* what) { if (rc) throw std::runtime_error(std::string(what) + ": " + mnt753_last_error()); }

// cuda_prover_piecewise.cu:18-53.  Overwrites ca (and cb, cc), like the reference.
template <typename B>
typename B::vector_Fr* compute_H(size_t d, typename B::vector_Fr* ca, typename B::vector_Fr* cb, typename B::vector_Fr* cc) {
  auto domain = B::get_evaluation_domain(d + 1);
  if (g_fused_h) {
    auto H_res = B::compute_H_fused(domain, ca, cb, cc);
    B::delete_evaluation_domain(domain);
    return H_res;
  }
  B::domain_iFFT(domain, ca);
  B::domain_iFFT(domain, cb);
  B::domain_cosetFFT(domain, ca);
  B::domain_cosetFFT(domain, cb);
  auto H_tmp = ca;  // ca stores H
  size_t m = B::domain_get_m(domain);
  B::vector_Fr_muleq(H_tmp, cb, m);
  B::domain_iFFT(domain, cc);
  B::domain_cosetFFT(domain, cc);
  B::vector_Fr_subeq(H_tmp, cc, m);
  B::domain_divide_by_Z_on_coset(domain, H_tmp);
  B::domain_icosetFFT(domain, H_tmp);
  typename B::vector_Fr* H_res = B::vector_Fr_zeros(m + 1);
  B::vector_Fr_copy_into(H_tmp, H_res, m);
  B::delete_evaluation_domain(domain);
  return H_res;
}

// cuda_prover_piecewise.cu:55-98, split at the line where the reference opens its timing window (libsnark/main.cpp:201-203):
// the parameters are loaded once and stay resident (window tables, workspaces, evaluation domain); every (input, output)
// pair after that is one proof.  With a single pair this is exactly the reference's run_prover.
template <typename B>
void prove_one(typename B::groth16_params* params, const char* input_path, const char* output_path, clk::time_point t0, bool first,
               typename B::r1cs* cs = nullptr) {
  const size_t primary_input_size = 1;
  auto t_main = clk::now();
  // compute-r1cs: the input file holds w and r only; ca / cb / cc come from the constraint system, evaluated on the device
  auto input = cs ? B::read_witness(input_path, params, cs) : B::read_input(input_path, params);
  auto t_in = clk::now();
  if (!g_quiet) printf("load inputs (started in the background): %.3fs\n", secs(t_main, t_in));

  auto w = B::input_w(input);
  auto ca = B::input_ca(input), cb = B::input_cb(input), cc = B::input_cc(input);
  auto pA = B::params_A(params); auto pB2 = B::params_B2(params);
  // Same operations as cuda_prover_piecewise.cu:64-81, in an order that follows the data: B::read_input streams the
  // file in the background (w first), B::multiexp_* only enqueue work, so the G2 MSM -- the longest, and it needs nothing
  // but w -- starts as soon as w is on the device, while ca / cb / cc are still loading.
  typename B::G2* evaluation_Bt2 = B::multiexp_G2(w, pB2, B::params_m(params) + 1);
  typename B::G1 *evaluation_At, *C;
  auto w_off = B::vector_Fr_offset(w, primary_input_size + 1);
  typename B::vector_Fr* coefficients_for_H;
  auto r = B::input_r(input);
  auto t_h = clk::now();
  clk::time_point t_msm, t_c;
  if (g_fused_c && g_explicit_c) {
    // Ht + Lt + r Bt1 is one group element: one MSM over the concatenated base set H | L | B1 with the scalars h | w_off | r w
    // (B::groth16_C) instead of three MSMs, a scalar multiplication and two additions -- what an MSM pays per call and per bucket
    // is paid once.  It needs coefficients_for_H, so compute_H goes first, right behind the G2 MSM.
    coefficients_for_H = compute_H<B>(B::params_d(params), ca, cb, cc);
    t_h = clk::now();
    if (g_c_first) C = B::groth16_C(params, coefficients_for_H, w_off, w, r);
    evaluation_At = B::multiexp_G1(w, pA, B::params_m(params) + 1);
    if (!g_c_first) C = B::groth16_C(params, coefficients_for_H, w_off, w, r);
    // touching a result waits for its MSM and runs the host tail of its bucket reduction (c - 1 doublings and additions: 0.33 ms for
    // an Fq3 point).  In the order the device finishes them -- the G2 MSM was enqueued first, A's point kernels are ordered behind
    // C's on the small sets -- so that every tail but the last runs under the kernels still in flight (profiles/r05/mnt6753_prove_timeline.txt)
    (void)B::G2_words(evaluation_Bt2); (void)B::G1_words(C); (void)B::G1_words(evaluation_At);
    t_msm = t_c = clk::now();
  } else {
    auto pB1 = B::params_B1(params); auto pH = B::params_H(params); auto pL = B::params_L(params);
    typename B::G1 *evaluation_Bt1, *evaluation_Lt, *evaluation_Ht;
    if (g_h_first) {
      // compute_H right behind the G2 MSM: its short kernels are not starved by four G1 accumulation phases, and the H MSM
      // overlaps the others instead of running alone at the end
      coefficients_for_H = compute_H<B>(B::params_d(params), ca, cb, cc);
      t_h = clk::now();
      evaluation_Ht = B::multiexp_G1(coefficients_for_H, pH, B::params_d(params));
      evaluation_At = B::multiexp_G1(w, pA, B::params_m(params) + 1);
      evaluation_Bt1 = B::multiexp_G1(w, pB1, B::params_m(params) + 1);
      evaluation_Lt = B::multiexp_G1(w_off, pL, B::params_m(params) - 1);
    } else {
      evaluation_At = B::multiexp_G1(w, pA, B::params_m(params) + 1);
      evaluation_Bt1 = B::multiexp_G1(w, pB1, B::params_m(params) + 1);
      evaluation_Lt = B::multiexp_G1(w_off, pL, B::params_m(params) - 1);
      coefficients_for_H = compute_H<B>(B::params_d(params), ca, cb, cc);
      t_h = clk::now();
      evaluation_Ht = B::multiexp_G1(coefficients_for_H, pH, B::params_d(params));
    }
    // the five MSMs run concurrently on their base sets' streams; touching the results waits for them.  (With fused parameters
    // Bt1, Lt and Ht are not computed at all: the wrapper recognises Ht + (Lt + r Bt1) below and runs it as one MSM -- touching
    // them here would force the three separate ones.)
    (void)B::G1_words(evaluation_At); (void)B::G2_words(evaluation_Bt2);
    if (!g_fused_c || g_touch_all) { (void)B::G1_words(evaluation_Bt1); (void)B::G1_words(evaluation_Ht); (void)B::G1_words(evaluation_Lt); }
    t_msm = clk::now();
    auto scaled_Bt1 = B::G1_scale(r, evaluation_Bt1);
    auto Lt1_plus_scaled_Bt1 = B::G1_add(evaluation_Lt, scaled_Bt1);
    C = B::G1_add(evaluation_Ht, Lt1_plus_scaled_Bt1);
    (void)B::G1_words(C);
    t_c = clk::now();
    B::delete_G1(evaluation_Bt1); B::delete_G1(evaluation_Ht); B::delete_G1(evaluation_Lt);
    B::delete_G1(scaled_Bt1); B::delete_G1(Lt1_plus_scaled_Bt1);
    B::delete_vector_G1(pB1); B::delete_vector_G1(pH); B::delete_vector_G1(pL);
  }
  B::groth16_output_write(evaluation_At, evaluation_Bt2, C, output_path);
  auto t_out = clk::now();
  if (!g_quiet) {
    printf("G2 MSM enqueued + compute_H (input streaming in): %.3fs\nremaining MSM time: %.3fs\nC = Ht + Lt + r*Bt1: %.3fs%s\ngpu: %.4fs\nstore: %.3fs\n", secs(t_in, t_h), secs(t_h, t_msm),
           secs(t_msm, t_c), g_fused_c ? " (one MSM over H | L | B1)" : "", secs(t_in, t_c), secs(t_c, t_out));
    printf("input file on the device after: %.3fs (background loader)\n", B::input_load_seconds(input));
    printf("Total time from input to output: %.4fs\n", secs(t_main, t_out));
    if (first) printf("Total wall (incl. load params): %.3fs\n", secs(t0, t_out));
  }

  B::delete_G1(evaluation_At); B::delete_G2(evaluation_Bt2); B::delete_G1(C);
  B::delete_vector_Fr(coefficients_for_H); B::delete_vector_Fr(w); B::delete_vector_Fr(w_off);
  B::delete_vector_Fr(ca); B::delete_vector_Fr(cb); B::delete_vector_Fr(cc);
  B::delete_vector_G1(pA); B::delete_vector_G2(pB2);
  B::delete_field(r);   // (the reference's wrapper has no delete_field and its driver leaks the element)
  B::delete_groth16_input(input);
}

// --peer-bench: what DESIGN.md section 5 ASSUMES (2.0 ms for 100 MB over one xGMI link), measured on the box the prover runs on: the
// copies of the sharded prove -- the transformed cb / cc 1 -> 0 and 2 -> 0 (cuda_prover_piecewise.cu:24-34 keeps them on one device;
// here they cross), a slice of coefficients_for_H 0 -> g -- as one 100 MB mnt753_copy_peer_async each, best of three, with what the
// platform granted for the pair.  One line per copy on stdout; bench.py --gpus N carries them.
static void peer_bench() {
  const int n = mnt753_device_count();
  if (n < 2) { printf("peer copy: one device, nothing to measure\n"); return; }
  const size_t bytes = (size_t)100 << 20;
  std::vector<void*> buf(n, nullptr);
  for (int g = 0; g < n; ++g) { ck(mnt753_set_device(g), "mnt753_set_device"); ck(mnt753_dev_alloc(&buf[g], bytes), "mnt753_dev_alloc"); ck(mnt753_dev_memset(buf[g], g, bytes), "mnt753_dev_memset"); }
  std::vector<std::pair<int, int>> pairs;   // (source, destination)
  pairs.emplace_back(1, 0);
  if (n > 2) pairs.emplace_back(2, 0);
  for (int g = 1; g < n; ++g) pairs.emplace_back(0, g);
  static const char* const names[] = {"same GPU (logical devices share it)", "direct (peer access)", "staged through host memory"};
  for (auto pr : pairs) {
    int how = MNT753_PEER_STAGED;
    ck(mnt753_enable_peer_access(pr.second, pr.first, &how), "mnt753_enable_peer_access");
    double best = 1e30;
    for (int k = 0; k < 4; ++k) {     // the first pass warms the path
      ck(mnt753_set_device(pr.first), "mnt753_set_device"); ck(mnt753_sync(nullptr), "mnt753_sync");
      ck(mnt753_set_device(pr.second), "mnt753_set_device"); ck(mnt753_sync(nullptr), "mnt753_sync");
      const auto t0 = clk::now();
      ck(mnt753_copy_peer_async(pr.second, buf[pr.second], pr.first, buf[pr.first], bytes), "mnt753_copy_peer_async");
      ck(mnt753_set_device(pr.second), "mnt753_set_device"); ck(mnt753_sync(nullptr), "mnt753_sync");
      const double ms = 1e3 * secs(t0, clk::now());
      if (k > 0 && ms < best) best = ms;
    }
    printf("peer copy 100 MB device %d -> %d: %.3f ms (%.1f GB/s), %s\n", pr.first, pr.second, best, bytes / best / 1e6, names[how >= 0 && how <= 2 ? how : 2]);
  }
  for (int g = 0; g < n; ++g) { ck(mnt753_set_device(g), "mnt753_set_device"); ck(mnt753_dev_free(buf[g]), "mnt753_dev_free"); }
  ck(mnt753_set_device(0), "mnt753_set_device");
}

// jobs: (input, output) pairs; all proved against the same resident parameters
template <typename B>
void run_prover(const char* params_path, const std::vector<std::pair<std::string, std::string>>& jobs, const char* r1cs_path = nullptr) {
  if (g_gpus > 0) B::use_devices(g_gpus);
  B::fuse_C(g_fused_c);
  if (g_fold_rccl >= 0) B::fold_over_rccl(g_fold_rccl != 0);
  // The reference's CLI is a one-shot process: parameters loaded per invocation, one proof, exit (libsnark/main.cpp:196-203, :274-293).
  // Invoked the same way -- one job, no --repeat / --serve, one device -- this prover builds no window tables and runs no warm-up MSM:
  // 2.7 s of table kernels and 0.7 s of warm-up would buy 0.1 s on the one proof.  A resident prover (several jobs, --repeat, --serve)
  // and a sharded one (--gpus) keep them.  --tables / --one-shot force either; MNT753_MSM_PRECOMP stays the override underneath.
  {
    const char* eg = getenv("MNT753_GPUS");
    const bool several_devices = g_gpus > 1 || (g_gpus == 0 && eg && atoi(eg) > 1);
    const bool one = g_one_shot >= 0 ? g_one_shot != 0 : (jobs.size() == 1 && !g_serve && !several_devices);
    B::one_shot(one);
    if (!g_quiet && one) printf("one-shot prover: no window tables (one job, no --repeat / --serve; --tables builds them)\n");
  }
  B::init_public_params();
  if (g_peer_bench) peer_bench();
  auto t0 = clk::now();
  auto params = B::read_params(params_path);
  typename B::r1cs* cs = r1cs_path ? B::read_r1cs(r1cs_path) : nullptr;
  auto t_params = clk::now();
  if (!g_quiet) printf("load params: %.3fs\n", secs(t0, t_params));
  bool first = true;
  for (const auto& job : jobs) {
    if (!g_quiet && jobs.size() > 1) printf("-- proof %s -> %s\n", job.first.c_str(), job.second.c_str());
    prove_one<B>(params, job.first.c_str(), job.second.c_str(), t0, first, cs);
    first = false;
  }
  if (g_serve) {
    // Resident job feed: the parameters (window tables, workspaces, evaluation domain: ~6 s and ~54 GB of HBM to build for the 2^20
    // set) are paid once per HOST process, every job after that costs its prove time.  One job per line on stdin,
    // "<input path> <output path>"; after each, one line "proved <output path> <seconds>" (or "failed <output path>: <reason>") on
    // stdout, flushed, so a driver can pipeline requests.  EOF ends the service.
    char line[8192];
    while (fgets(line, sizeof(line), stdin)) {
      char in_path[4096], out_path[4096];
      if (sscanf(line, "%4095s %4095s", in_path, out_path) != 2) continue;
      const auto tj = clk::now();
      try {
        prove_one<B>(params, in_path, out_path, t0, false, cs);
        printf("proved %s %.3f\n", out_path, secs(tj, clk::now()));
      } catch (const std::exception& e) {
        printf("failed %s: %s\n", out_path, e.what());
      }
      fflush(stdout);
    }
  }
  if (cs) B::delete_r1cs(cs);
  B::delete_groth16_params(params);
}

// ./main_hip <curve> complete <keys> <input|witness> <challenge_proof> <full_proof> [--s-file <Fr> | --s-seed N]
//
// The step AFTER the hot path: the challenge prover stops at (A, B, C) = (sum w_i A_i, sum w_i B_i,
// Ht + Lt + r Bt1); libsnark/main.cpp:312-319 shows how the reference completes that to a Groth16 proof a verifier accepts:
//     A' = alpha + A + r delta,    B' = beta + B + s delta,    C' = C + s A' + r beta          (r from the input file, s fresh)
// keys = alpha_g1 | beta_g1 | beta_g2 | delta_g1 | delta_g2 in the wire format (oracle/ref_groth16.cpp mints them from the
// reference's generator).  O(1) group operations on the host through the C ABI; no GPU needed.
static void slurp(const char* path, long offset_from_end, void* dst, size_t bytes) {
  FILE* f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  if (offset_from_end && fseek(f, -offset_from_end, SEEK_END) != 0) { fclose(f); throw std::runtime_error(std::string("seek failed: ") + path); }
  if (fread(dst, 1, bytes, f) != bytes) { fclose(f); throw std::runtime_error(std::string("short read: ") + path); }
  fclose(f);
}
static int complete_proof(int curve, const char* keys_path, const char* input_path, const char* challenge_path, const char* out_path,
                          const char* s_file, uint64_t s_seed) {
  const size_t g1 = 24, g2 = mnt753_affine_words(curve, MNT753_G2);
  std::vector<uint64_t> keys(3 * g1 + 2 * g2), proof(2 * g1 + g2);
  slurp(keys_path, 0, keys.data(), keys.size() * 8);
  slurp(challenge_path, 0, proof.data(), proof.size() * 8);
  uint64_t r[12], s[12];
  slurp(input_path, 96, r, 96);
  if (s_file) slurp(s_file, 0, s, 96); else ck(mnt753_synth_scalars(curve, s_seed, 1, s), "mnt753_synth_scalars");
  const uint64_t *alpha1 = keys.data(), *beta1 = alpha1 + g1, *beta2 = beta1 + g1, *delta1 = beta2 + g2, *delta2 = delta1 + g1;
  auto lift = [&](int group, const uint64_t* aff, uint64_t* proj) { ck(mnt753_point_from_affine(curve, group, aff, proj), "mnt753_point_from_affine"); };
  uint64_t pa[36], pb[108], pc[36], t1[36], t2[108], u1[36], u2[108];
  // A' = alpha + A + r delta
  lift(MNT753_G1, proof.data(), pa); lift(MNT753_G1, alpha1, t1);
  ck(mnt753_point_add(curve, MNT753_G1, t1, pa, pa), "mnt753_point_add");
  lift(MNT753_G1, delta1, t1); ck(mnt753_point_scale(curve, MNT753_G1, r, t1, u1), "mnt753_point_scale");
  ck(mnt753_point_add(curve, MNT753_G1, pa, u1, pa), "mnt753_point_add");
  // B' = beta + B + s delta
  lift(MNT753_G2, proof.data() + g1, pb); lift(MNT753_G2, beta2, t2);
  ck(mnt753_point_add(curve, MNT753_G2, t2, pb, pb), "mnt753_point_add");
  lift(MNT753_G2, delta2, t2); ck(mnt753_point_scale(curve, MNT753_G2, s, t2, u2), "mnt753_point_scale");
  ck(mnt753_point_add(curve, MNT753_G2, pb, u2, pb), "mnt753_point_add");
  // C' = C + s A' + r beta
  lift(MNT753_G1, proof.data() + g1 + g2, pc);
  ck(mnt753_point_scale(curve, MNT753_G1, s, pa, u1), "mnt753_point_scale");
  ck(mnt753_point_add(curve, MNT753_G1, pc, u1, pc), "mnt753_point_add");
  lift(MNT753_G1, beta1, t1); ck(mnt753_point_scale(curve, MNT753_G1, r, t1, u1), "mnt753_point_scale");
  ck(mnt753_point_add(curve, MNT753_G1, pc, u1, pc), "mnt753_point_add");
  std::vector<uint64_t> out(2 * g1 + g2);
  ck(mnt753_point_to_affine(curve, MNT753_G1, pa, out.data()), "mnt753_point_to_affine");
  ck(mnt753_point_to_affine(curve, MNT753_G2, pb, out.data() + g1), "mnt753_point_to_affine");
  ck(mnt753_point_to_affine(curve, MNT753_G1, pc, out.data() + g1 + g2), "mnt753_point_to_affine");
  FILE* f = fopen(out_path, "wb");
  if (!f) throw std::runtime_error(std::string("cannot open output file ") + out_path);
  fwrite(out.data(), 8, out.size(), f);
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  setbuf(stdout, NULL);
  if (argc >= 7 && !strcmp(argv[2], "complete")) {
    const int curve = !strcmp(argv[1], "MNT4753") ? 0 : (!strcmp(argv[1], "MNT6753") ? 1 : -1);
    if (curve < 0) { fprintf(stderr, "unknown curve %s\n", argv[1]); return 2; }
    const char* s_file = nullptr; uint64_t s_seed = 0x73656564ull;
    for (int i = 7; i + 1 < argc; i += 2) {
      if (!strcmp(argv[i], "--s-file")) s_file = argv[i + 1];
      else if (!strcmp(argv[i], "--s-seed")) s_seed = strtoull(argv[i + 1], nullptr, 0);
    }
    try { return complete_proof(curve, argv[3], argv[4], argv[5], argv[6], s_file, s_seed); }
    catch (const std::exception& e) { fprintf(stderr, "main_hip: %s\n", e.what()); return 1; }
  }
  if (argc >= 3 && !strcmp(argv[2], "self-test")) {
    // `main_hip <curve> self-test`: the known-answer checks of this build at their widest (level 2: also over window tables)
    if (mnt753_init(0) != 0) { fprintf(stderr, "main_hip: %s\n", mnt753_last_error()); return 1; }
    const int rc = mnt753_self_test(2);
    if (rc != 0) { fprintf(stderr, "main_hip: %s\n", mnt753_last_error()); return 1; }
    printf("self-test: all known answers of the reference agree (level 2)\n");
    return 0;
  }
  if (argc < 6) {
    fprintf(stderr, "usage: %s MNT4753|MNT6753 compute <params> <input> <output> [<input2> <output2> ...] [--repeat N] [--serve] [--gpus N] [--tables | --one-shot] [--unfused-h] [--unfused-c] [--ref-order] [--fold rccl|host] [--quiet]\n"
                    "  further (input, output) pairs and --repeat prove against the parameters that are already resident on the GPU\n", argv[0]);
    return 2;
  }
  // compute-r1cs <params> <r1cs> <witness> <output>: one more positional argument than compute
  const bool with_r1cs = !strcmp(argv[2], "compute-r1cs");
  if (with_r1cs && argc < 7) { fprintf(stderr, "usage: %s <curve> compute-r1cs <params> <r1cs> <witness> <output>\n", argv[0]); return 2; }
  const int a0 = with_r1cs ? 5 : 4;
  std::vector<std::pair<std::string, std::string>> jobs;
  jobs.emplace_back(argv[a0], argv[a0 + 1]);
  int repeat = 1;
  for (int i = a0 + 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--repeat") && i + 1 < argc) { repeat = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { g_gpus = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--fold") && i + 1 < argc) { g_fold_rccl = !strcmp(argv[++i], "rccl") ? 1 : 0; continue; }
    if (argv[i][0] != '-' && i + 1 < argc && argv[i + 1][0] != '-') { jobs.emplace_back(argv[i], argv[i + 1]); ++i; continue; }
    if (!strcmp(argv[i], "--fused-h")) g_fused_h = true;
    else if (!strcmp(argv[i], "--unfused-h")) g_fused_h = false;
    else if (!strcmp(argv[i], "--quiet")) g_quiet = true;
    else if (!strcmp(argv[i], "--serve")) g_serve = true;
    else if (!strcmp(argv[i], "--h-first")) g_h_first = true;
    else if (!strcmp(argv[i], "--h-last")) g_h_first = false;
    else if (!strcmp(argv[i], "--ref-order")) { g_h_first = false; g_explicit_c = false; }   // the call sequence of cuda_prover_piecewise.cu:64-90
    else if (!strcmp(argv[i], "--unfused-c")) g_fused_c = false;
    else if (!strcmp(argv[i], "--c-last")) g_c_first = false;
    else if (!strcmp(argv[i], "--touch-all")) g_touch_all = true;
    else if (!strcmp(argv[i], "--fused-c")) g_fused_c = true;
    else if (!strcmp(argv[i], "--peer-bench")) g_peer_bench = true;
    else if (!strcmp(argv[i], "--tables")) g_one_shot = 0;      // build the window tables even for a single proof
    else if (!strcmp(argv[i], "--one-shot")) g_one_shot = 1;    // no window tables, no warm-up MSM, whatever the job list
    else {
      // an option this prover does not have (or one that lost its argument), or an input without its output: refuse, do not guess
      fprintf(stderr, argv[i][0] == '-' ? "main_hip: unknown option %s\n" : "main_hip: input %s without an output path\n", argv[i]);
      return 2;
    }
  }
  std::string curve(argv[1]), mode(argv[2]);
  try {
    if (mode != "compute" && mode != "compute-r1cs") { fprintf(stderr, "unknown mode %s\n", argv[2]); return 2; }
    if (repeat > 1) { const auto one = jobs; for (int k = 1; k < repeat; ++k) jobs.insert(jobs.end(), one.begin(), one.end()); }
    const char* r1cs_path = with_r1cs ? argv[4] : nullptr;
    if (curve == "MNT4753") run_prover<mnt4753_hip>(argv[3], jobs, r1cs_path);
    else if (curve == "MNT6753") run_prover<mnt6753_hip>(argv[3], jobs, r1cs_path);
    else { fprintf(stderr, "unknown curve %s\n", argv[1]); return 2; }
  } catch (const std::exception& e) {
    fprintf(stderr, "main_hip: %s\n", e.what());
    return 1;
  }
  return 0;
}
