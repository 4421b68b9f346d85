/* mnt753_hip.h -- C ABI of libmnt753_hip.so: the MI355X (gfx950) implementation of the Groth16 prover
 * hot path of MinaProtocol/snark-challenge-prover-reference (MSM over G1/G2 and radix-2 FFT over Fr of
 * MNT4753 / MNT6753).
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).  The C++ class pair mnt4753_hip / mnt6753_hip in
 * include/prover_hip_functions.hpp mirrors libsnark/prover_reference_include/prover_reference_functions.hpp
 * member for member on top of this ABI; INTEGRATION.md shows the reference-side binding.
 *
 * Data formats (identical to the reference's files, libsnark/serialization.hpp:22-121):
 *   Fr / Fq element : 12 little-endian uint64 limbs, Montgomery form with R = 2^768, fully reduced.
 *   G1 affine       : x | y                      (24 u64);  y == 0 encodes the identity.
 *   G2 affine       : x.c0 | x.c1 [| x.c2] | y.c0 | y.c1 [| y.c2]   (48 u64 MNT4753, 72 u64 MNT6753).
 *   projective      : X | Y | Z  (homogeneous projective, identity = (0 : 1 : 0)) -- the in-memory form
 *                     of libff::mnt4753_G1 (depends/libff/libff/algebra/curves/mnt753/mnt4753/mnt4753_g1.hpp).
 *
 * Conventions: every function returns 0 on success and a negative MNT753_E* code on failure;
 * mnt753_last_error() returns a message for the calling thread's last failure.  Pointers named dev_*
 * are HIP device pointers (e.g. torch tensor.data_ptr(), or mnt753_dev_alloc); all others are host
 * pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).  One context per
 * process; calls are serialised by the caller (same threading contract as the reference wrapper).
 * There is NO CPU fallback: without a HIP device every compute entry point returns MNT753_ENODEV.
 */
#ifndef MNT753_HIP_H
#define MNT753_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNT753_CURVE_MNT4753 0
#define MNT753_CURVE_MNT6753 1
#define MNT753_G1 1
#define MNT753_G2 2

#define MNT753_OK 0
#define MNT753_EINVAL (-1)   /* bad argument (null pointer, size, curve / group id) */
#define MNT753_ENODEV (-2)   /* no HIP device / library not initialised */
#define MNT753_EHIP (-3)     /* a HIP runtime call failed */
#define MNT753_ENOMEM (-4)   /* device or host allocation failed */
#define MNT753_EDOMAIN (-5)  /* FFT size is not a supported power of two for this field */
#define MNT753_ESELFTEST (-6) /* mnt753_self_test: a known-answer check failed -- this build must not be used to prove */

/* FFT kinds: libfqfft basic_radix2_domain::{FFT,iFFT,cosetFFT,icosetFFT}
 * (depends/libfqfft/libfqfft/evaluation_domain/domains/basic_radix2_domain.tcc:62-96);
 * the coset shift is Fr::multiplicative_generator as in libsnark/prover_reference_functions.cpp:227-238. */
#define MNT753_FFT 0
#define MNT753_IFFT 1
#define MNT753_COSET_FFT 2
#define MNT753_ICOSET_FFT 3

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* replaces B::init_public_params (prover_reference_functions.hpp:25).  device = HIP device ordinal. */
int mnt753_init(int device);
/* Several GPUs in one process: logical devices 0 .. n-1 = HIP devices 0 .. n-1.  A base set, a domain
 * or a buffer lives on the device that is current for the calling thread when it is created (mnt753_set_device; device 0
 * after initialisation, also in new threads); entry points that take a base set run on that set's device whatever the
 * current one is.  The reference shards an MSM over OpenMP threads as contiguous slices and sums the partial results
 * serially (depends/libff/libff/algebra/scalar_multiplication/multiexp.tcc:417-440); the wrapper classes do the same over
 * devices: one base set per device and slice; every device streams its own slice of the witness from the input file
 * (mnt753_load_file_to_device on that device), only the slices of coefficients_for_H travel device 0 -> device g
 * (mnt753_copy_peer_async, xGMI); one projective point back per device, folded on the host in rank order with mnt753_point_add.
 * MNT753_SHARE_DEVICE=1 (development) maps the logical devices onto however many are visible. */
int mnt753_init_devices(int n_devices);
/* Known-answer self-test of THIS build on the current device: expected words computed by the reference's own code (libff's
 * multi_exp_with_mixed_addition, libfqfft's compute_H call sequence, libff's group classes) are embedded in the library as constants
 * (csrc/mnt753_selftest_data.h, minted by tools/gen_selftest_data.py through oracle/_ref -- data, none of the oracle's code).
 * level 0: the host tails (point decoding, addition, doubling, affine output) on one libff record per group; level 1: plus a 256-point
 * MSM per (curve, group) -- with and without the batched-affine levels of the large sets in front of it -- and compute_H on a 2^8
 * domain per curve (~0.15 s for both curves, most of it the allocations of the small base sets); level 2: plus the same MSMs over a window table (~0.35 s).  0 if everything agrees, MNT753_ESELFTEST (and
 * a message naming the check) if one word differs, another code if a call failed.  Why: a proof is only as sound as the build that
 * computed it, and the compiler has miscompiled kernels of this library before; the reference carries a check of the same
 * purpose around its prover (libsnark/main.cpp:295-343).  B::init_public_params runs level 1 for its curve once per process (MNT753_SELFTEST=0
 * skips it, MNT753_SELFTEST=2 raises it); `main_hip <curve> self-test` runs level 2. */
int mnt753_self_test(int level);
/* the same for one curve (MNT753_CURVE_MNT4753 / MNT753_CURVE_MNT6753): what B::init_public_params of that curve's class runs */
int mnt753_self_test_curve(int curve, int level);
int mnt753_device_count(void);
int mnt753_set_device(int logical_device);
int mnt753_get_device(void);   /* the calling thread's current logical device (0 before mnt753_set_device) */
/* Peer access: after the call the copy engines and kernels of `device` may address the memory of `peer` directly
 * (hipDeviceEnablePeerAccess), so that a copy between the two goes over their xGMI link instead of staging through host memory.
 * *how (may be NULL) = MNT753_PEER_SAME (both logical devices are one GPU: MNT753_SHARE_DEVICE), MNT753_PEER_DIRECT, or
 * MNT753_PEER_STAGED (the platform offers no peer access for the pair; copies still work, through the host).  Idempotent.
 * mnt753_init_devices asks for every ordered pair; the wrapper asks again and says under MNT753_TRACE=1 what it got.  The one-GPU
 * reference has nothing to compare (cuda_prover_piecewise.cu:24-34 keeps ca / cb / cc on one device); this is what the sharding of
 * SURVEY.md section 8e adds.  A mnt753_copy_peer_async is a command of the DESTINATION's default queue: the destination's copy
 * engine (ROCr's SDMA; a blit kernel on its CUs where SDMA is turned off) reads the source's memory over the link. */
enum { MNT753_PEER_SAME = 0, MNT753_PEER_DIRECT = 1, MNT753_PEER_STAGED = 2 };
int mnt753_enable_peer_access(int device, int peer, int* how);
int mnt753_copy_peer(int dst_device, void* dev_dst, int src_device, const void* dev_src, size_t bytes);
/* The same without blocking the host: the copy is enqueued on dst_device's default stream and starts after everything enqueued so
 * far on src_device's default stream (an event recorded there, waited for on the destination) -- how the slices of
 * coefficients_for_H leave device 0 behind compute_H (cuda_prover_piecewise.cu:79-81) while the host goes on enqueueing.  An MSM
 * started afterwards on dst_device with stream == NULL (mnt753_msm_start) is ordered behind the copy. */
int mnt753_copy_peer_async(int dst_device, void* dev_dst, int src_device, const void* dev_src, size_t bytes);
/* The exchange of a sharded MSM over RCCL ("Collective"): every logical device contributes `words` u64 -- its
 * partial points, multiexp.tcc:417-431 lifted to devices --, host_in[g] being device g's block in host memory (where mnt753_msm_finish
 * put it); afterwards host_out holds the n_devices blocks in rank order, ready for the serial fold of multiexp.tcc:433-438
 * (mnt753_point_add).  A single-process communicator over the devices of mnt753_init_devices (ncclCommInitAll), one ncclAllGather per
 * device in a group call, xGMI between the GPUs; librccl is loaded on first use.  MNT753_ENODEV if librccl cannot be loaded or the
 * logical devices are not distinct GPUs (MNT753_SHARE_DEVICE).  The wrapper classes fold on the host by default -- a partial point
 * arrives there anyway -- and take this path with MNT753_FOLD=rccl (main_hip --fold rccl). */
int mnt753_exchange_points(const uint64_t* const* host_in, size_t words, uint64_t* host_out);
double mnt753_exchange_last_us(void);   /* duration of the last exchange: staging, collective, copy back */
const char* mnt753_last_error(void);
/* number of words (uint64) of one element / point of the given kind */
size_t mnt753_affine_words(int curve, int group);      /* 24, 48 (MNT4753 G2) or 72 (MNT6753 G2) */
size_t mnt753_projective_words(int curve, int group);  /* 36, 72 or 108 */

/* ---- device memory helpers (for hosts that do not bring their own allocator) ------------------- */
int mnt753_dev_alloc(void** dev_ptr, size_t bytes);
int mnt753_dev_free(void* dev_ptr);
int mnt753_copy_h2d(void* dev_dst, const void* src, size_t bytes);
int mnt753_copy_d2h(void* dst, const void* dev_src, size_t bytes);
int mnt753_copy_d2d(void* dev_dst, const void* dev_src, size_t bytes);
int mnt753_dev_memset(void* dev_dst, int value, size_t bytes);
int mnt753_sync(void* stream);
/* free / total memory of the calling thread's current device (hipMemGetInfo): the wrapper prints what a parameter set occupies
 * (window tables, workspaces, the pooled buffers of the batched-affine levels) under MNT753_TRACE_LOAD=1 */
int mnt753_dev_mem_info(size_t* free_bytes, size_t* total_bytes);
/* Stream `bytes` of file `path` starting at `file_offset` into device memory (double-buffered pinned staging, reads
 * overlap the H2D copies); blocks the calling thread until the data is on the device.  Thread-safe: the wrapper's input
 * loader calls it from a background thread while the main thread launches kernels (replaces the 6.3 M fread calls of
 * the reference's groth16_input constructor, prover_reference_functions.cpp:48-76). */
int mnt753_load_file_to_device(const char* path, size_t file_offset, size_t bytes, void* dev_dst);

/* ---- MSM --------------------------------------------------------------------------------------
 * A base set is the device-resident, pre-converted image of a vector_G1 / vector_G2 of the parameters
 * (B::params_A/B1/L/H/B2, prover_reference_functions.hpp:66-70).  The reference loads parameters before
 * its timing window opens (libsnark/main.cpp:201-203); creating a base set belongs to that phase. */
typedef struct mnt753_bases mnt753_bases;
/* affine: n points in the wire format above; on_device != 0 if `affine` is a device pointer */
int mnt753_bases_create(int curve, int group, const uint64_t* affine, int on_device, size_t n, mnt753_bases** out);
int mnt753_bases_free(mnt753_bases* b);
size_t mnt753_bases_size(const mnt753_bases* b);

/* replaces B::multiexp_G1 / B::multiexp_G2 (prover_reference_functions.hpp:49-52):
 *   result = sum_{i < n} scalars[i] * bases[base_offset + i]
 * scalars: n Fr elements, wire format (Montgomery), device pointer if scalars_on_device != 0.
 * out_projective: host buffer of mnt753_projective_words() u64 -- X | Y | Z, Montgomery, fully reduced
 * (NOT normalised to Z = 1; feed it to mnt753_point_to_affine / mnt753_point_add). */
int mnt753_msm(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n,
               uint64_t* out_projective, void* stream);
/* The same, asynchronous: _start enqueues every kernel of the MSM and returns; _finish waits for it and writes the
 * result.  stream == NULL uses a non-blocking stream owned by the base set, ordered after all work already enqueued on
 * the default stream -- so the five MSMs of one proof (A, B1, B2, L, H: independent, cuda_prover_piecewise.cu:71-81) run
 * concurrently and the latency-bound reduction tail of one overlaps the bucket accumulation of another.  At most one
 * MSM may be in flight per base set. */
int mnt753_msm_start(mnt753_bases* b, size_t base_offset, const uint64_t* scalars, int scalars_on_device, size_t n, void* stream);
int mnt753_msm_finish(mnt753_bases* b, uint64_t* out_projective);
/* Order the throughput phases of two MSMs that are in flight together: the point kernels (pairing levels, accumulate: the ones that
 * fill the chip) of b's NEXT mnt753_msm_start begin when those of `first`'s latest mnt753_msm_start have ended; b's sort still runs
 * as early as it can.  Without it the two interleave, finish together, and both latency-bound tails (edge merge, bucket reduction:
 * 1.3 - 3 ms of a 2^20-point MSM during which the chip idles) end the prove; with it first's tail runs under b's point kernels and
 * only b's is left -- so the MSM with the shortest tail goes last (A behind C in the prove: cuda_prover_piecewise.cu:71-81 starts
 * them in an order that does not matter there).  Worth 1 - 3 % for the small sets (MNT6753 2^15: 15.7 -> 15.2 and 15.3 -> 15.2 ms per proof in two series); two
 * 2^20-point MSMs gain more from filling each other's kernel ends interleaved (measured 157.2 -> 158.4 ms ordered), so the wrapper
 * orders only below 2^18 points.  Both sets on one device; call it after first's mnt753_msm_start.  The order refers to first's
 * latest start AT THE TIME b starts; it is dropped (no wait) if `first` is freed before b's next start. */
int mnt753_msm_order_after(mnt753_bases* b, const mnt753_bases* first);

/* Window tables of the base sets created FROM NOW ON: mode 1 (default) -- a set of 4096 points or more gets the table of its window
 * multiples 2^(cw) P_i at creation (0.33 s of kernels per 2^20 G1 points, 1.3 s for 2^20 G2 points; an MSM over it then takes
 * 23.7 ms instead of 38-43); mode 0 -- no tables and no batched-affine levels (whose buffers are tens of GB to allocate): what a
 * process that will prove ONCE wants, because the reference's CLI is such a
 * process (libsnark/main.cpp:196-203 loads the parameters per invocation, :274-293) and a table never pays for itself in one proof.
 * The wrapper's B::one_shot / main_hip's default for a single job use it.  MNT753_MSM_PRECOMP=0 / 1 in the environment overrides
 * both.  Returns the previous mode. */
int mnt753_msm_set_window_table(int mode);
/* window size override (0 = automatic); returns previous value.  Tuning knob, not part of the reference. */
int mnt753_msm_set_window_bits(int c);
/* time of the last mnt753_msm call's kernels in milliseconds (HIP events on the launch stream):
 * index 0 = total, 1 = digits+sort, 2 = bucket accumulation kernel, 3 = bucket reduction, 4 = host tail */
int mnt753_msm_last_timing(float out_ms[5]);
/* plan of the last mnt753_msm call: [0] window bits c, [1] windows W, [2] 1 if the precomputed window table was used,
 * [3] sorted entries per accumulate lane T */
int mnt753_msm_last_plan(int out[4]);
/* Levels of the batched-affine pairing pass the last G1 MSM ran before its accumulate kernel (0 = none). */
int mnt753_msm_last_pair_levels(void);
/* irregular pairing levels the last MSM ran behind the regular ones (csrc/msm_kernels.hip.h, "irregular levels") */
int mnt753_msm_last_irr_levels(void);

/* ---- small group operations on the host (O(1) work per proof) ------------------------------------ */
/* replaces B::G1_add (hpp:32) */
int mnt753_point_add(int curve, int group, const uint64_t* a_proj, const uint64_t* b_proj, uint64_t* out_proj);
/* replaces B::G1_scale (hpp:33): scalar is an Fr element in wire (Montgomery) form */
int mnt753_point_scale(int curve, int group, const uint64_t* scalar, const uint64_t* p_proj, uint64_t* out_proj);
/* to_affine_coordinates + write_g1/write_g2 encoding (serialization.hpp:44-67): identity -> all zero */
int mnt753_point_to_affine(int curve, int group, const uint64_t* p_proj, uint64_t* out_affine);
/* read_g1 / read_g2 decoding (serialization.hpp:84-111): y == 0 -> identity */
int mnt753_point_from_affine(int curve, int group, const uint64_t* affine, uint64_t* out_proj);

/* ---- FFT over Fr ---------------------------------------------------------------------------------- */
typedef struct mnt753_domain mnt753_domain;
/* replaces B::get_evaluation_domain (hpp:30): m must be a power of two <= 2^s (s = 30 MNT4753, 15 MNT6753) */
int mnt753_domain_create(int curve, size_t m, mnt753_domain** out);
int mnt753_domain_free(mnt753_domain* d);
size_t mnt753_domain_size(const mnt753_domain* d);   /* B::domain_get_m (hpp:47) */
/* replaces B::domain_iFFT / domain_cosetFFT / domain_icosetFFT (hpp:42-45) and libfqfft FFT; in place on
 * a device vector of m Fr elements in wire format.  The transforms of one domain share its work buffer: calls on different
 * streams are ordered on the device one after the other (an event per domain); host threads must not enter the same domain
 * at the same time. */
int mnt753_fft(mnt753_domain* d, int kind, uint64_t* dev_vec, void* stream);
/* replaces B::domain_divide_by_Z_on_coset (hpp:46) */
int mnt753_divide_by_z_on_coset(mnt753_domain* d, uint64_t* dev_vec, void* stream);
/* replaces B::vector_Fr_muleq / vector_Fr_subeq (hpp:35-36): a[i] = a[i] (*|-) b[i], i < n */
int mnt753_vec_muleq(int curve, uint64_t* dev_a, const uint64_t* dev_b, size_t n, void* stream);
int mnt753_vec_subeq(int curve, uint64_t* dev_a, const uint64_t* dev_b, size_t n, void* stream);
/* dst[i] = src[i] * k, i < n, k one Fr element in HOST memory (wire format); dst may be src.  Used to fold the factor r of
 * r * Bt1 (cuda_prover_piecewise.cu:85-87: B::G1_scale(B::input_r(inputs), evaluation_Bt1)) into the scalars of that sum, so that
 * Ht + Lt + r Bt1 becomes one multi-scalar multiplication (B::groth16_C, include/prover_hip_functions.hpp). */
int mnt753_vec_scale(int curve, uint64_t* dev_dst, const uint64_t* dev_src, const uint64_t* host_scalar, size_t n, void* stream);
/* the whole of compute_H (cuda_prover_piecewise.cu:18-53) resident on the device:
 * dev_ca / dev_cb / dev_cc hold m elements each and are overwritten; dev_h receives m + 1 elements
 * (coefficients_for_H, last entry 0). */
int mnt753_compute_h(mnt753_domain* d, uint64_t* dev_ca, uint64_t* dev_cb, uint64_t* dev_cc, uint64_t* dev_h,
                     void* stream);
/* The two halves of compute_H, for a prover that spreads it over the devices of a node (task parallelism, SURVEY.md section 8e: the
 * transform is not sharded, but ca, cb and cc are independent until the pointwise step, cuda_prover_piecewise.cu:24-34):
 *   _chain : x <- cosetFFT(iFFT(x)) in place on one vector of m elements (:24-27 for ca / cb, :33-34 for cc);
 *   _finish: a <- icosetFFT((a * b - c) / Z), h <- a | 0 (:29-47); b and c are only read.
 * A domain (its tables and work buffer) lives on the device that was current when it was created; both calls run there whatever the
 * calling thread's current device is, and every vector handed to them must be on that device (mnt753_copy_peer_async brings the
 * chained cb / cc from the devices that transformed them).  mnt753_compute_h == three _chain + one _finish on one device. */
int mnt753_compute_h_chain(mnt753_domain* d, uint64_t* dev_vec, void* stream);
int mnt753_compute_h_finish(mnt753_domain* d, uint64_t* dev_a, const uint64_t* dev_b, const uint64_t* dev_c, uint64_t* dev_h, void* stream);
int mnt753_domain_device(const mnt753_domain* d);   /* logical device the domain lives on */

/* ---- witness-map front end (the step before the hot path) -------------------------------------------
 * The reference evaluates the constraint system on the assignment on the CPU before the prover starts
 * (libsnark/generate_parameters.cpp:44-57 writes the result into the input file as ca / cb / cc; it is the first loop of
 * r1cs_to_qap_witness_map, libsnark/reductions/r1cs_to_qap/r1cs_to_qap.tcc:223-237).  Here the constraint system lives on the
 * device as three CSR matrices (a, b, c) over the variables (1, w_1 .. w_m) -- column 0 is the constant one, i.e. w[0] of the
 * input file -- and the evaluation is one kernel:
 *   ca[i] = <a_i, w>, cb[i] = <b_i, w>, cc[i] = <c_i, w> for i < nc;  ca[nc + i] = w[i] for i <= num_inputs;  zero above.
 * row_ptr[k]: nc + 1 offsets (row_ptr[k][0] = 0), col[k]: variable indices 0 .. m, coeff[k]: Fr elements in wire form;
 * host pointers.  out_len = the evaluation domain size (d + 1 >= nc + num_inputs + 1). */
typedef struct mnt753_r1cs mnt753_r1cs;
int mnt753_r1cs_create(int curve, uint64_t num_inputs, uint64_t m, uint64_t nc, const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                       const uint64_t* const coeff[3], mnt753_r1cs** out);
int mnt753_r1cs_free(mnt753_r1cs* r);
size_t mnt753_r1cs_domain_size(const mnt753_r1cs* r);   /* nc + num_inputs + 1 */
size_t mnt753_r1cs_num_variables(const mnt753_r1cs* r); /* m: dev_w of mnt753_r1cs_evaluate must hold m + 1 elements */
size_t mnt753_r1cs_num_inputs(const mnt753_r1cs* r);
int mnt753_r1cs_evaluate(mnt753_r1cs* r, const uint64_t* dev_w, uint64_t* dev_ca, uint64_t* dev_cb, uint64_t* dev_cc, size_t out_len, void* stream);

/* ---- uniform scalars (host) ---------------------------------------------------------------------------
 * n elements of Fr of the curve, uniform in [0, r) by rejection on 753-bit draws of a seeded SplitMix64, in wire form.  What the
 * product needs them for: the scalars of the warm-up MSMs at parameter-load time (B::read_params) and the prover's second random
 * element s of `main_hip complete` when no file names one (main.cpp:312-319 draws it with libff's random_element). */
int mnt753_synth_scalars(int curve, uint64_t seed, size_t n, uint64_t* out_scalars);

/* The synthetic base points with known discrete logarithms and the device-level test hooks are NOT part of this library: they live
 * in libmnt753_hip_test.so (include/mnt753_hip_test.h), which tests/, bench.py and __graft_entry__.smoke() load beside it. */

#ifdef __cplusplus
}
#endif
#endif /* MNT753_HIP_H */
