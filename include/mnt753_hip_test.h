/* libmnt753_hip_test.so -- TEST INFRASTRUCTURE beside the product library (libmnt753_hip.so, include/mnt753_hip.h), which it links
 * against: the synthetic base points with known discrete logarithms that stand in for libsnark/generate_parameters.cpp on a box
 * with neither the reference nor its parameter files, and the device-level known-answer hooks under the MSM.  Loaded by tests/,
 * bench.py and __graft_entry__.smoke() only; main_hip and the B:: wrapper never see it (readelf -d: they need libmnt753_hip.so
 * alone). */
#ifndef MNT753_HIP_TEST_H
#define MNT753_HIP_TEST_H
#include "mnt753_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- deterministic synthetic inputs (host) ---------------------------------------------------------
 * Stand-in for libsnark/generate_parameters.cpp on machines that have neither the reference nor its
 * parameter files: bases with known discrete logarithms base[k] = e_k * G, uniform scalars, and the exact
 * value of sum scalars[k] * base[k] computed from the e_k (one scalar multiplication) for parity checks
 * at sizes no CPU implementation finishes quickly. */
int mnt753_synth_points(int curve, int group, uint64_t seed, size_t n, uint64_t* out_affine, int threads);
int mnt753_synth_expected_msm(int curve, int group, uint64_t seed, size_t n, const uint64_t* scalars, uint64_t* out_projective);

/* ---- test hooks (tests/ only; not used by the prover) -------------------------------------------------
 * The device field layer element-wise on n pairs of Fp elements in wire form (host pointers in and out), for known-answer
 * tests against libff's Fp_model (depends/libff/libff/algebra/fields/fp.tcc:161-186 mul_reduce, :405-417 +=, :491-508 -=,
 * :641-685 invert, :227-238 as_bigint).  mod: 0 = modulus A (Fr of MNT4753 / Fq of MNT6753), 1 = modulus B.
 * op: 0 a*b, 1 a+b, 2 a-b, 3 a^-1 (0 -> 0), 4 as_bigint(a), 5 -a, 6 a^2 (dedicated squaring), 7 wire->device->wire,
 * 8 a*b + a*a (fused two-product multiplier), 9 13*a (small-constant multiplier). */
int mnt753_test_field_op(int mod, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* The coordinate field of G2 on the device, element-wise on n pairs of elements in wire form (c0 | c1 [| c2], host pointers):
 * Fq2 = Fq[u]/(u^2 - 13) on MNT4753 (depends/libff/libff/algebra/fields/fp2.tcc:79-90 mul, :118-126 squared, :129-142 inverse,
 * :58-70 + and -), Fq3 = Fq[u]/(u^3 - 11) on MNT6753 (fp3.tcc:83-96, :107-123, :126-143, :59-74).  split: 0 = the one-lane form
 * (Karatsuba through one multiplier instance), 1 = the lane-split form the G2 point kernels run (two / three lanes per element,
 * fused multi-product multipliers, ds_bpermute exchange).
 * op: 0 a*b, 1 a*a, 2 a^-1, 3 a+b, 4 a-b, 5 -a, 6 (a == b) as the element 1 or 0 (the zero test the kernels branch on). */
int mnt753_test_ext_op(int curve, int split, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Every form of the group law the MSM kernels contain, on n pairs of points.  p_proj / q_proj / out_proj: projective X | Y | Z in
 * wire form (any representative; Z == 0 is the identity), host pointers.  group: MNT753_G1 / MNT753_G2; split (G2 only): 0 = one
 * lane per point, 1 = the lane-split configuration.  Reference: operator+ / mixed_add / dbl of mnt4753_G1 (depends/libff/libff/
 * algebra/curves/mnt753/mnt4753/mnt4753_g1.cpp:134-207, :265-313, :315-346), mnt4753_G2 (mnt4753_g2.cpp:150-223, :281-329,
 * :331-362), mnt6753_G1 (mnt6753_g1.cpp), mnt6753_G2 (mnt6753_g2.cpp:156-229, :287-335, :337-368).
 * op: 0 P + Q through the point VM (bucket reduction, edge merge);  1 2P through the VM (window table; equal points);
 *     2 P + Q with Q affine (Q's Z is taken as 1 unless 0) through the VM's mixed addition (bucket accumulation);
 *     3 the same as straight-line code (bucket accumulation of the base fields and the two-lane Fq2; other fields: as op 2);
 *     4 P + Q with two point-lanes per addition (narrow steps of the bucket reduction, edge merge of Fq3);
 *     5 P + Q as straight-line code (wide steps of the bucket reduction, base fields; other fields: as op 0);
 *     6 P + Q with one GROUP of lanes per addition (8 for the base fields, 16 for Fq2 / Fq3: the narrowest steps of the bucket
 *       reduction and the levels of the edge merge of a short lane list; split = 0 only). */
int mnt753_test_point_op(int curve, int group, int split, int op, const uint64_t* p_proj, const uint64_t* q_proj, size_t n, uint64_t* out_proj);

#ifdef __cplusplus
}
#endif
#endif /* MNT753_HIP_TEST_H */
