/* mnt753_oracle.h -- interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see mnt753_oracle.c). */
#ifndef MNT753_ORACLE_H
#define MNT753_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* all element / point arguments use the reference's wire format (12 x u64 Montgomery R = 2^768; points affine,
 * y == 0 is the identity).  curve: 0 = MNT4753, 1 = MNT6753; group: 1 = G1, 2 = G2; mod: 0 = A, 1 = B. */
int oracle_field_op(int mod, int op, const uint64_t* a, const uint64_t* b, uint64_t* out); /* op 0 mul,1 add,2 sub,3 inv,4 as_bigint,5 neg */
int oracle_ext_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out); /* Fq2 / Fq3 of G2: op 0 mul,1 sqr,2 inv,3 add,4 sub,5 neg */
int oracle_point_op(int curve, int group, int op, const uint64_t* p, const uint64_t* q, uint64_t* out); /* op 0 add,1 dbl,2 sub,3 scalar(q)*p */
int oracle_msm(int curve, int group, const uint64_t* bases, const uint64_t* scalars, size_t n, size_t chunks, uint64_t* out_affine);
int oracle_fft(int curve, int kind, uint64_t* vec, size_t m);   /* kind 0 FFT, 1 iFFT, 2 cosetFFT, 3 icosetFFT */
int oracle_divide_by_z_on_coset(int curve, uint64_t* vec, size_t m);
int oracle_compute_h(int curve, uint64_t* ca, uint64_t* cb, uint64_t* cc, uint64_t* h, size_t m);
int oracle_prove(int curve, const char* params_path, const char* input_path, const char* output_path, size_t chunks, double* timings4);
int oracle_r1cs_evaluate(int curve, uint64_t num_inputs, uint64_t nc, const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                         const uint64_t* const coeff[3], const uint64_t* w, uint64_t* ca, uint64_t* cb, uint64_t* cc, size_t out_len);
int oracle_complete_proof(int curve, const uint64_t* keys, const uint64_t* proof, const uint64_t* r, const uint64_t* s, uint64_t* out);
int oracle_max_threads(void);
#ifdef __cplusplus
}
#endif
#endif
