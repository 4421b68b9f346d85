/* mnt753_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's CPU algorithms for the Groth16 prover hot path
 * (MinaProtocol/snark-challenge-prover-reference).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or run this file; the product (libmnt753_hip.so) never
 * links or calls it.  Parity status: PINNED -- checked against golden vectors minted by the reference's
 * own libff / libfqfft / libsnark code (tests/golden/, made by oracle/mint_golden.cpp linked against the
 * reference sources; see tests/test_oracle_golden.py), and against the reference binaries in oracle/_ref
 * when they are present.
 *
 * Every function cites the reference code it follows (paths relative to the reference tree):
 *   F  = depends/libff/libff/algebra/fields
 *   C  = depends/libff/libff/algebra/curves/mnt753
 *   SM = depends/libff/libff/algebra/scalar_multiplication
 *   Q  = depends/libfqfft/libfqfft/evaluation_domain
 *   S  = libsnark
 * GMP (mpn_*) calls of the reference are restated with unsigned __int128 limb arithmetic; they are exact
 * integer primitives, so the results are identical by construction.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "mnt753_oracle.h"
#include "mnt753_oracle_constants.h"

typedef unsigned __int128 u128;
#define NLIMB 12

/* ------------------------------------------------------------------------------------------------
 * bigint helpers (F/bigint.tcc)
 * ---------------------------------------------------------------------------------------------- */
static int big_cmp(const uint64_t* a, const uint64_t* b, int n) { /* mpn_cmp */
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return -1;
  }
  return 0;
}
static uint64_t big_add(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) { /* mpn_add_n */
  u128 c = 0;
  for (int i = 0; i < n; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t big_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) { /* mpn_sub_n */
  uint64_t bw = 0;
  for (int i = 0; i < n; ++i) {
    u128 d = (u128)a[i] - b[i] - bw;
    r[i] = (uint64_t)d;
    bw = (uint64_t)(d >> 64) & 1;
  }
  return bw;
}
static int big_is_zero(const uint64_t* a, int n) { uint64_t o = 0; for (int i = 0; i < n; ++i) o |= a[i]; return o == 0; }
/* bigint<n>::test_bit, F/bigint.tcc:149-163 */
static int big_test_bit(const uint64_t* a, size_t bitno) {
  if (bitno >= 64 * NLIMB) return 0;
  return (int)((a[bitno >> 6] >> (bitno & 63)) & 1);
}
/* bigint<n>::num_bits, F/bigint.tcc:102-129 */
static size_t big_num_bits(const uint64_t* a) {
  for (int i = NLIMB - 1; i >= 0; --i) {
    if (a[i]) return (size_t)(64 * i + 64 - __builtin_clzll(a[i]));
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Fp_model<12, modulus>  (F/fp.tcc).  mod = 0: modulus A, 1: modulus B (constants header).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t l[NLIMB]; } fp_t;

static const uint64_t* MODP(int mod) { return ORACLE_MOD[mod]; }

/* Fp_model::mul_reduce generic path, F/fp.tcc:161-186: full product, then 12 rounds
 * k = inv * res[i]; res += k * modulus << (64 i); one conditional subtraction. */
static void fp_mul(fp_t* r, const fp_t* a, const fp_t* b, int mod) {
  uint64_t res[2 * NLIMB + 1];
  memset(res, 0, sizeof(res));
  for (int i = 0; i < NLIMB; ++i) { /* mpn_mul_n */
    u128 c = 0;
    for (int j = 0; j < NLIMB; ++j) {
      c += (u128)a->l[j] * b->l[i] + res[i + j];
      res[i + j] = (uint64_t)c;
      c >>= 64;
    }
    res[i + NLIMB] = (uint64_t)c;
  }
  const uint64_t* p = MODP(mod);
  for (int i = 0; i < NLIMB; ++i) {
    uint64_t k = ORACLE_INV[mod] * res[i];
    u128 c = 0;
    for (int j = 0; j < NLIMB; ++j) { /* mpn_addmul_1 */
      c += (u128)k * p[j] + res[i + j];
      res[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int j = i + NLIMB; j < 2 * NLIMB + 1 && c; ++j) { /* mpn_add_1 */
      c += res[j];
      res[j] = (uint64_t)c;
      c >>= 64;
    }
  }
  if (res[2 * NLIMB] || big_cmp(res + NLIMB, p, NLIMB) >= 0) big_sub(res + NLIMB, res + NLIMB, p, NLIMB);
  memcpy(r->l, res + NLIMB, sizeof(r->l));
}
/* operator+=, F/fp.tcc:405-417 */
static void fp_add(fp_t* r, const fp_t* a, const fp_t* b, int mod) {
  uint64_t t[NLIMB];
  uint64_t carry = big_add(t, a->l, b->l, NLIMB);
  if (carry || big_cmp(t, MODP(mod), NLIMB) >= 0) big_sub(t, t, MODP(mod), NLIMB);
  memcpy(r->l, t, sizeof(t));
}
/* operator-=, F/fp.tcc:491-508 */
static void fp_sub(fp_t* r, const fp_t* a, const fp_t* b, int mod) {
  uint64_t t[NLIMB];
  if (big_cmp(a->l, b->l, NLIMB) < 0) {
    uint64_t s[NLIMB + 1];
    s[NLIMB] = big_add(s, a->l, MODP(mod), NLIMB);
    uint64_t bb[NLIMB + 1];
    memcpy(bb, b->l, sizeof(b->l));
    bb[NLIMB] = 0;
    uint64_t d[NLIMB + 1];
    big_sub(d, s, bb, NLIMB + 1);
    memcpy(t, d, sizeof(t));
  } else {
    big_sub(t, a->l, b->l, NLIMB);
  }
  memcpy(r->l, t, sizeof(t));
}
static int fp_is_zero(const fp_t* a) { return big_is_zero(a->l, NLIMB); }
static int fp_eq(const fp_t* a, const fp_t* b) { return memcmp(a->l, b->l, sizeof(a->l)) == 0; }
static void fp_set_zero(fp_t* r) { memset(r->l, 0, sizeof(r->l)); }
static void fp_set_one(fp_t* r, int mod) { memcpy(r->l, ORACLE_ONE[mod], sizeof(r->l)); } /* R mod p */
/* unary minus, F/fp.tcc:574-591 */
static void fp_neg(fp_t* r, const fp_t* a, int mod) {
  if (fp_is_zero(a)) { *r = *a; return; }
  big_sub(r->l, MODP(mod), a->l, NLIMB);
}
/* as_bigint, F/fp.tcc:227-238: multiply by the integer 1 (leaves Montgomery form) */
static void fp_as_bigint(uint64_t out[NLIMB], const fp_t* a, int mod) {
  fp_t one, r;
  fp_set_zero(&one);
  one.l[0] = 1;
  fp_mul(&r, a, &one, mod);
  memcpy(out, r.l, sizeof(r.l));
}
/* Fp_model(long) constructor: x * R via mul_reduce(Rsquared), F/fp.tcc:196-215 */
static void fp_from_u64(fp_t* r, uint64_t v, int mod) {
  fp_t t, r2;
  fp_set_zero(&t);
  t.l[0] = v;
  memcpy(r2.l, ORACLE_R2[mod], sizeof(r2.l));
  fp_mul(r, &t, &r2, mod);
}
/* power with a plain bigint exponent: libff power(), F/field_utils / exponentiation.tcc */
static void fp_pow(fp_t* r, const fp_t* a, const uint64_t* e, int nlimbs, int mod) {
  fp_t res;
  fp_set_one(&res, mod);
  int found_one = 0;
  for (long i = (long)nlimbs * 64 - 1; i >= 0; --i) {
    if (found_one) fp_mul(&res, &res, &res, mod);
    if ((e[i >> 6] >> (i & 63)) & 1) {
      found_one = 1;
      fp_mul(&res, &res, a, mod);
    }
  }
  *r = res;
}
/* invert, F/fp.tcc:641-685.  The reference runs mpn_gcdext on the Montgomery residue and multiplies by
 * R^3; the inverse is unique, so it is restated as a^(p-2) (Fermat). */
static void fp_inv(fp_t* r, const fp_t* a, int mod) {
  uint64_t e[NLIMB];
  memcpy(e, MODP(mod), sizeof(e));
  e[0] -= 2;
  fp_pow(r, a, e, NLIMB, mod);
}

/* ------------------------------------------------------------------------------------------------
 * Extension fields: Fp2_model (F/fp2.tcc), Fp3_model (F/fp3.tcc), and Fp itself as degree 1.
 * An element is up to three base-field coefficients c0, c1, c2 (order of all_base_field_elements,
 * F/fp2.tcc:18-21, F/fp3.tcc:18-21, which is also the serialisation order).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { fp_t c[3]; } fe_t;
typedef struct {
  int mod;       /* base-field modulus index */
  int deg;       /* 1, 2 or 3 */
  uint64_t nr;   /* non_residue: 13 (Fq2 of MNT4753, C/mnt4753/mnt4753_init.cpp:105), 11 (Fq3 of MNT6753, C/mnt6753/mnt6753_init.cpp:109) */
} field_t;

static void fe_add(fe_t* r, const fe_t* a, const fe_t* b, const field_t* f) { for (int k = 0; k < f->deg; ++k) fp_add(&r->c[k], &a->c[k], &b->c[k], f->mod); }
static void fe_sub(fe_t* r, const fe_t* a, const fe_t* b, const field_t* f) { for (int k = 0; k < f->deg; ++k) fp_sub(&r->c[k], &a->c[k], &b->c[k], f->mod); }
static void fe_neg(fe_t* r, const fe_t* a, const field_t* f) { for (int k = 0; k < f->deg; ++k) fp_neg(&r->c[k], &a->c[k], f->mod); }
static int fe_is_zero(const fe_t* a, const field_t* f) { for (int k = 0; k < f->deg; ++k) if (!fp_is_zero(&a->c[k])) return 0; return 1; }
static int fe_eq(const fe_t* a, const fe_t* b, const field_t* f) { for (int k = 0; k < f->deg; ++k) if (!fp_eq(&a->c[k], &b->c[k])) return 0; return 1; }
static void fe_set_zero(fe_t* r) { memset(r, 0, sizeof(*r)); }
static void fe_set_one(fe_t* r, const field_t* f) { fe_set_zero(r); fp_set_one(&r->c[0], f->mod); }
static void fp_mul_nr(fp_t* r, const fp_t* a, const field_t* f) { fp_t n; fp_from_u64(&n, f->nr, f->mod); fp_mul(r, &n, a, f->mod); }

static void fe_mul(fe_t* r, const fe_t* x, const fe_t* y, const field_t* f) {
  const int m = f->mod;
  if (f->deg == 1) { fp_mul(&r->c[0], &x->c[0], &y->c[0], m); return; }
  if (f->deg == 2) {
    /* Fp2 Karatsuba, F/fp2.tcc:79-90: (aA + nr bB, (a+b)(A+B) - aA - bB) */
    fp_t aA, bB, s, t, u;
    fp_mul(&aA, &x->c[0], &y->c[0], m);
    fp_mul(&bB, &x->c[1], &y->c[1], m);
    fp_add(&s, &x->c[0], &x->c[1], m);
    fp_add(&t, &y->c[0], &y->c[1], m);
    fp_mul(&u, &s, &t, m);
    fp_sub(&u, &u, &aA, m);
    fp_sub(&u, &u, &bB, m);
    fp_mul_nr(&s, &bB, f);
    fp_add(&r->c[0], &aA, &s, m);
    r->c[1] = u;
    return;
  }
  /* Fp3 Karatsuba, F/fp3.tcc:83-96 */
  fp_t aA, bB, cC, s, t, u, c0, c1, c2;
  const fp_t *a = &x->c[0], *b = &x->c[1], *c = &x->c[2], *A = &y->c[0], *B = &y->c[1], *Cc = &y->c[2];
  fp_mul(&aA, a, A, m); fp_mul(&bB, b, B, m); fp_mul(&cC, c, Cc, m);
  fp_add(&s, b, c, m); fp_add(&t, B, Cc, m); fp_mul(&u, &s, &t, m); fp_sub(&u, &u, &bB, m); fp_sub(&u, &u, &cC, m);
  fp_mul_nr(&u, &u, f); fp_add(&c0, &aA, &u, m);                                   /* aA + nr((b+c)(B+C) - bB - cC) */
  fp_add(&s, a, b, m); fp_add(&t, A, B, m); fp_mul(&u, &s, &t, m); fp_sub(&u, &u, &aA, m); fp_sub(&u, &u, &bB, m);
  fp_mul_nr(&s, &cC, f); fp_add(&c1, &u, &s, m);                                    /* (a+b)(A+B) - aA - bB + nr cC */
  fp_add(&s, a, c, m); fp_add(&t, A, Cc, m); fp_mul(&u, &s, &t, m); fp_sub(&u, &u, &aA, m); fp_add(&u, &u, &bB, m);
  fp_sub(&c2, &u, &cC, m);                                                          /* (a+c)(A+C) - aA + bB - cC */
  r->c[0] = c0; r->c[1] = c1; r->c[2] = c2;
}
/* squared(): Fp -> mul (F/fp.tcc:635-638); Fp2 -> squared_complex (F/fp2.tcc:118-126);
 * Fp3 -> CH-SQR2 (F/fp3.tcc:107-123) */
static void fe_sqr(fe_t* r, const fe_t* x, const field_t* f) {
  const int m = f->mod;
  if (f->deg == 1) { fp_mul(&r->c[0], &x->c[0], &x->c[0], m); return; }
  if (f->deg == 2) {
    fp_t ab, s, t, u, nb, nab;
    const fp_t *a = &x->c[0], *b = &x->c[1];
    fp_mul(&ab, a, b, m);
    fp_add(&s, a, b, m);
    fp_mul_nr(&nb, b, f);
    fp_add(&t, a, &nb, m);
    fp_mul(&u, &s, &t, m);
    fp_sub(&u, &u, &ab, m);
    fp_mul_nr(&nab, &ab, f);
    fp_sub(&r->c[0], &u, &nab, m);     /* (a+b)(a+nr b) - ab - nr ab */
    fp_add(&r->c[1], &ab, &ab, m);     /* 2ab */
    return;
  }
  fp_t s0, ab, s1, s2, bc, s3, s4, t, c0, c1, c2;
  const fp_t *a = &x->c[0], *b = &x->c[1], *c = &x->c[2];
  fp_mul(&s0, a, a, m);
  fp_mul(&ab, a, b, m); fp_add(&s1, &ab, &ab, m);
  fp_sub(&t, a, b, m); fp_add(&t, &t, c, m); fp_mul(&s2, &t, &t, m);
  fp_mul(&bc, b, c, m); fp_add(&s3, &bc, &bc, m);
  fp_mul(&s4, c, c, m);
  fp_mul_nr(&t, &s3, f); fp_add(&c0, &s0, &t, m);            /* s0 + nr s3 */
  fp_mul_nr(&t, &s4, f); fp_add(&c1, &s1, &t, m);            /* s1 + nr s4 */
  fp_add(&t, &s1, &s2, m); fp_add(&t, &t, &s3, m); fp_sub(&t, &t, &s0, m); fp_sub(&c2, &t, &s4, m);
  r->c[0] = c0; r->c[1] = c1; r->c[2] = c2;
}
/* inverse: Fp (F/fp.tcc:641-685), Fp2 (F/fp2.tcc:129-142), Fp3 (F/fp3.tcc:126-143) */
static void fe_inv(fe_t* r, const fe_t* x, const field_t* f) {
  const int m = f->mod;
  if (f->deg == 1) { fp_inv(&r->c[0], &x->c[0], m); return; }
  if (f->deg == 2) {
    fp_t t0, t1, t2, t3, nt1;
    const fp_t *a = &x->c[0], *b = &x->c[1];
    fp_mul(&t0, a, a, m); fp_mul(&t1, b, b, m);
    fp_mul_nr(&nt1, &t1, f); fp_sub(&t2, &t0, &nt1, m);      /* t2 = a^2 - nr b^2 */
    fp_inv(&t3, &t2, m);
    fp_mul(&r->c[0], a, &t3, m);
    fp_mul(&t0, b, &t3, m); fp_neg(&r->c[1], &t0, m);
    return;
  }
  fp_t t0, t1, t2, t3, t4, t5, c0, c1, c2, u, v, t6;
  const fp_t *a = &x->c[0], *b = &x->c[1], *c = &x->c[2];
  fp_mul(&t0, a, a, m); fp_mul(&t1, b, b, m); fp_mul(&t2, c, c, m);
  fp_mul(&t3, a, b, m); fp_mul(&t4, a, c, m); fp_mul(&t5, b, c, m);
  fp_mul_nr(&u, &t5, f); fp_sub(&c0, &t0, &u, m);
  fp_mul_nr(&u, &t2, f); fp_sub(&c1, &u, &t3, m);
  fp_sub(&c2, &t1, &t4, m);
  fp_mul(&u, c, &c1, m); fp_mul(&v, b, &c2, m); fp_add(&u, &u, &v, m); fp_mul_nr(&u, &u, f);
  fp_mul(&v, a, &c0, m); fp_add(&u, &u, &v, m);
  fp_inv(&t6, &u, m);
  fp_mul(&r->c[0], &t6, &c0, m); fp_mul(&r->c[1], &t6, &c1, m); fp_mul(&r->c[2], &t6, &c2, m);
}

/* ------------------------------------------------------------------------------------------------
 * Groups: mnt4753_G1 / mnt4753_G2 / mnt6753_G1 / mnt6753_G2 (C/mnt4753/mnt4753_g1.cpp etc.)
 * homogeneous projective coordinates, identity (0 : 1 : 0).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { fe_t X, Y, Z; } pt_t;
typedef struct {
  int curve, group;
  field_t f;     /* coordinate field */
  int fr_mod;    /* scalar field modulus index */
} group_t;

static void group_init(group_t* g, int curve, int group) {
  g->curve = curve; g->group = group;
  if (curve == 0) { /* MNT4753: Fq = B, Fr = A  (C/mnt4753/mnt4753_init.hpp:23-24) */
    g->f.mod = 1; g->fr_mod = 0;
    g->f.deg = group == 1 ? 1 : 2; g->f.nr = 13;
  } else {          /* MNT6753: Fq = A, Fr = B  (C/mnt6753/mnt6753_init.hpp:23-24) */
    g->f.mod = 0; g->fr_mod = 1;
    g->f.deg = group == 1 ? 1 : 3; g->f.nr = 11;
  }
}
static void fe_mul_u64(fe_t* r, const fe_t* a, int k, uint64_t v, const field_t* f) { fp_t c; fp_from_u64(&c, v, f->mod); fp_mul(&r->c[k], &c, &a->c[k], f->mod); }
/* coeff_a * elt for G1 (a = 2, C/mnt4753/mnt4753_init.cpp:119; a = 11, C/mnt6753/mnt6753_init.cpp:130);
 * mnt4753_G2::mul_by_a (C/mnt4753/mnt4753_g2.cpp:31-34, coefficients mnt4753_init.cpp:127-128);
 * mnt6753_G2::mul_by_a (C/mnt6753/mnt6753_g2.cpp:38-41, coefficients mnt6753_init.cpp:140-142) */
static void pt_mul_by_a(fe_t* r, const fe_t* e, const group_t* g) {
  const field_t* f = &g->f;
  fe_t t;
  fe_set_zero(&t);
  if (g->group == 1) { fe_mul_u64(&t, e, 0, g->curve == 0 ? 2 : 11, f); }
  else if (g->curve == 0) { fe_mul_u64(&t, e, 0, 2 * 13, f); fe_mul_u64(&t, e, 1, 2 * 13, f); }
  else {
    fp_t k; const int m = f->mod;
    fp_from_u64(&k, 11 * 11, m); fp_mul(&t.c[0], &k, &e->c[1], m);
    fp_mul(&t.c[1], &k, &e->c[2], m);
    fp_from_u64(&k, 11, m); fp_mul(&t.c[2], &k, &e->c[0], m);
  }
  *r = t;
}
static void pt_set_zero(pt_t* p, const group_t* g) { fe_set_zero(&p->X); fe_set_one(&p->Y, &g->f); fe_set_zero(&p->Z); } /* G1_zero, mnt4753_init.cpp:135-137 */
static int pt_is_zero(const pt_t* p, const group_t* g) { return fe_is_zero(&p->X, &g->f) && fe_is_zero(&p->Z, &g->f); } /* mnt4753_g1.cpp:95-98 */

/* dbl(), mnt4753_g1.cpp:315-346 (dbl-2007-bl) */
static void pt_dbl(pt_t* r, const pt_t* p, const group_t* g) {
  const field_t* f = &g->f;
  if (pt_is_zero(p, g)) { *r = *p; return; }
  fe_t XX, ZZ, w, Y1Z1, s, ss, sss, R, RR, B, h, t, u;
  fe_sqr(&XX, &p->X, f);
  fe_sqr(&ZZ, &p->Z, f);
  pt_mul_by_a(&w, &ZZ, g); fe_add(&t, &XX, &XX, f); fe_add(&t, &t, &XX, f); fe_add(&w, &w, &t, f);
  fe_mul(&Y1Z1, &p->Y, &p->Z, f);
  fe_add(&s, &Y1Z1, &Y1Z1, f);
  fe_sqr(&ss, &s, f);
  fe_mul(&sss, &s, &ss, f);
  fe_mul(&R, &p->Y, &s, f);
  fe_sqr(&RR, &R, f);
  fe_add(&t, &p->X, &R, f); fe_sqr(&B, &t, f); fe_sub(&B, &B, &XX, f); fe_sub(&B, &B, &RR, f);
  fe_sqr(&h, &w, f); fe_add(&t, &B, &B, f); fe_sub(&h, &h, &t, f);
  fe_mul(&r->X, &h, &s, f);
  fe_sub(&t, &B, &h, f); fe_mul(&u, &w, &t, f); fe_add(&t, &RR, &RR, f); fe_sub(&r->Y, &u, &t, f);
  r->Z = sss;
}
/* operator+, mnt4753_g1.cpp:134-207 (add-1998-cmo-2 with the doubling check) */
static void pt_add(pt_t* r, const pt_t* p, const pt_t* q, const group_t* g) {
  const field_t* f = &g->f;
  if (pt_is_zero(p, g)) { *r = *q; return; }
  if (pt_is_zero(q, g)) { *r = *p; return; }
  fe_t X1Z2, X2Z1, Y1Z2, Y2Z1;
  fe_mul(&X1Z2, &p->X, &q->Z, f);
  fe_mul(&X2Z1, &p->Z, &q->X, f);
  fe_mul(&Y1Z2, &p->Y, &q->Z, f);
  fe_mul(&Y2Z1, &p->Z, &q->Y, f);
  if (fe_eq(&X1Z2, &X2Z1, f) && fe_eq(&Y1Z2, &Y2Z1, f)) { pt_dbl(r, p, g); return; }
  fe_t Z1Z2, u, uu, v, vv, vvv, R, A, t, t2;
  fe_mul(&Z1Z2, &p->Z, &q->Z, f);
  fe_sub(&u, &Y2Z1, &Y1Z2, f);
  fe_sqr(&uu, &u, f);
  fe_sub(&v, &X2Z1, &X1Z2, f);
  fe_sqr(&vv, &v, f);
  fe_mul(&vvv, &v, &vv, f);
  fe_mul(&R, &vv, &X1Z2, f);
  fe_mul(&A, &uu, &Z1Z2, f); fe_add(&t, &vvv, &R, f); fe_add(&t, &t, &R, f); fe_sub(&A, &A, &t, f);
  pt_t o;
  fe_mul(&o.X, &v, &A, f);
  fe_sub(&t, &R, &A, f); fe_mul(&t, &u, &t, f); fe_mul(&t2, &vvv, &Y1Z2, f); fe_sub(&o.Y, &t, &t2, f);
  fe_mul(&o.Z, &vvv, &Z1Z2, f);
  *r = o;
}
/* to_affine_coordinates, mnt4753_g1.cpp:68-83 */
static void pt_to_affine(pt_t* p, const group_t* g) {
  const field_t* f = &g->f;
  if (pt_is_zero(p, g)) { pt_set_zero(p, g); return; }
  fe_t zi;
  fe_inv(&zi, &p->Z, f);
  fe_mul(&p->X, &p->X, &zi, f);
  fe_mul(&p->Y, &p->Y, &zi, f);
  fe_set_one(&p->Z, f);
}
/* scalar_mul, depends/libff/libff/algebra/curves/curve_utils.tcc:14-35 (double-and-add, MSB first) */
static void pt_scalar_mul(pt_t* r, const pt_t* base, const uint64_t* e, const group_t* g) {
  pt_t res;
  pt_set_zero(&res, g);
  int found_one = 0;
  for (long i = (long)big_num_bits(e) - 1; i >= 0; --i) {
    if (found_one) pt_dbl(&res, &res, g);
    if (big_test_bit(e, (size_t)i)) { found_one = 1; pt_add(&res, &res, base, g); }
  }
  *r = res;
}

/* ---- wire codec (S/serialization.hpp:22-121) ---- */
static int group_affine_words(const group_t* g) { return 24 * g->f.deg; }
/* read_g1 / read_g2: y == 0 -> zero(); else (x, y, one) */
static void pt_from_wire(pt_t* p, const uint64_t* w, const group_t* g) {
  const int d = g->f.deg;
  fe_set_zero(&p->X); fe_set_zero(&p->Y);
  for (int k = 0; k < d; ++k) {
    memcpy(p->X.c[k].l, w + 12 * k, 96);
    memcpy(p->Y.c[k].l, w + 12 * (d + k), 96);
  }
  if (fe_is_zero(&p->Y, &g->f)) { pt_set_zero(p, g); return; }
  fe_set_one(&p->Z, &g->f);
}
/* write_g1 / write_g2: zero -> all-zero x and y; else to_affine then X, Y */
static void pt_to_wire(uint64_t* w, const pt_t* p, const group_t* g) {
  const int d = g->f.deg;
  if (pt_is_zero(p, g)) { memset(w, 0, (size_t)group_affine_words(g) * 8); return; }
  pt_t a = *p;
  pt_to_affine(&a, g);
  for (int k = 0; k < d; ++k) {
    memcpy(w + 12 * k, a.X.c[k].l, 96);
    memcpy(w + 12 * (d + k), a.Y.c[k].l, 96);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Multi-exponentiation (SM/multiexp.tcc)
 * ---------------------------------------------------------------------------------------------- */
/* libff::log2 = ceil(log2 n), depends/libff/libff/common/utils.cpp:32-45 */
static size_t ceil_log2(size_t n) {
  size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
  while (n > 1) { n >>= 1; r++; }
  return r;
}
/* multi_exp_inner<BDLO12>, SM/multiexp.tcc:165-282 (USE_MIXED_ADDITION is off in the reference build,
 * build.sh:4 / CMakeLists.txt defaults, so buckets use operator+) */
static void msm_inner_bdlo12(pt_t* out, const pt_t* bases, const fp_t* exps, size_t length, const group_t* g) {
  size_t log2_length = ceil_log2(length);
  size_t c = log2_length - (log2_length / 3 - 2);
  uint64_t (*bn)[NLIMB] = (uint64_t (*)[NLIMB])malloc(length * sizeof(*bn));
  size_t num_bits = 0;
  for (size_t i = 0; i < length; ++i) {
    fp_as_bigint(bn[i], &exps[i], g->fr_mod);
    size_t nb = big_num_bits(bn[i]);
    if (nb > num_bits) num_bits = nb;
  }
  size_t num_groups = (num_bits + c - 1) / c;
  pt_t result;
  pt_set_zero(&result, g);
  int result_nonzero = 0;
  pt_t* buckets = (pt_t*)malloc(sizeof(pt_t) << c);
  unsigned char* bucket_nonzero = (unsigned char*)malloc((size_t)1 << c);
  for (size_t k = num_groups - 1; k <= num_groups; k--) {
    if (result_nonzero)
      for (size_t i = 0; i < c; ++i) pt_dbl(&result, &result, g);
    memset(bucket_nonzero, 0, (size_t)1 << c);
    for (size_t i = 0; i < length; ++i) {
      size_t id = 0;
      for (size_t j = 0; j < c; ++j)
        if (big_test_bit(bn[i], k * c + j)) id |= (size_t)1 << j;
      if (id == 0) continue;
      if (bucket_nonzero[id]) pt_add(&buckets[id], &buckets[id], &bases[i], g);
      else { buckets[id] = bases[i]; bucket_nonzero[id] = 1; }
    }
    pt_t running_sum;
    pt_set_zero(&running_sum, g);
    int running_sum_nonzero = 0;
    for (size_t i = ((size_t)1 << c) - 1; i > 0; --i) {
      if (bucket_nonzero[i]) {
        if (running_sum_nonzero) pt_add(&running_sum, &running_sum, &buckets[i], g);
        else { running_sum = buckets[i]; running_sum_nonzero = 1; }
      }
      if (running_sum_nonzero) {
        if (result_nonzero) pt_add(&result, &result, &running_sum, g);
        else { result = running_sum; result_nonzero = 1; }
      }
    }
  }
  free(buckets); free(bucket_nonzero); free(bn);
  *out = result;
}
/* multi_exp (chunked OpenMP driver), SM/multiexp.tcc:402-441 */
static void msm_chunked(pt_t* out, const pt_t* bases, const fp_t* exps, size_t total, size_t chunks, const group_t* g) {
  if (total < chunks || chunks == 1) {
    if (total == 0) { pt_set_zero(out, g); return; }
    msm_inner_bdlo12(out, bases, exps, total, g);
    return;
  }
  const size_t one = total / chunks;
  pt_t* partial = (pt_t*)malloc(sizeof(pt_t) * chunks);
#ifdef _OPENMP
#pragma omp parallel for
#endif
  for (size_t i = 0; i < chunks; ++i) {
    size_t lo = i * one, hi = (i == chunks - 1) ? total : (i + 1) * one;
    msm_inner_bdlo12(&partial[i], bases + lo, exps + lo, hi - lo, g);
  }
  pt_t fin;
  pt_set_zero(&fin, g);
  for (size_t i = 0; i < chunks; ++i) pt_add(&fin, &fin, &partial[i], g);
  free(partial);
  *out = fin;
}
/* multi_exp_with_mixed_addition, SM/multiexp.tcc:443-496: skip scalar 0, add bases with scalar 1 */
static void msm_with_mixed_addition(pt_t* out, const pt_t* bases, const fp_t* exps, size_t n, size_t chunks, const group_t* g) {
  fp_t zero, one;
  fp_set_zero(&zero);
  fp_set_one(&one, g->fr_mod);
  fp_t* p = (fp_t*)malloc(sizeof(fp_t) * (n ? n : 1));
  pt_t* gg = (pt_t*)malloc(sizeof(pt_t) * (n ? n : 1));
  size_t m = 0;
  pt_t acc;
  pt_set_zero(&acc, g);
  for (size_t i = 0; i < n; ++i) {
    if (fp_eq(&exps[i], &zero)) continue;
    if (fp_eq(&exps[i], &one)) { pt_add(&acc, &acc, &bases[i], g); continue; }
    p[m] = exps[i]; gg[m] = bases[i]; ++m;
  }
  pt_t rest;
  if (m == 0) pt_set_zero(&rest, g);  /* libff would take log2(0); the reference never hits it (dense witness) */
  else msm_chunked(&rest, gg, p, m, chunks, g);
  pt_add(out, &acc, &rest, g);
  free(p); free(gg);
}

/* ------------------------------------------------------------------------------------------------
 * basic_radix2_domain (Q/domains/basic_radix2_domain.tcc, basic_radix2_domain_aux.tcc)
 * ---------------------------------------------------------------------------------------------- */
/* libff::bitreverse, depends/libff/libff/common/utils.cpp:60-69 */
static size_t bitreverse(size_t n, size_t l) { size_t r = 0; for (size_t k = 0; k < l; ++k) { r = (r << 1) | (n & 1); n >>= 1; } return r; }
/* get_root_of_unity, depends/libff/libff/algebra/fields/field_utils.tcc:40-89.
 * MNT4753 Fr (modulus A): s = 30, root_of_unity squared (s - logn) times (:78-86).
 * MNT6753 Fr (modulus B): small_subgroup_defined, so omega = full_root_of_unity^(5^2) squared
 * (s - logn) times (:59-70; s = 15, mnt6753_init.cpp:66,73-76). */
static int fr_root_of_unity(fp_t* omega, size_t n, int fr_mod) {
  size_t logn = ceil_log2(n);
  if (n != ((size_t)1 << logn) || logn > (size_t)ORACLE_FR_S[fr_mod]) return -1;
  fp_t w;
  if (fr_mod == 1) {
    fp_t full;
    memcpy(full.l, ORACLE_FR_FULL_ROOT_B, sizeof(full.l));
    uint64_t q = 5;
    fp_pow(&w, &full, &q, 1, fr_mod);
    fp_pow(&w, &w, &q, 1, fr_mod);
  } else {
    memcpy(w.l, ORACLE_FR_ROOT_A, sizeof(w.l));
  }
  for (size_t i = (size_t)ORACLE_FR_S[fr_mod]; i > logn; --i) fp_mul(&w, &w, &w, fr_mod);
  *omega = w;
  return 0;
}
/* _basic_serial_radix2_FFT, Q/domains/basic_radix2_domain_aux.tcc:167-202 (for MNT6753's Fr the reference
 * takes _basic_serial_mixed_radix_FFT, :45-165, whose q_adicity == 0 branch is the same bit-reversal +
 * radix-2 passes; the OpenMP _basic_parallel_radix2_FFT, :217-319, computes the same vector) */
static void fr_fft_serial(fp_t* a, size_t n, const fp_t* omega, int mod) {
  const size_t logn = ceil_log2(n);
  for (size_t k = 0; k < n; ++k) {
    size_t rk = bitreverse(k, logn);
    if (k < rk) { fp_t t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  size_t m = 1;
  for (size_t s = 1; s <= logn; ++s) {
    uint64_t e = (uint64_t)(n / (2 * m));
    fp_t w_m;
    fp_pow(&w_m, omega, &e, 1, mod);
#ifdef _OPENMP
#pragma omp parallel for if (n >= 4096)
#endif
    for (size_t k = 0; k < n; k += 2 * m) {
      fp_t w;
      fp_set_one(&w, mod);
      for (size_t j = 0; j < m; ++j) {
        fp_t t;
        fp_mul(&t, &w, &a[k + j + m], mod);
        fp_sub(&a[k + j + m], &a[k + j], &t, mod);
        fp_add(&a[k + j], &a[k + j], &t, mod);
        fp_mul(&w, &w, &w_m, mod);
      }
    }
    m *= 2;
  }
}
/* _multiply_by_coset, aux.tcc:321-330 */
static void fr_multiply_by_coset(fp_t* a, size_t n, const fp_t* g, int mod) {
  fp_t u = *g;
  for (size_t i = 1; i < n; ++i) { fp_mul(&a[i], &a[i], &u, mod); fp_mul(&u, &u, g, mod); }
}
static void fr_generator(fp_t* g, int mod) { fp_from_u64(g, 17, mod); } /* multiplicative_generator, mnt4753_init.cpp:68 / mnt6753_init.cpp:69 */
/* FFT :62-68, iFFT :70-82, cosetFFT :84-89, icosetFFT :91-96 of basic_radix2_domain.tcc */
static int fr_domain_fft(fp_t* a, size_t n, int kind, int mod) {
  fp_t omega, g, ginv, t;
  if (n <= 1 || fr_root_of_unity(&omega, n, mod)) return -1;   /* ctor :25-60 */
  fr_generator(&g, mod);
  if (kind == 2) fr_multiply_by_coset(a, n, &g, mod);
  if (kind == 0 || kind == 2) { fr_fft_serial(a, n, &omega, mod); return 0; }
  fp_inv(&t, &omega, mod);
  fr_fft_serial(a, n, &t, mod);
  fp_t sconst, nn;
  fp_from_u64(&nn, (uint64_t)n, mod);
  fp_inv(&sconst, &nn, mod);
  for (size_t i = 0; i < n; ++i) fp_mul(&a[i], &a[i], &sconst, mod);
  if (kind == 3) { fp_inv(&ginv, &g, mod); fr_multiply_by_coset(a, n, &ginv, mod); }
  return 0;
}
/* divide_by_Z_on_coset, basic_radix2_domain.tcc:125-134; compute_vanishing_polynomial :113-116 */
static void fr_divide_by_z_on_coset(fp_t* a, size_t n, int mod) {
  fp_t g, z, one, zi;
  fr_generator(&g, mod);
  uint64_t e = (uint64_t)n;
  fp_pow(&z, &g, &e, 1, mod);
  fp_set_one(&one, mod);
  fp_sub(&z, &z, &one, mod);
  fp_inv(&zi, &z, mod);
  for (size_t i = 0; i < n; ++i) fp_mul(&a[i], &a[i], &zi, mod);
}
/* compute_H<B>, cuda_prover_piecewise.cu:18-53 == S/main.cpp:104-163; h has n + 1 entries */
static int fr_compute_h(fp_t* ca, fp_t* cb, fp_t* cc, fp_t* h, size_t n, int mod) {
  if (fr_domain_fft(ca, n, 1, mod)) return -1;
  fr_domain_fft(cb, n, 1, mod);
  fr_domain_fft(ca, n, 2, mod);
  fr_domain_fft(cb, n, 2, mod);
  for (size_t i = 0; i < n; ++i) fp_mul(&ca[i], &ca[i], &cb[i], mod);
  fr_domain_fft(cc, n, 1, mod);
  fr_domain_fft(cc, n, 2, mod);
  for (size_t i = 0; i < n; ++i) fp_sub(&ca[i], &ca[i], &cc[i], mod);
  fr_divide_by_z_on_coset(ca, n, mod);
  fr_domain_fft(ca, n, 3, mod);
  memcpy(h, ca, n * sizeof(fp_t));
  fp_set_zero(&h[n]);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Exported C interface (ctypes from tests/)
 * ---------------------------------------------------------------------------------------------- */
static int valid_cg(int curve, int group) { return (curve == 0 || curve == 1) && (group == 1 || group == 2); }

int oracle_field_op(int mod, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  if (mod < 0 || mod > 1) return -1;
  fp_t x, y, r;
  memcpy(x.l, a, 96);
  if (b) memcpy(y.l, b, 96); else fp_set_zero(&y);
  switch (op) {
    case 0: fp_mul(&r, &x, &y, mod); break;
    case 1: fp_add(&r, &x, &y, mod); break;
    case 2: fp_sub(&r, &x, &y, mod); break;
    case 3: fp_inv(&r, &x, mod); break;
    case 4: fp_as_bigint(r.l, &x, mod); break;
    case 5: fp_neg(&r, &x, mod); break;
    default: return -1;
  }
  memcpy(out, r.l, 96);
  return 0;
}
/* the coordinate field of G2 (Fq2 of MNT4753, Fq3 of MNT6753), elements c0 | c1 [| c2] in wire form: op 0 a*b (F/fp2.tcc:79-90,
 * F/fp3.tcc:83-96), 1 a.squared() (F/fp2.tcc:118-126, F/fp3.tcc:107-123), 2 a.inverse() (F/fp2.tcc:129-142, F/fp3.tcc:126-143), 3 a+b, 4 a-b, 5 -a */
int oracle_ext_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  if (curve < 0 || curve > 1) return -1;
  group_t g; group_init(&g, curve, 2);
  fe_t x, y, r;
  fe_set_zero(&x); fe_set_zero(&y); fe_set_zero(&r);
  for (int k = 0; k < g.f.deg; ++k) { memcpy(x.c[k].l, a + 12 * k, 96); if (b) memcpy(y.c[k].l, b + 12 * k, 96); }
  switch (op) {
    case 0: fe_mul(&r, &x, &y, &g.f); break;
    case 1: fe_sqr(&r, &x, &g.f); break;
    case 2: fe_inv(&r, &x, &g.f); break;
    case 3: fe_add(&r, &x, &y, &g.f); break;
    case 4: fe_sub(&r, &x, &y, &g.f); break;
    case 5: fe_neg(&r, &x, &g.f); break;
    default: return -1;
  }
  for (int k = 0; k < g.f.deg; ++k) memcpy(out + 12 * k, r.c[k].l, 96);
  return 0;
}
/* op: 0 = P + Q, 1 = dbl(P), 2 = P - Q, 3 = scalar (Fr wire, in q) * P.   inputs / output affine wire */
int oracle_point_op(int curve, int group, int op, const uint64_t* p, const uint64_t* q, uint64_t* out) {
  if (!valid_cg(curve, group)) return -1;
  group_t g; group_init(&g, curve, group);
  pt_t P, Q, Rr;
  pt_from_wire(&P, p, &g);
  switch (op) {
    case 0: pt_from_wire(&Q, q, &g); pt_add(&Rr, &P, &Q, &g); break;
    case 1: pt_dbl(&Rr, &P, &g); break;
    case 2: pt_from_wire(&Q, q, &g); fe_neg(&Q.Y, &Q.Y, &g.f); pt_add(&Rr, &P, &Q, &g); break;
    case 3: { fp_t s; memcpy(s.l, q, 96); uint64_t e[NLIMB]; fp_as_bigint(e, &s, g.fr_mod); pt_scalar_mul(&Rr, &P, e, &g); } break;
    default: return -1;
  }
  pt_to_wire(out, &Rr, &g);
  return 0;
}
/* B::multiexp_G1 / multiexp_G2 (S/prover_reference_functions.cpp:27-40, 247-266): BDLO12, `chunks` as
 * omp_get_max_threads() would give; bases affine wire, scalars Fr wire, out affine wire */
int oracle_msm(int curve, int group, const uint64_t* bases, const uint64_t* scalars, size_t n, size_t chunks, uint64_t* out_affine) {
  if (!valid_cg(curve, group) || chunks == 0) return -1;
  group_t g; group_init(&g, curve, group);
  const int aw = group_affine_words(&g);
  pt_t* pts = (pt_t*)malloc(sizeof(pt_t) * (n ? n : 1));
  fp_t* ex = (fp_t*)malloc(sizeof(fp_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) { pt_from_wire(&pts[i], bases + i * aw, &g); memcpy(ex[i].l, scalars + i * 12, 96); }
  pt_t r;
  msm_with_mixed_addition(&r, pts, ex, n, chunks, &g);
  pt_to_wire(out_affine, &r, &g);
  free(pts); free(ex);
  return 0;
}
/* Evaluation of the constraint system on the assignment: the first loop of r1cs_to_qap_witness_map
 * (S/reductions/r1cs_to_qap/r1cs_to_qap.tcc:223-237) = S/generate_parameters.cpp:44-57.  Matrices in CSR over the variables
 * (1, w_1 .. w_m) -- column 0 is the constant one = w[0] of the input file; linear_combination::evaluate, relations/variable.tcc.
 * ca, cb, cc: out_len elements each. */
int oracle_r1cs_evaluate(int curve, uint64_t num_inputs, uint64_t nc, const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                         const uint64_t* const coeff[3], const uint64_t* w, uint64_t* ca, uint64_t* cb, uint64_t* cc, size_t out_len) {
  if (curve < 0 || curve > 1 || out_len < nc + num_inputs + 1) return -1;
  const int mod = curve == 0 ? 0 : 1;
  uint64_t* outv[3] = {ca, cb, cc};
  for (int k = 0; k < 3; ++k) {
    memset(outv[k], 0, 96 * out_len);
    for (uint64_t i = 0; i < nc; ++i) {
      fp_t acc; memset(&acc, 0, sizeof(acc));
      for (uint64_t t = row_ptr[k][i]; t < row_ptr[k][i + 1]; ++t) {
        fp_t c, x, p;
        memcpy(c.l, coeff[k] + 12 * t, 96); memcpy(x.l, w + 12 * (size_t)col[k][t], 96);
        fp_mul(&p, &c, &x, mod);
        fp_add(&acc, &acc, &p, mod);
      }
      memcpy(outv[k] + 12 * i, acc.l, 96);
    }
  }
  for (uint64_t i = 0; i <= num_inputs; ++i) memcpy(ca + 12 * (nc + i), w + 12 * i, 96);   /* input consistency rows (:223-227) */
  return 0;
}
/* Completion of a challenge proof to a full Groth16 proof, S/main.cpp:312-319:
 *   A' = alpha + A + r delta,  B' = beta + B + s delta,  C' = C + s A' + r beta.
 * keys = alpha_g1 | beta_g1 | beta_g2 | delta_g1 | delta_g2, proof / out = A | B | C, all affine wire; r, s Fr wire. */
int oracle_complete_proof(int curve, const uint64_t* keys, const uint64_t* proof, const uint64_t* r, const uint64_t* s, uint64_t* out) {
  if (curve < 0 || curve > 1) return -1;
  group_t g1, g2; group_init(&g1, curve, 1); group_init(&g2, curve, 2);
  const int w1 = group_affine_words(&g1), w2 = group_affine_words(&g2);
  const uint64_t *alpha1 = keys, *beta1 = keys + w1, *beta2 = beta1 + w1, *delta1 = beta2 + w2, *delta2 = delta1 + w1;
  fp_t fr, fs; memcpy(fr.l, r, 96); memcpy(fs.l, s, 96);
  uint64_t er[NLIMB], es[NLIMB];
  fp_as_bigint(er, &fr, g1.fr_mod); fp_as_bigint(es, &fs, g1.fr_mod);
  pt_t A, B, C, t, u;
  pt_from_wire(&A, proof, &g1); pt_from_wire(&t, alpha1, &g1); pt_add(&A, &t, &A, &g1);
  pt_from_wire(&t, delta1, &g1); pt_scalar_mul(&u, &t, er, &g1); pt_add(&A, &A, &u, &g1);
  pt_from_wire(&B, proof + w1, &g2); pt_from_wire(&t, beta2, &g2); pt_add(&B, &t, &B, &g2);
  pt_from_wire(&t, delta2, &g2); pt_scalar_mul(&u, &t, es, &g2); pt_add(&B, &B, &u, &g2);
  pt_from_wire(&C, proof + w1 + w2, &g1); pt_scalar_mul(&u, &A, es, &g1); pt_add(&C, &C, &u, &g1);
  pt_from_wire(&t, beta1, &g1); pt_scalar_mul(&u, &t, er, &g1); pt_add(&C, &C, &u, &g1);
  pt_to_wire(out, &A, &g1); pt_to_wire(out + w1, &B, &g2); pt_to_wire(out + w1 + w2, &C, &g1);
  return 0;
}
int oracle_fft(int curve, int kind, uint64_t* vec, size_t m) {
  if (curve < 0 || curve > 1 || kind < 0 || kind > 3) return -1;
  return fr_domain_fft((fp_t*)vec, m, kind, curve == 0 ? 0 : 1);
}
int oracle_divide_by_z_on_coset(int curve, uint64_t* vec, size_t m) {
  if (curve < 0 || curve > 1) return -1;
  fr_divide_by_z_on_coset((fp_t*)vec, m, curve == 0 ? 0 : 1);
  return 0;
}
int oracle_compute_h(int curve, uint64_t* ca, uint64_t* cb, uint64_t* cc, uint64_t* h, size_t m) {
  if (curve < 0 || curve > 1) return -1;
  return fr_compute_h((fp_t*)ca, (fp_t*)cb, (fp_t*)cc, (fp_t*)h, m, curve == 0 ? 0 : 1);
}

/* run_prover<B>, cuda_prover_piecewise.cu:55-98 == S/main.cpp:187-272; file layouts
 * S/generate_parameters.cpp:60-108, readers S/prover_reference_functions.cpp:48-116, writer :347-356 */
static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }
int oracle_prove(int curve, const char* params_path, const char* input_path, const char* output_path, size_t chunks, double* timings) {
  if (curve < 0 || curve > 1 || chunks == 0) return -1;
  group_t g1, g2; group_init(&g1, curve, 1); group_init(&g2, curve, 2);
  const int frm = g1.fr_mod;
  FILE* pf = fopen(params_path, "rb");
  if (!pf) return -2;
  uint64_t d, m;
  if (read_exact(pf, &d, 8) || read_exact(pf, &m, 8)) { fclose(pf); return -2; }
  const size_t a1 = 24, a2 = (size_t)group_affine_words(&g2);
  uint64_t* A = (uint64_t*)malloc(8 * a1 * (m + 1)); uint64_t* B1 = (uint64_t*)malloc(8 * a1 * (m + 1));
  uint64_t* B2 = (uint64_t*)malloc(8 * a2 * (m + 1)); uint64_t* L = (uint64_t*)malloc(8 * a1 * (m - 1));
  uint64_t* H = (uint64_t*)malloc(8 * a1 * d);
  int bad = read_exact(pf, A, 8 * a1 * (m + 1)) | read_exact(pf, B1, 8 * a1 * (m + 1)) | read_exact(pf, B2, 8 * a2 * (m + 1)) |
            read_exact(pf, L, 8 * a1 * (m - 1)) | read_exact(pf, H, 8 * a1 * d);
  fclose(pf);
  if (bad) return -2;
  double t_start = 0;
#ifdef _OPENMP
  t_start = omp_get_wtime();
#endif
  FILE* inf = fopen(input_path, "rb");
  if (!inf) return -3;
  uint64_t* w = (uint64_t*)malloc(96 * (m + 1));
  fp_t* ca = (fp_t*)malloc(96 * (d + 1)); fp_t* cb = (fp_t*)malloc(96 * (d + 1)); fp_t* cc = (fp_t*)malloc(96 * (d + 1));
  fp_t r;
  bad = read_exact(inf, w, 96 * (m + 1)) | read_exact(inf, ca, 96 * (d + 1)) | read_exact(inf, cb, 96 * (d + 1)) |
        read_exact(inf, cc, 96 * (d + 1)) | read_exact(inf, r.l, 96);
  fclose(inf);
  if (bad) return -3;
  double t_loaded = 0;
#ifdef _OPENMP
  t_loaded = omp_get_wtime();
#endif
  fp_t* h = (fp_t*)malloc(96 * (d + 2));
  if (fr_compute_h(ca, cb, cc, h, d + 1, frm)) return -4;
  double t_h = 0;
#ifdef _OPENMP
  t_h = omp_get_wtime();
#endif
  uint64_t At[24], Bt1[24], Bt2[72], Ht[24], Lt[24];
  oracle_msm(curve, 1, A, w, m + 1, chunks, At);
  oracle_msm(curve, 1, B1, w, m + 1, chunks, Bt1);
  oracle_msm(curve, 2, B2, w, m + 1, chunks, Bt2);
  oracle_msm(curve, 1, H, (const uint64_t*)h, d, chunks, Ht);
  oracle_msm(curve, 1, L, w + 12 * 2, m - 1, chunks, Lt);   /* vector_Fr_offset(w, primary_input_size + 1) */
  double t_msm = 0;
#ifdef _OPENMP
  t_msm = omp_get_wtime();
#endif
  /* C = Ht + (Lt + r * Bt1) */
  pt_t pB1, pH, pL, sc, t1, Cc;
  pt_from_wire(&pB1, Bt1, &g1); pt_from_wire(&pH, Ht, &g1); pt_from_wire(&pL, Lt, &g1);
  uint64_t e[NLIMB];
  fp_as_bigint(e, &r, frm);
  pt_scalar_mul(&sc, &pB1, e, &g1);
  pt_add(&t1, &pL, &sc, &g1);
  pt_add(&Cc, &pH, &t1, &g1);
  uint64_t Cw[24];
  pt_to_wire(Cw, &Cc, &g1);
  FILE* of = fopen(output_path, "wb");
  if (!of) return -5;
  fwrite(At, 8, 24, of); fwrite(Bt2, 8, a2, of); fwrite(Cw, 8, 24, of);
  fclose(of);
  if (timings) {
    double t_end = 0;
#ifdef _OPENMP
    t_end = omp_get_wtime();
#endif
    timings[0] = t_loaded - t_start; timings[1] = t_h - t_loaded; timings[2] = t_msm - t_h; timings[3] = t_end - t_start;
  }
  free(A); free(B1); free(B2); free(L); free(H); free(w); free(ca); free(cb); free(cc); free(h);
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#ifdef ORACLE_MAIN
/* ./main <curve> compute <params> <input> <output>  -- same CLI as S/main.cpp:274-293 */
int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s MNT4753|MNT6753 compute <params> <input> <output>\n", argv[0]); return 2; }
  int curve = strcmp(argv[1], "MNT4753") == 0 ? 0 : (strcmp(argv[1], "MNT6753") == 0 ? 1 : -1);
  if (curve < 0 || strcmp(argv[2], "compute") != 0) return 2;
  double t[4];
  int rc = oracle_prove(curve, argv[3], argv[4], argv[5], (size_t)oracle_max_threads(), t);
  if (rc) { fprintf(stderr, "oracle_prove failed: %d\n", rc); return 1; }
  printf("load inputs: %.3fs\ncompute_H: %.3fs\nmultiexp: %.3fs\nTotal time from input to output: %.3fs\n", t[0], t[1], t[2], t[3]);
  return 0;
}
#endif
