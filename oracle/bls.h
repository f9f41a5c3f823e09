/*
 * oracle/bls.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the arithmetic the reference obtains from its third-party
 * dependencies blst (>=0.3.16) and blstrs 0.7.1 (reference:
 * crates/cryptography/bls12_381/Cargo.toml:17,20; type aliases
 * crates/cryptography/bls12_381/src/lib.rs:23-42).  Those crates are NOT vendored
 * in /root/reference, so this file restates the published BLS12-381 algorithms
 * (Montgomery fields, short-Weierstrass Jacobian group law, ZCash point encoding,
 * optimal-ate pairing) and is pinned end-to-end by the reference's own golden
 * vectors (tests/golden, converted from /root/reference/test_vectors).
 *
 * Nothing under rust-eth-kzg_amd/ (the product) may include or link this.
 */
#ifndef KZG_ORACLE_BLS_H
#define KZG_ORACLE_BLS_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } fr_t; /* Montgomery form, R = 2^256 */
typedef struct { uint64_t l[6]; } fp_t; /* Montgomery form, R = 2^384 */
typedef struct { fp_t x, y, z; } g1_t;  /* Jacobian; identity <=> z == 0 */
typedef struct { fp_t x, y; int inf; } g1a_t; /* affine */

typedef struct { fp_t c0, c1; } fp2_t;          /* c0 + c1*u,  u^2 = -1 */
typedef struct { fp2_t c0, c1, c2; } fp6_t;     /* c0 + c1*v + c2*v^2, v^3 = 1+u */
typedef struct { fp6_t c0, c1; } fp12_t;        /* c0 + c1*w, w^2 = v */
typedef struct { fp2_t x, y; int inf; } g2a_t;

void bls_init(void); /* idempotent; computes Montgomery constants */

/* ---- Fr ---- */
extern fr_t FR_ONE, FR_ZERO;
void fr_add(fr_t *o, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *o, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *o, const fr_t *a);
void fr_mul(fr_t *o, const fr_t *a, const fr_t *b);
void fr_inv(fr_t *o, const fr_t *a);
void fr_pow_u64(fr_t *o, const fr_t *a, uint64_t e);
int  fr_is_zero(const fr_t *a);
int  fr_eq(const fr_t *a, const fr_t *b);
void fr_from_u64(fr_t *o, uint64_t v);
int  fr_from_be(fr_t *o, const uint8_t b[32]);          /* 0 ok, -1 if >= r */
void fr_from_be_reduce(fr_t *o, const uint8_t b[32]);   /* value mod r */
void fr_to_be(uint8_t b[32], const fr_t *a);
void fr_to_le_canon(uint64_t o[4], const fr_t *a);      /* out of Montgomery */
void fr_root_of_unity(fr_t *o, unsigned log_n);         /* 7^((r-1)/2^log_n) */
void fr_batch_inverse(fr_t *v, size_t n);               /* all nonzero */

/* ---- Fp ---- */
extern fp_t FP_ONE, FP_ZERO;
void fp_add(fp_t *o, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *o, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *o, const fp_t *a);
void fp_mul(fp_t *o, const fp_t *a, const fp_t *b);
void fp_sqr(fp_t *o, const fp_t *a);
void fp_inv(fp_t *o, const fp_t *a);
int  fp_sqrt(fp_t *o, const fp_t *a);                   /* 1 if square */
int  fp_is_zero(const fp_t *a);
int  fp_eq(const fp_t *a, const fp_t *b);
int  fp_from_be(fp_t *o, const uint8_t b[48]);          /* 0 ok, -1 if >= p */
void fp_to_be(uint8_t b[48], const fp_t *a);
int  fp_is_lex_largest(const fp_t *a);                  /* a > (p-1)/2 */
void fp_from_u64(fp_t *o, uint64_t v);

/* ---- G1 ---- */
void g1_set_inf(g1_t *p);
int  g1_is_inf(const g1_t *p);
void g1_from_affine(g1_t *o, const g1a_t *a);
void g1_to_affine(g1a_t *o, const g1_t *p);
void g1_dbl(g1_t *o, const g1_t *p);
void g1_add(g1_t *o, const g1_t *p, const g1_t *q);
void g1_add_affine(g1_t *o, const g1_t *p, const g1a_t *q);
void g1_neg(g1_t *o, const g1_t *p);
void g1_sub(g1_t *o, const g1_t *p, const g1_t *q);
void g1_mul(g1_t *o, const g1_t *p, const fr_t *k);     /* variable-base scalar mul */
int  g1_eq(const g1_t *p, const g1_t *q);
void g1_batch_normalize(g1a_t *o, const g1_t *p, size_t n);
void g1_compress(uint8_t out[48], const g1a_t *a);
/* returns 0 ok; -1 bad encoding / not on curve; -2 not in subgroup (only if check) */
int  g1_decompress(g1a_t *o, const uint8_t in[48], int subgroup_check);
int  g1_in_subgroup(const g1a_t *a);
void g1_generator(g1a_t *o);
void g1_msm(g1_t *o, const g1a_t *pts, const fr_t *k, size_t n); /* Pippenger */

/* ---- pairing ---- */
int  g2_decompress(g2a_t *o, const uint8_t in[96]);
void g2_neg(g2a_t *o, const g2a_t *a);
void g2_generator(g2a_t *o);
/* product of e(P_i, Q_i) == 1 ? */
int  pairing_product_is_one(const g1a_t *P, const g2a_t *Q, size_t n);

/* ---- sha256 ---- */
void sha256(uint8_t out[32], const uint8_t *data, size_t len);

#ifdef __cplusplus
}
#endif
#endif
