/*
 * oracle/g1.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * BLS12-381 G1 (y^2 = x^3 + 4 over Fp): Jacobian group law, scalar multiplication,
 * ZCash 48-byte compressed encoding, subgroup check, batch normalisation, Pippenger MSM.
 * Restates the blstrs/blst entry points the reference calls:
 *   G1Projective + / double / * Scalar      crates/cryptography/polynomial/src/fft.rs:9-35,164-177
 *   g1_batch_normalize                      crates/cryptography/bls12_381/src/lib.rs:56-104
 *   G1Affine::from_compressed / to_compressed  crates/serialization/src/lib.rs:69-99,84-86
 *   g1_lincomb -> multi_exp (Pippenger)     crates/cryptography/bls12_381/src/lincomb.rs:7-30
 */
#include "bls.h"
#include <stdlib.h>

void g1_set_inf(g1_t *p) { p->x = FP_ONE; p->y = FP_ONE; p->z = FP_ZERO; }
int g1_is_inf(const g1_t *p) { return fp_is_zero(&p->z); }
void g1_from_affine(g1_t *o, const g1a_t *a) {
    if (a->inf) { g1_set_inf(o); return; }
    o->x = a->x; o->y = a->y; o->z = FP_ONE;
}
void g1_to_affine(g1a_t *o, const g1_t *p) {
    if (g1_is_inf(p)) { o->inf = 1; o->x = FP_ZERO; o->y = FP_ZERO; return; }
    fp_t zi, zi2, zi3;
    fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&o->x, &p->x, &zi2); fp_mul(&o->y, &p->y, &zi3); o->inf = 0;
}
void g1_neg(g1_t *o, const g1_t *p) { o->x = p->x; fp_neg(&o->y, &p->y); o->z = p->z; }

void g1_dbl(g1_t *o, const g1_t *p) {
    if (g1_is_inf(p)) { g1_set_inf(o); return; }
    fp_t A, B, C, D, E, F, t, x3, y3, z3;
    fp_sqr(&A, &p->x); fp_sqr(&B, &p->y); fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B); fp_sqr(&t, &t); fp_sub(&t, &t, &A); fp_sub(&t, &t, &C); fp_add(&D, &t, &t);
    fp_add(&E, &A, &A); fp_add(&E, &E, &A);
    fp_sqr(&F, &E);
    fp_sub(&x3, &F, &D); fp_sub(&x3, &x3, &D);
    fp_sub(&t, &D, &x3); fp_mul(&y3, &E, &t);
    fp_add(&C, &C, &C); fp_add(&C, &C, &C); fp_add(&C, &C, &C);
    fp_sub(&y3, &y3, &C);
    fp_mul(&z3, &p->y, &p->z); fp_add(&z3, &z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}
void g1_add(g1_t *o, const g1_t *p, const g1_t *q) {
    if (g1_is_inf(p)) { *o = *q; return; }
    if (g1_is_inf(q)) { *o = *p; return; }
    fp_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&u1, &p->x, &z2z2); fp_mul(&u2, &q->x, &z1z1);
    fp_mul(&s1, &p->y, &q->z); fp_mul(&s1, &s1, &z2z2);
    fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
    fp_sub(&h, &u2, &u1); fp_sub(&r, &s2, &s1);
    if (fp_is_zero(&h)) {
        if (fp_is_zero(&r)) { g1_dbl(o, p); return; }
        g1_set_inf(o); return;
    }
    fp_add(&r, &r, &r);
    fp_add(&i, &h, &h); fp_sqr(&i, &i);
    fp_mul(&j, &h, &i);
    fp_mul(&v, &u1, &i);
    fp_sqr(&x3, &r); fp_sub(&x3, &x3, &j); fp_sub(&x3, &x3, &v); fp_sub(&x3, &x3, &v);
    fp_sub(&t, &v, &x3); fp_mul(&y3, &r, &t);
    fp_mul(&t, &s1, &j); fp_add(&t, &t, &t); fp_sub(&y3, &y3, &t);
    fp_add(&z3, &p->z, &q->z); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &z1z1); fp_sub(&z3, &z3, &z2z2);
    fp_mul(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
void g1_add_affine(g1_t *o, const g1_t *p, const g1a_t *q) {
    g1_t qq; g1_from_affine(&qq, q); g1_add(o, p, &qq);
}
void g1_sub(g1_t *o, const g1_t *p, const g1_t *q) { g1_t n; g1_neg(&n, q); g1_add(o, p, &n); }
int g1_eq(const g1_t *p, const g1_t *q) {
    int pi = g1_is_inf(p), qi = g1_is_inf(q);
    if (pi || qi) return pi && qi;
    fp_t z1z1, z2z2, a, b;
    fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
    fp_mul(&a, &p->x, &z2z2); fp_mul(&b, &q->x, &z1z1);
    if (!fp_eq(&a, &b)) return 0;
    fp_mul(&a, &p->y, &q->z); fp_mul(&a, &a, &z2z2);
    fp_mul(&b, &q->y, &p->z); fp_mul(&b, &b, &z1z1);
    return fp_eq(&a, &b);
}
static void g1_mul_limbs(g1_t *o, const g1_t *p, const uint64_t *k, int nl) {
    /* fixed 4-bit window, MSB first */
    g1_t tab[16]; g1_set_inf(&tab[0]); tab[1] = *p;
    for (int i = 2; i < 16; i++) g1_add(&tab[i], &tab[i - 1], p);
    g1_t acc; g1_set_inf(&acc);
    for (int i = 16 * nl - 1; i >= 0; i--) {
        for (int d = 0; d < 4; d++) g1_dbl(&acc, &acc);
        unsigned w = (unsigned)(k[i / 16] >> (4 * (i % 16))) & 15;
        if (w) g1_add(&acc, &acc, &tab[w]);
    }
    *o = acc;
}
void g1_mul(g1_t *o, const g1_t *p, const fr_t *k) {
    uint64_t c[4]; fr_to_le_canon(c, k);
    g1_mul_limbs(o, p, c, 4);
}
void g1_batch_normalize(g1a_t *o, const g1_t *p, size_t n) {
    /* identities are handled out-of-band exactly as the reference does (lib.rs:61-79) */
    fp_t *pre = (fp_t *)malloc((n + 1) * sizeof(fp_t));
    fp_t acc = FP_ONE;
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!g1_is_inf(&p[i])) fp_mul(&acc, &acc, &p[i].z);
    }
    fp_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (g1_is_inf(&p[i])) { o[i].inf = 1; o[i].x = FP_ZERO; o[i].y = FP_ZERO; continue; }
        fp_t zi, zi2, zi3;
        fp_mul(&zi, &acc, &pre[i]);
        fp_mul(&acc, &acc, &p[i].z);
        fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
        fp_mul(&o[i].x, &p[i].x, &zi2); fp_mul(&o[i].y, &p[i].y, &zi3); o[i].inf = 0;
    }
    free(pre);
}
void g1_compress(uint8_t out[48], const g1a_t *a) {
    if (a->inf) { memset(out, 0, 48); out[0] = 0xc0; return; }
    fp_to_be(out, &a->x);
    out[0] |= 0x80;
    if (fp_is_lex_largest(&a->y)) out[0] |= 0x20;
}
int g1_in_subgroup(const g1a_t *a) {
    if (a->inf) return 1;
    /* definitional check: [r]P == O */
    static const uint64_t R_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL,
                                      0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    g1_t p, q; g1_from_affine(&p, a);
    g1_mul_limbs(&q, &p, R_MOD, 4);
    return g1_is_inf(&q);
}
int g1_decompress(g1a_t *o, const uint8_t in[48], int subgroup_check) {
    uint8_t b[48]; memcpy(b, in, 48);
    int compressed = (b[0] >> 7) & 1, infinity = (b[0] >> 6) & 1, sign = (b[0] >> 5) & 1;
    if (!compressed) return -1;
    b[0] &= 0x1f;
    if (infinity) {
        if (sign) return -1;
        for (int i = 0; i < 48; i++) if (b[i]) return -1;
        o->inf = 1; o->x = FP_ZERO; o->y = FP_ZERO; return 0;
    }
    fp_t x, y, y2, four;
    if (fp_from_be(&x, b)) return -1;
    fp_sqr(&y2, &x); fp_mul(&y2, &y2, &x); fp_from_u64(&four, 4); fp_add(&y2, &y2, &four);
    if (!fp_sqrt(&y, &y2)) return -1;
    if (fp_is_lex_largest(&y) != sign) fp_neg(&y, &y);
    o->x = x; o->y = y; o->inf = 0;
    if (subgroup_check && !g1_in_subgroup(o)) return -2;
    return 0;
}
void g1_generator(g1a_t *o) {
    static const uint8_t G[48] = {
        0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
        0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
        0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    g1_decompress(o, G, 0);
}

/* Pippenger bucket MSM (the algorithm class of blst's multi_exp; only the group element
   it returns is observable). Zero scalars / identity points are skipped as g1_lincomb does. */
void g1_msm(g1_t *o, const g1a_t *pts, const fr_t *k, size_t n) {
    g1_t acc; g1_set_inf(&acc);
    if (n == 0) { *o = acc; return; }
    unsigned c = n < 8 ? 2 : n < 64 ? 4 : n < 1024 ? 7 : n < 16384 ? 10 : 12;
    unsigned nwin = (255 + c - 1) / c;
    size_t nb = (size_t)1 << c;
    uint64_t (*sc)[4] = malloc(n * sizeof *sc);
    for (size_t i = 0; i < n; i++) fr_to_le_canon(sc[i], &k[i]);
    g1_t *bk = (g1_t *)malloc(nb * sizeof(g1_t));
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (unsigned d = 0; d < c; d++) g1_dbl(&acc, &acc);
        for (size_t b = 0; b < nb; b++) g1_set_inf(&bk[b]);
        for (size_t i = 0; i < n; i++) {
            if (pts[i].inf) continue;
            unsigned bit = (unsigned)w * c;
            uint64_t v = sc[i][bit / 64] >> (bit % 64);
            if (bit % 64 + c > 64 && bit / 64 + 1 < 4) v |= sc[i][bit / 64 + 1] << (64 - bit % 64);
            v &= nb - 1;
            if (v) g1_add_affine(&bk[v], &bk[v], &pts[i]);
        }
        g1_t run, sum; g1_set_inf(&run); g1_set_inf(&sum);
        for (size_t b = nb - 1; b >= 1; b--) { g1_add(&run, &run, &bk[b]); g1_add(&sum, &sum, &run); }
        g1_add(&acc, &acc, &sum);
    }
    free(bk); free(sc);
    *o = acc;
}
