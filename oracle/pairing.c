/*
 * oracle/pairing.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * 2-pairing product check used by the cell-proof batch verifier
 * (reference: multi_pairings, crates/cryptography/bls12_381/src/lib.rs:45-50, called from
 * crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:251-259; G2Prepared inputs :88-90).
 * The reference delegates to blstrs/blst; this is the textbook ate pairing written for
 * clarity, not speed: G2 points are untwisted into E(Fp12) and the Miller loop runs in
 * affine coordinates over Fp12; the final exponentiation is f^(p^6-1) then a plain
 * square-and-multiply by (p^6+1)/r.
 * Tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v).
 * Twist (M-type): E'(Fp2): y^2 = x^3 + 4(1+u);  (x,y) -> (x/w^2, y/w^3) in E(Fp12).
 */
#include "bls.h"

/* ---------------- Fp2 ---------------- */
static void fp2_add(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_add(&o->c0, &a->c0, &b->c0); fp_add(&o->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2_t *o, const fp2_t *a, const fp2_t *b) { fp_sub(&o->c0, &a->c0, &b->c0); fp_sub(&o->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2_t *o, const fp2_t *a) { fp_neg(&o->c0, &a->c0); fp_neg(&o->c1, &a->c1); }
static void fp2_mul(fp2_t *o, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1); fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&o->c0, &t0, &t1); fp_add(&o->c1, &t2, &t3);
}
static void fp2_sqr(fp2_t *o, const fp2_t *a) { fp2_mul(o, a, a); }
static void fp2_conj(fp2_t *o, const fp2_t *a) { o->c0 = a->c0; fp_neg(&o->c1, &a->c1); }
static void fp2_inv(fp2_t *o, const fp2_t *a) {
    fp_t n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&o->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&o->c1, &t);
}
static void fp2_mul_xi(fp2_t *o, const fp2_t *a) { /* * (1+u) */
    fp_t t0, t1; fp_sub(&t0, &a->c0, &a->c1); fp_add(&t1, &a->c0, &a->c1); o->c0 = t0; o->c1 = t1;
}
static int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_pow(fp2_t *o, const fp2_t *a, const uint64_t *e, int nl) {
    fp2_t acc; acc.c0 = FP_ONE; acc.c1 = FP_ZERO;
    for (int i = 64 * nl - 1; i >= 0; i--) {
        fp2_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, a);
    }
    *o = acc;
}
static int fp2_sqrt(fp2_t *o, const fp2_t *a) {
    /* p = 3 mod 4 (Adj & Rodriguez-Henriquez, Alg. 9) */
    static const uint64_t E1[6] = { /* (p-3)/4 */
        0xee7fbfffffffeaaaULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
        0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
    static const uint64_t E2[6] = { /* (p-1)/2 */
        0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
        0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
    if (fp2_is_zero(a)) { *o = *a; return 1; }
    fp2_t a1, alpha, a0, x0, t, minus_one, x;
    fp2_pow(&a1, a, E1, 6);
    fp2_mul(&x0, &a1, a);
    fp2_mul(&alpha, &a1, &x0);
    fp2_conj(&t, &alpha); fp2_mul(&a0, &t, &alpha);
    fp_neg(&minus_one.c0, &FP_ONE); minus_one.c1 = FP_ZERO;
    if (fp2_eq(&a0, &minus_one)) return 0;
    if (fp2_eq(&alpha, &minus_one)) { /* x = u * x0 */
        fp_neg(&x.c0, &x0.c1); x.c1 = x0.c0;
    } else {
        fp2_t b; t = alpha; fp_add(&t.c0, &t.c0, &FP_ONE);
        fp2_pow(&b, &t, E2, 6);
        fp2_mul(&x, &b, &x0);
    }
    fp2_sqr(&t, &x);
    if (!fp2_eq(&t, a)) return 0;
    *o = x; return 1;
}

/* ---------------- Fp6 ---------------- */
static void fp6_add(fp6_t *o, const fp6_t *a, const fp6_t *b) { fp2_add(&o->c0, &a->c0, &b->c0); fp2_add(&o->c1, &a->c1, &b->c1); fp2_add(&o->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6_t *o, const fp6_t *a, const fp6_t *b) { fp2_sub(&o->c0, &a->c0, &b->c0); fp2_sub(&o->c1, &a->c1, &b->c1); fp2_sub(&o->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6_t *o, const fp6_t *a) { fp2_neg(&o->c0, &a->c0); fp2_neg(&o->c1, &a->c1); fp2_neg(&o->c2, &a->c2); }
static void fp6_mul(fp6_t *o, const fp6_t *a, const fp6_t *b) {
    fp2_t a0b0, a1b1, a2b2, t, u, c0, c1, c2;
    fp2_mul(&a0b0, &a->c0, &b->c0); fp2_mul(&a1b1, &a->c1, &b->c1); fp2_mul(&a2b2, &a->c2, &b->c2);
    fp2_mul(&t, &a->c1, &b->c2); fp2_mul(&u, &a->c2, &b->c1); fp2_add(&t, &t, &u); fp2_mul_xi(&t, &t); fp2_add(&c0, &a0b0, &t);
    fp2_mul(&t, &a->c0, &b->c1); fp2_mul(&u, &a->c1, &b->c0); fp2_add(&t, &t, &u); fp2_mul_xi(&u, &a2b2); fp2_add(&c1, &t, &u);
    fp2_mul(&t, &a->c0, &b->c2); fp2_mul(&u, &a->c2, &b->c0); fp2_add(&t, &t, &u); fp2_add(&c2, &t, &a1b1);
    o->c0 = c0; o->c1 = c1; o->c2 = c2;
}
static void fp6_mul_v(fp6_t *o, const fp6_t *a) { fp2_t t; fp2_mul_xi(&t, &a->c2); o->c2 = a->c1; o->c1 = a->c0; o->c0 = t; }
static void fp6_inv(fp6_t *o, const fp6_t *a) {
    fp2_t t0, t1, t2, d, x, y;
    fp2_sqr(&t0, &a->c0); fp2_mul(&x, &a->c1, &a->c2); fp2_mul_xi(&x, &x); fp2_sub(&t0, &t0, &x);
    fp2_sqr(&t1, &a->c2); fp2_mul_xi(&t1, &t1); fp2_mul(&x, &a->c0, &a->c1); fp2_sub(&t1, &t1, &x);
    fp2_sqr(&t2, &a->c1); fp2_mul(&x, &a->c0, &a->c2); fp2_sub(&t2, &t2, &x);
    fp2_mul(&d, &a->c0, &t0);
    fp2_mul(&x, &a->c2, &t1); fp2_mul(&y, &a->c1, &t2); fp2_add(&x, &x, &y); fp2_mul_xi(&x, &x); fp2_add(&d, &d, &x);
    fp2_inv(&d, &d);
    fp2_mul(&o->c0, &t0, &d); fp2_mul(&o->c1, &t1, &d); fp2_mul(&o->c2, &t2, &d);
}
static int fp6_is_zero(const fp6_t *a) { return fp2_is_zero(&a->c0) && fp2_is_zero(&a->c1) && fp2_is_zero(&a->c2); }

/* ---------------- Fp12 ---------------- */
static void fp12_one(fp12_t *o) { memset(o, 0, sizeof *o); o->c0.c0.c0 = FP_ONE; }
static void fp12_add(fp12_t *o, const fp12_t *a, const fp12_t *b) { fp6_add(&o->c0, &a->c0, &b->c0); fp6_add(&o->c1, &a->c1, &b->c1); }
static void fp12_sub(fp12_t *o, const fp12_t *a, const fp12_t *b) { fp6_sub(&o->c0, &a->c0, &b->c0); fp6_sub(&o->c1, &a->c1, &b->c1); }
static void fp12_mul(fp12_t *o, const fp12_t *a, const fp12_t *b) {
    fp6_t t0, t1, t2, c0, c1;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_mul_v(&t2, &t1); fp6_add(&c0, &t0, &t2);
    fp6_mul(&t0, &a->c0, &b->c1); fp6_mul(&t1, &a->c1, &b->c0); fp6_add(&c1, &t0, &t1);
    o->c0 = c0; o->c1 = c1;
}
static void fp12_conj(fp12_t *o, const fp12_t *a) { o->c0 = a->c0; fp6_neg(&o->c1, &a->c1); }
static void fp12_inv(fp12_t *o, const fp12_t *a) {
    fp6_t t0, t1;
    fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1);
    fp6_inv(&t0, &t0);
    fp6_mul(&o->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&o->c1, &t1);
}
static int fp12_is_one(const fp12_t *a) {
    fp12_t one; fp12_one(&one);
    return memcmp(a, &one, sizeof one) == 0;
}
static void fp12_from_fp(fp12_t *o, const fp_t *a) { memset(o, 0, sizeof *o); o->c0.c0.c0 = *a; }

/* ---------------- G2 ---------------- */
static int fp2_lex_largest(const fp2_t *y) {
    if (!fp_is_zero(&y->c1)) return fp_is_lex_largest(&y->c1);
    return fp_is_lex_largest(&y->c0);
}
int g2_decompress(g2a_t *o, const uint8_t in[96]) {
    uint8_t b[96]; memcpy(b, in, 96);
    int compressed = (b[0] >> 7) & 1, infinity = (b[0] >> 6) & 1, sign = (b[0] >> 5) & 1;
    if (!compressed) return -1;
    b[0] &= 0x1f;
    if (infinity) {
        if (sign) return -1;
        for (int i = 0; i < 96; i++) if (b[i]) return -1;
        memset(o, 0, sizeof *o); o->inf = 1; return 0;
    }
    fp2_t x, y, y2, bcoef;
    if (fp_from_be(&x.c1, b) || fp_from_be(&x.c0, b + 48)) return -1;
    fp2_sqr(&y2, &x); fp2_mul(&y2, &y2, &x);
    fp_from_u64(&bcoef.c0, 4); bcoef.c1 = bcoef.c0;
    fp2_add(&y2, &y2, &bcoef);
    if (!fp2_sqrt(&y, &y2)) return -1;
    if (fp2_lex_largest(&y) != sign) fp2_neg(&y, &y);
    o->x = x; o->y = y; o->inf = 0;
    return 0;
}
void g2_neg(g2a_t *o, const g2a_t *a) { o->x = a->x; fp2_neg(&o->y, &a->y); o->inf = a->inf; }

/* ---------------- pairing ---------------- */
typedef struct { fp12_t x, y; } e12_t;

static void untwist(e12_t *o, const g2a_t *q) {
    fp12_t w, w2, w3, x, y;
    memset(&w, 0, sizeof w); w.c1.c0.c0 = FP_ONE;
    fp12_mul(&w2, &w, &w); fp12_mul(&w3, &w2, &w);
    fp12_inv(&w2, &w2); fp12_inv(&w3, &w3);
    memset(&x, 0, sizeof x); x.c0.c0 = q->x;
    memset(&y, 0, sizeof y); y.c0.c0 = q->y;
    fp12_mul(&o->x, &x, &w2); fp12_mul(&o->y, &y, &w3);
}
/* line through T and S (or tangent when dbl) evaluated at P; then T <- T + S (or 2T) */
static void line_and_step(fp12_t *l, e12_t *T, const e12_t *S, int dbl, const fp12_t *xP, const fp12_t *yP) {
    fp12_t lam, num, den, t, x3, y3;
    if (dbl) {
        fp12_mul(&num, &T->x, &T->x); fp12_add(&t, &num, &num); fp12_add(&num, &t, &num);
        fp12_add(&den, &T->y, &T->y);
    } else {
        fp12_sub(&num, &S->y, &T->y); fp12_sub(&den, &S->x, &T->x);
    }
    fp12_inv(&den, &den); fp12_mul(&lam, &num, &den);
    /* l = (yP - yT) - lam*(xP - xT) */
    fp12_sub(&t, xP, &T->x); fp12_mul(&t, &lam, &t);
    fp12_sub(l, yP, &T->y); fp12_sub(l, l, &t);
    fp12_mul(&x3, &lam, &lam); fp12_sub(&x3, &x3, &T->x); fp12_sub(&x3, &x3, dbl ? &T->x : &S->x);
    fp12_sub(&t, &T->x, &x3); fp12_mul(&y3, &lam, &t); fp12_sub(&y3, &y3, &T->y);
    T->x = x3; T->y = y3;
}
static void miller_loop(fp12_t *f, const g1a_t *P, const g2a_t *Qa) {
    const uint64_t Z = 0xd201000000010000ULL; /* |z|; the sign only conjugates f, irrelevant for an ==1 check of a product */
    e12_t Q, T; fp12_t xP, yP, l;
    untwist(&Q, Qa); T = Q;
    fp12_from_fp(&xP, &P->x); fp12_from_fp(&yP, &P->y);
    fp12_one(f);
    for (int i = 62; i >= 0; i--) {
        fp12_mul(f, f, f);
        line_and_step(&l, &T, &T, 1, &xP, &yP); fp12_mul(f, f, &l);
        if ((Z >> i) & 1) { line_and_step(&l, &T, &Q, 0, &xP, &yP); fp12_mul(f, f, &l); }
    }
}
static void final_exponentiation(fp12_t *o, const fp12_t *f) {
    static const uint64_t E[32] = { /* (p^6+1)/r */
        0x8739e1cdc0705d6aULL, 0x09a5256de0381a16ULL, 0x9cf0f70a61c791e2ULL, 0x3a09c4497903f76eULL,
        0x2d7271563890f133ULL, 0x224741b36fec7760ULL, 0x338259c22a12bd40ULL, 0x38ee1cd4778e0de7ULL,
        0xc3b5ef4b188a20b0ULL, 0x1d615d49e2764d7bULL, 0x816101ddd076117dULL, 0xf007c01e7ebe3afcULL,
        0x27d7bd90935021c3ULL, 0xc3b5e2f557c0b15fULL, 0x5e886c94c4f82384ULL, 0xee6a95db11e63f56ULL,
        0x2b822f514a9c4f6fULL, 0x12d6a874d21b73daULL, 0x1304275ef499dffbULL, 0x967878febcb95d1fULL,
        0x4744497f8b2f2922ULL, 0x85a2e707f0841855ULL, 0x9f0c50126c802eecULL, 0xfb46e197bd2fa489ULL,
        0x548ce0809bc5f61aULL, 0xcf56fb1573beaa8cULL, 0xad7375a3763bdf7cULL, 0xe0ec9031179bdeccULL,
        0x6579aea83c48c1daULL, 0xdbf85ae664cf5bb3ULL, 0x7b6f235c55ca7566ULL, 0x000028b314877503ULL};
    fp12_t t, fi, acc;
    fp12_conj(&t, f); fp12_inv(&fi, f); fp12_mul(&t, &t, &fi); /* f^(p^6-1) */
    fp12_one(&acc);
    for (int i = 64 * 32 - 1; i >= 0; i--) {
        fp12_mul(&acc, &acc, &acc);
        if ((E[i / 64] >> (i % 64)) & 1) fp12_mul(&acc, &acc, &t);
    }
    *o = acc;
}
int pairing_product_is_one(const g1a_t *P, const g2a_t *Q, size_t n) {
    fp12_t f, g; fp12_one(&f);
    for (size_t i = 0; i < n; i++) {
        if (P[i].inf || Q[i].inf) continue; /* e(O, Q) = e(P, O) = 1 */
        miller_loop(&g, &P[i], &Q[i]);
        fp12_mul(&f, &f, &g);
    }
    final_exponentiation(&g, &f);
    return fp12_is_one(&g);
}
void g2_generator(g2a_t *o) { memset(o, 0, sizeof *o); o->inf = 1; /* unused: the SRS supplies [1]_2 */ }
