/*
 * oracle/field.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * BLS12-381 base field Fp (381 bit) and scalar field Fr (255 bit), Montgomery form,
 * 64-bit limbs with unsigned __int128.  Restates what the reference gets from
 * blstrs::{Fp,Scalar} (crates/cryptography/bls12_381/src/lib.rs:23-42; batch inversion
 * follows crates/cryptography/bls12_381/src/batch_inversion.rs:17-57; the wide reduction
 * follows reduce_bytes_to_scalar_bias, crates/cryptography/bls12_381/src/lib.rs:128-140).
 */
#include "bls.h"
#include <stdlib.h>

typedef unsigned __int128 u128;

static const uint64_t P_MOD[6] = {
    0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
    0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t R_MOD[4] = {
    0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

static uint64_t P_N0, R_N0;
static uint64_t P_R2[6], R_R2[4];
fr_t FR_ONE, FR_ZERO;
fp_t FP_ONE, FP_ZERO;
static int g_init = 0;

/* ---------- generic n-limb helpers (n is a compile-time constant at every call site) ---------- */
static inline int ge_n(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline uint64_t add_n(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t sub_n(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t br = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void mod_add(uint64_t *o, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6]; uint64_t c = add_n(t, a, b, n);
    if (c || ge_n(t, m, n)) sub_n(t, t, m, n);
    memcpy(o, t, 8 * n);
}
static inline void mod_sub(uint64_t *o, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6];
    if (sub_n(t, a, b, n)) add_n(t, t, m, n);
    memcpy(o, t, 8 * n);
}
static inline void mont_mul(uint64_t *o, const uint64_t *a, const uint64_t *b, const uint64_t *m,
                            uint64_t n0, int n) {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * n0;
        c = (u128)q * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) {
            c += (u128)q * m[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[n]; t[n - 1] = (uint64_t)c;
        t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    if (t[n] || ge_n(t, m, n)) sub_n(t, t, m, n);
    memcpy(o, t, 8 * n);
}
static uint64_t compute_n0(uint64_t m0) { /* -m0^{-1} mod 2^64 */
    uint64_t x = 1;
    for (int i = 0; i < 7; i++) x *= 2 - m0 * x;
    return (uint64_t)(0 - x);
}
static void compute_r2(uint64_t *r2, const uint64_t *m, int n) {
    /* 2^(2*64n) mod m by doubling */
    uint64_t t[6] = {0}; t[0] = 1;
    for (int i = 0; i < 2 * 64 * n; i++) mod_add(t, t, t, m, n);
    memcpy(r2, t, 8 * n);
}

void bls_init(void) {
    if (g_init) return;
    P_N0 = compute_n0(P_MOD[0]); R_N0 = compute_n0(R_MOD[0]);
    compute_r2(P_R2, P_MOD, 6); compute_r2(R_R2, R_MOD, 4);
    memset(&FR_ZERO, 0, sizeof FR_ZERO); memset(&FP_ZERO, 0, sizeof FP_ZERO);
    uint64_t one4[4] = {1, 0, 0, 0}, one6[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(FR_ONE.l, one4, R_R2, R_MOD, R_N0, 4);
    mont_mul(FP_ONE.l, one6, P_R2, P_MOD, P_N0, 6);
    g_init = 1;
}

/* ---------------- Fr ---------------- */
void fr_add(fr_t *o, const fr_t *a, const fr_t *b) { mod_add(o->l, a->l, b->l, R_MOD, 4); }
void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { mod_sub(o->l, a->l, b->l, R_MOD, 4); }
void fr_neg(fr_t *o, const fr_t *a) { mod_sub(o->l, FR_ZERO.l, a->l, R_MOD, 4); }
void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { mont_mul(o->l, a->l, b->l, R_MOD, R_N0, 4); }
int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof *a) == 0; }
void fr_from_u64(fr_t *o, uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    mont_mul(o->l, t, R_R2, R_MOD, R_N0, 4);
}
void fr_to_le_canon(uint64_t o[4], const fr_t *a) {
    uint64_t one[4] = {1, 0, 0, 0};
    mont_mul(o, a->l, one, R_MOD, R_N0, 4);
}
static void fr_pow_limbs(fr_t *o, const fr_t *a, const uint64_t *e, int n) {
    fr_t acc = FR_ONE, base = *a;
    for (int i = 0; i < 64 * n; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &base);
        fr_mul(&base, &base, &base);
    }
    *o = acc;
}
void fr_pow_u64(fr_t *o, const fr_t *a, uint64_t e) { fr_pow_limbs(o, a, &e, 1); }
void fr_inv(fr_t *o, const fr_t *a) {
    uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0};
    sub_n(e, R_MOD, two, 4);
    fr_pow_limbs(o, a, e, 4);
}
int fr_from_be(fr_t *o, const uint8_t b[32]) {
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j];
        t[i] = v;
    }
    if (ge_n(t, R_MOD, 4)) return -1;
    mont_mul(o->l, t, R_R2, R_MOD, R_N0, 4);
    return 0;
}
void fr_from_be_reduce(fr_t *o, const uint8_t b[32]) {
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j];
        t[i] = v;
    }
    /* 2^256 < 3r, so at most two subtractions */
    while (ge_n(t, R_MOD, 4)) sub_n(t, t, R_MOD, 4);
    mont_mul(o->l, t, R_R2, R_MOD, R_N0, 4);
}
void fr_to_be(uint8_t b[32], const fr_t *a) {
    uint64_t t[4]; fr_to_le_canon(t, a);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(t[i] >> (56 - 8 * j));
}
void fr_root_of_unity(fr_t *o, unsigned log_n) {
    /* 7^((r-1)/2^32) is blstrs' ROOT_OF_UNITY; square down as Domain::compute_generator_for_size does
       (crates/cryptography/polynomial/src/domain.rs:84-101). */
    uint64_t e[4], one[4] = {1, 0, 0, 0};
    sub_n(e, R_MOD, one, 4);
    /* e = (r-1) >> 32 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 32) | (i < 3 ? (e[i + 1] << 32) : 0);
    fr_t seven; fr_from_u64(&seven, 7);
    fr_t w; fr_pow_limbs(&w, &seven, e, 4);
    for (unsigned i = log_n; i < 32; i++) fr_mul(&w, &w, &w);
    *o = w;
}
void fr_batch_inverse(fr_t *v, size_t n) {
    if (n == 0) return;
    fr_t *pre = (fr_t *)malloc(n * sizeof(fr_t));
    fr_t acc = FR_ONE;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; fr_mul(&acc, &acc, &v[i]); }
    fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        fr_t t; fr_mul(&t, &acc, &pre[i]);
        fr_mul(&acc, &acc, &v[i]);
        v[i] = t;
    }
    free(pre);
}

/* ---------------- Fp ---------------- */
void fp_add(fp_t *o, const fp_t *a, const fp_t *b) { mod_add(o->l, a->l, b->l, P_MOD, 6); }
void fp_sub(fp_t *o, const fp_t *a, const fp_t *b) { mod_sub(o->l, a->l, b->l, P_MOD, 6); }
void fp_neg(fp_t *o, const fp_t *a) { mod_sub(o->l, FP_ZERO.l, a->l, P_MOD, 6); }
/* The portable product-scanning form above is the CHECKER.  The timed native build of bench.py's cpu_baseline leg (and only it:
 * -DORACLE_ADX -march=native) multiplies in Fp with the BMI2 / ADX instructions a production CPU library uses -- mulx for the
 * 64 x 64 -> 128 products, two independent carry chains per row (the compiler schedules them as adcx / adox or adc) -- so that the one
 * CPU number beside the GPU's is not softer than it needs to be (blst-class assembly is 1.3-2x the __int128 form).
 * oracle_fp_mul_selftest compares the two forms on random pairs (tests/test_oracle_units.py: 10^6 of them). */
void fp_mul_portable(fp_t *o, const fp_t *a, const fp_t *b) { mont_mul(o->l, a->l, b->l, P_MOD, P_N0, 6); }
#if defined(ORACLE_ADX) && defined(__ADX__) && defined(__BMI2__)
#include <immintrin.h>
#define ORACLE_ADX_ACTIVE 1
static inline void mont_mul6_adx(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    typedef unsigned long long ull;
    ull t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7;
    const ull p0 = P_MOD[0], p1 = P_MOD[1], p2 = P_MOD[2], p3 = P_MOD[3], p4 = P_MOD[4], p5 = P_MOD[5], n0 = P_N0;
    for (int i = 0; i < 6; i++) {
        ull l0, l1, l2, l3, l4, l5, h0, h1, h2, h3, h4, h5;
        unsigned char c, d;
        const ull bi = b[i];
        /* t += a * b[i]: the low halves on one carry chain, the high halves on another */
        l0 = _mulx_u64(a[0], bi, &h0); l1 = _mulx_u64(a[1], bi, &h1); l2 = _mulx_u64(a[2], bi, &h2);
        l3 = _mulx_u64(a[3], bi, &h3); l4 = _mulx_u64(a[4], bi, &h4); l5 = _mulx_u64(a[5], bi, &h5);
        c = _addcarryx_u64(0, t0, l0, &t0); c = _addcarryx_u64(c, t1, l1, &t1); c = _addcarryx_u64(c, t2, l2, &t2);
        c = _addcarryx_u64(c, t3, l3, &t3); c = _addcarryx_u64(c, t4, l4, &t4); c = _addcarryx_u64(c, t5, l5, &t5);
        c = _addcarryx_u64(c, t6, 0, &t6);
        t7 = c;
        d = _addcarryx_u64(0, t1, h0, &t1); d = _addcarryx_u64(d, t2, h1, &t2); d = _addcarryx_u64(d, t3, h2, &t3);
        d = _addcarryx_u64(d, t4, h3, &t4); d = _addcarryx_u64(d, t5, h4, &t5); d = _addcarryx_u64(d, t6, h5, &t6);
        t7 += d;
        /* t += q * p with q = t0 * n0: t0 becomes 0; shift one limb down */
        const ull q = t0 * n0;
        l0 = _mulx_u64(q, p0, &h0); l1 = _mulx_u64(q, p1, &h1); l2 = _mulx_u64(q, p2, &h2);
        l3 = _mulx_u64(q, p3, &h3); l4 = _mulx_u64(q, p4, &h4); l5 = _mulx_u64(q, p5, &h5);
        c = _addcarryx_u64(0, t0, l0, &t0); c = _addcarryx_u64(c, t1, l1, &t1); c = _addcarryx_u64(c, t2, l2, &t2);
        c = _addcarryx_u64(c, t3, l3, &t3); c = _addcarryx_u64(c, t4, l4, &t4); c = _addcarryx_u64(c, t5, l5, &t5);
        c = _addcarryx_u64(c, t6, 0, &t6);
        t7 += c;
        d = _addcarryx_u64(0, t1, h0, &t0); d = _addcarryx_u64(d, t2, h1, &t1); d = _addcarryx_u64(d, t3, h2, &t2);
        d = _addcarryx_u64(d, t4, h3, &t3); d = _addcarryx_u64(d, t5, h4, &t4); d = _addcarryx_u64(d, t6, h5, &t5);
        t6 = t7 + d;
    }
    uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
    if (t6 || ge_n(t, P_MOD, 6)) sub_n(t, t, P_MOD, 6);
    memcpy(o, t, 48);
}
void fp_mul(fp_t *o, const fp_t *a, const fp_t *b) { mont_mul6_adx(o->l, a->l, b->l); }
#else
#define ORACLE_ADX_ACTIVE 0
void fp_mul(fp_t *o, const fp_t *a, const fp_t *b) { mont_mul(o->l, a->l, b->l, P_MOD, P_N0, 6); }
#endif
int oracle_fp_mul_uses_adx(void) { return ORACLE_ADX_ACTIVE; }
/* fp_mul (whichever form this build uses) against the portable form on n pseudo-random pairs of values below p, chained so that
 * every product feeds the next pair; returns the number of mismatches */
long oracle_fp_mul_selftest(long n, uint64_t seed) {
    bls_init();
    fp_t x, y, r1, r2;
    uint64_t st = seed | 1;
    for (int i = 0; i < 6; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x.l[i] = st; st ^= st << 13; st ^= st >> 7; st ^= st << 17; y.l[i] = st; }
    x.l[5] &= 0x0fffffffffffffffULL; y.l[5] &= 0x0fffffffffffffffULL;
    long bad = 0;
    for (long k = 0; k < n; k++) {
        fp_mul(&r1, &x, &y);
        fp_mul_portable(&r2, &x, &y);
        if (memcmp(&r1, &r2, sizeof r1)) bad++;
        x = y;
        y = r2;
        if ((k & 1023) == 1023) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; y.l[0] ^= st; y.l[5] &= 0x0fffffffffffffffULL; }
    }
    return bad;
}
void fp_sqr(fp_t *o, const fp_t *a) { fp_mul(o, a, a); }
int fp_is_zero(const fp_t *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0;
}
int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof *a) == 0; }
void fp_from_u64(fp_t *o, uint64_t v) {
    uint64_t t[6] = {v, 0, 0, 0, 0, 0};
    mont_mul(o->l, t, P_R2, P_MOD, P_N0, 6);
}
static void fp_pow_limbs(fp_t *o, const fp_t *a, const uint64_t *e, int n) {
    fp_t acc = FP_ONE;
    int started = 0;
    for (int i = 64 * n - 1; i >= 0; i--) {
        if (started) fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) { fp_mul(&acc, &acc, a); started = 1; }
    }
    *o = acc;
}
void fp_inv(fp_t *o, const fp_t *a) {
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    sub_n(e, P_MOD, two, 6);
    fp_pow_limbs(o, a, e, 6);
}
int fp_sqrt(fp_t *o, const fp_t *a) {
    /* p = 3 mod 4: candidate a^((p+1)/4) */
    uint64_t e[6], one[6] = {1, 0, 0, 0, 0, 0};
    add_n(e, P_MOD, one, 6); /* p+1 < 2^384 */
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? (e[i + 1] << 62) : 0);
    fp_t r, c; fp_pow_limbs(&r, a, e, 6);
    fp_sqr(&c, &r);
    *o = r;
    return fp_eq(&c, a);
}
static void fp_to_canon(uint64_t t[6], const fp_t *a) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(t, a->l, one, P_MOD, P_N0, 6);
}
int fp_from_be(fp_t *o, const uint8_t b[48]) {
    uint64_t t[6];
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(5 - i) * 8 + j];
        t[i] = v;
    }
    if (ge_n(t, P_MOD, 6)) return -1;
    mont_mul(o->l, t, P_R2, P_MOD, P_N0, 6);
    return 0;
}
void fp_to_be(uint8_t b[48], const fp_t *a) {
    uint64_t t[6]; fp_to_canon(t, a);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(t[i] >> (56 - 8 * j));
}
int fp_is_lex_largest(const fp_t *a) {
    /* a > (p-1)/2 */
    uint64_t t[6], h[6];
    fp_to_canon(t, a);
    for (int i = 0; i < 6; i++) h[i] = (P_MOD[i] >> 1) | (i < 5 ? (P_MOD[i + 1] << 63) : 0); /* (p-1)/2 */
    /* t > h  <=> !(h >= t) */
    return !ge_n(h, t, 6);
}
