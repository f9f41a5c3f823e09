/*
 * oracle/kzg.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; see kzg_oracle.h).
 *
 * Plain-C restatement of the reference's hot path, function by function:
 *   Domain / fft_inplace / dit        crates/cryptography/polynomial/src/domain.rs:41-223, fft.rs:46-177,223-275
 *   Booth digits                      crates/cryptography/bls12_381/src/booth_encoding.rs:4-46
 *   fixed-base window MSM             crates/cryptography/bls12_381/src/fixed_base_msm_window.rs:54-168
 *   batched affine addition           crates/cryptography/bls12_381/src/batch_addition.rs:14-39,142-232
 *   FK20 prover                       crates/cryptography/kzg_multi_open/src/fk20/prover.rs:64-228
 *   h-polynomial commitments          .../fk20/h_poly.rs:18-68, toeplitz.rs:132-144, batch_toeplitz.rs:34-126
 *   FK20 batch verifier               .../fk20/verifier.rs:58-384, cosets.rs:89-112
 *   Reed-Solomon erasure recovery     crates/cryptography/erasure_codes/src/reed_solomon.rs:111-131,220-262,332-384
 *   recovery glue                     crates/eip7594/src/recovery.rs:22-151, .../fk20/cosets.rs:141-198
 *   DASContext entry points           crates/eip7594/src/prover.rs:100-171, verifier.rs:49-164
 *   (de)serialisation                 crates/serialization/src/lib.rs:36-156
 */
#include "kzg_oracle.h"
#include "bls.h"
#include <stdlib.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define N_BLOB 4096
#define N_EXT 8192
#define N_CELLS 128
#define CELL_LEN 64
#define BYTES_PER_CELL 2048
#define WBITS 8

/* ------------------------------------------------------------------ */
/* bit reversal (fft.rs:223-253)                                        */
static unsigned log2_pow2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }
static size_t reverse_bits(size_t v, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
#define DEFINE_BRP(NAME, T)                                        \
    static void NAME(T *a, size_t n) {                             \
        unsigned ln = log2_pow2(n);                                \
        for (size_t k = 0; k < n; k++) {                           \
            size_t rk = reverse_bits(k, ln);                       \
            if (k < rk) { T t = a[k]; a[k] = a[rk]; a[rk] = t; }   \
        }                                                          \
    }
DEFINE_BRP(brp_fr, fr_t)
DEFINE_BRP(brp_g1, g1_t)

/* ------------------------------------------------------------------ */
/* Domain (domain.rs:41-81)                                             */
typedef struct {
    size_t n; unsigned log_n;
    fr_t *roots;
    fr_t generator, generator_inv, size_inv;
    fr_t *omegas, *tw_bo, *omegas_inv, *tw_inv_bo;
} domain_t;

static void precompute_omegas(fr_t *out, const fr_t *omega, size_t n) { /* fft.rs:261-266 */
    unsigned ln = log2_pow2(n);
    for (unsigned s = 0; s < ln; s++) fr_pow_u64(&out[s], omega, (uint64_t)(n / ((size_t)1 << (s + 1))));
}
static void precompute_twiddles_bo(fr_t *out, const fr_t *omega, size_t n) { /* fft.rs:269-275 */
    fr_t t = FR_ONE;
    for (size_t i = 0; i < n / 2; i++) { out[i] = t; fr_mul(&t, &t, omega); }
    brp_fr(out, n / 2);
}
static void domain_init(domain_t *d, size_t n) {
    d->n = n; d->log_n = log2_pow2(n);
    fr_root_of_unity(&d->generator, d->log_n);
    fr_inv(&d->generator_inv, &d->generator);
    fr_t sz; fr_from_u64(&sz, (uint64_t)n); fr_inv(&d->size_inv, &sz);
    d->roots = malloc(n * sizeof(fr_t));
    d->roots[0] = FR_ONE;
    for (size_t i = 1; i < n; i++) fr_mul(&d->roots[i], &d->roots[i - 1], &d->generator);
    d->omegas = malloc((d->log_n + 1) * sizeof(fr_t));
    d->omegas_inv = malloc((d->log_n + 1) * sizeof(fr_t));
    d->tw_bo = malloc((n / 2 + 1) * sizeof(fr_t));
    d->tw_inv_bo = malloc((n / 2 + 1) * sizeof(fr_t));
    precompute_omegas(d->omegas, &d->generator, n);
    precompute_twiddles_bo(d->tw_bo, &d->generator, n);
    precompute_omegas(d->omegas_inv, &d->generator_inv, n);
    precompute_twiddles_bo(d->tw_inv_bo, &d->generator_inv, n);
}
static void domain_free(domain_t *d) {
    free(d->roots); free(d->omegas); free(d->omegas_inv); free(d->tw_bo); free(d->tw_inv_bo);
}

/* ------------------------------------------------------------------ */
/* fft_inplace for Fr (fft.rs:46-177)                                   */
static fr_t FR_MINUS_ONE;
static inline void dit_fr(fr_t *a, fr_t *b, const fr_t *tw) { /* fft.rs:164-177 */
    fr_t t;
    if (fr_eq(tw, &FR_ONE)) t = *b;
    else if (fr_eq(tw, &FR_MINUS_ONE)) fr_neg(&t, b);
    else if (fr_is_zero(b)) t = FR_ZERO;
    else fr_mul(&t, b, tw);
    *b = *a;
    fr_add(a, a, &t);
    fr_sub(b, b, &t);
}
static void dit_layer_fr(fr_t *blocks, size_t len, size_t half, const fr_t *omega) { /* fft.rs:90-112 */
    for (size_t s = 0; s < len; s += 2 * half) {
        fr_t tw = FR_ONE;
        for (size_t k = 0; k < half; k++) {
            dit_fr(&blocks[s + k], &blocks[s + half + k], &tw);
            fr_mul(&tw, &tw, omega);
        }
    }
}
static void dit_layer_bo_fr(fr_t *blocks, size_t len, size_t half, const fr_t *tw_bo) { /* fft.rs:138-158 */
    size_t bi = 0;
    for (size_t s = 0; s < len; s += 2 * half, bi++)
        for (size_t k = 0; k < half; k++) dit_fr(&blocks[s + k], &blocks[s + half + k], &tw_bo[bi]);
}
static void fft_inplace_fr(const fr_t *omegas, const fr_t *tw_bo, fr_t *v, size_t n, int par) {
    unsigned log_n = log2_pow2(n), mid = (log_n + 1) / 2;
    brp_fr(v, n);
    size_t c1 = (size_t)1 << mid;
    /* first_half: fft.rs:71-82 (maybe_par_chunks_mut) */
#pragma omp parallel for if (par) schedule(static)
    for (size_t c = 0; c < n / c1; c++)
        for (unsigned layer = 0; layer < mid; layer++)
            dit_layer_fr(v + c * c1, c1, (size_t)1 << layer, &omegas[layer]);
    brp_fr(v, n);
    size_t c2 = (size_t)1 << (log_n - mid);
    /* second_half: fft.rs:118-132 */
#pragma omp parallel for if (par) schedule(static)
    for (size_t c = 0; c < n / c2; c++)
        for (unsigned layer = mid; layer < log_n; layer++)
            dit_layer_bo_fr(v + c * c2, c2, (size_t)1 << (log_n - 1 - layer), tw_bo + (c << (layer - mid)));
    brp_fr(v, n);
}
/* Domain::fft_scalars / ifft_scalars / coset variants (domain.rs:117-142,199-223); v has d->n entries */
static void fft_scalars(const domain_t *d, fr_t *v, int par) { fft_inplace_fr(d->omegas, d->tw_bo, v, d->n, par); }
static void ifft_scalars(const domain_t *d, fr_t *v, int par) {
    fft_inplace_fr(d->omegas_inv, d->tw_inv_bo, v, d->n, par);
    for (size_t i = 0; i < d->n; i++) fr_mul(&v[i], &v[i], &d->size_inv);
}
static void coset_fft_scalars(const domain_t *d, fr_t *v, const fr_t *gen, int par) {
    fr_t s = FR_ONE;
    for (size_t i = 0; i < d->n; i++) { fr_mul(&v[i], &v[i], &s); fr_mul(&s, &s, gen); }
    fft_scalars(d, v, par);
}
static void coset_ifft_scalars(const domain_t *d, fr_t *v, const fr_t *gen_inv, int par) {
    ifft_scalars(d, v, par);
    fr_t s = FR_ONE;
    for (size_t i = 0; i < d->n; i++) { fr_mul(&v[i], &v[i], &s); fr_mul(&s, &s, gen_inv); }
}

/* ------------------------------------------------------------------ */
/* fft_inplace for G1Projective (same network; `b * twiddle` is a scalar multiplication)  */
static inline void dit_g1(g1_t *a, g1_t *b, const fr_t *tw) {
    g1_t t;
    if (fr_eq(tw, &FR_ONE)) t = *b;
    else if (fr_eq(tw, &FR_MINUS_ONE)) g1_neg(&t, b);
    else if (g1_is_inf(b)) g1_set_inf(&t);
    else g1_mul(&t, b, tw);
    g1_t a0 = *a;
    g1_add(a, &a0, &t);
    g1_sub(b, &a0, &t);
}
static void dit_layer_g1(g1_t *blocks, size_t len, size_t half, const fr_t *omega) {
    for (size_t s = 0; s < len; s += 2 * half) {
        fr_t tw = FR_ONE;
        for (size_t k = 0; k < half; k++) {
            dit_g1(&blocks[s + k], &blocks[s + half + k], &tw);
            fr_mul(&tw, &tw, omega);
        }
    }
}
static void dit_layer_bo_g1(g1_t *blocks, size_t len, size_t half, const fr_t *tw_bo) {
    size_t bi = 0;
    for (size_t s = 0; s < len; s += 2 * half, bi++)
        for (size_t k = 0; k < half; k++) dit_g1(&blocks[s + k], &blocks[s + half + k], &tw_bo[bi]);
}
static void fft_inplace_g1(const fr_t *omegas, const fr_t *tw_bo, g1_t *v, size_t n, int par) {
    unsigned log_n = log2_pow2(n), mid = (log_n + 1) / 2;
    brp_g1(v, n);
    size_t c1 = (size_t)1 << mid;
#pragma omp parallel for if (par) schedule(static)
    for (size_t c = 0; c < n / c1; c++)
        for (unsigned layer = 0; layer < mid; layer++)
            dit_layer_g1(v + c * c1, c1, (size_t)1 << layer, &omegas[layer]);
    brp_g1(v, n);
    size_t c2 = (size_t)1 << (log_n - mid);
#pragma omp parallel for if (par) schedule(static)
    for (size_t c = 0; c < n / c2; c++)
        for (unsigned layer = mid; layer < log_n; layer++)
            dit_layer_bo_g1(v + c * c2, c2, (size_t)1 << (log_n - 1 - layer), tw_bo + (c << (layer - mid)));
    brp_g1(v, n);
}
static void fft_g1(const domain_t *d, g1_t *v, int par) { fft_inplace_g1(d->omegas, d->tw_bo, v, d->n, par); }
/* Domain::ifft_g1_take_n (domain.rs:172-194): only the first take_n outputs are scaled/returned */
static void ifft_g1_take_n(const domain_t *d, g1_t *v, size_t take_n, int par) {
    fft_inplace_g1(d->omegas_inv, d->tw_inv_bo, v, d->n, par);
    for (size_t i = 0; i < take_n; i++) g1_mul(&v[i], &v[i], &d->size_inv);
}

/* ------------------------------------------------------------------ */
/* Booth digit (booth_encoding.rs:4-46); el = 32 little-endian bytes    */
static int get_booth_index(unsigned window_index, unsigned window_size, const uint8_t *el) {
    unsigned skip_bits = window_index * window_size; skip_bits = skip_bits ? skip_bits - 1 : 0;
    unsigned skip_bytes = skip_bits / 8;
    uint8_t v[4] = {0, 0, 0, 0};
    for (unsigned i = 0; i < 4 && skip_bytes + i < 32; i++) v[i] = el[skip_bytes + i];
    uint32_t tmp = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    if (window_index == 0) tmp <<= 1;
    tmp >>= skip_bits - skip_bytes * 8;
    tmp &= (1u << (window_size + 1)) - 1;
    int sign = (tmp & (1u << window_size)) == 0;
    tmp = (tmp + 1) >> 1;
    if (sign) return (int)tmp;
    return -(int)((~(tmp - 1)) & ((1u << window_size) - 1));
}

/* ------------------------------------------------------------------ */
/* multi_batch_addition_binary_tree_stride (batch_addition.rs:142-232).
   buckets[b] holds cnt[b] affine points (no identities); sums[b] receives the Jacobian sum.
   Unlike the reference (which documents P = -Q as unhandled, batch_addition.rs:27-39) the
   restatement is correct there: such a pair sums to the identity and is dropped. */
#define BATCH_INVERSE_THRESHOLD 16
static void multi_batch_addition(g1a_t **buckets, size_t *cnt, size_t nb, g1_t *sums) {
    size_t total = 0, maxlen = 0;
    for (size_t b = 0; b < nb; b++) { g1_set_inf(&sums[b]); total += cnt[b]; if (cnt[b] > maxlen) maxlen = cnt[b]; }
    fp_t *den = malloc((total / 2 + 1) * sizeof(fp_t));
    fp_t *pre = malloc((total / 2 + 1) * sizeof(fp_t));
    for (;;) {
        size_t work = 0;
        for (size_t b = 0; b < nb; b++) work += cnt[b] / 2;
        if (work <= BATCH_INVERSE_THRESHOLD) break;
        for (size_t b = 0; b < nb; b++)
            if (cnt[b] & 1) { cnt[b]--; g1_add_affine(&sums[b], &sums[b], &buckets[b][cnt[b]]); }
        /* denominators: choose_add_or_double (batch_addition.rs:33-39) */
        size_t nd = 0;
        for (size_t b = 0; b < nb; b++)
            for (size_t i = 0; i + 1 < cnt[b]; i += 2) {
                const g1a_t *p1 = &buckets[b][i], *p2 = &buckets[b][i + 1];
                if (fp_eq(&p1->x, &p2->x)) {
                    if (fp_eq(&p1->y, &p2->y) && !fp_is_zero(&p1->y)) fp_add(&den[nd], &p2->y, &p2->y);
                    else den[nd] = FP_ONE; /* P + (-P): pair cancels */
                } else fp_sub(&den[nd], &p2->x, &p1->x);
                nd++;
            }
        /* batch_inverse_scratch_pad (batch_inversion.rs:17-57) */
        fp_t acc = FP_ONE;
        for (size_t i = 0; i < nd; i++) { pre[i] = acc; fp_mul(&acc, &acc, &den[i]); }
        fp_inv(&acc, &acc);
        for (size_t i = nd; i-- > 0;) {
            fp_t t; fp_mul(&t, &acc, &pre[i]); fp_mul(&acc, &acc, &den[i]); den[i] = t;
        }
        /* point_add_double (batch_addition.rs:14-25) */
        size_t off = 0;
        for (size_t b = 0; b < nb; b++) {
            size_t out = 0;
            for (size_t i = 0; i + 1 < cnt[b]; i += 2, off++) {
                g1a_t p1 = buckets[b][i], p2 = buckets[b][i + 1];
                fp_t lambda, x, y, t;
                if (fp_eq(&p1.x, &p2.x)) {
                    if (!(fp_eq(&p1.y, &p2.y) && !fp_is_zero(&p1.y))) continue; /* cancels to identity */
                    fp_sqr(&t, &p1.x); fp_add(&lambda, &t, &t); fp_add(&lambda, &lambda, &t);
                    fp_mul(&lambda, &lambda, &den[off]);
                } else {
                    fp_sub(&t, &p2.y, &p1.y); fp_mul(&lambda, &t, &den[off]);
                }
                fp_sqr(&x, &lambda); fp_sub(&x, &x, &p1.x); fp_sub(&x, &x, &p2.x);
                fp_sub(&t, &p1.x, &x); fp_mul(&y, &lambda, &t); fp_sub(&y, &y, &p1.y);
                buckets[b][out].x = x; buckets[b][out].y = y; buckets[b][out].inf = 0; out++;
            }
            cnt[b] = out;
        }
    }
    for (size_t b = 0; b < nb; b++)
        for (size_t i = 0; i < cnt[b]; i++) g1_add_affine(&sums[b], &sums[b], &buckets[b][i]);
    free(den); free(pre);
}

/* ------------------------------------------------------------------ */
struct oracle_ctx {
    int use_precomp, threads;
    g1a_t *g1s;      /* 4096 monomial SRS points */
    g2a_t tau_pow_n, neg_g2_gen, tau_g2;  /* [tau^64]_2, -[1]_2, [tau]_2 */
    domain_t d128, d4096, d8192, d64;
    g1a_t *fft_srs;  /* [128][64]: row j = the 64 bases of MSM j (batch_toeplitz.rs:61) */
    g1a_t *tables;   /* [128][64][128] width-8 tables (fixed_base_msm_window.rs:69-82) or NULL */
    fr_t coset_gens_br[N_CELLS], coset_gens_inv_br[N_CELLS], coset_gens_pow_n_br[N_CELLS];
    fr_t rs_coset_gen, rs_coset_gen_inv;
};

/* FixedBaseMSMPrecompWindow::msm (fixed_base_msm_window.rs:102-168), 64 scalars */
static void fixed_base_msm_precomp(const g1a_t *table /*[64][128]*/, const fr_t *scalars, size_t n, g1_t *out) {
    const unsigned nwin = 255 / WBITS + 1;
    uint8_t (*sb)[32] = malloc(n * 32);
    for (size_t i = 0; i < n; i++) {
        uint64_t c[4]; fr_to_le_canon(c, &scalars[i]);
        for (int k = 0; k < 32; k++) sb[i][k] = (uint8_t)(c[k / 8] >> (8 * (k % 8)));
    }
    g1a_t *store = malloc((size_t)nwin * n * sizeof(g1a_t));
    g1a_t *buckets[64]; size_t cnt[64];
    for (unsigned w = 0; w < nwin; w++) {
        buckets[w] = store + (size_t)w * n; cnt[w] = 0;
        for (size_t i = 0; i < n; i++) {
            int idx = get_booth_index(w, WBITS, sb[i]);
            if (idx == 0) continue;
            g1a_t p = table[i * (1u << (WBITS - 1)) + (size_t)(abs(idx) - 1)];
            if (p.inf) continue; /* identity base (padded SRS vectors can FFT to identity only in theory) */
            if (idx < 0) fp_neg(&p.y, &p.y);
            buckets[w][cnt[w]++] = p;
        }
    }
    g1_t sums[64];
    multi_batch_addition(buckets, cnt, nwin, sums);
    g1_t result = sums[nwin - 1];
    for (int w = (int)nwin - 2; w >= 0; w--) {
        for (unsigned d = 0; d < WBITS; d++) g1_dbl(&result, &result);
        g1_add(&result, &result, &sums[w]);
    }
    *out = result;
    free(store); free(sb);
}

/* ------------------------------------------------------------------ */
static int load_srs(oracle_ctx *c, const uint8_t *srs, size_t len) {
    if (len < 16 || memcmp(srs, "KZGSRS01", 8)) return -1;
    uint32_t n1, n2; memcpy(&n1, srs + 8, 4); memcpy(&n2, srs + 12, 4);
    if (n1 != N_BLOB || n2 != 65 || len != 16 + (size_t)n1 * 48 + (size_t)n2 * 96) return -1;
    c->g1s = malloc(N_BLOB * sizeof(g1a_t));
    int bad = 0;
    /* unchecked decompression, as trusted_setup/src/lib.rs:80-86 */
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < N_BLOB; i++) bad |= g1_decompress(&c->g1s[i], srs + 16 + 48 * (size_t)i, 0) != 0;
    if (bad) return -1;
    const uint8_t *g2 = srs + 16 + 48 * (size_t)N_BLOB;
    g2a_t gen;
    if (g2_decompress(&gen, g2)) return -1;              /* [1]_2 */
    if (g2_decompress(&c->tau_pow_n, g2 + 96 * CELL_LEN)) return -1; /* [tau^64]_2 (verifier.rs:88) */
    g2_neg(&c->neg_g2_gen, &gen);                          /* -[1]_2 (verifier.rs:90) */
    if (g2_decompress(&c->tau_g2, g2 + 96)) return -1;     /* [tau]_2 (eip4844 verification key) */
    return 0;
}

oracle_ctx *oracle_ctx_new(const uint8_t *srs, size_t srs_len, int use_precomp, int threads) {
    bls_init();
    fr_neg(&FR_MINUS_ONE, &FR_ONE);
    oracle_ctx *c = calloc(1, sizeof *c);
    c->use_precomp = use_precomp; c->threads = threads < 1 ? 1 : threads;
#ifdef _OPENMP
    omp_set_num_threads(c->threads);
#endif
    if (load_srs(c, srs, srs_len)) { free(c); return NULL; }
    domain_init(&c->d128, 128); domain_init(&c->d4096, N_BLOB); domain_init(&c->d8192, N_EXT); domain_init(&c->d64, CELL_LEN);

    /* FK20Prover::new (prover.rs:64-125): srs_truncated = reverse(g1s).skip(64); vectors = take_every_nth(.,64) */
    g1a_t *trunc = malloc((N_BLOB - CELL_LEN) * sizeof(g1a_t));
    for (int i = 0; i < N_BLOB - CELL_LEN; i++) trunc[i] = c->g1s[N_BLOB - 1 - CELL_LEN - i];
    c->fft_srs = malloc(128 * 64 * sizeof(g1a_t));
    /* BatchToeplitzMatrixVecMul::new (batch_toeplitz.rs:34-78): FFT_128 of each zero-padded vector, normalise, transpose */
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < CELL_LEN; i++) {
        g1_t v[128]; g1a_t a[128];
        size_t k = 0;
        for (size_t m = (size_t)i; m < (size_t)(N_BLOB - CELL_LEN); m += CELL_LEN) g1_from_affine(&v[k++], &trunc[m]);
        for (; k < 128; k++) g1_set_inf(&v[k]);
        fft_g1(&c->d128, v, 0);
        g1_batch_normalize(a, v, 128);
        for (int j = 0; j < 128; j++) c->fft_srs[j * 64 + i] = a[j];
    }
    free(trunc);
    if (use_precomp) {
        const size_t T = 1u << (WBITS - 1);
        c->tables = malloc((size_t)128 * 64 * T * sizeof(g1a_t));
#pragma omp parallel for schedule(dynamic, 16)
        for (int q = 0; q < 128 * 64; q++) {
            g1_t cur, tab[1u << (WBITS - 1)];
            g1_from_affine(&cur, &c->fft_srs[q]);
            for (size_t t = 0; t < T; t++) { tab[t] = cur; g1_add_affine(&cur, &cur, &c->fft_srs[q]); }
            g1_batch_normalize(c->tables + (size_t)q * T, tab, T);
        }
    }
    /* FK20Verifier::new (verifier.rs:58-108), coset_gens (cosets.rs:89-112) */
    for (size_t i = 0; i < N_CELLS; i++) {
        fr_pow_u64(&c->coset_gens_br[i], &c->d8192.generator, reverse_bits(i, 7));
        fr_inv(&c->coset_gens_inv_br[i], &c->coset_gens_br[i]);
        fr_pow_u64(&c->coset_gens_pow_n_br[i], &c->coset_gens_br[i], CELL_LEN);
    }
    /* ReedSolomon::new (reed_solomon.rs:111-131): coset generator = MULTIPLICATIVE_GENERATOR = 7 */
    fr_from_u64(&c->rs_coset_gen, 7); fr_inv(&c->rs_coset_gen_inv, &c->rs_coset_gen);
    return c;
}
void oracle_ctx_free(oracle_ctx *c) {
    if (!c) return;
    domain_free(&c->d128); domain_free(&c->d4096); domain_free(&c->d8192); domain_free(&c->d64);
    free(c->g1s); free(c->fft_srs); free(c->tables); free(c);
}

/* ------------------------------------------------------------------ */
/* deserialize_blob_to_scalars (serialization/src/lib.rs:36-63) */
static int deserialize_scalars(fr_t *out, const uint8_t *bytes, size_t n) {
    for (size_t i = 0; i < n; i++) if (fr_from_be(&out[i], bytes + 32 * i)) return ORACLE_ERR_SCALAR;
    return ORACLE_OK;
}
/* Input::Data -> PolyCoeff (prover.rs:177-180) */
static void data_to_poly(const oracle_ctx *c, fr_t *v) {
    brp_fr(v, N_BLOB);
    ifft_scalars(&c->d4096, v, c->threads > 1);
}

/* compute_h_poly_commitments (h_poly.rs:18-57) + sum_matrix_vector_mul (batch_toeplitz.rs:85-126) */
static void compute_h_poly_commitments(const oracle_ctx *c, const fr_t *poly, g1_t *h /*128, first 64 valid*/) {
    int par = c->threads > 1;
    fr_t *rev = malloc(N_BLOB * sizeof(fr_t));
    for (int i = 0; i < N_BLOB; i++) rev[i] = poly[N_BLOB - 1 - i];           /* polynomial.reverse() */
    fr_t *col_ffts = malloc(64 * 128 * sizeof(fr_t));
#pragma omp parallel for if (par) schedule(static)
    for (int i = 0; i < CELL_LEN; i++) {
        fr_t *cm = col_ffts + (size_t)i * 128, row[64];
        for (int k = 0; k < 64; k++) row[k] = rev[i + 64 * k];                 /* take_every_nth */
        /* CirculantMatrix::from_toeplitz (toeplitz.rs:132-144): col = [row0,0..], ext = [0,row63,..,row1] */
        cm[0] = row[0];
        for (int k = 1; k < 64; k++) cm[k] = FR_ZERO;
        cm[64] = FR_ZERO;
        for (int k = 1; k < 64; k++) cm[64 + k] = row[64 - k];
        fft_scalars(&c->d128, cm, 0);
    }
#pragma omp parallel for if (par) schedule(dynamic)
    for (int j = 0; j < 128; j++) {
        fr_t sc[64];
        for (int i = 0; i < 64; i++) sc[i] = col_ffts[(size_t)i * 128 + j];   /* transpose */
        if (c->use_precomp) fixed_base_msm_precomp(c->tables + (size_t)j * 64 * (1u << (WBITS - 1)), sc, 64, &h[j]);
        else g1_msm(&h[j], c->fft_srs + (size_t)j * 64, sc, 64);              /* FixedBaseMSM::NoPrecomp -> g1_lincomb */
    }
    ifft_g1_take_n(&c->d128, h, CELL_LEN, par);
    free(col_ffts); free(rev);
}

/* compute_coset_evaluations (prover.rs:158-165) + serialize_cells */
static void compute_cells_from_poly(const oracle_ctx *c, const fr_t *poly, uint8_t *cells) {
    fr_t *ev = malloc(N_EXT * sizeof(fr_t));
    memcpy(ev, poly, N_BLOB * sizeof(fr_t));
    for (int i = N_BLOB; i < N_EXT; i++) ev[i] = FR_ZERO;
    fft_scalars(&c->d8192, ev, c->threads > 1);
    brp_fr(ev, N_EXT);
    for (int i = 0; i < N_EXT; i++) fr_to_be(cells + 32 * (size_t)i, &ev[i]);
    free(ev);
}
/* compute_multi_opening_proofs_poly_coeff (prover.rs:203-228) */
static void proofs_and_cells_from_poly(const oracle_ctx *c, const fr_t *poly, uint8_t *cells, uint8_t *proofs) {
    g1_t h[128]; g1a_t pa[128];
    compute_h_poly_commitments(c, poly, h);
    for (int i = CELL_LEN; i < 128; i++) g1_set_inf(&h[i]);
    fft_g1(&c->d128, h, c->threads > 1);
    brp_g1(h, 128);
    g1_batch_normalize(pa, h, 128);
    for (int i = 0; i < 128; i++) g1_compress(proofs + 48 * (size_t)i, &pa[i]);
    compute_cells_from_poly(c, poly, cells);
}

int oracle_blob_to_kzg_commitment(const oracle_ctx *c, const uint8_t *blob, uint8_t *out) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t));
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK) {
        data_to_poly(c, v);
        g1_t r; g1a_t a;
        g1_msm(&r, c->g1s, v, N_BLOB);   /* CommitKey::commit_g1 -> g1_lincomb (commit_key.rs:38-44) */
        g1_to_affine(&a, &r); g1_compress(out, &a);
    }
    free(v);
    return rc;
}
int oracle_compute_cells_and_kzg_proofs(const oracle_ctx *c, const uint8_t *blob, uint8_t *cells, uint8_t *proofs) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t));
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK) { data_to_poly(c, v); proofs_and_cells_from_poly(c, v, cells, proofs); }
    free(v);
    return rc;
}
int oracle_compute_cells(const oracle_ctx *c, const uint8_t *blob, uint8_t *cells) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t));
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK) { data_to_poly(c, v); compute_cells_from_poly(c, v, cells); }
    free(v);
    return rc;
}

/* ------------------------------------------------------------------ */
/* verify_cell_kzg_proof_batch (eip7594/src/verifier.rs:72-164, fk20/verifier.rs:129-384) */
int oracle_verify_cell_kzg_proof_batch(const oracle_ctx *c, size_t n_commitments, const uint8_t *commitments,
                                       size_t n_indices, const uint64_t *cell_indices, size_t n_cells,
                                       const uint8_t *cells, size_t n_proofs, const uint8_t *proofs, int *verified) {
    *verified = 0;
    /* deduplicate_with_indices (verifier.rs:49-65): by byte equality, first-occurrence order */
    uint8_t *uniq = malloc(n_commitments * 48 + 1);
    uint64_t *row = malloc((n_commitments + 1) * sizeof(uint64_t));
    size_t m = 0;
    for (size_t i = 0; i < n_commitments; i++) {
        size_t k = 0;
        for (; k < m; k++) if (!memcmp(uniq + 48 * k, commitments + 48 * i, 48)) break;
        if (k == m) { memcpy(uniq + 48 * m, commitments + 48 * i, 48); m++; }
        row[i] = k;
    }
    int rc = ORACLE_OK;
    /* validation (verifier.rs:123-164) */
    if (!(n_commitments == n_indices && n_commitments == n_cells && n_commitments == n_proofs)) rc = ORACLE_ERR_INPUT;
    if (rc == ORACLE_OK) for (size_t i = 0; i < n_indices; i++) if (cell_indices[i] >= N_CELLS) { rc = ORACLE_ERR_INPUT; break; }
    size_t n = n_cells;
    if (rc != ORACLE_OK || n == 0) { if (rc == ORACLE_OK) *verified = 1; free(uniq); free(row); return rc; }

    g1a_t *comm = malloc(m * sizeof(g1a_t)), *prf = malloc(n * sizeof(g1a_t));
    fr_t *evals = malloc(n * CELL_LEN * sizeof(fr_t));
    fr_t *rp = malloc(n * sizeof(fr_t)), *wrp = malloc(n * sizeof(fr_t)), *weights = calloc(m, sizeof(fr_t));
    uint8_t *hin = NULL;
    for (size_t i = 0; i < m && rc == ORACLE_OK; i++) if (g1_decompress(&comm[i], uniq + 48 * i, 1)) rc = ORACLE_ERR_G1;
    if (rc == ORACLE_OK) {
        int bad = 0;
#pragma omp parallel for if (c->threads > 1) schedule(dynamic, 8) reduction(| : bad)
        for (size_t i = 0; i < n; i++) bad |= g1_decompress(&prf[i], proofs + 48 * i, 1) != 0;
        if (bad) rc = ORACLE_ERR_G1;
    }
    if (rc == ORACLE_OK) rc = deserialize_scalars(evals, cells, n * CELL_LEN);
    if (rc != ORACLE_OK) goto done;

    /* compute_fiat_shamir_challenge (verifier.rs:269-328) */
    {
        size_t hlen = 16 + 8 * 4 + m * 48 + n * (8 + 8 + CELL_LEN * 32 + 48), off = 0;
        hin = malloc(hlen);
        memcpy(hin, "RCKZGCBATCH__V1_", 16); off = 16;
        uint64_t hdr[4] = {N_BLOB, CELL_LEN, m, n};
        for (int k = 0; k < 4; k++) for (int b = 0; b < 8; b++) hin[off++] = (uint8_t)(hdr[k] >> (56 - 8 * b));
        for (size_t i = 0; i < m; i++) { g1_compress(hin + off, &comm[i]); off += 48; }
        for (size_t k = 0; k < n; k++) {
            for (int b = 0; b < 8; b++) hin[off++] = (uint8_t)(row[k] >> (56 - 8 * b));
            for (int b = 0; b < 8; b++) hin[off++] = (uint8_t)(cell_indices[k] >> (56 - 8 * b));
            for (int e = 0; e < CELL_LEN; e++) { fr_to_be(hin + off, &evals[k * CELL_LEN + e]); off += 32; }
            g1_compress(hin + off, &prf[k]); off += 48;
        }
        uint8_t dig[32]; sha256(dig, hin, hlen);
        fr_t r; fr_from_be_reduce(&r, dig);
        fr_t cur = FR_ONE;                               /* compute_powers (verifier.rs:333-343) */
        for (size_t k = 0; k < n; k++) { rp[k] = cur; fr_mul(&cur, &cur, &r); }
    }
    g1_t sum_proofs, wsum_proofs, sum_comm, comm_interp;
    g1_msm(&sum_proofs, prf, rp, n);                                            /* step 2 */
    for (size_t k = 0; k < n; k++) fr_mul(&wrp[k], &rp[k], &c->coset_gens_pow_n_br[cell_indices[k]]);
    g1_msm(&wsum_proofs, prf, wrp, n);                                          /* step 3 */
    for (size_t k = 0; k < n; k++) fr_add(&weights[row[k]], &weights[row[k]], &rp[k]);
    g1_msm(&sum_comm, comm, weights, m);                                        /* step 4 */
    {   /* compute_sum_interpolation_poly (verifier.rs:348-384) */
        fr_t acc[CELL_LEN];
        for (int i = 0; i < CELL_LEN; i++) acc[i] = FR_ZERO;
        for (size_t k = 0; k < n; k++) {
            fr_t e[CELL_LEN];
            memcpy(e, evals + k * CELL_LEN, sizeof e);
            brp_fr(e, CELL_LEN);
            coset_ifft_scalars(&c->d64, e, &c->coset_gens_inv_br[cell_indices[k]], 0);
            for (int i = 0; i < CELL_LEN; i++) { fr_t t; fr_mul(&t, &e[i], &rp[k]); fr_add(&acc[i], &acc[i], &t); }
        }
        g1_msm(&comm_interp, c->g1s, acc, CELL_LEN);   /* VerificationKey::commit_g1: first 65 SRS points */
    }
    {   /* step 6: pairing check (verifier.rs:242-259) */
        g1_t pin; g1_sub(&pin, &sum_comm, &comm_interp); g1_add(&pin, &pin, &wsum_proofs);
        g1_t both[2] = {sum_proofs, pin}; g1a_t aff[2];
        g1_batch_normalize(aff, both, 2);
        g2a_t q[2] = {c->tau_pow_n, c->neg_g2_gen};
        *verified = pairing_product_is_one(aff, q, 2);
    }
done:
    free(hin); free(comm); free(prf); free(evals); free(rp); free(wrp); free(weights); free(uniq); free(row);
    return rc;
}

/* ------------------------------------------------------------------ */
/* recover_cells_and_kzg_proofs (eip7594/src/prover.rs:156-171, recovery.rs:22-151) */
int oracle_recover_cells_and_kzg_proofs(const oracle_ctx *c, size_t n_cells, const uint8_t *cells,
                                        size_t n_indices, const uint64_t *cell_indices,
                                        uint8_t *out_cells, uint8_t *out_proofs) {
    /* validate_recovery_inputs (recovery.rs:90-146) */
    if (n_indices != n_cells) return ORACLE_ERR_INPUT;
    for (size_t i = 0; i < n_indices; i++) if (cell_indices[i] >= N_CELLS) return ORACLE_ERR_INPUT;
    for (size_t i = 1; i < n_indices; i++) if (!(cell_indices[i - 1] < cell_indices[i])) return ORACLE_ERR_INPUT;
    if (n_indices < N_CELLS / 2 || n_indices > N_CELLS) return ORACLE_ERR_INPUT;

    int par = c->threads > 1, rc = ORACLE_OK;
    fr_t *e = calloc(N_EXT, sizeof(fr_t)), *z = calloc(N_EXT, sizeof(fr_t)), *zc = malloc(N_EXT * sizeof(fr_t));
    /* recover_evaluations_in_domain_order (cosets.rs:141-198) */
    for (size_t k = 0; k < n_cells && rc == ORACLE_OK; k++)
        rc = deserialize_scalars(e + CELL_LEN * cell_indices[k], cells + BYTES_PER_CELL * k, CELL_LEN);
    if (rc != ORACLE_OK) goto done;
    brp_fr(e, N_EXT);
    {
        int present[N_CELLS] = {0};
        for (size_t k = 0; k < n_cells; k++) present[reverse_bits(cell_indices[k], 7)] = 1;
        /* construct_vanishing_poly_from_block_erasures (reed_solomon.rs:220-262), vanishing_poly (poly_coeff.rs:109-115) */
        fr_t zp[N_CELLS + 1]; size_t deg = 0; zp[0] = FR_ONE;
        for (size_t i = 0; i < N_CELLS; i++) {
            if (present[i]) continue;
            fr_t root = c->d128.roots[i], nr; fr_neg(&nr, &root);
            /* multiply by (x - root): new[k] = old[k-1] - root*old[k], in place from the top */
            zp[deg + 1] = zp[deg];
            for (size_t k = deg; k >= 1; k--) {
                fr_t t; fr_mul(&t, &zp[k], &nr);
                fr_add(&zp[k], &t, &zp[k - 1]);
            }
            fr_mul(&zp[0], &zp[0], &nr);
            deg++;
        }
        for (size_t i = 0; i <= deg; i++) z[i * (N_EXT / N_CELLS)] = zp[i];
    }
    memcpy(zc, z, N_EXT * sizeof(fr_t));
    /* recover_polynomial_coefficient_erasure_pattern (reed_solomon.rs:332-384) */
    fft_scalars(&c->d8192, z, par);                                   /* z_eval */
    for (int i = 0; i < N_EXT; i++) fr_mul(&e[i], &e[i], &z[i]);      /* ez_eval */
    ifft_scalars(&c->d8192, e, par);                                  /* dz_coeffs */
    coset_fft_scalars(&c->d8192, e, &c->rs_coset_gen, par);           /* dz_coset_eval */
    coset_fft_scalars(&c->d8192, zc, &c->rs_coset_gen, par);          /* z coset eval */
    fr_batch_inverse(zc, N_EXT);
    for (int i = 0; i < N_EXT; i++) fr_mul(&e[i], &e[i], &zc[i]);
    coset_ifft_scalars(&c->d8192, e, &c->rs_coset_gen_inv, par);      /* d_coeffs */
    for (int i = N_BLOB; i < N_EXT; i++) if (!fr_is_zero(&e[i])) { rc = ORACLE_ERR_RECOVERY; break; }
    if (rc == ORACLE_OK) proofs_and_cells_from_poly(c, e, out_cells, out_proofs);
done:
    free(e); free(z); free(zc);
    return rc;
}

/* ------------------------------------------------------------------ */
/* EIP-4844 single-point operations (crates/eip4844/src/{prover,verifier}.rs,
   crates/cryptography/kzg_single_open/src/{prover,verifier}.rs) */

/* divide_by_linear (kzg_single_open/src/prover.rs:50-65): quotient of poly by (X - z) and remainder y = poly(z) */
static void divide_by_linear(const fr_t *poly, size_t n, const fr_t *z, fr_t *quotient /*n-1*/, fr_t *y) {
    fr_t k = FR_ZERO;
    for (size_t i = n; i-- > 0;) {
        fr_t t; fr_add(&t, &poly[i], &k);
        if (i > 0) quotient[i - 1] = t; else *y = t;
        fr_mul(&k, z, &t);
    }
}
/* compute_fiat_shamir_challenge (eip4844/src/verifier.rs:155-196) */
static void fs_challenge_4844(fr_t *z, const uint8_t *blob, const uint8_t *commitment) {
    size_t len = 16 + 16 + 131072 + 48;
    uint8_t *in = malloc(len);
    memcpy(in, "FSBLOBVERIFY_V1_", 16);
    memset(in + 16, 0, 16); in[16 + 14] = 0x10; /* u128 big-endian 4096 */
    memcpy(in + 32, blob, 131072);
    memcpy(in + 32 + 131072, commitment, 48);
    uint8_t dig[32]; sha256(dig, in, len);
    fr_from_be_reduce(z, dig);
    free(in);
}
/* Prover::compute_kzg_proof (kzg_single_open/src/prover.rs:33-46) on a polynomial in monomial form */
static void kzg_proof_from_poly(const oracle_ctx *c, const fr_t *poly, const fr_t *z, uint8_t *out_proof, fr_t *y) {
    fr_t *q = malloc(N_BLOB * sizeof(fr_t));
    divide_by_linear(poly, N_BLOB, z, q, y);
    g1_t r; g1a_t a;
    g1_msm(&r, c->g1s, q, N_BLOB - 1);
    g1_to_affine(&a, &r); g1_compress(out_proof, &a);
    free(q);
}
int oracle_compute_kzg_proof(const oracle_ctx *c, const uint8_t *blob, const uint8_t *z_bytes, uint8_t *out_proof, uint8_t *out_y) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t)), z, y;
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK && fr_from_be(&z, z_bytes)) rc = ORACLE_ERR_SCALAR;
    if (rc == ORACLE_OK) { data_to_poly(c, v); kzg_proof_from_poly(c, v, &z, out_proof, &y); fr_to_be(out_y, &y); }
    free(v);
    return rc;
}
int oracle_compute_blob_kzg_proof(const oracle_ctx *c, const uint8_t *blob, const uint8_t *commitment, uint8_t *out_proof) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t)), z, y;
    g1a_t cm;
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK && g1_decompress(&cm, commitment, 1)) rc = ORACLE_ERR_G1;
    if (rc == ORACLE_OK) { data_to_poly(c, v); fs_challenge_4844(&z, blob, commitment); kzg_proof_from_poly(c, v, &z, out_proof, &y); }
    free(v);
    return rc;
}
/* Verifier::verify_kzg_proof (kzg_single_open/src/verifier.rs:33-57).  The reference pairs (C - yG, -G2) with
   (pi, [tau - z]_2); by bilinearity that product equals e(C - yG + z pi, -G2) * e(pi, [tau]_2), which needs no G2
   arithmetic (the same rearrangement the reference's own batch verifier uses, verifier.rs:76-107). */
static int verify_single(const oracle_ctx *c, const g1a_t *cm, const fr_t *z, const fr_t *y, const g1a_t *pi) {
    g1a_t pts[3] = {*cm, c->g1s[0], *pi};
    fr_t sc[3] = {FR_ONE, FR_ZERO, *z};
    fr_neg(&sc[1], y);
    g1_t lhs; g1_msm(&lhs, pts, sc, 3);
    g1_t both[2]; both[0] = lhs; g1_from_affine(&both[1], pi);
    g1a_t aff[2]; g1_batch_normalize(aff, both, 2);
    g2a_t q[2] = {c->neg_g2_gen, c->tau_g2};
    return pairing_product_is_one(aff, q, 2);
}
int oracle_verify_kzg_proof(const oracle_ctx *c, const uint8_t *commitment, const uint8_t *z_bytes, const uint8_t *y_bytes,
                            const uint8_t *proof, int *verified) {
    g1a_t cm, pi; fr_t z, y;
    *verified = 0;
    if (g1_decompress(&cm, commitment, 1)) return ORACLE_ERR_G1;
    if (g1_decompress(&pi, proof, 1)) return ORACLE_ERR_G1;
    if (fr_from_be(&z, z_bytes) || fr_from_be(&y, y_bytes)) return ORACLE_ERR_SCALAR;
    *verified = verify_single(c, &cm, &z, &y, &pi);
    return ORACLE_OK;
}
/* PolyCoeff::eval (polynomial/src/poly_coeff.rs:59-65) */
static void poly_eval(fr_t *out, const fr_t *poly, size_t n, const fr_t *x) {
    fr_t r = FR_ZERO;
    for (size_t i = n; i-- > 0;) { fr_mul(&r, &r, x); fr_add(&r, &r, &poly[i]); }
    *out = r;
}
int oracle_verify_blob_kzg_proof(const oracle_ctx *c, const uint8_t *blob, const uint8_t *commitment, const uint8_t *proof, int *verified) {
    fr_t *v = malloc(N_BLOB * sizeof(fr_t)), z, y;
    g1a_t cm, pi;
    *verified = 0;
    int rc = deserialize_scalars(v, blob, N_BLOB);
    if (rc == ORACLE_OK && g1_decompress(&cm, commitment, 1)) rc = ORACLE_ERR_G1;
    if (rc == ORACLE_OK && g1_decompress(&pi, proof, 1)) rc = ORACLE_ERR_G1;
    if (rc == ORACLE_OK) {
        fs_challenge_4844(&z, blob, commitment);
        data_to_poly(c, v);
        poly_eval(&y, v, N_BLOB, &z);
        *verified = verify_single(c, &cm, &z, &y, &pi);
    }
    free(v);
    return rc;
}
/* verify_blob_kzg_proof_batch (eip4844/src/verifier.rs:81-143, :201-262; kzg_single_open/src/verifier.rs:59-108) */
int oracle_verify_blob_kzg_proof_batch(const oracle_ctx *c, size_t n_blobs, const uint8_t *blobs, size_t n_commitments,
                                       const uint8_t *commitments, size_t n_proofs, const uint8_t *proofs, int *verified) {
    *verified = 0;
    if (!(n_blobs == n_commitments && n_blobs == n_proofs)) return ORACLE_ERR_INPUT;
    size_t n = n_blobs;
    int rc = ORACLE_OK;
    fr_t *polys = malloc((n * N_BLOB + 1) * sizeof(fr_t));
    g1a_t *cm = malloc((n + 1) * sizeof(g1a_t)), *pi = malloc((n + 1) * sizeof(g1a_t));
    fr_t *zs = malloc((n + 1) * sizeof(fr_t)), *ys = malloc((n + 1) * sizeof(fr_t));
    for (size_t i = 0; i < n && rc == ORACLE_OK; i++) rc = deserialize_scalars(polys + i * N_BLOB, blobs + i * 131072, N_BLOB);
    for (size_t i = 0; i < n && rc == ORACLE_OK; i++) if (g1_decompress(&cm[i], commitments + 48 * i, 1)) rc = ORACLE_ERR_G1;
    for (size_t i = 0; i < n && rc == ORACLE_OK; i++) if (g1_decompress(&pi[i], proofs + 48 * i, 1)) rc = ORACLE_ERR_G1;
    if (rc == ORACLE_OK) {
        for (size_t i = 0; i < n; i++) {
            fs_challenge_4844(&zs[i], blobs + i * 131072, commitments + 48 * i);
            data_to_poly(c, polys + i * N_BLOB);
            poly_eval(&ys[i], polys + i * N_BLOB, N_BLOB, &zs[i]);
        }
        /* compute_r_powers_for_verify_kzg_proof_batch */
        size_t hlen = 16 + 8 + 8 + n * (48 + 32 + 32 + 48), off = 0;
        uint8_t *hin = malloc(hlen);
        memcpy(hin, "RCKZGBATCH___V1_", 16); off = 16;
        uint64_t hdr[2] = {N_BLOB, n};
        for (int k = 0; k < 2; k++) for (int b = 0; b < 8; b++) hin[off++] = (uint8_t)(hdr[k] >> (56 - 8 * b));
        for (size_t i = 0; i < n; i++) {
            memcpy(hin + off, commitments + 48 * i, 48); off += 48;
            fr_to_be(hin + off, &zs[i]); off += 32;
            fr_to_be(hin + off, &ys[i]); off += 32;
            memcpy(hin + off, proofs + 48 * i, 48); off += 48;
        }
        uint8_t dig[32]; sha256(dig, hin, hlen); free(hin);
        fr_t r; fr_from_be_reduce(&r, dig);
        /* lhs = sum r^i C_i - (sum r^i y_i) G + sum r^i z_i pi_i ; rhs = sum r^i pi_i */
        g1a_t *pts = malloc((2 * n + 1) * sizeof(g1a_t)); fr_t *sc = malloc((2 * n + 1) * sizeof(fr_t));
        fr_t cur = FR_ONE, ysum = FR_ZERO;
        for (size_t i = 0; i < n; i++) {
            pts[i] = cm[i]; sc[i] = cur;
            pts[n + 1 + i] = pi[i]; fr_mul(&sc[n + 1 + i], &cur, &zs[i]);
            fr_t t; fr_mul(&t, &cur, &ys[i]); fr_add(&ysum, &ysum, &t);
            fr_mul(&cur, &cur, &r);
        }
        pts[n] = c->g1s[0]; fr_neg(&sc[n], &ysum);
        g1_t both[2];
        g1_msm(&both[0], pts, sc, 2 * n + 1);
        g1_msm(&both[1], pi, sc, n);   /* sc[0..n) = r^i */
        g1a_t aff[2]; g1_batch_normalize(aff, both, 2);
        g2a_t q[2] = {c->neg_g2_gen, c->tau_g2};
        *verified = pairing_product_is_one(aff, q, 2);
        free(pts); free(sc);
    }
    free(polys); free(cm); free(pi); free(zs); free(ys);
    return rc;
}

/* ------------------------------------------------------------------ */
/* stage-level entry points for kernel parity tests */
int oracle_fr_ntt(uint8_t *data, size_t n, int inverse, int coset) {
    bls_init(); fr_neg(&FR_MINUS_ONE, &FR_ONE);
    domain_t d; domain_init(&d, n);
    fr_t *v = malloc(n * sizeof(fr_t));
    int rc = deserialize_scalars(v, data, n);
    if (rc == ORACLE_OK) {
        fr_t g, gi; fr_from_u64(&g, 7); fr_inv(&gi, &g);
        if (coset == 1) coset_fft_scalars(&d, v, &g, 0);
        else if (coset == 2) coset_ifft_scalars(&d, v, &gi, 0);
        else if (inverse) ifft_scalars(&d, v, 0);
        else fft_scalars(&d, v, 0);
        for (size_t i = 0; i < n; i++) fr_to_be(data + 32 * i, &v[i]);
    }
    free(v); domain_free(&d);
    return rc;
}
int oracle_g1_fft(uint8_t *points, size_t n, int inverse) {
    bls_init(); fr_neg(&FR_MINUS_ONE, &FR_ONE);
    domain_t d; domain_init(&d, n);
    g1_t *v = malloc(n * sizeof(g1_t)); g1a_t *a = malloc(n * sizeof(g1a_t));
    int rc = 0;
    for (size_t i = 0; i < n && !rc; i++) { rc = g1_decompress(&a[i], points + 48 * i, 0); g1_from_affine(&v[i], &a[i]); }
    if (!rc) {
        if (inverse) ifft_g1_take_n(&d, v, n, 1); else fft_g1(&d, v, 1);
        g1_batch_normalize(a, v, n);
        for (size_t i = 0; i < n; i++) g1_compress(points + 48 * i, &a[i]);
    }
    free(v); free(a); domain_free(&d);
    return rc;
}
int oracle_g1_msm(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t *out) {
    bls_init();
    g1a_t *a = malloc((n + 1) * sizeof(g1a_t)); fr_t *k = malloc((n + 1) * sizeof(fr_t));
    int rc = 0;
    for (size_t i = 0; i < n && !rc; i++) {
        rc = g1_decompress(&a[i], points + 48 * i, 0);
        if (!rc && fr_from_be(&k[i], scalars + 32 * i)) rc = -3;
    }
    if (!rc) { g1_t r; g1a_t ra; g1_msm(&r, a, k, n); g1_to_affine(&ra, &r); g1_compress(out, &ra); }
    free(a); free(k);
    return rc;
}
int oracle_g1_mul(const uint8_t *point, const uint8_t *scalar, uint8_t *out) {
    bls_init();
    g1a_t a; fr_t k; g1_t p, r;
    if (g1_decompress(&a, point, 0)) return -1;
    if (fr_from_be(&k, scalar)) return -3;
    g1_from_affine(&p, &a); g1_mul(&r, &p, &k); g1_to_affine(&a, &r); g1_compress(out, &a);
    return 0;
}
int oracle_g1_validate(const uint8_t *point, int subgroup_check) {
    bls_init(); g1a_t a; return g1_decompress(&a, point, subgroup_check);
}
int oracle_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    bls_init(); fr_t x, y;
    if (fr_from_be(&x, a) || fr_from_be(&y, b)) return -1;
    fr_mul(&x, &x, &y); fr_to_be(out, &x); return 0;
}
int oracle_fp_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    bls_init(); fp_t x, y;
    if (fp_from_be(&x, a) || fp_from_be(&y, b)) return -1;
    fp_mul(&x, &x, &y); fp_to_be(out, &x); return 0;
}
void oracle_sha256(const uint8_t *data, size_t len, uint8_t *out) { sha256(out, data, len); }
/* get_booth_index (booth_encoding.rs:4-46) on a 32-byte little-endian scalar */
int oracle_booth_index(unsigned window_index, unsigned window_size, const uint8_t *el_le32) { return get_booth_index(window_index, window_size, el_le32); }
/* reduce_bytes_to_scalar_bias (bls12_381/src/lib.rs:128-140): 32-byte BE in, canonical 32-byte BE out */
void oracle_reduce_bytes_to_scalar(const uint8_t *in_be32, uint8_t *out_be32) { bls_init(); fr_t x; fr_from_be_reduce(&x, in_be32); fr_to_be(out_be32, &x); }
