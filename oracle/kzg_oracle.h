/*
 * oracle/kzg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle for the hot path of crate-crypto/rust-eth-kzg: a plain-C restatement of
 * the reference algorithm (citations in kzg.c).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so; the product
 * (rust-eth-kzg_amd/, libc_eth_kzg.so) never links or calls it.
 *
 * Parity pinning: every function below is checked against the reference's own golden
 * vectors (tests/golden, converted from /root/reference/test_vectors) by
 * tests/test_oracle_vectors.py.
 */
#ifndef KZG_ORACLE_H
#define KZG_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

enum {
    ORACLE_OK = 0,
    ORACLE_ERR_SCALAR = 1,      /* a 32-byte field element >= r */
    ORACLE_ERR_G1 = 2,          /* bad G1 encoding / not on curve / not in subgroup */
    ORACLE_ERR_INPUT = 3,       /* length / index validation failed */
    ORACLE_ERR_RECOVERY = 4,    /* recovered polynomial has the wrong degree, etc. */
};

/* srs: the flat binary written by tests/golden/make_fixtures.py
   ("KZGSRS01" | n_g1 | n_g2 | 48-B G1 monomial points | 96-B G2 monomial points).
   use_precomp != 0 => width-8 fixed-base window tables (UsePrecomp::Yes{width:8});
   threads: number of OpenMP threads used for the axes maybe_rayon parallelises (>=1). */
oracle_ctx *oracle_ctx_new(const uint8_t *srs, size_t srs_len, int use_precomp, int threads);
void oracle_ctx_free(oracle_ctx *ctx);

int oracle_blob_to_kzg_commitment(const oracle_ctx *ctx, const uint8_t *blob /*131072*/, uint8_t *out /*48*/);
int oracle_compute_cells_and_kzg_proofs(const oracle_ctx *ctx, const uint8_t *blob,
                                        uint8_t *cells /*128*2048*/, uint8_t *proofs /*128*48*/);
int oracle_compute_cells(const oracle_ctx *ctx, const uint8_t *blob, uint8_t *cells);
/* arrays are flat: commitments n*48, cells n*2048, proofs n*48 */
int oracle_verify_cell_kzg_proof_batch(const oracle_ctx *ctx, size_t n_commitments, const uint8_t *commitments,
                                       size_t n_indices, const uint64_t *cell_indices,
                                       size_t n_cells, const uint8_t *cells,
                                       size_t n_proofs, const uint8_t *proofs, int *verified);
int oracle_recover_cells_and_kzg_proofs(const oracle_ctx *ctx, size_t n_cells, const uint8_t *cells,
                                        size_t n_indices, const uint64_t *cell_indices,
                                        uint8_t *out_cells, uint8_t *out_proofs);

/* EIP-4844 single-point operations (crates/eip4844); flat arrays as above */
int oracle_compute_kzg_proof(const oracle_ctx *ctx, const uint8_t *blob, const uint8_t *z /*32*/, uint8_t *out_proof /*48*/, uint8_t *out_y /*32*/);
int oracle_compute_blob_kzg_proof(const oracle_ctx *ctx, const uint8_t *blob, const uint8_t *commitment /*48*/, uint8_t *out_proof /*48*/);
int oracle_verify_kzg_proof(const oracle_ctx *ctx, const uint8_t *commitment, const uint8_t *z, const uint8_t *y, const uint8_t *proof, int *verified);
int oracle_verify_blob_kzg_proof(const oracle_ctx *ctx, const uint8_t *blob, const uint8_t *commitment, const uint8_t *proof, int *verified);
int oracle_verify_blob_kzg_proof_batch(const oracle_ctx *ctx, size_t n_blobs, const uint8_t *blobs, size_t n_commitments,
                                       const uint8_t *commitments, size_t n_proofs, const uint8_t *proofs, int *verified);

/* ---- stage-level entry points used by the kernel parity tests (canonical encodings) ---- */
/* Fr NTT over n = 2^k elements given as 32-byte big-endian canonical scalars.
   inverse != 0: multiply by n^-1 afterwards (Domain::ifft_scalars). coset: 0 none,
   1 = coset_fft_scalars with generator 7 (pre-scale), 2 = coset_ifft_scalars (post-scale by 7^-i). */
int oracle_fr_ntt(uint8_t *data, size_t n, int inverse, int coset);
/* G1 FFT over n = 2^k compressed points (48 B each); inverse => also * n^-1 */
int oracle_g1_fft(uint8_t *points, size_t n, int inverse);
/* sum_i k_i * P_i; points compressed (unchecked), scalars 32-B BE */
int oracle_g1_msm(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t *out);
/* k*P */
int oracle_g1_mul(const uint8_t *point, const uint8_t *scalar, uint8_t *out);
/* decompress + (optional) subgroup check + recompress; returns 0 / -1 / -2 like g1_decompress */
int oracle_g1_validate(const uint8_t *point, int subgroup_check);
/* a*b mod r, a+b mod r on 32-B BE canonical scalars */
int oracle_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out);
/* a*b mod p on 48-B BE canonical */
int oracle_fp_mul(const uint8_t *a, const uint8_t *b, uint8_t *out);
void oracle_sha256(const uint8_t *data, size_t len, uint8_t *out);
int oracle_booth_index(unsigned window_index, unsigned window_size, const uint8_t *el_le32);
void oracle_reduce_bytes_to_scalar(const uint8_t *in_be32, uint8_t *out_be32);

#ifdef __cplusplus
}
#endif
#endif
