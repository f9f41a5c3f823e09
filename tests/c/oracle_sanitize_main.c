/* Sanitizer driver for the CPU oracle (-fsanitize=address,undefined on the CPU-compilable code; never
 * on the GPU).  Runs every oracle entry point once on a synthetic blob: commitment, cells + proofs (with and without
 * window tables), verification of a few cells, recovery from half of the cells, the EIP-4844 operations and the
 * stage-level helpers.  Built and run by tests/test_sanitizers.py; exit code 0 and an empty sanitizer log = pass. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kzg_oracle.h"

#define BLOB 131072
#define CELLS 128
#define CELL 2048

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *srs = malloc((size_t)len);
    if (fread(srs, 1, (size_t)len, f) != (size_t)len) return 1;
    fclose(f);
    uint8_t *blob = malloc(BLOB), *cells = malloc(CELLS * CELL), *proofs = malloc(CELLS * 48), *cells2 = malloc(CELLS * CELL),
            *proofs2 = malloc(CELLS * 48);
    for (int i = 0; i < 4096; i++) {  /* element i = a small deterministic value, canonical */
        memset(blob + 32 * i, 0, 32);
        uint32_t v = 2654435761u * (uint32_t)(i + 1);
        blob[32 * i + 1] = (uint8_t)(v >> 24); blob[32 * i + 17] = (uint8_t)(v >> 16); blob[32 * i + 30] = (uint8_t)(v >> 8); blob[32 * i + 31] = (uint8_t)v;
    }
    int bad = 0;
    for (int precomp = 0; precomp < 2; precomp++) {
        oracle_ctx *ctx = oracle_ctx_new(srs, (size_t)len, precomp, 2);
        if (!ctx) return 2;
        uint8_t comm[48];
        bad |= oracle_blob_to_kzg_commitment(ctx, blob, comm);
        bad |= oracle_compute_cells_and_kzg_proofs(ctx, blob, precomp ? cells2 : cells, precomp ? proofs2 : proofs);
        if (precomp) {
            bad |= memcmp(cells, cells2, CELLS * CELL) != 0 || memcmp(proofs, proofs2, CELLS * 48) != 0;
            /* verify 5 cells, one of them twice */
            uint64_t idx[6] = {0, 1, 64, 127, 5, 5};
            uint8_t *vc = malloc(6 * CELL), *vp = malloc(6 * 48), *vcm = malloc(6 * 48);
            for (int k = 0; k < 6; k++) { memcpy(vc + k * CELL, cells + idx[k] * CELL, CELL); memcpy(vp + k * 48, proofs + idx[k] * 48, 48); memcpy(vcm + k * 48, comm, 48); }
            int ok = 0;
            bad |= oracle_verify_cell_kzg_proof_batch(ctx, 6, vcm, 6, idx, 6, vc, 6, vp, &ok);
            bad |= !ok;
            vp[0] ^= 1;  /* a corrupted proof: either "false" or a decoding error, never a crash */
            (void)oracle_verify_cell_kzg_proof_batch(ctx, 6, vcm, 6, idx, 6, vc, 6, vp, &ok);
            free(vc); free(vp); free(vcm);
            /* recover from the odd cells */
            uint64_t hidx[64];
            uint8_t *half = malloc(64 * CELL);
            for (int k = 0; k < 64; k++) { hidx[k] = (uint64_t)(2 * k + 1); memcpy(half + k * CELL, cells + (2 * k + 1) * CELL, CELL); }
            bad |= oracle_recover_cells_and_kzg_proofs(ctx, 64, half, 64, hidx, cells2, proofs2);
            bad |= memcmp(cells, cells2, CELLS * CELL) != 0 || memcmp(proofs, proofs2, CELLS * 48) != 0;
            free(half);
            /* EIP-4844 */
            uint8_t z[32] = {0}, y[32], pr[48], bpr[48];
            z[31] = 7;
            bad |= oracle_compute_kzg_proof(ctx, blob, z, pr, y);
            bad |= oracle_verify_kzg_proof(ctx, comm, z, y, pr, &ok);
            bad |= !ok;
            bad |= oracle_compute_blob_kzg_proof(ctx, blob, comm, bpr);
            bad |= oracle_verify_blob_kzg_proof(ctx, blob, comm, bpr, &ok);
            bad |= !ok;
            bad |= oracle_verify_blob_kzg_proof_batch(ctx, 1, blob, 1, comm, 1, bpr, &ok);
            bad |= !ok;
        }
        oracle_ctx_free(ctx);
    }
    /* stage-level helpers */
    uint8_t d[32 * 64];
    for (int i = 0; i < 64; i++) { memset(d + 32 * i, 0, 32); d[32 * i + 31] = (uint8_t)(i + 1); }
    bad |= oracle_fr_ntt(d, 64, 0, 0);
    bad |= oracle_fr_ntt(d, 64, 1, 0);
    bad |= d[31] != 1 || d[32 * 63 + 31] != 64;
    uint8_t dig[32];
    oracle_sha256((const uint8_t *)"abc", 3, dig);
    bad |= dig[0] != 0xba;
    free(srs); free(blob); free(cells); free(proofs); free(cells2); free(proofs2);
    printf("oracle sanitize run: %s\n", bad ? "FAILED" : "ok");
    return bad ? 3 : 0;
}
