/*
 * tests/c/abi_runner.c -- plain-C consumer of include/c_eth_kzg.h, the way a cgo / P-Invoke / Nim caller links
 * the reference's c_eth_kzg library (bindings/golang/prover.go:4-10).  Used by tests/test_c_abi_runner.py:
 *   abi_runner compute <blob file> <out file>      out = 128*2048 cell bytes | 128*48 proof bytes | 48 commitment bytes
 *   abi_runner verify  <blob file>                 computes, then verifies all 128 cells; prints "verified=1"
 *   abi_runner multi   <blob file>                 the same blob 5 times through eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi
 *                                                  over every visible GPU (one context each); prints "multi=1 devices=N"
 * Exit code 0 on success, 2 when the library reports Err (message on stderr).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c_eth_kzg.h"

#define BLOB 131072
#define CELLS 128
#define CELL 2048

static int fail(CResult r, const char *what) {
    fprintf(stderr, "%s: %s\n", what, r.error_msg ? r.error_msg : "(no message)");
    eth_kzg_free_error_message(r.error_msg);
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s compute|verify <blob> [out]\n", argv[0]); return 1; }
    static uint8_t blob[BLOB], cells[CELLS][CELL], proofs[CELLS][48], commitment[48];
    FILE *f = fopen(argv[2], "rb");
    if (!f || fread(blob, 1, BLOB, f) != BLOB) { fprintf(stderr, "cannot read blob\n"); return 1; }
    fclose(f);
    if (eth_kzg_constant_bytes_per_cell() != CELL || eth_kzg_constant_cells_per_ext_blob() != CELLS ||
        eth_kzg_constant_bytes_per_proof() != 48) return 1;

    DASContext *ctx = eth_kzg_das_context_new(true);
    uint8_t *cell_ptrs[CELLS], *proof_ptrs[CELLS];
    for (int i = 0; i < CELLS; i++) { cell_ptrs[i] = cells[i]; proof_ptrs[i] = proofs[i]; }
    CResult r = eth_kzg_compute_cells_and_kzg_proofs(ctx, blob, cell_ptrs, proof_ptrs);
    if (r.status != Ok) return fail(r, "compute_cells_and_kzg_proofs");
    r = eth_kzg_blob_to_kzg_commitment(ctx, blob, commitment);
    if (r.status != Ok) return fail(r, "blob_to_kzg_commitment");

    if (!strcmp(argv[1], "compute")) {
        if (argc < 4) return 1;
        FILE *o = fopen(argv[3], "wb");
        if (!o) return 1;
        fwrite(cells, 1, sizeof cells, o);
        fwrite(proofs, 1, sizeof proofs, o);
        fwrite(commitment, 1, sizeof commitment, o);
        fclose(o);
    } else if (!strcmp(argv[1], "multi")) {
        /* single-process fan-out: one context per visible GPU, a 5-blob batch (blob 3 invalid) cut into slices */
        enum { NB = 5 };
        int ndev = eth_kzg_amd_device_count();
        if (ndev < 1) return 1;
        if (ndev > 8) ndev = 8;
        const DASContext *ctxs[8];
        ctxs[0] = ctx;
        for (int d = 1; d < ndev; d++) ctxs[d] = eth_kzg_amd_das_context_new_on_device(true, d);
        static uint8_t bad[BLOB], mc[NB][CELLS][CELL], mp[NB][CELLS][48];
        memset(bad, 0xff, sizeof bad);
        const uint8_t *blobs[NB] = {blob, blob, blob, bad, blob};
        uint8_t *cp[NB][CELLS], *pp[NB][CELLS];
        uint8_t *const *cpp[NB];
        uint8_t *const *ppp[NB];
        int32_t st[NB];
        for (int b = 0; b < NB; b++) {
            for (int i = 0; i < CELLS; i++) { cp[b][i] = mc[b][i]; pp[b][i] = mp[b][i]; }
            cpp[b] = cp[b]; ppp[b] = pp[b];
        }
        r = eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi(ctxs, (uint64_t)ndev, NB, blobs, cpp, ppp, st);
        if (r.status != Ok) return fail(r, "compute_cells_and_kzg_proofs_batch_multi");
        int good = st[3] != 0;
        for (int b = 0; b < NB; b++)
            if (b != 3) good = good && st[b] == 0 && !memcmp(mc[b], cells, sizeof cells) && !memcmp(mp[b], proofs, sizeof proofs);
        printf("multi=%d devices=%d\n", good, ndev);
        for (int d = 1; d < ndev; d++) eth_kzg_das_context_free((DASContext *)ctxs[d]);
    } else {
        const uint8_t *comm_ptrs[CELLS], *ccell[CELLS], *cproof[CELLS];
        uint64_t idx[CELLS];
        for (int i = 0; i < CELLS; i++) { comm_ptrs[i] = commitment; ccell[i] = cells[i]; cproof[i] = proofs[i]; idx[i] = (uint64_t)i; }
        bool ok = false;
        r = eth_kzg_verify_cell_kzg_proof_batch(ctx, CELLS, comm_ptrs, CELLS, idx, CELLS, ccell, CELLS, cproof, &ok);
        if (r.status != Ok) return fail(r, "verify_cell_kzg_proof_batch");
        printf("verified=%d\n", ok ? 1 : 0);
        /* recover from the even cells and compare */
        static uint8_t rc[CELLS][CELL], rp[CELLS][48];
        uint8_t *rc_ptrs[CELLS], *rp_ptrs[CELLS];
        const uint8_t *half[CELLS / 2];
        uint64_t hidx[CELLS / 2];
        for (int i = 0; i < CELLS; i++) { rc_ptrs[i] = rc[i]; rp_ptrs[i] = rp[i]; }
        for (int i = 0; i < CELLS / 2; i++) { half[i] = cells[2 * i]; hidx[i] = (uint64_t)(2 * i); }
        r = eth_kzg_recover_cells_and_proofs(ctx, CELLS / 2, half, CELLS / 2, hidx, rc_ptrs, rp_ptrs);
        if (r.status != Ok) return fail(r, "recover_cells_and_proofs");
        printf("recovered=%d\n", memcmp(rc, cells, sizeof cells) == 0 && memcmp(rp, proofs, sizeof proofs) == 0);
    }
    eth_kzg_das_context_free(ctx);
    return 0;
}
