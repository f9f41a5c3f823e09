/* Seconds from process start to the first compute_cells_and_kzg_proofs result for a plain C consumer of libc_eth_kzg.so
 * (no interpreter in front): the figure next to the reference's "Initialize context" bench (crates/eip7594/benches/
 * benchmark-mt.rs:103-113).  Prints one JSON line.  Build: make -C tools/first_result ; run on a GPU box. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "c_eth_kzg.h"

enum { FIELD_ELEMENTS_PER_BLOB = 4096, BYTES_PER_BLOB = 32 * 4096, CELLS_PER_EXT_BLOB = 128, BYTES_PER_CELL = 2048 };

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(void) {
    const double t0 = now();
    static uint8_t blob[BYTES_PER_BLOB];
    for (int i = 0; i < FIELD_ELEMENTS_PER_BLOB; i++) {  /* canonical field elements: top byte 0 */
        unsigned x = 2654435761u * (unsigned)(i + 1);
        for (int k = 1; k < 32; k++) { blob[32 * i + k] = (uint8_t)(x >> 24); x = x * 1664525u + 1013904223u; }
    }
    static uint8_t cells[CELLS_PER_EXT_BLOB][BYTES_PER_CELL], proofs[CELLS_PER_EXT_BLOB][48];
    uint8_t *pc[CELLS_PER_EXT_BLOB], *pp[CELLS_PER_EXT_BLOB];
    for (int k = 0; k < CELLS_PER_EXT_BLOB; k++) { pc[k] = cells[k]; pp[k] = proofs[k]; }
    DASContext* ctx = eth_kzg_das_context_new(true);
    const double t1 = now();
    CResult r = eth_kzg_compute_cells_and_kzg_proofs(ctx, blob, pc, pp);
    const double t2 = now();
    if (r.status != Ok) { fprintf(stderr, "first_result: %s\n", r.error_msg ? r.error_msg : "error"); return 1; }
    unsigned sum = 0;
    for (int k = 0; k < CELLS_PER_EXT_BLOB; k++) for (int j = 0; j < 48; j++) sum = sum * 31 + proofs[k][j];
    r = eth_kzg_compute_cells_and_kzg_proofs(ctx, blob, pc, pp);
    const double t3 = now();
    printf("{\"constructor_returned_s\": %.3f, \"first_result_s\": %.3f, \"second_result_ms\": %.2f, \"proofs_checksum\": %u}\n", t1 - t0,
           t2 - t0, 1e3 * (t3 - t2), sum);
    fflush(stdout);
    eth_kzg_das_context_free(ctx);
    return 0;
}
