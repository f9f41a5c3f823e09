/*
 * N-API addon over include/c_eth_kzg.h: the reference's Node binding (bindings/node/src/lib.rs, index.d.ts -- napi-rs over the
 * Rust crate) re-expressed over the C ABI, so that `libc_eth_kzg.so` is a drop-in for Node callers too.  Same surface for
 * the EIP-7594 path: class DasContextJs { constructor(), static create({usePrecomp}), blobToKzgCommitment,
 * computeCellsAndKzgProofs, computeCells, recoverCellsAndKzgProofs, verifyCellKzgProofBatch, and the EIP-4844 operations
 * computeKzgProof, computeBlobKzgProof, verifyKzgProof, verifyBlobKzgProof, verifyBlobKzgProofBatch } and the async* forms, which
 * run on libuv worker threads against ONE shared context, like the reference's (`async fn` over Arc<DASContext>,
 * lib.rs:92-299).  Plain C, N-API version 6 (BigInt cell indices).  Build: bindings/node/Makefile.
 */
#include <node_api.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c_eth_kzg.h"

#define BLOB 131072
#define CELLS 128
#define CELL 2048
#define G1 48

#define NAPI_OK(call)                                                   \
    do {                                                                \
        if ((call) != napi_ok) {                                        \
            napi_throw_error(env, NULL, "N-API call failed: " #call);   \
            return NULL;                                                \
        }                                                               \
    } while (0)

enum { JOB_COMMIT, JOB_CELLS_PROOFS, JOB_CELLS, JOB_RECOVER, JOB_VERIFY,
       /* EIP-4844 single-point operations (index.d.ts: computeKzgProof ... verifyBlobKzgProofBatch) */
       JOB_KZG_PROOF, JOB_BLOB_PROOF, JOB_VERIFY_KZG, JOB_VERIFY_BLOB, JOB_VERIFY_BLOB_BATCH };

typedef struct {
    int kind;
    const DASContext *ctx;
    /* inputs, copied out of the JS buffers so that a worker thread may read them */
    uint8_t *blob;                 /* BLOB bytes */
    uint64_t n;                    /* recover / verify: number of cells */
    uint8_t *cells_in, *proofs_in, *commitments_in;
    uint64_t *indices;
    /* outputs */
    uint8_t *cells_out, *proofs_out; /* CELLS*CELL, CELLS*G1 */
    uint8_t commitment[G1];
    uint8_t *blobs_in;             /* verifyBlobKzgProofBatch: n blobs */
    uint8_t small[3][G1];          /* 4844 inputs: commitment / z / y / proof, as the operation needs them */
    uint8_t proof_out[G1], y_out[32];
    bool verified;
    CResult res;
    /* async plumbing */
    napi_async_work work;
    napi_deferred deferred;
} Job;

static void job_free(Job *j) {
    if (!j) return;
    free(j->blob); free(j->cells_in); free(j->proofs_in); free(j->commitments_in); free(j->indices); free(j->blobs_in);
    free(j->cells_out); free(j->proofs_out);
    if (j->res.error_msg) eth_kzg_free_error_message(j->res.error_msg);
    free(j);
}

static void job_run(Job *j) {
    uint8_t *cp[CELLS], *pp[CELLS];
    if (j->cells_out) for (int i = 0; i < CELLS; i++) cp[i] = j->cells_out + (size_t)i * CELL;
    if (j->proofs_out) for (int i = 0; i < CELLS; i++) pp[i] = j->proofs_out + (size_t)i * G1;
    switch (j->kind) {
        case JOB_COMMIT: j->res = eth_kzg_blob_to_kzg_commitment(j->ctx, j->blob, j->commitment); break;
        case JOB_CELLS_PROOFS: j->res = eth_kzg_compute_cells_and_kzg_proofs(j->ctx, j->blob, cp, pp); break;
        case JOB_CELLS: j->res = eth_kzg_compute_cells(j->ctx, j->blob, cp); break;
        case JOB_RECOVER: {
            const uint8_t **in = malloc((j->n ? j->n : 1) * sizeof *in);
            for (uint64_t k = 0; k < j->n; k++) in[k] = j->cells_in + k * CELL;
            j->res = eth_kzg_recover_cells_and_proofs(j->ctx, j->n, in, j->n, j->indices, cp, pp);
            free(in);
            break;
        }
        case JOB_KZG_PROOF: j->res = eth_kzg_compute_kzg_proof(j->ctx, j->blob, j->small[0], j->proof_out, j->y_out); break;
        case JOB_BLOB_PROOF: j->res = eth_kzg_compute_blob_kzg_proof(j->ctx, j->blob, j->small[0], j->proof_out); break;
        case JOB_VERIFY_KZG: j->res = eth_kzg_verify_kzg_proof(j->ctx, j->small[0], j->small[1], j->small[2], j->commitment, &j->verified); break;
        case JOB_VERIFY_BLOB: j->res = eth_kzg_verify_blob_kzg_proof(j->ctx, j->blob, j->small[0], j->small[1], &j->verified); break;
        case JOB_VERIFY_BLOB_BATCH: {
            const uint64_t n = j->n ? j->n : 1;
            const uint8_t **b = malloc(n * sizeof *b), **c = malloc(n * sizeof *c), **p = malloc(n * sizeof *p);
            for (uint64_t k = 0; k < j->n; k++) { b[k] = j->blobs_in + k * BLOB; c[k] = j->commitments_in + k * G1; p[k] = j->proofs_in + k * G1; }
            j->res = eth_kzg_verify_blob_kzg_proof_batch(j->ctx, j->n, b, j->n, c, j->n, p, &j->verified);
            free(b); free(c); free(p);
            break;
        }
        default: {
            const uint64_t n = j->n ? j->n : 1;
            const uint8_t **c = malloc(n * sizeof *c), **l = malloc(n * sizeof *l), **p = malloc(n * sizeof *p);
            for (uint64_t k = 0; k < j->n; k++) { c[k] = j->commitments_in + k * G1; l[k] = j->cells_in + k * CELL; p[k] = j->proofs_in + k * G1; }
            j->res = eth_kzg_verify_cell_kzg_proof_batch(j->ctx, j->n, c, j->n, j->indices, j->n, l, j->n, p, &j->verified);
            free(c); free(l); free(p);
        }
    }
}

/* ---- JS <-> C helpers ------------------------------------------------------------------------------------------ */
static bool get_bytes(napi_env env, napi_value v, size_t want, const char *what, uint8_t **out) {
    bool is_ta = false;
    napi_typedarray_type t;
    size_t len = 0;
    void *data = NULL;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, v, &t, &len, &data, NULL, NULL) != napi_ok || t != napi_uint8_array || len != want) {
        char msg[128];
        snprintf(msg, sizeof msg, "%s must be a Uint8Array of %zu bytes", what, want);
        napi_throw_error(env, NULL, msg);
        return false;
    }
    *out = malloc(want ? want : 1);
    memcpy(*out, data, want);
    return true;
}
/* array of Uint8Array(item) -> one flat malloc'ed buffer; *n receives the count */
static bool get_byte_arrays(napi_env env, napi_value arr, size_t item, const char *what, uint8_t **out, uint64_t *n) {
    bool is_arr = false;
    uint32_t len = 0;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, arr, &len) != napi_ok) {
        napi_throw_error(env, NULL, "expected an array of Uint8Array");
        return false;
    }
    uint8_t *buf = malloc((size_t)(len ? len : 1) * item);
    for (uint32_t i = 0; i < len; i++) {
        napi_value e;
        uint8_t *one = NULL;
        if (napi_get_element(env, arr, i, &e) != napi_ok || !get_bytes(env, e, item, what, &one)) { free(buf); return false; }
        memcpy(buf + (size_t)i * item, one, item);
        free(one);
    }
    *out = buf;
    *n = len;
    return true;
}
static bool get_indices(napi_env env, napi_value arr, uint64_t **out, uint64_t *n) {  /* Array<number | bigint> */
    bool is_arr = false;
    uint32_t len = 0;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, arr, &len) != napi_ok) {
        napi_throw_error(env, NULL, "cellIndices must be an array");
        return false;
    }
    uint64_t *buf = malloc((size_t)(len ? len : 1) * sizeof *buf);
    for (uint32_t i = 0; i < len; i++) {
        napi_value e;
        napi_valuetype t;
        bool ok = napi_get_element(env, arr, i, &e) == napi_ok && napi_typeof(env, e, &t) == napi_ok;
        if (ok && t == napi_bigint) {
            bool lossless = false;
            ok = napi_get_value_bigint_uint64(env, e, &buf[i], &lossless) == napi_ok && lossless;
        } else if (ok && t == napi_number) {
            int64_t v = -1;
            ok = napi_get_value_int64(env, e, &v) == napi_ok && v >= 0;
            buf[i] = (uint64_t)v;
        } else ok = false;
        if (!ok) { free(buf); napi_throw_error(env, NULL, "cellIndices must hold non-negative numbers or bigints"); return false; }
    }
    *out = buf;
    *n = len;
    return true;
}
static napi_value make_u8(napi_env env, const uint8_t *src, size_t len) {
    void *data = NULL;
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, len, &data, &ab));
    memcpy(data, src, len);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, len, ab, 0, &ta));
    return ta;
}
static napi_value make_u8_array(napi_env env, const uint8_t *src, size_t count, size_t item) {
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, count, &arr));
    for (size_t i = 0; i < count; i++) {
        napi_value e = make_u8(env, src + i * item, item);
        if (!e) return NULL;
        NAPI_OK(napi_set_element(env, arr, (uint32_t)i, e));
    }
    return arr;
}
static const char *job_name(int kind) {
    static const char *names[] = {"blob_to_kzg_commitment", "compute_cells_and_kzg_proofs", "compute_cells", "recover_cells_and_kzg_proofs",
                                  "verify_cell_kzg_proof_batch", "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof",
                                  "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch"};
    return names[kind];
}
/* the JS value of a finished job, or NULL with *err set to an Error object */
static napi_value job_result(napi_env env, Job *j, napi_value *err) {
    *err = NULL;
    if (j->res.status != Ok) {
        char msg[512];
        snprintf(msg, sizeof msg, "failed to compute %s: %s", job_name(j->kind), j->res.error_msg ? j->res.error_msg : "error");
        napi_value m;
        if (napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &m) == napi_ok) napi_create_error(env, NULL, m, err);
        return NULL;
    }
    switch (j->kind) {
        case JOB_COMMIT: return make_u8(env, j->commitment, G1);
        case JOB_CELLS: return make_u8_array(env, j->cells_out, CELLS, CELL);
        case JOB_VERIFY: case JOB_VERIFY_KZG: case JOB_VERIFY_BLOB: case JOB_VERIFY_BLOB_BATCH: {
            napi_value b;
            NAPI_OK(napi_get_boolean(env, j->verified, &b));
            return b;
        }
        case JOB_BLOB_PROOF: return make_u8(env, j->proof_out, G1);
        case JOB_KZG_PROOF: {  /* [proof, y] (index.d.ts: Array<Uint8Array>) */
            napi_value arr, p = make_u8(env, j->proof_out, G1), y = make_u8(env, j->y_out, 32);
            if (!p || !y) return NULL;
            NAPI_OK(napi_create_array_with_length(env, 2, &arr));
            NAPI_OK(napi_set_element(env, arr, 0, p));
            NAPI_OK(napi_set_element(env, arr, 1, y));
            return arr;
        }
        default: {
            napi_value obj, c = make_u8_array(env, j->cells_out, CELLS, CELL), p = make_u8_array(env, j->proofs_out, CELLS, G1);
            if (!c || !p) return NULL;
            NAPI_OK(napi_create_object(env, &obj));
            NAPI_OK(napi_set_named_property(env, obj, "cells", c));
            NAPI_OK(napi_set_named_property(env, obj, "proofs", p));
            return obj;
        }
    }
}

static void async_execute(napi_env env, void *data) { (void)env; job_run((Job *)data); }
static void async_complete(napi_env env, napi_status status, void *data) {
    Job *j = data;
    napi_value err = NULL, val = status == napi_ok ? job_result(env, j, &err) : NULL;
    if (val) napi_resolve_deferred(env, j->deferred, val);
    else {
        if (!err) {
            bool pending = false;
            if (napi_is_exception_pending(env, &pending) == napi_ok && pending) napi_get_and_clear_last_exception(env, &err);
            else { napi_value m; napi_create_string_utf8(env, "eth_kzg: conversion failed", NAPI_AUTO_LENGTH, &m); napi_create_error(env, NULL, m, &err); }
        }
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_async_work(env, j->work);
    job_free(j);
}

/* one entry point per (kind, sync | async): data = kind * 2 + is_async */
static napi_value method(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4], self;
    void *data = NULL;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, &data));
    const int kind = (int)((intptr_t)data >> 1), is_async = (int)((intptr_t)data & 1);
    DASContext **holder = NULL;
    NAPI_OK(napi_unwrap(env, self, (void **)&holder));
    Job *j = calloc(1, sizeof *j);
    j->kind = kind;
    j->ctx = *holder;
    bool ok = true;
    if (kind <= JOB_CELLS) {
        ok = argc >= 1 && get_bytes(env, argv[0], BLOB, "blob", &j->blob);
    } else if (kind == JOB_RECOVER) {
        uint64_t ni = 0;
        ok = argc >= 2 && get_indices(env, argv[0], &j->indices, &ni) && get_byte_arrays(env, argv[1], CELL, "cell", &j->cells_in, &j->n);
        if (ok && ni != j->n) { napi_throw_error(env, NULL, "cellIndices and cells differ in length"); ok = false; }
    } else if (kind >= JOB_KZG_PROOF) {
        /* fixed-size operands copied into j->small / j->commitment; `take` fetches one of them */
        #define TAKE(arg, size, what, dst) do { uint8_t *t_ = NULL; ok = ok && argc > (arg) && get_bytes(env, argv[arg], size, what, &t_); if (ok) { memcpy(dst, t_, size); } free(t_); } while (0)
        if (kind == JOB_KZG_PROOF) { ok = argc >= 2 && get_bytes(env, argv[0], BLOB, "blob", &j->blob); TAKE(1, 32, "z", j->small[0]); }
        else if (kind == JOB_BLOB_PROOF) { ok = argc >= 2 && get_bytes(env, argv[0], BLOB, "blob", &j->blob); TAKE(1, G1, "commitment", j->small[0]); }
        else if (kind == JOB_VERIFY_KZG) { TAKE(0, G1, "commitment", j->small[0]); TAKE(1, 32, "z", j->small[1]); TAKE(2, 32, "y", j->small[2]); TAKE(3, G1, "proof", j->commitment); }
        else if (kind == JOB_VERIFY_BLOB) { ok = argc >= 3 && get_bytes(env, argv[0], BLOB, "blob", &j->blob); TAKE(1, G1, "commitment", j->small[0]); TAKE(2, G1, "proof", j->small[1]); }
        else {
            uint64_t nc = 0, np = 0;
            ok = argc >= 3 && get_byte_arrays(env, argv[0], BLOB, "blob", &j->blobs_in, &j->n) &&
                 get_byte_arrays(env, argv[1], G1, "commitment", &j->commitments_in, &nc) && get_byte_arrays(env, argv[2], G1, "proof", &j->proofs_in, &np);
            if (ok && (nc != j->n || np != j->n)) { napi_throw_error(env, NULL, "blobs, commitments and proofs differ in length"); ok = false; }
        }
        #undef TAKE
    } else {
        uint64_t ni = 0, nl = 0, np = 0;
        ok = argc >= 4 && get_byte_arrays(env, argv[0], G1, "commitment", &j->commitments_in, &j->n) && get_indices(env, argv[1], &j->indices, &ni) &&
             get_byte_arrays(env, argv[2], CELL, "cell", &j->cells_in, &nl) && get_byte_arrays(env, argv[3], G1, "proof", &j->proofs_in, &np);
        if (ok && (ni != j->n || nl != j->n || np != j->n)) { napi_throw_error(env, NULL, "commitments, cellIndices, cells and proofs differ in length"); ok = false; }
    }
    if (!ok) {
        bool pending = false;
        if (napi_is_exception_pending(env, &pending) == napi_ok && !pending) napi_throw_error(env, NULL, "wrong arguments");
        job_free(j);
        return NULL;
    }
    if (kind == JOB_CELLS_PROOFS || kind == JOB_CELLS || kind == JOB_RECOVER) j->cells_out = malloc((size_t)CELLS * CELL);
    if (kind == JOB_CELLS_PROOFS || kind == JOB_RECOVER) j->proofs_out = malloc((size_t)CELLS * G1);
    if (!is_async) {
        job_run(j);
        napi_value err = NULL, val = job_result(env, j, &err);
        job_free(j);
        if (!val && err) napi_throw(env, err);
        return val;
    }
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "eth_kzg_async", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, NULL, name, async_execute, async_complete, j, &j->work));
    NAPI_OK(napi_queue_async_work(env, j->work));
    return promise;
}

/* ---- class DasContextJs ----------------------------------------------------------------------------------------- */
static void ctx_finalize(napi_env env, void *data, void *hint) {
    (void)env; (void)hint;
    DASContext **holder = data;
    eth_kzg_das_context_free(*holder);
    free(holder);
}
static napi_value ctor(napi_env env, napi_callback_info info) {  /* new DasContextJs([usePrecomp = true]) */
    size_t argc = 1;
    napi_value argv[1], self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, NULL));
    bool use_precomp = true;
    if (argc >= 1) {
        napi_valuetype t;
        if (napi_typeof(env, argv[0], &t) == napi_ok && t == napi_boolean) napi_get_value_bool(env, argv[0], &use_precomp);
    }
    DASContext **holder = malloc(sizeof *holder);
    *holder = eth_kzg_das_context_new(use_precomp);
    NAPI_OK(napi_wrap(env, self, holder, ctx_finalize, NULL, NULL));
    return self;
}
static napi_ref g_ctor;
static napi_value create(napi_env env, napi_callback_info info) {  /* DasContextJs.create({usePrecomp}) */
    size_t argc = 1;
    napi_value argv[1], cons, inst, flag;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    bool use_precomp = true;
    if (argc >= 1) {
        napi_value v;
        bool has = false;
        if (napi_has_named_property(env, argv[0], "usePrecomp", &has) == napi_ok && has &&
            napi_get_named_property(env, argv[0], "usePrecomp", &v) == napi_ok)
            napi_get_value_bool(env, v, &use_precomp);
    }
    NAPI_OK(napi_get_reference_value(env, g_ctor, &cons));
    NAPI_OK(napi_get_boolean(env, use_precomp, &flag));
    NAPI_OK(napi_new_instance(env, cons, 1, &flag, &inst));
    return inst;
}

#define METHOD(name, kind, is_async) {name, NULL, method, NULL, NULL, NULL, napi_default, (void *)(intptr_t)((kind) * 2 + (is_async))}
static napi_value init(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        METHOD("blobToKzgCommitment", JOB_COMMIT, 0), METHOD("asyncBlobToKzgCommitment", JOB_COMMIT, 1),
        METHOD("computeCellsAndKzgProofs", JOB_CELLS_PROOFS, 0), METHOD("asyncComputeCellsAndKzgProofs", JOB_CELLS_PROOFS, 1),
        METHOD("computeCells", JOB_CELLS, 0), METHOD("asyncComputeCells", JOB_CELLS, 1),
        METHOD("recoverCellsAndKzgProofs", JOB_RECOVER, 0), METHOD("asyncRecoverCellsAndKzgProofs", JOB_RECOVER, 1),
        METHOD("verifyCellKzgProofBatch", JOB_VERIFY, 0), METHOD("asyncVerifyCellKzgProofBatch", JOB_VERIFY, 1),
        METHOD("computeKzgProof", JOB_KZG_PROOF, 0), METHOD("asyncComputeKzgProof", JOB_KZG_PROOF, 1),
        METHOD("computeBlobKzgProof", JOB_BLOB_PROOF, 0), METHOD("asyncComputeBlobKzgProof", JOB_BLOB_PROOF, 1),
        METHOD("verifyKzgProof", JOB_VERIFY_KZG, 0), METHOD("asyncVerifyKzgProof", JOB_VERIFY_KZG, 1),
        METHOD("verifyBlobKzgProof", JOB_VERIFY_BLOB, 0), METHOD("asyncVerifyBlobKzgProof", JOB_VERIFY_BLOB, 1),
        METHOD("verifyBlobKzgProofBatch", JOB_VERIFY_BLOB_BATCH, 0), METHOD("asyncVerifyBlobKzgProofBatch", JOB_VERIFY_BLOB_BATCH, 1),
        {"create", NULL, create, NULL, NULL, NULL, napi_static, NULL},
    };
    napi_value cls;
    NAPI_OK(napi_define_class(env, "DasContextJs", NAPI_AUTO_LENGTH, ctor, NULL, sizeof props / sizeof props[0], props, &cls));
    NAPI_OK(napi_create_reference(env, cls, 1, &g_ctor));
    NAPI_OK(napi_set_named_property(env, exports, "DasContextJs", cls));
    const struct { const char *name; uint32_t v; } consts[] = {
        {"BYTES_PER_COMMITMENT", G1}, {"BYTES_PER_PROOF", G1}, {"BYTES_PER_FIELD_ELEMENT", 32}, {"BYTES_PER_BLOB", BLOB},
        {"MAX_NUM_COLUMNS", CELLS}, {"BYTES_PER_CELL", CELL}};
    for (size_t i = 0; i < sizeof consts / sizeof consts[0]; i++) {
        napi_value v;
        NAPI_OK(napi_create_uint32(env, consts[i].v, &v));
        NAPI_OK(napi_set_named_property(env, exports, consts[i].name, v));
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
