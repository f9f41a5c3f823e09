/*
 * c_eth_kzg.h -- C ABI of the MI355X-native KZG engine.
 *
 * Drop-in for the header cbindgen generates for the reference's `c_eth_kzg` crate
 * (reference: bindings/c/build.rs:24-29; symbols in bindings/c/src/lib.rs, mirrored by
 * bindings/nim/nim_code/nim_eth_kzg/header.nim and bindings/csharp/.../native_methods.g.cs).
 * Every `eth_kzg_*` prototype below has the reference's exact signature and calling
 * convention; the citation after each one is the Rust definition it replaces.
 * The `eth_kzg_amd_*` entry points are non-breaking additions: batched and
 * device-resident forms of the same operations, which is what a GPU needs to be fed.
 *
 * Ownership: inputs are borrowed for the duration of the call; outputs are written into
 * caller-allocated memory; only `CResult.error_msg` and the context are library-allocated
 * (free with eth_kzg_free_error_message / eth_kzg_das_context_free).
 * Threading: a context may be used from many threads at once.  Prover calls take one of several scratch sets; the
 * recovery / commitment / EIP-4844 calls run on up to $ETH_KZG_AMD_SERIAL_LANES (default 4) engine lanes that the context
 * creates on demand, so calls from different threads overlap on the GPU; of concurrent eth_kzg_verify_cell_kzg_proof_batch
 * callers one at a time takes the latency path and those that arrive meanwhile are combined into many-verification passes
 * behind the ABI (three pass slots; same verdicts and error split).
 */
#ifndef C_ETH_KZG_H
#define C_ETH_KZG_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bindings/c/src/lib.rs:119-123 */
typedef enum CResultStatus {
    Ok,
    Err,
} CResultStatus;

/* bindings/c/src/lib.rs:128-132 */
typedef struct CResult {
    CResultStatus status;
    char *error_msg;
} CResult;

/* bindings/c/src/lib.rs:50-52 (opaque) */
typedef struct DASContext DASContext;

/* bindings/c/src/lib.rs:79-92.  use_precomp = true selects precomputed window tables (the reference: width 8,
 * RECOMMENDED_PRECOMP_WIDTH, on the CPU; here the widest tables that fit in HBM: a GLV table of 16-bit windows for FK20,
 * 206 GB, and one of nine windows for commitments, 35 GB, when $ETH_KZG_AMD_TABLE_GB=max; by default the tables take 108 GB
 * -- nine windows each --; narrower automatically when memory is short or the budget bounds them), false the 2.4 GB sixteen-window tables and nothing wider (the
 * tables a use_precomp = true context starts on); results are identical.
 * Progressive start: the call returns as soon as small start tables are up (about 0.5 s) and every entry point works from
 * then on; the wide tables are built by a helper thread, in pieces of under a gigabyte so that no other HIP call of the process
 * waits long, and taken into use group by group (eth_kzg_amd_tables_ready, eth_kzg_amd_table_groups_ready;
 * $ETH_KZG_AMD_PROGRESSIVE=0 builds them before returning).  The embedded mainnet trusted setup is loaded; GPU 0 (or the
 * ordinal in $ETH_KZG_AMD_DEVICE) is used.  Aborts if no MI355X-class GPU is usable: there is no CPU fallback. */
DASContext *eth_kzg_das_context_new(bool use_precomp);

/* bindings/c/src/lib.rs:109-116.  NULL-safe. */
void eth_kzg_das_context_free(DASContext *ctx);

/* bindings/c/src/lib.rs:171-180.  NULL-safe. */
void eth_kzg_free_error_message(char *c_message);

/* bindings/c/src/lib.rs:196-205 -> blob_to_kzg_commitment.rs:8-38.
 * blob: 131072 bytes; out: 48 bytes. */
CResult eth_kzg_blob_to_kzg_commitment(const DASContext *ctx, const uint8_t *blob, uint8_t *out);

/* bindings/c/src/lib.rs:226-236 -> compute_cells_and_kzg_proofs.rs:8-33.
 * out_cells: 128 pointers to 2048-byte buffers; out_proofs: 128 pointers to 48-byte buffers. */
CResult eth_kzg_compute_cells_and_kzg_proofs(const DASContext *ctx, const uint8_t *blob, uint8_t **out_cells,
                                             uint8_t **out_proofs);

/* bindings/c/src/lib.rs:255-263 */
CResult eth_kzg_compute_cells(const DASContext *ctx, const uint8_t *blob, uint8_t **out_cells);

/* bindings/c/src/lib.rs:309-335 -> verify_cells_and_kzg_proofs_batch.rs:9-49.
 * commitments: one 48-byte commitment per cell (NOT deduplicated). An invalid proof is
 * `Ok` with *verified = false; malformed input is `Err` (lib.rs:272-280). A zero length means
 * the matching pointer is not dereferenced. */
CResult eth_kzg_verify_cell_kzg_proof_batch(const DASContext *ctx, uint64_t commitments_length,
                                            const uint8_t *const *commitments, uint64_t cell_indices_length,
                                            const uint64_t *cell_indices, uint64_t cells_length,
                                            const uint8_t *const *cells, uint64_t proofs_length,
                                            const uint8_t *const *proofs, bool *verified);

/* bindings/c/src/lib.rs:366-386 -> recover_cells_and_kzg_proofs.rs:10-39 */
CResult eth_kzg_recover_cells_and_proofs(const DASContext *ctx, uint64_t cells_length, const uint8_t *const *cells,
                                         uint64_t cell_indices_length, const uint64_t *cell_indices,
                                         uint8_t **out_cells, uint8_t **out_proofs);

/* bindings/c/src/lib.rs:395-405 */
uint64_t eth_kzg_constant_bytes_per_cell(void);
uint64_t eth_kzg_constant_bytes_per_proof(void);
uint64_t eth_kzg_constant_cells_per_ext_blob(void);

/* EIP-4844 single-point operations (bindings/c/src/lib.rs:423-566 -> compute_kzg_proof.rs, compute_blob_kzg_proof.rs,
 * verify_kzg_proof.rs, verify_blob_kzg_proof.rs, verify_blob_kzg_proof_batch.rs), on the same GPU kernels.
 * z, y: 32-byte big-endian field elements; invalid proof => `Ok` + *verified = false; malformed input => `Err`. */
CResult eth_kzg_compute_kzg_proof(const DASContext *ctx, const uint8_t *blob, const uint8_t *z, uint8_t *out_proof,
                                  uint8_t *out_y);
CResult eth_kzg_compute_blob_kzg_proof(const DASContext *ctx, const uint8_t *blob, const uint8_t *commitment,
                                       uint8_t *out_proof);
CResult eth_kzg_verify_kzg_proof(const DASContext *ctx, const uint8_t *commitment, const uint8_t *z, const uint8_t *y,
                                 const uint8_t *proof, bool *verified);
CResult eth_kzg_verify_blob_kzg_proof(const DASContext *ctx, const uint8_t *blob, const uint8_t *commitment,
                                      const uint8_t *proof, bool *verified);
CResult eth_kzg_verify_blob_kzg_proof_batch(const DASContext *ctx, uint64_t blobs_length, const uint8_t *const *blobs,
                                            uint64_t commitments_length, const uint8_t *const *commitments,
                                            uint64_t proofs_length, const uint8_t *const *proofs, bool *verified);

/* ---------------------------------------------------------------------------------------------
 * MI355X additions (no counterpart in the reference; its single-blob ABI cannot feed a GPU).
 * Status arrays receive one int per blob: 0 ok, 1 = a field element >= r, 2 = bad G1 point,
 * 3 = invalid input, 4 = recovery failed.  The CResult is `Err` only for call-level failures.
 */

/* Context on an explicit GPU ordinal (one process per GPU under torch.distributed/RCCL). */
DASContext *eth_kzg_amd_das_context_new_on_device(bool use_precomp, int device_ordinal);

/* The constructor that never kills the host process.  eth_kzg_das_context_new keeps the reference's behaviour -- the Rust
 * constructor panics across the FFI when it cannot build a context (bindings/c/src/lib.rs:79-92), this library aborts when there
 * is no usable GPU or not even the 2.4 GB start tables fit -- which is hostile inside a consensus client.  This form returns NULL
 * and fills *result (status Err, error_msg to be freed with eth_kzg_free_error_message; may be NULL) instead; on success *result
 * is Ok.  table_budget_gb bounds the HBM the two window tables take together: > 0 = that many GB (the commitment table takes the
 * widest GLV table within a third of it, the FK20 table the widest within the rest: 108 GB -> nine windows each, 44 -> ten, 22 ->
 * eleven, 3 -> sixteen), 0 = $ETH_KZG_AMD_TABLE_GB or else the default of 108 GB (-7 % against the widest tables at 44 % of their
 * memory), < 0 = whatever the HBM still holds (242 GB on an idle GPU: eight windows of 16 bits for FK20, nine for commitments). */
DASContext *eth_kzg_amd_das_context_try_new(bool use_precomp, int device_ordinal, double table_budget_gb, CResult *result);

/* A context over a DEVICE LIST: one engine per listed GPU behind the one pointer, so that a host written against the
 * reference -- one context, shared by all its threads (bindings/c/src/lib.rs:79-92, bindings/node/src/lib.rs:35,75,
 * bindings/golang/prover.go) -- uses every GPU of the node through the UNCHANGED sixteen symbols:
 *   - a single-problem call (eth_kzg_compute_cells_and_kzg_proofs, _verify_cell_kzg_proof_batch, _recover_cells_and_proofs,
 *     _blob_to_kzg_commitment, the EIP-4844 calls) runs on the device with the least work in flight;
 *   - a host-pointer batch (eth_kzg_amd_*_batch, _verify_cell_kzg_proof_batch_many) is cut into contiguous slices, slice d =
 *     [n d / D, n (d + 1) / D) on the d-th listed device, one host thread each, the caller's buffers are the gather target
 *     (no collective); the call's error is the lowest failing device's;
 *   - a device-resident call (eth_kzg_amd_*_device) runs on the listed device that owns the caller's buffers (`Err` if none does);
 *   - the communicator, profiling and table-introspection calls address the FIRST listed device (eth_kzg_amd_table_bytes sums
 *     the distinct devices, eth_kzg_amd_tables_ready reports the slowest one).
 * Results are byte-identical to a one-device context's.  eth_kzg_das_context_new reads the list from ETH_KZG_AMD_DEVICES
 * ("0,1,2,3" or "all"; unset: ETH_KZG_AMD_DEVICE, else GPU 0).  An ordinal may repeat (two engines on one GPU share its window
 * tables): the form the one-GPU tests use.  Returns NULL and fills *result (as _try_new) when any device fails. */
DASContext *eth_kzg_amd_das_context_new_on_devices(bool use_precomp, const int32_t *device_ordinals, uint64_t n_devices,
                                                   double table_budget_gb, CResult *result);
/* the context's device list: writes up to `capacity` ordinals, returns how many devices the context spans */
uint64_t eth_kzg_amd_context_devices(const DASContext *ctx, int32_t *out_ordinals, uint64_t capacity);
/* Version of the eth_kzg_amd_* additions (the sixteen eth_kzg_* symbols never change): 6 = this header.  Bumped whenever a
 * symbol's meaning changes or one is removed, so that an out-of-tree consumer can check at load time. */
int eth_kzg_amd_abi_version(void);

/* Host-pointer batches: n blobs; out_cells[b] / out_proofs[b] are arrays of 128 pointers as in the
 * single-blob calls (either may be NULL to skip that output). */
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_batch(const DASContext *ctx, uint64_t n, const uint8_t *const *blobs,
                                                       uint8_t *const *const *out_cells,
                                                       uint8_t *const *const *out_proofs, int32_t *status);
CResult eth_kzg_amd_blob_to_kzg_commitment_batch(const DASContext *ctx, uint64_t n, const uint8_t *const *blobs,
                                                 uint8_t *const *out, int32_t *status);

/* n independent recoveries in one pass (BASELINE.json config 5). Blob b supplies cells_lengths[b] cells
 * (cells[b][k], 2048 B each) with indices cell_indices[b][k]; outputs as in the batch call above. status[b] = 0 ok,
 * 1 a cell holds a non-canonical field element, 3 invalid indices / counts, 4 the cells are not consistent. */
CResult eth_kzg_amd_recover_cells_and_proofs_batch(const DASContext *ctx, uint64_t n, const uint64_t *cells_lengths,
                                                   const uint8_t *const *const *cells,
                                                   const uint64_t *cell_indices_lengths,
                                                   const uint64_t *const *cell_indices, uint8_t *const *const *out_cells,
                                                   uint8_t *const *const *out_proofs, int32_t *status);

/* Device-resident recovery (BASELINE.json config 5 for pipelines that hold extended blobs in HBM): d_cells is the flat
 * [n][128][2048] layout on this GPU; present_masks (HOST memory) holds two 64-bit words per blob, bit (c % 64) of word
 * (c / 64) set when cell c carries data - missing cells are never read.  Replaces the cell list + index list of
 * eth_kzg_recover_cells_and_proofs (bindings/c/src/lib.rs:366): ascending unique indices hold by construction, a blob
 * needs at least 64 present cells.  Outputs as in eth_kzg_amd_compute_cells_and_kzg_proofs_device (either may be NULL);
 * status[b] (HOST, n entries) = 0 ok, 1 non-canonical field element, 3 fewer than 64 cells, 4 cells not consistent with
 * a degree < 4096 polynomial; outputs of a failed blob are unspecified.  The decode runs on the library's stream, ordered
 * after the work already queued on hip_stream (event) and completed before the call returns (the status words come back
 * from it); the cells and proofs kernels are enqueued on hip_stream behind it (NULL: library stream, synchronised). */
CResult eth_kzg_amd_recover_cells_and_proofs_device(const DASContext *ctx, uint64_t n, const uint8_t *d_cells,
                                                    const uint64_t *present_masks, uint8_t *d_out_cells,
                                                    uint8_t *d_out_proofs, int32_t *status, void *hip_stream);

/* verify_cell_kzg_proof_batch sharded over GPUs (BASELINE.json config 3).  The verification
 * equation of crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:139-260 is linear in the cells once the
 * Fiat-Shamir challenge (taken over the WHOLE batch) is fixed, so each rank evaluates its slice
 * [shard_begin, shard_end) of the cell list and emits a 96-byte record (two compressed G1 partial sums);
 * the records of all ranks are all-gathered and handed to _combine, which adds them and runs the one pairing check.
 * Arguments before shard_begin are exactly those of eth_kzg_verify_cell_kzg_proof_batch and must describe the whole
 * batch on every rank.  Malformed input in the rank's slice (or in the shared lengths / indices) => Err.
 * The union of the slices must be [0, cells_length); an empty slice is fine. */
#define ETH_KZG_AMD_VERIFY_PARTIAL_BYTES 96
CResult eth_kzg_amd_verify_cell_kzg_proof_batch_partial(const DASContext *ctx, uint64_t commitments_length,
                                                        const uint8_t *const *commitments, uint64_t cell_indices_length,
                                                        const uint64_t *cell_indices, uint64_t cells_length,
                                                        const uint8_t *const *cells, uint64_t proofs_length,
                                                        const uint8_t *const *proofs, uint64_t shard_begin,
                                                        uint64_t shard_end, uint8_t *out_partial /* 96 */);
CResult eth_kzg_amd_verify_cell_kzg_proof_batch_combine(const DASContext *ctx, uint64_t n_partials,
                                                        const uint8_t *partials /* n_partials * 96 */, bool *verified);

/* MANY independent verifications in one call -- the throughput form of eth_kzg_verify_cell_kzg_proof_batch (the reference gets
 * its throughput by verifying from many threads on one context, bindings/node/src/lib.rs:92-299).  Problem b is described by
 * commitments[b] / cell_indices[b] / cells[b] / proofs[b] with the four lengths *_lengths[b], exactly as one call of the
 * single form; the problems share every GPU launch (one lane per scalar multiplication), their transcripts are hashed on host
 * threads in parallel, and the pairing checks of a pass are folded into one with 127-bit weights derived from all the
 * challenges (error 2^-127, the argument of the powers of r inside one batch); in a pass that contains wrong proofs they are
 * found by folding sub-ranges of the resident weighted sums (a handful of pairings per wrong proof instead of one per problem)
 * down to single problems, so a false verdict is always exact and per problem.  A problem may hold at most 2^24 - 1 cells.  Per problem: status[b] = 0 and verified[b] = the verdict, or
 * status[b] = 1 (a cell holds a non-canonical field element), 2 (bad G1 encoding / not in the subgroup), 3 (invalid lengths
 * or indices) and verified[b] = false -- what the single form reports as Err (status may be NULL).  The CResult is Err only
 * for a call-level (device) failure.  An empty problem verifies (verifier.rs:90-93). */
CResult eth_kzg_amd_verify_cell_kzg_proof_batch_many(const DASContext *ctx, uint64_t n_batches,
                                                     const uint64_t *commitments_lengths,
                                                     const uint8_t *const *const *commitments,
                                                     const uint64_t *cell_indices_lengths,
                                                     const uint64_t *const *cell_indices, const uint64_t *cells_lengths,
                                                     const uint8_t *const *const *cells, const uint64_t *proofs_lengths,
                                                     const uint8_t *const *const *proofs, bool *verified, int32_t *status);

/* Device-resident batches: flat buffers already in this GPU's HBM.
 *   d_blobs        n * 131072 bytes
 *   d_out_cells    n * 128 * 2048 bytes   (may be NULL)
 *   d_out_proofs   n * 128 * 48 bytes     (may be NULL)
 *   d_out          n * 48 bytes
 *   status         n ints in HOST memory, or NULL: then nothing is copied back and, if
 *                  `hip_stream` is non-NULL, the call returns without synchronising (work is
 *                  enqueued on that hipStream_t; NULL = the context's own stream, synchronised).
 * Inputs produced by work the caller has queued on `hip_stream` are waited for on the device; with NULL that is the
 * work queued so far on the DEFAULT stream (the library's own streams are non-blocking streams: without this a kernel
 * of the caller's default stream that writes d_blobs could still be running when the call reads them).
 * Calls on one stream are ordered.  The context owns a few sets of intermediate buffers; a call takes a free one and
 * makes its stream wait for the previous user of that set, so asynchronous calls on different streams are safe too
 * (they overlap while sets are free, otherwise they queue). */
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_device(const DASContext *ctx, uint64_t n, const uint8_t *d_blobs,
                                                        uint8_t *d_out_cells, uint8_t *d_out_proofs, int32_t *status,
                                                        void *hip_stream);
CResult eth_kzg_amd_blob_to_kzg_commitment_device(const DASContext *ctx, uint64_t n, const uint8_t *d_blobs,
                                                  uint8_t *d_out, int32_t *status, void *hip_stream);

/* verify_cell_kzg_proof_batch on flat arrays in this GPU's HBM: n * 48 commitment bytes (one per cell, NOT deduplicated, as in
 * the host form), n indices, n * 2048 cell bytes, n * 48 proof bytes.  The GPU's part reads cells and proofs where they are;
 * the transcript hash runs on a host core, so its bytes come down (in chunks, under the hash) and the call is synchronous;
 * work already queued on `hip_stream` (NULL: the default stream) that produces the inputs is waited for.  Result and error behaviour as eth_kzg_verify_cell_kzg_proof_batch. */
CResult eth_kzg_amd_verify_cell_kzg_proof_batch_device(const DASContext *ctx, uint64_t n, const uint8_t *d_commitments,
                                                       const uint64_t *d_cell_indices, const uint8_t *d_cells,
                                                       const uint8_t *d_proofs, bool *verified, void *hip_stream);

/* Introspection used by bench.py / DESIGN.md: bytes of both window tables resident in HBM; nominal window width w of the FK20
 * table in use (a GLV table: ceil(128 / w) windows over the two 128-bit halves of each scalar, packed 96-byte entries: 16
 * gathered additions per base at w = 16, 32 at w = 8 -- the start tables, and all there is with use_precomp = false). */
uint64_t eth_kzg_amd_table_bytes(const DASContext *ctx);
/* NOMINAL: since ABI version 5 the windows of a table have mixed widths (w = 15: two windows of 15 bits and seven of 14) -- do
 * not derive the additions per base from it; eth_kzg_amd_window_count gives the windows per 128-bit GLV half directly (gathered
 * additions per base = 2 x that: 16 on the widest tables, 18 on the default ones, 32 on the start tables). */
int eth_kzg_amd_window_bits(const DASContext *ctx);
int eth_kzg_amd_window_count(const DASContext *ctx);
/* DEPRECATED (kept for consumers linked against the round-4 header): always 1 -- every window table is a GLV table now. */
int eth_kzg_amd_glv_table(const DASContext *ctx);
/* 1 once the final window tables are in use, 0 while the context still runs on its start tables (waits up to wait_ms
 * milliseconds for the switch; negative: until it happened), 2 if the wide tables could not be built (memory) and the
 * context stays on what it has.  Results never depend on the table in use. */
int eth_kzg_amd_tables_ready(const DASContext *ctx, int wait_ms);
/* The wide FK20 table is allocated in pieces of under a gigabyte and used GROUP BY GROUP while the helper thread builds it (the
 * 128 MSM groups of a blob: the groups already built run on the wide table, the rest on the start table): this returns how
 * many of the 128 groups of the table under construction are in use (128 once it is complete or when nothing is being built). */
int eth_kzg_amd_table_groups_ready(const DASContext *ctx);
/* How the window tables in use were allocated: out4[0] = milliseconds spent in hipMalloc for their pieces, out4[1] = the longest
 * single allocation (ms), out4[2] = number of pieces, out4[3] = bytes.  On an idle GPU a piece takes 2 ms; a piece that takes
 * seconds is the driver waiting for memory that another process has just freed to be wiped, and the calling process's GPU
 * queues stand still for as long (tools/alloc_test/probe_stall.cpp): the one start-up stall the library cannot remove. */
void eth_kzg_amd_table_build_info(const DASContext *ctx, double *out4);
/* The compiled linear map that replaces the two G1 transforms of the prover: out4 = constant multiplications, point
 * additions, point doublings per blob, kernel launches per call. */
void eth_kzg_amd_linmap_info(const DASContext *ctx, int32_t *out4);

/* Per-stage HIP-event timing for bench.py's roofline leg.  Stages, in order: blob_to_coeffs,
 * coeffs_to_cells, fk20_scalars, msm_fixed, g1_ifft, g1_fft, compress, g1_linmap.  get_stage_times writes the
 * milliseconds and kernel-launch counts accumulated since the previous call (it synchronises the
 * device) and returns the number of stages. */
void eth_kzg_amd_set_profiling(const DASContext *ctx, int on);
int eth_kzg_amd_get_stage_times(const DASContext *ctx, double *ms, uint64_t *launches, int n);

/* ---- multi-GPU (north_star: blob batches shard across the GPUs of a node; one all-gather of the proof vectors) ----
 *
 * (1) One process per GPU (torchrun / MPI style).  The library owns the RCCL communicator: rank 0 obtains an id,
 *     the caller distributes its 128 bytes by whatever it already has (MPI_Bcast, torch.distributed, a file), every
 *     rank attaches its context, and eth_kzg_amd_all_gather is one ncclAllGather over xGMI on the given stream:
 *     d_recv receives world * bytes_per_rank bytes, rank r's slab at offset r * bytes_per_rank.  RCCL is bound with
 *     dlopen when the first of these functions is called, so the library itself does not depend on it: a copy the
 *     process has already mapped (PyTorch ships one) is reused, otherwise librccl.so.1 is loaded.
 *     eth_kzg_amd_comm_init is COLLECTIVE (ncclCommInitRank): every rank must enter it or none; call
 *     eth_kzg_amd_comm_probe first (local, cheap: Ok iff RCCL can be bound and the context has no communicator yet;
 *     optionally reports which library file was bound) and agree on the result before any rank calls _init.
 *     eth_kzg_amd_comm_info returns the rank and size the communicator itself reports (ncclCommUserRank/Count). */
CResult eth_kzg_amd_comm_probe(const DASContext *ctx, char *out_library_path /* may be NULL */, uint64_t path_capacity);
CResult eth_kzg_amd_comm_unique_id(uint8_t *out_id /* 128 */);
CResult eth_kzg_amd_comm_init(DASContext *ctx, const uint8_t *id /* 128 */, int rank, int world);
CResult eth_kzg_amd_comm_info(const DASContext *ctx, int *out_rank, int *out_world);
CResult eth_kzg_amd_all_gather(const DASContext *ctx, const void *d_send, void *d_recv, uint64_t bytes_per_rank,
                               void *hip_stream);
void eth_kzg_amd_comm_destroy(DASContext *ctx);
/* (2) One process, several GPUs: contexts[d] created with eth_kzg_amd_das_context_new_on_device(.., d).  The batch
 *     is cut into contiguous slices, one per context, each served by its own host thread through the host-pointer
 *     path above; the caller's buffers are the gather target, so there is no collective.  status as in the batch call. */
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi(const DASContext *const *contexts, uint64_t n_contexts,
                                                             uint64_t n, const uint8_t *const *blobs,
                                                             uint8_t *const *const *out_cells,
                                                             uint8_t *const *const *out_proofs, int32_t *status);
/* number of GPUs visible to the library (hipGetDeviceCount) */
int eth_kzg_amd_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* C_ETH_KZG_H */
