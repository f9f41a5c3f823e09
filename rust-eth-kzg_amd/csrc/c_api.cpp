// C ABI of the MI355X KZG engine: the reference's `eth_kzg_*` symbols (bindings/c/src/lib.rs) over
// kzg::Engine, plus the batched / device-resident `eth_kzg_amd_*` additions.  See include/c_eth_kzg.h.
#include "../../include/c_eth_kzg.h"
#include "../../include/c_eth_kzg_test_hooks.h"
#include "engine.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct DASContext {
    kzg::Engine* engine;
};

namespace {

constexpr int CELLS = 128;

CResult ok() { return CResult{Ok, nullptr}; }
CResult err(const std::string& m) {  // CResult::with_error, bindings/c/src/lib.rs:146-153
    char* s = (char*)malloc(m.size() + 1);
    memcpy(s, m.c_str(), m.size() + 1);
    return CResult{Err, s};
}
const char* status_text(int st) {
    switch (st) {
        case kzg::ERR_SCALAR: return "Serialization(CouldNotDeserializeScalar)";
        case kzg::ERR_G1: return "Serialization(CouldNotDeserializeG1Point)";
        case kzg::ERR_INPUT: return "InvalidInput";
        case kzg::ERR_RECOVERY: return "Recovery(PolynomialHasInvalidLength)";
        default: return "DeviceError";
    }
}
kzg::Engine* eng(const DASContext* ctx) {
    // `assert!(!ctx.is_null())` in the reference (e.g. compute_cells_and_kzg_proofs.rs:14): abort, like a Rust panic across FFI
    if (!ctx || !ctx->engine) {
        fprintf(stderr, "c_eth_kzg: context pointer is null\n");
        abort();
    }
    return ctx->engine;
}
CResult device_err(kzg::Engine* e) { return err("DeviceError(" + e->last_error() + ")"); }

// the one constructor: NULL + message when the context cannot be built (no usable GPU, not even the 3.7 GB start tables fit)
DASContext* try_make_ctx(bool use_precomp, int device, double table_budget_gb, std::string* why) {
    DASContext* c = nullptr;
    try {
        c = new DASContext{nullptr};
        c->engine = new kzg::Engine(use_precomp, device, nullptr, table_budget_gb);
        return c;
    } catch (const std::exception& e) {
        delete c;
        if (why) *why = e.what();
        return nullptr;
    }
}
DASContext* make_ctx(bool use_precomp, int device) {
    std::string why;
    DASContext* c = try_make_ctx(use_precomp, device, 0, &why);
    if (!c) {
        // The reference panics when the context cannot be built (bad SRS); here: no usable GPU.  No CPU fallback.  A host that must
        // not die calls eth_kzg_amd_das_context_try_new instead.
        fprintf(stderr, "c_eth_kzg: cannot create the MI355X context: %s\n", why.c_str());
        abort();
    }
    return c;
}

}  // namespace

extern "C" {

DASContext* eth_kzg_das_context_new(bool use_precomp) {
    return make_ctx(use_precomp, kzg::Knobs::from_env().device);  // ETH_KZG_AMD_DEVICE (knobs.hpp), read here, once per context
}
DASContext* eth_kzg_amd_das_context_new_on_device(bool use_precomp, int device_ordinal) {
    return make_ctx(use_precomp, device_ordinal);
}
DASContext* eth_kzg_amd_das_context_try_new(bool use_precomp, int device_ordinal, double table_budget_gb, CResult* result) {
    std::string why;
    DASContext* c = try_make_ctx(use_precomp, device_ordinal, table_budget_gb, &why);
    if (result) *result = c ? ok() : err("ContextCreation(" + why + ")");
    return c;
}
void eth_kzg_das_context_free(DASContext* ctx) {
    if (!ctx) return;
    eth_kzg_amd_comm_destroy(ctx);
    delete ctx->engine;
    delete ctx;
}
void eth_kzg_free_error_message(char* c_message) {
    if (c_message) free(c_message);
}

CResult eth_kzg_blob_to_kzg_commitment(const DASContext* ctx, const uint8_t* blob, uint8_t* out) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int st = 0;
    const uint8_t* blobs[1] = {blob};
    uint8_t* outs[1] = {out};
    if (e->blob_to_kzg_commitment_host(1, blobs, outs, &st)) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_compute_cells_and_kzg_proofs(const DASContext* ctx, const uint8_t* blob, uint8_t** out_cells,
                                             uint8_t** out_proofs) {
    kzg::Engine* e = eng(ctx);
    int st = 0;
    const uint8_t* blobs[1] = {blob};
    uint8_t* const* cells[1] = {out_cells};
    uint8_t* const* proofs[1] = {out_proofs};
    if (e->compute_cells_and_kzg_proofs_host(1, blobs, cells, proofs, &st)) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_compute_cells(const DASContext* ctx, const uint8_t* blob, uint8_t** out_cells) {
    kzg::Engine* e = eng(ctx);
    int st = 0;
    const uint8_t* blobs[1] = {blob};
    uint8_t* const* cells[1] = {out_cells};
    if (e->compute_cells_and_kzg_proofs_host(1, blobs, cells, nullptr, &st)) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_verify_cell_kzg_proof_batch(const DASContext* ctx, uint64_t commitments_length,
                                            const uint8_t* const* commitments, uint64_t cell_indices_length,
                                            const uint64_t* cell_indices, uint64_t cells_length,
                                            const uint8_t* const* cells, uint64_t proofs_length,
                                            const uint8_t* const* proofs, bool* verified) {
    // up to one caller per engine lane runs the latency path; callers beyond that are combined into many-verification passes
    kzg::Engine* e = eng(ctx);
    int ver = 0;
    int st = e->verify_cell_kzg_proof_batch_combined(commitments_length, commitments, cell_indices_length, cell_indices,
                                                     cells_length, cells, proofs_length, proofs, &ver);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    if (st) return err(status_text(st));
    *verified = ver != 0;
    return ok();
}

CResult eth_kzg_amd_verify_cell_kzg_proof_batch_many(const DASContext* ctx, uint64_t n_batches, const uint64_t* commitments_lengths,
                                                     const uint8_t* const* const* commitments, const uint64_t* cell_indices_lengths,
                                                     const uint64_t* const* cell_indices, const uint64_t* cells_lengths,
                                                     const uint8_t* const* const* cells, const uint64_t* proofs_lengths,
                                                     const uint8_t* const* const* proofs, bool* verified, int32_t* status) {
    kzg::Engine* e = eng(ctx);  // own lock, stream and scratch inside (verify_many.hip): no lane needed
    if (n_batches == 0) return ok();
    if (n_batches > (1u << 24)) return err("InvalidInput");
    try {
        std::vector<int> ver(n_batches), st(n_batches);
        const int rc = e->verify_cell_kzg_proof_batch_many_host(n_batches, commitments_lengths, commitments, cell_indices_lengths,
                                                                cell_indices, cells_lengths, cells, proofs_lengths, proofs, ver.data(), st.data());
        if (rc == kzg::ERR_DEVICE) return device_err(e);
        for (uint64_t b = 0; b < n_batches; b++) {
            verified[b] = st[b] == 0 && ver[b] != 0;
            if (status) status[b] = st[b];
        }
        return ok();
    } catch (const std::exception& ex) {
        return err(std::string("DeviceError(") + ex.what() + ")");
    }
}

CResult eth_kzg_amd_verify_cell_kzg_proof_batch_device(const DASContext* ctx, uint64_t n, const uint8_t* d_commitments,
                                                       const uint64_t* d_cell_indices, const uint8_t* d_cells,
                                                       const uint8_t* d_proofs, bool* verified, void* hip_stream) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_cell_kzg_proof_batch_device(n, d_commitments, d_cell_indices, d_cells, d_proofs, &ver, (hipStream_t)hip_stream);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    if (st) return err(status_text(st));
    *verified = ver != 0;
    return ok();
}

CResult eth_kzg_amd_recover_cells_and_proofs_device(const DASContext* ctx, uint64_t n, const uint8_t* d_cells,
                                                    const uint64_t* present_masks, uint8_t* d_out_cells, uint8_t* d_out_proofs,
                                                    int32_t* status, void* hip_stream) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    static_assert(sizeof(int32_t) == sizeof(int), "status array");
    int st = e->recover_cells_and_kzg_proofs_device((int)n, d_cells, present_masks, d_out_cells, d_out_proofs, (int*)status,
                                                    (hipStream_t)hip_stream);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

// Sharded verification: per-rank partial, then one combine over the gathered records.
CResult eth_kzg_amd_verify_cell_kzg_proof_batch_partial(const DASContext* ctx, uint64_t commitments_length,
                                                        const uint8_t* const* commitments, uint64_t cell_indices_length,
                                                        const uint64_t* cell_indices, uint64_t cells_length,
                                                        const uint8_t* const* cells, uint64_t proofs_length,
                                                        const uint8_t* const* proofs, uint64_t shard_begin,
                                                        uint64_t shard_end, uint8_t* out_partial) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int st = e->verify_cell_kzg_proof_batch_partial_host(commitments_length, commitments, cell_indices_length, cell_indices,
                                                         cells_length, cells, proofs_length, proofs, shard_begin, shard_end,
                                                         out_partial);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_amd_verify_cell_kzg_proof_batch_combine(const DASContext* ctx, uint64_t n_partials, const uint8_t* partials,
                                                        bool* verified) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_cell_kzg_proof_batch_combine_host(n_partials, partials, &ver);
    if (st) return err(status_text(st));
    *verified = ver != 0;
    return ok();
}

CResult eth_kzg_recover_cells_and_proofs(const DASContext* ctx, uint64_t cells_length, const uint8_t* const* cells,
                                         uint64_t cell_indices_length, const uint64_t* cell_indices,
                                         uint8_t** out_cells, uint8_t** out_proofs) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int st = e->recover_cells_and_kzg_proofs_host(cells_length, cells, cell_indices_length, cell_indices, out_cells, out_proofs);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

uint64_t eth_kzg_constant_bytes_per_cell(void) { return 2048; }
uint64_t eth_kzg_constant_bytes_per_proof(void) { return 48; }
uint64_t eth_kzg_constant_cells_per_ext_blob(void) { return CELLS; }

// EIP-4844 single-point operations (bindings/c/src/lib.rs:423-566)
static CResult finish(kzg::Engine* e, int st) {
    if (st == kzg::ERR_DEVICE) return device_err(e);
    return st ? err(status_text(st)) : ok();
}
CResult eth_kzg_compute_kzg_proof(const DASContext* ctx, const uint8_t* blob, const uint8_t* z, uint8_t* out_proof, uint8_t* out_y) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    return finish(e, e->compute_kzg_proof_host(blob, z, out_proof, out_y));
}
CResult eth_kzg_compute_blob_kzg_proof(const DASContext* ctx, const uint8_t* blob, const uint8_t* commitment, uint8_t* out) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    return finish(e, e->compute_blob_kzg_proof_host(blob, commitment, out));
}
CResult eth_kzg_verify_kzg_proof(const DASContext* ctx, const uint8_t* commitment, const uint8_t* z, const uint8_t* y,
                                 const uint8_t* proof, bool* verified) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_kzg_proof_host(commitment, z, y, proof, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}
CResult eth_kzg_verify_blob_kzg_proof(const DASContext* ctx, const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof,
                                      bool* verified) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_blob_kzg_proof_host(blob, commitment, proof, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}
CResult eth_kzg_verify_blob_kzg_proof_batch(const DASContext* ctx, uint64_t blobs_length, const uint8_t* const* blobs,
                                            uint64_t commitments_length, const uint8_t* const* commitments, uint64_t proofs_length,
                                            const uint8_t* const* proofs, bool* verified) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_blob_kzg_proof_batch_host(blobs_length, blobs, commitments_length, commitments, proofs_length, proofs, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}

// ---------------------------------------------------------------------------------------------
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_batch(const DASContext* ctx, uint64_t n, const uint8_t* const* blobs,
                                                       uint8_t* const* const* out_cells, uint8_t* const* const* out_proofs,
                                                       int32_t* status) {
    kzg::Engine* e = eng(ctx);
    std::vector<int> st(n);
    if (e->compute_cells_and_kzg_proofs_host((int)n, blobs, out_cells, out_proofs, st.data())) return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
    return ok();
}
CResult eth_kzg_amd_blob_to_kzg_commitment_batch(const DASContext* ctx, uint64_t n, const uint8_t* const* blobs,
                                                 uint8_t* const* out, int32_t* status) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    std::vector<int> st(n);
    if (e->blob_to_kzg_commitment_host((int)n, blobs, out, st.data())) return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
    return ok();
}
CResult eth_kzg_amd_recover_cells_and_proofs_batch(const DASContext* ctx, uint64_t n, const uint64_t* cells_lengths,
                                                   const uint8_t* const* const* cells, const uint64_t* cell_indices_lengths,
                                                   const uint64_t* const* cell_indices, uint8_t* const* const* out_cells,
                                                   uint8_t* const* const* out_proofs, int32_t* status) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    std::vector<int> st(n);
    if (e->recover_cells_and_kzg_proofs_batch_host((int)n, cells_lengths, cells, cell_indices_lengths, cell_indices, out_cells,
                                                   out_proofs, st.data()))
        return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
    return ok();
}
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_device(const DASContext* ctx, uint64_t n, const uint8_t* d_blobs,
                                                        uint8_t* d_out_cells, uint8_t* d_out_proofs, int32_t* status,
                                                        void* hip_stream) {
    kzg::Engine* e = eng(ctx);
    bool sync = hip_stream == nullptr;
    if (e->compute_cells_and_kzg_proofs_device((int)n, d_blobs, d_out_cells, d_out_proofs, status, (hipStream_t)hip_stream, sync))
        return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = status[i] ? kzg::ERR_SCALAR : 0;
    return ok();
}
CResult eth_kzg_amd_blob_to_kzg_commitment_device(const DASContext* ctx, uint64_t n, const uint8_t* d_blobs, uint8_t* d_out,
                                                  int32_t* status, void* hip_stream) {
    auto lane = eng(ctx)->lease_serial();
    kzg::Engine* e = lane.e;
    bool sync = hip_stream == nullptr;
    if (e->blob_to_kzg_commitment_device((int)n, d_blobs, d_out, status, (hipStream_t)hip_stream, sync)) return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = status[i] ? kzg::ERR_SCALAR : 0;
    return ok();
}
void eth_kzg_amd_set_profiling(const DASContext* ctx, int on) { eng(ctx)->set_profiling(on != 0); }
int eth_kzg_amd_get_stage_times(const DASContext* ctx, double* ms, uint64_t* launches, int n) {
    double m[kzg::Engine::ST_COUNT];
    uint64_t l[kzg::Engine::ST_COUNT];
    eng(ctx)->get_stage_times(m, l);
    for (int i = 0; i < n && i < kzg::Engine::ST_COUNT; i++) { ms[i] = m[i]; launches[i] = l[i]; }
    return kzg::Engine::ST_COUNT;
}
uint64_t eth_kzg_amd_table_bytes(const DASContext* ctx) { return eng(ctx)->table_bytes(); }
int eth_kzg_amd_window_bits(const DASContext* ctx) { return eng(ctx)->window_bits(); }
int eth_kzg_amd_tables_ready(const DASContext* ctx, int wait_ms) { return eng(ctx)->tables_ready(wait_ms); }
void eth_kzg_amd_table_build_info(const DASContext* ctx, double* out4) { eng(ctx)->table_build_info(out4); }
int eth_kzg_amd_table_groups_ready(const DASContext* ctx) { return eng(ctx)->table_groups_ready(kzg::Engine::TAB_FK); }
void eth_kzg_amd_linmap_info(const DASContext* ctx, int32_t* out4) {
    for (int i = 0; i < 4; i++) out4[i] = eng(ctx)->linmap_info()[i];
}

int eth_kzg_amd_test_fr_ntt4096(const DASContext* ctx, const uint8_t* in, uint8_t* out, int inverse_dit) {
    return eng(ctx)->test_fr_ntt4096(in, out, inverse_dit);
}
int eth_kzg_amd_test_g1_fft128(const DASContext* ctx, const uint8_t* in, uint8_t* out, int n_lanes, int inverse) {
    return eng(ctx)->test_g1_fft128(in, out, n_lanes, inverse);
}
int eth_kzg_amd_test_fixed_msm(const DASContext* ctx, const uint8_t* scalars, int n_msm, uint8_t* out) {
    return eng(ctx)->test_fixed_msm(scalars, n_msm, out);
}
int eth_kzg_amd_test_g1_decompress(const DASContext* ctx, const uint8_t* in, int n, int subgroup_check, int32_t* status,
                                   uint8_t* out) {
    return eng(ctx)->test_g1_decompress(in, n, subgroup_check, status, out);
}
int eth_kzg_amd_test_field_mul(const DASContext* ctx, const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp) {
    return eng(ctx)->test_field_mul(a, b, out, n, is_fp);
}

}  // extern "C"
