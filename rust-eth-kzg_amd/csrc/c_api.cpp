// C ABI of the MI355X KZG engine: the reference's `eth_kzg_*` symbols (bindings/c/src/lib.rs) over
// kzg::Engine, plus the batched / device-resident `eth_kzg_amd_*` additions.  See include/c_eth_kzg.h.
//
// A context may span a DEVICE LIST (c_ctx.hpp): every entry point below first decides WHERE a call runs --
//   single-problem calls (the reference's sixteen symbols)   the least-loaded device (DevicePicker)
//   host-pointer batches (_batch, _many)                     contiguous slices, one per device (fan_out_slices)
//   device-resident forms (_device)                          the device that owns the caller's buffers
// -- and a one-device context takes none of these paths.
#include "../../include/c_eth_kzg.h"
#include "c_ctx.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int CELLS = 128;
// Batched entry points bound their count: the engine indexes blobs with `int`, and a count in the billions is a corrupted argument,
// not a workload (2^24 blobs are 2 TB of input).  Malformed input is `Err`, never a crash (bindings/c/src/lib.rs:272-280).
constexpr uint64_t MAX_BATCH = 1u << 24;

CResult ok() { return CResult{Ok, nullptr}; }
CResult err(const std::string& m) {  // CResult::with_error, bindings/c/src/lib.rs:146-153
    char* s = (char*)malloc(m.size() + 1);
    if (!s) return CResult{Err, nullptr};
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
const DASContext* live(const DASContext* ctx) {
    // `assert!(!ctx.is_null())` in the reference (e.g. compute_cells_and_kzg_proofs.rs:14): abort, like a Rust panic across FFI
    if (!ctx || !ctx->engine) {
        fprintf(stderr, "c_eth_kzg: context pointer is null\n");
        abort();
    }
    return ctx;
}
kzg::Engine* eng(const DASContext* ctx) { return live(ctx)->engine; }
std::string device_text(kzg::Engine* e) { return "DeviceError(" + e->last_error() + ")"; }
CResult device_err(kzg::Engine* e) { return err(device_text(e)); }

// where a single-problem call runs: the context's one engine, or the least-loaded device of its list.  `weight` = the call's size
// in units that compare across entry points (cells: a blob is 128 of them).
struct Picked {
    kzg::Engine* e;
    kzg::DevicePicker::Ticket ticket;
};
Picked pick(const DASContext* ctx, uint64_t weight) {
    live(ctx);
    auto t = ctx->picker->pick(weight ? weight : 1);
    return Picked{ctx->engines[(size_t)t.device()], std::move(t)};
}
// the engine whose GPU holds a device pointer of the caller (device-resident forms); NULL: none of the context's devices does
kzg::Engine* engine_of_pointer(const DASContext* ctx, const void* p) {
    live(ctx);
    if (ctx->engines.size() == 1 || !p) return ctx->engine;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    for (kzg::Engine* e : ctx->engines)
        if (e->device() == a.device) return e;
    return nullptr;
}
CResult slices_result(const DASContext* ctx, const std::pair<int, std::string>& r) {
    if (r.first < 0) return ok();
    if (ctx->engines.size() == 1) return err(r.second);
    return err("device " + std::to_string(ctx->engines[(size_t)r.first]->device()) + ": " + r.second);
}

void free_ctx(DASContext* c) {
    if (!c) return;
    for (kzg::Engine* e : c->engines) delete e;
    delete c;
}
// the one constructor: NULL + message when the context cannot be built (no usable GPU, not even the 3.7 GB start tables fit).
// The engines of a device list are built side by side (a constructor is 0.1 - 2 s of set-up kernels and allocations per GPU).
DASContext* try_make_ctx(bool use_precomp, const std::vector<int>& devices, double table_budget_gb, std::string* why) {
    DASContext* c = nullptr;
    try {
        if (devices.empty() || devices.size() > 64) throw std::runtime_error("empty or oversized device list");
        c = new DASContext;
        const size_t D = devices.size();
        c->engines.assign(D, nullptr);
        std::vector<std::string> errs(D);
        auto make = [&](size_t d) {
            try {
                c->engines[d] = new kzg::Engine(use_precomp, devices[d], nullptr, table_budget_gb);
            } catch (const std::exception& e) {
                errs[d] = e.what();
                if (errs[d].empty()) errs[d] = "unknown failure";
            } catch (...) {  // (nothing may leave a helper thread)
                errs[d] = "unknown failure";
            }
        };
        // Engines on DIFFERENT GPUs are built side by side (they share nothing but the mutex-guarded registries); entries that repeat
        // an ordinal -- the form the one-GPU tests use -- are built one after the other: the later one finds the tables the earlier
        // one has published.  (Same-GPU constructors side by side work as well -- the table registry is made for concurrent
        // constructors, tools/probe_concurrent_ctors.py -- but a list 0,0 built that way kept 1.4 GB of HBM after injected constructor
        // faults on two of the test boxes, which is not understood; the sequential order is the one the soaks have run on.)
        std::vector<std::thread> th;
        std::vector<char> deferred(D, 0);
        for (size_t d = 1; d < D; d++) {
            for (size_t k = 0; k < d; k++) deferred[d] |= devices[k] == devices[d];
            if (deferred[d]) continue;
            try {
                th.emplace_back(make, d);
            } catch (const std::exception&) {  // no thread to be had: built on this one, below
                deferred[d] = 1;
            }
        }
        make(0);
        for (auto& t : th) t.join();
        for (size_t d = 1; d < D; d++)
            if (deferred[d]) make(d);
        for (size_t d = 0; d < D; d++)
            if (!c->engines[d]) throw std::runtime_error(D == 1 ? errs[d] : "device " + std::to_string(devices[d]) + ": " + errs[d]);
        c->engine = c->engines[0];
        c->picker.reset(new kzg::DevicePicker((int)D));
        return c;
    } catch (const std::exception& e) {
        free_ctx(c);
        if (why) *why = e.what();
        return nullptr;
    }
}
DASContext* make_ctx(bool use_precomp, const std::vector<int>& devices) {
    std::string why;
    DASContext* c = try_make_ctx(use_precomp, devices, 0, &why);
    if (!c) {
        // The reference panics when the context cannot be built (bad SRS); here: no usable GPU.  No CPU fallback.  A host that must
        // not die calls eth_kzg_amd_das_context_try_new instead.
        fprintf(stderr, "c_eth_kzg: cannot create the MI355X context: %s\n", why.c_str());
        abort();
    }
    return c;
}
std::vector<int> env_device_list() {  // ETH_KZG_AMD_DEVICES (a list, or "all"), else ETH_KZG_AMD_DEVICE, else GPU 0 -- read once per context
    const kzg::Knobs k = kzg::Knobs::from_env();
    if (k.devices.size() == 1 && k.devices[0] < 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        std::vector<int> all;
        for (int d = 0; d < n; d++) all.push_back(d);
        if (all.empty()) all.push_back(0);  // (the constructor reports the missing GPU)
        return all;
    }
    if (!k.devices.empty()) return k.devices;
    return {k.device};
}

}  // namespace

extern "C" {

DASContext* eth_kzg_das_context_new(bool use_precomp) { return make_ctx(use_precomp, env_device_list()); }
DASContext* eth_kzg_amd_das_context_new_on_device(bool use_precomp, int device_ordinal) {
    return make_ctx(use_precomp, {device_ordinal});
}
DASContext* eth_kzg_amd_das_context_try_new(bool use_precomp, int device_ordinal, double table_budget_gb, CResult* result) {
    std::string why;
    DASContext* c = try_make_ctx(use_precomp, {device_ordinal}, table_budget_gb, &why);
    if (result) *result = c ? ok() : err("ContextCreation(" + why + ")");
    return c;
}
DASContext* eth_kzg_amd_das_context_new_on_devices(bool use_precomp, const int32_t* device_ordinals, uint64_t n_devices,
                                                   double table_budget_gb, CResult* result) {
    std::string why;
    DASContext* c = nullptr;
    if (!device_ordinals || n_devices == 0 || n_devices > 64) why = "InvalidInput: the device list holds 1 to 64 ordinals";
    else c = try_make_ctx(use_precomp, std::vector<int>(device_ordinals, device_ordinals + n_devices), table_budget_gb, &why);
    if (result) *result = c ? ok() : err("ContextCreation(" + why + ")");
    return c;
}
uint64_t eth_kzg_amd_context_devices(const DASContext* ctx, int32_t* out_ordinals, uint64_t capacity) {
    live(ctx);
    for (size_t d = 0; d < ctx->engines.size() && d < capacity && out_ordinals; d++) out_ordinals[d] = ctx->engines[d]->device();
    return ctx->engines.size();
}
void eth_kzg_das_context_free(DASContext* ctx) {
    if (!ctx) return;
    eth_kzg_amd_comm_destroy(ctx);
    free_ctx(ctx);
}
void eth_kzg_free_error_message(char* c_message) {
    if (c_message) free(c_message);
}

CResult eth_kzg_blob_to_kzg_commitment(const DASContext* ctx, const uint8_t* blob, uint8_t* out) {
    Picked p = pick(ctx, CELLS);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    int st = 0;
    const uint8_t* blobs[1] = {blob};
    uint8_t* outs[1] = {out};
    if (e->blob_to_kzg_commitment_host(1, blobs, outs, &st)) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_compute_cells_and_kzg_proofs(const DASContext* ctx, const uint8_t* blob, uint8_t** out_cells,
                                             uint8_t** out_proofs) {
    Picked p = pick(ctx, CELLS);
    kzg::Engine* e = p.e;
    int st = 0;
    const uint8_t* blobs[1] = {blob};
    uint8_t* const* cells[1] = {out_cells};
    uint8_t* const* proofs[1] = {out_proofs};
    if (e->compute_cells_and_kzg_proofs_host(1, blobs, cells, proofs, &st)) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_compute_cells(const DASContext* ctx, const uint8_t* blob, uint8_t** out_cells) {
    Picked p = pick(ctx, CELLS / 16);
    kzg::Engine* e = p.e;
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
    Picked p = pick(ctx, cells_length / 8 + 1);
    kzg::Engine* e = p.e;
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
    if (n_batches > MAX_BATCH) return err("InvalidInput");  // (before the context is looked at: a corrupted count is rejected whatever else is wrong)
    live(ctx);  // own lock, stream and scratch inside (verify_many.hip): no lane needed
    if (n_batches == 0) return ok();
    try {
        std::vector<int> ver(n_batches), st(n_batches);
        // by PROBLEM over the device list: problems are independent (each has its own Fiat-Shamir transcript)
        const auto r = kzg::fan_out_slices((int)ctx->engines.size(), n_batches, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
            kzg::Engine* e = ctx->engines[(size_t)d];
            const int rc = e->verify_cell_kzg_proof_batch_many_host(hi - lo, commitments_lengths + lo, commitments + lo, cell_indices_lengths + lo,
                                                                    cell_indices + lo, cells_lengths + lo, cells + lo, proofs_lengths + lo, proofs + lo,
                                                                    ver.data() + lo, st.data() + lo);
            return rc == kzg::ERR_DEVICE ? device_text(e) : std::string();
        });
        if (r.first >= 0) return slices_result(ctx, r);
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
    kzg::Engine* owner = engine_of_pointer(ctx, d_cells);
    if (!owner) return err("InvalidInput: the buffers are on no device of this context");
    auto lane = owner->lease_serial();
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
    if (n > MAX_BATCH) return err("InvalidInput");
    kzg::Engine* owner = engine_of_pointer(ctx, d_cells);
    if (!owner) return err("InvalidInput: the buffers are on no device of this context");
    auto lane = owner->lease_serial();
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
    Picked p = pick(ctx, (shard_end > shard_begin ? shard_end - shard_begin : 0) / 8 + 1);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    int st = e->verify_cell_kzg_proof_batch_partial_host(commitments_length, commitments, cell_indices_length, cell_indices,
                                                         cells_length, cells, proofs_length, proofs, shard_begin, shard_end,
                                                         out_partial);
    if (st == kzg::ERR_DEVICE) return device_err(e);
    return st ? err(status_text(st)) : ok();
}

CResult eth_kzg_amd_verify_cell_kzg_proof_batch_combine(const DASContext* ctx, uint64_t n_partials, const uint8_t* partials,
                                                        bool* verified) {
    if (n_partials > MAX_BATCH) return err("InvalidInput");
    Picked p = pick(ctx, 1);
    auto lane = p.e->lease_serial();
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
    Picked p = pick(ctx, 2 * CELLS);
    auto lane = p.e->lease_serial();
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
    Picked p = pick(ctx, CELLS);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    return finish(e, e->compute_kzg_proof_host(blob, z, out_proof, out_y));
}
CResult eth_kzg_compute_blob_kzg_proof(const DASContext* ctx, const uint8_t* blob, const uint8_t* commitment, uint8_t* out) {
    Picked p = pick(ctx, CELLS);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    return finish(e, e->compute_blob_kzg_proof_host(blob, commitment, out));
}
CResult eth_kzg_verify_kzg_proof(const DASContext* ctx, const uint8_t* commitment, const uint8_t* z, const uint8_t* y,
                                 const uint8_t* proof, bool* verified) {
    Picked p = pick(ctx, 1);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_kzg_proof_host(commitment, z, y, proof, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}
CResult eth_kzg_verify_blob_kzg_proof(const DASContext* ctx, const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof,
                                      bool* verified) {
    Picked p = pick(ctx, CELLS);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_blob_kzg_proof_host(blob, commitment, proof, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}
CResult eth_kzg_verify_blob_kzg_proof_batch(const DASContext* ctx, uint64_t blobs_length, const uint8_t* const* blobs,
                                            uint64_t commitments_length, const uint8_t* const* commitments, uint64_t proofs_length,
                                            const uint8_t* const* proofs, bool* verified) {
    // ONE random linear combination over the whole batch (crates/eip4844/src/verifier.rs): not cut over devices
    Picked p = pick(ctx, blobs_length < MAX_BATCH ? blobs_length * CELLS + 1 : 1);
    auto lane = p.e->lease_serial();
    kzg::Engine* e = lane.e;
    int ver = 0;
    int st = e->verify_blob_kzg_proof_batch_host(blobs_length, blobs, commitments_length, commitments, proofs_length, proofs, &ver);
    if (!st) *verified = ver != 0;
    return finish(e, st);
}

// ---------------------------------------------------------------------------------------------
// Host-pointer batches: contiguous slices over the context's device list (the caller's buffers are the gather target; no
// collective), the whole batch on the one engine of a one-device context.
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_batch(const DASContext* ctx, uint64_t n, const uint8_t* const* blobs,
                                                       uint8_t* const* const* out_cells, uint8_t* const* const* out_proofs,
                                                       int32_t* status) {
    if (n > MAX_BATCH) return err("InvalidInput");
    live(ctx);
    if (n == 0) return ok();
    if (!blobs) return err("InvalidInput");
    try {
        std::vector<int> st(n);
        const auto r = kzg::fan_out_slices((int)ctx->engines.size(), n, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
            kzg::Engine* e = ctx->engines[(size_t)d];
            return e->compute_cells_and_kzg_proofs_host((int)(hi - lo), blobs + lo, out_cells ? out_cells + lo : nullptr,
                                                        out_proofs ? out_proofs + lo : nullptr, st.data() + lo)
                       ? device_text(e) : std::string();
        });
        if (r.first >= 0) return slices_result(ctx, r);
        if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
        return ok();
    } catch (const std::exception& ex) {
        return err(std::string("DeviceError(") + ex.what() + ")");
    }
}
CResult eth_kzg_amd_blob_to_kzg_commitment_batch(const DASContext* ctx, uint64_t n, const uint8_t* const* blobs,
                                                 uint8_t* const* out, int32_t* status) {
    if (n > MAX_BATCH) return err("InvalidInput");
    live(ctx);
    if (n == 0) return ok();
    if (!blobs || !out) return err("InvalidInput");
    try {
        std::vector<int> st(n);
        const auto r = kzg::fan_out_slices((int)ctx->engines.size(), n, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
            auto lane = ctx->engines[(size_t)d]->lease_serial();
            kzg::Engine* e = lane.e;
            return e->blob_to_kzg_commitment_host((int)(hi - lo), blobs + lo, out + lo, st.data() + lo) ? device_text(e) : std::string();
        });
        if (r.first >= 0) return slices_result(ctx, r);
        if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
        return ok();
    } catch (const std::exception& ex) {
        return err(std::string("DeviceError(") + ex.what() + ")");
    }
}
CResult eth_kzg_amd_recover_cells_and_proofs_batch(const DASContext* ctx, uint64_t n, const uint64_t* cells_lengths,
                                                   const uint8_t* const* const* cells, const uint64_t* cell_indices_lengths,
                                                   const uint64_t* const* cell_indices, uint8_t* const* const* out_cells,
                                                   uint8_t* const* const* out_proofs, int32_t* status) {
    if (n > MAX_BATCH) return err("InvalidInput");
    live(ctx);
    if (n == 0) return ok();
    if (!cells_lengths || !cells || !cell_indices_lengths || !cell_indices) return err("InvalidInput");
    try {
        std::vector<int> st(n);
        const auto r = kzg::fan_out_slices((int)ctx->engines.size(), n, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
            auto lane = ctx->engines[(size_t)d]->lease_serial();
            kzg::Engine* e = lane.e;
            return e->recover_cells_and_kzg_proofs_batch_host((int)(hi - lo), cells_lengths + lo, cells + lo, cell_indices_lengths + lo,
                                                              cell_indices + lo, out_cells ? out_cells + lo : nullptr,
                                                              out_proofs ? out_proofs + lo : nullptr, st.data() + lo)
                       ? device_text(e) : std::string();
        });
        if (r.first >= 0) return slices_result(ctx, r);
        if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
        return ok();
    } catch (const std::exception& ex) {
        return err(std::string("DeviceError(") + ex.what() + ")");
    }
}
CResult eth_kzg_amd_compute_cells_and_kzg_proofs_device(const DASContext* ctx, uint64_t n, const uint8_t* d_blobs,
                                                        uint8_t* d_out_cells, uint8_t* d_out_proofs, int32_t* status,
                                                        void* hip_stream) {
    if (n > MAX_BATCH) return err("InvalidInput");
    kzg::Engine* e = engine_of_pointer(ctx, d_blobs);
    if (!e) return err("InvalidInput: the buffers are on no device of this context");
    bool sync = hip_stream == nullptr;
    if (e->compute_cells_and_kzg_proofs_device((int)n, d_blobs, d_out_cells, d_out_proofs, status, (hipStream_t)hip_stream, sync))
        return device_err(e);
    if (status) for (uint64_t i = 0; i < n; i++) status[i] = status[i] ? kzg::ERR_SCALAR : 0;
    return ok();
}
CResult eth_kzg_amd_blob_to_kzg_commitment_device(const DASContext* ctx, uint64_t n, const uint8_t* d_blobs, uint8_t* d_out,
                                                  int32_t* status, void* hip_stream) {
    if (n > MAX_BATCH) return err("InvalidInput");
    kzg::Engine* owner = engine_of_pointer(ctx, d_blobs);
    if (!owner) return err("InvalidInput: the buffers are on no device of this context");
    auto lane = owner->lease_serial();
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
// table queries: per device they describe the context's FIRST device; the byte count is the whole list's
uint64_t eth_kzg_amd_table_bytes(const DASContext* ctx) {
    live(ctx);
    uint64_t b = 0;
    std::vector<int> seen;
    for (kzg::Engine* e : ctx->engines) {  // two engines on one GPU share its tables: counted once
        bool dup = false;
        for (int d : seen) dup |= d == e->device();
        if (!dup) { b += e->table_bytes(); seen.push_back(e->device()); }
    }
    return b;
}
int eth_kzg_amd_window_bits(const DASContext* ctx) { return eng(ctx)->window_bits(); }
int eth_kzg_amd_window_count(const DASContext* ctx) { return eng(ctx)->window_count(); }
int eth_kzg_amd_glv_table(const DASContext* ctx) { (void)live(ctx); return 1; }  // deprecated: every table has been a GLV table since round 5
int eth_kzg_amd_abi_version(void) { return 6; }
int eth_kzg_amd_tables_ready(const DASContext* ctx, int wait_ms) {
    live(ctx);
    // the slowest device decides: 1 only when every device is on its final tables, 2 when a build failed somewhere, else 0
    int all = 1;
    for (kzg::Engine* e : ctx->engines) {
        const int s = e->tables_ready(wait_ms);
        if (s == 2) all = 2;
        else if (s == 0 && all != 2) all = 0;
    }
    return all;
}
void eth_kzg_amd_table_build_info(const DASContext* ctx, double* out4) { eng(ctx)->table_build_info(out4); }
int eth_kzg_amd_table_groups_ready(const DASContext* ctx) { return eng(ctx)->table_groups_ready(kzg::Engine::TAB_FK); }
void eth_kzg_amd_linmap_info(const DASContext* ctx, int32_t* out4) {
    for (int i = 0; i < 4; i++) out4[i] = eng(ctx)->linmap_info()[i];
}

}  // extern "C"
