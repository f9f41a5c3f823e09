// Multi-GPU entry points of the C ABI (include/c_eth_kzg.h, "multi-GPU"): blob batches shard across the GPUs of a node
// with no data-path collective; the only exchange north_star names is one all-gather of the proof vectors, done here by
// RCCL (ncclAllGather over xGMI) on a communicator the library owns.  RCCL is dlopen'ed on first use, so loading
// libc_eth_kzg.so never depends on it.  The single-process form fans a host-pointer batch out over one context per GPU
// with one host thread each; the caller's buffers are the gather target.
// Reference analogue: the per-blob independence that maybe_rayon's par_iter exploits (crates/eip7594/src/prover.rs,
// crates/maybe_rayon), stretched over devices.
#include "../../include/c_eth_kzg.h"
#include "c_ctx.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

CResult ok() { return CResult{Ok, nullptr}; }
CResult err(const std::string& m) {
    char* s = (char*)malloc(m.size() + 1);
    memcpy(s, m.c_str(), m.size() + 1);
    return CResult{Err, s};
}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why, path;
    bool shared_with_process = false;  // the handle is the copy some other component (torch) had already mapped
};
// An RCCL that the process has already mapped wins (torch ships its own copy under torch/lib and maps it with the
// Python extension; a second copy in the same address space would keep separate bootstrap / topology state and the two
// are not guaranteed to be the same build): dl_iterate_phdr finds it by name and RTLD_NOLOAD hands back a handle to that
// very object.  Only when none is mapped is librccl.so.1 loaded from the usual places.
int find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out) {
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) != 0) return 0;
    *(std::string*)out = name;
    return 1;
}
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string mapped;
        dl_iterate_phdr(find_mapped_rccl, &mapped);
        if (!mapped.empty()) {
            r.handle = dlopen(mapped.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) { r.path = mapped; r.shared_with_process = true; }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) r.path = name;
        }
        if (!r.handle) { r.why = "RCCL not found (librccl.so)"; return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.handle, "ncclCommUserRank");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { r.why = "RCCL symbols missing in " + r.path; r.handle = nullptr; }
    });
    return r;
}
std::string nccl_text(ncclResult_t e) {
    Rccl& r = rccl();
    return std::string("RCCL: ") + (r.GetErrorString ? r.GetErrorString(e) : "error") + " (" + std::to_string((int)e) + ")";
}

}  // namespace

namespace kzg {
// the communicator a context owns (Engine keeps it as an opaque pointer so that engine.hpp stays free of RCCL types)
struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};
}  // namespace kzg

extern "C" {

static_assert(sizeof(ncclUniqueId) == 128, "the ABI passes the id as 128 bytes");

CResult eth_kzg_amd_comm_unique_id(uint8_t* out_id) {
    Rccl& r = rccl();
    if (!r.handle) return err(r.why);
    ncclUniqueId id;
    ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return err(nccl_text(e));
    memcpy(out_id, &id, 128);
    return ok();
}

CResult eth_kzg_amd_comm_probe(const DASContext* ctx, char* out_library_path, uint64_t path_capacity) {
    if (!ctx || !ctx->engine) abort();
    Rccl& r = rccl();
    if (!r.handle) return err(r.why);
    if (ctx->engine->comm()) return err("communicator already attached to this context");
    if (out_library_path && path_capacity) {
        std::string p = r.path + (r.shared_with_process ? " (already mapped by the process)" : " (loaded by libc_eth_kzg)");
        size_t n = p.size() < path_capacity - 1 ? p.size() : (size_t)path_capacity - 1;
        memcpy(out_library_path, p.data(), n);
        out_library_path[n] = 0;
    }
    return ok();
}

CResult eth_kzg_amd_comm_info(const DASContext* ctx, int* out_rank, int* out_world) {
    if (!ctx || !ctx->engine) abort();
    Rccl& r = rccl();
    kzg::Comm* c = ctx->engine->comm();
    if (!r.handle || !c) return err("no communicator: call eth_kzg_amd_comm_init first");
    int rank = c->rank, world = c->world;
    // what the communicator itself reports (not what the caller passed in), where RCCL exports the queries
    if (r.CommCount && r.CommCount(c->comm, &world) != ncclSuccess) return err("RCCL: ncclCommCount failed");
    if (r.CommUserRank && r.CommUserRank(c->comm, &rank) != ncclSuccess) return err("RCCL: ncclCommUserRank failed");
    if (out_rank) *out_rank = rank;
    if (out_world) *out_world = world;
    return ok();
}

CResult eth_kzg_amd_comm_init(DASContext* ctx, const uint8_t* id_bytes, int rank, int world) {
    if (!ctx || !ctx->engine) abort();
    Rccl& r = rccl();
    if (!r.handle) return err(r.why);
    if (world < 1 || rank < 0 || rank >= world) return err("InvalidInput");
    kzg::Engine* e = ctx->engine;
    if (e->comm()) return err("communicator already attached to this context");
    if (hipSetDevice(e->device()) != hipSuccess) return err("DeviceError(hipSetDevice)");
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    auto* c = new kzg::Comm;
    c->rank = rank;
    c->world = world;
    ncclResult_t rc = r.CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { delete c; return err(nccl_text(rc)); }
    e->set_comm(c);
    return ok();
}

CResult eth_kzg_amd_all_gather(const DASContext* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream) {
    if (!ctx || !ctx->engine) abort();
    Rccl& r = rccl();
    kzg::Comm* c = ctx->engine->comm();
    if (!r.handle || !c) return err("no communicator: call eth_kzg_amd_comm_init first");
    if (hipSetDevice(ctx->engine->device()) != hipSuccess) return err("DeviceError(hipSetDevice)");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->engine->stream();
    ncclResult_t rc = r.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclUint8, c->comm, st);
    if (rc != ncclSuccess) return err(nccl_text(rc));
    if (!hip_stream && hipStreamSynchronize(st) != hipSuccess) return err("DeviceError(hipStreamSynchronize)");
    return ok();
}

void eth_kzg_amd_comm_destroy(DASContext* ctx) {
    if (!ctx || !ctx->engine) return;
    kzg::Comm* c = ctx->engine->comm();
    if (!c) return;
    (void)hipSetDevice(ctx->engine->device());
    if (rccl().handle) (void)rccl().CommDestroy(c->comm);
    delete c;
    ctx->engine->set_comm(nullptr);
}

int eth_kzg_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

CResult eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi(const DASContext* const* contexts, uint64_t n_contexts, uint64_t n,
                                                             const uint8_t* const* blobs, uint8_t* const* const* out_cells,
                                                             uint8_t* const* const* out_proofs, int32_t* status) {
    if (n_contexts == 0 || n_contexts > 64 || !contexts || n > (1u << 24)) return err("InvalidInput");
    for (uint64_t d = 0; d < n_contexts; d++)
        if (!contexts[d] || !contexts[d]->engine) abort();
    if (n == 0) return ok();
    if (!blobs) return err("InvalidInput");
    try {
        // contiguous slices, sizes differing by at most one blob (rust-eth-kzg_amd/sharding.py: shard_bounds; host_sync.hpp: the
        // protocol a context over a device list uses for every batched entry point)
        std::vector<int> st(n);
        const auto r = kzg::fan_out_slices((int)n_contexts, n, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
            kzg::Engine* e = contexts[d]->engine;
            return e->compute_cells_and_kzg_proofs_host((int)(hi - lo), blobs + lo, out_cells ? out_cells + lo : nullptr,
                                                        out_proofs ? out_proofs + lo : nullptr, st.data() + lo)
                       ? (e->last_error().empty() ? std::string("unknown failure") : e->last_error()) : std::string();
        });
        if (r.first >= 0) return err("DeviceError(device " + std::to_string(contexts[r.first]->engine->device()) + ": " + r.second + ")");
        if (status) for (uint64_t i = 0; i < n; i++) status[i] = st[i];
        return ok();
    } catch (const std::exception& ex) {
        return err(std::string("DeviceError(") + ex.what() + ")");
    }
}

}  // extern "C"
