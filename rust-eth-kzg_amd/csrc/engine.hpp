// Host-side mirror of the reference's DASContext for the GPU hot path
// (reference: crates/eip7594/src/lib.rs:41-87 DASContext, prover.rs:62-171 ProverContext,
//  verifier.rs:72-112).  One Engine = one context on one GPU.
#pragma once
#include "knobs.hpp"
#include "host_sync.hpp"
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>
#include <memory>

namespace kzg {

namespace pairing { struct G2Prepared; }
struct Comm;  // multi_gpu.cpp
struct G1Affine;  // curve.hpp
struct Fr8 { uint32_t v[8]; };
struct Fp12w { uint32_t v[12]; };

// A verification problem's cell list is bounded: the powers of the Fiat-Shamir challenge come from a 24-entry table r^(2^i)
// (k_verify.hip: k_verify_scalars, k_verify_many.hip: k_vm_scalars) and positions are 32-bit on the device.  2^24 cells are 34 GB
// of input; longer lists are rejected as invalid input by every verification entry point.
constexpr uint64_t MAX_CELLS_PER_VERIFICATION = (1u << 24) - 1;

enum Status : int {
    OK = 0,
    ERR_SCALAR = 1,    // a field element >= r   (SerializationError::CouldNotDeserializeScalar)
    ERR_G1 = 2,        // bad G1 encoding / not on curve / not in subgroup
    ERR_INPUT = 3,     // length / index validation failed
    ERR_RECOVERY = 4,  // recovered polynomial has non-zero high coefficients
    ERR_DEVICE = 5,    // HIP failure
};

// Scoped scratch buffers of the host-pointer entry points come from per-context pools instead of hipMalloc / hipFree
// (both cost 0.1 - 1 ms and hipFree drains the device): blocks are kept by power-of-two size class and reused; the pool
// is emptied with the context.  Callers hold Engine::mu_, so no locking here.
class BufferPool {
public:
    explicit BufferPool(bool pinned_host) : host_(pinned_host) {}
    ~BufferPool();
    void* get(size_t bytes, size_t* cls);
    void put(void* p, size_t cls);
private:
    bool host_;
    std::vector<std::pair<size_t, void*>> free_;
    size_t pooled_ = 0;
};
class Engine;
struct PoolBuf {  // RAII: a block of at least `bytes` from the context's device (or pinned-host) pool
    void* p = nullptr;
    PoolBuf(Engine& e, size_t bytes, bool pinned_host = false);
    ~PoolBuf();
    PoolBuf(const PoolBuf&) = delete;
    PoolBuf& operator=(const PoolBuf&) = delete;
private:
    Engine& e_;
    size_t cls_ = 0;
    bool host_;
};

// One set of device scratch for a prover call: coefficients, MSM scalars, G1 arrays, status words, the staging buffers of
// the host-pointer entry point and the streams / events that order its use.  The engine owns a few of them, so that
// independent calls (and the chunks of one large host-pointer call) overlap on the GPU instead of queueing on one lock.
struct Work {
    std::mutex mu;  // held while a call enqueues on / owns this set
    int cap = 0;    // blobs the arrays below hold
    void *coeffs = nullptr, *canon = nullptr, *scalars = nullptr, *X = nullptr;
    int* status = nullptr;
    void *circ_table = nullptr, *slp_arena = nullptr;
    size_t slp_arena_bytes = 0;
    hipEvent_t done = nullptr;  // recorded after the last kernel that touches the set; the next user's stream waits on it
    // host-pointer path: device and pinned-host staging for one chunk, a compute stream, a copy stream, events
    int stage_cap = 0;
    uint8_t *d_in = nullptr, *d_cells = nullptr, *d_proofs = nullptr;
    uint8_t *h_in = nullptr, *h_cells = nullptr, *h_proofs = nullptr;
    int* h_status = nullptr;
    hipStream_t stream = nullptr, copy = nullptr;
    hipEvent_t ev_in = nullptr, ev_cells = nullptr, ev_cells_host = nullptr, ev_done = nullptr;
    hipEvent_t ev_coeffs = nullptr, ev_side = nullptr;  // small batches: coefficients ready / cells written on the second stream
    std::vector<hipEvent_t> sub_events;  // per sub-batch of a host-pointer call: [2i] cells computed, [2i+1] cells on the host
};
// (HostPool, the combiner, pass-slot and lane leases: host_sync.hpp -- HIP-free, driven under ThreadSanitizer on the CPU)
class Engine {
public:
    friend struct PoolBuf;
    struct SharedTable;  // engine.hip
    // use_precomp: true -> the widest GLV window tables inside the budget (engine_tables.hip: build_final_tables; the reference's
    //              UsePrecomp::Yes uses width 8 on the CPU), false -> the sixteen-window tables only (2.4 GB, twice the additions of
    //              eight windows).  Results identical.  Tables are shared by the contexts of a device.
    // primary: the context's engine when this one is an auxiliary lane of it (lease_serial): the lane shares the primary's
    // window tables by reading through to its view -- no registry look-up, no builder thread, no lock shared with a build
    // table_budget_gb: > 0: upper bound for both window tables together; 0: $ETH_KZG_AMD_TABLE_GB, else DEFAULT_TABLE_BUDGET_GB;
    // < 0: whatever the HBM still holds
    Engine(bool use_precomp, int device, const Engine* primary = nullptr, double table_budget_gb = 0);
    // The default memory budget of the window tables: the nine-window GLV tables (nominal width 15: two windows of 15 bits and seven
    // of 14, 18 gathered additions per base) for FK20 (70.9 GB) and for the commitments (35.4 GB).  Measured on one box, same
    // resident 2048-blob batch: -8 % against the widest FK20 table (eight windows of 16 bits: 206 GB) at 43 % of the memory; the next
    // step down (ten windows, 29 GB) costs another 6 %.  Taking "whatever HBM holds" is a decision for the host application
    // (ETH_KZG_AMD_TABLE_GB=max, or the budget argument of eth_kzg_amd_das_context_try_new).
    static constexpr double DEFAULT_TABLE_BUDGET_GB = 108.0;
    ~Engine();
    Engine(const Engine&) = delete;

    // The verification / recovery / commitment / EIP-4844 entry points serialise on one lock per Engine (they share its stream
    // and scratch).  So that calls from different host threads overlap, as they do on the reference's immutable context
    // (bindings/node/src/lib.rs:92-299), the context keeps up to ETH_KZG_AMD_SERIAL_LANES (default 4) engines: the primary one
    // and auxiliaries created on demand (0.1 s each; the window tables are shared, so no memory to speak of).  lease_serial()
    // hands out a free one -- the primary when it is idle -- and blocks only when all are busy.
    using SerialLease = LanePool<Engine>::Lease;  // {Engine* e; std::unique_lock<std::mutex> busy;}
    friend class LanePool<Engine>;

    SerialLease lease_serial();

    int device() const { return dev_; }
    Comm* comm() const { return comm_; }       // RCCL communicator attached by eth_kzg_amd_comm_init (multi_gpu.cpp)
    void set_comm(Comm* c) { comm_ = c; }
    hipStream_t stream() const { return stream_; }
    const std::string& last_error() const;  // of the calling thread's last failed call

    // ---- device-resident batch entry points (flat buffers in HBM, launched on `stream`) ----
    // d_blobs: n * 131072 B.  d_cells: n * 128 * 2048 B.  d_proofs: n * 128 * 48 B.  d_commitments: n * 48 B.
    // h_status: n ints (host), 0 = ok.  Any output pointer may be null to skip that product.
    int compute_cells_and_kzg_proofs_device(int n, const uint8_t* d_blobs, uint8_t* d_cells, uint8_t* d_proofs,
                                            int* h_status, hipStream_t stream, bool sync);
    int blob_to_kzg_commitment_device(int n, const uint8_t* d_blobs, uint8_t* d_commitments, int* h_status,
                                      hipStream_t stream, bool sync);

    // ---- host-buffer entry points (what the reference's C ABI hands over) ----
    int compute_cells_and_kzg_proofs_host(int n, const uint8_t* const* blobs, uint8_t* const* const* cells,
                                          uint8_t* const* const* proofs, int* h_status);
    int blob_to_kzg_commitment_host(int n, const uint8_t* const* blobs, uint8_t* const* out, int* h_status);

    // verify_cell_kzg_proof_batch (eip7594/src/verifier.rs:72-112): returns a Status; *verified set on OK.
    int verify_cell_kzg_proof_batch_host(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                         const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                         uint64_t n_proofs, const uint8_t* const* proofs, int* verified);
    // the same check on flat arrays in this GPU's HBM: n * 48 commitment bytes (one per cell, not deduplicated), n u64 indices,
    // n * 2048 cell bytes, n * 48 proof bytes
    int verify_cell_kzg_proof_batch_device(uint64_t n, const uint8_t* d_commitments, const uint64_t* d_cell_indices,
                                           const uint8_t* d_cells, const uint8_t* d_proofs, int* verified, hipStream_t stream);
    // MANY independent verifications in one call (verify_many.hip): problem b has n_*[b] entries in commitments[b] / cell_indices[b]
    // / cells[b] / proofs[b]; status[b] = a Status (OK, ERR_INPUT, ERR_G1, ERR_SCALAR), verified[b] set when OK.  Returns
    // ERR_DEVICE on a HIP failure, OK otherwise.  Does not take mu_: runs next to the other entry points.
    int verify_cell_kzg_proof_batch_many_host(uint64_t n_batches, const uint64_t* n_commitments, const uint8_t* const* const* commitments,
                                              const uint64_t* n_indices, const uint64_t* const* cell_indices, const uint64_t* n_cells,
                                              const uint8_t* const* const* cells, const uint64_t* n_proofs,
                                              const uint8_t* const* const* proofs, int* verified, int* status);
    // verify_cell_kzg_proof_batch as the reference's users call it -- one problem per call, from many threads on one context
    // (bindings/node/src/lib.rs:92-299) -- WITHOUT asking them to batch: the first callers take the latency-optimised path
    // (verify_cell_kzg_proof_batch_host, one per engine lane); callers that arrive while every lane is verifying
    // are COMBINED: they queue their problems, one of them becomes the leader and runs everything queued as one
    // many-verification pass, the others sleep until their verdict is in.  Same verdicts and error split as the single path.
    int verify_cell_kzg_proof_batch_combined(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                             const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                             uint64_t n_proofs, const uint8_t* const* proofs, int* verified);
    // the same check sharded over ranks: every rank passes the WHOLE batch (the Fiat-Shamir transcript covers it) and its
    // slice [lo, hi) of the cell list, gets 96 bytes back; the gathered records go to _combine on any rank.
    int verify_cell_kzg_proof_batch_partial_host(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                                 const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                                 uint64_t n_proofs, const uint8_t* const* proofs, uint64_t lo, uint64_t hi,
                                                 uint8_t* out96);
    int verify_cell_kzg_proof_batch_combine_host(uint64_t n_partials, const uint8_t* partials96, int* verified);
    // recover_cells_and_kzg_proofs (eip7594/src/prover.rs:156-171)
    int recover_cells_and_kzg_proofs_host(uint64_t n_cells, const uint8_t* const* cells, uint64_t n_indices,
                                          const uint64_t* cell_indices, uint8_t* const* out_cells,
                                          uint8_t* const* out_proofs);

    // EIP-4844 single-point operations (crates/eip4844/src/{prover,verifier}.rs); Status returns
    int compute_kzg_proof_host(const uint8_t* blob, const uint8_t* z, uint8_t* out_proof, uint8_t* out_y);
    int compute_blob_kzg_proof_host(const uint8_t* blob, const uint8_t* commitment, uint8_t* out_proof);
    int verify_kzg_proof_host(const uint8_t* commitment, const uint8_t* z, const uint8_t* y, const uint8_t* proof, int* verified);
    int verify_blob_kzg_proof_host(const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof, int* verified);
    int verify_blob_kzg_proof_batch_host(uint64_t n_blobs, const uint8_t* const* blobs, uint64_t n_commitments,
                                         const uint8_t* const* commitments, uint64_t n_proofs, const uint8_t* const* proofs,
                                         int* verified);

    // device-resident recovery: flat [R][128][2048] cells in HBM + a 128-bit presence mask per blob (host); see verify.hip
    int recover_cells_and_kzg_proofs_device(int R, const uint8_t* d_cells, const uint64_t* present_masks, uint8_t* d_out_cells,
                                            uint8_t* d_out_proofs, int* status, hipStream_t stream);
    // batched form: R independent recoveries in one pass (ragged cell lists); status[r] per blob
    int recover_cells_and_kzg_proofs_batch_host(int R, const uint64_t* n_cells, const uint8_t* const* const* cells,
                                                const uint64_t* n_indices, const uint64_t* const* cell_indices,
                                                uint8_t* const* const* out_cells, uint8_t* const* const* out_proofs,
                                                int* status);

    // ---- stage-level hooks for the kernel parity tests (host buffers, canonical encodings) ----
    int test_fr_ntt4096(const uint8_t* in_be, uint8_t* out_be, int inverse_dit);
    int test_g1_fft128(const uint8_t* in_compressed, uint8_t* out_compressed, int n_lanes, int inverse);
    int test_fixed_msm(const uint8_t* scalars_be, int n_msm, uint8_t* out_compressed);
    int test_g1_decompress(const uint8_t* in, int n, int subgroup_check, int* h_status, uint8_t* out_recompressed);
    int test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp);

    // ---- per-stage HIP-event timing (bench.py's roofline leg) ----
    enum Stage { ST_BLOB_TO_COEFFS = 0, ST_COEFFS_TO_CELLS, ST_FK20_SCALARS, ST_MSM_FIXED, ST_G1_IFFT, ST_G1_FFT,
                 ST_COMPRESS, ST_G1_LINMAP, ST_COUNT };
    void set_profiling(bool on);
    // sums since the last call: ms[ST_COUNT], launches[ST_COUNT]; synchronises the device
    void get_stage_times(double* ms, uint64_t* launches);

    // ---- window tables.  A context is usable as soon as its START tables are up (sixteen-window GLV tables: FK20 1.6 GB,
    // commitments 0.8 GB); the wide ones (the widest GLV tables inside the budget: engine_tables.hip, build_final_tables) are
    // built by a helper thread and published with one pointer swap: every MSM launch takes a snapshot of the view, results
    // are identical for every table (tests), only the speed changes.  ETH_KZG_AMD_PROGRESSIVE=0 builds them before the
    // constructor returns.
    struct TableView {  // a snapshot: main = the complete table calls run on, next = a wider one under construction (its
                        // leading ready groups are used already; launch_msm); c / bytes describe main
        std::shared_ptr<SharedTable> main, next;
        int c = 0;          // window width
        size_t bytes = 0;
    };
    enum TableSel { TAB_FK = 0, TAB_SRS = 1 };
    TableView table_view(TableSel which) const;
    // 1 = final tables in place, 0 = still on the start tables (waits up to wait_ms; < 0: until done), 2 = the wide build
    // failed (out of memory ...) and the context stays on what it has
    int tables_ready(int wait_ms);
    // groups of the table under construction that MSMs already run on (all of them once it is complete)
    int table_groups_ready(TableSel which) const;
    // how the tables in use were allocated: out[0] = milliseconds spent in hipMalloc for their pieces (both tables), out[1] = the
    // longest single hipMalloc (ms), out[2] = pieces, out[3] = bytes.  A long single allocation is the driver waiting for memory
    // another process freed to be wiped: the process's GPU queues stand still meanwhile (tools/alloc_test/probe_stall.cpp)
    void table_build_info(double* out4) const;
    void stop_builder();  // abandon an unfinished build of the wide tables and join the helper thread
    size_t table_bytes() const { return table_view(TAB_FK).bytes + table_view(TAB_SRS).bytes; }
    int window_bits() const { return table_view(TAB_FK).c; }  // NOMINAL width of the FK20 table in use: its windows are ceil(128 / c) of mixed widths (launch.hpp)
    int window_count() const { const int c = window_bits(); return c > 0 ? (128 + c - 1) / c : 0; }  // windows per GLV half = gathered additions per base / 2
    const int* linmap_info() const { return slp_info_; }

private:
    void construct();           // the constructor's body; a throw is followed by teardown()
    void teardown() noexcept;   // the destructor's body
    void start_builder();       // progressive start: registers the engine and starts the table builder thread (engine_tables.hip)
    void init_constants();
    void init_linmap(const Fr8* w8192_mont);  // host copy of omega_8192^k
    void init_srs();
    void settle_streams();
    bool streams_overlap(hipStream_t a, hipStream_t b);
    void init_fk20();
    void init_verifier();
    int open_blobs_at(int n, const uint8_t* const* blobs, const Fr8* z_mont, bool want_proofs, uint8_t* h_proofs, Fr8* h_y_canon,
                      int* h_status);
    int pairing_check_4844(const void* d_points, const std::vector<Fr8>& sc0, const std::vector<Fr8>& sc1);
    // dsrc (device-resident form): the cells and proofs already sit in HBM -- they are copied device to device into the arena
    // instead of being gathered on the host and uploaded; the pointer arrays then address their pinned host mirror, which the
    // transcript hash reads, and whose cells arrive in chunks (an event per chunk)
    // where a single verification stages and computes: the engine's own stream / arena / pinned slab (under mu_), or those of a pass
    // slot (the caller holds the slot): so that up to four single verifications run side by side, each on its own hardware queue
    struct VerifyScratch {
        hipStream_t stream = nullptr;
        void* dev = nullptr;
        size_t dev_cap = 0;
        uint8_t* pin = nullptr;
        size_t pin_cap = 0;
    };
    struct VerifyDeviceSource {
        const uint8_t *d_cells, *d_proofs;  // flat [n][2048] / [n][48] in this GPU's memory
        hipEvent_t* chunk_events;           // chunk j = cells [j * chunk_cells, (j + 1) * chunk_cells) of the mirror
        int chunk_cells, n_chunks;
    };
    int verify_cells_partial(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                             const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells, uint64_t n_proofs,
                             const uint8_t* const* proofs, uint64_t lo, uint64_t hi, G1Affine* out2, bool* empty,
                             const VerifyDeviceSource* dsrc = nullptr, VerifyScratch* vs = nullptr);
    bool verify_cells_pairing(const G1Affine* pts2) const;
    // the same check with the second pair's Miller loop on a thread of the staging pool (the latency path of a single verification)
    bool verify_cells_pairing_split(const G1Affine* pts2);
    int rs_decode(int R, const uint8_t* d_cells, bool flat_source, const std::vector<int>& slot, const std::vector<int>& stof,
                  const std::vector<uint32_t>& present, int* st_out);
    int recover_batch_to_coeffs(int R, const uint64_t* n_cells, const uint8_t* const* const* cells,
                                const uint64_t* const* cell_indices, int* st_out);
    void ensure_workspace(int n);  // work_[0]; also makes stream_ wait for the last asynchronous call that used it
    void ensure_workspace(Work& w, int n, bool release = false);
    void ensure_staging(Work& w, int n);
    void run_proofs_from_coeffs(int n, uint8_t* d_proofs, hipStream_t st) { run_proofs_from_coeffs(work_[0], n, d_proofs, st); }
    // msm_cut > 0 (host-pointer path, scalars precomputed): the MSM stage in two launches around the cut -- PROOFS_HEAD issues the
    // arena set-up and the MSMs of blobs [0, msm_cut) and returns, PROOFS_TAIL the MSMs of [msm_cut, n) and everything after
    enum ProofsPhase { PROOFS_ALL = 0, PROOFS_HEAD = 1, PROOFS_TAIL = 2 };
    void run_proofs_from_coeffs(Work& w, int n, uint8_t* d_proofs, hipStream_t st, const TableView* tv_pre = nullptr, ProofsPhase phase = PROOFS_ALL,
                                int msm_cut = 0);
    void enqueue_compute(Work& w, int n, const uint8_t* d_blobs, uint8_t* d_cells, uint8_t* d_proofs, hipStream_t st,
                         hipEvent_t after_cells);
    Work& lease_work(int first, int last);  // locks and returns a free set among work_[first..last] (waits for whichever frees first)
    void release_work(Work& w);
    std::mutex lease_mu_;
    std::condition_variable lease_cv_;
    void set_error(const std::exception& e);
    void launch_msm(const void* scalars, TableSel table, void* out, int n_groups, int n_slices, int out_stride,
                    int brp_bits, hipStream_t st, int out_fmt = 0 /* launch::FMT_JACQ */);
    void launch_msm(const void* scalars, const TableView& tv, bool scalars_split, void* out, int n_groups, int n_slices, int out_stride,
                    int brp_bits, hipStream_t st, int out_fmt = 0 /* launch::FMT_JACQ */);
    void launch_msm_range(const void* scalars, const SharedTable& t, int g0, int gcnt, void* out, int n_groups, int n_slices, int out_stride,
                          int brp_bits, hipStream_t st, int out_fmt = 0);
    long msm_waves(long msms, int c) const;  // waves of the MSM launch launch_msm_range would issue (the side-stream decision of enqueue_compute)
    void build_final_tables();  // the wide tables: on the helper thread (progressive start) or inline
    void publish(TableSel which, const std::shared_ptr<SharedTable>& main, const std::shared_ptr<SharedTable>& next);
    void g1_fft128_full(void* X, int stride, int inverse, hipStream_t st);

    struct StageMark { int stage; int launches; hipEvent_t a, b; };
    int mark_begin(int stage, hipStream_t st);  // -> handle for mark_end (-1 when profiling is off)
    void mark_end(int mark, int launches, hipStream_t st);
    bool profiling_ = false;
    std::vector<StageMark> marks_;
    unsigned marks_gen_ = 0;
    std::mutex marks_mu_;

    int dev_ = 0;
    Comm* comm_ = nullptr;
    bool use_precomp_ = true;
    Knobs knobs_;            // every environment knob, read once when the context is created (knobs.hpp)
    int want_glv_c_ = 0;     // ETH_KZG_AMD_GLV_WINDOW: this GLV width exactly (0: the widest that fits memory and budget)
    double table_budget_gb_ = DEFAULT_TABLE_BUDGET_GB;  // upper bound for both tables together (constructor argument, ETH_KZG_AMD_TABLE_GB, or the default); <= 0: what the HBM holds
    mutable std::mutex tab_mu_;
    std::condition_variable tab_cv_;
    Published<SharedTable> pub_[2];  // per table kind: main / next / retired (host_sync.hpp); start tables stay alive for kernels already in flight
    int tables_state_ = 0;  // 0 building, 1 final, 2 wide build failed (guarded by tab_mu_)
    std::string tables_error_;
    std::thread builder_;
    std::atomic<bool> cancel_build_{false};
    bool progressive_build_ = false;  // the wide tables are built next to callers on the start tables: the builder leaves room on the GPU
    hipStream_t build_stream_ = nullptr;
    int wave_slots_ = 2048;  // CUs x 4 SIMDs x 2 waves: what one round of a ~240-VGPR point kernel occupies
    int device_batch_max_ = 4096;  // a device-resident prover call runs as sub-batches of at most this many blobs (ETH_KZG_AMD_DEVICE_BATCH_MAX)
    int msm_chunks_ = -1;    // -1: pick per launch (launch_msm); otherwise forced by ETH_KZG_AMD_MSM_CHUNKS
    bool msm_split_ = true;  // small batches: two lanes per MSM window where they still fit one round of the wave slots (ETH_KZG_AMD_MSM_SPLIT=0: off)
    hipStream_t stream_ = nullptr;
    std::recursive_mutex mu_;  // the verification / recovery / EIP-4844 / commitment paths and work_[0]: one call at a time
    static constexpr int NW = 4;  // work_[0]: the paths under mu_; work_[1..3]: compute_cells(_and_kzg_proofs) calls, concurrently
    Work work_[NW];
    std::unique_ptr<HostPool> host_pool_;
    std::once_flag host_pool_once_;

    // constants in HBM
    void* d_w8192_ = nullptr;     // Fr[8192] omega_8192^k, Montgomery
    void* d_w29_ = nullptr;       // the same in the unsaturated 9 x 29-bit form (36 B each) for the prover's Fr stages
    void* d_naf_ = nullptr;       // u32[128][2][33]: width-w NAF digits (signed bytes) of the GLV halves of omega_128^k
    Fp12w beta_;                  // cube root of unity in Fp: (beta x, y) = [lambda](x, y)
    void* d_srs_ = nullptr;       // G1Affine[4096] monomial SRS
    void* d_fk_bases_ = nullptr;  // G1Affine[128][64] FFT'd SRS vectors (batch_toeplitz.rs:46-61)
    BufferPool dev_pool_{false}, pin_pool_{true};
    Fr8 n_inv4096_, inv128_;

    // verifier / recovery constants
    void* d_coset_ = nullptr;      // Fr[8192] 7^i      (ReedSolomon coset generator, reed_solomon.rs:129)
    void* d_coset_inv_ = nullptr;  // Fr[8192] 7^-i
    Fr8 inv64_, n_inv8192_;
    std::shared_ptr<pairing::G2Prepared> g2_tau_, g2_neg_gen_;  // [tau^64]_2 and -[1]_2 (verifier.rs:88-90)
    std::shared_ptr<pairing::G2Prepared> g2_tau1_;             // [tau]_2 (EIP-4844 verification key)

    // verify workspace: one device arena + one pinned host staging buffer, grown on demand (guarded by mu_)
    void* v_dev_ = nullptr;
    size_t v_dev_cap_ = 0;
    uint8_t* v_pin_ = nullptr;
    size_t v_pin_cap_ = 0;
    hipStream_t v_side_ = nullptr;  // verification, round 3's form: the subgroup tests run here next to the point shifts on stream_
    bool v_two_streams_ = false;    // (both chains run in ONE launch on stream_, k_pip_shift_subgroup; the two-stream form of round 3 shared hardware queues)
    hipEvent_t v_decoded_ = nullptr, v_checked_ = nullptr;

    // serial-path lanes (lease_serial)
    const Engine* primary_ = nullptr;   // set in an auxiliary lane
    bool auxiliary_ = false;
    std::mutex lane_busy_;              // held while a leased call runs on THIS engine
    LanePool<Engine> lanes_;            // the auxiliary lanes of a context (host_sync.hpp)
    int max_lanes_ = 4;

    // combiner of concurrent single verifications (verify_many.hip: verify_cell_kzg_proof_batch_combined)
    struct VerifyRequest;
    Combiner<VerifyRequest> combiner_{VM_SLOTS};  // at most VM_SLOTS leaders run a pass at a time (host_sync.hpp)
    std::atomic<int> verify_inflight_{0};  // single verifications on the latency path right now
    int verify_lanes_ = 1;                 // concurrent single verifications on the latency path ; the others are combined
    int comb_max_cells_ = 1024;            // larger problems always take the single path (their transcript hash is the bound)

    // many-verification path (verify_many.hip): its own lock, stream, device arena and pinned slab
    // THREE pass slots, each with its own lock, stream, device arena and pinned slab: while one pass is on the GPU the other one's
    // staging, transcript hashes and pairing checks run on the host threads (concurrent single calls are combined into passes:
    // verify_cell_kzg_proof_batch_combined lets two leaders run at a time)
    static constexpr int VM_SLOTS = 3;
    struct VmSlot {  // (its lock: vm_slots_)
        VerifyScratch vs;   // for a "pass" of ONE problem: the single path's arena on this slot's stream
        void* dev = nullptr;
        size_t dev_cap = 0;
        uint8_t* pin = nullptr;
        size_t pin_cap = 0;
    };
    VmSlot vm_slot_[VM_SLOTS];
    SlotSet<VM_SLOTS> vm_slots_;  // leases of the pass slots (host_sync.hpp)
    std::unique_ptr<HostPool> stage_pool_;  // staging of single verifications (a few threads: the lane + the pass slots run side by side)
    std::once_flag stage_pool_once_;
    std::unique_ptr<HostPool> vm_pool_;  // the host threads of the many-verification passes (hashes, staging, pairing checks)
    std::once_flag vm_pool_once_;
    bool vm_search_ = true;  // ETH_KZG_AMD_VM_SEARCH=0: a pass whose folded check fails is re-checked problem by problem (round 3's form)
    int vm_small_max_ = -1;  // passes of at most this many problems take the short-chain form (verify_many.hip); -1: 2 x host threads
    uint8_t* vd_pin_ = nullptr;  // device-resident verification: the host mirror the transcript hash reads (grow-only, guarded by mu_)
    size_t vd_pin_cap_ = 0;
    static constexpr int VD_CHUNKS = 8;
    hipEvent_t vd_events_[VD_CHUNKS] = {};

    // small-batch circulant form of the two G1 transforms: term list, doubling tables (allocated on first use)
    void *d_circ_terms_ = nullptr;
    // verification batches of at least this many cells build byte-shifted point copies before the challenge is known
    // (k_verify.hip: k_pip_shift): behind the transcript hash for large batches, next to the subgroup tests (second stream) for
    // small ones.  ETH_KZG_AMD_PIP_SHIFT_MIN raises it (tests: the windowed form as the cross-check)
    int pip_shift_min_ = 1;
    int circ_T_ = 0, circ_per_lane_ = 0, circ_max_ = 2;  // largest batch on the circulant form: the measured cross-over with the compiled linear map (engine.hip: the constructor)
    Fr8 seg_shift_[3];  // 2^32, 2^64, 2^96 in Montgomery form
    // the two G1 transforms as one compiled linear map (g1_linmap.hpp, k_g1slp.hip).  SEVERAL compilations of the same map:
    // the throughput optimum (fewest point operations: 350 constant multiplications, 8-way Toom-Cook) for batches that fill
    // the chip, and depth-optimised ones for batches that do not -- there a dependency level lasts as long as its longest
    // operation and a constant multiplication has a SIMD to itself as long as waves <= SIMDs, so more multiplications with
    // shallower, doubling-free evaluation / interpolation trees (Karatsuba: 712 multiplications, 13 levels of single
    // additions instead of 25 levels with runs of up to 7 doublings) are faster.  Picked per launch by the number of 64-blob
    // lane groups (pick_slp_program); built on first use except the two tuned schedules.
    struct SlpLaunch { int kind, first, count; };
    enum SlpProgramId { SLP_TUNED_FUSED = 0, SLP_TUNED = 1, SLP_KARATSUBA = 2, SLP_DEPTH_456 = 3, SLP_DEPTH_606 = 4, SLP_DEPTH_372 = 5, SLP_COUNT = 6 };
    struct SlpProgram {
        std::vector<SlpLaunch> launches;
        void *d_words = nullptr, *d_naf = nullptr;  // d_naf may be shared with another program of the same plan (owns_naf)
        bool owns_naf = false, ready = false;
        int n_slots = 0;
        int info[4] = {0, 0, 0, 0};  // constant multiplications, additions, doublings, launches
    };
    SlpProgram slp_prog_[SLP_COUNT];
    std::mutex slp_build_mu_;
    std::vector<Fr8> w128_;                       // omega_128^e, kept for the programs built on first use
    const SlpProgram& slp_program(int id);        // builds it if need be (engine.hip: build_slp_program)
    void build_slp_program(int id);
    int pick_slp_program(int lanes) const;        // lanes = blobs rounded up to 64
    int slp_force_ = -1;                          // ETH_KZG_AMD_SLP_PROGRAM: this program at every batch size (tests, A/B runs)
    // a phase = one multiplication launch (level0 < 0) or a run of cheap dependency levels executed by ONE ticket-walking launch
    bool arena_signed_ = true;  // the prover's, the recovery's and the commitments' G1 points live in the signed 13 x 30-bit form from the MSM sums to the compression (ETH_KZG_AMD_ARENA_SIGNED=0: the 14 x 29-bit points and kernels of rounds 2-5, for A/B runs and as the tests' cross-check)
    int slp_fuse_min_ = 1024;  // measured: 512 blobs 5.47 (plain) against 5.55 ms (fused), 2048 blobs 16.13 against 15.93 ms
    int slp_info_[4] = {0, 0, 0, 0};  // constant multiplications, additions, doublings, launches of the tuned program
    Fr8 half_;  // 1/2 in Montgomery form: the scaling folded into the MSM scalars in linear-map mode

    // work_[0] under its historical names (grown on demand, guarded by mu_)
    int& cap_ = work_[0].cap;
    void *&d_coeffs_ = work_[0].coeffs, *&d_canon_ = work_[0].canon, *&d_scalars_ = work_[0].scalars, *&d_X_ = work_[0].X;
    int*& d_status_ = work_[0].status;
    uint8_t *d_in_ = nullptr, *d_cells_ = nullptr, *d_proofs_ = nullptr;  // staging for host API
    int stage_cap_ = 0;
};

}  // namespace kzg
