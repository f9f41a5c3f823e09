// Shared by the translation units of the engine (engine.hip: context life cycle and constants; engine_tables.hip: window tables,
// their registry and builder; engine_prover.hip: the prover paths; engine_testhooks.hip: stage-level test hooks).  Not installed.
#pragma once
#include "engine.hpp"
#include "curve.hpp"
#include "g1_linmap.hpp"
#include "launch.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <tuple>

extern "C" const unsigned char kzg_srs_begin[];
extern "C" const unsigned char kzg_srs_end[];

namespace kzg {

// a failed HIP call; `code` lets a caller tell an exhausted HBM (retry with a smaller sub-batch) from a broken device
struct HipError : std::runtime_error {
    hipError_t code;
    HipError(hipError_t c, const std::string& what) : std::runtime_error(what), code(c) {}
    bool out_of_memory() const { return code == hipErrorOutOfMemory || code == hipErrorMemoryAllocation; }
};
#define HIPCK(x)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess)                                                                                 \
            throw HipError(e_, std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + \
                                   std::to_string(__LINE__));                                                 \
    } while (0)

// batches up to this many lanes (blobs rounded up to 64) use the direct 8 x 16 G1 transforms (k_g1fft.hip)
// up to how many blobs the MSM stage runs one block per MSM (k_msm_glv_flat) instead of two lanes per window: measured on one box
// (tools/ab_flat_msm_max.sh, profiles/r6_ab_flat_msm_max.log).  With one lane per addition in the windowed kernel's tree the cross-over
// was 12 blobs; since that tree takes four lanes per addition it is 8 again: 9 blobs 1.61 -> 1.60 ms per step, 10: 1.65 -> 1.60, 12: 1.71 -> 1.61
#ifndef KZG_FLAT_MSM_MAX_SLICES  // (experiment builds of that A/B)
#define KZG_FLAT_MSM_MAX_SLICES 8
#endif
static constexpr int FLAT_MSM_MAX_SLICES = KZG_FLAT_MSM_MAX_SLICES;
static constexpr int SIDE_CELLS_MAX = 256;  // batches up to this size compute their cells on the work set's second stream, next to the proof stages (64 blobs: 0.08 of 3.7 ms)
static constexpr int N_BLOB = 4096, N_EXT = 8192, N_CELLS = 128, CELL_LEN = 64, BYTES_PER_BLOB = 131072, BYTES_PER_CELL = 2048;
static_assert(sizeof(Fr) == launch::SIZEOF_FR && sizeof(G1Affine) == launch::SIZEOF_G1AFFINE && sizeof(G1Jac) == launch::SIZEOF_G1JAC, "layout");

double trace_clock_ms();  // milliseconds since the library first asked (ETH_KZG_AMD_TRACE lines of different threads on one time line)

// A table is NOT one allocation.  Mapping 200+ GB with one hipMalloc takes the driver seconds during which every other HIP
// call of the process waits (measured in round 3: a 214 GB hipMalloc on the helper thread stalled the caller's launches for
// 4.3 s; the virtual-memory API that would back one address range piece by piece produced GPU memory faults on ROCm 7.0.2 and
// is gone).  Instead the kernels reach a table through a device array of BLOCK pointers -- two per group: its lower and
// upper windows (launch::TabBlocks) -- and the blocks live in PIECES of at most
// ~0.85 GB (one 0.8 GB block of the widest table; many blocks of a small one), each its own hipMalloc of a few milliseconds:
//   * other threads' HIP calls slip in between the pieces (tests/test_gpu_tables.py: a caller every 5 ms never waits long),
//   * a build is abandoned within one piece,
//   * the table is usable GROUP BY GROUP while it is built: ready_groups counts the leading groups whose entries are final,
//     an MSM stage runs those on the new table and the rest on the table the context started on (Engine::launch_msm).
struct Engine::SharedTable {
    int dev = 0, kind = 0, c = 0;  // kind: 2 = FK20 (128 groups of 64 bases), 3 = commitments (the monomial SRS as 64 groups of 64); both GLV tables
                                   // of nominal width c (launch.hpp: W = glv_windows(c) windows of mixed widths).  (0 and 1 were the plain tables of rounds 1-4.)
    int n_groups = 0, nb = 64;
    static constexpr int halves = 2;       // blocks per group: the lower and the upper windows
    size_t bytes = 0;                      // of all blocks
    size_t block_entries[2] = {0, 0};      // entries of a group's two blocks
    std::vector<void*> pieces;
    std::vector<void*> h_blocks;           // host copy of the pointer array (entries of unallocated blocks are null)
    void** d_blocks = nullptr;             // device: [n_groups * halves]
    int blocks_allocated = 0;
    std::atomic<int> ready_groups{0};      // leading groups whose entries are final and whose pointers are on the device
    std::atomic<int> state{0};             // 0 under construction, 1 complete, 2 abandoned (cancelled / out of memory): what is ready stays usable
    std::string why;                       // of state 2
    static size_t entry_bytes() { return launch::SIZEOF_TABP; }
    size_t block_bytes(int b) const { return block_entries[b % halves] * entry_bytes(); }
    void shape(int device, int kind_, int width, int groups) {
        dev = device; kind = kind_; c = width; n_groups = groups;
        const int WLc = launch::glv_lower_windows(c), Wc = launch::glv_windows(c);
        block_entries[0] = launch::glv_entries_per_base(c, 0, WLc) * (size_t)nb;
        block_entries[1] = launch::glv_entries_per_base(c, WLc, Wc) * (size_t)nb;
        bytes = 0;
        for (int b = 0; b < halves; b++) bytes += block_bytes(b) * (size_t)n_groups;
        h_blocks.assign((size_t)n_groups * halves, nullptr);
        pieces.reserve((size_t)n_groups * halves);  // never reallocated: table_build_info reads its size from other threads while the builder appends
    }
    // time spent in hipMalloc for the pieces (microseconds: total, longest single call) and their count -- written by the builder thread,
    // read by eth_kzg_amd_table_build_info from caller threads while the build runs: atomics, not plain doubles
    std::atomic<uint64_t> alloc_us{0}, alloc_us_max{0}, piece_count{0};
    bool trace_allocs = false;  // (ETH_KZG_AMD_TRACE of the context that builds the table)
    // allocate pieces until blocks [0, block_end) exist; false: out of memory (why is set) or cancelled
    bool alloc_until(int block_end, const std::atomic<bool>* cancel) {
        constexpr size_t PIECE = (size_t)850 << 20, HEADROOM = (size_t)8 << 30;  // (one piece for the whole table was round 3's single hipMalloc: seconds of stall for every HIP call of the process)
        const int total = n_groups * halves;
        if (block_end > total) block_end = total;
        if (!d_blocks) {
            if (hipMalloc((void**)&d_blocks, (size_t)total * sizeof(void*)) != hipSuccess) { (void)hipGetLastError(); d_blocks = nullptr; why = "hipMalloc of the block pointer array failed"; return false; }
            // hipMemset on device memory is ASYNCHRONOUS with respect to the host and runs on the NULL stream, which does not order
            // against this library's non-blocking streams: without the wait the zeroes could land AFTER the first chunk's block
            // pointers had been uploaded on the build stream -- the builder's kernels then wrote their entries to 0 + offset
            // (round 6: "write access to a read-only page at 0x500000", seen on a busy GPU only: four contexts under 16 threads)
            (void)hipMemset(d_blocks, 0, (size_t)total * sizeof(void*));
            (void)hipStreamSynchronize(nullptr);
        }
        while (blocks_allocated < block_end) {
            if (cancel && cancel->load()) { why = "cancelled"; return false; }
            int n = 0;
            size_t sz = 0;
            while (blocks_allocated + n < total && (n == 0 || sz + block_bytes(blocks_allocated + n) <= PIECE)) { sz += block_bytes(blocks_allocated + n); n++; }
            // fill_table has checked that the whole table fits, but another process (or thread) may have taken memory since: a table
            // that grows to the GPU's last byte leaves nothing for anybody's kernel scratch or batch buffers, and the runtime
            // aborts the process whose queue asks next (seen with two ranks racing for one GPU).  Stop a piece early instead.
            size_t free_now = 0, total_now = 0;
            if (hipMemGetInfo(&free_now, &total_now) == hipSuccess && free_now < sz + HEADROOM) {
                why = "not enough free device memory (taken by someone else since the build began)";
                return false;
            }
            void* p = nullptr;
            const auto a0 = std::chrono::steady_clock::now();
            const hipError_t e = hipMalloc(&p, sz);  // (physically contiguous pieces, hipDeviceMallocContiguous, measured the same in round 4)
            const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
            alloc_us.fetch_add((uint64_t)(dt * 1e3), std::memory_order_relaxed);
            if ((uint64_t)(dt * 1e3) > alloc_us_max.load(std::memory_order_relaxed)) alloc_us_max.store((uint64_t)(dt * 1e3), std::memory_order_relaxed);
            if (dt > 100 && trace_allocs) fprintf(stderr, "[context] @%.0f ms: hipMalloc of table piece %zu (%.2f GB) took %.0f ms\n", trace_clock_ms(), pieces.size(), sz / 1e9, dt);
            if (e != hipSuccess) { (void)hipGetLastError(); why = std::string("hipMalloc of a table piece: ") + hipGetErrorString(e); return false; }
            pieces.push_back(p);
            piece_count.store(pieces.size(), std::memory_order_release);
            // the HIP runtime serialises allocations and other calls on locks that are not fair: a thread that allocates piece
            // after piece without a pause can keep another thread's launch waiting for many pieces in a row
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            char* q = (char*)p;
            for (int k = 0; k < n; k++) { h_blocks[blocks_allocated + k] = q; q += block_bytes(blocks_allocated + k); }
            blocks_allocated += n;
        }
        return true;
    }
    ~SharedTable() {
        (void)hipSetDevice(dev);
        for (void* p : pieces) (void)hipFree(p);
        if (d_blocks) (void)hipFree(d_blocks);
    }
};
// engines whose helper thread may still be building tables, and the exit handler that stops them (engine.hip)
extern std::atomic<bool> g_exiting;
extern std::mutex g_engines_mu;
extern std::vector<Engine*> g_engines;
void stop_all_builders_at_exit();
struct BuildCancelled {};  // thrown out of a table build when its context (or the process) is going away
inline size_t glv_table_bytes(int c, int n_groups = 128) { return launch::table_glv_entries(c, n_groups, 64) * launch::SIZEOF_TABP; }

}  // namespace kzg
