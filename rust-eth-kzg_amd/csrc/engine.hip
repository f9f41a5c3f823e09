// Engine: context set-up and kernel orchestration for the KZG hot path on one MI355X.
// Mirrors ProverContext::new / FK20Prover::new (reference: crates/eip7594/src/prover.rs:62-94,
// crates/cryptography/kzg_multi_open/src/fk20/prover.rs:64-125, batch_toeplitz.rs:34-78) and the
// per-blob pipeline of compute_multi_opening_proofs (fk20/prover.rs:173-228) as a batch of kernels.
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

#define HIPCK(x)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + \
                                     std::to_string(__LINE__));                                               \
    } while (0)

// batches up to this many lanes (blobs rounded up to 64) use the direct 8 x 16 G1 transforms (k_g1fft.hip)
static constexpr int FLAT_MSM_MAX_SLICES = 8;  // measured: 1 blob 0.23 ms (vs 1.0), 16 blobs 1.6 ms (vs 1.06): one block per MSM pays while the chip is not full
static constexpr int SIDE_CELLS_MAX = 256;  // batches up to this size compute their cells on the work set's second stream, next to the proof stages (64 blobs: 0.08 of 3.7 ms)
static constexpr int LATENCY_MODE_MAX_LANES = 128;  // direct 8 x 16 transforms up to two 64-blob groups (128 blobs: 14.6 ms; the radix-2 network needs 17 ms at any batch below ~1000)
static constexpr int N_BLOB = 4096, N_EXT = 8192, N_CELLS = 128, CELL_LEN = 64, BYTES_PER_BLOB = 131072, BYTES_PER_CELL = 2048;
static_assert(sizeof(Fr) == launch::SIZEOF_FR && sizeof(G1Affine) == launch::SIZEOF_G1AFFINE && sizeof(G1Jac) == launch::SIZEOF_G1JAC, "layout");

// ---------------------------------------------------------------------------------------------
// host-side field helpers (field.hpp compiled for the host)
static Fr fr_from_u64(uint64_t v) {
    Fr a = zero<FrParams>();
    a.v[0] = (uint32_t)v;
    a.v[1] = (uint32_t)(v >> 32);
    return to_mont(a);
}
static Fr fr_pow_limbs(const Fr& base, const uint32_t* e, int nl) {
    Fr acc = one<FrParams>();
    for (int i = 32 * nl - 1; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
    }
    return acc;
}
// omega_n = 7^((r-1)/n)  (blstrs ROOT_OF_UNITY squared down; domain.rs:84-101)
static Fr fr_root_of_unity(int log_n) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
    e[0] -= 1;
    // e >>= log_n
    for (int s = 0; s < log_n; s++) {
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
    }
    return fr_pow_limbs(fr_from_u64(7), e, 8);
}
// GLV split of a public twiddle (host side, once per context).
// phi(x, y) = (beta x, y) acts on G1 as multiplication by lambda = z^2 - 1 (z the BLS parameter), a 128-bit cube
// root of unity mod r.  k = k1 + k2 lambda with k2 = floor(k / lambda), both < 2^128; each half is then recoded in
// width-w non-adjacent form (init_constants; consumed by mul_by_twiddle in k_g1fft.hip and by k_g1circ.hip's term list).
typedef unsigned __int128 u128;
static const u128 GLV_LAMBDA = ((u128)0xac45a4010001a402ULL << 64) | 0x00000000ffffffffULL;
static void glv_split(const Fr& canon, u128& k1, u128& k2) {
    // binary long division of the 256-bit integer by lambda (remainder < lambda < 2^128; one extra carry bit)
    u128 rem = 0, q = 0;
    for (int i = 255; i >= 0; i--) {
        bool top = (rem >> 127) & 1;
        rem = (rem << 1) | ((canon.v[i >> 5] >> (i & 31)) & 1);
        q <<= 1;  // quotient < 2^128: the bits shifted out here are zero
        if (top || rem >= GLV_LAMBDA) { rem -= GLV_LAMBDA; q |= 1; }
    }
    k1 = rem;
    k2 = q;
}
// GLV split + width-w NAF of one public constant (Montgomery form): 2 x TWIDDLE_WORDS words, one signed byte per digit
// (consumed by mul_by_recoded, g1_mulc.hpp).  Both steps are verified by recomputing the constant from its digits.
static void recode_glv_wnaf(const Fr& k_mont, const Fr& lambda_mont, uint32_t* out) {
    constexpr int TWW = launch::TWIDDLE_WORDS, W = launch::TWIDDLE_WNAF_W;
    u128 r, q;
    glv_split(from_mont(k_mont), r, q);  // k = r + q lambda, 0 <= r < lambda, 0 <= q <= lambda + 1
    // The split is unique only up to the lattice of (a, b) with a + b lambda = 0 mod r; the four representatives around
    // the origin -- (r, q), (r - lambda, q + 1), (r - 1, q - lambda - 1), (r - lambda - 1, q - lambda) -- cost different
    // numbers of additions, so take the cheapest (the constant is public and recoded once; ~4 % fewer instructions per
    // multiplication on average).  Halves are signed: magnitude + sign, a negative half flips its digits.
    struct Half { u128 mag; bool neg; };
    auto signed_sub = [](u128 a, u128 b) { return a >= b ? Half{a - b, false} : Half{b - a, true}; };  // a - b
    const Half cand[4][2] = {
        {Half{r, false}, Half{q, false}},
        {signed_sub(r, GLV_LAMBDA), Half{q + 1, false}},
        {signed_sub(r, 1), signed_sub(q, GLV_LAMBDA + 1)},
        {signed_sub(r, GLV_LAMBDA + 1), signed_sub(q, GLV_LAMBDA)},
    };
    auto fr_of = [](const Half& h) {
        Fr a = zero<FrParams>();
        for (int i = 0; i < 4; i++) a.v[i] = (uint32_t)(h.mag >> (32 * i));
        a = to_mont(a);
        return h.neg ? neg(a) : a;
    };
    // width-w NAF of a magnitude into signed bytes (flipped when the half is negative); returns (length, non-zero digits)
    auto naf = [&](const Half& h, int8_t* dg, int& len, int& weight) {
        u128 v = h.mag;
        len = weight = 0;
        for (int t = 0; v != 0; t++, v >>= 1) {
            if (t >= 4 * TWW) throw std::runtime_error("constant recoding too long");
            if (!(v & 1)) continue;
            int d = (int)(v & ((1u << W) - 1));        // v mods 2^w: odd residue in (-2^(w-1), 2^(w-1))
            if (d >= (1 << (W - 1))) d -= 1 << W;
            if (d < 0) v += (u128)(-d); else v -= (u128)d;
            dg[t] = (int8_t)(h.neg ? -d : d);
            len = t + 1;
            weight++;
        }
    };
    double best_cost = 0;
    int best = -1;
    std::vector<uint32_t> best_words;
    for (int c = 0; c < 4; c++) {
        if (!eq(add(fr_of(cand[c][0]), mul(fr_of(cand[c][1]), lambda_mont)), k_mont)) throw std::runtime_error("GLV split failed");
        std::vector<uint32_t> words(2 * TWW, 0u);
        int len[2], wt[2];
        for (int h = 0; h < 2; h++) naf(cand[c][h], reinterpret_cast<int8_t*>(&words[(size_t)h * TWW]), len[h], wt[h]);
        const double cost = linmap::COST_DBL * std::max(len[0], len[1]) + 5.0e3 * (wt[0] + wt[1]);  // doublings shared, mixed additions
        if (best < 0 || cost < best_cost) { best = c; best_cost = cost; best_words = words; }
    }
    for (int i = 0; i < 2 * TWW; i++) out[i] = best_words[i];
    for (int h = 0; h < 2; h++) {  // the digits must add up to the signed half again
        const int8_t* dg = reinterpret_cast<const int8_t*>(out + (size_t)h * TWW);
        Fr back = zero<FrParams>(), pw = one<FrParams>();
        for (int t = 0; t < 4 * TWW; t++, pw = add(pw, pw)) {
            const int d = dg[t];
            if (!d) continue;
            Fr term = pw;
            for (int m = 1; m < (d < 0 ? -d : d); m++) term = add(term, pw);  // |d| * 2^t
            back = d < 0 ? sub(back, term) : add(back, term);
        }
        if (!eq(back, fr_of(cand[best][h]))) throw std::runtime_error("constant recoding failed");
    }
}
// milliseconds since the library first asked (ETH_KZG_AMD_TRACE lines of different threads on one time line)
static double trace_clock_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// The text of a device error belongs to the call that failed, and calls run concurrently: keep it per thread.
static thread_local std::string t_last_error;
const std::string& Engine::last_error() const { return t_last_error; }
void Engine::set_error(const std::exception& e) { t_last_error = e.what(); }

HostPool::HostPool(int threads, int device) {
    for (int i = 0; i < threads; i++) th_.emplace_back([this, device] { run(device); });
}
HostPool::~HostPool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
}
void HostPool::submit(std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push_back(std::move(fn));
    }
    cv_.notify_one();
}
void HostPool::run(int device) {
    (void)hipSetDevice(device);
    for (;;) {
        std::function<void()> fn;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;
            fn = std::move(q_.front());
            q_.pop_front();
        }
        fn();
    }
}

// ---------------------------------------------------------------------------------------------
BufferPool::~BufferPool() {
    for (auto& f : free_) {
        if (host_) (void)hipHostFree(f.second);
        else (void)hipFree(f.second);
    }
}
void* BufferPool::get(size_t bytes, size_t* cls) {
    size_t c = 256;
    while (c < bytes) c <<= 1;
    *cls = c;
    for (size_t i = 0; i < free_.size(); i++)
        if (free_[i].first == c) {
            void* p = free_[i].second;
            free_[i] = free_.back();
            free_.pop_back();
            pooled_ -= c;
            return p;
        }
    void* p = nullptr;
    hipError_t e = host_ ? hipHostMalloc(&p, c, hipHostMallocDefault) : hipMalloc(&p, c);
    if (e != hipSuccess) {  // give the cached blocks back and retry once
        (void)hipGetLastError();
        for (auto& f : free_) { if (host_) (void)hipHostFree(f.second); else (void)hipFree(f.second); }
        free_.clear();
        pooled_ = 0;
        e = host_ ? hipHostMalloc(&p, c, hipHostMallocDefault) : hipMalloc(&p, c);
        if (e != hipSuccess) throw std::runtime_error(host_ ? "hipHostMalloc failed" : "hipMalloc failed");
    }
    return p;
}
void BufferPool::put(void* p, size_t cls) {
    constexpr size_t KEEP = 6ull << 30;  // cached bytes per pool; larger working sets fall back to plain free
    if (pooled_ + cls > KEEP) {
        if (host_) (void)hipHostFree(p); else (void)hipFree(p);
        return;
    }
    free_.emplace_back(cls, p);
    pooled_ += cls;
}
PoolBuf::PoolBuf(Engine& e, size_t bytes, bool pinned_host) : e_(e), host_(pinned_host) {
    p = (host_ ? e_.pin_pool_ : e_.dev_pool_).get(bytes ? bytes : 1, &cls_);
}
PoolBuf::~PoolBuf() {
    (void)hipStreamSynchronize(e_.stream_);  // nothing in flight may still use the block (free when the stream is idle)
    (host_ ? e_.pin_pool_ : e_.dev_pool_).put(p, cls_);
}

// ---------------------------------------------------------------------------------------------
// Engines whose helper thread may still be building tables.  A process that exits without freeing its context (legal for the
// reference's API users: the context is often a process-lifetime object) must not tear the HIP runtime down under that
// thread: the handler below runs before the runtime's own exit handlers (it is registered later), raises the cancel flags
// and waits for the builders, each of which gives up within one 2 GB piece.
static std::atomic<bool> g_exiting{false};
static std::mutex g_engines_mu;
static std::vector<Engine*> g_engines;
void Engine::stop_builder() {
    cancel_build_.store(true);
    if (builder_.joinable()) builder_.join();
}
static void stop_all_builders_at_exit() {
    g_exiting.store(true);
    std::vector<Engine*> live;
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        live = g_engines;
    }
    for (Engine* e : live) e->stop_builder();
}

Engine::SerialLease Engine::lease_serial() {
    SerialLease L;
    L.busy = std::unique_lock<std::mutex>(lane_busy_, std::try_to_lock);
    if (L.busy.owns_lock() || auxiliary_ || max_lanes_ <= 1) {
        if (!L.busy.owns_lock()) L.busy.lock();
        L.e = this;
        return L;
    }
    bool create = false;
    {
        std::lock_guard<std::mutex> lk(lanes_mu_);
        for (auto& a : aux_) {
            L.busy = std::unique_lock<std::mutex>(a->lane_busy_, std::try_to_lock);
            if (L.busy.owns_lock()) { L.e = a.get(); return L; }
        }
        if ((int)aux_.size() + lanes_pending_ + 1 < max_lanes_) { lanes_pending_++; create = true; }
    }
    if (create) {
        // a new lane is built OUTSIDE lanes_mu_ (0.1 s of set-up kernels): callers that find the existing lanes busy meanwhile
        // queue on one of those instead of behind this constructor; the lane shares this engine's window tables (no registry
        // look-up, no lock a table build could hold)
        std::unique_ptr<Engine> fresh;
        try {
            fresh.reset(new Engine(use_precomp_, dev_, /*primary=*/this));
        } catch (const std::exception&) {
            (void)hipGetLastError();  // no room for another lane: queue on an existing one
        }
        std::lock_guard<std::mutex> lk(lanes_mu_);
        lanes_pending_--;
        if (fresh) {
            L.busy = std::unique_lock<std::mutex>(fresh->lane_busy_);
            L.e = fresh.get();
            aux_.push_back(std::move(fresh));
            return L;
        }
    }
    // all lanes busy: wait on one of them, round-robin
    Engine* pick = this;
    {
        std::lock_guard<std::mutex> lk(lanes_mu_);
        const unsigned k = lane_rr_.fetch_add(1) % (unsigned)(aux_.size() + 1);
        if (k > 0) pick = aux_[k - 1].get();
    }
    L.busy = std::unique_lock<std::mutex>(pick->lane_busy_);
    L.e = pick;
    return L;
}

Engine::Engine(bool use_precomp, int device, const Engine* primary) : dev_(device), use_precomp_(use_precomp), primary_(primary), auxiliary_(primary != nullptr) {
    if (const char* s = getenv("ETH_KZG_AMD_SERIAL_LANES")) {
        const int v = atoi(s);
        if (v >= 1 && v <= 16) max_lanes_ = v;
    }
    if (use_precomp) {
        // default: the widest GLV table that fits (16-bit windows: 206 GB, 16 gathered additions per base), see build_final_tables
        if (const char* s = getenv("ETH_KZG_AMD_WINDOW")) {  // tuning knob: plain FK20 table of this window width (8, 10, 12, 13, 14)
            int c = atoi(s);
            if (c == 8 || c == 10 || c == 12 || c == 13 || c == 14) want_plain_c_ = c;
        }
        if (const char* s = getenv("ETH_KZG_AMD_GLV_WINDOW")) {  // tuning knob: exactly this GLV width (16, 15, 14, 12, 8)
            int c = atoi(s);
            if (launch::glv_width_supported(c)) want_glv_c_ = c;
        }
        if (const char* s = getenv("ETH_KZG_AMD_TABLE_GB")) {  // memory budget for the window tables (both together), in GB
            const double g = atof(s);
            if (g > 0) table_budget_gb_ = g;
        }
    }
    if (const char* s = getenv("ETH_KZG_AMD_VERIFY_LANES")) { const int v = atoi(s); if (v >= 0 && v <= 16) verify_lanes_ = v; }
    if (const char* s = getenv("ETH_KZG_AMD_VM_SMALL")) vm_small_max_ = atoi(s);  // 0 disables the short-chain form of small passes
    if (const char* s = getenv("ETH_KZG_AMD_VM_SEARCH")) vm_search_ = atoi(s) != 0;  // tests: the per-problem re-check of round 3 as the cross-check
    if (const char* s = getenv("ETH_KZG_AMD_PIP_SHIFT_MIN")) {  // tuning knob: smallest cell count verified with byte-shifted point copies
        const int v = atoi(s);
        if (v >= 1) pip_shift_min_ = v;
    }
    // largest batch on the circulant form: its cost grows by 0.3 ms per blob (1 blob 1.48 ms, 4: 2.33, 5: 3.0, 8: 3.5), the compiled
    // map in its Karatsuba compilation is flat (5 - 8 blobs: 2.5 - 2.7 ms with the flat MSM) -- round 3's cross-over, against the
    // tuned program, was 8
    circ_max_ = 4;
    if (const char* s = getenv("ETH_KZG_AMD_CIRC_MAX")) {  // tuning knob: largest batch served by the circulant form (0 disables it)
        int v = atoi(s);
        if (v >= 0 && v <= 64) circ_max_ = v;
    }
    const bool trace = getenv("ETH_KZG_AMD_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[context] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    HIPCK(hipSetDevice(dev_));
    {
        int cus = 0;
        HIPCK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_));
        wave_slots_ = cus * 4 * 2;  // point kernels hold ~240 VGPRs: two waves per SIMD
        if (const char* s = getenv("ETH_KZG_AMD_MSM_CHUNKS")) {
            int v = atoi(s);
            if (v == 0 || v == 1 || v == 2 || v == 4 || v == 8) msm_chunks_ = v;
        }
    }
    HIPCK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (!primary_) {  // the table builder runs at the lowest stream priority: callers' kernels are dispatched ahead of its waves
        int prio_low = 0, prio_high = 0;
        HIPCK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIPCK(hipStreamCreateWithPriority(&build_stream_, hipStreamNonBlocking, prio_low));
    }
    if (const char* s = getenv("ETH_KZG_AMD_VERIFY_SIDE_STREAM")) v_two_streams_ = atoi(s) != 0;
    if (v_two_streams_) HIPCK(hipStreamCreateWithFlags(&v_side_, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&v_decoded_, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&v_checked_, hipEventDisableTiming));
    for (Work& w : work_) {
        HIPCK(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_in, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_cells, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_cells_host, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_coeffs, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&w.ev_side, hipEventDisableTiming));
        if (&w == &work_[0]) continue;  // the paths under mu_ run on stream_
        if (primary_) continue;         // an engine lane serves the serial paths only: no prover streams (HIP maps all of a process's streams onto four hardware queues)
        // ROCm multiplexes streams onto a few hardware queues per priority level and a queue runs in order: a copy stream
        // that shares its queue with a compute stream delivers the cells only after the MSMs.  Copy streams get the high
        // priority level, i.e. queues of their own.
        int prio_low = 0, prio_high = 0;
        HIPCK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIPCK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        HIPCK(hipStreamCreateWithPriority(&w.copy, hipStreamNonBlocking, prio_high));
    }
    launch::init_attributes();
    if (!primary_) settle_streams();
    lap("HIP runtime + stream");
    init_constants();
    lap("twiddles, recodings");
    init_srs();
    lap("SRS decompress");
    init_fk20();
    lap("FK20 bases + start tables");
    init_verifier();
    HIPCK(hipStreamSynchronize(stream_));
    lap("verifier (G2 lines, cosets)");
}

// ROCm maps a process's streams onto four hardware queues per priority level, and which streams end up sharing a queue depends
// on what the host application created before this context (a PyTorch process holds queues of its own).  Two streams on one
// queue run in order: the four concurrent single verifications of a multi-threaded caller (the lane on stream_, the pass slots
// on the work sets' streams) then take turns.  So the mapping is measured, not assumed: a one-wave kernel that stays resident
// for 0.3 ms is put on two streams at once, and if the pair takes twice that, the second stream is replaced by a fresh one
// (the rejected ones are kept until the end, so that the runtime's least-used-queue rule moves on).  The test errs to one side
// only -- a busy GPU can make an overlapping pair look serial, never the reverse -- and the whole probe takes a few ms.
bool Engine::streams_overlap(hipStream_t a, hipStream_t b) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev_) != hipSuccess || khz <= 0) khz = 100000;
    const double spin_us = 300.0;
    const uint64_t ticks = (uint64_t)(spin_us * 1e-3 * khz);
    for (int t = 0; t < 3; t++) {
        HIPCK(hipStreamSynchronize(a));
        HIPCK(hipStreamSynchronize(b));
        const auto t0 = std::chrono::steady_clock::now();
        launch::spin(ticks, a);
        launch::spin(ticks, b);
        HIPCK(hipStreamSynchronize(a));
        HIPCK(hipStreamSynchronize(b));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < 1.6 * spin_us) return true;
    }
    return false;
}
void Engine::settle_streams() {
    if (const char* s = getenv("ETH_KZG_AMD_SETTLE_STREAMS"))
        if (atoi(s) == 0) return;
    const bool trace = getenv("ETH_KZG_AMD_TRACE_STREAMS") != nullptr;
    launch::spin(1, stream_);  // first launch of the kernel (and first use of the stream) outside the measurement
    std::vector<hipStream_t> chosen{stream_}, rejected;
    bool gave_up = false;
    for (int i = 1; i < NW; i++) {
        Work& w = work_[i];
        if (!w.stream || gave_up) continue;
        for (int attempt = 0;; attempt++) {
            launch::spin(1, w.stream);
            bool ok = true;
            for (hipStream_t c : chosen) ok = ok && streams_overlap(c, w.stream);
            if (ok || attempt == 5) {
                if (trace) fprintf(stderr, "[eth_kzg_amd] work set %d: stream %s after %d replacement(s)\n", i, ok ? "runs beside the others" : "still shares a queue", attempt);
                if (!ok) gave_up = true;  // a GPU this busy (another process's kernels) shows every pair as serial: no point in trying the other sets
                break;
            }
            rejected.push_back(w.stream);
            HIPCK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        }
        chosen.push_back(w.stream);
    }
    for (hipStream_t r : rejected) hipStreamDestroy(r);
}

Engine::~Engine() {
    hipSetDevice(dev_);
    {
        std::lock_guard<std::mutex> lk(lanes_mu_);
        aux_.clear();
    }
    stop_builder();  // an unfinished build of the wide tables is abandoned (within one 2 GB piece)
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        g_engines.erase(std::remove(g_engines.begin(), g_engines.end(), this), g_engines.end());
    }
    if (build_stream_) hipStreamDestroy(build_stream_);
    host_pool_.reset();  // joins the helper threads before anything they might touch goes away
    vm_pool_.reset();
    stage_pool_.reset();
    void* ptrs[] = {d_w8192_, d_w29_, d_naf_, d_srs_, d_fk_bases_, d_in_, d_cells_, d_proofs_, d_coset_, d_coset_inv_, d_circ_terms_, d_slp_levels_};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (SlpProgram& P : slp_prog_) {
        if (P.d_words) hipFree(P.d_words);
        if (P.d_naf && P.owns_naf) hipFree(P.d_naf);
    }
    for (Work& w : work_) {
        void* dev[] = {w.coeffs, w.canon, w.scalars, w.X, w.status, w.dft_tmp, w.dft_prod, w.circ_table, w.slp_arena, w.slp_sync, w.d_in, w.d_cells, w.d_proofs, w.msm_partial};
        for (void* p : dev)
            if (p) hipFree(p);
        void* pin[] = {w.h_in, w.h_cells, w.h_proofs, w.h_status};
        for (void* p : pin)
            if (p) hipHostFree(p);
        hipEvent_t evs[] = {w.done, w.ev_in, w.ev_cells, w.ev_cells_host, w.ev_done, w.ev_coeffs, w.ev_side};
        for (hipEvent_t e : evs)
            if (e) hipEventDestroy(e);
        for (hipEvent_t e : w.sub_events) hipEventDestroy(e);
        if (w.stream) hipStreamDestroy(w.stream);
        if (w.copy) hipStreamDestroy(w.copy);
    }
    if (v_dev_) hipFree(v_dev_);
    if (v_pin_) hipHostFree(v_pin_);
    for (VmSlot& v : vm_slot_) {
        if (v.dev) hipFree(v.dev);
        if (v.pin) hipHostFree(v.pin);
        if (v.vs.dev) hipFree(v.vs.dev);
        if (v.vs.pin) hipHostFree(v.vs.pin);
    }
    if (vd_pin_) hipHostFree(vd_pin_);
    for (hipEvent_t e : vd_events_)
        if (e) hipEventDestroy(e);
    if (v_side_) hipStreamDestroy(v_side_);
    if (v_decoded_) hipEventDestroy(v_decoded_);
    if (v_checked_) hipEventDestroy(v_checked_);
    if (stream_) hipStreamDestroy(stream_);
}

void Engine::init_constants() {
    // omega_8192 powers (Domain::new roots, domain.rs:55-62)
    std::vector<Fr> w(N_EXT);
    Fr g = fr_root_of_unity(13);
    w[0] = one<FrParams>();
    for (int i = 1; i < N_EXT; i++) w[i] = mul(w[i - 1], g);
    if (!eq(mul(w[N_EXT - 1], g), one<FrParams>())) throw std::runtime_error("omega_8192 has wrong order");
    HIPCK(hipMalloc(&d_w8192_, N_EXT * sizeof(Fr)));
    HIPCK(hipMemcpy(d_w8192_, w.data(), N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
    {   // the same table in the unsaturated 9 x 29-bit form of the prover's Fr stages (k_ntt.hip, fr29.hpp)
        std::vector<uint32_t> w29((size_t)N_EXT * 9);
        launch::ntt_twiddles29(w.data(), w29.data());
        HIPCK(hipMalloc(&d_w29_, w29.size() * 4));
        HIPCK(hipMemcpy(d_w29_, w29.data(), w29.size() * 4, hipMemcpyHostToDevice));
    }
    // GLV + width-w NAF recoding of omega_128^k = w[64k] (the G1-FFT twiddles)
    {
        Fr lam = zero<FrParams>();
        for (int i = 0; i < 4; i++) lam.v[i] = (uint32_t)(GLV_LAMBDA >> (32 * i));
        Fr lm = to_mont(lam);
        if (!is_zero(add(add(sqr(lm), lm), one<FrParams>()))) throw std::runtime_error("GLV lambda is not a cube root of unity");
        // width-w NAF of both GLV halves of every twiddle omega_128^k = w[64 k] (k_g1fft.hip: mul_by_twiddle)
        constexpr int TWW = launch::TWIDDLE_WORDS;
        std::vector<uint32_t> tw((size_t)128 * 2 * TWW, 0u);
        for (int k = 0; k < 128; k++) recode_glv_wnaf(w[64 * k], lm, &tw[(size_t)k * 2 * TWW]);
        HIPCK(hipMalloc(&d_naf_, tw.size() * 4));
        HIPCK(hipMemcpy(d_naf_, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
        // Small-batch circulant form (k_g1circ.hip): symbol c_d = sum_{t<64} omega_128^(d t); its non-zero entries,
        // GLV-split and NAF-recoded, become one flat list of +-D[k-d][half][t] terms dealt round-robin to 256 lanes.
        std::vector<uint32_t> plus, minus;
        int max_t = 0;
        for (int d = 0; d < 128; d++) {
            Fr c = zero<FrParams>();
            for (int t = 0; t < 64; t++) c = add(c, w[(64 * d * t) & (N_EXT - 1)]);
            if (d != 0 && (d & 1) == 0) {
                if (!is_zero(c)) throw std::runtime_error("circulant symbol: even entry is not zero");
                continue;
            }
            u128 kk[2];
            glv_split(from_mont(c), kk[0], kk[1]);
            Fr back = zero<FrParams>();  // the recoded terms must add up to c_d again
            for (int half = 0; half < 2; half++) {
                u128 k = kk[half];
                Fr pw = half ? lm : one<FrParams>();  // 2^t (half 0) or 2^t lambda (half 1)
                for (int t = 0; k != 0; t++, k >>= 1, pw = add(pw, pw)) {
                    if (!(k & 1)) continue;
                    const int dg = 2 - (int)(k & 3);  // NAF digit +-1
                    if (dg < 0) k += 1;               // k - dg; the shift then drops the zero low bit
                    const uint32_t word = (uint32_t)d | ((uint32_t)t << 7) | ((uint32_t)half << 15) | ((dg < 0 ? 1u : 0u) << 16) | (1u << 17);
                    (dg < 0 ? minus : plus).push_back(word);
                    back = dg < 0 ? sub(back, pw) : add(back, pw);
                    if (t > max_t) max_t = t;
                }
            }
            if (!eq(back, c)) throw std::runtime_error("circulant term list does not reproduce its symbol");
        }
        if ((int)plus.size() < launch::CIRC_LANES || max_t >= 255) throw std::runtime_error("circulant term list is malformed");
        std::vector<uint32_t> all(plus);  // positive terms first: row 0 (every lane's starting value) needs no negation
        all.insert(all.end(), minus.begin(), minus.end());
        circ_T_ = max_t + 1;
        circ_per_lane_ = ((int)all.size() + launch::CIRC_LANES - 1) / launch::CIRC_LANES;
        all.resize((size_t)circ_per_lane_ * launch::CIRC_LANES, 0u);  // padding: valid bit clear
        HIPCK(hipMalloc(&d_circ_terms_, all.size() * 4));
        HIPCK(hipMemcpy(d_circ_terms_, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    }
    init_linmap(reinterpret_cast<const Fr8*>(w.data()));
    for (int sgi = 0; sgi < 3; sgi++) {  // Montgomery forms of 2^32, 2^64, 2^96 (k_fk20_scalars' segment copies)
        Fr v = zero<FrParams>();
        v.v[sgi + 1] = 1;
        v = to_mont(v);
        memcpy(&seg_shift_[sgi], &v, 32);
    }
    Fr i4096 = inv(fr_from_u64(N_BLOB)), i128 = inv(fr_from_u64(128));
    memcpy(&n_inv4096_, &i4096, 32);
    memcpy(&inv128_, &i128, 32);
}

// The two G1 transforms of the prover as one straight-line program of point operations (g1_linmap.hpp): built, checked
// against the definition of the map over Fr (plan and scheduled slot program), constants recoded, uploaded.
static linmap::Strategy slp_strategy(int id) {
    linmap::Strategy s;
    s.allow_toom8 = getenv("ETH_KZG_AMD_NO_TOOM8") == nullptr;
    auto fixed = [&](int k4, int k8, int k16, int k32) {
        s.tuned = false;
        s.balanced_lincomb = true;
        s.hankel_split = {{2, 2}, {4, k4}, {8, k8}, {16, k16}, {32, k32}};
    };
    // (tools/linmap_explore.cpp lists every assignment of splits with its multiplication count and the latency of its cheap
    // levels; these are points of that Pareto front)
    switch (id) {
        case 2: fixed(2, 2, 2, 2); break;  // Karatsuba throughout: 712 multiplications, 13 levels of single additions
        case 3: fixed(4, 2, 4, 2); break;  // 456 multiplications, 18 levels, at most two doublings in front of an addition
        case 4: fixed(2, 2, 2, 4); break;  // 606 multiplications, 15 levels
        case 5: fixed(4, 2, 4, 8); break;  // 372 multiplications, 19 levels (8-way split of the 32-point products only)
        default: break;                    // tuned by operation count: 350 multiplications
    }
    return s;
}
void Engine::build_slp_program(int id) {
    const Fr* w128p = reinterpret_cast<const Fr*>(w128_.data());
    const std::vector<Fr> w128(w128p, w128p + 128);
    const bool verbose = getenv("ETH_KZG_AMD_TRACE") != nullptr;
    linmap::Plan plan = linmap::build_fk20_proofs_plan(w128, slp_strategy(id), verbose);
    {   // the executor leaves output p in arena slot 128 + p, and the proofs are wanted in bit-reversed FFT order
        std::vector<linmap::Ref> perm(128);
        for (int p = 0; p < 128; p++) {
            int k = 0;
            for (int b = 0; b < 7; b++) k |= ((p >> b) & 1) << (6 - b);
            perm[p] = plan.outputs[k];
        }
        plan.outputs = perm;
    }
    // a + b / a - b pairs as ONE operation only in the large-batch schedule: for batches that leave the chip part empty a
    // step lasts as long as its longest operation, and the fused pair is 15 % longer than an addition (64 blobs: 2.65 against
    // 2.73 ms for the map; 2048 blobs: 16.15 against 16.0 ms -- fewer, fuller rounds win there)
    const linmap::Schedule sched = linmap::make_schedule(plan, /*fuse_add_sub=*/id == SLP_TUNED_FUSED);
    // self-check: definition of the map vs the plan vs the scheduled slot program, over Fr
    uint64_t st = 0x853c49e6748fea9bull + (uint64_t)id;
    for (int it = 0; it < 2; it++) {
        std::vector<Fr> in(128);
        for (auto& v : in) {
            for (int i = 0; i < 8; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v.v[i] = (uint32_t)(st >> 16); }
            v.v[7] &= 0x3fffffffu;
        }
        const auto want = linmap::fk20_proofs_map_by_definition(w128, in);
        const auto got1 = linmap::run_over_fr(plan, in);
        const auto got2 = linmap::run_schedule_over_fr(sched, plan.consts, 128, 128, in);
        for (int p = 0; p < 128; p++) {
            int k = 0;
            for (int b = 0; b < 7; b++) k |= ((p >> b) & 1) << (6 - b);
            if (!eq(want[k], got1[p]) || !eq(want[k], got2[p])) throw std::runtime_error("FK20 proofs map: compiled program differs from its definition");
        }
    }
    SlpProgram& P = slp_prog_[id];
    if (id == SLP_TUNED && slp_prog_[SLP_TUNED_FUSED].ready) {  // same plan, same constants
        P.d_naf = slp_prog_[SLP_TUNED_FUSED].d_naf;
    } else {
        Fr lam = zero<FrParams>();
        for (int i = 0; i < 4; i++) lam.v[i] = (uint32_t)(GLV_LAMBDA >> (32 * i));
        const Fr lm = to_mont(lam);
        constexpr int TWW = launch::TWIDDLE_WORDS;
        std::vector<uint32_t> naf(plan.consts.size() * 2 * TWW);
        for (size_t c = 0; c < plan.consts.size(); c++) recode_glv_wnaf(plan.consts[c], lm, &naf[c * 2 * TWW]);
        HIPCK(hipMalloc(&P.d_naf, naf.size() * 4));
        P.owns_naf = true;
        HIPCK(hipMemcpy(P.d_naf, naf.data(), naf.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCK(hipMalloc(&P.d_words, sched.words.size() * 4));
    HIPCK(hipMemcpy(P.d_words, sched.words.data(), sched.words.size() * 4, hipMemcpyHostToDevice));
    P.launches.clear();
    for (auto& L : sched.launches) P.launches.push_back(SlpLaunch{(int)L.kind, L.first, L.count});
    P.n_slots = sched.n_slots;
    P.info[0] = (int)plan.count(linmap::OP_MULC);
    P.info[1] = (int)(plan.count(linmap::OP_ADD) + plan.count(linmap::OP_SUB));
    P.info[2] = (int)plan.doublings();
    P.info[3] = (int)sched.launches.size();
    if (verbose) fprintf(stderr, "[linmap] program %d: %d multiplications, %d additions, %d doublings, %d launches, %d slots\n", id, P.info[0], P.info[1], P.info[2], P.info[3], P.n_slots);
    P.ready = true;
}
const Engine::SlpProgram& Engine::slp_program(int id) {
    std::lock_guard<std::mutex> lk(slp_build_mu_);
    if (!slp_prog_[id].ready) build_slp_program(id);
    return slp_prog_[id];
}
// Which compilation of the map serves a batch of `lanes` (a multiple of 64).  A constant multiplication is one wave per 64-blob
// lane group and lasts 1.3 ms with a SIMD to itself, 2.3 ms when two share one (the chip has wave_slots_ / 2 SIMDs), and from
// two waves per SIMD on the launch is throughput (1.2 us per wave): so the multiplication count M is chosen so that M x groups
// stays within the SIMDs (<= 2 groups) or the wave slots (<= 5 groups), and within that the program with the shallowest cheap
// levels wins (tools/linmap_explore.cpp).  Measured per group count on one box (profiles/r4_linmap_programs.log; g1_linmap ms):
//   blobs           32    64    128   192   256   320   384   448   512   768   1024  1536
//   350 (tuned)    2.50  2.55  2.71  3.82  3.87  3.98  5.14  5.19  5.36  7.72  8.98* 13.2*      (* with fused a + b / a - b pairs)
//   712 Karatsuba  1.70  1.77  3.04  4.29  4.43  5.56  6.91        8.30
//   456            1.92  1.98  2.14  3.28  3.38  4.60  4.81  5.85  6.07  8.59  10.9
//   606            1.78  1.85  3.08  3.20  4.43  4.56  5.83        7.16
//   372            2.21  2.22  2.35  3.54  3.61  3.72  4.96  4.95  5.16  7.64  9.05  13.4
int Engine::pick_slp_program(int lanes) const {
    if (slp_force_ >= 0) return slp_force_;
    const int groups = lanes / 64, simds = wave_slots_ / 2;
    if (712 * groups <= simds) return SLP_KARATSUBA;
    if (456 * groups <= simds) return SLP_DEPTH_456;
    if (606 * groups <= wave_slots_) return SLP_DEPTH_606;
    if (456 * groups <= wave_slots_) return SLP_DEPTH_456;
    if (372 * groups <= wave_slots_) return SLP_DEPTH_372;
    if (456 * groups <= wave_slots_ * 3 / 2) return SLP_DEPTH_456;  // 6 groups: 1.34 rounds of shorter levels beat 1.03 rounds of the tuned program's
    if (groups <= 12) return SLP_DEPTH_372;
    return lanes >= slp_fuse_min_ ? SLP_TUNED_FUSED : SLP_TUNED;
}
void Engine::init_linmap(const Fr8* w8192_mont) {
    static_assert(sizeof(Fr8) == sizeof(Fr), "layout");
    if (const char* e = getenv("ETH_KZG_AMD_G1FFT")) {  // tuning knob: "radix2" keeps the butterfly network of k_g1fft.hip
        if (!strcmp(e, "radix2")) { use_linmap_ = false; return; }
    }
    if (const char* e = getenv("ETH_KZG_AMD_SLP_PROGRAM")) {  // tuning knob / tests: one compilation of the map at every batch size
        const int v = atoi(e);
        if (v >= 0 && v < SLP_COUNT) slp_force_ = v;
    }
    w128_.resize(128);
    for (int e = 0; e < 128; e++) w128_[e] = w8192_mont[64 * e];
    build_slp_program(SLP_TUNED_FUSED);
    build_slp_program(SLP_TUNED);
    const SlpProgram& fused = slp_prog_[SLP_TUNED_FUSED];
    {   // the two schedules of the tuned plan must agree (the arena is sized per program; inputs and outputs sit at the same slots in all)
        if (fused.info[0] != slp_prog_[SLP_TUNED].info[0]) throw std::runtime_error("FK20 proofs map: the two schedules differ");
    }
    // phases for the ticket walker (k_g1slp.hip: k_slp_walk): every maximal run of cheap launches becomes ONE launch
    {
        const auto& launches = fused.launches;
        std::vector<int> lf, lc;
        slp_phases_.clear();
        for (size_t i = 0; i < launches.size();) {
            if (launches[i].kind == (int)linmap::OP_MULC) { slp_phases_.push_back(SlpPhase{(int)i, -1, 0, 0, 0}); i++; continue; }
            SlpPhase ph{(int)i, (int)lf.size(), 0, 0, 0};
            while (i < launches.size() && launches[i].kind != (int)linmap::OP_MULC) {
                lf.push_back(launches[i].first);
                lc.push_back(launches[i].count);
                ph.n_levels++;
                ph.max_count = std::max(ph.max_count, launches[i].count);
                ph.total_ops += launches[i].count;
                i++;
            }
            slp_phases_.push_back(ph);
        }
        slp_max_levels_ = 0;
        for (auto& ph : slp_phases_) slp_max_levels_ = std::max(slp_max_levels_, ph.n_levels);
        HIPCK(hipMalloc(&d_slp_levels_, (lf.size() * 2 + 2) * sizeof(int)));
        HIPCK(hipMemcpy(d_slp_levels_, lf.data(), lf.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy((int*)d_slp_levels_ + lf.size(), lc.data(), lc.size() * sizeof(int), hipMemcpyHostToDevice));
        slp_level_total_ = (int)lf.size();
        // Measured (MI355X, profiles/r3_slp_walk_ab.log): the walker LOSES to one launch per level at every batch size -- g1_linmap
        // 2.75 -> 2.89 ms at 64 blobs, 5.60 -> 7.27 at 512, 18.1 -> 23.0 at 2048 -- because every operation pays an agent-scope
        // release (buffer_wbl2: the XCD's L2 writes its dirty lines back) and an acquire (buffer_inv: the CU's L1 goes cold), which
        // cost more than the part-empty rounds at the 41 level boundaries they remove.  Kept as an option (and parity-tested);
        // the default stays one launch per dependency level.
        slp_walk_ = false;
        if (const char* e = getenv("ETH_KZG_AMD_SLP_WALK")) slp_walk_ = atoi(e) != 0;
    }
    for (int i = 0; i < 3; i++) slp_info_[i] = fused.info[i];
    slp_info_[3] = slp_walk_ ? (int)slp_phases_.size() : fused.info[3];
    const Fr h = inv(fr_from_u64(2));
    memcpy(&half_, &h, 32);
    use_linmap_ = true;
}

void Engine::init_srs() {
    // embedded trusted setup: "KZGSRS01" | n_g1 | n_g2 | g1 monomial (48 B each) | g2 monomial (96 B each)
    const unsigned char* p = kzg_srs_begin;
    size_t len = (size_t)(kzg_srs_end - kzg_srs_begin);
    uint32_t n1, n2;
    if (len < 16 || memcmp(p, "KZGSRS01", 8)) throw std::runtime_error("bad embedded SRS");
    memcpy(&n1, p + 8, 4);
    memcpy(&n2, p + 12, 4);
    if (n1 != (uint32_t)N_BLOB || len != 16 + (size_t)n1 * 48 + (size_t)n2 * 96) throw std::runtime_error("bad embedded SRS size");
    uint8_t* d_bytes;
    int* d_st;
    HIPCK(hipMalloc(&d_bytes, (size_t)N_BLOB * 48));
    HIPCK(hipMalloc(&d_st, N_BLOB * sizeof(int)));
    HIPCK(hipMalloc(&d_srs_, N_BLOB * sizeof(G1Affine)));
    HIPCK(hipMemcpy(d_bytes, p + 16, (size_t)N_BLOB * 48, hipMemcpyHostToDevice));
    {
        Fp12w nobeta{};
        launch::g1_decompress(d_bytes, d_srs_, d_st, N_BLOB, 0, nobeta, stream_);
    }
    std::vector<int> st(N_BLOB);
    HIPCK(hipMemcpyAsync(st.data(), d_st, N_BLOB * sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCK(hipStreamSynchronize(stream_));
    for (int s : st)
        if (s) throw std::runtime_error("embedded SRS point failed to decompress");
    HIPCK(hipFree(d_bytes));
    HIPCK(hipFree(d_st));
    // beta: the cube root of unity in Fp with (beta x, y) = [lambda](x, y); pick it by testing on [tau]_1
    {
        G1Affine P;
        HIPCK(hipMemcpy(&P, (const G1Affine*)d_srs_ + 1, sizeof(G1Affine), hipMemcpyDeviceToHost));
        uint32_t lam[4];
        for (int i = 0; i < 4; i++) lam[i] = (uint32_t)(GLV_LAMBDA >> (32 * i));
        G1Affine Q = to_affine(scalar_mul<4>(to_jac(P), lam));
        Fp bx = mul(Q.x, inv(P.x));  // beta = x(lambda P) / x(P)
        Fp b3 = mul(sqr(bx), bx);
        if (!eq(b3, one<FpParams>()) || eq(bx, one<FpParams>()) || !eq(Q.y, P.y)) throw std::runtime_error("GLV endomorphism check failed");
        memcpy(&beta_, &bx, 48);
    }
}

// Window tables are immutable once built and depend only on (device, which bases, width), so the contexts of one
// process share them: the second DASContext on a GPU costs neither another 206 GB nor another build
// (the reference's Java test creates several contexts, LibEthKZGTest.java:32).  The last context to go frees the table.
//
// A table is NOT one allocation.  Mapping 200+ GB with one hipMalloc takes the driver seconds during which every other HIP
// call of the process waits (measured in round 3: a 214 GB hipMalloc on the helper thread stalled the caller's launches for
// 4.3 s; the virtual-memory API that would back one address range piece by piece produced GPU memory faults on ROCm 7.0.2 and
// is gone).  Instead the kernels reach a table through a device array of BLOCK pointers -- one block per group for a plain
// table, two for a GLV table (its lower and upper windows; launch::TabBlocks) -- and the blocks live in PIECES of at most
// ~0.85 GB (one 0.8 GB block of the widest table; many blocks of a small one), each its own hipMalloc of a few milliseconds:
//   * other threads' HIP calls slip in between the pieces (tests/test_gpu_tables.py: a caller every 5 ms never waits long),
//   * a build is abandoned within one piece,
//   * the table is usable GROUP BY GROUP while it is built: ready_groups counts the leading groups whose entries are final,
//     an MSM stage runs those on the new table and the rest on the table the context started on (Engine::launch_msm).
struct Engine::SharedTable {
    int dev = 0, kind = 0, c = 0;  // kind: 0 = commitments (plain, monomial SRS as [64][64]), 1 = FK20 plain, 2 = FK20 GLV
    bool glv = false;
    int n_groups = 0, nb = 64, halves = 1;
    size_t bytes = 0;                      // of all blocks
    size_t block_entries[2] = {0, 0};      // entries of a group's block(s)
    std::vector<void*> pieces;
    std::vector<void*> h_blocks;           // host copy of the pointer array (entries of unallocated blocks are null)
    void** d_blocks = nullptr;             // device: [n_groups * halves]
    int blocks_allocated = 0;
    std::atomic<int> ready_groups{0};      // leading groups whose entries are final and whose pointers are on the device
    std::atomic<int> state{0};             // 0 under construction, 1 complete, 2 abandoned (cancelled / out of memory): what is ready stays usable
    std::string why;                       // of state 2
    size_t entry_bytes() const { return glv ? launch::SIZEOF_TABP : launch::SIZEOF_TABQ; }
    size_t block_bytes(int b) const { return block_entries[b % halves] * entry_bytes(); }
    void shape(int device, int kind_, int width, int groups) {
        dev = device; kind = kind_; c = width; glv = kind_ == 2; n_groups = groups;
        halves = glv ? 2 : 1;
        if (glv) {
            const size_t per_window = (size_t)nb << (c - 1);
            block_entries[0] = per_window * launch::glv_lower_windows(c);
            block_entries[1] = per_window * (launch::glv_windows(c) - launch::glv_lower_windows(c));
        } else {
            block_entries[0] = launch::table_entries(c, 1, nb);
        }
        bytes = 0;
        for (int b = 0; b < halves; b++) bytes += block_bytes(b) * (size_t)n_groups;
        h_blocks.assign((size_t)n_groups * halves, nullptr);
        pieces.reserve((size_t)n_groups * halves);  // never reallocated: table_build_info reads its size from other threads while the builder appends
    }
    double alloc_ms = 0, alloc_ms_max = 0;  // time spent in hipMalloc for the pieces: total and the longest single call (trace)
    // allocate pieces until blocks [0, block_end) exist; false: out of memory (why is set) or cancelled
    bool alloc_until(int block_end, const std::atomic<bool>* cancel) {
        // (ETH_KZG_AMD_TABLE_PIECE_MB: experiments only -- one piece for the whole table is round 3's single hipMalloc)
        static const size_t PIECE = [] { const char* e = getenv("ETH_KZG_AMD_TABLE_PIECE_MB"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 850) << 20; }();
        const int total = n_groups * halves;
        if (block_end > total) block_end = total;
        if (!d_blocks) {
            if (hipMalloc((void**)&d_blocks, (size_t)total * sizeof(void*)) != hipSuccess) { (void)hipGetLastError(); d_blocks = nullptr; why = "hipMalloc of the block pointer array failed"; return false; }
            (void)hipMemset(d_blocks, 0, (size_t)total * sizeof(void*));
        }
        while (blocks_allocated < block_end) {
            if (cancel && cancel->load()) { why = "cancelled"; return false; }
            int n = 0;
            size_t sz = 0;
            while (blocks_allocated + n < total && (n == 0 || sz + block_bytes(blocks_allocated + n) <= PIECE)) { sz += block_bytes(blocks_allocated + n); n++; }
            void* p = nullptr;
            const auto a0 = std::chrono::steady_clock::now();
            // (ETH_KZG_AMD_TABLE_CONTIGUOUS=1: experiments -- physically contiguous pieces, in the hope of larger page-table fragments)
            static const bool contiguous = [] { const char* e = getenv("ETH_KZG_AMD_TABLE_CONTIGUOUS"); return e && atoi(e) != 0; }();
            const hipError_t e = contiguous ? hipExtMallocWithFlags(&p, sz, hipDeviceMallocContiguous) : hipMalloc(&p, sz);
            const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
            alloc_ms += dt;
            alloc_ms_max = std::max(alloc_ms_max, dt);
            if (dt > 100 && getenv("ETH_KZG_AMD_TRACE")) fprintf(stderr, "[context] @%.0f ms: hipMalloc of table piece %zu (%.2f GB) took %.0f ms\n", trace_clock_ms(), pieces.size(), sz / 1e9, dt);
            if (e != hipSuccess) { (void)hipGetLastError(); why = std::string("hipMalloc of a table piece: ") + hipGetErrorString(e); return false; }
            pieces.push_back(p);
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
struct BuildCancelled {};  // thrown out of a table build when its context (or the process) is going away
static std::mutex g_tables_mu;  // the registry below: held for look-ups and inserts only, never across a build
static std::map<std::tuple<int, int, int>, std::weak_ptr<Engine::SharedTable>> g_tables;  // (device, kind, width)
static std::mutex g_build_mu;   // one builder of WIDE tables at a time per process (the helper threads of several contexts queue here)

static size_t plain_table_bytes(int c, int n_groups) { return launch::table_entries(c, n_groups, 64) * launch::SIZEOF_TABQ; }
static size_t glv_table_bytes(int c) { return launch::table_glv_entries(c, 128, 64) * launch::SIZEOF_TABP; }

// Fill a table the caller has just created: pieces are allocated a chunk of groups ahead of the builder kernels, every
// finished chunk is published through ready_groups.  Returns false (state 2, `why` set) if the device cannot hold it;
// throws BuildCancelled when `cancel` is raised (state 2 as well).  The groups that are ready stay usable either way.
// gentle: the build shares the GPU with callers on the start tables (progressive start): one group per launch -- 512 waves, one
// per SIMD on half the chip's SIMDs, so a caller's kernels find free SIMDs at once instead of waiting for 1,500 builder
// waves that run 17 ms -- at twice the build time, which the allocation of the pieces hides anyway.
static bool fill_table(Engine::SharedTable& t, const void* bases, hipStream_t st, const std::atomic<bool>* cancel, bool gentle = false) {
    const bool trace = getenv("ETH_KZG_AMD_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int c = t.c, nb = t.nb;
    const bool fast = t.glv || c >= 10;  // plain width 8 (128 entries per window = two wave steps) is quicker with the simple builder
    const size_t per_group = t.glv ? launch::table_glv_entries(c, 1, nb) : launch::table_entries(c, 1, nb);
    const size_t scratch_per_entry = t.glv ? 168 : fast ? 56 : sizeof(G1Jac);
    int chunk = (int)((t.glv || fast ? (9ull << 30) : (24ull << 30)) / (per_group * scratch_per_entry));
    if (chunk < 1) chunk = 1;
    if (chunk > t.n_groups) chunk = t.n_groups;
    if (gentle) {  // <= ~512 builder waves in flight (a wave per (base, window): 64 x W per group)
        const int W = t.glv ? launch::glv_windows(c) : (255 + c) / c;
        chunk = std::max(1, std::min(chunk, 512 / (nb * W)));
    }
    const size_t side_bytes = t.glv ? launch::table_glv_side_bytes(c, chunk, nb) : fast ? launch::table_fast_side_bytes(c, chunk, nb) : 0;
    void *scratch = nullptr, *side = nullptr;
    int* d_err = nullptr;
    auto cleanup = [&] {
        (void)hipStreamSynchronize(st);
        if (scratch) (void)hipFree(scratch);
        if (side) (void)hipFree(side);
        if (d_err) (void)hipFree(d_err);
        scratch = side = nullptr;
        d_err = nullptr;
    };
    auto give_up = [&](const std::string& why) {
        cleanup();
        t.why = why;
        t.state.store(2);
        return false;
    };
    size_t free_b = 0, total_b = 0;
    HIPCK(hipMemGetInfo(&free_b, &total_b));
    const size_t need = t.bytes + per_group * chunk * scratch_per_entry + side_bytes + (8ull << 30);  // + head-room for batches
    if (need > free_b) return give_up("not enough free device memory");
    if (hipMalloc(&scratch, per_group * chunk * scratch_per_entry) != hipSuccess || (side_bytes && hipMalloc(&side, side_bytes) != hipSuccess) ||
        hipMalloc(&d_err, sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        return give_up("hipMalloc of the builder's scratch failed");
    }
    if (trace) fprintf(stderr, "[context] @%.0f ms:  table kind %d width %d: %.1f GB in pieces, scratch %.1f GB allocated  %8.1f ms\n", trace_clock_ms(), t.kind, c, t.bytes / 1e9,
                       (per_group * chunk * scratch_per_entry + side_bytes) / 1e9, ms());
    try {
        HIPCK(hipMemsetAsync(d_err, 0, sizeof(int), st));
        for (int g0 = 0; g0 < t.n_groups; g0 += chunk) {
            const int g = std::min(chunk, t.n_groups - g0);
            if (cancel && cancel->load()) throw BuildCancelled{};
            // the pieces of this chunk are allocated while the previous chunk's kernels still run
            if (!t.alloc_until((g0 + g) * t.halves, cancel)) {
                if (cancel && cancel->load()) throw BuildCancelled{};
                if (trace) fprintf(stderr, "[context]   table kind %d width %d: stopped at group %d of %d (%s)\n", t.kind, c, g0, t.n_groups, t.why.c_str());
                return give_up(t.why);
            }
            HIPCK(hipStreamSynchronize(st));  // the previous chunk has left the scratch: its groups are final
            t.ready_groups.store(g0, std::memory_order_release);
            HIPCK(hipMemcpyAsync(t.d_blocks + (size_t)g0 * t.halves, t.h_blocks.data() + (size_t)g0 * t.halves, (size_t)g * t.halves * sizeof(void*),
                                 hipMemcpyHostToDevice, st));
            const char* b = (const char*)bases + (size_t)g0 * nb * sizeof(G1Affine);
            void* const* blocks = t.d_blocks + (size_t)g0 * t.halves;
            if (t.glv) {
                if (!launch::build_table_glv(c, b, blocks, scratch, side, g, nb, d_err, st)) throw std::runtime_error("GLV table width not built in");
            } else if (fast) {
                if (!launch::build_table_fast(c, b, blocks, scratch, side, g, nb, d_err, st)) throw std::runtime_error("table width not built in");
            } else {
                launch::build_table(c, b, blocks, scratch, g, nb, st);
            }
        }
        HIPCK(hipStreamSynchronize(st));
        int err = 0;
        HIPCK(hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost));
        if (err) throw std::runtime_error("window table: a base point of small order");
    } catch (const BuildCancelled&) {
        give_up("cancelled: the context is being freed");
        throw;
    } catch (const std::exception& e) {
        give_up(e.what());
        throw;
    }
    cleanup();
    t.ready_groups.store(t.n_groups, std::memory_order_release);
    t.state.store(1);
    if (trace) fprintf(stderr, "[context]   table kind %d width %d: built            %8.1f ms (%zu pieces: hipMalloc %.1f ms in all, longest %.1f ms)\n", t.kind, c, ms(),
                       t.pieces.size(), t.alloc_ms, t.alloc_ms_max);
    return true;
}

// the live table of (device, kind, width) in the registry, whatever its state (null: none, or only an abandoned one)
static std::shared_ptr<Engine::SharedTable> find_table(int dev, int kind, int w) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    auto it = g_tables.find(std::make_tuple(dev, kind, w));
    if (it == g_tables.end()) return nullptr;
    auto t = it->second.lock();
    if (!t || t->state.load() == 2) return nullptr;
    return t;
}
// find_table, or a new (empty, state 0) table registered under the key; *created tells which
static std::shared_ptr<Engine::SharedTable> find_or_create_table(int dev, int kind, int w, int n_groups, bool* created) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    auto& slot = g_tables[std::make_tuple(dev, kind, w)];
    auto t = slot.lock();
    *created = false;
    if (t && t->state.load() != 2) return t;
    t = std::make_shared<Engine::SharedTable>();
    t->shape(dev, kind, w, n_groups);
    slot = t;
    *created = true;
    return t;
}
// a COMPLETE table of (device, kind, width): found, awaited (another thread is building it) or built here; null if it does not fit
static std::shared_ptr<Engine::SharedTable> obtain_table(int dev, int kind, int w, const void* bases, int n_groups, hipStream_t st,
                                                         bool only_if_live = false, const std::atomic<bool>* cancel = nullptr) {
    if (only_if_live) {
        auto t = find_table(dev, kind, w);
        return t && t->state.load() == 1 ? t : nullptr;
    }
    if (cancel && cancel->load()) throw BuildCancelled{};
    bool created = false;
    auto t = find_or_create_table(dev, kind, w, n_groups, &created);
    if (created) return fill_table(*t, bases, st, cancel) ? t : nullptr;
    while (t->state.load() == 0) {  // another context's thread is building it
        if (cancel && cancel->load()) throw BuildCancelled{};
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return t->state.load() == 1 ? t : nullptr;
}

Engine::TableView Engine::table_view(TableSel which) const {
    if (primary_) return primary_->table_view(which);  // an engine lane reads through to the context's engine
    std::lock_guard<std::mutex> lk(tab_mu_);
    return views_[which];
}
// main: the complete table calls run on; next: a wider one under construction whose ready groups are used already
void Engine::publish(TableSel which, const std::shared_ptr<SharedTable>& main, const std::shared_ptr<SharedTable>& next) {
    std::lock_guard<std::mutex> lk(tab_mu_);
    TableView& v = views_[which];
    if (v.main && v.main != main) retired_.push_back(v.main);  // kernels in flight may still read it
    if (v.next && v.next != next && v.next != main) retired_.push_back(v.next);
    v.main = main;
    v.next = next;
    v.c = main ? main->c : 0;
    v.glv = main ? main->glv : false;
    v.bytes = main ? main->bytes : 0;
}
int Engine::tables_ready(int wait_ms) {
    if (primary_) return const_cast<Engine*>(primary_)->tables_ready(wait_ms);
    std::unique_lock<std::mutex> lk(tab_mu_);
    if (wait_ms < 0) tab_cv_.wait(lk, [&] { return tables_state_ != 0; });
    else if (wait_ms > 0) tab_cv_.wait_for(lk, std::chrono::milliseconds(wait_ms), [&] { return tables_state_ != 0; });
    return tables_state_;
}
int Engine::table_groups_ready(TableSel which) const {
    if (primary_) return primary_->table_groups_ready(which);
    const TableView v = table_view(which);
    if (v.next) return v.next->ready_groups.load(std::memory_order_acquire);
    std::lock_guard<std::mutex> lk(tab_mu_);
    return tables_state_ != 0 && v.main ? v.main->n_groups : 0;  // nothing wider under construction yet (or ever): 0 until the builder is done
}

void Engine::table_build_info(double* out4) const {
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (TableSel sel : {TAB_FK, TAB_SRS}) {
        const TableView v = table_view(sel);
        for (const SharedTable* t : {v.main.get(), v.next.get()}) {
            if (!t) continue;
            out4[0] += t->alloc_ms;
            out4[1] = std::max(out4[1], t->alloc_ms_max);
            out4[2] += (double)t->pieces.size();
            out4[3] += (double)t->bytes;
        }
    }
}

void Engine::init_fk20() {
    // 64 G1-FFT_128 of the SRS vectors: the 64 vectors ride on the 64 lanes of the FFT kernel.
    void* X;
    HIPCK(hipMalloc(&X, 128 * 64 * launch::SIZEOF_JACQ));
    HIPCK(hipMalloc(&d_fk_bases_, 128 * 64 * sizeof(G1Affine)));
    launch::fk20_srs_vectors(d_srs_, X, stream_);
    g1_fft128_full(X, 64, /*inverse=*/0, stream_);  // DIF: natural in, bit-reversed out
    launch::fk20_gather_bases(X, d_fk_bases_, stream_);
    HIPCK(hipStreamSynchronize(stream_));
    HIPCK(hipFree(X));
    if (primary_) return;  // an engine lane: the tables are the context's (table_view reads through)
    if (!use_precomp_) {  // UsePrecomp::No: the 0.8 GB width-4 tables, nothing else to build
        auto srs = obtain_table(dev_, 0, 4, d_srs_, 64, stream_), fk = obtain_table(dev_, 1, 4, d_fk_bases_, 128, stream_);
        if (!srs || !fk) throw std::runtime_error("not enough device memory for the window tables");
        publish(TAB_SRS, srs, nullptr);
        publish(TAB_FK, fk, nullptr);
        std::lock_guard<std::mutex> lk2(tab_mu_);
        tables_state_ = 1;
        return;
    }
    bool progressive = true;
    if (const char* e = getenv("ETH_KZG_AMD_PROGRESSIVE")) progressive = atoi(e) != 0;
    if (!progressive) {
        build_final_tables();
        if (!table_view(TAB_FK).main || !table_view(TAB_SRS).main) throw std::runtime_error("not enough device memory for the window tables: " + tables_error_);
        return;
    }
    // Progressive start (the reference's "Initialize context" bench, benchmark-mt.rs:103-113): serve from small tables at once --
    // or from whatever wider table another context of this process already holds -- and build the wide ones on a helper thread.
    {
        std::shared_ptr<SharedTable> fk, srs;
        if (!want_plain_c_)
            for (int w : launch::GLV_WIDTHS)
                if (!fk) fk = obtain_table(dev_, 2, w, nullptr, 128, stream_, /*only_if_live=*/true);
        if (!fk) fk = obtain_table(dev_, 2, 8, d_fk_bases_, 128, stream_);
        for (int w : {13, 12, 10, 8})
            if (!srs) srs = obtain_table(dev_, 0, w, nullptr, 64, stream_, true);
        if (!srs) srs = obtain_table(dev_, 0, 8, d_srs_, 64, stream_);
        if (!fk || !srs) throw std::runtime_error("not enough device memory for the start window tables");
        publish(TAB_FK, fk, nullptr);
        publish(TAB_SRS, srs, nullptr);
    }
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        static bool registered = false;
        if (!registered) { atexit(stop_all_builders_at_exit); registered = true; }
        g_engines.push_back(this);
    }
    progressive_build_ = true;
    builder_ = std::thread([this] {
        (void)hipSetDevice(dev_);
        build_final_tables();
    });
}

// The wide tables, widest first, each taken from the process-wide registry if another context of this GPU holds it already:
//   commitments: plain width 13 (43 GB; 20 windows), 12, 10, 8
//   FK20: GLV width 16 (206 GB; 16 gathered additions per base), 15 (116 GB; 18), 14 (64 GB; 20), 12 (18 GB; 22), 8 (1.6 GB; 32)
//         -- GLV first at every size: the endomorphism halves the memory per window bit (a plain width-14 table costs
//         163 GB for 19 additions) -- or the plain width ETH_KZG_AMD_WINDOW names;
// bounded by what the HBM still holds (another process may own part of it) and by ETH_KZG_AMD_TABLE_GB (both tables
// together; the commitment table gets at most 18 % of it).  A table this thread creates is published as the view's `next`
// BEFORE it is filled, so the MSMs use its groups as they become ready.  Never throws: a failure leaves the context on the
// tables it has.
void Engine::build_final_tables() {
    int state = 1;
    std::string why;
    // wider than `now`, from the registry or built here; attach = publish as `next` while it is filled
    // A table this context had attached as `next` may have been ABANDONED by the context that was filling it (freed mid-build):
    // its pieces must go before another 206 GB can be allocated.  Nothing refers to it once the view is republished, except
    // kernels already in flight and a caller that took its snapshot a moment ago: wait for both, then let go of it.
    auto drop_abandoned = [&](TableSel sel) {
        const TableView cur = table_view(sel);
        if (!cur.next || cur.next->state.load() != 2) return;
        publish(sel, cur.main, nullptr);
        (void)hipDeviceSynchronize();
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(tab_mu_);
        retired_.erase(std::remove_if(retired_.begin(), retired_.end(), [](const std::shared_ptr<SharedTable>& r) { return r->state.load() == 2; }), retired_.end());
    };
    auto widen = [&](TableSel sel, int kind, int w, const void* bases, int n_groups) -> std::shared_ptr<SharedTable> {
        if (cancel_build_.load()) throw BuildCancelled{};
        drop_abandoned(sel);
        bool created = false;
        auto t = find_or_create_table(dev_, kind, w, n_groups, &created);
        const TableView cur = table_view(sel);
        const bool same_form = cur.main && cur.main->glv == t->glv && cur.main->n_groups == t->n_groups;
        if (t->state.load() == 0 && same_form) publish(sel, cur.main, t);  // its ready groups serve at once
        if (created) {
            if (!fill_table(*t, bases, build_stream_, &cancel_build_, /*gentle=*/progressive_build_)) {
                if (t->ready_groups.load() == 0) publish(sel, cur.main, nullptr);
                return nullptr;
            }
        } else {
            while (t->state.load() == 0) {  // another context's helper thread is filling it
                if (cancel_build_.load()) throw BuildCancelled{};
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
            if (t->state.load() != 1) return nullptr;
        }
        return t;
    };
    try {
        const double p0 = trace_clock_ms();
        launch::preload_code_objects();  // before the first piece is allocated: no caller's first launch of a kernel waits behind a hipMalloc
        if (getenv("ETH_KZG_AMD_TRACE")) fprintf(stderr, "[context] @%.0f ms: code objects preloaded in %.0f ms\n", trace_clock_ms(), trace_clock_ms() - p0);
        // Another context of the process may be filling the wide tables right now (its helper thread holds g_build_mu until it is
        // done): this context uses their ready groups meanwhile instead of sitting on its start tables for the other's build.
        auto attach_growing = [&] {
            for (TableSel sel : {TAB_SRS, TAB_FK}) {
                const TableView cur = table_view(sel);
                if (!cur.main || (cur.next && cur.next->state.load() == 0)) continue;
                std::shared_ptr<SharedTable> growing;
                if (sel == TAB_SRS) {
                    for (int w : {13, 12, 10})
                        if (!growing && w > cur.c) { auto t = find_table(dev_, 0, w); if (t && t->state.load() == 0) growing = t; }
                } else if (!want_plain_c_) {
                    for (int w : launch::GLV_WIDTHS)
                        if (!growing && (!cur.glv || w > cur.c)) { auto t = find_table(dev_, 2, w); if (t && t->state.load() == 0) growing = t; }
                }
                if (growing && growing->glv == cur.main->glv && growing->n_groups == cur.main->n_groups) publish(sel, cur.main, growing);
            }
        };
        std::unique_lock<std::mutex> lk(g_build_mu, std::defer_lock);  // one builder of wide tables at a time per process
        while (!lk.try_lock()) {
            attach_growing();
            if (cancel_build_.load()) throw BuildCancelled{};
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
        if (cancel_build_.load()) throw BuildCancelled{};
        const double budget = table_budget_gb_ > 0 ? table_budget_gb_ * 1e9 : 1e18;
        const TableView srs_now = table_view(TAB_SRS), fk_now = table_view(TAB_FK);
        std::shared_ptr<SharedTable> srs;
        for (int w : {13, 12, 10, 8}) {
            if (srs) break;
            if (srs_now.main && w <= srs_now.c) break;  // nothing wider than what is in use fits
            if ((double)plain_table_bytes(w, 64) > std::max(0.18 * budget, 2.2e9)) continue;
            srs = widen(TAB_SRS, 0, w, d_srs_, 64);
        }
        if (srs) publish(TAB_SRS, srs, nullptr);
        const double left = budget - (double)table_view(TAB_SRS).bytes;
        std::shared_ptr<SharedTable> fk;
        if (want_plain_c_) {
            static const int widths[] = {14, 13, 12, 10, 8};
            for (int w : widths) {
                if (fk || w > want_plain_c_) continue;
                fk = widen(TAB_FK, 1, w, d_fk_bases_, 128);
            }
        } else {
            for (int w : launch::GLV_WIDTHS) {
                if (fk) break;
                if (want_glv_c_ && w != want_glv_c_) continue;
                if (fk_now.main && fk_now.glv && w <= fk_now.c) break;
                if ((double)glv_table_bytes(w) > std::max(left, 1.7e9)) continue;
                fk = widen(TAB_FK, 2, w, d_fk_bases_, 128);
            }
        }
        if (fk) publish(TAB_FK, fk, nullptr);
        if (!table_view(TAB_FK).main) {  // not even the narrowest GLV table fits: the plain width-4 tables (0.8 GB)
            auto f4 = obtain_table(dev_, 1, 4, d_fk_bases_, 128, build_stream_);
            if (f4) publish(TAB_FK, f4, nullptr);
        }
        if (!table_view(TAB_SRS).main) {
            auto s4 = obtain_table(dev_, 0, 4, d_srs_, 64, build_stream_);
            if (s4) publish(TAB_SRS, s4, nullptr);
        }
    } catch (const BuildCancelled&) {
        state = 2;
        why = "cancelled: the context is being freed";
    } catch (const std::exception& e) {
        (void)hipGetLastError();
        state = 2;
        why = e.what();
    }
    if (state == 2 && getenv("ETH_KZG_AMD_TRACE")) fprintf(stderr, "[context] wide tables not built: %s\n", why.c_str());
    std::lock_guard<std::mutex> lk(tab_mu_);
    tables_state_ = state;
    tables_error_ = why;
    tab_cv_.notify_all();
}

// Per-stage HIP events (bench.py's roofline leg).  Meant for one caller at a time: marks of concurrent calls would interleave.
void Engine::set_profiling(bool on) { std::lock_guard<std::mutex> lk(marks_mu_); profiling_ = on; }
// Marks are paired by the index mark_begin returns: concurrent prover calls (work sets 1..3, side streams) each close their
// own mark instead of "the last one pushed" (ADVICE r2).
int Engine::mark_begin(int stage, hipStream_t st) {
    if (!profiling_) return -1;
    std::lock_guard<std::mutex> lk(marks_mu_);
    StageMark m{stage, 0, nullptr, nullptr};
    HIPCK(hipEventCreate(&m.a));
    HIPCK(hipEventCreate(&m.b));
    HIPCK(hipEventRecord(m.a, st));
    marks_.push_back(m);
    return (int)((marks_gen_ & 0x7ff) << 20) | ((int)marks_.size() - 1);  // generation: get_stage_times may clear the list under a running call
}
void Engine::mark_end(int mark, int launches, hipStream_t st) {
    if (mark < 0) return;
    std::lock_guard<std::mutex> lk(marks_mu_);
    const int idx = mark & 0xfffff;
    if (((mark >> 20) & 0x7ff) != (int)(marks_gen_ & 0x7ff) || idx >= (int)marks_.size()) return;  // collected in between
    marks_[idx].launches = launches;
    HIPCK(hipEventRecord(marks_[idx].b, st));
}
void Engine::get_stage_times(double* ms, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(marks_mu_);
    for (int i = 0; i < ST_COUNT; i++) { ms[i] = 0; launches[i] = 0; }
    hipSetDevice(dev_);
    hipDeviceSynchronize();
    for (auto& m : marks_) {
        float t = 0;
        if (hipEventElapsedTime(&t, m.a, m.b) == hipSuccess) { ms[m.stage] += t; launches[m.stage] += m.launches; }
        hipEventDestroy(m.a);
        hipEventDestroy(m.b);
    }
    marks_.clear();
    marks_gen_++;
}

void Engine::ensure_workspace(int n) {
    ensure_workspace(work_[0], n);
    HIPCK(hipStreamWaitEvent(stream_, work_[0].done, 0));
}
void Engine::ensure_workspace(Work& w, int n) {
    if (n <= w.cap) return;
    int cap = ((n + 63) / 64) * 64;
    void** ptrs[] = {&w.coeffs, &w.canon, &w.scalars, &w.X, (void**)&w.status};
    for (void** p : ptrs)
        if (*p) { HIPCK(hipFree(*p)); *p = nullptr; }
    w.cap = 0;
    HIPCK(hipMalloc(&w.coeffs, (size_t)cap * N_BLOB * sizeof(Fr)));
    if (&w == &work_[0]) HIPCK(hipMalloc(&w.canon, (size_t)cap * N_BLOB * sizeof(Fr)));  // canonical coefficients: commitment / EIP-4844 paths only
    HIPCK(hipMalloc(&w.scalars, (size_t)cap * 128 * 64 * sizeof(Fr)));
    HIPCK(hipMalloc(&w.X, (size_t)cap * 128 * launch::SIZEOF_JACQ));
    HIPCK(hipMalloc(&w.status, (size_t)cap * sizeof(int)));
    w.cap = cap;
}
// staging of the host-pointer prover entry point: one chunk of blobs in, its cells and proofs out (device + pinned host)
void Engine::ensure_staging(Work& w, int n) {
    if (n <= w.stage_cap) return;
    void* dev[] = {w.d_in, w.d_cells, w.d_proofs};
    for (void* p : dev)
        if (p) HIPCK(hipFree(p));
    void* pin[] = {w.h_in, w.h_cells, w.h_proofs, w.h_status};
    for (void* p : pin)
        if (p) HIPCK(hipHostFree(p));
    w.d_in = w.d_cells = w.d_proofs = w.h_in = w.h_cells = w.h_proofs = nullptr;
    w.h_status = nullptr;
    w.stage_cap = 0;
    HIPCK(hipMalloc(&w.d_in, (size_t)n * BYTES_PER_BLOB));
    HIPCK(hipMalloc(&w.d_cells, (size_t)n * N_CELLS * BYTES_PER_CELL));
    HIPCK(hipMalloc(&w.d_proofs, (size_t)n * N_CELLS * 48));
    HIPCK(hipHostMalloc(&w.h_in, (size_t)n * BYTES_PER_BLOB, hipHostMallocDefault));
    HIPCK(hipHostMalloc(&w.h_cells, (size_t)n * N_CELLS * BYTES_PER_CELL, hipHostMallocDefault));
    HIPCK(hipHostMalloc(&w.h_proofs, (size_t)n * N_CELLS * 48, hipHostMallocDefault));
    HIPCK(hipHostMalloc(&w.h_status, (size_t)n * sizeof(int), hipHostMallocDefault));
    w.stage_cap = n;
}
// a free set among work_[first..last]: the first one whose lock is free, else wait for `first`
Work& Engine::lease_work(int first, int last) {
    // any free set; when all are taken, wait for WHICHEVER frees first (release_work notifies) instead of queueing on one of them
    std::unique_lock<std::mutex> lk(lease_mu_);
    for (;;) {
        for (int i = first; i <= last; i++)
            if (work_[i].mu.try_lock()) return work_[i];
        lease_cv_.wait_for(lk, std::chrono::milliseconds(2));
    }
}
void Engine::release_work(Work& w) {
    w.mu.unlock();
    lease_cv_.notify_one();
}

// ---------------------------------------------------------------------------------------------
void Engine::launch_msm(const void* scalars, TableSel which, void* out, int n_groups, int n_slices, int out_stride,
                        int brp_bits, hipStream_t st) {
    launch_msm(scalars, table_view(which), false, out, n_groups, n_slices, out_stride, brp_bits, st);
}
// tv: ONE snapshot of the table view (the builder thread may publish a wider table at any time); scalars_split: the producer
// has stored the scalars as balanced GLV halves already (only meaningful for a GLV table).  While a wider table is under
// construction its leading ready groups run on it and the rest on the complete table: two launches, one MSM stage.
void Engine::launch_msm(const void* scalars, const TableView& tv, bool scalars_split, void* out, int n_groups, int n_slices,
                        int out_stride, int brp_bits, hipStream_t st, void* partial) {
    const SharedTable* main = tv.main.get();
    const SharedTable* next = tv.next.get();
    int ready = 0;
    if (next && next->glv == main->glv) ready = std::min(n_groups, next->ready_groups.load(std::memory_order_acquire));
    if (main->glv && !scalars_split) launch::glv_split(const_cast<void*>(scalars), (size_t)n_groups * n_slices * 64, st);  // in place: they feed nothing else
    if (ready > 0) launch_msm_range(scalars, *next, 0, ready, out, n_groups, n_slices, out_stride, brp_bits, st, partial);
    if (ready < n_groups) launch_msm_range(scalars, *main, ready, n_groups - ready, out, n_groups, n_slices, out_stride, brp_bits, st, partial);
}
// groups [g0, g0 + gcnt) of every slice on table t
void Engine::launch_msm_range(const void* scalars, const SharedTable& t, int g0, int gcnt, void* out, int n_groups, int n_slices,
                              int out_stride, int brp_bits, hipStream_t st, void* partial) {
    const launch::TabBlocks tb{(const void* const*)t.d_blocks, g0, gcnt};
    const int c = t.c;
    const long msms = (long)gcnt * n_slices;
    if (t.glv) {
        int mode = 1;
        if (n_slices <= FLAT_MSM_MAX_SLICES && circ_max_ > 0) mode = 0;
        else if (msm_chunks_ >= 0) mode = msm_chunks_ == 0 ? 1 : msm_chunks_ == 1 ? 3 : msm_chunks_ == 2 ? 4 : msm_chunks_ == 8 ? 5 : 2;  // tuning knob / tests
        else if ((msms * 4 + 63) / 64 >= (long)wave_slots_) {
            // The chip is full: four chunks per MSM.  Measured on one box, alternating runs (MSM stage, ms; S = lanes per MSM of
            // the lane kernels):        blobs   256   512   768   1024  1536  2048         3072
            //   four chunks                     6.2  11.5  16.8  21.4  31.1  40.5-40.7   59.7
            //   a lane per MSM (S = 1)          9.9  18.5  19.9  22.0  31.3  41.0-42.0   60.9
            //   a lane per GLV half (S = 2)     9.2  11.9  21.4  24.9  34.3  44.9        61.9
            // The lane kernels save the folds and the barriers, and at 2048 / 3072 blobs their waves make exact rounds -- and
            // still lose 1-2 %: 16384 short waves dealt out as slots free up balance the SIMDs better than 4096 long ones.
            mode = 2;
        }
        launch::msm_glv(c, mode, scalars, tb, out, n_groups, n_slices, 64, out_stride, brp_bits, beta_, st);
        return;
    }
    if (n_slices <= FLAT_MSM_MAX_SLICES && circ_max_ > 0) {
        launch::msm_fixed_flat(c, scalars, tb, out, n_groups, n_slices, 64, out_stride, brp_bits, st);
        return;
    }
    // Large batches: threads own a chunk of the windows of an MSM (S = 4 chunks: the fold is two additions per ~300, and
    // the waves are short enough for the tail of a launch not to matter; measured equal or better than S = 1, 2 at every
    // batch that fills the chip).  Below one round of the chip's 2-per-SIMD wave slots the windowed kernel (one thread
    // per window) has more parallelism.
    int S = 0;
    if (msm_chunks_ >= 0) S = msm_chunks_;  // tuning knob ETH_KZG_AMD_MSM_CHUNKS: 0 = windowed kernel, 1/2/4 = chunked
    else if ((msms * 4 + 63) / 64 >= (long)wave_slots_) S = 4;
    if (S) launch::msm_fixed_chunked(c, scalars, tb, out, n_groups, n_slices, 64, out_stride, brp_bits, S, st);
    else launch::msm_fixed(c, scalars, tb, out, n_groups, n_slices, 64, out_stride, brp_bits, st);
}

// inverse FFT_128, DIT, input at bit-reversed positions, only outputs 0..63 produced (domain.rs:172-194;
// the 128^-1 scaling is folded into the MSM scalars).
void Engine::g1_ifft128_take64(void* X, int stride, hipStream_t st) {
    for (int half = 1; half <= 32; half <<= 1)
        launch::g1_fft_layer(X, stride, half, 128 / (2 * half), 1, 0, d_naf_, beta_, st);
    launch::g1_fft_layer(X, stride, 64, 1, 1, 3, d_naf_, beta_, st);
}
// forward FFT_128 of (h || O): DIF, natural in, bit-reversed out = the proof order (prover.rs:214-222).
void Engine::g1_fft128_from64(void* X, int stride, hipStream_t st) {
    launch::g1_fft_layer(X, stride, 64, 1, 0, 2, d_naf_, beta_, st);
    for (int half = 32; half >= 1; half >>= 1)
        launch::g1_fft_layer(X, stride, half, 128 / (2 * half), 0, 1, d_naf_, beta_, st);
}
// full FFT_128.  forward: DIF natural -> bit-reversed.  inverse: DIT bit-reversed -> natural (no scaling).
void Engine::g1_fft128_full(void* X, int stride, int inverse, hipStream_t st) {
    if (!inverse) {
        for (int half = 64; half >= 1; half >>= 1)
            launch::g1_fft_layer(X, stride, half, 128 / (2 * half), 0, 1, d_naf_, beta_, st);
    } else {
        for (int half = 1; half <= 64; half <<= 1)
            launch::g1_fft_layer(X, stride, half, 128 / (2 * half), 1, 0, d_naf_, beta_, st);
    }
}

// stages C..G of SURVEY 3.2 from coefficients already in w.coeffs
// tv_pre: the scalars of all n blobs are in w.scalars already, computed in the form of THIS view (the host-pointer path does it
// sub-batch by sub-batch under the uploads)
void Engine::run_proofs_from_coeffs(Work& w, int n, uint8_t* d_proofs, hipStream_t st, const TableView* tv_pre, ProofsPhase phase, int msm_cut) {
    const int bp = ((n + 63) / 64) * 64;
    // one or two blobs: the MSM also delivers 2^32 u, 2^64 u, 2^96 u (scaled copies of the scalars, same tables), which
    // cuts the doubling chain of the circulant form into four parallel quarters (needs 32 * 4 >= T - 1 doublings)
    const int segs = (n > circ_max_ || circ_T_ > 129) ? 1 : n <= 2 ? 4 : n <= 4 ? 2 : 1;
    const Fr8 two_segments[3] = {seg_shift_[1], seg_shift_[1], seg_shift_[1]};  // 2^64
    // beyond the small-batch circulant kernel the two transforms run as one compiled linear map (g1_linmap.hpp), which wants
    // the MSM outputs halved instead of divided by 128 and in natural Fourier order in the first 128 arena slots
    const bool linmap_mode = use_linmap_ && n > circ_max_;
    void* X = w.X;
    const SlpProgram* prog = nullptr;
    int mulc_coop_lanes = 0;  // > 0: so few blobs that the constant multiplications take several lanes per blob (launch::g1_slp_launch)
    if (linmap_mode) {
        int which = slp_walk_ ? (int)SLP_TUNED_FUSED : pick_slp_program(bp);
        // 33 .. 64 blobs: the constant multiplications run with two lanes per blob = two waves per operation (k_g1slp.hip), so the
        // compilation with 456 of them (912 waves, one per SIMD) replaces the one with 712 that a lane per blob takes
        static const int pair64 = [] { const char* e = getenv("ETH_KZG_AMD_SLP_PAIR64"); return e ? atoi(e) : 1; }();
        if (pair64 && !slp_walk_ && slp_force_ < 0 && bp == 64 && n > 32 && launch::coop_points_max() > 0) which = SLP_DEPTH_456;
        mulc_coop_lanes = n <= 32 ? n : (bp == 64 && which == SLP_DEPTH_456) ? n : 0;
        prog = &slp_program(which);
        const size_t need = (size_t)prog->n_slots * bp * launch::SIZEOF_JACQ;
        if (need > w.slp_arena_bytes) {
            if (w.slp_arena) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(w.slp_arena)); w.slp_arena = nullptr; }
            HIPCK(hipMalloc(&w.slp_arena, need));
            w.slp_arena_bytes = need;
        }
        X = w.slp_arena;
    }
    const TableView tv = tv_pre ? *tv_pre : table_view(TAB_FK);  // one snapshot for the scalars' form AND the MSM that reads them
    // EXPERIMENT (ETH_KZG_AMD_OVERLAP_HALVES=1): the lane groups in two halves, half A's linear map on the work
    // set's second (high-priority) stream next to half B's MSM.  Measured, not adopted: see DESIGN.md section 5.
    static const bool overlap_halves = [] { const char* e = getenv("ETH_KZG_AMD_OVERLAP_HALVES"); return e && atoi(e) != 0; }();
    if (overlap_halves && phase == PROOFS_ALL && linmap_mode && !slp_walk_ && bp >= 128 && w.copy && !profiling_ && segs == 1 && st != w.copy) {
        const int G = bp / 64, lanesA = ((G + 1) / 2) * 64, lanesB = bp - lanesA, nA = std::min(n, lanesA), nB = n - nA;
        const SlpProgram* pA = &slp_program(pick_slp_program(lanesA));
        const SlpProgram* pB = nB > 0 ? &slp_program(pick_slp_program(lanesB)) : nullptr;
        const size_t need = (size_t)std::max(pA->n_slots, pB ? pB->n_slots : 0) * bp * launch::SIZEOF_JACQ;
        if (need > w.slp_arena_bytes) {
            if (w.slp_arena) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipStreamSynchronize(w.copy)); HIPCK(hipFree(w.slp_arena)); w.slp_arena = nullptr; }
            HIPCK(hipMalloc(&w.slp_arena, need));
            w.slp_arena_bytes = need;
        }
        char* A = (char*)w.slp_arena;
        if (!tv_pre) launch::fk20_scalars(n, w.coeffs, w.scalars, d_w29_, half_, 1, seg_shift_, tv.glv, st);
        launch::g1_set_inf(A, (size_t)128 * bp, st);
        launch_msm(w.scalars, tv, tv.glv, A, 128, nA, bp, 0, st);
        HIPCK(hipEventRecord(w.ev_coeffs, st));
        if (nB > 0) launch_msm((char*)w.scalars + (size_t)nA * 128 * 64 * sizeof(Fr), tv, tv.glv, A + (size_t)lanesA * launch::SIZEOF_JACQ, 128, nB, bp, 0, st);
        HIPCK(hipStreamWaitEvent(w.copy, w.ev_coeffs, 0));
        for (auto& L : pA->launches)
            launch::g1_slp_launch(L.kind, A, bp, (const uint32_t*)pA->d_words + (size_t)L.first * 4, L.count, pA->d_naf, beta_, w.copy, lanesA);
        launch::g1_compress(A + (size_t)128 * bp * launch::SIZEOF_JACQ, d_proofs, 128, bp, nA, w.copy);
        HIPCK(hipEventRecord(w.ev_side, w.copy));
        if (nB > 0) {
            char* AB = A + (size_t)lanesA * launch::SIZEOF_JACQ;
            for (auto& L : pB->launches)
                launch::g1_slp_launch(L.kind, AB, bp, (const uint32_t*)pB->d_words + (size_t)L.first * 4, L.count, pB->d_naf, beta_, st, lanesB);
            launch::g1_compress(AB + (size_t)128 * bp * launch::SIZEOF_JACQ, d_proofs + (size_t)nA * 128 * 48, 128, bp, nB, st);
        }
        HIPCK(hipStreamWaitEvent(st, w.ev_side, 0));
        return;
    }
    if (!tv_pre) {
        const int mk1 = mark_begin(ST_FK20_SCALARS, st);
        launch::fk20_scalars(n, w.coeffs, w.scalars, d_w29_, linmap_mode ? half_ : inv128_, segs, segs == 2 ? two_segments : seg_shift_, tv.glv, st);
        mark_end(mk1, 1, st);
    }
    if (phase != PROOFS_ALL) {
        // Two MSM launches around a cut (a multiple of 64 blobs): the head is issued while the rest of the batch is still on the
        // link.  The scalars are blob-major, the outputs lane-major with stride bp: a sub-range is a pointer offset on both.
        if (!tv_pre || !linmap_mode || segs != 1 || msm_cut <= 0 || msm_cut >= n || msm_cut % 64) throw std::logic_error("run_proofs_from_coeffs: bad MSM cut");
        // (the head on a stream of its own, next to the later sub-batches' light stages and joined before the linear map, was
        // measured too: no gain, profiles/r4_early_msm_ab.log)
        if (phase == PROOFS_HEAD) {
            launch::g1_set_inf(X, (size_t)128 * bp, st);
            launch_msm(w.scalars, tv, tv.glv, X, 128, msm_cut, bp, 0, st);
            return;
        }
        launch_msm((char*)w.scalars + (size_t)msm_cut * 128 * 64 * sizeof(Fr), tv, tv.glv, (char*)X + (size_t)msm_cut * launch::SIZEOF_JACQ, 128, n - msm_cut, bp, 0, st);
    }
    if (phase == PROOFS_ALL) launch::g1_set_inf(X, (size_t)128 * bp, st);
    const bool latency_mode = bp <= LATENCY_MODE_MAX_LANES;  // few 64-blob groups: direct 8 x 16 transforms, 4 rounds instead of 14
    const int mk2 = mark_begin(ST_MSM_FIXED, st);
    void* partial = nullptr;
    if (tv.glv && msm_chunks_ == 8 && segs == 1) {  // the row-sharing schedule's chunk sums
        const size_t need = (size_t)4 * 128 * bp * launch::SIZEOF_JACQ;
        if (need > w.msm_partial_bytes) {
            if (w.msm_partial) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(w.msm_partial)); w.msm_partial = nullptr; }
            HIPCK(hipMalloc(&w.msm_partial, need));
            w.msm_partial_bytes = need;
        }
        partial = w.msm_partial;
    }
    if (phase == PROOFS_ALL) launch_msm(w.scalars, tv, tv.glv, X, 128, segs * n, bp, (latency_mode || linmap_mode) ? 0 : 7, st, partial);
    mark_end(mk2, 1, st);
    if (linmap_mode) {
        const int mk3 = mark_begin(ST_G1_LINMAP, st);
        int n_launches = 0;
        if (slp_walk_) {
            // the constant multiplications as one launch each (there is one), every run of cheap levels as ONE ticket-walking launch
            const size_t ints = launch::g1_slp_walk_sync_ints(bp / 64, slp_max_levels_);
            if (ints > w.slp_sync_ints) {
                if (w.slp_sync) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(w.slp_sync)); w.slp_sync = nullptr; }
                HIPCK(hipMalloc(&w.slp_sync, 2 * ints * sizeof(int)));  // two phases may be in flight back to back: one block each
                w.slp_sync_ints = ints;
            }
            int walk = 0;
            for (auto& ph : slp_phases_) {
                if (ph.level0 < 0) {
                    const SlpLaunch& L = prog->launches[ph.launch0];
                    launch::g1_slp_launch(L.kind, w.slp_arena, bp, (const uint32_t*)prog->d_words + (size_t)L.first * 4, L.count, prog->d_naf, beta_, st);
                } else {
                    launch::g1_slp_walk(w.slp_arena, bp, (const uint32_t*)prog->d_words, (const int*)d_slp_levels_ + ph.level0,
                                        (const int*)d_slp_levels_ + slp_level_total_ + ph.level0, ph.n_levels, ph.max_count, ph.total_ops,
                                        w.slp_sync + (size_t)(walk & 1) * w.slp_sync_ints, wave_slots_, st);
                    if (getenv("ETH_KZG_AMD_SLP_DEBUG")) {  // debugging aid: the walker's counters after the phase
                        fprintf(stderr, "[slp walk %d] launched: level0 %d n_levels %d max_count %d total_ops %d level_total %d\n", walk, ph.level0, ph.n_levels,
                                ph.max_count, ph.total_ops, slp_level_total_);
                        HIPCK(hipStreamSynchronize(st));
                        std::vector<int> h(ints);
                        HIPCK(hipMemcpy(h.data(), w.slp_sync + (size_t)(walk & 1) * w.slp_sync_ints, ints * sizeof(int), hipMemcpyDeviceToHost));
                        const int G = bp / 64;
                        const size_t eo = 256 + 16 * (size_t)G * ph.n_levels;
                        fprintf(stderr, "[slp walk %d] groups %d levels %d tickets drawn:", walk, G, ph.n_levels);
                        for (int sh = 0; sh < std::min(G, 8); sh++) fprintf(stderr, " %d", h[32 * sh]);
                        fprintf(stderr, "\n  done[group 0]:");
                        for (int l = 0; l < ph.n_levels; l++) fprintf(stderr, " %d", h[256 + 16 * l]);
                        fprintf(stderr, "\n  error %d reporters %d first: ticket %d level %d group %d op %d saw %d need %d tickets %d groups_here %d\n", h[eo], h[eo + 1],
                                h[eo + 4], h[eo + 5], h[eo + 6], h[eo + 7], h[eo + 8], h[eo + 9], h[eo + 10], h[eo + 11]);
                        fprintf(stderr, "  kernel saw: total %d tickets %d n_levels %d level_count[0] %d\n", h[eo + 12], h[eo + 13], h[eo + 14], h[eo + 15]);
                    }
                    walk++;
                }
                n_launches++;
            }
        } else {
            for (auto& L : prog->launches)
                launch::g1_slp_launch(L.kind, w.slp_arena, bp, (const uint32_t*)prog->d_words + (size_t)L.first * 4, L.count, prog->d_naf, beta_, st, 0,
                                      mulc_coop_lanes);
            n_launches = (int)prog->launches.size();
        }
        mark_end(mk3, n_launches, st);
        const int mk4 = mark_begin(ST_COMPRESS, st);
        launch::g1_compress((const char*)w.slp_arena + (size_t)128 * bp * launch::SIZEOF_JACQ, d_proofs, 128, bp, n, st);
        mark_end(mk4, 1, st);
        return;
    }
    if (n <= circ_max_) {  // a handful of blobs: the two transforms as one circulant product (k_g1circ.hip)
        if (!w.circ_table) HIPCK(hipMalloc(&w.circ_table, launch::g1_circ_table_bytes(circ_max_, circ_T_)));
        const int mk5 = mark_begin(ST_G1_IFFT, st);
        launch::g1_circ128(w.X, bp, n, segs, w.circ_table, circ_T_, d_circ_terms_, circ_per_lane_, beta_, st);
        mark_end(mk5, 2, st);
    } else if (latency_mode) {
        if (!w.dft_tmp) {
            HIPCK(hipMalloc(&w.dft_tmp, (size_t)128 * LATENCY_MODE_MAX_LANES * launch::SIZEOF_JACQ));
            HIPCK(hipMalloc(&w.dft_prod, (size_t)128 * 16 * LATENCY_MODE_MAX_LANES * launch::SIZEOF_JACQ));
        }
        const int mk6 = mark_begin(ST_G1_IFFT, st);
        launch::g1_dft128_direct(w.X, w.dft_tmp, w.dft_prod, bp, 128, 64, 1, 0, d_naf_, beta_, st);  // h = first 64 outputs
        mark_end(mk6, 2, st);
        const int mk7 = mark_begin(ST_G1_FFT, st);
        launch::g1_dft128_direct(w.X, w.dft_tmp, w.dft_prod, bp, 64, 128, 0, 1, d_naf_, beta_, st);  // proofs, bit-reversed
        mark_end(mk7, 2, st);
    } else {
        const int mk8 = mark_begin(ST_G1_IFFT, st);
        g1_ifft128_take64(w.X, bp, st);
        mark_end(mk8, 7, st);  // 7 layers (each = one twiddle-multiplication launch + one butterfly launch)
        const int mk9 = mark_begin(ST_G1_FFT, st);
        g1_fft128_from64(w.X, bp, st);
        mark_end(mk9, 7, st);
    }
    const int mk10 = mark_begin(ST_COMPRESS, st);
    launch::g1_compress(w.X, d_proofs, 128, bp, n, st);
    mark_end(mk10, 1, st);
}

// the kernels of one prover call on `st`, scratch from `w`; optionally records `after_cells` once the cells are written
void Engine::enqueue_compute(Work& w, int n, const uint8_t* d_blobs, uint8_t* d_cells, uint8_t* d_proofs, hipStream_t st,
                             hipEvent_t after_cells) {
    ensure_workspace(w, n);
    HIPCK(hipStreamWaitEvent(st, w.done, 0));  // an earlier call may still be using this set on another stream
    HIPCK(hipMemsetAsync(w.status, 0, n * sizeof(int), st));
    const int mk11 = mark_begin(ST_BLOB_TO_COEFFS, st);
    launch::blob_to_coeffs(n, d_blobs, w.coeffs, nullptr, w.status, d_w29_, n_inv4096_, st);
    mark_end(mk11, 1, st);
    // a handful of blobs is a chain of latencies: the cells (one 8192-point transform, 0.08 ms) then run on the set's second
    // stream next to the proof stages instead of in front of them
    const bool side = d_cells && d_proofs && n <= SIDE_CELLS_MAX && w.copy && !profiling_;
    if (side) {
        HIPCK(hipEventRecord(w.ev_coeffs, st));
        HIPCK(hipStreamWaitEvent(w.copy, w.ev_coeffs, 0));
        launch::coeffs_to_cells(n, w.coeffs, d_cells, d_w29_, w.copy);
        if (after_cells) HIPCK(hipEventRecord(after_cells, w.copy));
        HIPCK(hipEventRecord(w.ev_side, w.copy));
        run_proofs_from_coeffs(w, n, d_proofs, st);
        HIPCK(hipStreamWaitEvent(st, w.ev_side, 0));
        return;
    }
    if (d_cells) {
        const int mk12 = mark_begin(ST_COEFFS_TO_CELLS, st);
        launch::coeffs_to_cells(n, w.coeffs, d_cells, d_w29_, st);
        mark_end(mk12, 1, st);
    }
    if (after_cells) HIPCK(hipEventRecord(after_cells, st));
    if (d_proofs) run_proofs_from_coeffs(w, n, d_proofs, st);
}

int Engine::compute_cells_and_kzg_proofs_device(int n, const uint8_t* d_blobs, uint8_t* d_cells, uint8_t* d_proofs,
                                                int* h_status, hipStream_t st, bool sync) {
    if (n <= 0) return OK;
    Work* held = nullptr;
    try {
        HIPCK(hipSetDevice(dev_));
        Work& w = lease_work(1, NW - 1);
        held = &w;
        if (!st) {  // NULL: the library's stream, ordered behind whatever the caller has queued on the default stream so far
            st = w.stream;
            HIPCK(hipEventRecord(w.ev_in, nullptr));
            HIPCK(hipStreamWaitEvent(st, w.ev_in, 0));
        }
        enqueue_compute(w, n, d_blobs, d_cells, d_proofs, st, nullptr);
        if (h_status) HIPCK(hipMemcpyAsync(h_status, w.status, n * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCK(hipEventRecord(w.done, st));
        HIPCK(hipGetLastError());
        release_work(w);
        held = nullptr;
        if (sync || h_status) HIPCK(hipStreamSynchronize(st));
    } catch (const std::exception& e) {
        if (held) release_work(*held);
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::blob_to_kzg_commitment_device(int n, const uint8_t* d_blobs, uint8_t* d_commitments, int* h_status,
                                          hipStream_t st, bool sync) {
    if (n <= 0) return OK;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        if (!st) {  // NULL: the library's stream, ordered behind whatever the caller has queued on the default stream so far
            st = stream_;
            HIPCK(hipEventRecord(work_[0].ev_in, nullptr));
            HIPCK(hipStreamWaitEvent(st, work_[0].ev_in, 0));
        }
        ensure_workspace(n);
        HIPCK(hipStreamWaitEvent(st, work_[0].done, 0));  // an earlier asynchronous call on another stream may still use the workspace
        const int bp = ((n + 63) / 64) * 64;
        HIPCK(hipMemsetAsync(d_status_, 0, n * sizeof(int), st));
        // commit = MSM_4096(coeffs, g1_monomial)  (fk20/prover.rs:128-145, commit_key.rs:38-44):
        // 64 groups of 64 bases through the window-table kernel, then a fold over the groups.
        launch::blob_to_coeffs(n, d_blobs, d_coeffs_, d_canon_, d_status_, d_w29_, n_inv4096_, st);
        launch::g1_set_inf(d_X_, (size_t)64 * bp, st);
        launch_msm(d_canon_, TAB_SRS, d_X_, 64, n, bp, 0, st);
        launch::g1_sum_positions(d_X_, 64, bp, n, st);
        launch::g1_compress(d_X_, d_commitments, 1, bp, n, st);
        if (h_status) HIPCK(hipMemcpyAsync(h_status, d_status_, n * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCK(hipEventRecord(work_[0].done, st));
        HIPCK(hipGetLastError());
        if (sync || h_status) HIPCK(hipStreamSynchronize(st));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------
// host-buffer entry points: stage through device buffers owned by the engine
// The reference's entry point (bindings/c/src/lib.rs:226-236) and its batched form: host pointers in, 256 caller
// buffers per blob out.  The light per-blob stages run per sub-batch as the blobs arrive -- helper threads gather them
// into pinned memory, the upload, blob_to_coeffs and coeffs_to_cells follow on the compute stream, and the cells (95 %
// of the output bytes) go back on a copy stream and are scattered to the caller's buffers by the helper threads --
// while the heavy stages (fixed-base MSMs, the G1 linear map) run ONCE over the whole batch at its saturated rate.
int Engine::compute_cells_and_kzg_proofs_host(int n, const uint8_t* const* blobs, uint8_t* const* const* cells,
                                              uint8_t* const* const* proofs, int* h_status) {
    if (n <= 0) return OK;
    constexpr int SUPER = 4096, SUB = 256, PART = 32;
    const bool threaded = n >= 32;  // small calls: everything on the calling thread (latency)
    const bool trace = getenv("ETH_KZG_AMD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // ETH_KZG_AMD_TRACE_SLOW=<ms>: report the steps of a call that took longer (which HIP call waited, and for how long)
    static const double slow_ms = [] { const char* e = getenv("ETH_KZG_AMD_TRACE_SLOW"); return e ? atof(e) : 0.0; }();
    double step_at[8] = {0};
    int n_steps = 0;
    auto step = [&]() { if (slow_ms > 0 && n_steps < 8) step_at[n_steps++] = now_ms(); };
    struct SlowReport {
        const double& limit; double* at; int& n; std::function<double()> now;
        ~SlowReport() {
            if (limit <= 0 || now() < limit) return;
            fprintf(stderr, "[host-batch] @%.0f ms: slow call, %.1f ms; steps (lease, enqueue sub-batches, enqueue proofs, events recorded, cells back, all back):", trace_clock_ms(), now());
            for (int i = 0; i < n; i++) fprintf(stderr, " %.1f", at[i]);
            fprintf(stderr, "\n");
        }
    } slow_report{slow_ms, step_at, n_steps, now_ms};
    Work* held = nullptr;
    std::atomic<int> failed{0};
    std::mutex err_mu;
    std::string err_text;
    auto fail = [&](const std::exception& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!failed.exchange(1)) err_text = e.what();
    };
    std::atomic<int> outstanding{0};  // helper-thread tasks of this call still running or queued
    auto nap = [] { std::this_thread::sleep_for(std::chrono::microseconds(30)); };  // waits below are tens of microseconds to milliseconds long
    auto drain = [&]() { while (outstanding.load(std::memory_order_acquire) > 0) nap(); };
    try {
        HIPCK(hipSetDevice(dev_));
        if (threaded)
            std::call_once(host_pool_once_, [this] {
                // memcpy helpers: the gather of 2048 blobs is 268 MB and the MSMs cannot start before its last byte is uploaded,
                // so its bandwidth is exposed time (4 threads: ~10 ms, 8: ~5 ms).  ETH_KZG_AMD_HOST_THREADS overrides.
                int t = 8;
                const unsigned hw = std::thread::hardware_concurrency();
                if (hw && (int)hw < 2 * t) t = (int)hw / 2 > 1 ? (int)hw / 2 : 1;
                if (const char* e = getenv("ETH_KZG_AMD_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) t = v; }
                host_pool_.reset(new HostPool(t, dev_));
            });
        Work& w = lease_work(1, NW - 1);
        held = &w;
        step();
        for (int s0 = 0; s0 < n && !failed.load(); s0 += SUPER) {
            const int ns = std::min(SUPER, n - s0);
            // sub-batch boundaries: 256 blobs each, but the first one short (64) so that the link starts carrying blobs 0.15 ms into
            // the call instead of 0.6 ms (the gather of 256 blobs)
            std::vector<int> cut{0};
            if (ns > SUB) cut.push_back(SUB / 4);
            if (ns > SUB) cut.push_back(SUB);
            while (cut.back() < ns) cut.push_back(std::min(ns, cut.back() + SUB));
            const int n_sub = (int)cut.size() - 1;
            ensure_workspace(w, ns);
            ensure_staging(w, ns);
            while ((int)w.sub_events.size() < 3 * n_sub) {
                hipEvent_t e;
                HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                w.sub_events.push_back(e);
            }
            HIPCK(hipStreamWaitEvent(w.stream, w.done, 0));
            HIPCK(hipMemsetAsync(w.status, 0, ns * sizeof(int), w.stream));
            // Everything the helper tasks reach by reference is declared here, BEFORE the guard that waits for them: on any
            // way out of this scope (an exception included) the guard runs first and the objects die after the last task.
            std::vector<std::atomic<int>> gathered(n_sub);  // parts of sub-batch i still to copy
            std::function<void(int, int, int)> scatter_cells;
            std::function<void(int, int)> scatter_proofs;
            struct Drain {
                std::function<void()> f;
                ~Drain() { f(); }
            } drain_on_exit{drain};
            // gather tasks for the whole super-batch, in order
            for (int i = 0; i < n_sub; i++) {
                const int lo = cut[i], hi = cut[i + 1], parts = (hi - lo + PART - 1) / PART;
                gathered[i].store(threaded ? parts : 0, std::memory_order_relaxed);
                if (!threaded) {
                    for (int b = lo; b < hi; b++) memcpy(w.h_in + (size_t)b * BYTES_PER_BLOB, blobs[s0 + b], BYTES_PER_BLOB);
                    continue;
                }
                for (int p0 = lo; p0 < hi; p0 += PART) {
                    const int p1 = std::min(hi, p0 + PART);
                    outstanding.fetch_add(1, std::memory_order_relaxed);
                    host_pool_->submit([&, i, p0, p1, s0] {
                        for (int b = p0; b < p1; b++) memcpy(w.h_in + (size_t)b * BYTES_PER_BLOB, blobs[s0 + b], BYTES_PER_BLOB);
                        gathered[i].fetch_sub(1, std::memory_order_release);
                        outstanding.fetch_sub(1, std::memory_order_release);
                    });
                }
            }
            // scatter of one sub-batch's status words and cells once its copy-stream event has fired; a task that finds the
            // event pending goes back to the end of the queue instead of blocking a helper thread
            scatter_cells = [&](int i, int lo, int hi) {
                const hipError_t q = hipEventQuery(w.sub_events[2 * i + 1]);
                if (q == hipErrorNotReady) {
                    nap();
                    host_pool_->submit([&, i, lo, hi] { scatter_cells(i, lo, hi); });
                    return;
                }
                try {
                    HIPCK(q);
                    for (int b = lo; b < hi; b++) {
                        if (h_status) h_status[s0 + b] = w.h_status[b] ? ERR_SCALAR : OK;
                        if (w.h_status[b] || !cells) continue;
                        const uint8_t* src = w.h_cells + (size_t)b * N_CELLS * BYTES_PER_CELL;
                        for (int k = 0; k < N_CELLS; k++) memcpy(cells[s0 + b][k], src + (size_t)k * BYTES_PER_CELL, BYTES_PER_CELL);
                    }
                } catch (const std::exception& e) { fail(e); }
                outstanding.fetch_sub(1, std::memory_order_release);
            };
            // a handful of blobs: the cells are computed on the copy stream, next to the proof stages (enqueue_compute does the same)
            const bool side = cells && proofs && ns <= circ_max_;
            // batches that take the compiled linear map: the MSM scalars are computed sub-batch by sub-batch under the uploads
            // (0.77 ms per 2048 blobs that used to sit between the last upload and the MSMs)
            const bool early_scalars = proofs && use_linmap_ && ns > circ_max_;
            const TableView tv_call = table_view(TAB_FK);
            // A batch of several rounds of the chip: the MSMs of the first 256 blobs are launched as soon as THEIR scalars exist,
            // 0.7 ms into the call, and run while the other 235 MB are on the link; the MSMs of the rest follow.  (An MSM launch costs
            // ~1.3 ms beyond its share of the work -- the last waves of a launch -- so the head is as small as will still cover
            // the uploads: 256 blobs = 6 ms.)  ETH_KZG_AMD_EARLY_MSM=<blobs> moves the cut, 0 = one launch after the last upload.
            static const int early_msm_blobs = [] { const char* e = getenv("ETH_KZG_AMD_EARLY_MSM"); return e ? atoi(e) : 256; }();
            int msm_cut = 0;
            if (early_scalars && early_msm_blobs > 0 && ns >= 4 * early_msm_blobs && msm_chunks_ != 8 && !slp_walk_)
                for (int c : cut)
                    if (!msm_cut && c >= early_msm_blobs && c % 64 == 0 && c < ns) msm_cut = c;
            // The uploads run back to back on the copy stream (its own hardware queue), the light per-blob stages follow on the compute
            // stream sub-batch by sub-batch: in one stream the copy engine idled during the kernels and the kernels during the copies
            // (rocprofv3 --memory-copy-trace: 7.6 ms until the MSMs could start, for 5.2 ms of link time)
            for (int i = 0; i < n_sub; i++) {
                const int lo = cut[i], hi = cut[i + 1], nb = hi - lo;
                while (gathered[i].load(std::memory_order_acquire) > 0) std::this_thread::yield();  // a few hundred microseconds: the gather of 32 MB
                HIPCK(hipMemcpyAsync(w.d_in + (size_t)lo * BYTES_PER_BLOB, w.h_in + (size_t)lo * BYTES_PER_BLOB, (size_t)nb * BYTES_PER_BLOB,
                                     hipMemcpyHostToDevice, w.copy));
                HIPCK(hipEventRecord(w.sub_events[2 * n_sub + i], w.copy));
                HIPCK(hipStreamWaitEvent(w.stream, w.sub_events[2 * n_sub + i], 0));
                launch::blob_to_coeffs(nb, w.d_in + (size_t)lo * BYTES_PER_BLOB, (char*)w.coeffs + (size_t)lo * N_BLOB * sizeof(Fr), nullptr,
                                       w.status + lo, d_w29_, n_inv4096_, w.stream);
                if (cells && !side)
                    launch::coeffs_to_cells(nb, (char*)w.coeffs + (size_t)lo * N_BLOB * sizeof(Fr),
                                            w.d_cells + (size_t)lo * N_CELLS * BYTES_PER_CELL, d_w29_, w.stream);
                HIPCK(hipEventRecord(w.sub_events[2 * i], w.stream));
                if (early_scalars)
                    launch::fk20_scalars(nb, (char*)w.coeffs + (size_t)lo * N_BLOB * sizeof(Fr), (char*)w.scalars + (size_t)lo * 128 * 64 * sizeof(Fr),
                                         d_w29_, half_, 1, seg_shift_, tv_call.glv, w.stream);
                if (msm_cut && hi == msm_cut) run_proofs_from_coeffs(w, ns, w.d_proofs, w.stream, &tv_call, PROOFS_HEAD, msm_cut);
                if (trace) fprintf(stderr, "[host-batch] sub-batch %d (%d blobs) enqueued at %.2f ms\n", i, nb, now_ms());
            }
            // The cells go back on the same copy stream, i.e. behind the LAST upload: the fixed-base MSMs run once over the whole batch
            // and cannot start before every blob is up, so until then the PCIe link belongs to the uploads; the 537 MB of cells then
            // have the 50 ms of the heavy stages to come down.
            for (int i = 0; i < n_sub; i++) {
                const int lo = cut[i], hi = cut[i + 1], nb = hi - lo;
                HIPCK(hipStreamWaitEvent(w.copy, w.sub_events[2 * i], 0));
                if (side)
                    launch::coeffs_to_cells(nb, (char*)w.coeffs + (size_t)lo * N_BLOB * sizeof(Fr),
                                            w.d_cells + (size_t)lo * N_CELLS * BYTES_PER_CELL, d_w29_, w.copy);
                HIPCK(hipMemcpyAsync(w.h_status + lo, w.status + lo, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, w.copy));
                if (cells)
                    HIPCK(hipMemcpyAsync(w.h_cells + (size_t)lo * N_CELLS * BYTES_PER_CELL, w.d_cells + (size_t)lo * N_CELLS * BYTES_PER_CELL,
                                         (size_t)nb * N_CELLS * BYTES_PER_CELL, hipMemcpyDeviceToHost, w.copy));
                HIPCK(hipEventRecord(w.sub_events[2 * i + 1], w.copy));
                if (threaded) {
                    for (int p0 = lo; p0 < hi; p0 += PART) {
                        const int p1 = std::min(hi, p0 + PART);
                        outstanding.fetch_add(1, std::memory_order_relaxed);
                        host_pool_->submit([&, i, p0, p1] { scatter_cells(i, p0, p1); });
                    }
                }
            }
            step();
            if (proofs) {
                if (msm_cut) run_proofs_from_coeffs(w, ns, w.d_proofs, w.stream, &tv_call, PROOFS_TAIL, msm_cut);
                else run_proofs_from_coeffs(w, ns, w.d_proofs, w.stream, early_scalars ? &tv_call : nullptr);
                HIPCK(hipMemcpyAsync(w.h_proofs, w.d_proofs, (size_t)ns * N_CELLS * 48, hipMemcpyDeviceToHost, w.stream));
            }
            step();
            HIPCK(hipEventRecord(w.ev_done, w.stream));
            HIPCK(hipEventRecord(w.done, w.stream));
            HIPCK(hipGetLastError());
            step();
            if (!threaded) {  // small call: status and cells as soon as they are back, proofs at the end, all on this thread
                HIPCK(hipEventSynchronize(w.sub_events[1]));
                for (int b = 0; b < ns; b++) {
                    if (h_status) h_status[s0 + b] = w.h_status[b] ? ERR_SCALAR : OK;
                    if (w.h_status[b] || !cells) continue;
                    const uint8_t* src = w.h_cells + (size_t)b * N_CELLS * BYTES_PER_CELL;
                    for (int k = 0; k < N_CELLS; k++) memcpy(cells[s0 + b][k], src + (size_t)k * BYTES_PER_CELL, BYTES_PER_CELL);
                }
            }
            step();
            HIPCK(hipEventSynchronize(w.ev_done));
            HIPCK(hipStreamSynchronize(w.copy));
            step();
            if (trace) fprintf(stderr, "[host-batch] proofs of %d blobs back at %.2f ms\n", ns, now_ms());
            if (proofs) {
                scatter_proofs = [&, s0](int lo, int hi) {
                    for (int b = lo; b < hi; b++) {
                        if (w.h_status[b]) continue;
                        const uint8_t* src = w.h_proofs + (size_t)b * N_CELLS * 48;
                        for (int k = 0; k < N_CELLS; k++) memcpy(proofs[s0 + b][k], src + (size_t)k * 48, 48);
                    }
                };
                if (threaded && ns >= 256) {
                    const int parts = 8;
                    for (int t = 1; t < parts; t++) {
                        outstanding.fetch_add(1, std::memory_order_relaxed);
                        host_pool_->submit([&, t] { scatter_proofs(t * ns / parts, (t + 1) * ns / parts); outstanding.fetch_sub(1, std::memory_order_release); });
                    }
                    scatter_proofs(0, ns / parts);
                } else scatter_proofs(0, ns);
            }
            // drain_on_exit: the helper tasks are done before the pinned buffers are reused by the next super-batch
        }
        release_work(w);
        held = nullptr;
        if (trace) fprintf(stderr, "[host-batch] %d blobs delivered at %.2f ms\n", n, now_ms());
        if (failed.load()) throw std::runtime_error(err_text);
    } catch (const std::exception& e) {
        if (held) {  // let the streams drain and the helper tasks finish before the set is handed back, whatever happened
            (void)hipStreamSynchronize(held->stream);
            (void)hipStreamSynchronize(held->copy);
            drain();
            release_work(*held);
        }
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::blob_to_kzg_commitment_host(int n, const uint8_t* const* blobs, uint8_t* const* out, int* h_status) {
    if (n <= 0) return OK;
    std::lock_guard<std::recursive_mutex> whole_call(mu_);
    uint8_t* d_out = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        try {
            HIPCK(hipSetDevice(dev_));
            if (n > stage_cap_) {
                if (d_in_) { HIPCK(hipFree(d_in_)); HIPCK(hipFree(d_cells_)); HIPCK(hipFree(d_proofs_)); }
                HIPCK(hipMalloc(&d_in_, (size_t)n * BYTES_PER_BLOB));
                HIPCK(hipMalloc(&d_cells_, (size_t)n * N_CELLS * BYTES_PER_CELL));
                HIPCK(hipMalloc(&d_proofs_, (size_t)n * N_CELLS * 48));
                stage_cap_ = n;
            }
            for (int b = 0; b < n; b++)
                HIPCK(hipMemcpyAsync(d_in_ + (size_t)b * BYTES_PER_BLOB, blobs[b], BYTES_PER_BLOB, hipMemcpyHostToDevice, stream_));
            d_out = d_proofs_;  // reuse staging
        } catch (const std::exception& e) {
            set_error(e);
            return ERR_DEVICE;
        }
    }
    std::vector<int> st(n);
    int rc = blob_to_kzg_commitment_device(n, d_in_, d_out, st.data(), stream_, true);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        std::vector<uint8_t> h((size_t)n * 48);
        HIPCK(hipMemcpy(h.data(), d_out, h.size(), hipMemcpyDeviceToHost));
        for (int b = 0; b < n; b++) {
            if (h_status) h_status[b] = st[b] ? ERR_SCALAR : OK;
            if (!st[b]) memcpy(out[b], h.data() + (size_t)b * 48, 48);
        }
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------
// stage-level test hooks
int Engine::test_fr_ntt4096(const uint8_t* in_be, uint8_t* out_be, int inverse_dit) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        uint8_t *di, *dout;
        HIPCK(hipMalloc(&di, BYTES_PER_BLOB));
        HIPCK(hipMalloc(&dout, BYTES_PER_BLOB));
        HIPCK(hipMemcpy(di, in_be, BYTES_PER_BLOB, hipMemcpyHostToDevice));
        launch::test_ntt4096(di, dout, d_w29_, n_inv4096_, inverse_dit, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out_be, dout, BYTES_PER_BLOB, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di));
        HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// in/out: [lane][128][48 B]; both directions natural in -> natural out (inverse is unscaled)
int Engine::test_g1_fft128(const uint8_t* in, uint8_t* out, int n_lanes, int inverse) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        int stride = ((n_lanes + 63) / 64) * 64;
        size_t bytes = (size_t)n_lanes * 128 * 48;
        uint8_t *di, *dout;
        void* X;
        HIPCK(hipMalloc(&di, bytes));
        HIPCK(hipMalloc(&dout, bytes));
        size_t nx = (size_t)128 * stride;
        const size_t PS = launch::SIZEOF_JACQ;
        HIPCK(hipMalloc(&X, nx * PS));
        HIPCK(hipMemcpy(di, in, bytes, hipMemcpyHostToDevice));
        launch::g1_set_inf(X, nx, stream_);
        launch::test_load_points(di, X, n_lanes, stride, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        std::vector<uint8_t> hx(nx * PS), hy(nx * PS);
        auto brp = [](int v) { int r = 0; for (int i = 0; i < 7; i++) r |= ((v >> i) & 1) << (6 - i); return r; };
        auto permute = [&]() {
            HIPCK(hipMemcpy(hx.data(), X, nx * PS, hipMemcpyDeviceToHost));
            for (int p = 0; p < 128; p++) memcpy(&hy[(size_t)brp(p) * stride * PS], &hx[(size_t)p * stride * PS], stride * PS);
            HIPCK(hipMemcpy(X, hy.data(), nx * PS, hipMemcpyHostToDevice));
        };
        if (inverse) permute();  // DIT wants bit-reversed input
        g1_fft128_full(X, stride, inverse, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        if (!inverse) permute();  // DIF leaves bit-reversed output
        launch::g1_compress(X, dout, 128, stride, n_lanes, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(dout)); HIPCK(hipFree(X));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// scalars: [n_msm][128 groups][64] BE -> out [n_msm][128][48]: the 128 fixed-base MSMs of stage D
int Engine::test_fixed_msm(const uint8_t* scalars_be, int n_msm, uint8_t* out) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        size_t ns = (size_t)n_msm * 128 * 64;
        int stride = ((n_msm + 63) / 64) * 64;
        uint8_t *di, *dout;
        void *sc, *X;
        HIPCK(hipMalloc(&di, ns * 32));
        HIPCK(hipMalloc(&sc, ns * sizeof(Fr)));
        HIPCK(hipMalloc(&X, (size_t)128 * stride * launch::SIZEOF_JACQ));
        HIPCK(hipMalloc(&dout, (size_t)n_msm * 128 * 48));
        HIPCK(hipMemcpy(di, scalars_be, ns * 32, hipMemcpyHostToDevice));
        launch::test_scalars_be(di, sc, ns, stream_);
        launch::g1_set_inf(X, (size_t)128 * stride, stream_);
        launch_msm(sc, TAB_FK, X, 128, n_msm, stride, 0, stream_);
        launch::g1_compress(X, dout, 128, stride, n_msm, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, (size_t)n_msm * 128 * 48, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(sc)); HIPCK(hipFree(X)); HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::test_g1_decompress(const uint8_t* in, int n, int subgroup_check, int* h_status, uint8_t* out) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        uint8_t *di, *dout;
        void* pts;
        int* st;
        HIPCK(hipMalloc(&di, (size_t)n * 48)); HIPCK(hipMalloc(&dout, (size_t)n * 48));
        HIPCK(hipMalloc(&pts, (size_t)n * sizeof(G1Affine))); HIPCK(hipMalloc(&st, n * sizeof(int)));
        HIPCK(hipMemcpy(di, in, (size_t)n * 48, hipMemcpyHostToDevice));
        launch::g1_decompress(di, pts, st, n, subgroup_check, beta_, stream_);
        launch::test_recompress(pts, dout, n, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(h_status, st, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(out, dout, (size_t)n * 48, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(dout)); HIPCK(hipFree(pts)); HIPCK(hipFree(st));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        size_t nb = (size_t)n * (is_fp ? 48 : 32);
        uint8_t *da, *db, *dout;
        HIPCK(hipMalloc(&da, nb)); HIPCK(hipMalloc(&db, nb)); HIPCK(hipMalloc(&dout, nb));
        HIPCK(hipMemcpy(da, a, nb, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(db, b, nb, hipMemcpyHostToDevice));
        launch::test_field_mul(da, db, dout, n, is_fp, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, nb, hipMemcpyDeviceToHost));
        HIPCK(hipFree(da)); HIPCK(hipFree(db)); HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

}  // namespace kzg
