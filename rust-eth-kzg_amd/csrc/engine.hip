// Engine: context set-up and kernel orchestration for the KZG hot path on one MI355X.
// Mirrors ProverContext::new / FK20Prover::new (reference: crates/eip7594/src/prover.rs:62-94,
// crates/cryptography/kzg_multi_open/src/fk20/prover.rs:64-125, batch_toeplitz.rs:34-78) and the
// per-blob pipeline of compute_multi_opening_proofs (fk20/prover.rs:173-228) as a batch of kernels.
#include "engine_internal.hpp"

namespace kzg {

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
double trace_clock_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// The text of a device error belongs to the call that failed, and calls run concurrently: keep it per thread.
static thread_local std::string t_last_error;
const std::string& Engine::last_error() const { return t_last_error; }
void Engine::set_error(const std::exception& e) { t_last_error = e.what(); }

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
std::atomic<bool> g_exiting{false};
std::mutex g_engines_mu;
std::vector<Engine*> g_engines;
void Engine::stop_builder() {
    cancel_build_.store(true);
    if (builder_.joinable()) builder_.join();
}
void stop_all_builders_at_exit() {
    g_exiting.store(true);
    std::vector<Engine*> live;
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        live = g_engines;
    }
    for (Engine* e : live) e->stop_builder();
}

// the primary when it is idle, an idle auxiliary lane, a new lane built OUTSIDE the pool's lock (0.1 s of set-up kernels: callers that
// find the existing lanes busy meanwhile queue on one of those instead of behind this constructor; the lane shares this engine's
// window tables -- no registry look-up, no lock a table build could hold), else wait on a lane, round-robin: host_sync.hpp (LanePool)
Engine::SerialLease Engine::lease_serial() {
    return lanes_.lease(this, max_lanes_, auxiliary_, [this] {
        try {
            return std::unique_ptr<Engine>(new Engine(use_precomp_, dev_, /*primary=*/this));
        } catch (...) {
            (void)hipGetLastError();  // no room for another lane: the caller queues on an existing one
            throw;
        }
    });
}

Engine::Engine(bool use_precomp, int device, const Engine* primary, double table_budget_gb) : dev_(device), use_precomp_(use_precomp), primary_(primary), auxiliary_(primary != nullptr) {
    // every environment knob is read here, once (knobs.hpp); the auxiliary engines of a context copy their context's values
    knobs_ = primary ? primary->knobs_ : Knobs::from_env();
    if (knobs_.serial_lanes) max_lanes_ = knobs_.serial_lanes;
    if (knobs_.device_batch_max) device_batch_max_ = ((knobs_.device_batch_max + 63) / 64) * 64;
    if (use_precomp) {
        // default: the widest GLV table inside the budget (108 GB: nine windows, 71 GB, 18 gathered additions per base; the commitment table likewise, 35 GB), see build_final_tables
        if (launch::glv_width_supported(knobs_.glv_window)) want_glv_c_ = knobs_.glv_window;
        // memory budget for the window tables (both together), in GB: the constructor's argument, else ETH_KZG_AMD_TABLE_GB
        // (a number, or "max" = whatever the HBM still holds -- the behaviour up to round 4), else DEFAULT_TABLE_BUDGET_GB
        if (table_budget_gb != 0) table_budget_gb_ = table_budget_gb;
        else if (knobs_.table_budget_gb != 0) table_budget_gb_ = knobs_.table_budget_gb;
    }
    vm_search_ = knobs_.vm_search;
    arena_signed_ = knobs_.arena_signed;
    msm_split_ = knobs_.msm_split;
    if (knobs_.pip_shift_min >= 1) pip_shift_min_ = knobs_.pip_shift_min;
    // largest batch on the circulant form.  Its cost grows with every blob (1 blob 1.11 ms, 2: 1.36, 3: 1.78, 4: 1.87); the compiled
    // map with four lanes per blob and the flat MSM is flat (3 .. 8 blobs: 1.65 .. 1.86 ms).  Round 3's cross-over was 8 blobs, round
    // 4's 4; since the round-6 changes below 16 blobs (two lanes per MSM window, the pair forms) it is 2: 3 blobs 1.78 -> 1.65 ms,
    // 4 blobs 1.87 -> 1.70 ms (tools/ab_circ_max.sh, three alternated rounds on one box, profiles/r6_ab_circ_max.log).
#ifdef KZG_CIRC_MAX  // (experiment builds of that A/B; <= 4)
    circ_max_ = KZG_CIRC_MAX;
#else
    circ_max_ = 2;
#endif
    // Everything below can throw (HIPCK, the SRS checks).  A constructor that throws runs no destructor: teardown() frees what
    // was built so far -- streams, events, device buffers, the published start tables' references -- and the exception goes on to
    // eth_kzg_amd_das_context_try_new, which promises NULL + a message and never a dead process.
    try {
        construct();
    } catch (...) {
        teardown();
        throw;
    }
}

void Engine::construct() {
    const bool trace = knobs_.trace;
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
        if (knobs_.msm_chunks >= 0) msm_chunks_ = knobs_.msm_chunks;
    }
    HIPCK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (!primary_) {  // the table builder runs at the lowest stream priority: callers' kernels are dispatched ahead of its waves
        int prio_low = 0, prio_high = 0;
        HIPCK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIPCK(hipStreamCreateWithPriority(&build_stream_, hipStreamNonBlocking, prio_low));
    }
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
    if (knobs_.fault == "constructor" && !primary_) throw std::runtime_error("injected fault (ETH_KZG_AMD_FAULT=constructor)");
    start_builder();  // last: nothing after it can throw
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
    const bool trace = knobs_.trace;
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

Engine::~Engine() { teardown(); }

// Frees whatever the engine holds; every member it looks at is null / empty until it has been created, so it also serves a
// constructor that failed half-way.
void Engine::teardown() noexcept {
    (void)hipSetDevice(dev_);
    lanes_.clear();
    stop_builder();  // an unfinished build of the wide tables is abandoned (within one 2 GB piece)
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        g_engines.erase(std::remove(g_engines.begin(), g_engines.end(), this), g_engines.end());
    }
    if (build_stream_) hipStreamDestroy(build_stream_);
    host_pool_.reset();  // joins the helper threads before anything they might touch goes away
    vm_pool_.reset();
    stage_pool_.reset();
    void* ptrs[] = {d_w8192_, d_w29_, d_naf_, d_srs_, d_fk_bases_, d_in_, d_cells_, d_proofs_, d_coset_, d_coset_inv_, d_circ_terms_};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (SlpProgram& P : slp_prog_) {
        if (P.d_words) hipFree(P.d_words);
        if (P.d_naf && P.owns_naf) hipFree(P.d_naf);
    }
    for (Work& w : work_) {
        void* dev[] = {w.coeffs, w.canon, w.scalars, w.X, w.status, w.circ_table, w.slp_arena, w.d_in, w.d_cells, w.d_proofs};
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
    s.allow_toom8 = true;
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
    const bool verbose = knobs_.trace;
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
// levels wins (tools/linmap_explore.cpp).  Measured per group count on one box (profiles/archive/r4_linmap_programs.log; g1_linmap ms):
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
    if (knobs_.slp_program >= 0 && knobs_.slp_program < SLP_COUNT) slp_force_ = knobs_.slp_program;  // tests: one compilation of the map at every batch size
    w128_.resize(128);
    for (int e = 0; e < 128; e++) w128_[e] = w8192_mont[64 * e];
    build_slp_program(SLP_TUNED_FUSED);
    build_slp_program(SLP_TUNED);
    const SlpProgram& fused = slp_prog_[SLP_TUNED_FUSED];
    {   // the two schedules of the tuned plan must agree (the arena is sized per program; inputs and outputs sit at the same slots in all)
        if (fused.info[0] != slp_prog_[SLP_TUNED].info[0]) throw std::runtime_error("FK20 proofs map: the two schedules differ");
    }
    for (int i = 0; i < 3; i++) slp_info_[i] = fused.info[i];
    slp_info_[3] = fused.info[3];
    const Fr h = inv(fr_from_u64(2));
    memcpy(&half_, &h, 32);
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
}  // namespace kzg
