// The prover paths of a context: compute_cells_and_kzg_proofs (device-resident and host-pointer forms), compute_cells,
// blob_to_kzg_commitment -- workspaces, the MSM stage over the table views, the G1 linear map, per-stage timing marks.
// Reference: compute_multi_opening_proofs (crates/cryptography/kzg_multi_open/src/fk20/prover.rs:173-228), crates/eip7594/src/prover.rs:117-148.
#include "engine_internal.hpp"

namespace kzg {

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
void Engine::ensure_workspace(Work& w, int n, bool release) {
    if (n <= w.cap && !release) return;
    int cap = ((n + 63) / 64) * 64;
    void** ptrs[] = {&w.coeffs, &w.canon, &w.scalars, &w.X, (void**)&w.status};
    for (void** p : ptrs)
        if (*p) { HIPCK(hipFree(*p)); *p = nullptr; }
    w.cap = 0;
    if (release) {  // give the set's memory back (a sub-batch did not fit: the next attempt is half the size)
        if (w.slp_arena) { HIPCK(hipFree(w.slp_arena)); w.slp_arena = nullptr; w.slp_arena_bytes = 0; }
        return;
    }
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
                        int brp_bits, hipStream_t st, int out_fmt) {
    launch_msm(scalars, table_view(which), false, out, n_groups, n_slices, out_stride, brp_bits, st, out_fmt);
}
// out_fmt: launch::FMT_JACQ (the default: 14 x 29-bit sums) or FMT_JACS (the signed 13 x 30-bit sums everything after the MSM computes
// in); `out` addresses points of that format
// tv: ONE snapshot of the table view (the builder thread may publish a wider table at any time); scalars_split: the producer
// has stored the scalars as balanced GLV halves already (only meaningful for a GLV table).  While a wider table is under
// construction its leading ready groups run on it and the rest on the complete table: two launches, one MSM stage.
void Engine::launch_msm(const void* scalars, const TableView& tv, bool scalars_split, void* out, int n_groups, int n_slices,
                        int out_stride, int brp_bits, hipStream_t st, int out_fmt) {
    const SharedTable* main = tv.main.get();
    const SharedTable* next = tv.next.get();
    int ready = 0;
    if (next) ready = std::min(n_groups, next->ready_groups.load(std::memory_order_acquire));
    if (!scalars_split) launch::glv_split(const_cast<void*>(scalars), (size_t)n_groups * n_slices * 64, st);  // in place: they feed nothing else
    if (ready > 0) launch_msm_range(scalars, *next, 0, ready, out, n_groups, n_slices, out_stride, brp_bits, st, out_fmt);
    if (ready < n_groups) launch_msm_range(scalars, *main, ready, n_groups - ready, out, n_groups, n_slices, out_stride, brp_bits, st, out_fmt);
}
// waves of the MSM launch over `msms` MSMs on a table of nominal width c, as launch_msm_range will schedule it (flat launches aside)
long Engine::msm_waves(long msms, int c) const {
    const long W = launch::glv_windows(c);
    if (msm_chunks_ > 0 || (msm_chunks_ < 0 && (msms * 4 + 63) / 64 >= (long)wave_slots_)) return msms * 4;  // four chunks per MSM: 64 MSMs x 4 waves per block
    if (msm_chunks_ < 0 && msm_split_ && (msms * 4 * W + 63) / 64 <= (long)wave_slots_ / 2) return (msms * 4 * W + 63) / 64;
    return (msms * 2 * W + 63) / 64;
}
// groups [g0, g0 + gcnt) of every slice on table t
void Engine::launch_msm_range(const void* scalars, const SharedTable& t, int g0, int gcnt, void* out, int n_groups, int n_slices,
                              int out_stride, int brp_bits, hipStream_t st, int out_fmt) {
    const launch::TabBlocks tb{(const void* const*)t.d_blocks, g0, gcnt};
    const int c = t.c;
    const long msms = (long)gcnt * n_slices;
    int mode = 1;  // a lane per (MSM, window)
    if (n_slices <= FLAT_MSM_MAX_SLICES && circ_max_ > 0) mode = 0;  // a handful of blobs: one block per MSM
    else if (msm_chunks_ >= 0) mode = msm_chunks_ == 0 ? 1 : 2;  // tests: 0 = the windowed kernel, anything else = four chunks per MSM
    else if (msm_split_ && (msms * 4 * launch::glv_windows(c) + 63) / 64 <= (long)wave_slots_ / 2) mode = 3;  // <= 16 blobs on eight windows: two lanes per window while that still leaves a SIMD per wave -- the chain of dependent additions is halved (16 blobs: 0.70 -> 0.49 ms; with two waves per SIMD, 17 .. 32 blobs, it was measured 0.05 ms SLOWER)
    else if ((msms * 4 + 63) / 64 >= (long)wave_slots_) {
        // The chip is full: four chunks per MSM, 16384 short waves dealt out as slots free up.  (A lane per MSM and a lane per GLV
        // half -- no folds, no barriers, exact rounds of long waves -- were measured 1-2 % SLOWER in rounds 3 and 4 and are gone.)
        mode = 2;
    }
    if (mode == 1 && msm_chunks_ < 0 && msm_split_) {
        // A lane per window fills one round of wave slots with 64 blobs on eight windows -- and with 56 on nine (the default tables):
        // the blobs beyond that used to open a second round of full-length waves on a nearly empty chip (64 blobs on the default
        // tables: MSM 1.66 ms against 1.13 on eight windows).  An overflow of at most a quarter round runs as its own launch: a block per
        // MSM for a handful of blobs, two lanes per window (half-length chains) beyond (tools/sweep_default_tables.sh: 64 blobs on the
        // default tables 3.12 -> 3.03 ms with the second form alone; 65 .. 80 blobs on eight windows 3.9 -> 3.5 ms).
        const long per_slice = (long)gcnt * 2 * launch::glv_windows(c), slots = (long)wave_slots_ * 64, lanes = per_slice * n_slices;
        const int n_a = (int)(slots / per_slice);
        if (lanes > slots && lanes * 4 <= slots * 5 && n_a >= 1 && n_a < n_slices) {
            const size_t pt = out_fmt == launch::FMT_JACS ? launch::SIZEOF_JACS : launch::SIZEOF_JACQ;
            launch::msm_glv(c, 1, scalars, tb, out, n_groups, n_a, 64, out_stride, brp_bits, beta_, st, out_fmt);
            const int n_b = n_slices - n_a;
            launch::msm_glv(c, (n_b <= FLAT_MSM_MAX_SLICES && circ_max_ > 0) ? 0 : 3, (const char*)scalars + (size_t)n_a * n_groups * 64 * sizeof(Fr), tb,
                            (char*)out + (size_t)n_a * pt, n_groups, n_b, 64, out_stride, brp_bits, beta_, st, out_fmt);
            return;
        }
    }
    launch::msm_glv(c, mode, scalars, tb, out, n_groups, n_slices, 64, out_stride, brp_bits, beta_, st, out_fmt);
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
    const bool linmap_mode = n > circ_max_;
    void* X = w.X;
    const SlpProgram* prog = nullptr;
    int mulc_coop_lanes = 0;  // > 0: so few blobs that the constant multiplications take several lanes per blob (launch::g1_slp_launch)
    if (linmap_mode) {
        int which = pick_slp_program(bp);
        // 33 .. 64 blobs: the constant multiplications run with two lanes per blob = two waves per operation (k_g1slp.hip), so the
        // compilation with 456 of them (912 waves, one per SIMD) replaces the one with 712 that a lane per blob takes
        if (slp_force_ < 0 && bp == 64 && n > 32 && launch::coop_points_max() > 0) which = SLP_DEPTH_456;
        mulc_coop_lanes = n <= 32 ? n : (bp == 64 && which == SLP_DEPTH_456) ? n : 0;
        prog = &slp_program(which);
        const size_t need = (size_t)prog->n_slots * bp * launch::SIZEOF_JACQ;  // (sized for the larger of the two point formats)
        if (need > w.slp_arena_bytes) {
            if (w.slp_arena) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(w.slp_arena)); w.slp_arena = nullptr; w.slp_arena_bytes = 0; }
            HIPCK(hipMalloc(&w.slp_arena, need));
            w.slp_arena_bytes = need;
        }
        X = w.slp_arena;
    }
    // The arena's point format: every kernel between the scalars and the proofs' bytes -- MSM, constant multiplications, additions,
    // compression; a lane, two or four lanes per blob -- is a kernel of the signed 13 x 30-bit field, and the arena holds its points
    // (JacS, 156 B) as they are: ONE Fp representation under the whole path at every batch size, no conversions.
    // The 14 x 29-bit field (JacQ) is the cross-check ETH_KZG_AMD_ARENA_SIGNED=0 (and what verification and EIP-4844 compute in).
    const int fmt = arena_signed_ ? launch::FMT_JACS : launch::FMT_JACQ;
    const size_t pt = fmt == launch::FMT_JACS ? launch::SIZEOF_JACS : launch::SIZEOF_JACQ;
    const TableView tv = tv_pre ? *tv_pre : table_view(TAB_FK);  // one snapshot for the scalars' form AND the MSM that reads them
    if (!tv_pre) {
        const int mk1 = mark_begin(ST_FK20_SCALARS, st);
        launch::fk20_scalars(n, w.coeffs, w.scalars, d_w29_, linmap_mode ? half_ : inv128_, segs, segs == 2 ? two_segments : seg_shift_, st);
        mark_end(mk1, 1, st);
    }
    if (phase != PROOFS_ALL) {
        // Two MSM launches around a cut (a multiple of 64 blobs): the head is issued while the rest of the batch is still on the
        // link.  The scalars are blob-major, the outputs lane-major with stride bp: a sub-range is a pointer offset on both.
        if (!tv_pre || !linmap_mode || segs != 1 || msm_cut <= 0 || msm_cut >= n || msm_cut % 64) throw std::logic_error("run_proofs_from_coeffs: bad MSM cut");
        // (the head on a stream of its own, next to the later sub-batches' light stages and joined before the linear map, was
        // measured too: no gain, profiles/archive/r4_early_msm_ab.log)
        if (phase == PROOFS_HEAD) {
            launch::g1_set_inf(X, (size_t)128 * bp, st, fmt);
            launch_msm(w.scalars, tv, true, X, 128, msm_cut, bp, 0, st, fmt);
            return;
        }
        launch_msm((char*)w.scalars + (size_t)msm_cut * 128 * 64 * sizeof(Fr), tv, true, (char*)X + (size_t)msm_cut * pt, 128, n - msm_cut, bp, 0, st, fmt);
    }
    if (phase == PROOFS_ALL) launch::g1_set_inf(X, (size_t)128 * bp, st, fmt);
    const int mk2 = mark_begin(ST_MSM_FIXED, st);
    if (phase == PROOFS_ALL) launch_msm(w.scalars, tv, true, X, 128, segs * n, bp, 0, st, fmt);
    mark_end(mk2, 1, st);
    if (linmap_mode) {
        const int mk3 = mark_begin(ST_G1_LINMAP, st);
        int n_launches = 0;
        for (auto& L : prog->launches)
            launch::g1_slp_launch(L.kind, w.slp_arena, bp, (const uint32_t*)prog->d_words + (size_t)L.first * 4, L.count, prog->d_naf, beta_, st, 0,
                                  mulc_coop_lanes, fmt, n);
        n_launches = (int)prog->launches.size();
        mark_end(mk3, n_launches, st);
        const int mk4 = mark_begin(ST_COMPRESS, st);
        launch::g1_compress((const char*)w.slp_arena + (size_t)128 * bp * pt, d_proofs, 128, bp, n, st, fmt);
        mark_end(mk4, 1, st);
        return;
    }
    if (n <= circ_max_) {  // a handful of blobs: the two transforms as one circulant product (k_g1circ.hip)
        if (!w.circ_table) HIPCK(hipMalloc(&w.circ_table, launch::g1_circ_table_bytes(circ_max_, circ_T_)));
        const int mk5 = mark_begin(ST_G1_IFFT, st);
        launch::g1_circ128(w.X, bp, n, segs, w.circ_table, circ_T_, d_circ_terms_, circ_per_lane_, beta_, st, fmt);
        mark_end(mk5, 2, st);
    } else throw std::logic_error("run_proofs_from_coeffs: a batch above the circulant form needs the compiled linear map");
    const int mk10 = mark_begin(ST_COMPRESS, st);
    launch::g1_compress(w.X, d_proofs, 128, bp, n, st, fmt);
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
    // ... unless the MSM launch that follows puts a wave on (nearly) every SIMD and no second one: a block of the cells kernel needs a
    // whole CU (144 KB of LDS, 16 waves) and then neither starts nor lets the map's waves in -- measured: 28 and 32 blobs 2.57 ms with
    // the side stream against 2.20 ms without, 24 blobs 2.12 against 2.21, 40 blobs 2.72 against 2.82 (tools/sweep_small_batches.sh)
    bool side = d_cells && d_proofs && n <= SIDE_CELLS_MAX && w.copy && !profiling_;
    if (side && n > FLAT_MSM_MAX_SLICES) {
        const long simds = wave_slots_ / 2, waves = msm_waves(128L * n, table_view(TAB_FK).c);
        if (waves > simds * 4 / 5 && waves <= simds) side = false;
    }
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
        // A batch larger than the scratch the HBM still holds beside the window tables (0.6 MB per blob: coefficients, MSM scalars, the
        // linear map's arena) runs as sub-batches on the same stream: at most device_batch_max_ blobs each (4096: twice what fills the
        // chip), and half of that again whenever an allocation of the work set fails -- down to 64 blobs, below which the failure is the
        // caller's.  Stages already enqueued for a sub-batch that is then retried smaller are recomputed (same outputs).
        int sub = std::min(n, device_batch_max_);
        for (int b0 = 0; b0 < n;) {
            const int nb = std::min(sub, n - b0);
            try {
                enqueue_compute(w, nb, d_blobs + (size_t)b0 * BYTES_PER_BLOB, d_cells ? d_cells + (size_t)b0 * N_CELLS * BYTES_PER_CELL : nullptr,
                                d_proofs ? d_proofs + (size_t)b0 * N_CELLS * 48 : nullptr, st, nullptr);
            } catch (const HipError& e) {
                if (!e.out_of_memory() || sub <= 64) throw;
                (void)hipGetLastError();
                HIPCK(hipStreamSynchronize(st));  // kernels of the failed attempt may still read the set's buffers
                ensure_workspace(w, 0, /*release=*/true);
                sub = std::max(64, ((sub / 2 + 63) / 64) * 64);
                continue;
            }
            if (h_status) HIPCK(hipMemcpyAsync(h_status + b0, w.status, nb * sizeof(int), hipMemcpyDeviceToHost, st));
            b0 += nb;
        }
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
    if (n > device_batch_max_) {  // sub-batches (0.5 MB of scratch per blob), as compute_cells_and_kzg_proofs_device
        for (int b0 = 0; b0 < n; b0 += device_batch_max_) {
            const int nb = std::min(device_batch_max_, n - b0);
            const int rc = blob_to_kzg_commitment_device(nb, d_blobs + (size_t)b0 * BYTES_PER_BLOB, d_commitments + (size_t)b0 * 48,
                                                         h_status ? h_status + b0 : nullptr, st, sync);
            if (rc) return rc;
        }
        return OK;
    }
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
        const int fmt = arena_signed_ ? launch::FMT_JACS : launch::FMT_JACQ;  // (the points' form: run_proofs_from_coeffs)
        launch::g1_set_inf(d_X_, (size_t)64 * bp, st, fmt);
        launch_msm(d_canon_, TAB_SRS, d_X_, 64, n, bp, 0, st, fmt);
        launch::g1_sum_positions(d_X_, 64, bp, n, st, fmt);
        launch::g1_compress(d_X_, d_commitments, 1, bp, n, st, fmt);
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
    const bool trace = knobs_.trace;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // with ETH_KZG_AMD_TRACE: report the steps of a call that took longer than 50 ms (which HIP call waited, and for how long)
    const double slow_ms = knobs_.trace ? 50.0 : 0.0;
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
                if (knobs_.host_threads) t = knobs_.host_threads;
                host_pool_.reset(new HostPool(t, [d = dev_] { (void)hipSetDevice(d); }));
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
            const bool early_scalars = proofs && ns > circ_max_;
            const TableView tv_call = table_view(TAB_FK);
            // A batch of several rounds of the chip: the MSMs of the first 256 blobs are launched as soon as THEIR scalars exist,
            // 0.7 ms into the call, and run while the other 235 MB are on the link; the MSMs of the rest follow.  (An MSM launch costs
            // ~1.3 ms beyond its share of the work -- the last waves of a launch -- so the head is as small as will still cover
            // the uploads: 256 blobs = 6 ms; cut at 512: +1.4 ms, one launch after the last upload: +2.0 ms, profiles/archive/r4_early_msm_ab.log.)
            constexpr int early_msm_blobs = 256;
            int msm_cut = 0;
            if (early_scalars && early_msm_blobs > 0 && ns >= 4 * early_msm_blobs)
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
                                         d_w29_, half_, 1, seg_shift_, w.stream);
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

}  // namespace kzg
