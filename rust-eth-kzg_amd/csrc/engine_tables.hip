// Window tables of a context: the process-wide registry (one table per GPU, kind and width, shared by contexts), the builder that
// allocates a table in pieces and fills it group by group, the views the MSM launches snapshot, the progressive start.
// Replaces precompute_points / FixedBaseMSMPrecompWindow::new (crates/cryptography/bls12_381/src/fixed_base_msm_window.rs:69-82).
#include "engine_internal.hpp"

namespace kzg {

static std::mutex g_tables_mu;  // the registry below: held for look-ups and inserts only, never across a build
static std::map<std::tuple<int, int, int>, std::weak_ptr<Engine::SharedTable>> g_tables;  // (device, kind, width)
// one builder of WIDE tables at a time per GPU (the helper threads of several contexts of one device queue here; the contexts of a
// device list -- c_api.cpp, ETH_KZG_AMD_DEVICES -- build side by side, one per device)
static std::mutex& build_mutex_of(int device) {
    static std::mutex reg;
    static std::map<int, std::unique_ptr<std::mutex>> mu;
    std::lock_guard<std::mutex> lk(reg);
    auto& m = mu[device];
    if (!m) m.reset(new std::mutex);
    return *m;
}


// Fill a table the caller has just created: pieces are allocated a chunk of groups ahead of the builder kernels, every
// finished chunk is published through ready_groups.  Returns false (state 2, `why` set) if the device cannot hold it;
// throws BuildCancelled when `cancel` is raised (state 2 as well).  The groups that are ready stay usable either way.
// gentle: the build shares the GPU with callers on the start tables (progressive start): one group per launch -- 512 waves, one
// per SIMD on half the chip's SIMDs, so a caller's kernels find free SIMDs at once instead of waiting for 1,500 builder
// waves that run 17 ms -- at twice the build time, which the allocation of the pieces hides anyway.
static bool fill_table(Engine::SharedTable& t, const void* bases, hipStream_t st, const std::atomic<bool>* cancel, bool gentle = false) {
    const bool trace = t.trace_allocs;
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int c = t.c, nb = t.nb;
    const size_t per_group = launch::table_glv_entries(c, 1, nb);
    const size_t scratch_per_entry = 168;
    int chunk = (int)((9ull << 30) / (per_group * scratch_per_entry));
    if (chunk < 1) chunk = 1;
    if (chunk > t.n_groups) chunk = t.n_groups;
    if (gentle) chunk = std::max(1, std::min(chunk, 512 / (nb * launch::glv_windows(c))));  // <= ~512 builder waves in flight (a wave per (base, window): 64 x W per group)
    const size_t side_bytes = launch::table_glv_side_bytes(c, chunk, nb);
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
            if (!launch::build_table_glv(c, b, blocks, scratch, side, g, nb, d_err, st)) throw std::runtime_error("GLV table width not built in");
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
                       t.pieces.size(), t.alloc_us.load() / 1e3, t.alloc_us_max.load() / 1e3);
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

// One snapshot per MSM launch; publish / retire / reap: host_sync.hpp (Published), driven under ThreadSanitizer on the CPU
Engine::TableView Engine::table_view(TableSel which) const {
    if (primary_) return primary_->table_view(which);  // an engine lane reads through to the context's engine
    const auto s = pub_[which].snapshot();
    TableView v;
    v.main = s.main;
    v.next = s.next;
    v.c = s.main ? s.main->c : 0;
    v.bytes = s.main ? s.main->bytes : 0;
    return v;
}
// main: the complete table calls run on; next: a wider one under construction whose ready groups are used already; what falls out
// of the view is retired (kernels in flight may still read it) until nobody refers to it
void Engine::publish(TableSel which, const std::shared_ptr<SharedTable>& main, const std::shared_ptr<SharedTable>& next) {
    pub_[which].publish(main, next);
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
            out4[0] += t->alloc_us.load(std::memory_order_relaxed) / 1e3;
            out4[1] = std::max(out4[1], t->alloc_us_max.load(std::memory_order_relaxed) / 1e3);
            out4[2] += (double)t->piece_count.load(std::memory_order_acquire);
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
    if (!use_precomp_) {  // UsePrecomp::No: the sixteen-window tables (1.6 + 0.8 GB: what a use_precomp = true context STARTS on), nothing else to build
        auto srs = obtain_table(dev_, 3, 8, d_srs_, 64, stream_), fk = obtain_table(dev_, 2, 8, d_fk_bases_, 128, stream_);
        if (!srs || !fk) throw std::runtime_error("not enough device memory for the window tables");
        publish(TAB_SRS, srs, nullptr);
        publish(TAB_FK, fk, nullptr);
        std::lock_guard<std::mutex> lk2(tab_mu_);
        tables_state_ = 1;
        return;
    }
    const bool progressive = knobs_.progressive;
    if (!progressive) {
        build_final_tables();
        if (!table_view(TAB_FK).main || !table_view(TAB_SRS).main) throw std::runtime_error("not enough device memory for the window tables: " + tables_error_);
        return;
    }
    // Progressive start (the reference's "Initialize context" bench, benchmark-mt.rs:103-113): serve from small tables at once --
    // or from whatever wider table another context of this process already holds -- and build the wide ones on a helper thread.
    {
        std::shared_ptr<SharedTable> fk, srs;
        for (int w : launch::GLV_WIDTHS)
            if (!fk) fk = obtain_table(dev_, 2, w, nullptr, 128, stream_, /*only_if_live=*/true);
        if (!fk) fk = obtain_table(dev_, 2, 8, d_fk_bases_, 128, stream_);
        for (int w : launch::GLV_WIDTHS)
            if (!srs) srs = obtain_table(dev_, 3, w, nullptr, 64, stream_, true);
        if (!srs) srs = obtain_table(dev_, 3, 8, d_srs_, 64, stream_);
        if (!fk || !srs) throw std::runtime_error("not enough device memory for the start window tables");
        publish(TAB_FK, fk, nullptr);
        publish(TAB_SRS, srs, nullptr);
    }
    progressive_build_ = true;  // start_builder(), the constructor's LAST step, starts the helper thread
}

// The helper thread of a progressive start.  Called when everything else of the context stands (a constructor that
// throws after this point would unwind a joinable std::thread -- std::terminate -- and leave `this` in g_engines).
void Engine::start_builder() {
    if (!progressive_build_ || primary_) return;
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        static bool registered = false;
        if (!registered) { atexit(stop_all_builders_at_exit); registered = true; }
        g_engines.push_back(this);
    }
    builder_ = std::thread([this] {
        (void)hipSetDevice(dev_);
        build_final_tables();
    });
}

// The wide tables, widest first, each taken from the process-wide registry if another context of this GPU holds it already. Both are
// GLV tables (the endomorphism halves the memory per window bit: a plain width-14 FK20 table costs 163 GB for 19 additions, the
// plain width-13 commitment table of rounds 2-4 cost 43 GB for the 20 additions that ten GLV windows give in 14.5 GB), their W
// windows of mixed widths that cover the 128-bit half exactly (launch.hpp):
//   FK20 (128 groups):       8 windows 206 GB (16 gathered additions per base) . 9: 71 GB (18) . 10: 29 GB (20) . 11: 14.5 GB (22) . 16: 1.6 GB (32)
//   commitments (64 groups): 9 windows 35 GB (18) . 10: 14.5 GB (20) . 11: 7.3 GB (22) . 16: 0.8 GB (32)
// bounded by what the HBM still holds (another process may own part of it) and by ETH_KZG_AMD_TABLE_GB (both tables together; the
// commitment table gets at most a third of it; default Engine::DEFAULT_TABLE_BUDGET_GB = 108 = nine windows each).  A table this
// thread creates is published as the view's `next` BEFORE it is filled, so the MSMs use its groups as they become ready.  Never
// throws: a failure leaves the context on the tables it has.
void Engine::build_final_tables() {
    int state = 1;
    std::string why;
    // wider than `now`, from the registry or built here; attach = publish as `next` while it is filled
    // A table this context had attached as `next` may have been ABANDONED by the context that was filling it (freed mid-build):
    // its pieces must go before another 206 GB can be allocated.  The invariant: a table lives as long as a shared_ptr to it does
    // -- the views, `retired_`, and the snapshot (TableView) every MSM launch holds until its kernels are enqueued; hipFree itself
    // waits for kernels in flight.  So: unpublish it (it moves to `retired_`), wait until `retired_` holds the ONLY reference (a
    // caller's snapshot lives microseconds), and drop it HERE, on the builder thread -- the hundreds of hipFree calls of a 200 GB
    // table never run on a caller's hot path, and no device-wide synchronisation waits behind other callers' queued work
    // (the former hipDeviceSynchronize + sleep(20 ms) was a timing heuristic).  A snapshot that outlives the bound
    // leaves the table in `retired_` for the next reaping; the allocation below then falls back to a narrower table.
    auto drop_abandoned = [&](TableSel sel) {
        {
            const TableView cur = table_view(sel);  // (this snapshot is itself a reference: it ends with the block)
            if (cur.next && cur.next->state.load() == 2) publish(sel, cur.main, nullptr);
        }
        // ~SharedTable (hipFree of every piece) runs inside reap(): outside the lock, on this thread
        int waiting = 0;
        for (int spin = 0; spin < 2000 && (waiting = pub_[sel].reap([](const SharedTable& r) { return r.state.load() == 2; })) > 0; spin++)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        return waiting == 0;  // false: somebody held a snapshot of an abandoned table for two seconds; its memory is still taken
    };
    auto widen = [&](TableSel sel, int kind, int w, const void* bases, int n_groups) -> std::shared_ptr<SharedTable> {
        if (cancel_build_.load()) throw BuildCancelled{};
        if (!drop_abandoned(sel)) {  // do not allocate a wide table on top of memory an abandoned one still holds
            why = "an abandoned table is still referenced: its memory is not free yet";
            return nullptr;
        }
        bool created = false;
        auto t = find_or_create_table(dev_, kind, w, n_groups, &created);
        const TableView cur = table_view(sel);
        const bool same_form = cur.main && cur.main->n_groups == t->n_groups;
        if (t->state.load() == 0 && same_form) publish(sel, cur.main, t);  // its ready groups serve at once
        if (created) {
            t->trace_allocs = knobs_.trace;
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
        if (knobs_.trace) fprintf(stderr, "[context] @%.0f ms: code objects preloaded in %.0f ms\n", trace_clock_ms(), trace_clock_ms() - p0);
        // Another context of the process may be filling the wide tables right now (its helper thread holds the device's build mutex until it is
        // done): this context uses their ready groups meanwhile instead of sitting on its start tables for the other's build.
        auto attach_growing = [&] {
            for (TableSel sel : {TAB_SRS, TAB_FK}) {
                const TableView cur = table_view(sel);
                if (!cur.main || (cur.next && cur.next->state.load() == 0)) continue;
                std::shared_ptr<SharedTable> growing;
                if (sel == TAB_SRS) {
                    for (int w : launch::GLV_WIDTHS)
                        if (!growing && w > cur.c) { auto t = find_table(dev_, 3, w); if (t && t->state.load() == 0) growing = t; }
                } else {
                    for (int w : launch::GLV_WIDTHS)
                        if (!growing && w > cur.c) { auto t = find_table(dev_, 2, w); if (t && t->state.load() == 0) growing = t; }
                }
                if (growing && growing->n_groups == cur.main->n_groups) publish(sel, cur.main, growing);
            }
        };
        std::unique_lock<std::mutex> lk(build_mutex_of(dev_), std::defer_lock);  // one builder of wide tables at a time per GPU
        while (!lk.try_lock()) {
            attach_growing();
            if (cancel_build_.load()) throw BuildCancelled{};
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
        if (cancel_build_.load()) throw BuildCancelled{};
        const double budget = table_budget_gb_ > 0 ? table_budget_gb_ * 1e9 : 1e18;
        const TableView srs_now = table_view(TAB_SRS), fk_now = table_view(TAB_FK);
        std::shared_ptr<SharedTable> srs;
        for (int w : launch::GLV_WIDTHS) {
            if (srs) break;
            if (w == 16) continue;  // (eight windows for the commitments would be 103 GB: the FK20 table has the better use for them)
            if (srs_now.main && w <= srs_now.c) break;  // nothing wider than what is in use fits
            if ((double)glv_table_bytes(w, 64) > std::max(budget / 3, 0.9e9)) continue;  // (a third of the default 108 GB = the nine-window table)
            srs = widen(TAB_SRS, 3, w, d_srs_, 64);
        }
        if (srs) publish(TAB_SRS, srs, nullptr);
        const double left = budget - (double)table_view(TAB_SRS).bytes;
        std::shared_ptr<SharedTable> fk;
        for (int w : launch::GLV_WIDTHS) {
            if (fk) break;
            if (want_glv_c_ && w != want_glv_c_) continue;
            if (fk_now.main && w <= fk_now.c) break;
            if ((double)glv_table_bytes(w) > std::max(left, 1.7e9)) continue;
            fk = widen(TAB_FK, 2, w, d_fk_bases_, 128);
        }
        if (fk) publish(TAB_FK, fk, nullptr);
        // The tables this context STARTED on (1.6 + 0.8 GB, complete: state 1) were retired by the two publishes above and are not
        // counted in the stated budget: they go as soon as no MSM launch's snapshot refers to them -- a snapshot lives for
        // the microseconds of an enqueue -- here, on the builder thread (hipFree waits for kernels in flight).  A start table that
        // another context of the device still runs on has other owners and stays.
        for (TableSel sel : {TAB_SRS, TAB_FK})
            for (int spin = 0; spin < 200 && pub_[sel].reap([](const SharedTable&) { return true; }) > 0; spin++)
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
    } catch (const BuildCancelled&) {
        state = 2;
        why = "cancelled: the context is being freed";
    } catch (const std::exception& e) {
        (void)hipGetLastError();
        state = 2;
        why = e.what();
    }
    if (state == 2 && knobs_.trace) fprintf(stderr, "[context] wide tables not built: %s\n", why.c_str());
    std::lock_guard<std::mutex> lk(tab_mu_);
    tables_state_ = state;
    tables_error_ = why;
    tab_cv_.notify_all();
}

}  // namespace kzg
