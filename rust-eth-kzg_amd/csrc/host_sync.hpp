// The host-side concurrency protocols of a context, free of HIP: one context is shared by many host threads (the reference's
// usage: bindings/node/src/lib.rs:35,92-299, bindings/java/.../LibEthKZGTest.java:28-37), and what makes that work --
//   HostPool       a persistent pool of helper threads (hashing, staging, pairings)
//   Combiner       concurrent single verifications combined into passes: leader election, follower wake-up, failure fan-out
//   SlotSet        pass slots: take a free one, prefer one whose partner resource is idle, else queue round-robin
//   LanePool       engine lanes: the primary when idle, an idle auxiliary, a new one built OUTSIDE the lock, else queue
//   Published      a value shared by readers' snapshots and replaced by a builder: publish / snapshot / retire / reap, with
//                  group-by-group progress (ready count) that readers take with acquire semantics
// -- lives here so that it can be driven WITHOUT a GPU: tests/c/test_host_sync.cpp runs 32 threads of mixed operations against
// fake device passes (that sleep, throw, or report wrong proofs) while contexts are created and freed, under ThreadSanitizer and
// under AddressSanitizer + UBSan (tests/test_sanitizers.py).  engine.hip / verify_many.hip use exactly these classes.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace kzg {

// ---------------------------------------------------------------------------------------------------------------------------
// A persistent pool of helper threads.  thread_init runs first on every worker (the engine binds its GPU there).
class HostPool {
public:
    explicit HostPool(int threads, std::function<void()> thread_init = nullptr) {
        for (int i = 0; i < threads; i++) th_.emplace_back([this, thread_init] {
            if (thread_init) thread_init();
            run();
        });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    HostPool(const HostPool&) = delete;
    void submit(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(std::move(fn));
        }
        cv_.notify_one();
    }
    int threads() const { return (int)th_.size(); }

private:
    void run() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // stop_ and nothing left: queued work is always finished first
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    std::vector<std::thread> th_;
    std::deque<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};
// fork-join over [0, n) on the calling thread and up to `threads` - 1 workers of a persistent pool (work stealing by an atomic
// counter; the first exception is rethrown).  Round 3 started fresh std::threads here, five times per pass: ~0.5 ms of thread
// churn on a pass that should last 4.
template <class F>
inline void parallel_for(int n, int threads, HostPool* pool, F fn) {
    if (n <= 0) return;
    if (threads > n) threads = n;
    if (threads <= 1 || !pool) { for (int i = 0; i < n; i++) fn(i); return; }
    struct Shared {
        std::atomic<int> next{0};
        std::atomic<unsigned> state{0};  // helpers inside body | CLOSED: the caller has left its own share and admits no more
        std::exception_ptr err;
        std::mutex mu;
        std::condition_variable cv;
    };
    constexpr unsigned CLOSED = 0x80000000u;
    auto sh = std::make_shared<Shared>();  // outlives this frame: a helper the pool gets to late finds the door closed and leaves
    auto body = [sh, n, &fn] {
        try {
            for (int i; (i = sh->next.fetch_add(1)) < n;) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> lk(sh->mu);
            if (!sh->err) sh->err = std::current_exception();
            sh->next.store(n);
        }
    };
    for (int t = 0; t < threads - 1; t++)
        pool->submit([sh, body, CLOSED] {
            unsigned s = sh->state.load();
            do {
                if (s & CLOSED) return;  // too late: nothing of the caller's frame may be touched any more
            } while (!sh->state.compare_exchange_weak(s, s + 1));
            body();
            if (sh->state.fetch_sub(1) == (CLOSED | 1u)) { std::lock_guard<std::mutex> lk(sh->mu); sh->cv.notify_all(); }
        });
    body();
    {   // no new helper may enter; those inside are waited for (fn and the caller's captures die with this frame)
        std::unique_lock<std::mutex> lk(sh->mu);
        if (sh->state.fetch_or(CLOSED) != 0) sh->cv.wait(lk, [&] { return sh->state.load() == CLOSED; });
    }
    if (sh->err) std::rethrow_exception(sh->err);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Combiner: callers that arrive while passes are running queue their requests; up to max_leaders of them become LEADERS, each taking
// everything queued so far (its own request included) as one batch and running it outside the lock; the others sleep until their
// request is marked done.  A leader must release its followers whatever happens inside the pass: run_pass's exceptions are caught and
// turned into fail(batch, what).  Request needs two bool members the combiner owns: done, taken.
template <class Request>
class Combiner {
public:
    explicit Combiner(int max_leaders) : max_leaders_(max_leaders) {}
    // run_pass(std::vector<Request*>& batch): fills every request's result.  fail(batch, const std::string& what): marks them failed.
    template <class RunPass, class Fail>
    void submit(Request& me, RunPass&& run_pass, Fail&& fail) {
        std::unique_lock<std::mutex> lk(mu_);
        queue_.push_back(&me);
        while (!me.done) {
            if (me.taken || running_ >= max_leaders_) { cv_.wait(lk); continue; }
            running_++;
            std::vector<Request*> batch;
            batch.swap(queue_);
            for (Request* r : batch) r->taken = true;
            lk.unlock();
            try {
                run_pass(batch);
            } catch (const std::exception& e) {
                fail(batch, std::string(e.what()));
            } catch (...) {
                fail(batch, std::string("unknown failure"));
            }
            lk.lock();
            for (Request* r : batch) r->done = true;
            running_--;
            cv_.notify_all();
        }
    }
    int running() const { std::lock_guard<std::mutex> lk(mu_); return running_; }

private:
    mutable std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Request*> queue_;
    int running_ = 0;
    const int max_leaders_;
};

// ---------------------------------------------------------------------------------------------------------------------------
// SlotSet: N resources with a lock each.  acquire(): a free slot that `preferred(k)` likes, else any free slot, else queue on one in
// turn.  The lock travels with the lease.
template <int N>
class SlotSet {
public:
    struct Lease {
        int index = -1;
        std::unique_lock<std::mutex> lock;
    };
    template <class Pred>
    Lease acquire(Pred&& preferred) {
        Lease L;
        for (int pass = 0; pass < 2 && L.index < 0; pass++)
            for (int k = 0; k < N && L.index < 0; k++) {
                if (pass == 0 && !preferred(k)) continue;
                std::unique_lock<std::mutex> t(mu_[k], std::try_to_lock);
                if (t.owns_lock()) { L.lock = std::move(t); L.index = k; }
            }
        if (L.index < 0) {
            L.index = (int)(rr_.fetch_add(1) % (unsigned)N);
            L.lock = std::unique_lock<std::mutex>(mu_[L.index]);
        }
        return L;
    }
    Lease acquire() { return acquire([](int) { return true; }); }
    std::mutex& mutex(int k) { return mu_[k]; }

private:
    std::mutex mu_[N];
    std::atomic<unsigned> rr_{0};
};
// "is this mutex free right now?" -- a peek, not a lease
inline bool mutex_is_free(std::mutex& m) {
    if (!m.try_lock()) return false;
    m.unlock();
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// LanePool: lanes of T (T has a public std::mutex lane_busy_).  lease(primary, make): the primary when it is idle; an idle auxiliary;
// a new auxiliary -- built by make() OUTSIDE the pool's lock (seconds of set-up must not block callers that could queue on an
// existing lane), at most max_lanes - 1 of them, a failed construction just queues; else wait on a lane, round-robin.
template <class T>
class LanePool {
public:
    struct Lease {
        T* e = nullptr;
        std::unique_lock<std::mutex> busy;
    };
    template <class Make>
    Lease lease(T* primary, int max_lanes, bool primary_only, Make&& make) {
        Lease L;
        L.busy = std::unique_lock<std::mutex>(primary->lane_busy_, std::try_to_lock);
        if (L.busy.owns_lock() || primary_only || max_lanes <= 1) {
            if (!L.busy.owns_lock()) L.busy.lock();
            L.e = primary;
            return L;
        }
        bool create = false;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto& a : aux_) {
                L.busy = std::unique_lock<std::mutex>(a->lane_busy_, std::try_to_lock);
                if (L.busy.owns_lock()) { L.e = a.get(); return L; }
            }
            if ((int)aux_.size() + pending_ + 1 < max_lanes) { pending_++; create = true; }
        }
        if (create) {
            std::unique_ptr<T> fresh;
            try {
                fresh = make();
            } catch (...) {
                fresh.reset();  // no room for another lane: queue on an existing one
            }
            std::lock_guard<std::mutex> lk(mu_);
            pending_--;
            if (fresh) {
                L.busy = std::unique_lock<std::mutex>(fresh->lane_busy_);
                L.e = fresh.get();
                aux_.push_back(std::move(fresh));
                return L;
            }
        }
        T* pick = primary;
        {
            std::lock_guard<std::mutex> lk(mu_);
            const unsigned k = rr_.fetch_add(1) % (unsigned)(aux_.size() + 1);
            if (k > 0) pick = aux_[k - 1].get();
        }
        L.busy = std::unique_lock<std::mutex>(pick->lane_busy_);
        L.e = pick;
        return L;
    }
    // the owner's destructor: lanes are destroyed before the primary's own resources
    void clear() {
        std::lock_guard<std::mutex> lk(mu_);
        aux_.clear();
    }
    size_t size() const { std::lock_guard<std::mutex> lk(mu_); return aux_.size(); }

private:
    mutable std::mutex mu_;
    std::vector<std::unique_ptr<T>> aux_;
    int pending_ = 0;
    std::atomic<unsigned> rr_{0};
};

// ---------------------------------------------------------------------------------------------------------------------------
// Published<T>: `main` is the complete value callers run on, `next` a wider one under construction whose leading parts are usable
// (T::ready_groups, an atomic the builder raises with release and readers load with acquire).  Readers take a SNAPSHOT (two
// shared_ptr copies under the lock) and keep it for one launch; publish() replaces the pair and moves what fell out to `retired`
// (a launch in flight may still read it); reap(dead) drops retired values nobody refers to any more ON THE CALLING THREAD -- the
// builder's -- so that a 200 GB table is never destroyed on a caller's hot path.
template <class T>
class Published {
public:
    struct View {
        std::shared_ptr<T> main, next;
    };
    View snapshot() const {
        std::lock_guard<std::mutex> lk(mu_);
        return View{main_, next_};
    }
    void publish(const std::shared_ptr<T>& main, const std::shared_ptr<T>& next) {
        std::lock_guard<std::mutex> lk(mu_);
        if (main_ && main_ != main) retired_.push_back(main_);
        if (next_ && next_ != next && next_ != main) retired_.push_back(next_);
        main_ = main;
        next_ = next;
    }
    // retired values for which dead(value) holds and that only the retired list still refers to; returns how many are still waited for
    template <class Pred>
    int reap(Pred&& dead) {
        std::vector<std::shared_ptr<T>> gone;
        int waiting = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto it = retired_.begin(); it != retired_.end();) {
                if (!dead(**it)) { ++it; continue; }
                if (it->use_count() == 1) { gone.push_back(std::move(*it)); it = retired_.erase(it); }
                else { waiting++; ++it; }
            }
        }
        gone.clear();  // destructors run here, outside the lock, on the caller of reap()
        return waiting;
    }
    size_t retired() const { std::lock_guard<std::mutex> lk(mu_); return retired_.size(); }

private:
    mutable std::mutex mu_;
    std::shared_ptr<T> main_, next_;
    std::vector<std::shared_ptr<T>> retired_;
};

// ---------------------------------------------------------------------------------------------------------------------------
// A context over a DEVICE LIST (c_api.cpp: ETH_KZG_AMD_DEVICES / eth_kzg_amd_das_context_new_on_devices): one engine per GPU behind
// the reference's unchanged symbols.  The reference's hosts create ONE context and call it from many threads
// (bindings/node/src/lib.rs:35,75; bindings/c/src/lib.rs:79-92); this is what spreads those calls over the node.
//   DevicePicker     single calls: the device with the least work in flight (weights = units of the call: blobs, cells), ties broken
//                    round-robin so that equal loads do not all land on device 0; the ticket gives the weight back when the call ends
//   fan_out_slices   batched calls: contiguous slices [n d / D, n (d + 1) / D) -- the split of rust-eth-kzg_amd/sharding.py and of
//                    eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi --, one thread per non-empty slice (slice 0 on the caller),
//                    every slice runs to its end whatever the others do, the error of the LOWEST failing device is the call's
class DevicePicker {
public:
    explicit DevicePicker(int n_devices) : load_(n_devices > 0 ? n_devices : 1) {
        for (auto& l : load_) l.store(0, std::memory_order_relaxed);
    }
    class Ticket {
    public:
        Ticket(DevicePicker* p, int device, uint64_t weight) : p_(p), device_(device), weight_(weight) {}
        Ticket(Ticket&& o) noexcept : p_(o.p_), device_(o.device_), weight_(o.weight_) { o.p_ = nullptr; }
        Ticket(const Ticket&) = delete;
        Ticket& operator=(const Ticket&) = delete;
        ~Ticket() { if (p_) p_->load_[device_].fetch_sub(weight_, std::memory_order_acq_rel); }
        int device() const { return device_; }
    private:
        DevicePicker* p_;
        int device_;
        uint64_t weight_;
    };
    Ticket pick(uint64_t weight = 1) {
        const int n = (int)load_.size();
        if (n == 1) { load_[0].fetch_add(weight, std::memory_order_acq_rel); return Ticket(this, 0, weight); }
        // Two callers may read the same minimum and pick the same device: the loads are a hint, not a reservation -- the engines
        // serialise or overlap calls themselves (lanes, work sets).  The rotating start spreads callers that see equal loads.
        const unsigned start = rr_.fetch_add(1, std::memory_order_relaxed);
        int best = (int)(start % (unsigned)n);
        uint64_t best_load = load_[best].load(std::memory_order_acquire);
        for (int k = 1; k < n && best_load != 0; k++) {
            const int d = (int)((start + (unsigned)k) % (unsigned)n);
            const uint64_t l = load_[d].load(std::memory_order_acquire);
            if (l < best_load) { best = d; best_load = l; }
        }
        load_[best].fetch_add(weight, std::memory_order_acq_rel);
        return Ticket(this, best, weight);
    }
    uint64_t load(int device) const { return load_[device].load(std::memory_order_acquire); }
    int devices() const { return (int)load_.size(); }

private:
    std::vector<std::atomic<uint64_t>> load_;
    std::atomic<unsigned> rr_{0};
};

inline uint64_t slice_begin(uint64_t n, int n_devices, int d) { return (uint64_t)((unsigned __int128)n * (unsigned)d / (unsigned)n_devices); }

// run(device, lo, hi) -> "" on success, else the error text.  Returns {-1, ""} or {lowest failing device, its text}.
// An exception out of run() is that slice's error (never out of a thread).
template <class F>
inline std::pair<int, std::string> fan_out_slices(int n_devices, uint64_t n, F run) {
    if (n_devices < 1) n_devices = 1;
    std::vector<std::string> why((size_t)n_devices);
    std::vector<char> failed((size_t)n_devices, 0);
    auto guarded = [&](int d, uint64_t lo, uint64_t hi) {
        try {
            why[d] = run(d, lo, hi);
        } catch (const std::exception& e) {
            why[d] = std::string("exception: ") + e.what();
        } catch (...) {
            why[d] = "unknown exception";
        }
        failed[d] = !why[d].empty();
    };
    std::vector<std::thread> th;
    int first = -1;
    for (int d = 0; d < n_devices; d++) {
        const uint64_t lo = slice_begin(n, n_devices, d), hi = slice_begin(n, n_devices, d + 1);
        if (lo == hi) continue;
        if (first < 0) { first = d; continue; }  // the caller's own slice: run last, after the helpers have been started
        try {
            th.emplace_back(guarded, d, lo, hi);
        } catch (const std::exception& e) {  // no thread to be had: the slice runs on the caller
            guarded(d, lo, hi);
        }
    }
    if (first >= 0) guarded(first, slice_begin(n, n_devices, first), slice_begin(n, n_devices, first + 1));
    for (auto& t : th) t.join();
    for (int d = 0; d < n_devices; d++)
        if (failed[d]) return {d, why[d]};
    return {-1, std::string()};
}

}  // namespace kzg
