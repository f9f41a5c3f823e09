// The host-side concurrency protocols of a context (csrc/host_sync.hpp: HostPool + parallel_for, Combiner, SlotSet, LanePool,
// Published) driven WITHOUT a GPU: 32 threads of mixed operations on shared fake contexts whose "device pass" sleeps, throws, or
// reports wrong proofs, while a builder thread publishes, fills, abandons and reaps tables and a third party creates and frees
// contexts.  Built with -fsanitize=thread and again with -fsanitize=address,undefined by tests/test_sanitizers.py: a lock-scope
// bug of the kind ADVICE r3 found by reading (a lane used by two callers, a follower never woken, a table read before its groups
// were published, a table destroyed under a reader) is a sanitizer report or a wrong count here.
// The contract being defended: one context shared by many threads (bindings/node/src/lib.rs:35,92-299,
// bindings/java/java_code/src/test/java/ethereum/cryptography/LibEthKZGTest.java:28-37).
#include "host_sync.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
using namespace kzg;

static std::atomic<long> g_bad{0}, g_checks{0};
#define EXPECT(c, what) do { g_checks++; if (!(c)) { g_bad++; fprintf(stderr, "FAILED: %s (line %d)\n", what, __LINE__); } } while (0)

struct FakeTable {  // SharedTable: groups become readable as ready_groups is raised with release
    static constexpr int GROUPS = 64;
    int id;
    long entry[GROUPS];  // plain memory: written by the builder BEFORE the release store that publishes the group
    std::atomic<int> ready_groups{0};
    std::atomic<int> state{0};  // 0 building, 1 complete, 2 abandoned
    static std::atomic<int> alive, destroyed_on_builder, destroyed_elsewhere;
    explicit FakeTable(int i) : id(i) { for (long& e : entry) e = -1; alive++; }
    ~FakeTable();
};
std::atomic<int> FakeTable::alive{0}, FakeTable::destroyed_on_builder{0}, FakeTable::destroyed_elsewhere{0};
static thread_local bool t_is_builder = false;
FakeTable::~FakeTable() {
    alive--;
    (t_is_builder ? destroyed_on_builder : destroyed_elsewhere)++;
}

struct Request {  // VerifyRequest
    long payload = 0;
    int verified = -1, status = -1;
    std::string error;
    bool done = false, taken = false;
};
struct FakeLane {  // an engine lane: lane_busy_ guards `in_use`, a plain int
    std::mutex lane_busy_;
    int in_use = 0;
    long calls = 0;
};
struct FakeContext {
    HostPool pool{4};
    Combiner<Request> combiner{3};
    SlotSet<3> slots;
    int slot_in_use[3] = {0, 0, 0};  // plain: guarded by the slot's lock
    std::mutex work_mu[3];           // the "work sets" a pass prefers to find idle
    LanePool<FakeLane> lanes;
    FakeLane primary;
    Published<FakeTable> view;
    std::atomic<long> passes{0}, failed_passes{0}, lanes_made{0};
};

static void nap(int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }

// one fake many-verification pass on a pass slot: verdict = payload % 7 != 0 ("a wrong proof" every seventh), every 23rd pass throws
static void run_pass(FakeContext& c, std::vector<Request*>& batch) {
    const long n = c.passes.fetch_add(1);
    auto lease = c.slots.acquire([&](int k) { return mutex_is_free(c.work_mu[k]); });
    EXPECT(c.slot_in_use[lease.index] == 0, "a pass slot leased twice");
    c.slot_in_use[lease.index] = 1;
    nap(30 + (int)(n % 5) * 40);
    if (n % 23 == 22) { c.slot_in_use[lease.index] = 0; c.failed_passes++; throw std::runtime_error("device fault"); }
    parallel_for((int)batch.size(), 4, &c.pool, [&](int i) {
        batch[i]->verified = batch[i]->payload % 7 != 0;
        batch[i]->status = 0;
    });
    c.slot_in_use[lease.index] = 0;
}
static void verify(FakeContext& c, long payload) {
    Request me;
    me.payload = payload;
    c.combiner.submit(me, [&](std::vector<Request*>& b) { run_pass(c, b); },
                      [&](std::vector<Request*>& b, const std::string& what) { for (Request* r : b) { r->status = 1; r->verified = 0; r->error = what; } });
    EXPECT(me.done, "a request returned before it was done");
    if (me.status == 0) EXPECT(me.verified == (payload % 7 != 0), "wrong verdict");
    else EXPECT(me.status == 1 && me.error == "device fault", "a failed pass must fail its followers with the reason");
}
static void lane_call(FakeContext& c) {
    auto L = c.lanes.lease(&c.primary, 4, false, [&] { c.lanes_made++; nap(300); if (c.lanes_made.load() % 3 == 2) throw std::runtime_error("no room for a lane"); return std::unique_ptr<FakeLane>(new FakeLane); });
    EXPECT(L.e && L.busy.owns_lock(), "a lane lease without its lock");
    EXPECT(L.e->in_use == 0, "a lane used by two callers");
    L.e->in_use = 1;
    L.e->calls++;
    nap(40);
    L.e->in_use = 0;
}
static void msm_stage(FakeContext& c) {  // Engine::launch_msm: one snapshot, the ready groups of `next`, the rest on `main`
    auto v = c.view.snapshot();
    if (!v.main) return;
    int ready = 0;
    if (v.next) ready = v.next->ready_groups.load(std::memory_order_acquire);
    for (int g = 0; g < FakeTable::GROUPS; g++) {
        const FakeTable& t = g < ready ? *v.next : *v.main;
        EXPECT(t.entry[g] == (long)t.id * 1000 + g, "a table group read before it was published");
    }
}
static void builder(FakeContext& c, std::atomic<bool>& stop) {
    t_is_builder = true;
    int id = 1;
    auto full = [&](int i) {
        auto t = std::make_shared<FakeTable>(i);
        for (int g = 0; g < FakeTable::GROUPS; g++) t->entry[g] = (long)i * 1000 + g;
        t->ready_groups.store(FakeTable::GROUPS, std::memory_order_release);
        t->state.store(1);
        return t;
    };
    c.view.publish(full(id++), nullptr);
    while (!stop.load()) {
        auto main = c.view.snapshot().main;
        auto next = std::make_shared<FakeTable>(id++);
        c.view.publish(main, next);  // published BEFORE it is filled: its groups serve as they become ready
        const bool abandon = id % 4 == 0;
        for (int g = 0; g < FakeTable::GROUPS; g += 8) {
            for (int k = g; k < g + 8; k++) next->entry[k] = (long)next->id * 1000 + k;
            next->ready_groups.store(g + 8, std::memory_order_release);
            nap(20);
            if (abandon && g == 24) break;
        }
        if (abandon) {
            next->state.store(2);
            c.view.publish(main, nullptr);  // unpublish: it moves to `retired`
        } else {
            next->state.store(1);
            c.view.publish(next, nullptr);  // the old main retires
            main->state.store(2);
        }
        main.reset();
        next.reset();
        for (int spin = 0; spin < 2000 && c.view.reap([](const FakeTable& t) { return t.state.load() == 2; }) > 0; spin++) nap(50);
    }
    // the last tables: everything still published or retired dies here, on the builder's thread
    c.view.publish(nullptr, nullptr);
    for (int spin = 0; spin < 4000 && c.view.reap([](const FakeTable&) { return true; }) > 0; spin++) nap(50);
}

// The protocol of a context over a device list (c_api.cpp): single calls take a DevicePicker ticket (least load, returned when the
// call ends), batched calls fan out as contiguous slices with error fan-in -- many threads at once on one picker, fake devices that
// sleep / fail / throw.  Checked: the loads return to zero, no device is starved or flooded, every index of every batch is served
// exactly once by the device whose slice holds it, the reported error is the LOWEST failing device's, exceptions stay inside.
static void device_list_protocol(int threads, int iters) {
    constexpr int D = 4;
    DevicePicker picker(D);
    std::atomic<long> calls[D], in_flight[D], worst[D];
    for (int d = 0; d < D; d++) { calls[d] = 0; in_flight[d] = 0; worst[d] = 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::mt19937 rng(99 + t);
            for (int i = 0; i < iters; i++) {
                if (rng() % 4) {  // a single call
                    auto ticket = picker.pick(1 + rng() % 128);
                    const int d = ticket.device();
                    EXPECT(d >= 0 && d < D, "picker: device out of range");
                    calls[d]++;
                    const long now = ++in_flight[d];
                    long w = worst[d].load();
                    while (now > w && !worst[d].compare_exchange_weak(w, now)) {}
                    nap(20 + (int)(rng() % 60));
                    --in_flight[d];
                } else {  // a batch of n items over the D devices, device `bad` failing, device `boom` throwing
                    const uint64_t n = rng() % 23;
                    const int bad = (int)(rng() % (D + 3)), boom = (int)(rng() % (D + 5));
                    std::vector<std::atomic<int>> served(n ? n : 1);
                    for (auto& a : served) a = 0;
                    auto r = fan_out_slices(D, n, [&](int d, uint64_t lo, uint64_t hi) -> std::string {
                        if (lo != slice_begin(n, D, d) || hi != slice_begin(n, D, d + 1)) return "wrong slice";
                        for (uint64_t k = lo; k < hi; k++) served[k] += 1 + 16 * d;
                        nap(10);
                        if (d == boom) throw std::runtime_error("device lost");
                        return d == bad ? "bad " + std::to_string(d) : std::string();
                    });
                    int expect = -1;
                    for (int d = 0; d < D && expect < 0; d++)
                        if ((d == bad || d == boom) && slice_begin(n, D, d) != slice_begin(n, D, d + 1)) expect = d;
                    EXPECT(r.first == expect, "fan_out: not the lowest failing device");
                    EXPECT((r.first < 0) == r.second.empty(), "fan_out: error text and verdict disagree");
                    if (expect >= 0 && expect == boom) EXPECT(r.second.find("device lost") != std::string::npos, "fan_out: the exception's text is lost");
                    for (uint64_t k = 0; k < n; k++) {
                        int owner = 0;
                        while (slice_begin(n, D, owner + 1) <= k) owner++;
                        EXPECT(served[k].load() == 1 + 16 * owner, "fan_out: an item served twice, never, or by the wrong device");
                    }
                }
            }
        });
    for (auto& t : th) t.join();
    long total = 0, lo = 1L << 60, hi = 0;
    for (int d = 0; d < D; d++) {
        EXPECT(picker.load(d) == 0, "picker: load not returned");
        total += calls[d].load();
        lo = std::min(lo, calls[d].load());
        hi = std::max(hi, calls[d].load());
    }
    EXPECT(lo * 2 >= hi, "picker: one device got more than twice the calls of another");
    printf("device list: %ld single calls over %d devices (%ld .. %ld per device), deepest queue %ld %ld %ld %ld\n", total, D, lo, hi,
           worst[0].load(), worst[1].load(), worst[2].load(), worst[3].load());
    DevicePicker one(1);
    { auto t = one.pick(5); EXPECT(t.device() == 0 && one.load(0) == 5, "picker: one device"); }
    EXPECT(one.load(0) == 0, "picker: one device, load returned");
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 300;
    {
        FakeContext shared[2];
        std::atomic<bool> stop{false};
        std::thread b0([&] { builder(shared[0], stop); }), b1([&] { builder(shared[1], stop); });
        // contexts created and freed meanwhile, each used briefly from two threads
        std::thread churn([&] {
            for (int r = 0; r < 12; r++) {
                auto* c = new FakeContext;
                std::thread t([&] { for (int i = 0; i < 20; i++) { verify(*c, r * 100 + i); lane_call(*c); } });
                for (int i = 0; i < 20; i++) verify(*c, r * 1000 + i);
                t.join();
                c->lanes.clear();
                delete c;
            }
        });
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                std::mt19937 rng(1234 + t);
                for (int i = 0; i < iters; i++) {
                    FakeContext& c = shared[(t + i) & 1];
                    switch (rng() % 6) {
                        case 0: case 1: verify(c, (long)t * 100000 + i); break;
                        case 2: lane_call(c); break;
                        case 3: msm_stage(c); break;
                        case 4: {  // a prover call holding a work set: passes should prefer the other slots, and must stay correct either way
                            std::lock_guard<std::mutex> lk(c.work_mu[i % 3]);
                            nap(60);
                            break;
                        }
                        default: {
                            std::atomic<int> sum{0};
                            bool threw = false;
                            try {
                                parallel_for(40, 5, &c.pool, [&](int k) { if (i % 11 == 0 && k == 17) throw std::runtime_error("one index fails"); sum += k; });
                            } catch (const std::runtime_error&) { threw = true; }
                            EXPECT(threw == (i % 11 == 0) && (threw || sum.load() == 780), "parallel_for");
                        }
                    }
                }
            });
        for (auto& t : th) t.join();
        churn.join();
        stop.store(true);
        b0.join();
        b1.join();
        for (auto& c : shared) {
            EXPECT(c.combiner.running() == 0, "a leader left with its pass counted as running");
            EXPECT(c.lanes.size() <= 3, "more lanes than allowed");
            c.lanes.clear();
        }
        printf("passes %ld + %ld (failed %ld + %ld), lanes built %ld + %ld\n", shared[0].passes.load(), shared[1].passes.load(),
               shared[0].failed_passes.load(), shared[1].failed_passes.load(), shared[0].lanes_made.load(), shared[1].lanes_made.load());
    }
    device_list_protocol(threads, iters);
    EXPECT(FakeTable::alive.load() == 0, "a table outlived its context");
    EXPECT(FakeTable::destroyed_elsewhere.load() == 0, "a table was destroyed on a caller's thread instead of the builder's");
    printf("tables destroyed on the builder thread: %d, elsewhere: %d\n", FakeTable::destroyed_on_builder.load(), FakeTable::destroyed_elsewhere.load());
    printf("host_sync: %ld checks, %ld mismatches\n", g_checks.load(), g_bad.load());
    return g_bad.load() ? 1 : 0;
}
