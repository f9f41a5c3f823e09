// eth_kzg_amd_verify_cell_kzg_proof_batch_many: MANY independent verify_cell_kzg_proof_batch problems in one call.
// Reference semantics per problem: DASContext::verify_cell_kzg_proof_batch (crates/eip7594/src/verifier.rs:72-164) ->
// FK20Verifier::verify_multi_opening (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:129-260); the reference gets
// its throughput by verifying from many threads on one context (bindings/node/src/lib.rs:92-299,
// crates/cryptography/bls12_381/src/lib.rs:45-50).  Here the problems of a call share every GPU launch:
//   host threads : validation + de-duplication per problem, staging into one pinned slab, one SHA-256 transcript per problem
//                  (in parallel, behind the GPU's decoding; a persistent pool), then ONE 2-pairing check per pass: the problems' two
//                  G1 sums are folded on the GPU with 127-bit weights derived from all the challenges; in a pass whose folded check
//                  fails (some proof is wrong) the wrong problems are SEARCHED by folding sub-ranges of the resident weighted sums,
//                  one pairing per probe ($ETH_KZG_AMD_VM_SEARCH=0: re-checked one by one; $ETH_KZG_AMD_VM_FOLD=0: never folded)
//   GPU          : decode + subgroup-check all points, decode all cells, per-cell interpolation (challenge-free: behind the
//                  hashes), then per-problem scalars / weights / interpolation sums, ONE LANE PER SCALAR MULTIPLICATION
//                  (k_verify_many.hip), the 64-term interpolation commitments from the commitment window table, per-problem sums;
//                  small passes (concurrent single calls combined) take a short-chain form: everything after the challenge in
//                  one launch, no fold
// Three pass slots, each with its own lock, device arena and pinned slab, on the streams of the prover's three work sets (HIP maps
// a process's streams onto four hardware queues): passes run side by side, a large call is cut into three concurrent parts, and a
// "pass" of one problem is the single path on the slot.  Nothing here takes the context's big lock.
#include "engine.hpp"
#include "curve29.hpp"
#include "host_pairing.hpp"
#include "launch.hpp"
#include "sha256.hpp"

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>

namespace kzg {

#define HIPCK(x)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + \
                                     std::to_string(__LINE__));                                               \
    } while (0)
#define SYNC_CHECKED(stream)                 \
    do {                                     \
        HIPCK(hipGetLastError());            \
        HIPCK(hipStreamSynchronize(stream)); \
        HIPCK(hipGetLastError());            \
    } while (0)

static constexpr int N_BLOB = 4096, N_CELLS = 128, CELL_LEN = 64, BYTES_PER_CELL = 2048;

namespace {
int host_threads(const Knobs& k) {
    int t = 16;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (int)hw < t) t = (int)hw;
    if (k.host_threads) t = k.host_threads;
    return t;
}
// reduce_bytes_to_scalar_bias (crates/cryptography/bls12_381/src/lib.rs:128-140): 256-bit big-endian integer mod r
Fr reduce_be32(const uint8_t* b) {
    Fr x;
    for (int i = 0; i < 8; i++)
        x.v[7 - i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    while (geq_mod<FrParams>(x.v)) {
        uint32_t t[8];
        sub_limbs<8>(t, x.v, FrParams::MOD);
        memcpy(x.v, t, 32);
    }
    return to_mont(x);
}
struct Problem {  // one verification of the call, after validation
    std::vector<const uint8_t*> uniq;  // de-duplicated commitments, first-occurrence order (verifier.rs:49-65)
    std::vector<int> row;              // per cell: index into uniq
    int n = 0, m = 0;                  // cells, unique commitments (0, 0 when the problem is skipped)
};
}  // namespace

int Engine::verify_cell_kzg_proof_batch_many_host(uint64_t n_batches, const uint64_t* n_commitments,
                                                  const uint8_t* const* const* commitments, const uint64_t* n_indices,
                                                  const uint64_t* const* cell_indices, const uint64_t* n_cells,
                                                  const uint8_t* const* const* cells, const uint64_t* n_proofs,
                                                  const uint8_t* const* const* proofs, int* verified, int* status) {
    const int B = (int)n_batches;
    for (int b = 0; b < B; b++) { verified[b] = 0; status[b] = OK; }
    if (B == 0) return OK;
    // A LARGE call is cut into parts that run as passes of their own on different pass slots at the same time: while one part's
    // products are on the GPU the next one's 100+ MB are still being staged and hashed by the host threads (in one pass the
    // staging of all 275 MB of a 1024-problem call came first, ~10 ms, and the GPU waited for it).  Each part folds its own
    // pairing check (one pass, round 3's form: 28.4-28.9 ms against 24.6-26.2 for 1024 problems).
    static thread_local bool in_part = false;
    constexpr int max_parts = VM_SLOTS < 3 ? (int)VM_SLOTS : 3;
    if (!in_part && max_parts > 1 && B >= 192) {
        uint64_t total_cells = 0;
        for (int b = 0; b < B; b++) total_cells += n_cells[b];
        if (total_cells >= 24576) {
            const int parts = max_parts;
            std::vector<int> cut{0};
            uint64_t acc = 0;
            for (int b = 0; b < B; b++) {  // equal shares of the cells
                acc += n_cells[b];
                if ((int)cut.size() < parts && acc * parts >= total_cells * cut.size() && b + 1 < B) cut.push_back(b + 1);
            }
            cut.push_back(B);
            std::vector<int> rcs(cut.size() - 1, OK);
            std::vector<std::string> errs(cut.size() - 1);
            auto run_part = [&](int p) {
                const int lo = cut[p], n = cut[p + 1] - lo;
                in_part = true;
                rcs[p] = verify_cell_kzg_proof_batch_many_host((uint64_t)n, n_commitments + lo, commitments + lo, n_indices + lo, cell_indices + lo, n_cells + lo,
                                                               cells + lo, n_proofs + lo, proofs + lo, verified + lo, status + lo);
                if (rcs[p] == ERR_DEVICE) errs[p] = last_error();
                in_part = false;
            };
            std::vector<std::thread> th;
            for (int p = 1; p + 1 < (int)cut.size(); p++) th.emplace_back(run_part, p);
            run_part(0);
            for (auto& t : th) t.join();
            for (size_t p = 0; p < rcs.size(); p++)
                if (rcs[p] != OK) { set_error(std::runtime_error(errs[p])); return rcs[p]; }
            return OK;
        }
    }
    const int T = host_threads(knobs_);
    std::call_once(vm_pool_once_, [&] { vm_pool_.reset(new HostPool(T > 1 ? T - 1 : 1, [d = dev_] { (void)hipSetDevice(d); })); });
    HostPool* pool = vm_pool_.get();
    // a free pass slot (lock, stream, arena, pinned slab); when all are taken, queue on one of them in turn (host_sync.hpp: SlotSet).
    // First choice: a free slot whose work set (whose in-order stream the pass runs on, below) is not held by a prover call right
    // now -- otherwise the pass and that call wait on each other's work and a fault of either surfaces in both; a peek,
    // not a lease: a prover call that takes the set a moment later shares the stream with the pass, which is correct, only serial
    auto slot_lease = vm_slots_.acquire([&](int k) { return mutex_is_free(work_[1 + k % (NW - 1)].mu); });
    VmSlot* slot = &vm_slot_[slot_lease.index];
    try {
        HIPCK(hipSetDevice(dev_));
        // A slot runs on the stream of one of the prover's three work sets instead of a stream of its own: HIP maps a process's
        // streams onto FOUR hardware queues per priority, and the context's stream + the three work-set streams are exactly four --
        // a fifth stream would share a queue with one of them and the two would take turns (seen in the rocprofv3 trace of four
        // concurrent callers).  A prover call that happens to hold that work set shares the (in-order) stream with the pass.
        const int slot_index = (int)(slot - vm_slot_);
        hipStream_t st = work_[1 + slot_index % (NW - 1)].stream ? work_[1 + slot_index % (NW - 1)].stream : stream_;
        // ---- per problem: validation (verifier.rs:123-164) and de-duplication
        std::vector<Problem> pr(B);
        parallel_for(B, T, pool, [&](int b) {
            const uint64_t nc = n_commitments[b];
            if (!(nc == n_indices[b] && nc == n_cells[b] && nc == n_proofs[b]) || nc > MAX_CELLS_PER_VERIFICATION) { status[b] = ERR_INPUT; return; }
            for (uint64_t i = 0; i < nc; i++)
                if (cell_indices[b][i] >= (uint64_t)N_CELLS) { status[b] = ERR_INPUT; return; }
            if (nc == 0) { verified[b] = 1; return; }  // verifier.rs:90-93
            Problem& p = pr[b];
            p.row.resize(nc);
            std::map<std::string, int> seen;
            for (uint64_t i = 0; i < nc; i++) {
                std::string key((const char*)commitments[b][i], 48);
                auto it = seen.find(key);
                if (it == seen.end()) { it = seen.emplace(key, (int)p.uniq.size()).first; p.uniq.push_back(commitments[b][i]); }
                p.row[i] = it->second;
            }
            p.n = (int)nc;
            p.m = (int)p.uniq.size();
        });
        // ---- chunks of problems: at most CHUNK_CELLS cells per pass (the pinned slab and the arena stay bounded)
        constexpr int CHUNK_CELLS = 131072;  // 1024 verifications of 128 cells: 275 MB of pinned staging, ~1 GB of device arena
        bool fold = true;  // one folded pairing check per pass instead of one per problem (falls back to per-problem checks when it fails)
        if (!knobs_.vm_fold) fold = false;  // (test hook: one pairing per problem)
        for (int b0 = 0; b0 < B;) {
            int b1 = b0, nn = 0, mm = 0;
            while (b1 < B && (b1 == b0 || nn + pr[b1].n <= CHUNK_CELLS)) { nn += pr[b1].n; mm += pr[b1].m; b1++; }
            const int Bc = b1 - b0, n = nn, m = mm;
            if (n == 0) { b0 = b1; continue; }
            // A pass of a few problems (concurrent single calls combined) is a chain of latencies, not work: it takes the SHORT-CHAIN
            // form -- subgroup tests and the interpolation commitments' terms inside the ONE launch of all scalar multiplications
            // (k_vm_mul_small), no fold (its 128-bit multiplication per problem is a 1.4 ms chain; <= 2 T pairing checks are one or
            // two rounds of the host threads) -- decode 0.45 + products 1.7 + sums 0.2 ms of GPU instead of ~6
            const bool small = vm_small_max_ != 0 && Bc <= (vm_small_max_ > 0 ? vm_small_max_ : 2 * T);
            std::vector<int> cell_start(Bc + 1, 0), row_start(Bc + 1, 0);
            for (int i = 0; i < Bc; i++) { cell_start[i + 1] = cell_start[i] + pr[b0 + i].n; row_start[i + 1] = row_start[i] + pr[b0 + i].m; }
            // ---- layout: [inputs, uploaded in one copy][device-only]; pinned slab = inputs + read-backs
            auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
            size_t o = 0;
            const size_t off_p = o; o += up((size_t)n * 48);
            const size_t off_c = o; o += up((size_t)m * 48);
            const size_t off_cells = o; o += up((size_t)n * BYTES_PER_CELL);
            const size_t off_idx = o; o += up((size_t)n * 4);
            const size_t off_row = o; o += up((size_t)n * 4);
            const size_t off_bat = o; o += up((size_t)n * 4);
            const size_t off_pos = o; o += up((size_t)n * 4);
            const size_t off_rowb = o; o += up((size_t)m * 4);
            const size_t off_cs = o; o += up((size_t)(Bc + 1) * 4);
            const size_t off_rs = o; o += up((size_t)(Bc + 1) * 4);
            const size_t in_bytes = o;
            const size_t off_pow = o; o += up((size_t)Bc * 24 * sizeof(Fr));  // second upload (after the hashes)
            const size_t off_rho = o; o += up((size_t)Bc * 16);               // folding weights, 128 bits per problem
            const size_t in2_bytes = o;
            // device only
            const size_t off_pts = o; o += up((size_t)(n + m) * sizeof(G1Affine));
            const size_t off_evals = o; o += up((size_t)n * CELL_LEN * sizeof(Fr));
            const size_t off_coef = o; o += up((size_t)n * CELL_LEN * sizeof(Fr));
            const size_t off_stp = o; o += up((size_t)(n + m) * 4);  // [proofs n | commitments m]
            const size_t off_ste = o; o += up((size_t)Bc * 4);
            const size_t off_rp = o; o += up((size_t)n * sizeof(Fr));
            const size_t off_s1 = o; o += up((size_t)n * sizeof(Fr));
            const size_t off_s2 = o; o += up((size_t)n * sizeof(Fr));
            const size_t off_w = o; o += up((size_t)m * sizeof(Fr));
            const size_t off_isc = o; o += up((size_t)Bc * 64 * sizeof(Fr));
            const size_t off_icm = o; o += up((size_t)Bc * launch::SIZEOF_JACQ);
            const size_t off_prod = o; o += up((size_t)(2 * n + m + (small ? 64 * Bc : 0)) * launch::SIZEOF_JACQ);
            const size_t off_out = o; o += up((size_t)2 * Bc * launch::SIZEOF_JACQ);
            const size_t off_fprod = o; o += up((size_t)2 * Bc * launch::SIZEOF_JACQ);
            const size_t off_fold = o; o += up((size_t)2 * launch::SIZEOF_JACQ);
            const int max_ranges = std::min(Bc, 2048);  // probes per level of the search for wrong proofs
            const size_t off_rng = o; o += up((size_t)max_ranges * 8);
            const size_t off_rsum = o; o += up((size_t)2 * max_ranges * launch::SIZEOF_JACQ);
            const size_t dev_bytes = o;
            // pinned read-backs behind the inputs
            size_t po = in2_bytes;
            const size_t poff_st = po; po += up((size_t)(n + m + Bc) * 4);
            const size_t poff_out = po; po += up((size_t)2 * Bc * launch::SIZEOF_JACQ);
            const size_t poff_fold = po; po += up((size_t)2 * launch::SIZEOF_JACQ);
            const size_t poff_rng = po; po += up((size_t)max_ranges * 8);
            const size_t poff_rsum = po; po += up((size_t)2 * max_ranges * launch::SIZEOF_JACQ);
            const size_t pin_bytes = po;
            if (dev_bytes > slot->dev_cap) {
                if (slot->dev) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(slot->dev)); slot->dev = nullptr; slot->dev_cap = 0; }
                HIPCK(hipMalloc(&slot->dev, dev_bytes + (dev_bytes >> 2)));
                slot->dev_cap = dev_bytes + (dev_bytes >> 2);
            }
            if (pin_bytes > slot->pin_cap) {
                if (slot->pin) { HIPCK(hipHostFree(slot->pin)); slot->pin = nullptr; slot->pin_cap = 0; }
                HIPCK(hipHostMalloc((void**)&slot->pin, pin_bytes + (pin_bytes >> 2), hipHostMallocDefault));
                slot->pin_cap = pin_bytes + (pin_bytes >> 2);
            }
            uint8_t* hb = slot->pin;
            uint8_t* db = (uint8_t*)slot->dev;
            int* h_idx = (int*)(hb + off_idx);
            int* h_row = (int*)(hb + off_row);
            int* h_bat = (int*)(hb + off_bat);
            int* h_pos = (int*)(hb + off_pos);
            int* h_rowb = (int*)(hb + off_rowb);
            memcpy(hb + off_cs, cell_start.data(), (size_t)(Bc + 1) * 4);
            memcpy(hb + off_rs, row_start.data(), (size_t)(Bc + 1) * 4);
            // ---- staging (parallel over problems)
            parallel_for(Bc, T, pool, [&](int i) {
                const int b = b0 + i;
                const Problem& p = pr[b];
                const int c0 = cell_start[i], r0 = row_start[i];
                for (int j = 0; j < p.m; j++) { memcpy(hb + off_c + (size_t)(r0 + j) * 48, p.uniq[j], 48); h_rowb[r0 + j] = i; }
                for (int k = 0; k < p.n; k++) {
                    memcpy(hb + off_p + (size_t)(c0 + k) * 48, proofs[b][k], 48);
                    memcpy(hb + off_cells + (size_t)(c0 + k) * BYTES_PER_CELL, cells[b][k], BYTES_PER_CELL);
                    h_idx[c0 + k] = (int)cell_indices[b][k];
                    h_row[c0 + k] = r0 + p.row[k];
                    h_bat[c0 + k] = i;
                    h_pos[c0 + k] = k;
                }
            });
            // stale contents of the persistent arena must fail closed: poison the status words and the result slots
            int* h_st = (int*)(hb + poff_st);
            memset(h_st, 0xff, (size_t)(n + m + Bc) * 4);
            memset(hb + poff_out, 0xff, (size_t)2 * Bc * launch::SIZEOF_JACQ);
            memset(hb + poff_fold, 0xff, (size_t)2 * launch::SIZEOF_JACQ);
            HIPCK(hipMemsetAsync(db + off_stp, 0xff, (size_t)(n + m) * 4, st));
            HIPCK(hipMemsetAsync(db + off_out, 0xff, (size_t)2 * Bc * launch::SIZEOF_JACQ, st));
            HIPCK(hipMemsetAsync(db + off_fold, 0xff, (size_t)2 * launch::SIZEOF_JACQ, st));
            HIPCK(hipMemsetAsync(db + off_ste, 0, (size_t)Bc * 4, st));
            HIPCK(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, st));
            // ---- challenge-free GPU work: decode (on curve) + subgroup tests of [proofs | commitments], cells, interpolation
            G1Affine* d_pts = (G1Affine*)(db + off_pts);
            int* d_stp = (int*)(db + off_stp);
            launch::g1_decode2(db + off_p, d_pts, d_stp, n, db + off_c, d_pts + n, d_stp + n, m, beta_, st);
            if (!small) launch::g1_subgroup2(d_pts, d_stp, n, d_pts + n, d_stp + n, m, beta_, st);  // (small: inside k_vm_mul_small)
            launch::cells_to_fr(db + off_cells, db + off_evals, nullptr, (int*)(db + off_ste), (const int*)(db + off_bat), nullptr, n, st);
            launch::interp_cells(db + off_evals, (const int*)(db + off_idx), d_w8192_, inv64_, db + off_coef, n, st);
            HIPCK(hipMemcpyAsync(h_st, d_stp, (size_t)(n + m) * 4, hipMemcpyDeviceToHost, st));
            HIPCK(hipMemcpyAsync(h_st + n + m, db + off_ste, (size_t)Bc * 4, hipMemcpyDeviceToHost, st));
            HIPCK(hipGetLastError());
            // ---- Fiat-Shamir challenges on the host threads meanwhile (verifier.rs:269-328): valid inputs are canonical
            // encodings, so the transcript is the input bytes themselves; then the table r^(2^i) per problem
            Fr* h_pow = (Fr*)(hb + off_pow);
            std::vector<uint8_t> digests((size_t)Bc * 32, 0);
            parallel_for(Bc, T, pool, [&](int i) {
                const int b = b0 + i;
                const Problem& p = pr[b];
                Fr* tab = h_pow + (size_t)i * 24;
                if (p.n == 0) { for (int j = 0; j < 24; j++) tab[j] = one<FrParams>(); return; }
                Sha256 sh;
                auto be64 = [](uint64_t v, uint8_t* out) { for (int q = 0; q < 8; q++) out[q] = (uint8_t)(v >> (56 - 8 * q)); };
                uint8_t hdr[16 + 32];
                memcpy(hdr, "RCKZGCBATCH__V1_", 16);
                be64(N_BLOB, hdr + 16); be64(CELL_LEN, hdr + 24); be64((uint64_t)p.m, hdr + 32); be64((uint64_t)p.n, hdr + 40);
                sh.update(hdr, sizeof hdr);
                for (int j = 0; j < p.m; j++) sh.update(p.uniq[j], 48);
                for (int k = 0; k < p.n; k++) {
                    uint8_t ix[16];
                    be64((uint64_t)p.row[k], ix); be64(cell_indices[b][k], ix + 8);
                    sh.update(ix, 16);
                    sh.update(cells[b][k], BYTES_PER_CELL);
                    sh.update(proofs[b][k], 48);
                }
                uint8_t* dig = &digests[(size_t)i * 32];
                sh.finish(dig);
                Fr cur = reduce_be32(dig);
                for (int j = 0; j < 24; j++) { tab[j] = cur; cur = sqr(cur); }
            });
            // ---- decoding verdicts per problem (order of the reference: commitments, proofs, cells): the challenge-free GPU work
            // has long finished under the hashes, so this wait is short; a problem with an error takes no part in what follows
            SYNC_CHECKED(st);
            // (order of the reference: commitments, proofs, cells)
            std::vector<char> live(Bc, 0);
            auto judge = [&]() {
                for (int i = 0; i < Bc; i++) {
                    const int b = b0 + i;
                    const Problem& p = pr[b];
                    live[i] = 0;
                    if (p.n == 0) continue;
                    const int c0 = cell_start[i], r0 = row_start[i];
                    int bad = OK;
                    for (int j = 0; j < p.m && !bad; j++) if (h_st[n + r0 + j]) bad = ERR_G1;
                    for (int k = 0; k < p.n && !bad; k++) if (h_st[c0 + k]) bad = ERR_G1;
                    if (!bad && h_st[n + m + i]) bad = ERR_SCALAR;
                    status[b] = bad;
                    if (!bad) live[i] = 1;
                }
            };
            judge();  // (a small pass judges again once its subgroup tests are in)
            // ---- folding weights: rho_b = SHA-256(seed || b) truncated to 127 bits, seed = SHA-256 over ALL challenges' digests of the
            // pass (so no weight can be predicted before every input byte is fixed); 0 for problems that are out
            uint32_t* h_rho = (uint32_t*)(hb + off_rho);
            int n_live = 0;
            for (int i = 0; i < Bc; i++) n_live += live[i];
            const bool folded = fold && n_live >= 2 && !small;
            if (folded) {
                uint8_t seed[32];
                {
                    Sha256 sh;
                    sh.update((const uint8_t*)"RCKZGCBATCHFOLD1", 16);
                    sh.update(digests.data(), digests.size());
                    sh.finish(seed);
                }
                parallel_for(Bc, T, pool, [&](int i) {
                    uint32_t* r4 = h_rho + 4 * (size_t)i;
                    if (!live[i]) { r4[0] = r4[1] = r4[2] = r4[3] = 0; return; }
                    Sha256 sh;
                    uint8_t ix[8], dg[32];
                    for (int q = 0; q < 8; q++) ix[q] = (uint8_t)((uint64_t)i >> (56 - 8 * q));
                    sh.update(seed, 32);
                    sh.update(ix, 8);
                    sh.finish(dg);
                    memcpy(r4, dg, 16);
                    r4[3] &= 0x7fffffffu;
                    if ((r4[0] | r4[1] | r4[2] | r4[3]) == 0) r4[0] = 1;
                });
            }
            HIPCK(hipMemcpyAsync(db + off_pow, hb + off_pow, in2_bytes - off_pow, hipMemcpyHostToDevice, st));  // power tables + weights
            // ---- per-problem scalars, weights, interpolation sums; every scalar multiplication; the two sums per problem
            const int* d_cs = (const int*)(db + off_cs);
            const int* d_rs = (const int*)(db + off_rs);
            launch::vm_scalars(db + off_pow, (const int*)(db + off_bat), (const int*)(db + off_pos), (const int*)(db + off_idx), d_w8192_,
                               db + off_rp, db + off_s1, db + off_s2, n, st);
            launch::vm_weights(db + off_rp, (const int*)(db + off_row), (const int*)(db + off_rowb), d_cs, db + off_w, m, st);
            launch::vm_interp_sum(db + off_coef, db + off_rp, d_cs, db + off_isc, Bc, st);
            if (small) {
                launch::vm_mul_small(d_pts, db + off_s1, db + off_s2, db + off_w, db + off_isc, d_srs_, db + off_prod, n, m, Bc, d_stp, beta_, st);
                launch::vm_reduce_small(db + off_prod, d_cs, d_rs, db + off_out, n, m, Bc, st);
                HIPCK(hipMemcpyAsync(h_st, d_stp, (size_t)(n + m) * 4, hipMemcpyDeviceToHost, st));  // with the subgroup verdicts now
            } else {
                launch::vm_mul(d_pts, db + off_s1, db + off_s2, db + off_w, db + off_prod, n, m, beta_, st);
                // - commit(interpolation polynomial): 64 fixed bases = group 0 of the commitment window table (verification_key.rs:66-70)
                launch_msm(db + off_isc, TAB_SRS, db + off_icm, 1, Bc, Bc, 0, st);
                launch::vm_reduce(db + off_prod, db + off_icm, d_cs, d_rs, db + off_out, n, Bc, st);
            }
            if (folded) {
                launch::vm_fold(db + off_out, (const uint32_t*)(db + off_rho), db + off_fprod, db + off_fold, Bc, beta_, st);
                HIPCK(hipMemcpyAsync(hb + poff_fold, db + off_fold, (size_t)2 * launch::SIZEOF_JACQ, hipMemcpyDeviceToHost, st));
            }
            HIPCK(hipMemcpyAsync(hb + poff_out, db + off_out, (size_t)2 * Bc * launch::SIZEOF_JACQ, hipMemcpyDeviceToHost, st));
            SYNC_CHECKED(st);
            if (small) judge();
            // ---- verdicts: ONE pairing check of the folded sums; only if that fails (some problem's proof is wrong) one per problem
            const JacQ* sums = (const JacQ*)(hb + poff_out);
            std::atomic<int> device_fault{0};
            auto poisoned = [](const JacQ& s) { return s.x.v[0] == 0xffffffffu && s.z.v[0] == 0xffffffffu; };
            bool all_true = false;
            if (folded) {
                const JacQ* f2 = (const JacQ*)(hb + poff_fold);
                if (poisoned(f2[0]) || poisoned(f2[1])) throw std::runtime_error("many-verification pass left no folded result");
                G1Affine pts[2] = {to_affine(jac_from_jacq(f2[0])), to_affine(jac_from_jacq(f2[1]))};
                all_true = verify_cells_pairing(pts);
            }
            auto check_single = [&](int i) {  // one problem's own pairing check on the sums read back
                G1Affine pts[2];
                for (int j = 0; j < 2; j++) {
                    const JacQ& s = sums[2 * (size_t)i + j];
                    if (poisoned(s)) { device_fault.store(1); return; }
                    pts[j] = to_affine(jac_from_jacq(s));
                }
                verified[b0 + i] = verify_cells_pairing(pts) ? 1 : 0;
            };
            if (all_true) {
                for (int i = 0; i < Bc; i++)
                    if (live[i]) {
                        if (poisoned(sums[2 * (size_t)i]) || poisoned(sums[2 * (size_t)i + 1])) { device_fault.store(1); break; }
                        verified[b0 + i] = 1;
                    }
            } else if (folded && vm_search_) {
                // Some proof of the pass is wrong.  The weighted pairs rho_b * sum_b are still in HBM: SEARCH for the wrong ones by
                // folding sub-ranges (k_vm_fold_ranges: one small launch per level) and checking each with one pairing on a host
                // thread -- a range whose fold passes holds only valid proofs (the same 2^-127 argument, the weights were fixed
                // after every input byte), a single problem whose weighted pair fails is wrong EXACTLY (rho_b != 0 and the
                // pairing group has prime order).  Ranges are split K ways with K chosen so that a level's probes fill the
                // host threads once: one wrong proof among 1024 costs ~36 pairings in 3 rounds instead of 1024 in 64 (round 3:
                // 85 ms per call against 29.6 ms for an all-valid one).  When wrong proofs are many the search stops paying:
                // from a quarter of the live problems suspect, the rest is checked one by one as before.
                std::vector<std::pair<int, int>> suspects{{0, Bc}};
                int* h_rng = (int*)(hb + poff_rng);
                const JacQ* h_rsum = (const JacQ*)(hb + poff_rsum);
                while (!suspects.empty()) {
                    long suspect_problems = 0;
                    for (auto& sr : suspects) suspect_problems += sr.second - sr.first;
                    if ((long)suspects.size() * 4 >= n_live || suspect_problems <= 2 * T) {  // few problems left, or wrong proofs everywhere
                        std::vector<int> todo;
                        for (auto& sr : suspects)
                            for (int i = sr.first; i < sr.second; i++)
                                if (live[i]) todo.push_back(i);
                        parallel_for((int)todo.size(), T, pool, [&](int q) { check_single(todo[q]); });
                        break;
                    }
                    int K = T / (int)suspects.size();
                    K = K < 2 ? 2 : K > 16 ? 16 : K;
                    std::vector<std::pair<int, int>> probes;
                    for (auto& sr : suspects) {
                        const int len = sr.second - sr.first, parts = std::min(K, len);
                        for (int q = 0; q < parts; q++) probes.emplace_back(sr.first + (int)((long)len * q / parts), sr.first + (int)((long)len * (q + 1) / parts));
                    }
                    // a level may need more probes than the range buffers hold (max_ranges = min(Bc, 2048): e.g. > 4096 small problems
                    // with > 1024 scattered wrong proofs): it then runs in batches over the same buffers -- never an error, the
                    // valid problems of the pass keep their verdicts
                    std::vector<char> fails(probes.size(), 0);
                    for (size_t q0 = 0; q0 < probes.size() && !device_fault.load(); q0 += (size_t)max_ranges) {
                        const size_t cnt = std::min(probes.size() - q0, (size_t)max_ranges);
                        for (size_t q = 0; q < cnt; q++) { h_rng[2 * q] = probes[q0 + q].first; h_rng[2 * q + 1] = probes[q0 + q].second; }
                        memset(hb + poff_rsum, 0xff, cnt * 2 * launch::SIZEOF_JACQ);
                        HIPCK(hipMemcpyAsync(db + off_rng, h_rng, cnt * 8, hipMemcpyHostToDevice, st));
                        launch::vm_fold_ranges(db + off_fprod, (const int*)(db + off_rng), db + off_rsum, (int)cnt, st);
                        HIPCK(hipMemcpyAsync(hb + poff_rsum, db + off_rsum, cnt * 2 * launch::SIZEOF_JACQ, hipMemcpyDeviceToHost, st));
                        SYNC_CHECKED(st);
                        parallel_for((int)cnt, T, pool, [&](int q) {
                            G1Affine pts[2];
                            for (int j = 0; j < 2; j++) {
                                const JacQ& sj = h_rsum[2 * (size_t)q + j];
                                if (poisoned(sj)) { device_fault.store(1); return; }
                                pts[j] = to_affine(jac_from_jacq(sj));
                            }
                            fails[q0 + q] = verify_cells_pairing(pts) ? 0 : 1;
                        });
                    }
                    if (device_fault.load()) break;
                    suspects.clear();
                    for (size_t q = 0; q < probes.size(); q++) {
                        const int lo = probes[q].first, hi = probes[q].second;
                        if (!fails[q]) {
                            for (int i = lo; i < hi; i++) if (live[i]) verified[b0 + i] = 1;
                        } else if (hi - lo == 1) {
                            verified[b0 + lo] = 0;  // exact: its own weighted pair fails the check
                        } else {
                            suspects.emplace_back(lo, hi);
                        }
                    }
                }
            } else {
                parallel_for(Bc, T, pool, [&](int i) {
                    if (live[i]) check_single(i);
                });
            }
            if (device_fault.load()) throw std::runtime_error("many-verification pass left no result");
            b0 = b1;
        }
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------------------------
struct Engine::VerifyRequest {
    uint64_t n[4];
    const uint8_t* const* commitments;
    const uint64_t* cell_indices;
    const uint8_t* const* cells;
    const uint8_t* const* proofs;
    int verified = 0, status = OK;
    bool done = false;
    bool taken = false;  // a leader has this request in its pass (its owner then only waits)
    std::string error;  // of a device failure of the pass that carried this request
};

int Engine::verify_cell_kzg_proof_batch_combined(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                                 const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                                 uint64_t n_proofs, const uint8_t* const* proofs, int* verified) {
    *verified = 0;
    const bool enabled = knobs_.verify_combine;
    // ONE caller at a time (verify_lanes_), a large batch, or the feature switched off: the latency-optimised single path (3.3 ms);
    // callers that arrive while it is taken are combined into passes on the three pass slots (a pass of 1-8 problems: 3.7-4.1 ms
    // in the short-chain form).  Round 3 sent the first four callers to four engine lanes; their streams shared hardware queues
    // with each other (HIP has four per process), and lanes cannot share launches the way a pass does: measured this round,
    // verifications/s at 2 / 4 / 8 / 16 / 32 threads: four lanes 450 / 650 / 906 / 1997 / 3703, one lane 523 / 681 / 1284 / 2291 / 4323
    if (!enabled || n_cells > (uint64_t)comb_max_cells_ || verify_inflight_.fetch_add(1) < verify_lanes_) {
        struct Leave { std::atomic<int>* c; bool on; ~Leave() { if (on) c->fetch_sub(1); } } leave{&verify_inflight_, enabled && n_cells <= (uint64_t)comb_max_cells_};
        auto lane = lease_serial();
        const int st = lane.e->verify_cell_kzg_proof_batch_host(n_commitments, commitments, n_indices, cell_indices, n_cells, cells, n_proofs,
                                                                proofs, verified);
        if (st == ERR_DEVICE && lane.e != this) set_error(std::runtime_error(lane.e->last_error()));
        return st;
    }
    verify_inflight_.fetch_sub(1);  // this call goes through the combiner instead
    VerifyRequest me;
    me.n[0] = n_commitments; me.n[1] = n_indices; me.n[2] = n_cells; me.n[3] = n_proofs;
    me.commitments = commitments; me.cell_indices = cell_indices; me.cells = cells; me.proofs = proofs;
    // leader election, follower wake-up and the release of the followers whatever happens in a pass: host_sync.hpp (Combiner), which
    // tests/c/test_host_sync.cpp drives under ThreadSanitizer with fake passes
    combiner_.submit(me, [&](std::vector<VerifyRequest*>& batch) {
    const size_t B = batch.size();
    // whatever happens in here (an allocation failure included) the followers must be released with a verdict: a leader that
    // left with comb_running_ set would leave every queued and every later caller waiting for ever
    int rc = ERR_DEVICE;
    std::string why;
    std::vector<int> ver, st;
    try {
      if (B == 1) {
        // a "pass" of one problem is the single path itself -- 3.3 ms against the 3.65 of a one-problem pass (its points are
        // shifted before the challenge; a pass multiplies by per-lane scalars after it) -- on a pass slot's stream and arena:
        // the latency lane and three slots make four single verifications side by side, one hardware queue each
        auto sl = vm_slots_.acquire();
        VmSlot* slot = &vm_slot_[sl.index];
        const int slot_index = (int)(slot - vm_slot_);
        slot->vs.stream = work_[1 + slot_index % (NW - 1)].stream ? work_[1 + slot_index % (NW - 1)].stream : stream_;
        VerifyRequest* r = batch[0];
        G1Affine pts[2];
        bool empty = false;
        ver.assign(1, 0);
        st.assign(1, (int)ERR_DEVICE);
        HIPCK(hipSetDevice(dev_));
        st[0] = verify_cells_partial(r->n[0], r->commitments, r->n[1], r->cell_indices, r->n[2], r->cells, r->n[3], r->proofs, 0, r->n[2], pts, &empty, nullptr, &slot->vs);
        sl.lock.unlock();  // the pairing needs no slot
        rc = st[0] == ERR_DEVICE ? (int)ERR_DEVICE : (int)OK;
        if (rc == ERR_DEVICE) why = last_error();
        if (st[0] == OK) ver[0] = (empty || verify_cells_pairing_split(pts)) ? 1 : 0;
      } else {
        std::vector<uint64_t> l0(B), l1(B), l2(B), l3(B);
        std::vector<const uint8_t* const*> pc(B), pl(B), pp(B);
        std::vector<const uint64_t*> pi(B);
        ver.assign(B, 0);
        st.assign(B, (int)ERR_DEVICE);
        for (size_t i = 0; i < B; i++) {
            l0[i] = batch[i]->n[0]; l1[i] = batch[i]->n[1]; l2[i] = batch[i]->n[2]; l3[i] = batch[i]->n[3];
            pc[i] = batch[i]->commitments; pi[i] = batch[i]->cell_indices; pl[i] = batch[i]->cells; pp[i] = batch[i]->proofs;
        }
        rc = verify_cell_kzg_proof_batch_many_host(B, l0.data(), pc.data(), l1.data(), pi.data(), l2.data(), pl.data(), l3.data(),
                                                   pp.data(), ver.data(), st.data());
        if (rc == ERR_DEVICE) why = last_error();
      }
    } catch (const std::exception& e) {
        rc = ERR_DEVICE;
        why = std::string("combined verification pass: ") + e.what();
    } catch (...) {
        rc = ERR_DEVICE;
        why = "combined verification pass: unknown failure";
    }
        for (size_t i = 0; i < B; i++) {
            batch[i]->status = (rc == ERR_DEVICE || i >= st.size()) ? (int)ERR_DEVICE : st[i];
            batch[i]->verified = i < ver.size() ? ver[i] : 0;
            batch[i]->error = why;
        }
    }, [&](std::vector<VerifyRequest*>& batch, const std::string& what) {
        for (VerifyRequest* r : batch) { r->status = ERR_DEVICE; r->verified = 0; r->error = "combined verification pass: " + what; }
    });
    if (me.status == ERR_DEVICE) set_error(std::runtime_error(me.error));  // the error text belongs to the calling thread
    *verified = me.status == OK ? me.verified : 0;
    return me.status;
}

}  // namespace kzg
