// verify_cell_kzg_proof_batch and recover_cells_and_kzg_proofs: host orchestration.
// Reference: DASContext::verify_cell_kzg_proof_batch (crates/eip7594/src/verifier.rs:49-164),
// FK20Verifier::{new, verify_multi_opening} (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:58-260),
// compute_fiat_shamir_challenge (:269-328), recover_polynomial_coeff (crates/eip7594/src/recovery.rs:22-151),
// ReedSolomon::{construct_vanishing_poly_from_block_erasures, recover_polynomial_coefficient}
// (crates/cryptography/erasure_codes/src/reed_solomon.rs:220-262,332-384).
// Work split: all G1 / Fr batch arithmetic on the GPU; the sequential SHA-256 transcript and the
// constant-size 2-pairing check on the host.
#include "engine.hpp"
#include <condition_variable>
#include <mutex>
#include "curve29.hpp"
#include "host_pairing.hpp"
#include "launch.hpp"
#include "sha256.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <exception>

extern "C" const unsigned char kzg_srs_begin[];

namespace kzg {

#define HIPCK(x)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + \
                                     std::to_string(__LINE__));                                               \
    } while (0)
// Kernel launches report failures only through the thread's last-error slot: look at it before trusting anything that
// is read back after the synchronisation (a stale status word or result point must never pass for a fresh one).
#define SYNC_CHECKED(stream)                 \
    do {                                     \
        HIPCK(hipGetLastError());            \
        HIPCK(hipStreamSynchronize(stream)); \
        HIPCK(hipGetLastError());            \
    } while (0)

static constexpr int N_BLOB = 4096, N_EXT = 8192, N_CELLS = 128, CELL_LEN = 64, BYTES_PER_CELL = 2048;

static Fr fr_u64(uint64_t v) {
    Fr a = zero<FrParams>();
    a.v[0] = (uint32_t)v;
    a.v[1] = (uint32_t)(v >> 32);
    return to_mont(a);
}
static Fr8 to8(const Fr& a) { Fr8 r; memcpy(&r, &a, 32); return r; }
static Fr from8(const Fr8& a) { Fr r; memcpy(&r, &a, 32); return r; }
static int brp7(int v) { int r = 0; for (int i = 0; i < 7; i++) r |= ((v >> i) & 1) << (6 - i); return r; }


void Engine::init_verifier() {
    launch::init_attributes_verify();
    pairing::init();
    // G2 points of the verification key: [1]_2 = g2_monomial[0], [tau^64]_2 = g2_monomial[64]
    const unsigned char* g2 = kzg_srs_begin + 16 + (size_t)N_BLOB * 48;
    pairing::G2Affine gen, tau;
    if (!pairing::g2_decompress(gen, g2) || !pairing::g2_decompress(tau, g2 + 96 * CELL_LEN))
        throw std::runtime_error("embedded SRS: G2 point failed to decompress");
    g2_tau_ = std::make_shared<pairing::G2Prepared>(pairing::prepare(tau));
    g2_neg_gen_ = std::make_shared<pairing::G2Prepared>(pairing::prepare(pairing::g2_neg(gen)));
    pairing::G2Affine tau1;
    if (!pairing::g2_decompress(tau1, g2 + 96)) throw std::runtime_error("embedded SRS: [tau]_2 failed to decompress");
    g2_tau1_ = std::make_shared<pairing::G2Prepared>(pairing::prepare(tau1));
    // coset shift tables 7^i, 7^-i
    std::vector<Fr> c(N_EXT), ci(N_EXT);
    Fr g = fr_u64(7), gi = inv(g);
    c[0] = ci[0] = one<FrParams>();
    for (int i = 1; i < N_EXT; i++) { c[i] = mul(c[i - 1], g); ci[i] = mul(ci[i - 1], gi); }
    HIPCK(hipMalloc(&d_coset_, N_EXT * sizeof(Fr)));
    HIPCK(hipMalloc(&d_coset_inv_, N_EXT * sizeof(Fr)));
    HIPCK(hipMemcpy(d_coset_, c.data(), N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_coset_inv_, ci.data(), N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
    inv64_ = to8(inv(fr_u64(64)));
    n_inv8192_ = to8(inv(fr_u64(N_EXT)));
}

// reduce_bytes_to_scalar_bias (crates/cryptography/bls12_381/src/lib.rs:128-140): 256-bit big-endian integer mod r
static Fr reduce_be32(const uint8_t* b) {
    Fr x;
    for (int i = 0; i < 8; i++)
        x.v[7 - i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    while (geq_mod<FrParams>(x.v)) {  // 2^256 < 3r: at most two subtractions
        uint32_t t[8];
        sub_limbs<8>(t, x.v, FrParams::MOD);
        memcpy(x.v, t, 32);
    }
    return to_mont(x);
}

// The verification equation is linear in the cells: with the challenge r taken over the WHOLE batch, the two G1
// pairing inputs are sums of per-cell terms, so a shard [lo, hi) of the cell list yields two partial points and the
// partials of all shards add up to the pairing inputs of the full batch.
// `out` receives the two partial points; *empty is set when the batch has no cells at all (verifier.rs:90-93).
int Engine::verify_cells_partial(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                 const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                 uint64_t n_proofs, const uint8_t* const* proofs, uint64_t lo, uint64_t hi,
                                 G1Affine* out, bool* empty, const VerifyDeviceSource* dsrc, VerifyScratch* vs) {
    *empty = false;
    out[0] = out[1] = aff_inf();
    // deduplicate_with_indices (verifier.rs:49-65): byte equality, first-occurrence order
    std::vector<const uint8_t*> uniq;
    std::vector<int> row(n_commitments);
    {
        std::map<std::string, int> seen;
        for (uint64_t i = 0; i < n_commitments; i++) {
            std::string key((const char*)commitments[i], 48);
            auto it = seen.find(key);
            if (it == seen.end()) { it = seen.emplace(key, (int)uniq.size()).first; uniq.push_back(commitments[i]); }
            row[i] = it->second;
        }
    }
    // validation (verifier.rs:123-164)
    if (!(n_commitments == n_indices && n_commitments == n_cells && n_commitments == n_proofs)) return ERR_INPUT;
    if (n_cells > MAX_CELLS_PER_VERIFICATION) return ERR_INPUT;  // (engine.hpp: the 24-entry power table, 32-bit positions)
    for (uint64_t i = 0; i < n_indices; i++)
        if (cell_indices[i] >= (uint64_t)N_CELLS) return ERR_INPUT;
    if (lo > hi || hi > n_cells) return ERR_INPUT;
    const int n_all = (int)n_cells, m = (int)uniq.size();
    if (n_all == 0) { *empty = true; return OK; }  // verifier.rs:90-93
    if (lo == hi) return OK;                       // an empty shard contributes the identity twice
    const int n = (int)(hi - lo), k0 = (int)lo;    // this shard: cells k0 .. k0+n, global exponents r^(k0+k)

    // the engine's own scratch under its lock, or a pass slot's (its holder called)
    std::unique_lock<std::recursive_mutex> lk(mu_, std::defer_lock);
    if (!vs) lk.lock();
    void*& a_dev = vs ? vs->dev : v_dev_;
    size_t& a_dev_cap = vs ? vs->dev_cap : v_dev_cap_;
    uint8_t*& a_pin = vs ? vs->pin : v_pin_;
    size_t& a_pin_cap = vs ? vs->pin_cap : v_pin_cap_;
    const bool two_streams = v_two_streams_ && !vs;
    const bool trace = knobs_.trace;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[verify] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    try {
        HIPCK(hipSetDevice(dev_));
        hipStream_t st = vs ? vs->stream : stream_;
        // ---- stage inputs: gather the caller's scattered buffers into ONE pinned host slab, one async copy to a
        // persistent device arena (no per-call hipMalloc)
        const size_t sz_c = (size_t)m * 48, sz_p = (size_t)n * 48, sz_cells = (size_t)n * BYTES_PER_CELL, sz_i = (size_t)n * sizeof(int);
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t off_c = 0, off_p = off_c + up(sz_c), off_cells = off_p + up(sz_p), off_idx = off_cells + up(sz_cells),
                     off_row = off_idx + up(sz_i), in_bytes = off_row + up(sz_i);
        // pinned read-back area behind the inputs: statuses (m + n + 1 ints) and the two result points
        const size_t off_hst = in_bytes, pin_bytes = off_hst + up(((size_t)m + n + 1) * sizeof(int)) + 256;
        const size_t npts = (size_t)n + m + 64;
        const int ib = n < 256 ? n : 256;
        size_t o = up(in_bytes);
        const size_t off_pts = o; o += up(npts * sizeof(G1Affine));
        const size_t off_evals = o; o += up((size_t)n * CELL_LEN * sizeof(Fr));
        const size_t off_stc = o; o += up((size_t)m * sizeof(int));
        const size_t off_stp = o; o += up(sz_i);
        const size_t off_ste = o; o += 256;
        const size_t off_rp = o; o += up((size_t)n * sizeof(Fr));
        const size_t off_s1 = o; o += up((size_t)n * sizeof(Fr));
        const size_t off_sB = o; o += up(npts * sizeof(Fr));
        const size_t off_part = o; o += up((size_t)ib * 64 * sizeof(Fr));
        // large batches take the byte-shifted lincombs (k_verify.hip): point copies and per-cell interpolation polynomials
        // are built behind the hash, so only the cheap half waits for the challenge
        const bool shifted = n >= pip_shift_min_;
        const size_t off_ws = o; o += up(shifted ? launch::pip_shift_workspace_bytes((int)npts) : launch::pip_workspace_bytes((int)npts));
        const size_t off_coef = o; o += shifted ? up((size_t)n * CELL_LEN * sizeof(Fr)) : 0;
        const size_t off_out = o; o += 512;  // two affine points, or two Jacobian sums (shifted form)
        if (o > a_dev_cap) {
            if (a_dev) { HIPCK(hipStreamSynchronize(st)); HIPCK(hipFree(a_dev)); }
            a_dev = nullptr;
            a_dev_cap = 0;
            HIPCK(hipMalloc(&a_dev, o + (o >> 2)));
            a_dev_cap = o + (o >> 2);
        }
        if (pin_bytes > a_pin_cap) {
            if (a_pin) HIPCK(hipHostFree(a_pin));
            a_pin = nullptr;
            a_pin_cap = 0;
            HIPCK(hipHostMalloc((void**)&a_pin, pin_bytes + (pin_bytes >> 2), hipHostMallocDefault));
            a_pin_cap = pin_bytes + (pin_bytes >> 2);
        }
        uint8_t* hb = a_pin;
        uint8_t* db = (uint8_t*)a_dev;
        uint8_t *hc = hb + off_c, *hp = hb + off_p, *hcells = hb + off_cells;
        int* hidx = (int*)(hb + off_idx);
        int* hrow = (int*)(hb + off_row);
        struct View { void* p; };
        View d_cb{db + off_c}, d_pb{db + off_p}, d_cellb{db + off_cells}, d_idx{db + off_idx}, d_row{db + off_row};
        View d_pts{db + off_pts}, d_evals{db + off_evals}, d_stc{db + off_stc}, d_stp{db + off_stp}, d_ste{db + off_ste};
        // one point array [proofs n | commitments m | 64 SRS points] so that the second lincomb is a single MSM
        G1Affine* d_prf_p = (G1Affine*)d_pts.p;
        G1Affine* d_comm_p = d_prf_p + n;
        int* stc = (int*)(hb + off_hst);  // pinned: the read-backs do not block
        int* stp = stc + m;
        int* ste_p = stp + n;
        // ---- staging, upload and deserialisation run on a helper thread while this one hashes the transcript straight from
        // the caller's buffers (the hash is the longest sequential piece of a verification)
        std::exception_ptr stage_error;
        // (a worker of the engine's small persistent pool: starting and joining a fresh thread per call was 50-100 us of a 3.3 ms call)
        std::call_once(stage_pool_once_, [this] { stage_pool_.reset(new HostPool(primary_ ? 1 : 4, [d = dev_] { (void)hipSetDevice(d); })); });
        struct StageDone {
            std::mutex mu;
            std::condition_variable cv;
            bool done = false;
            void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return done; }); }
        } staged;
        stage_pool_->submit([&]() {
            struct Signal { StageDone& s; ~Signal() { std::lock_guard<std::mutex> lk(s.mu); s.done = true; s.cv.notify_all(); } } signal{staged};
            try {
                HIPCK(hipSetDevice(dev_));
                for (int i = 0; i < m; i++) memcpy(hc + (size_t)i * 48, uniq[i], 48);
                for (int k = 0; k < n; k++) {
                    if (!dsrc) {
                        memcpy(hp + (size_t)k * 48, proofs[k0 + k], 48);
                        memcpy(hcells + (size_t)k * BYTES_PER_CELL, cells[k0 + k], BYTES_PER_CELL);
                    }
                    hidx[k] = (int)cell_indices[k0 + k];
                    hrow[k] = row[k0 + k];
                }
                // stale contents of the persistent arena must fail closed: poison every status word and the result slot
                // (device side and pinned read-back side) before anything is launched
                memset(stc, 0xff, ((size_t)m + n + 1) * sizeof(int));
                HIPCK(hipMemsetAsync(d_stc.p, 0xff, (size_t)m * sizeof(int), st));
                HIPCK(hipMemsetAsync(d_stp.p, 0xff, (size_t)n * sizeof(int), st));
                HIPCK(hipMemsetAsync(db + off_out, 0xff, 512, st));
                if (!dsrc) {
                    HIPCK(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, st));
                } else {  // device-resident form: only the small host-made parts go up; cells and proofs move inside HBM
                    HIPCK(hipMemcpyAsync(db + off_c, hb + off_c, sz_c, hipMemcpyHostToDevice, st));
                    HIPCK(hipMemcpyAsync(db + off_idx, hb + off_idx, in_bytes - off_idx, hipMemcpyHostToDevice, st));
                    HIPCK(hipMemcpyAsync(db + off_p, dsrc->d_proofs + (size_t)k0 * 48, sz_p, hipMemcpyDeviceToDevice, st));
                    HIPCK(hipMemcpyAsync(db + off_cells, dsrc->d_cells + (size_t)k0 * BYTES_PER_CELL, sz_cells, hipMemcpyDeviceToDevice, st));
                }
                HIPCK(hipMemsetAsync(d_ste.p, 0, sizeof(int), st));
                // deserialisation with on-curve + subgroup checks (serialization/src/lib.rs:69-99), on the GPU
                if (shifted) {
                    // decode on this stream; the subgroup tests (126 dependent doublings per point) on a second stream, next to
                    // everything that needs only the coordinates and not the challenge: the byte-shifted point copies (120
                    // dependent doublings per point) and the per-cell interpolation polynomials
                    launch::g1_decode2((const uint8_t*)d_pb.p, d_prf_p, (int*)d_stp.p, n, (const uint8_t*)d_cb.p, d_comm_p, (int*)d_stc.p, m, beta_, st);
                    launch::copy_affine(d_srs_, d_comm_p + m, 64, st);  // vk.g1s: the first 64 SRS points (verification_key.rs:66-70)
                    launch::cells_to_fr((const uint8_t*)d_cellb.p, d_evals.p, nullptr, (int*)d_ste.p, nullptr, nullptr, n, st);
                    if (two_streams) {  // round 3's form: the subgroup tests on a second stream (ETH_KZG_AMD_VERIFY_SIDE_STREAM=1)
                        HIPCK(hipEventRecord(v_decoded_, st));
                        HIPCK(hipStreamWaitEvent(v_side_, v_decoded_, 0));
                        launch::g1_subgroup2(d_prf_p, (int*)d_stp.p, n, d_comm_p, (int*)d_stc.p, m, beta_, v_side_);
                        HIPCK(hipEventRecord(v_checked_, v_side_));
                        launch::pip_shift_prepare(d_pts.p, (int)npts, (int)npts, db + off_ws, beta_, st);
                    } else {
                        launch::pip_shift_prepare_and_subgroup(d_pts.p, (int)npts, (int)npts, db + off_ws, d_prf_p, (int*)d_stp.p, n, d_comm_p,
                                                               (int*)d_stc.p, m, beta_, st);
                    }
                    launch::interp_cells(d_evals.p, (const int*)d_idx.p, d_w8192_, inv64_, db + off_coef, n, st);
                    if (two_streams) HIPCK(hipStreamWaitEvent(st, v_checked_, 0));
                } else {
                    launch::g1_decompress2((const uint8_t*)d_pb.p, d_prf_p, (int*)d_stp.p, n, (const uint8_t*)d_cb.p, d_comm_p, (int*)d_stc.p, m, beta_, st);
                    launch::copy_affine(d_srs_, d_comm_p + m, 64, st);  // vk.g1s: the first 64 SRS points (verification_key.rs:66-70)
                    launch::cells_to_fr((const uint8_t*)d_cellb.p, d_evals.p, nullptr, (int*)d_ste.p, nullptr, nullptr, n, st);
                }
                HIPCK(hipMemcpyAsync(stc, d_stc.p, m * sizeof(int), hipMemcpyDeviceToHost, st));
                HIPCK(hipMemcpyAsync(stp, d_stp.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
                HIPCK(hipMemcpyAsync(ste_p, d_ste.p, sizeof(int), hipMemcpyDeviceToHost, st));
                HIPCK(hipGetLastError());  // launch failures are per thread: this thread's would be lost with it
            } catch (...) {
                stage_error = std::current_exception();
            }
        });
        struct Joiner { StageDone& s; ~Joiner() { s.wait(); } } joiner{staged};  // whatever happens below, the task has left this frame first
        // ---- Fiat-Shamir challenge on the host while the GPU decompresses (verifier.rs:269-328).
        // Valid inputs are canonical encodings, so the transcript is the input bytes themselves.
        Sha256 sh;
        auto be64 = [](uint64_t v, uint8_t* o) { for (int b = 0; b < 8; b++) o[b] = (uint8_t)(v >> (56 - 8 * b)); };
        uint8_t hdr[16 + 32];
        memcpy(hdr, "RCKZGCBATCH__V1_", 16);
        be64(N_BLOB, hdr + 16); be64(CELL_LEN, hdr + 24); be64((uint64_t)m, hdr + 32); be64((uint64_t)n_all, hdr + 40);
        sh.update(hdr, sizeof hdr);
        for (int i = 0; i < m; i++) sh.update(uniq[i], 48);
        int chunks_in = 0;  // (device-resident form) chunks of the cells' host mirror that have arrived
        for (int k = 0; k < n_all; k++) {  // the transcript always covers the whole batch, whatever the shard
            while (dsrc && chunks_in < dsrc->n_chunks && k >= chunks_in * dsrc->chunk_cells) HIPCK(hipEventSynchronize(dsrc->chunk_events[chunks_in++]));
            uint8_t ix[16];
            be64((uint64_t)row[k], ix); be64(cell_indices[k], ix + 8);
            sh.update(ix, 16);
            sh.update(cells[k], BYTES_PER_CELL);
            sh.update(proofs[k], 48);
        }
        uint8_t dig[32];
        sh.finish(dig);
        Fr r = reduce_be32(dig);
        lap("sha256 transcript (host)");
        staged.wait();
        if (stage_error) std::rethrow_exception(stage_error);
        SYNC_CHECKED(st);
        lap("wait decompress/deserialise");
        for (int i = 0; i < m; i++) if (stc[i]) return ERR_G1;  // order of the reference: commitments, proofs, cells
        for (int i = 0; i < n; i++) if (stp[i]) return ERR_G1;
        if (*ste_p) return ERR_SCALAR;

        // ---- scalars
        Fr8 tab[24];
        Fr cur = r;
        for (int i = 0; i < 24; i++) { tab[i] = to8(cur); cur = sqr(cur); }
        View d_rp{db + off_rp}, d_s1{db + off_s1}, d_sB{db + off_sB};
        Fr* d_s2 = (Fr*)d_sB.p;
        Fr* d_w = d_s2 + n;
        Fr* d_interp = d_w + m;
        launch::verify_scalars(tab, k0, (const int*)d_idx.p, d_w8192_, d_rp.p, d_s1.p, d_s2, n, st);
        launch::verify_weights(d_rp.p, (const int*)d_row.p, d_w, n, m, st);
        View d_part{db + off_part};
        if (shifted) launch::interp_sum(db + off_coef, d_rp.p, d_part.p, ib, d_interp, n, st);
        else launch::interp(d_evals.p, (const int*)d_idx.p, d_rp.p, d_w8192_, inv64_, d_part.p, ib, d_interp, n, st);
        // ---- the four lincombs (verifier.rs:186,200,224,235) as two bucket MSMs over the shared point array:
        //   out[0] = sum r^k pi_k;   out[1] = sum r^k h^64 pi_k + sum w_row C_row - commit(interpolation poly)
        View d_ws{db + off_ws}, d_out{db + off_out};
        if (shifted) {
            launch::msm_pippenger2_shifted(d_s1.p, n, d_sB.p, n + m + 64, (int)npts, d_ws.p, d_out.p, st);
            JacQ sums[2];
            HIPCK(hipMemcpyAsync(sums, d_out.p, sizeof sums, hipMemcpyDeviceToHost, st));
            SYNC_CHECKED(st);
            for (int i = 0; i < 2; i++) {
                if (sums[i].x.v[0] == 0xffffffffu && sums[i].z.v[0] == 0xffffffffu) throw std::runtime_error("verification MSM left no result");
                out[i] = to_affine(jac_from_jacq(sums[i]));
            }
        } else {
            launch::msm_pippenger2(d_pts.p, d_s1.p, n, d_sB.p, n + m + 64, d_ws.p, d_out.p, beta_, st);
            HIPCK(hipMemcpyAsync(out, d_out.p, 2 * sizeof(G1Affine), hipMemcpyDeviceToHost, st));
            SYNC_CHECKED(st);
        }
        for (int i = 0; i < 2; i++)  // the poison pattern (or anything else that is not a reduced coordinate) is a device failure
            if (out[i].x.v[11] > FpParams::MOD[11] || out[i].y.v[11] > FpParams::MOD[11]) throw std::runtime_error("verification MSM left no result");
        lap("scalars+interp+lincombs (GPU)");
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// pairing check e(sum r^k pi_k, [tau^64]_2) * e(C - I + weighted proofs, -[1]_2) == 1 (verifier.rs:242-259)
bool Engine::verify_cells_pairing(const G1Affine* pts) const {
    const pairing::G2Prepared* q[2] = {g2_tau_.get(), g2_neg_gen_.get()};
    return pairing::product_is_one(pts, q, 2);
}

// A single verification waits for its pairing check on the calling thread (0.85 ms of 2.8).  The two Miller loops are
// independent, so the second one goes to a thread of the staging pool (idle by now: its staging task finished before the
// challenge) and the values meet in front of the one final exponentiation: 63 squarings + 68 line products per thread instead
// of 63 + 136 on one.  Whoever gets to the second loop first computes it -- a busy pool costs nothing.
bool Engine::verify_cells_pairing_split(const G1Affine* pts) {
    if (!stage_pool_) return verify_cells_pairing(pts);
    struct Shared {
        pairing::Fp12 f;
        G1Affine p;
        std::atomic<int> claimed{0}, done{0};
    };
    auto sh = std::make_shared<Shared>();
    sh->p = pts[1];
    const pairing::G2Prepared* q1 = g2_neg_gen_.get();
    stage_pool_->submit([sh, q1] {
        if (sh->claimed.exchange(1, std::memory_order_acq_rel)) return;
        sh->f = pairing::miller_loop(sh->p, *q1);
        sh->done.store(1, std::memory_order_release);
    });
    const pairing::Fp12 f0 = pairing::miller_loop(pts[0], *g2_tau_);
    if (!sh->claimed.exchange(1, std::memory_order_acq_rel)) {
        sh->f = pairing::miller_loop(sh->p, *q1);
    } else {
        while (!sh->done.load(std::memory_order_acquire)) std::this_thread::yield();
    }
    return pairing::final_exponentiation_is_one(pairing::fp12_mul(f0, sh->f));
}

// Device-resident form of the same check: the four flat arrays already sit in this GPU's HBM (cells straight from a prover or
// recovery call, for instance) and STAY there for the GPU's part: decoding, subgroup tests, shifted copies and interpolation
// read them after a device-to-device copy into the arena, at once.  What has to come down is what the Fiat-Shamir transcript
// hashes -- a sequential SHA-256 over every byte, which belongs on a host core (2 cycles per byte with SHA-NI: 7.2 ms for
// config 3's 17.6 MB, the floor of this call): commitments, indices and proofs first (0.9 MB; the host de-duplicates the
// commitments), then the cells in eight chunks that the hash consumes as they land (PCIe delivers 20x faster than it reads).
// Round 3 copied everything down, gathered it on the host and uploaded it again.
int Engine::verify_cell_kzg_proof_batch_device(uint64_t n, const uint8_t* d_commitments, const uint64_t* d_cell_indices,
                                               const uint8_t* d_cells, const uint8_t* d_proofs, int* verified, hipStream_t user_stream) {
    *verified = 0;
    if (n == 0) { *verified = 1; return OK; }  // verifier.rs:90-93
    if (n > MAX_CELLS_PER_VERIFICATION) return ERR_INPUT;
    std::lock_guard<std::recursive_mutex> lk(mu_);  // the pinned mirror is the context's (grow-only, reused by every call)
    G1Affine pts[2];
    bool empty = false;
    int rc = OK;
    try {
        std::vector<const uint8_t*> cp(n), lp(n), pp(n);  // inside the try: a bogus n must not unwind through the C ABI
        HIPCK(hipSetDevice(dev_));
        const size_t sz_c = n * 48, sz_i = n * sizeof(uint64_t), sz_l = n * (size_t)BYTES_PER_CELL, sz_p = n * 48;
        const size_t need = sz_c + sz_i + sz_l + sz_p;
        if (need > vd_pin_cap_) {
            if (vd_pin_) { HIPCK(hipHostFree(vd_pin_)); vd_pin_ = nullptr; vd_pin_cap_ = 0; }
            HIPCK(hipHostMalloc((void**)&vd_pin_, need + (need >> 2), hipHostMallocDefault));
            vd_pin_cap_ = need + (need >> 2);
        }
        for (hipEvent_t& e : vd_events_)
            if (!e) HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        uint8_t* pin = vd_pin_;
        uint8_t *pin_c = pin, *pin_i = pin + sz_c, *pin_p = pin + sz_c + sz_i, *pin_l = pin + sz_c + sz_i + sz_p;
        hipStream_t st = stream_;
        if (user_stream != st) {  // what the caller has queued on its stream (NULL: the default stream) produces the inputs
            HIPCK(hipEventRecord(v_decoded_, user_stream));
            HIPCK(hipStreamWaitEvent(st, v_decoded_, 0));
        }
        HIPCK(hipMemcpyAsync(pin_c, d_commitments, sz_c, hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(pin_i, d_cell_indices, sz_i, hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(pin_p, d_proofs, sz_p, hipMemcpyDeviceToHost, st));
        HIPCK(hipEventRecord(v_checked_, st));
        const int n_chunks = n >= 64 * VD_CHUNKS ? VD_CHUNKS : 1, chunk_cells = (int)((n + n_chunks - 1) / n_chunks);
        for (int j = 0; j < n_chunks; j++) {
            const size_t c0 = (size_t)j * chunk_cells, c1 = std::min<size_t>(n, c0 + chunk_cells);
            if (c1 > c0) HIPCK(hipMemcpyAsync(pin_l + c0 * BYTES_PER_CELL, d_cells + c0 * BYTES_PER_CELL, (c1 - c0) * BYTES_PER_CELL, hipMemcpyDeviceToHost, st));
            HIPCK(hipEventRecord(vd_events_[j], st));
        }
        HIPCK(hipEventSynchronize(v_checked_));  // commitments, indices, proofs are down: validation and de-duplication can start
        for (uint64_t k = 0; k < n; k++) {
            cp[k] = pin_c + k * 48;
            lp[k] = pin_l + k * (size_t)BYTES_PER_CELL;
            pp[k] = pin_p + k * 48;
        }
        const VerifyDeviceSource src{d_cells, d_proofs, vd_events_, chunk_cells, n_chunks};
        rc = verify_cells_partial(n, cp.data(), n, reinterpret_cast<const uint64_t*>(pin_i), n, lp.data(), n, pp.data(), 0, n, pts, &empty, &src);
        if (rc != OK) (void)hipStreamSynchronize(st);  // (an early error return: the copies into the mirror must not outlive the call)
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    if (rc) return rc;
    *verified = (empty || verify_cells_pairing_split(pts)) ? 1 : 0;
    return OK;
}

int Engine::verify_cell_kzg_proof_batch_host(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                             const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                             uint64_t n_proofs, const uint8_t* const* proofs, int* verified) {
    *verified = 0;
    G1Affine pts[2];
    bool empty = false;
    int st = verify_cells_partial(n_commitments, commitments, n_indices, cell_indices, n_cells, cells, n_proofs, proofs, 0,
                                  n_cells, pts, &empty);
    if (st) return st;
    const auto t0 = std::chrono::steady_clock::now();
    *verified = (empty || verify_cells_pairing_split(pts)) ? 1 : 0;
    if (knobs_.trace)
        fprintf(stderr, "[verify] %-28s %8.3f ms\n", "pairing check (host)",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return OK;
}

// Sharded form, step 1: this rank's cells [lo, hi) of the batch -> 96 bytes (two compressed G1 partial sums).
int Engine::verify_cell_kzg_proof_batch_partial_host(uint64_t n_commitments, const uint8_t* const* commitments,
                                                     uint64_t n_indices, const uint64_t* cell_indices, uint64_t n_cells,
                                                     const uint8_t* const* cells, uint64_t n_proofs,
                                                     const uint8_t* const* proofs, uint64_t lo, uint64_t hi, uint8_t* out96) {
    G1Affine pts[2];
    bool empty = false;
    int st = verify_cells_partial(n_commitments, commitments, n_indices, cell_indices, n_cells, cells, n_proofs, proofs, lo, hi,
                                  pts, &empty);
    if (st) return st;
    g1_compress(out96, pts[0]);
    g1_compress(out96 + 48, pts[1]);
    return OK;
}

// Sharded form, step 2: add the partials of all shards (group addition is not an RCCL reduction, so the gathered
// 96-byte records are summed on the host) and run the one pairing check.
int Engine::verify_cell_kzg_proof_batch_combine_host(uint64_t n_partials, const uint8_t* partials96, int* verified) {
    *verified = 0;
    G1Jac acc[2] = {jac_inf(), jac_inf()};
    for (uint64_t i = 0; i < n_partials; i++)
        for (int j = 0; j < 2; j++) {
            G1Affine a;
            if (g1_decompress(a, partials96 + i * 96 + 48 * j) != 0) return ERR_G1;
            acc[j] = add_mixed(acc[j], a);
        }
    G1Affine pts[2] = {to_affine(acc[0]), to_affine(acc[1])};
    // no cells anywhere => both sums are the identity and the product of pairings is 1, as the reference's early return
    *verified = verify_cells_pairing(pts) ? 1 : 0;
    return OK;
}

// ------------------------------------------------------------------------------------------------
// Recovery of R blobs at once.  Per blob: validated cell list -> coefficients in d_coeffs_[r].
// Returns per-blob statuses in `st_out`; blobs that fail validation / decoding are skipped by the caller.
int Engine::recover_batch_to_coeffs(int R, const uint64_t* n_cells, const uint8_t* const* const* cells,
                                    const uint64_t* const* cell_indices, int* st_out) {
    hipStream_t st = stream_;
    ensure_workspace(R);
    // host: presence masks (domain order) and the flattened cell list; the vanishing polynomials are built on the GPU
    std::vector<uint32_t> present((size_t)R * 4, 0xffffffffu);  // a blob that failed validation has nothing missing
    std::vector<int> slot, stof;
    size_t total_cells = 0;
    for (int r = 0; r < R; r++) total_cells += st_out[r] == OK ? n_cells[r] : 0;
    PoolBuf hcells_buf(*this, total_cells * BYTES_PER_CELL, true);  // pinned staging: the H2D copy below runs at link speed
    uint8_t* hcells = (uint8_t*)hcells_buf.p;
    const size_t hcells_bytes = total_cells * BYTES_PER_CELL;
    slot.reserve(total_cells);
    stof.reserve(total_cells);
    size_t pos = 0;
    for (int r = 0; r < R; r++) {
        if (st_out[r] != OK) continue;
        // domain-order index of a cell = bit-reversed cell index (cosets.rs:186-195); missing = complement (recovery.rs:69-75)
        uint32_t* m = &present[(size_t)r * 4];
        m[0] = m[1] = m[2] = m[3] = 0;
        for (uint64_t k = 0; k < n_cells[r]; k++) {
            const int i = brp7((int)cell_indices[r][k]);
            m[i >> 5] |= 1u << (i & 31);
        }
        for (uint64_t k = 0; k < n_cells[r]; k++) {
            memcpy(&hcells[pos * BYTES_PER_CELL], cells[r][k], BYTES_PER_CELL);
            slot.push_back(r * N_CELLS + (int)cell_indices[r][k]);  // scatter into blob r's 128 cell slots (cosets.rs:170-175)
            stof.push_back(r);
            pos++;
        }
    }
    PoolBuf d_cellb(*this, hcells_bytes);
    if (total_cells) HIPCK(hipMemcpyAsync(d_cellb.p, hcells, hcells_bytes, hipMemcpyHostToDevice, st));
    return rs_decode(R, (const uint8_t*)d_cellb.p, /*source index = list position*/ false, slot, stof, present, st_out);
}

// Reed-Solomon decode of R blobs whose present cells are listed in `slot` (blob * 128 + cell index), `stof` (blob of each
// list entry) and `present` (domain-order masks); the cell bytes are read from d_cells at list position k, or at
// slot[k] when the caller's buffer is the flat [R][128][2048] layout.  Leaves the coefficients in d_coeffs_.
int Engine::rs_decode(int R, const uint8_t* d_cells, bool flat_source, const std::vector<int>& slot, const std::vector<int>& stof,
                      const std::vector<uint32_t>& present, int* st_out) {
    hipStream_t st = stream_;
    const int n = (int)slot.size();
    Fr seven64 = fr_u64(7);
    for (int i = 0; i < 6; i++) seven64 = sqr(seven64);
    PoolBuf d_slot(*this, (size_t)n * sizeof(int)), d_stof(*this, (size_t)n * sizeof(int));
    PoolBuf d_E(*this, (size_t)R * N_EXT * sizeof(Fr)), d_T(*this, (size_t)R * N_EXT * sizeof(Fr)), d_U(*this, (size_t)R * N_EXT * sizeof(Fr));
    PoolBuf d_zp(*this, (size_t)R * 65 * sizeof(Fr)), d_deg(*this, R * sizeof(int)), d_present(*this, present.size() * 4);
    PoolBuf d_zeval(*this, (size_t)R * N_CELLS * sizeof(Fr)), d_zcinv(*this, (size_t)R * N_CELLS * sizeof(Fr)), d_st(*this, R * sizeof(int));
    if (n) {
        HIPCK(hipMemcpyAsync(d_slot.p, slot.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d_stof.p, stof.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
    }
    HIPCK(hipMemcpyAsync(d_present.p, present.data(), present.size() * 4, hipMemcpyHostToDevice, st));
    launch::rec_vanishing_poly((const uint32_t*)d_present.p, d_w8192_, d_zp.p, (int*)d_deg.p, R, st);
    HIPCK(hipMemsetAsync(d_E.p, 0, (size_t)R * N_EXT * sizeof(Fr), st));
    HIPCK(hipMemsetAsync(d_st.p, 0, R * sizeof(int), st));
    if (n) launch::cells_to_fr(d_cells, d_E.p, (const int*)d_slot.p, (int*)d_st.p, (const int*)d_stof.p,
                               flat_source ? (const int*)d_slot.p : nullptr, n, st);  // E in cell order
    launch::rec_vanishing(d_zp.p, (const int*)d_deg.p, d_w8192_, to8(seven64), d_zeval.p, d_zcinv.p, R, st);
    launch::rec_dit_half(R, d_E.p, d_zeval.p, d_T.p, d_w8192_, st);                                      // (E*Z) -> IFFT ...
    launch::rec_dit_last(R, d_T.p, d_coset_, n_inv8192_, d_U.p, nullptr, nullptr, d_w8192_, 0, st);      // ... * 7^i
    launch::rec_dif_half(R, d_U.p, d_zcinv.p, d_E.p, d_w8192_, st);                                      // coset FFT, / Z
    launch::rec_dit_half(R, d_E.p, nullptr, d_T.p, d_w8192_, st);                                        // coset IFFT ...
    launch::rec_dit_last(R, d_T.p, d_coset_inv_, n_inv8192_, nullptr, d_coeffs_, (int*)d_st.p, d_w8192_, 1, st);  // ... * 7^-i
    std::vector<int> hst(R);
    HIPCK(hipMemcpyAsync(hst.data(), d_st.p, R * sizeof(int), hipMemcpyDeviceToHost, st));
    SYNC_CHECKED(st);
    for (int r = 0; r < R; r++) {
        if (st_out[r] != OK) continue;
        if (hst[r] & 1) st_out[r] = ERR_SCALAR;
        else if (hst[r] & 4) st_out[r] = ERR_RECOVERY;
    }
    return OK;
}

// Device-resident form: d_cells is the flat [R][128][2048] extended-blob layout in HBM, present_masks[2 r .. 2 r + 1] the
// 128-bit set of cells that hold data (bit c of word c / 64); missing cells are never read.  Outputs as in the
// device-resident prover call; status[r] per blob, outputs of a failed blob are unspecified.
int Engine::recover_cells_and_kzg_proofs_device(int R, const uint8_t* d_cells, const uint64_t* present_masks, uint8_t* d_out_cells,
                                                uint8_t* d_out_proofs, int* status, hipStream_t user_stream) {
    if (R <= 0) return OK;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    if (R > device_batch_max_) {  // sub-batches on the same streams (the scratch of one pass is 0.9 MB per blob): see compute_cells_and_kzg_proofs_device
        for (int r0 = 0; r0 < R; r0 += device_batch_max_) {
            const int nr = std::min(device_batch_max_, R - r0);
            const int rc = recover_cells_and_kzg_proofs_device(nr, d_cells + (size_t)r0 * N_CELLS * BYTES_PER_CELL, present_masks + 2 * (size_t)r0,
                                                               d_out_cells ? d_out_cells + (size_t)r0 * N_CELLS * BYTES_PER_CELL : nullptr,
                                                               d_out_proofs ? d_out_proofs + (size_t)r0 * N_CELLS * 48 : nullptr, status + r0, user_stream);
            if (rc) return rc;
        }
        return OK;
    }
    try {
        HIPCK(hipSetDevice(dev_));
        ensure_workspace(R);  // also orders stream_ behind the previous asynchronous call that used the workspace
        {   // the decode runs on the library's stream: it must see what the caller's stream (NULL: the default stream) wrote into d_cells
            HIPCK(hipEventRecord(work_[0].ev_in, user_stream));
            HIPCK(hipStreamWaitEvent(stream_, work_[0].ev_in, 0));
        }
        std::vector<uint32_t> present((size_t)R * 4, 0xffffffffu);
        std::vector<int> slot, stof;
        for (int r = 0; r < R; r++) {
            const uint64_t m0 = present_masks[2 * r], m1 = present_masks[2 * r + 1];
            const int cnt = __builtin_popcountll(m0) + __builtin_popcountll(m1);
            status[r] = cnt < N_CELLS / 2 ? ERR_INPUT : OK;  // recovery.rs:90-146: at least half of the cells
            if (status[r] != OK) continue;
            uint32_t* m = &present[(size_t)r * 4];
            m[0] = m[1] = m[2] = m[3] = 0;
            for (int c = 0; c < N_CELLS; c++) {
                if (!(((c < 64 ? m0 : m1) >> (c & 63)) & 1)) continue;
                const int i = brp7(c);
                m[i >> 5] |= 1u << (i & 31);
                slot.push_back(r * N_CELLS + c);
                stof.push_back(r);
            }
        }
        int rc = rs_decode(R, d_cells, /*flat_source=*/true, slot, stof, present, status);
        if (rc) return rc;
        hipStream_t st = user_stream ? user_stream : stream_;
        if (d_out_cells) launch::coeffs_to_cells(R, d_coeffs_, d_out_cells, d_w29_, st);
        if (d_out_proofs) run_proofs_from_coeffs(R, d_out_proofs, st);
        HIPCK(hipEventRecord(work_[0].done, st));  // the next user of the workspace waits for these kernels (ensure_workspace)
        HIPCK(hipGetLastError());
        if (!user_stream) SYNC_CHECKED(st);
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// validate_recovery_inputs (recovery.rs:90-146)
static int validate_recovery(uint64_t n_cells, uint64_t n_indices, const uint64_t* cell_indices) {
    if (n_indices != n_cells) return ERR_INPUT;
    for (uint64_t i = 0; i < n_indices; i++)
        if (cell_indices[i] >= (uint64_t)N_CELLS) return ERR_INPUT;
    for (uint64_t i = 1; i < n_indices; i++)
        if (!(cell_indices[i - 1] < cell_indices[i])) return ERR_INPUT;
    if (n_indices < (uint64_t)N_CELLS / 2 || n_indices > (uint64_t)N_CELLS) return ERR_INPUT;
    return OK;
}

int Engine::recover_cells_and_kzg_proofs_batch_host(int R, const uint64_t* n_cells, const uint8_t* const* const* cells,
                                                    const uint64_t* n_indices, const uint64_t* const* cell_indices,
                                                    uint8_t* const* const* out_cells, uint8_t* const* const* out_proofs,
                                                    int* status) {
    if (R <= 0) return OK;
    for (int r = 0; r < R; r++) status[r] = validate_recovery(n_cells[r], n_indices[r], cell_indices[r]);
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        const bool trace = knobs_.trace;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!trace) return;
            auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[recover] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        int rc = recover_batch_to_coeffs(R, n_cells, cells, cell_indices, status);
        if (rc) return rc;
        lap("stage in + RS decode");
        // compute_multi_opening_proofs(Input::PolyCoeff) = stages C..I (prover.rs:164-170) for the whole batch
        PoolBuf d_c(*this, (size_t)R * N_CELLS * BYTES_PER_CELL), d_p(*this, (size_t)R * N_CELLS * 48);
        launch::coeffs_to_cells(R, d_coeffs_, (uint8_t*)d_c.p, d_w29_, stream_);
        run_proofs_from_coeffs(R, (uint8_t*)d_p.p, stream_);
        PoolBuf hc_buf(*this, (size_t)R * N_CELLS * BYTES_PER_CELL, true), hp_buf(*this, (size_t)R * N_CELLS * 48, true);
        const uint8_t* hc = (const uint8_t*)hc_buf.p;
        const uint8_t* hp = (const uint8_t*)hp_buf.p;
        HIPCK(hipMemcpyAsync(hc_buf.p, d_c.p, (size_t)R * N_CELLS * BYTES_PER_CELL, hipMemcpyDeviceToHost, stream_));
        HIPCK(hipMemcpyAsync(hp_buf.p, d_p.p, (size_t)R * N_CELLS * 48, hipMemcpyDeviceToHost, stream_));
        SYNC_CHECKED(stream_);
        lap("cells + proofs + D2H");
        for (int r = 0; r < R; r++) {
            if (status[r] != OK) continue;
            for (int k = 0; k < N_CELLS; k++) {
                memcpy(out_cells[r][k], &hc[((size_t)r * N_CELLS + k) * BYTES_PER_CELL], BYTES_PER_CELL);
                memcpy(out_proofs[r][k], &hp[((size_t)r * N_CELLS + k) * 48], 48);
            }
        }
        lap("scatter to caller buffers");
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::recover_cells_and_kzg_proofs_host(uint64_t n_cells, const uint8_t* const* cells, uint64_t n_indices,
                                              const uint64_t* cell_indices, uint8_t* const* out_cells,
                                              uint8_t* const* out_proofs) {
    int st = OK;
    const uint8_t* const* cl[1] = {cells};
    const uint64_t* ix[1] = {cell_indices};
    uint8_t* const* oc[1] = {out_cells};
    uint8_t* const* op[1] = {out_proofs};
    int rc = recover_cells_and_kzg_proofs_batch_host(1, &n_cells, cl, &n_indices, ix, oc, op, &st);
    return rc ? rc : st;
}

}  // namespace kzg
