// verify_cell_kzg_proof_batch and recover_cells_and_kzg_proofs: host orchestration.
// Reference: DASContext::verify_cell_kzg_proof_batch (crates/eip7594/src/verifier.rs:49-164),
// FK20Verifier::{new, verify_multi_opening} (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:58-260),
// compute_fiat_shamir_challenge (:269-328), recover_polynomial_coeff (crates/eip7594/src/recovery.rs:22-151),
// ReedSolomon::{construct_vanishing_poly_from_block_erasures, recover_polynomial_coefficient}
// (crates/cryptography/erasure_codes/src/reed_solomon.rs:220-262,332-384).
// Work split: all G1 / Fr batch arithmetic on the GPU; the sequential SHA-256 transcript and the
// constant-size 2-pairing check on the host.
#include "engine.hpp"
#include "curve.hpp"
#include "host_pairing.hpp"
#include "launch.hpp"
#include "sha256.hpp"

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

extern "C" const unsigned char kzg_srs_begin[];

namespace kzg {

#define HIPCK(x)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (x);                                                                                  \
        if (e_ != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + ":" + \
                                     std::to_string(__LINE__));                                               \
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

struct DevBuf {  // scoped device allocation
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) throw std::runtime_error("hipMalloc failed"); }
    ~DevBuf() { if (p) hipFree(p); }
    DevBuf(const DevBuf&) = delete;
};

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

int Engine::verify_cell_kzg_proof_batch_host(uint64_t n_commitments, const uint8_t* const* commitments, uint64_t n_indices,
                                             const uint64_t* cell_indices, uint64_t n_cells, const uint8_t* const* cells,
                                             uint64_t n_proofs, const uint8_t* const* proofs, int* verified) {
    *verified = 0;
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
    for (uint64_t i = 0; i < n_indices; i++)
        if (cell_indices[i] >= (uint64_t)N_CELLS) return ERR_INPUT;
    const int n = (int)n_cells, m = (int)uniq.size();
    if (n == 0) { *verified = 1; return OK; }  // verifier.rs:90-93

    std::lock_guard<std::mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        hipStream_t st = stream_;
        // ---- stage inputs
        std::vector<uint8_t> hc((size_t)m * 48), hp((size_t)n * 48), hcells((size_t)n * BYTES_PER_CELL);
        std::vector<int> hidx(n);
        for (int i = 0; i < m; i++) memcpy(&hc[(size_t)i * 48], uniq[i], 48);
        for (int k = 0; k < n; k++) {
            memcpy(&hp[(size_t)k * 48], proofs[k], 48);
            memcpy(&hcells[(size_t)k * BYTES_PER_CELL], cells[k], BYTES_PER_CELL);
            hidx[k] = (int)cell_indices[k];
        }
        DevBuf d_cb(hc.size()), d_pb(hp.size()), d_cellb(hcells.size());
        DevBuf d_comm((size_t)m * sizeof(G1Affine)), d_prf((size_t)n * sizeof(G1Affine)), d_evals((size_t)n * CELL_LEN * sizeof(Fr));
        DevBuf d_stc((size_t)m * sizeof(int)), d_stp((size_t)n * sizeof(int)), d_ste(sizeof(int));
        DevBuf d_idx((size_t)n * sizeof(int)), d_row((size_t)n * sizeof(int));
        HIPCK(hipMemcpyAsync(d_cb.p, hc.data(), hc.size(), hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d_pb.p, hp.data(), hp.size(), hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d_cellb.p, hcells.data(), hcells.size(), hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d_idx.p, hidx.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d_row.p, row.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(hipMemsetAsync(d_ste.p, 0, sizeof(int), st));
        // ---- deserialisation with on-curve + subgroup checks (serialization/src/lib.rs:69-99), on the GPU
        launch::g1_decompress((const uint8_t*)d_cb.p, d_comm.p, (int*)d_stc.p, m, 1, st);
        launch::g1_decompress((const uint8_t*)d_pb.p, d_prf.p, (int*)d_stp.p, n, 1, st);
        launch::cells_to_fr((const uint8_t*)d_cellb.p, d_evals.p, nullptr, (int*)d_ste.p, n, st);
        std::vector<int> stc(m), stp(n);
        int ste = 0;
        HIPCK(hipMemcpyAsync(stc.data(), d_stc.p, m * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(stp.data(), d_stp.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(&ste, d_ste.p, sizeof(int), hipMemcpyDeviceToHost, st));
        // ---- Fiat-Shamir challenge on the host while the GPU decompresses (verifier.rs:269-328).
        // Valid inputs are canonical encodings, so the transcript is the input bytes themselves.
        Sha256 sh;
        auto be64 = [](uint64_t v, uint8_t* o) { for (int b = 0; b < 8; b++) o[b] = (uint8_t)(v >> (56 - 8 * b)); };
        uint8_t hdr[16 + 32];
        memcpy(hdr, "RCKZGCBATCH__V1_", 16);
        be64(N_BLOB, hdr + 16); be64(CELL_LEN, hdr + 24); be64((uint64_t)m, hdr + 32); be64((uint64_t)n, hdr + 40);
        sh.update(hdr, sizeof hdr);
        sh.update(hc.data(), hc.size());
        for (int k = 0; k < n; k++) {
            uint8_t ix[16];
            be64((uint64_t)row[k], ix); be64(cell_indices[k], ix + 8);
            sh.update(ix, 16);
            sh.update(&hcells[(size_t)k * BYTES_PER_CELL], BYTES_PER_CELL);
            sh.update(&hp[(size_t)k * 48], 48);
        }
        uint8_t dig[32];
        sh.finish(dig);
        Fr r = reduce_be32(dig);
        HIPCK(hipStreamSynchronize(st));
        for (int s : stc) if (s) return ERR_G1;      // order of the reference: commitments, proofs, cells
        for (int s : stp) if (s) return ERR_G1;
        if (ste) return ERR_SCALAR;

        // ---- scalars
        Fr8 tab[24];
        Fr cur = r;
        for (int i = 0; i < 24; i++) { tab[i] = to8(cur); cur = sqr(cur); }
        DevBuf d_rp((size_t)n * sizeof(Fr)), d_s1((size_t)n * sizeof(Fr)), d_s2((size_t)n * sizeof(Fr)), d_w((size_t)m * sizeof(Fr));
        launch::verify_scalars(tab, (const int*)d_idx.p, d_w8192_, d_rp.p, d_s1.p, d_s2.p, n, st);
        launch::verify_weights(d_rp.p, (const int*)d_row.p, d_w.p, n, m, st);
        const int ib = n < 256 ? n : 256;
        DevBuf d_part((size_t)ib * 64 * sizeof(Fr)), d_interp(64 * sizeof(Fr));
        launch::interp(d_evals.p, (const int*)d_idx.p, d_rp.p, d_w8192_, inv64_, d_part.p, ib, d_interp.p, n, st);
        // ---- the four lincombs (verifier.rs:186,200,224,235): block partial sums, then two totals
        const int pb_n = (n + 63) / 64, pb_m = (m + 63) / 64;
        DevBuf d_parts((size_t)(2 * pb_n + pb_m + 1) * sizeof(G1Jac)), d_out(2 * sizeof(G1Affine));
        G1Jac* parts = (G1Jac*)d_parts.p;
        launch::lincomb_partial(d_prf.p, d_s1.p, n, parts, st);                       // sum r^k pi_k
        launch::lincomb_partial(d_prf.p, d_s2.p, n, parts + pb_n, st);                // sum r^k h^64 pi_k
        launch::lincomb_partial(d_comm.p, d_w.p, m, parts + 2 * pb_n, st);            // sum w_row C_row
        launch::lincomb_partial(d_srs_, d_interp.p, 64, parts + 2 * pb_n + pb_m, st); // - commit(interpolation poly)
        launch::lincomb_final(parts, pb_n, pb_n + pb_m + 1, d_out.p, st);
        G1Affine out[2];
        HIPCK(hipMemcpyAsync(out, d_out.p, sizeof out, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        // ---- pairing check e(sum r^k pi_k, [tau^64]_2) * e(C - I + weighted proofs, -[1]_2) == 1 (verifier.rs:242-259)
        const pairing::G2Prepared* q[2] = {g2_tau_.get(), g2_neg_gen_.get()};
        *verified = pairing::product_is_one(out, q, 2) ? 1 : 0;
    } catch (const std::exception& e) {
        err_ = e.what();
        return ERR_DEVICE;
    }
    return OK;
}

// ------------------------------------------------------------------------------------------------
// recovery: returns a Status; on OK the 4096 coefficients of the blob polynomial are in d_coeffs_[0].
int Engine::recover_to_coeffs(uint64_t n_cells, const uint8_t* const* cells, const uint64_t* cell_indices) {
    hipStream_t st = stream_;
    const int n = (int)n_cells;
    ensure_workspace(1);
    // Z'(y) = prod_{missing i} (y - omega_128^i) in domain order (recovery.rs:43-58, reed_solomon.rs:236-241,
    // poly_coeff.rs:109-115); indices in domain order are the bit-reversed cell indices (cosets.rs:186-195).
    bool present[N_CELLS] = {false};
    for (int k = 0; k < n; k++) present[brp7((int)cell_indices[k])] = true;
    Fr w128 = one<FrParams>();
    {   // omega_128 = 7^((r-1)/128)
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
        e[0] -= 1;
        for (int s = 0; s < 7; s++)
            for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
        Fr acc = one<FrParams>(), base = fr_u64(7);
        for (int i = 255; i >= 0; i--) { acc = sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base); }
        w128 = acc;
    }
    std::vector<Fr> roots(N_CELLS);
    roots[0] = one<FrParams>();
    for (int i = 1; i < N_CELLS; i++) roots[i] = mul(roots[i - 1], w128);
    std::vector<Fr> zp(1, one<FrParams>());
    for (int i = 0; i < N_CELLS; i++) {
        if (present[i]) continue;
        Fr nr = neg(roots[i]);
        zp.push_back(zp.back());
        for (size_t k = zp.size() - 2; k >= 1; k--) zp[k] = add(mul(zp[k], nr), zp[k - 1]);
        zp[0] = mul(zp[0], nr);
    }
    // per-cell values of Z on the domain and inverse values on the coset 7 * domain:
    //   Z(omega_8192^n) = Z'(omega_128^(n mod 128)),  Z(7 omega_8192^n) = Z'(7^64 omega_128^(n mod 128)),
    // and position q of the bit-reversed (cell-order) array has n mod 128 = brp7(q / 64).
    Fr seven64 = one<FrParams>();
    { Fr b = fr_u64(7); for (int i = 0; i < 6; i++) b = sqr(b); seven64 = b; }
    auto horner = [&](const Fr& x) { Fr acc = zero<FrParams>(); for (size_t k = zp.size(); k-- > 0;) acc = add(mul(acc, x), zp[k]); return acc; };
    std::vector<Fr> zeval(N_CELLS), zcinv(N_CELLS);
    for (int c = 0; c < N_CELLS; c++) {
        Fr x = roots[brp7(c)];
        zeval[c] = horner(x);
        zcinv[c] = inv(horner(mul(seven64, x)));  // never zero: Z has no roots on the coset (reed_solomon.rs:356-357)
    }
    std::vector<uint8_t> hcells((size_t)n * BYTES_PER_CELL);
    std::vector<int> slot(n);
    for (int k = 0; k < n; k++) { memcpy(&hcells[(size_t)k * BYTES_PER_CELL], cells[k], BYTES_PER_CELL); slot[k] = (int)cell_indices[k]; }
    DevBuf d_cellb(hcells.size()), d_slot(n * sizeof(int)), d_E(N_EXT * sizeof(Fr)), d_T(N_EXT * sizeof(Fr)), d_U(N_EXT * sizeof(Fr));
    DevBuf d_zeval(N_CELLS * sizeof(Fr)), d_zcinv(N_CELLS * sizeof(Fr)), d_st(sizeof(int));
    HIPCK(hipMemcpyAsync(d_cellb.p, hcells.data(), hcells.size(), hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_slot.p, slot.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_zeval.p, zeval.data(), N_CELLS * sizeof(Fr), hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_zcinv.p, zcinv.data(), N_CELLS * sizeof(Fr), hipMemcpyHostToDevice, st));
    HIPCK(hipMemsetAsync(d_E.p, 0, N_EXT * sizeof(Fr), st));
    HIPCK(hipMemsetAsync(d_st.p, 0, sizeof(int), st));
    launch::cells_to_fr((const uint8_t*)d_cellb.p, d_E.p, (const int*)d_slot.p, (int*)d_st.p, n, st);   // E in cell order
    launch::rec_dit_half(1, d_E.p, d_zeval.p, d_T.p, d_w8192_, st);                                      // (E*Z) -> IFFT ...
    launch::rec_dit_last(1, d_T.p, d_coset_, n_inv8192_, d_U.p, nullptr, nullptr, d_w8192_, 0, st);      // ... * 7^i
    launch::rec_dif_half(1, d_U.p, d_zcinv.p, d_E.p, d_w8192_, st);                                      // coset FFT, / Z
    launch::rec_dit_half(1, d_E.p, nullptr, d_T.p, d_w8192_, st);                                        // coset IFFT ...
    launch::rec_dit_last(1, d_T.p, d_coset_inv_, n_inv8192_, nullptr, d_coeffs_, (int*)d_st.p, d_w8192_, 1, st);  // ... * 7^-i
    int hst = 0;
    HIPCK(hipMemcpyAsync(&hst, d_st.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if (hst & 1) return ERR_SCALAR;
    if (hst & 4) return ERR_RECOVERY;
    return OK;
}

int Engine::recover_cells_and_kzg_proofs_host(uint64_t n_cells, const uint8_t* const* cells, uint64_t n_indices,
                                              const uint64_t* cell_indices, uint8_t* const* out_cells,
                                              uint8_t* const* out_proofs) {
    // validate_recovery_inputs (recovery.rs:90-146)
    if (n_indices != n_cells) return ERR_INPUT;
    for (uint64_t i = 0; i < n_indices; i++)
        if (cell_indices[i] >= (uint64_t)N_CELLS) return ERR_INPUT;
    for (uint64_t i = 1; i < n_indices; i++)
        if (!(cell_indices[i - 1] < cell_indices[i])) return ERR_INPUT;
    if (n_indices < (uint64_t)N_CELLS / 2 || n_indices > (uint64_t)N_CELLS) return ERR_INPUT;

    std::lock_guard<std::mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        int rc = recover_to_coeffs(n_cells, cells, cell_indices);
        if (rc) return rc;
        // compute_multi_opening_proofs(Input::PolyCoeff) = stages C..I (prover.rs:164-170)
        DevBuf d_c((size_t)N_CELLS * BYTES_PER_CELL), d_p((size_t)N_CELLS * 48);
        launch::coeffs_to_cells(1, d_coeffs_, (uint8_t*)d_c.p, d_w8192_, stream_);
        run_proofs_from_coeffs(1, (uint8_t*)d_p.p, stream_);
        std::vector<uint8_t> hc((size_t)N_CELLS * BYTES_PER_CELL), hp((size_t)N_CELLS * 48);
        HIPCK(hipMemcpyAsync(hc.data(), d_c.p, hc.size(), hipMemcpyDeviceToHost, stream_));
        HIPCK(hipMemcpyAsync(hp.data(), d_p.p, hp.size(), hipMemcpyDeviceToHost, stream_));
        HIPCK(hipStreamSynchronize(stream_));
        for (int k = 0; k < N_CELLS; k++) {
            memcpy(out_cells[k], &hc[(size_t)k * BYTES_PER_CELL], BYTES_PER_CELL);
            memcpy(out_proofs[k], &hp[(size_t)k * 48], 48);
        }
    } catch (const std::exception& e) {
        err_ = e.what();
        return ERR_DEVICE;
    }
    return OK;
}

}  // namespace kzg
