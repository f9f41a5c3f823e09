// EIP-4844 single-point operations on the same kernels (first "next" row).
// Reference: crates/eip4844/src/prover.rs:32-88 (compute_kzg_proof, compute_blob_kzg_proof),
// crates/eip4844/src/verifier.rs:18-262 (verify_kzg_proof, verify_blob_kzg_proof, verify_blob_kzg_proof_batch and the
// two Fiat-Shamir transcripts), crates/cryptography/kzg_single_open/src/{prover.rs:33-65, verifier.rs:33-108}.
// GPU: blob -> coefficients (k_blob_to_coeffs), quotient by (X - z) (k_quotient_by_linear), MSM against the
// monomial SRS window table, decompression with subgroup checks, bucket MSMs of the verification equation.
// Host: SHA-256 transcripts, the handful of Fr products of the batch weights, the 2-pairing check.
#include "engine.hpp"
#include "curve.hpp"
#include "host_pairing.hpp"
#include "launch.hpp"
#include "sha256.hpp"

#include <cstring>
#include <stdexcept>
#include <string>

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

static constexpr int N_BLOB = 4096, BYTES_PER_BLOB = 131072;


static bool fr_from_be_canonical(Fr& out_mont, const uint8_t* b) {  // deserialize_bytes_to_scalar (serialization/src/lib.rs:50-63)
    Fr x;
    for (int i = 0; i < 8; i++)
        x.v[7 - i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    if (geq_mod<FrParams>(x.v)) return false;
    out_mont = to_mont(x);
    return true;
}
static void fr_to_be(uint8_t* o, const Fr& canon) {
    for (int i = 0; i < 8; i++) {
        uint32_t w = canon.v[7 - i];
        o[4 * i] = (uint8_t)(w >> 24); o[4 * i + 1] = (uint8_t)(w >> 16); o[4 * i + 2] = (uint8_t)(w >> 8); o[4 * i + 3] = (uint8_t)w;
    }
}
static Fr reduce_be32_4844(const uint8_t* b) {  // reduce_bytes_to_scalar_bias
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
// compute_fiat_shamir_challenge (eip4844/src/verifier.rs:155-196)
static Fr fs_challenge(const uint8_t* blob, const uint8_t* commitment) {
    Sha256 sh;
    uint8_t hdr[32];
    memcpy(hdr, "FSBLOBVERIFY_V1_", 16);
    memset(hdr + 16, 0, 16);
    hdr[16 + 14] = 0x10;  // u128 big-endian 4096
    sh.update(hdr, 32);
    sh.update(blob, BYTES_PER_BLOB);
    sh.update(commitment, 48);
    uint8_t dig[32];
    sh.finish(dig);
    return reduce_be32_4844(dig);
}

// blobs -> (status, y_i canonical, optionally proof_i = commit(quotient_i)); z_i Montgomery.
// Shared by compute_kzg_proof / compute_blob_kzg_proof (want_proofs) and the blob verifiers (y only).
int Engine::open_blobs_at(int n, const uint8_t* const* blobs, const Fr8* z_mont, bool want_proofs, uint8_t* h_proofs,
                          Fr8* h_y_canon, int* h_status) {
    hipStream_t st = stream_;
    ensure_workspace(n);
    const int bp = ((n + 63) / 64) * 64;
    PoolBuf d_blobs(*this, (size_t)n * BYTES_PER_BLOB), d_z(*this, (size_t)n * 32), d_y(*this, (size_t)n * 32), d_pr(*this, (size_t)n * 48);
    for (int b = 0; b < n; b++)
        HIPCK(hipMemcpyAsync((uint8_t*)d_blobs.p + (size_t)b * BYTES_PER_BLOB, blobs[b], BYTES_PER_BLOB, hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_z.p, z_mont, (size_t)n * 32, hipMemcpyHostToDevice, st));
    HIPCK(hipMemsetAsync(d_status_, 0, n * sizeof(int), st));
    launch::blob_to_coeffs(n, (const uint8_t*)d_blobs.p, d_coeffs_, nullptr, d_status_, d_w29_, n_inv4096_, st);
    launch::quotient_by_linear(n, d_coeffs_, d_z.p, d_canon_, d_y.p, st);
    if (want_proofs) {
        // proof = g1_lincomb(g1s[..4095], quotient) (kzg_single_open/src/prover.rs:40-43): the commitment MSM path
        const int fmt = arena_signed_ ? launch::FMT_JACS : launch::FMT_JACQ;
        launch::g1_set_inf(d_X_, (size_t)64 * bp, st, fmt);
        launch_msm(d_canon_, TAB_SRS, d_X_, 64, n, bp, 0, st, fmt);
        launch::g1_sum_positions(d_X_, 64, bp, n, st, fmt);
        launch::g1_compress(d_X_, (uint8_t*)d_pr.p, 1, bp, n, st, fmt);
        HIPCK(hipMemcpyAsync(h_proofs, d_pr.p, (size_t)n * 48, hipMemcpyDeviceToHost, st));
    }
    HIPCK(hipMemcpyAsync(h_y_canon, d_y.p, (size_t)n * 32, hipMemcpyDeviceToHost, st));
    HIPCK(hipMemcpyAsync(h_status, d_status_, n * sizeof(int), hipMemcpyDeviceToHost, st));
    SYNC_CHECKED(st);
    return OK;
}

int Engine::compute_kzg_proof_host(const uint8_t* blob, const uint8_t* z_bytes, uint8_t* out_proof, uint8_t* out_y) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        Fr z;
        bool z_ok = fr_from_be_canonical(z, z_bytes);
        if (!z_ok) z = zero<FrParams>();
        Fr8 z8, y8;
        memcpy(&z8, &z, 32);
        int st = 0;
        uint8_t proof[48];
        const uint8_t* bl[1] = {blob};
        open_blobs_at(1, bl, &z8, true, proof, &y8, &st);
        if (st || !z_ok) return ERR_SCALAR;  // blob elements are checked first in the reference, then z
        memcpy(out_proof, proof, 48);
        Fr y;
        memcpy(&y, &y8, 32);
        fr_to_be(out_y, y);
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// decompress + subgroup-check a few points on the GPU; returns per-point status
static void check_points(Engine* eng, const uint8_t* bytes, int n, void* d_out_affine, int* h_status, hipStream_t st, const Fp12w& beta) {
    PoolBuf d_b(*eng, (size_t)n * 48), d_st(*eng, (size_t)n * sizeof(int));
    HIPCK(hipMemcpyAsync(d_b.p, bytes, (size_t)n * 48, hipMemcpyHostToDevice, st));
    launch::g1_decompress((const uint8_t*)d_b.p, d_out_affine, (int*)d_st.p, n, 1, beta, st);
    HIPCK(hipMemcpyAsync(h_status, d_st.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
    SYNC_CHECKED(st);

}

int Engine::compute_blob_kzg_proof_host(const uint8_t* blob, const uint8_t* commitment, uint8_t* out_proof) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        Fr z = fs_challenge(blob, commitment);
        Fr8 z8, y8;
        memcpy(&z8, &z, 32);
        int st = 0, cst = 0;
        uint8_t proof[48];
        const uint8_t* bl[1] = {blob};
        open_blobs_at(1, bl, &z8, true, proof, &y8, &st);
        if (st) return ERR_SCALAR;
        PoolBuf d_pt(*this, sizeof(G1Affine));
        check_points(this, commitment, 1, d_pt.p, &cst, stream_, beta_);  // only validated (prover.rs:73-75)
        if (cst) return ERR_G1;
        memcpy(out_proof, proof, 48);
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// e(sum_i a_i P_i, -[1]_2) * e(sum_i b_i Q_i, [tau]_2) == 1 with the two sums done as bucket MSMs on the GPU.
// d_points: [n_total] affine (job 0 = first n0 with sc0, job 1 = all n1 with sc1).  Returns 1 / 0.
int Engine::pairing_check_4844(const void* d_points, const std::vector<Fr8>& sc0, const std::vector<Fr8>& sc1) {
    hipStream_t st = stream_;
    const int n0 = (int)sc0.size(), n1 = (int)sc1.size();
    PoolBuf d_s0(*this, (size_t)n0 * 32), d_s1(*this, (size_t)n1 * 32), d_ws(*this, launch::pip_workspace_bytes(n1 > n0 ? n1 : n0)), d_out(*this, 2 * sizeof(G1Affine));
    HIPCK(hipMemcpyAsync(d_s0.p, sc0.data(), (size_t)n0 * 32, hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_s1.p, sc1.data(), (size_t)n1 * 32, hipMemcpyHostToDevice, st));
    launch::msm_pippenger2(d_points, d_s0.p, n0, d_s1.p, n1, d_ws.p, d_out.p, beta_, st);
    G1Affine out[2];
    HIPCK(hipMemcpyAsync(out, d_out.p, sizeof out, hipMemcpyDeviceToHost, st));
    SYNC_CHECKED(st);
    // out[0] = rhs (pairs with [tau]_2), out[1] = lhs (pairs with -[1]_2)
    const pairing::G2Prepared* q[2] = {g2_tau1_.get(), g2_neg_gen_.get()};
    return pairing::product_is_one(out, q, 2) ? 1 : 0;
}

static Fr8 canon8(const Fr& mont) { Fr c = from_mont(mont); Fr8 r; memcpy(&r, &c, 32); return r; }

// Verifier::verify_kzg_proof (kzg_single_open/src/verifier.rs:33-57): e(C - yG, -G2) e(pi, [tau - z]_2) == 1, evaluated as
// e(C - yG + z pi, -G2) e(pi, [tau]_2) == 1 (bilinearity; the shape the reference's batch verifier uses, :76-107).
int Engine::verify_kzg_proof_host(const uint8_t* commitment, const uint8_t* z_bytes, const uint8_t* y_bytes, const uint8_t* proof,
                                  int* verified) {
    *verified = 0;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        // point array [pi | C | G]
        PoolBuf d_pts(*this, 3 * sizeof(G1Affine));
        uint8_t two[96];
        memcpy(two, proof, 48);
        memcpy(two + 48, commitment, 48);
        int pst[2];
        check_points(this, two, 2, d_pts.p, pst, stream_, beta_);
        if (pst[1]) return ERR_G1;  // commitment first, then proof (eip4844/src/verifier.rs:29-33)
        if (pst[0]) return ERR_G1;
        Fr z, y;
        if (!fr_from_be_canonical(z, z_bytes)) return ERR_SCALAR;
        if (!fr_from_be_canonical(y, y_bytes)) return ERR_SCALAR;
        launch::copy_affine(d_srs_, (G1Affine*)d_pts.p + 2, 1, stream_);  // G = [1]_1 = g1_monomial[0]
        std::vector<Fr8> s0 = {canon8(one<FrParams>())};
        std::vector<Fr8> s1 = {canon8(z), canon8(one<FrParams>()), canon8(neg(y))};
        *verified = pairing_check_4844(d_pts.p, s0, s1);
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::verify_blob_kzg_proof_batch_host(uint64_t n_blobs, const uint8_t* const* blobs, uint64_t n_commitments,
                                             const uint8_t* const* commitments, uint64_t n_proofs, const uint8_t* const* proofs,
                                             int* verified) {
    *verified = 0;
    if (!(n_blobs == n_commitments && n_blobs == n_proofs)) return ERR_INPUT;  // eip4844/src/verifier.rs:87-95
    const int n = (int)n_blobs;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        // challenges z_i and evaluations y_i = p_i(z_i)
        std::vector<Fr> zs(n);
        std::vector<Fr8> z8(n), y8(n);
        std::vector<int> bst(n);
        for (int i = 0; i < n; i++) { zs[i] = fs_challenge(blobs[i], commitments[i]); memcpy(&z8[i], &zs[i], 32); }
        if (n) open_blobs_at(n, blobs, z8.data(), false, nullptr, y8.data(), bst.data());
        for (int i = 0; i < n; i++) if (bst[i]) return ERR_SCALAR;          // blobs first,
        // point array [proofs n | commitments n | G]
        PoolBuf d_pts(*this, (size_t)(2 * n + 1) * sizeof(G1Affine));
        std::vector<uint8_t> pb((size_t)2 * n * 48 + 1);
        std::vector<int> pst(2 * n + 1);
        for (int i = 0; i < n; i++) { memcpy(&pb[(size_t)i * 48], proofs[i], 48); memcpy(&pb[(size_t)(n + i) * 48], commitments[i], 48); }
        if (n) check_points(this, pb.data(), 2 * n, d_pts.p, pst.data(), stream_, beta_);
        for (int i = 0; i < n; i++) if (pst[n + i]) return ERR_G1;          // then commitments,
        for (int i = 0; i < n; i++) if (pst[i]) return ERR_G1;              // then proofs (verifier.rs:97-113)
        launch::copy_affine(d_srs_, (G1Affine*)d_pts.p + 2 * n, 1, stream_);
        // compute_r_powers_for_verify_kzg_proof_batch (verifier.rs:201-262)
        Sha256 sh;
        uint8_t hdr[32];
        memcpy(hdr, "RCKZGBATCH___V1_", 16);
        for (int b = 0; b < 8; b++) { hdr[16 + b] = (uint8_t)((uint64_t)N_BLOB >> (56 - 8 * b)); hdr[24 + b] = (uint8_t)((uint64_t)n >> (56 - 8 * b)); }
        sh.update(hdr, 32);
        for (int i = 0; i < n; i++) {
            uint8_t zy[64];
            fr_to_be(zy, from_mont(zs[i]));
            Fr yc;
            memcpy(&yc, &y8[i], 32);
            fr_to_be(zy + 32, yc);
            sh.update(commitments[i], 48);
            sh.update(zy, 64);
            sh.update(proofs[i], 48);
        }
        uint8_t dig[32];
        sh.finish(dig);
        Fr r = reduce_be32_4844(dig);
        // lhs = sum r^i C_i - (sum r^i y_i) G + sum r^i z_i pi_i ; rhs = sum r^i pi_i   (kzg_single_open/src/verifier.rs:76-99)
        std::vector<Fr8> s0(n), s1(2 * n + 1);
        Fr cur = one<FrParams>(), ysum = zero<FrParams>();
        for (int i = 0; i < n; i++) {
            Fr yc;
            memcpy(&yc, &y8[i], 32);
            s0[i] = canon8(cur);
            s1[i] = canon8(mul(cur, zs[i]));
            s1[n + i] = s0[i];
            ysum = add(ysum, mul(cur, to_mont(yc)));
            cur = mul(cur, r);
        }
        s1[2 * n] = canon8(neg(ysum));
        *verified = pairing_check_4844(d_pts.p, s0, s1);
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// verify_blob_kzg_proof (eip4844/src/verifier.rs:50-76): single-opening check at the Fiat-Shamir point
int Engine::verify_blob_kzg_proof_host(const uint8_t* blob, const uint8_t* commitment, const uint8_t* proof, int* verified) {
    *verified = 0;
    std::lock_guard<std::recursive_mutex> whole_call(mu_);
    int st = OK;
    Fr z, y;
    {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        try {
            HIPCK(hipSetDevice(dev_));
            z = fs_challenge(blob, commitment);
            Fr8 z8, y8;
            memcpy(&z8, &z, 32);
            int bst = 0;
            const uint8_t* bl[1] = {blob};
            open_blobs_at(1, bl, &z8, false, nullptr, &y8, &bst);
            if (bst) return ERR_SCALAR;
            memcpy(&y, &y8, 32);
        } catch (const std::exception& e) {
            set_error(e);
            return ERR_DEVICE;
        }
    }
    uint8_t zb[32], yb[32];
    fr_to_be(zb, from_mont(z));
    fr_to_be(yb, y);
    st = verify_kzg_proof_host(commitment, zb, yb, proof, verified);
    return st;
}

}  // namespace kzg
