// G1 helpers: identity fill, normalise + compress, decompression, small set-up gathers.
// Reference: g1_batch_normalize (crates/cryptography/bls12_381/src/lib.rs:56-104), serialize_g1_compressed /
// deserialize_compressed_g1 (crates/serialization/src/lib.rs:69-99), SRS vectors (fk20/prover.rs:88-104).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "curve30.hpp"
#include "g1_subgroup.hpp"
#include "g1_coop.hpp"
#include "g1_coop30.hpp"
#include "launch.hpp"

namespace kzg {

// The batched point arrays X[pos * stride + lane] hold JacQ (unsaturated Montgomery-406 coordinates, 168 B).
__global__ void k_g1_set_inf(JacQ* X, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[i] = jacq_inf();
}
__global__ void k_g1_set_inf_s(JacS* X, size_t n) {  // an array in the signed 13 x 30-bit form (launch::FMT_JACS)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[i] = jacs_inf();
}
// a point of either array format as saturated Jacobian coordinates (what the normalisation below computes in)
__device__ __forceinline__ G1Jac jac_of(const JacQ& p) { return jac_from_jacq(p); }
__device__ __forceinline__ G1Jac jac_of(const JacS& p) {
    G1Jac r;
    r.x = fp_from_fs(p.x);
    r.y = fp_from_fs(p.y);
    r.z = fp_from_fs(p.z);  // canonical: the identity arrives as z = 0
    return r;
}

// Stage G+I: normalise and compress (lib.rs:56-104, serialization/src/lib.rs:84-86).
// X[pos * stride + slice] -> out[(slice * n_pos + pos) * 48]
// A thread normalises NB = 4 consecutive positions of its slice with ONE inversion (the reference's g1_batch_normalize,
// lib.rs:56-104, is Montgomery's trick too): z0 z1 z2 z3 inverted once, the four inverses peeled off with 6 multiplications;
// an identity (z = 0) takes part as z = 1 and is encoded as the identity afterwards.  The binary-GCD inversion is ~80 % of a
// one-point normalisation, so this is ~3x fewer instructions per point; n_pos not a multiple of 4: the tail takes part as ones.
// (NB = 1 for batches that leave the chip part empty: there the kernel lasts as long as one thread's chain.)
template <int NB, class Pt>
__global__ __launch_bounds__(64) void k_g1_compress(const Pt* __restrict__ X, uint8_t* __restrict__ out, int n_pos,
                                                    int stride, int n_slices) {
    const int pos0 = blockIdx.x * NB, slice_of_lane = blockIdx.y * 64 + threadIdx.x;
    // the lanes behind the last blob repeat its work and store nothing (k_g1slp.hip, k_slp_mulc_s: a partly filled wave is the slower one)
    if ((int)(blockIdx.y * 64) >= n_slices) return;
    const int slice = slice_of_lane < n_slices ? slice_of_lane : n_slices - 1;
    G1Jac P[NB];
    Fp pre[NB];  // pre[i] = z_0 ... z_i (identities and the tail count as 1)
    bool inf[NB];
#pragma unroll 1
    for (int i = 0; i < NB; i++) {
        inf[i] = true;
        if (pos0 + i < n_pos) {
            P[i] = jac_of(X[(size_t)(pos0 + i) * stride + slice]);
            inf[i] = is_inf(P[i]);
        }
        const Fp z = inf[i] ? one<FpParams>() : P[i].z;
        pre[i] = i ? mul(pre[i - 1], z) : z;
    }
    Fp acc = inv_fast(pre[NB - 1]);  // (z_0 ... z_3)^-1
#pragma unroll 1
    for (int i = NB - 1; i >= 0; i--) {
        const Fp zi = i ? mul(acc, pre[i - 1]) : acc;  // z_i^-1
        if (i) acc = mul(acc, inf[i] ? one<FpParams>() : P[i].z);
        if (pos0 + i < n_pos) {
            G1Affine a = aff_inf();
            if (!inf[i]) {
                const Fp zi2 = sqr(zi);
                a.x = mul(P[i].x, zi2);
                a.y = mul(P[i].y, mul(zi2, zi));
            }
            uint8_t buf[48];
            g1_compress(buf, a);
            uint32_t* dst = reinterpret_cast<uint32_t*>(out + ((size_t)slice * n_pos + pos0 + i) * 48);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(buf);
            if (slice_of_lane < n_slices) {
#pragma unroll
                for (int k = 0; k < 12; k++) dst[k] = src[k];
            }
        }
    }
}

// sum over positions: out[slice] = sum_pos X[pos*stride + slice]   (final fold of the commitment MSM)
// A block per slice, a lane per position, a six-level tree through LDS (its idle lanes sharing the additions): the lane-per-slice loop this replaces was a chain of 63
// dependent additions -- 1.0 ms whatever the batch, two thirds of a single blob's commitment (round 4: 1.59 -> 0.7 ms), and
// still the longer way at 2048 blobs (32 waves for 1.0 ms against 2048 short ones).  X[slice] (position 0) is read by its own
// block only, so the sum may land there.
__device__ __forceinline__ void sum_fold64(JacQ* T, int t) { coop_tree_fold<64>(T, 32, t); }    // four lanes per addition (g1_coop.hpp)
__device__ __forceinline__ void sum_fold64(JacS* T, int t) { coop4_tree_fold<64>(T, 32, t); }   // ... in the signed field (g1_coop30.hpp)
__device__ __forceinline__ void set_inf(JacQ& p) { p = jacq_inf(); }
__device__ __forceinline__ void set_inf(JacS& p) { p = jacs_inf(); }
template <class Pt>
__global__ __launch_bounds__(64) void k_g1_sum_positions(Pt* __restrict__ X, int n_pos, int stride, int n_slices) {
    __shared__ Pt T[64];
    const int slice = blockIdx.x, t = threadIdx.x;
    Pt acc;
    if (t < n_pos) acc = X[(size_t)t * stride + slice];
    else set_inf(acc);
    for (int p = t + 64; p < n_pos; p += 64) acc = add(acc, X[(size_t)p * stride + slice]);
    T[t] = acc;
    sum_fold64(T, t);
    if (t == 0) X[slice] = T[0];
}

// ------------------------------------------------------------------------------------------------
// Decompression in the unsaturated field (the dependent chain of this kernel is what a single-proof verification waits
// for): same decisions as curve.hpp's g1_decompress + g1_in_subgroup_endo, which stay as the host / reference forms.
//   y = (x^3 + 4)^((p+1)/4) with fixed 4-bit windows: 380 squarings + <= 95 + 14 multiplications;
//   subgroup test [z^2]P - P == phi(P) with mixed additions of the affine P: 126 doublings + 11 mixed additions.
__device__ __forceinline__ Fq<2> fq_sqrt_candidate(const Fq<2>& a) {
    constexpr uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};  // (p + 1) / 4
    Fq<2> tbl[16];
    tbl[1] = a;
#pragma unroll 1
    for (int i = 2; i < 16; i++) tbl[i] = mul(tbl[i - 1], a);
    Fq<2> acc = relax<2>(fq_one());
    bool started = false;
#pragma unroll 1
    for (int nib = 95; nib >= 0; nib--) {
        const int d = (E[nib >> 3] >> ((nib & 7) * 4)) & 15;  // wave-uniform: the exponent is a constant
        if (started) {
#pragma unroll 1
            for (int k = 0; k < 4; k++) acc = sqr(acc);
        }
        if (d) {
            acc = started ? mul(acc, tbl[d]) : tbl[d];
            started = true;
        }
    }
    return acc;
}
// rc 0 ok (out = affine Montgomery-384 point), 1 bad encoding / x >= p / not on the curve, 2 not in the subgroup
// quad >= 0: four lanes decode the same point and share the doublings of its subgroup test (g1_coop.hpp)
__device__ __forceinline__ int g1_decompress_q(G1Affine& out, const uint8_t* in, bool subgroup_check, const Fq<1>& beta, int quad = -1) {
    const uint8_t b0 = in[0];
    const bool compressed = (b0 >> 7) & 1, infinity = (b0 >> 6) & 1, sign = (b0 >> 5) & 1;
    if (!compressed) return 1;
    Fp x;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t w = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | in[4 * i + 3];
        if (i == 0) w &= 0x1fffffffu;
        x.v[11 - i] = w;
    }
    if (infinity) {
        if (sign || !is_zero(x)) return 1;
        out = aff_inf();
        return 0;
    }
    if (geq_mod<FpParams>(x.v)) return 1;
    const Fp xm = to_mont(x);
    const Fq<1> xq = fq_from_fp(xm);
    Fq<1> four = fq_one();
    four = canonical(dbl2(four));
    const Fq<2> y2 = relax<2>(canonical(add(mul(sqr(xq), xq), four)));
    const Fq<2> yc = fq_sqrt_candidate(y2);
    if (!is_zero_slow(sub(sqr(yc), y2))) return 1;  // not a square: x is not the abscissa of a curve point
    Fp ym = fp_from_fq(yc);
    if (fp_is_lex_largest(ym) != sign) ym = neg(ym);
    out.x = xm;
    out.y = ym;
    if (!subgroup_check) return 0;
    AffQ pa;
    pa.x = xq;
    pa.y = fq_from_fp(ym);
    return (quad >= 0 ? g1_in_subgroup_coop(pa, beta, quad) : g1_in_subgroup_q(pa, beta)) ? 0 : 2;
}

// Decompression with validation: thread per point, one wave per block (launch bounds tell the compiler it may use the
// whole register file: without them it budgets for 1024-thread blocks and spills several hundred VGPRs).
// subgroup_check: 0 none (the embedded SRS, trusted_setup/src/lib.rs:80-86), 1 endomorphism test (production)
// Two independent (input, output, status) segments per launch: a verification decodes its proofs and its commitments in one
// go (a launch of 64 points costs the same ~1 ms of dependent doublings as one of 8192).
__global__ __launch_bounds__(64) void k_g1_decompress(const uint8_t* __restrict__ in0, G1Affine* __restrict__ out0,
                                                      int* __restrict__ status0, int n0, const uint8_t* __restrict__ in1,
                                                      G1Affine* __restrict__ out1, int* __restrict__ status1, int n1,
                                                      int subgroup_check, Fq<1> beta_q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 + n1) i = n0 + n1 - 1;  // (the lanes behind the last point repeat its work -- same values to the same places: a wave with few lanes in use is the slow one, k_g1slp.hip)
    const bool second = i >= n0;
    if (second) i -= n0;
    const uint8_t* in = second ? in1 : in0;
    G1Affine a;
    const int rc = g1_decompress_q(a, in + (size_t)i * 48, subgroup_check == 1, beta_q);
    if (rc) a = aff_inf();
    (second ? status1 : status0)[i] = rc;
    (second ? out1 : out0)[i] = a;
}
// k_g1_decompress with the endomorphism test for a few points (one, for the EIP-4844 calls): the 126 dependent doublings of the
// test are the launch's whole duration, so four lanes per point share them; 16 points per block, every lane of a quad
// writes the same result
__global__ __launch_bounds__(64) void k_g1_decompress_coop(const uint8_t* __restrict__ in0, G1Affine* __restrict__ out0,
                                                           int* __restrict__ status0, int n0, const uint8_t* __restrict__ in1,
                                                           G1Affine* __restrict__ out1, int* __restrict__ status1, int n1, Fq<1> beta_q) {
    int i = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (i >= n0 + n1) i = n0 + n1 - 1;  // (the lanes behind the last point repeat its work -- same values to the same places: a wave with few lanes in use is the slow one, k_g1slp.hip)
    const bool second = i >= n0;
    if (second) i -= n0;
    const uint8_t* in = second ? in1 : in0;
    G1Affine a;
    const int rc = g1_decompress_q(a, in + (size_t)i * 48, true, beta_q, threadIdx.x & 3);
    if (rc) a = aff_inf();
    (second ? status1 : status0)[i] = rc;
    (second ? out1 : out0)[i] = a;
}
__global__ __launch_bounds__(64) void k_g1_subgroup_coop(const G1Affine* __restrict__ pts0, int* __restrict__ status0, int n0,
                                                         const G1Affine* __restrict__ pts1, int* __restrict__ status1, int n1, Fq<1> beta_q) {
    int i = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (i >= n0 + n1) i = n0 + n1 - 1;
    const bool second = i >= n0;
    if (second) i -= n0;
    int* st = second ? status1 : status0;
    const int before = st[i];
    if (before != 0) return;
    const G1Affine a = (second ? pts1 : pts0)[i];
    if (is_inf(a)) return;
    if (!g1_in_subgroup_coop(affq_from_affine(a), beta_q, threadIdx.x & 3)) st[i] = 2;
}
// The subgroup test on its own, for points k_g1_decompress has decoded with subgroup_check = 0: a verification runs it on a
// second stream next to the work that only needs the coordinates (k_verify.hip: k_pip_shift), because each of the two is
// a millisecond-long dependent chain of doublings whatever the number of points.  status: 0 -> 0 or 2; others are kept.
__global__ __launch_bounds__(64) void k_g1_subgroup(const G1Affine* __restrict__ pts0, int* __restrict__ status0, int n0,
                                                    const G1Affine* __restrict__ pts1, int* __restrict__ status1, int n1, Fq<1> beta_q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 + n1) i = n0 + n1 - 1;
    const bool second = i >= n0;
    if (second) i -= n0;
    int* st = second ? status1 : status0;
    if (st[i] != 0) return;
    const G1Affine a = (second ? pts1 : pts0)[i];
    if (is_inf(a)) return;
    if (!g1_in_subgroup_q(affq_from_affine(a), beta_q)) st[i] = 2;
}
// The same decoding on the saturated reference forms, for the parity tests only: mode 2 = definitional [r]P == O,
// mode 3 = the saturated endomorphism test.
__global__ __launch_bounds__(64) void k_g1_decompress_reference(const uint8_t* __restrict__ in, G1Affine* __restrict__ out,
                                                                int* __restrict__ status, int n, int mode, Fp beta) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a;
    int rc = g1_decompress(a, in + (size_t)i * 48);
    if (rc == 0 && mode == 3 && !g1_in_subgroup_endo(a, beta)) rc = 2;
    if (rc == 0 && mode == 2 && !g1_in_subgroup(a)) rc = 2;
    if (rc) { status[i] = rc; a = aff_inf(); } else status[i] = 0;
    out[i] = a;
}

// SRS vectors of FK20 (fk20/prover.rs:88-104): t = reverse(g1s)[64:], S_i = t[i::64] (63 points) || O.
__global__ void k_fk20_srs_vectors(const G1Affine* __restrict__ srs, JacQ* __restrict__ X) {
    // X[pos * 64 + i], pos < 128, i < 64
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * 64) return;
    int pos = idx >> 6, i = idx & 63;
    int m = i + 64 * pos;  // index into t
    JacQ v = jacq_inf();
    if (pos < 63) v = jacq_from_jac(to_jac(srs[N_BLOB - 1 - CELL_LEN - m]));
    X[idx] = v;
}
// bases[j][i] = affine(X[brp7(j) * 64 + i])   (transpose of batch_toeplitz.rs:61)
__global__ void k_fk20_gather_bases(const JacQ* __restrict__ X, G1Affine* __restrict__ bases) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * 64) return;
    int j = idx >> 6, i = idx & 63;
    int q = (int)(__brev((unsigned)j) >> 25);
    bases[idx] = to_affine(jac_from_jacq(X[q * 64 + i]));
}
#ifdef KZG_TEST_HOOKS  // stage-level test kernels: compiled into libc_eth_kzg_hooks.so only (csrc/Makefile)
__global__ void k_test_load_points(const uint8_t* in, JacQ* X, int n_lanes, int stride) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * n_lanes) return;
    int pos = idx / n_lanes, lane = idx % n_lanes;
    G1Affine a;
    if (g1_decompress(a, in + ((size_t)lane * 128 + pos) * 48)) a = aff_inf();
    X[(size_t)pos * stride + lane] = jacq_from_jac(to_jac(a));
}
__global__ void k_test_recompress(const G1Affine* in, uint8_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t buf[48];
    g1_compress(buf, in[i]);
    for (int k = 0; k < 48; k++) out[(size_t)i * 48 + k] = buf[k];
}
#endif

// One wave that stays resident for a fixed wall-clock time: the engine's probe of which streams share a hardware queue.
__global__ void k_spin(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

namespace launch {
// Up to this many points a launch of the four-lanes-per-point kernels (g1_coop.hpp) still gives every wave a SIMD of its own
// (16 points per wave, 1,024 SIMDs); beyond it the chip fills and one lane per point is less work.  ETH_KZG_AMD_COOP_POINTS=<n>
// moves the limit, 0 switches the quad form off.
int coop_points_max() {
    static const int v = [] { const char* e = getenv("ETH_KZG_AMD_COOP_POINTS"); return e ? atoi(e) : 8192; }();
    return v;
}
void spin(uint64_t ticks, hipStream_t st) { k_spin<<<1, 64, 0, st>>>(ticks); }
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1misc() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_g1_set_inf));
}
void g1_set_inf(void* X, size_t n, hipStream_t st, int fmt) {
    if (fmt == FMT_JACS) k_g1_set_inf_s<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((JacS*)X, n);
    else k_g1_set_inf<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((JacQ*)X, n);
}
void g1_compress(const void* X, uint8_t* out, int n_pos, int stride, int n_slices, hipStream_t st, int fmt) {
    // measured (128 positions): 2048 lanes 0.39 -> 0.22 ms with four positions per thread; 512 lanes 0.155 -> 0.21, 64 lanes 0.15 -> 0.21 ms
    // (below two rounds of one-point waves the kernel lasts as long as one thread's chain)
    const bool four = (long)n_pos * stride >= 128L * 2048;
    const dim3 g4((n_pos + 3) / 4, stride / 64), g1(n_pos, stride / 64);
    if (fmt == FMT_JACS) {  // signed 13 x 30-bit points: what the prover, recovery and the commitments hand over
        if (four) k_g1_compress<4, JacS><<<g4, 64, 0, st>>>((const JacS*)X, out, n_pos, stride, n_slices);
        else k_g1_compress<1, JacS><<<g1, 64, 0, st>>>((const JacS*)X, out, n_pos, stride, n_slices);
    } else {
        if (four) k_g1_compress<4, JacQ><<<g4, 64, 0, st>>>((const JacQ*)X, out, n_pos, stride, n_slices);
        else k_g1_compress<1, JacQ><<<g1, 64, 0, st>>>((const JacQ*)X, out, n_pos, stride, n_slices);
    }
}
void g1_sum_positions(void* X, int n_pos, int stride, int n_slices, hipStream_t st, int fmt) {
    if (n_slices <= 0) return;
    if (fmt == FMT_JACS) k_g1_sum_positions<JacS><<<n_slices, 64, 0, st>>>((JacS*)X, n_pos, stride, n_slices);
    else k_g1_sum_positions<JacQ><<<n_slices, 64, 0, st>>>((JacQ*)X, n_pos, stride, n_slices);
}
void g1_decompress(const uint8_t* in, void* out, int* status, int n, int subgroup_check, const Fp12w& beta, hipStream_t st) {
    Fp b;
    for (int i = 0; i < 12; i++) b.v[i] = beta.v[i];
    if (subgroup_check >= 2) k_g1_decompress_reference<<<(n + 63) / 64, 64, 0, st>>>(in, (G1Affine*)out, status, n, subgroup_check, b);
    else if (subgroup_check == 1 && n <= coop_points_max())
        k_g1_decompress_coop<<<(n + 15) / 16, 64, 0, st>>>(in, (G1Affine*)out, status, n, nullptr, nullptr, nullptr, 0, fq_from_fp(b));
    else k_g1_decompress<<<(n + 63) / 64, 64, 0, st>>>(in, (G1Affine*)out, status, n, nullptr, nullptr, nullptr, 0, subgroup_check, fq_from_fp(b));
}
void g1_decompress2(const uint8_t* in0, void* out0, int* status0, int n0, const uint8_t* in1, void* out1, int* status1, int n1,
                    const Fp12w& beta, hipStream_t st) {
    Fp b;
    for (int i = 0; i < 12; i++) b.v[i] = beta.v[i];
    if (n0 + n1 <= coop_points_max())
        k_g1_decompress_coop<<<(n0 + n1 + 15) / 16, 64, 0, st>>>(in0, (G1Affine*)out0, status0, n0, in1, (G1Affine*)out1, status1, n1, fq_from_fp(b));
    else k_g1_decompress<<<(n0 + n1 + 63) / 64, 64, 0, st>>>(in0, (G1Affine*)out0, status0, n0, in1, (G1Affine*)out1, status1, n1, 1, fq_from_fp(b));
}
// the two halves of g1_decompress2 as separate launches (decode + on-curve; subgroup)
void g1_decode2(const uint8_t* in0, void* out0, int* status0, int n0, const uint8_t* in1, void* out1, int* status1, int n1,
                const Fp12w& beta, hipStream_t st) {
    Fp b;
    for (int i = 0; i < 12; i++) b.v[i] = beta.v[i];
    k_g1_decompress<<<(n0 + n1 + 63) / 64, 64, 0, st>>>(in0, (G1Affine*)out0, status0, n0, in1, (G1Affine*)out1, status1, n1, 0, fq_from_fp(b));
}
void g1_subgroup2(const void* pts0, int* status0, int n0, const void* pts1, int* status1, int n1, const Fp12w& beta, hipStream_t st) {
    Fp b;
    for (int i = 0; i < 12; i++) b.v[i] = beta.v[i];
    if (n0 + n1 <= coop_points_max())
        k_g1_subgroup_coop<<<(n0 + n1 + 15) / 16, 64, 0, st>>>((const G1Affine*)pts0, status0, n0, (const G1Affine*)pts1, status1, n1, fq_from_fp(b));
    else k_g1_subgroup<<<(n0 + n1 + 63) / 64, 64, 0, st>>>((const G1Affine*)pts0, status0, n0, (const G1Affine*)pts1, status1, n1, fq_from_fp(b));
}
void fk20_srs_vectors(const void* srs, void* X, hipStream_t st) { k_fk20_srs_vectors<<<128 * 64 / 256, 256, 0, st>>>((const G1Affine*)srs, (JacQ*)X); }
void fk20_gather_bases(const void* X, void* bases, hipStream_t st) { k_fk20_gather_bases<<<128 * 64 / 256, 256, 0, st>>>((const JacQ*)X, (G1Affine*)bases); }
#ifdef KZG_TEST_HOOKS
void test_load_points(const uint8_t* in, void* X, int n_lanes, int stride, hipStream_t st) {
    k_test_load_points<<<(128 * n_lanes + 255) / 256, 256, 0, st>>>(in, (JacQ*)X, n_lanes, stride);
}
void test_recompress(const void* pts, uint8_t* out, int n, hipStream_t st) { k_test_recompress<<<(n + 63) / 64, 64, 0, st>>>((const G1Affine*)pts, out, n); }
#endif
}  // namespace launch
}  // namespace kzg
