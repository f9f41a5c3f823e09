// Kernels of verify_cell_kzg_proof_batch and recover_cells_and_kzg_proofs.
// Reference: FK20Verifier::verify_multi_opening (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:129-260,
// compute_sum_interpolation_poly :348-384), g1_lincomb (crates/cryptography/bls12_381/src/lincomb.rs:7-30),
// ReedSolomon::recover_polynomial_coefficient (crates/cryptography/erasure_codes/src/reed_solomon.rs:332-384),
// recover_evaluations_in_domain_order (fk20/cosets.rs:141-198), deserialize_cells (serialization/src/lib.rs:107-114).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "g1_subgroup.hpp"
#include "g1_coop.hpp"
#include "launch.hpp"
#include "glv.hpp"

namespace kzg {

// LDS-resident 4096-point NTT helpers (same network as k_ntt.hip)
__device__ __forceinline__ Fr vl_load(const uint32_t* s, int idx) {
    Fr r;
#pragma unroll
    for (int l = 0; l < 8; l++) r.v[l] = s[l * N_BLOB + idx];
    return r;
}
__device__ __forceinline__ void vl_store(uint32_t* s, int idx, const Fr& a) {
#pragma unroll
    for (int l = 0; l < 8; l++) s[l * N_BLOB + idx] = a.v[l];
}
__device__ __forceinline__ void v_dit_inverse4096(uint32_t* s, const Fr* __restrict__ w8192) {
    for (int half = 1; half < N_BLOB; half <<= 1) {
        const int tw_step = N_EXT / (2 * half);
        for (int q = threadIdx.x; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = vl_load(s, i0), b = vl_load(s, i1);
            Fr t = j ? mul(b, w8192[(N_EXT - j * tw_step) & (N_EXT - 1)]) : b;
            vl_store(s, i0, add(a, t));
            vl_store(s, i1, sub(a, t));
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void v_dif_forward4096(uint32_t* s, const Fr* __restrict__ w8192) {
    for (int half = N_BLOB / 2; half >= 1; half >>= 1) {
        const int tw_step = N_EXT / (2 * half);
        for (int q = threadIdx.x; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = vl_load(s, i0), b = vl_load(s, i1);
            Fr d = sub(a, b);
            vl_store(s, i0, add(a, b));
            vl_store(s, i1, j ? mul(d, w8192[j * tw_step]) : d);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// cells (n x 2048 B, big-endian) -> Fr Montgomery [slot][64]; status |= 1 on a non-canonical element.
// slot_of == nullptr: slot = cell number (verify).  Otherwise the cell lands at slot_of[k] (recover scatter,
// cosets.rs:170-175); evals must be zero-filled first.
// src_of != nullptr: list entry k reads cell src_of[k] of the caller's buffer (flat device layout) instead of cell k.
__global__ void k_cells_to_fr(const uint8_t* __restrict__ cells, Fr* __restrict__ evals, const int* __restrict__ slot_of,
                              int* __restrict__ status, const int* __restrict__ status_of, const int* __restrict__ src_of, int n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * CELL_LEN) return;
    int k = idx >> 6, e = idx & 63;
    Fr x = load_fr_be(cells + ((size_t)(src_of ? src_of[k] : k) * CELL_LEN + e) * 32);
    if (geq_mod<FrParams>(x.v)) atomicOr(status + (status_of ? status_of[k] : 0), 1);
    int slot = slot_of ? slot_of[k] : k;
    evals[(size_t)slot * CELL_LEN + e] = to_mont(x);
}

// r^k for k < n from the table r^(2^i) (compute_powers, verifier.rs:333-343), plus the scalars of the
// two proof MSMs: s1[k] = r^k, s2[k] = r^k * h_{idx_k}^64 (verifier.rs:186-201), both out of Montgomery form.
struct PowTable { Fr p[24]; };
// k0: global position of this shard's first cell (0 for an unsharded batch).
__global__ void k_verify_scalars(PowTable tab, int k0, const int* __restrict__ cell_idx, const Fr* __restrict__ w8192,
                                 Fr* __restrict__ rp_mont, Fr* __restrict__ s1, Fr* __restrict__ s2, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr acc = one<FrParams>();
    const int e = k0 + k;
    for (int i = 0; i < 24; i++)
        if ((e >> i) & 1) acc = mul(acc, tab.p[i]);
    rp_mont[k] = acc;
    s1[k] = from_mont(acc);
    // coset generator h_c = omega_8192^brp7(c) (cosets.rs:89-112); h_c^64 = omega_128^brp7(c) = w8192[64 * brp7(c)]
    int bc = (int)(__brev((unsigned)cell_idx[k]) >> 25);
    s2[k] = from_mont(mul(acc, w8192[64 * bc]));
}
// weights[row] = sum_{k : row_k == row} r^k  (verifier.rs:216-219), canonical.  One block per row: 256 threads scan the
// cell list in strides and fold their partial sums through LDS (the one-thread-per-row loop this replaces took 2.5 ms
// of a 16 ms verification).
__global__ __launch_bounds__(256) void k_verify_weights(const Fr* __restrict__ rp_mont, const int* __restrict__ row,
                                                        Fr* __restrict__ weights, int n, int m) {
    __shared__ uint32_t part[8][256];
    const int r = blockIdx.x, t = threadIdx.x;
    Fr acc = zero<FrParams>();
    for (int k = t; k < n; k += 256)
        if (row[k] == r) acc = add(acc, rp_mont[k]);
#pragma unroll
    for (int l = 0; l < 8; l++) part[l][t] = acc.v[l];
    __syncthreads();
    for (int span = 128; span >= 1; span >>= 1) {
        if (t < span) {
            Fr a, b;
#pragma unroll
            for (int l = 0; l < 8; l++) { a.v[l] = part[l][t]; b.v[l] = part[l][t + span]; }
            a = add(a, b);
#pragma unroll
            for (int l = 0; l < 8; l++) part[l][t] = a.v[l];
        }
        __syncthreads();
    }
    if (t == 0) {
        Fr a;
#pragma unroll
        for (int l = 0; l < 8; l++) a.v[l] = part[l][0];
        weights[r] = from_mont(a);
    }
}

// compute_sum_interpolation_poly (verifier.rs:348-384): for every cell k, interpolate its 64 evaluations on the
// coset h_c <omega_64> (bit-reversed inside the cell, so the DIT network takes the cell as stored), un-shift by
// h_c^-i, scale by r^k and accumulate.  One wave per cell; partial[block][64] is folded by k_interp_fold.
// interp_cell: coefficient i of cell k's interpolation polynomial (lane i of the wave), without the factor r^k.
__device__ __forceinline__ Fr interp_cell(uint32_t (*s)[64], const Fr* __restrict__ evals, const int* __restrict__ cell_idx,
                                          const Fr* __restrict__ w8192, const Fr& inv64, int k, int i) {
    Fr x = evals[(size_t)k * CELL_LEN + i];
#pragma unroll
    for (int l = 0; l < 8; l++) s[l][i] = x.v[l];
    __syncthreads();
    for (int half = 1; half < 64; half <<= 1) {  // DIT, inverse twiddles omega_64^-j = w8192[8192 - 128 j * (32/half)]
        Fr a, b;
        int q = i & 31;
        int j = q & (half - 1);
        int i0 = ((q - j) << 1) + j, i1 = i0 + half;
        if (i < 32) {
#pragma unroll
            for (int l = 0; l < 8; l++) { a.v[l] = s[l][i0]; b.v[l] = s[l][i1]; }
            int e = (N_EXT - j * (N_EXT / (2 * half))) & (N_EXT - 1);
            Fr t = j ? mul(b, w8192[e]) : b;
            Fr u = add(a, t), v = sub(a, t);
#pragma unroll
            for (int l = 0; l < 8; l++) { s[l][i0] = u.v[l]; s[l][i1] = v.v[l]; }
        }
        __syncthreads();
    }
    Fr c;
#pragma unroll
    for (int l = 0; l < 8; l++) c.v[l] = s[l][i];
    __syncthreads();
    int bc = (int)(__brev((unsigned)cell_idx[k]) >> 25);
    Fr hinv_i = w8192[(N_EXT - ((bc * i) & (N_EXT - 1))) & (N_EXT - 1)];  // (h_c^-1)^i, domain.rs:214-223
    return mul(mul(c, inv64), hinv_i);
}
__global__ __launch_bounds__(64) void k_interp(const Fr* __restrict__ evals, const int* __restrict__ cell_idx,
                                               const Fr* __restrict__ rp_mont, const Fr* __restrict__ w8192, Fr inv64,
                                               Fr* __restrict__ partial, int n) {
    __shared__ uint32_t s[8][64];
    const int i = threadIdx.x;
    Fr acc = zero<FrParams>();
    for (int k = blockIdx.x; k < n; k += gridDim.x) acc = add(acc, mul(interp_cell(s, evals, cell_idx, w8192, inv64, k, i), rp_mont[k]));
    partial[(size_t)blockIdx.x * 64 + i] = acc;
}
// The same in two steps for large batches: the per-cell polynomials do not depend on the challenge, so they are computed
// while the host still hashes the transcript (k_interp_cells), and only the weighted sum waits for r (k_interp_sum).
__global__ __launch_bounds__(64) void k_interp_cells(const Fr* __restrict__ evals, const int* __restrict__ cell_idx,
                                                     const Fr* __restrict__ w8192, Fr inv64, Fr* __restrict__ coef, int n) {
    __shared__ uint32_t s[8][64];
    const int i = threadIdx.x;
    for (int k = blockIdx.x; k < n; k += gridDim.x) coef[(size_t)k * 64 + i] = interp_cell(s, evals, cell_idx, w8192, inv64, k, i);
}
// four waves per block, each a quarter of the block's cells; the block's sums meet in LDS
__device__ __forceinline__ Fr fold4(uint32_t (*s)[8][64], const Fr& mine, int q, int i) {
#pragma unroll
    for (int l = 0; l < 8; l++) s[q][l][i] = mine.v[l];
    __syncthreads();
    Fr acc = mine;
    if (q == 0)
        for (int o = 1; o < 4; o++) {
            Fr b;
#pragma unroll
            for (int l = 0; l < 8; l++) b.v[l] = s[o][l][i];
            acc = add(acc, b);
        }
    return acc;
}
__global__ __launch_bounds__(256) void k_interp_sum(const Fr* __restrict__ coef, const Fr* __restrict__ rp_mont,
                                                    Fr* __restrict__ partial, int n) {
    __shared__ uint32_t s[4][8][64];
    const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
    Fr acc = zero<FrParams>();
    for (int k = blockIdx.x * 4 + q; k < n; k += gridDim.x * 4) acc = add(acc, mul(coef[(size_t)k * 64 + i], rp_mont[k]));
    acc = fold4(s, acc, q, i);
    if (q == 0) partial[(size_t)blockIdx.x * 64 + i] = acc;
}
__global__ __launch_bounds__(256) void k_interp_fold(const Fr* __restrict__ partial, int nblocks, Fr* __restrict__ out_neg_canon) {
    __shared__ uint32_t s[4][8][64];
    const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
    Fr acc = zero<FrParams>();
    for (int b = q; b < nblocks; b += 4) acc = add(acc, partial[(size_t)b * 64 + i]);
    acc = fold4(s, acc, q, i);
    if (q == 0) out_neg_canon[i] = from_mont(neg(acc));  // the interpolation commitment enters the pairing input with a minus sign
}

// ------------------------------------------------------------------------------------------------
// Variable-base lincombs of the verifier (g1_lincomb -> blst Pippenger, lincomb.rs:7-30; call sites
// verifier.rs:186,200,224,235) as a bucket MSM on the GPU, two jobs at once (blockIdx.y):
//   job 0: sum_k r^k pi_k                                   (n points)
//   job 1: sum_k r^k h_k^64 pi_k + sum_row w_row C_row - sum_i I_i [tau^i]_1   (n + m + 64 points, one array)
// Every scalar is first split with the GLV endomorphism, k = k1 + k2 lambda (both < 2^128), and the job runs over the 2n
// pairs (P, k1), (phi P, k2): the bucket work is the same (one addition per point and window) but there are 16 windows
// instead of 32, and the final fold needs 120 dependent doublings instead of 248 (it was 2.1 of the 4.1 ms GPU time).
// Unsigned 8-bit windows, 255 buckets per window; arithmetic in the unsaturated field.
//   k_pip_to_affq  : points and their phi images (beta x, y) in the unsaturated form
//   k_pip_glv_split: k -> (k1, k2) by one multiplication with floor(2^256 / lambda) and one correction
//   k_pip_sort     : per (window, job): LDS histogram of the digits, exclusive scan, bucket-ordered index list
//   k_pip_buckets  : thread = (window, bucket): sum of its points (mixed additions)
//   k_pip_window   : per (window, job): S_w = sum_b b * B_b by a suffix scan + tree fold in LDS
//   k_pip_final    : sum_w 2^(8w) S_w (lane w doubles 8w times, LDS fold), converted to affine Montgomery-384
constexpr int PIP_C = 8, PIP_W = 16, PIP_B = 256;
struct PipJob {
    const Fr* scalars;  // canonical
    int n;
};
struct Half128 { uint32_t v[4]; };
// out[i] = P_i, out[n_max + i] = phi(P_i) = (beta x, y)  (the identity (0,0) maps to itself)
__global__ void k_pip_to_affq(const G1Affine* __restrict__ in, AffQ* __restrict__ out, int n, int n_max, Fq<1> beta) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffQ a = affq_from_affine(in[i]);
    out[i] = a;
    AffQ b = a;
    b.x = canonical(mul(a.x, beta));
    out[n_max + i] = b;
}
// halves[job][p] for p < n: k1 of scalar p; for n <= p < 2n: k2 of scalar p - n.   k = k1 + k2 * lambda exactly.
__global__ void k_pip_glv_split(PipJob j0, PipJob j1, int n_max, Half128* __restrict__ halves) {
    const int job = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const PipJob jb = job ? j1 : j0;
    if (i >= jb.n) return;
    const Fr k = jb.scalars[i];
    uint32_t r[4], q[4];
    glv_split_unsigned(k, r, q);  // glv.hpp: r = k mod lambda, q = floor(k / lambda)
    Half128 h1, h2;
    for (int l = 0; l < 4; l++) { h1.v[l] = r[l]; h2.v[l] = q[l]; }
    Half128* out = halves + (size_t)job * 2 * n_max;
    out[i] = h1;
    out[jb.n + i] = h2;
}
__device__ __forceinline__ int pip_digit(const Half128& s, int w) { return (s.v[w >> 2] >> ((w & 3) * 8)) & 255; }
// idx: [job][window][2 n_max] (resolved indices into the point array [P | phi P]); start: [job][window][257]
__global__ __launch_bounds__(256) void k_pip_sort(PipJob j0, PipJob j1, int n_max, const Half128* __restrict__ halves,
                                                  int* __restrict__ idx, int* __restrict__ start) {
    __shared__ int hist[PIP_B], cursor[PIP_B];
    const int w = blockIdx.x, job = blockIdx.y, t = threadIdx.x;
    const int n = (job ? j1 : j0).n;
    const Half128* hv = halves + (size_t)job * 2 * n_max;
    hist[t] = 0;
    __syncthreads();
    for (int k = t; k < 2 * n; k += 256) atomicAdd(&hist[pip_digit(hv[k], w)], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int b = 0; b < PIP_B; b++) { cursor[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    int* st = start + ((size_t)job * PIP_W + w) * (PIP_B + 1);
    st[t] = cursor[t];
    if (t == 0) st[PIP_B] = 2 * n;
    __syncthreads();
    int* out = idx + ((size_t)job * PIP_W + w) * 2 * n_max;
    for (int k = t; k < 2 * n; k += 256) {
        int d = pip_digit(hv[k], w);
        out[atomicAdd(&cursor[d], 1)] = k < n ? k : n_max + (k - n);
    }
}
// buckets: [job][window][256] JacQ
__global__ __launch_bounds__(64) void k_pip_buckets(const AffQ* __restrict__ pts, const int* __restrict__ idx,
                                                    const int* __restrict__ start, int n_max, JacQ* __restrict__ buckets) {
    const int job = blockIdx.y;
    const int g = blockIdx.x * 64 + threadIdx.x;  // window * 256 + bucket
    if (g >= PIP_W * PIP_B) return;
    const int w = g >> 8, b = g & 255;
    JacQ acc = jacq_inf();
    if (b) {
        const int* st = start + ((size_t)job * PIP_W + w) * (PIP_B + 1);
        const int* ix = idx + ((size_t)job * PIP_W + w) * 2 * n_max;
        for (int p = st[b]; p < st[b + 1]; p++) acc = add_mixed(acc, pts[ix[p]]);
    }
    buckets[(size_t)job * PIP_W * PIP_B + g] = acc;
}
__global__ __launch_bounds__(256) void k_pip_window(const JacQ* __restrict__ buckets, JacQ* __restrict__ wsum) {
    __shared__ JacQ T[PIP_B];
    const int w = blockIdx.x, W = gridDim.x, job = blockIdx.y, t = threadIdx.x;
    T[t] = buckets[((size_t)job * W + w) * PIP_B + t];  // bucket 0 is the identity
    __syncthreads();
    // suffix sums T[b] = sum_{j >= b} B_j  (Hillis-Steele), then sum_b T[b] = sum_b b * B_b
    for (int off = 1; off < PIP_B; off <<= 1) {
        JacQ v = T[t];
        bool has = t + off < PIP_B;
        JacQ o = has ? T[t + off] : v;
        __syncthreads();
        if (has) T[t] = add(v, o);
        __syncthreads();
    }
    if (t == 0) T[0] = jacq_inf();  // T[0] counted bucket 0's suffix; the weighted sum starts at b = 1
    coop_tree_fold<PIP_B>(T, PIP_B / 2, t);
    if (t == 0) wsum[job * W + w] = T[0];
}
// (the 16 windows take the wave's 16 quads: four lanes share each of the up to 120 dependent doublings, g1_coop.hpp)
__global__ __launch_bounds__(64) void k_pip_final(const JacQ* __restrict__ wsum, G1Affine* __restrict__ out_affine) {
    __shared__ JacQ T[PIP_W];
    const int job = blockIdx.x, t = threadIdx.x;
    static_assert(PIP_W * 4 == 64, "one quad per window");
    {
        const int w = t >> 2, quad = t & 3;
        JacQ acc = wsum[job * PIP_W + w];
#pragma unroll 1
        for (int k = 0; k < PIP_C * w; k++) acc = coop_dbl(acc, quad);
        if (quad == 0) T[w] = acc;
    }
    __syncthreads();
    for (int span = PIP_W / 2; span >= 1; span >>= 1) {
        if (t < span) T[t] = add(T[t], T[t + span]);
        __syncthreads();
    }
    if (t == 0) out_affine[job] = to_affine(jac_from_jacq(T[0]));
}

// ------------------------------------------------------------------------------------------------
// verify_cell_kzg_proof_batch: the same two lincombs with the points SHIFTED AHEAD OF TIME.  The challenge r -- and with
// it every scalar -- is known only after the host has hashed the whole transcript (17.6 MB, 7 ms, at config 3), while the
// points are there as soon as they are decoded.  So the 120 dependent doublings that the fold of 16 windows needs
// (k_pip_final above: 1.1 ms with one busy lane) move in front of the challenge: k_pip_shift builds, for every point,
// the 16 byte-shifted copies 2^(8p) P and their phi images (32 affine points, one inversion per input point) while the
// host hashes -- and next to the subgroup tests, an equally long chain, which run on a second stream (small batches have
// no hash to hide behind).  A 128-bit half scalar is then 16 independent bytes, each paired with its own copy, and the
// whole lincomb is ONE window: 32 n (point, byte) items into 255 buckets.  After the challenge: split, a three-kernel
// counting sort, two waves per bucket (128 lanes share a bucket's ~1040 items and fold through LDS), the suffix scan of
// k_pip_window; the host normalises the two sums.  No doubling at all: 3.4 -> 0.8 ms of GPU time behind the hash.
//   pts32[(phi * 16 + p) * n_max + i] = phi^phi(2^(8p) P_i)
constexpr int PS_P = 16, PS_SLICES = 64;
// quad >= 0: the four lanes of a quad work on point i together (g1_coop.hpp) -- they share each doubling and repeat the rest, every
// lane writing the same values; quad < 0: one lane per point
__device__ __forceinline__ void pip_shift_point(int i, const G1Affine* __restrict__ in, AffQ* __restrict__ pts32, JacQ* __restrict__ jac,
                                                Fq<2>* __restrict__ pre, int n_max, const Fq<1>& beta, int quad = -1) {
    auto put = [&](int p, const AffQ& a) {
        pts32[(size_t)p * n_max + i] = a;
        AffQ b = a;
        b.x = canonical(mul(a.x, beta));
        pts32[(size_t)(PS_P + p) * n_max + i] = b;
    };
    const AffQ P = affq_from_affine(in[i]);
    put(0, P);
    if (is_inf(P)) {  // the identity (0,0) is its own image and its own multiple
        for (int p = 1; p < PS_P; p++) put(p, P);
        return;
    }
    JacQ cur = to_jacq(P);
    Fq<2> prod = relax<2>(fq_one());
    for (int p = 1; p < PS_P; p++) {
        if (quad >= 0) { for (int k = 0; k < 8; k++) cur = coop_dbl(cur, quad); }
        else { for (int k = 0; k < 8; k++) cur = dbl(cur); }
        jac[(size_t)(p - 1) * n_max + i] = cur;
        pre[(size_t)(p - 1) * n_max + i] = prod;
        prod = mul(prod, cur.z);
    }
    // a point of G1 has odd order, so no multiple by a power of two is the identity; anything else was rejected by the
    // decoder and its (meaningless) copies are never used: inv(0) = 0 keeps even that case finite
    Fq<2> inv = relax<2>(fq_inv(prod));
    for (int p = PS_P - 1; p >= 1; p--) {
        const JacQ q = jac[(size_t)(p - 1) * n_max + i];
        const Fq<2> zi = mul(inv, pre[(size_t)(p - 1) * n_max + i]);
        inv = mul(inv, q.z);
        const Fq<2> zi2 = sqr(zi);
        AffQ a;
        a.x = reduce_once(mul(q.x, zi2));
        a.y = reduce_once(mul(q.y, mul(zi2, zi)));
        put(p, a);
    }
}
__global__ __launch_bounds__(64) void k_pip_shift(const G1Affine* __restrict__ in, AffQ* __restrict__ pts32, JacQ* __restrict__ jac,
                                                  Fq<2>* __restrict__ pre, int n, int n_max, Fq<1> beta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pip_shift_point(i, in, pts32, jac, pre, n_max, beta);
}
// The two challenge-free chains of a verification -- the byte-shifted copies (120 dependent doublings per point) and the
// subgroup tests of the decoded proofs and commitments (126 dependent doublings per point) -- in ONE launch: the first
// shift_blocks blocks shift, the rest test.  They used to be two launches on two streams so as to run side by side; HIP maps a
// process's streams onto four hardware queues, and with several verifications in flight on several engine lanes one lane's
// side stream landed on another lane's queue and the lanes took turns (rocprofv3 trace of two callers: 5.0 ms per round of
// two instead of 3.3).  One launch needs one stream per verification, and the waves of the two halves still run side by side.
// status: 0 -> 0 or 2; other values are kept (k_g1misc.hip: k_g1_subgroup does the same on its own).
__global__ __launch_bounds__(64) void k_pip_shift_subgroup(const G1Affine* __restrict__ in, AffQ* __restrict__ pts32, JacQ* __restrict__ jac,
                                                           Fq<2>* __restrict__ pre, int n, int n_max, int shift_blocks,
                                                           const G1Affine* __restrict__ pts0, int* __restrict__ status0, int n0,
                                                           const G1Affine* __restrict__ pts1, int* __restrict__ status1, int n1, Fq<1> beta) {
    if ((int)blockIdx.x < shift_blocks) {
        const int i = blockIdx.x * 64 + threadIdx.x;
        pip_shift_point(i < n ? i : n - 1, in, pts32, jac, pre, n_max, beta);  // (lanes behind the last point repeat it: see the quad form below)
        return;
    }
    int i = ((int)blockIdx.x - shift_blocks) * 64 + threadIdx.x;
    if (i >= n0 + n1) i = n0 + n1 - 1;
    const bool second = i >= n0;
    if (second) i -= n0;
    int* st = second ? status1 : status0;
    if (st[i] != 0) return;
    const G1Affine a = (second ? pts1 : pts0)[i];
    if (is_inf(a)) return;
    if (!g1_in_subgroup_q(affq_from_affine(a), beta)) st[i] = 2;
}
// The same launch for a verification of a few hundred points, where both chains are pure latency (every wave has a SIMD to
// itself): FOUR lanes per point share the doublings (g1_coop.hpp).  16 points per block.
__global__ __launch_bounds__(64) void k_pip_shift_subgroup_coop(const G1Affine* __restrict__ in, AffQ* __restrict__ pts32, JacQ* __restrict__ jac,
                                                                Fq<2>* __restrict__ pre, int n, int n_max, int shift_blocks,
                                                                const G1Affine* __restrict__ pts0, int* __restrict__ status0, int n0,
                                                                const G1Affine* __restrict__ pts1, int* __restrict__ status1, int n1, Fq<1> beta) {
    const int quad = threadIdx.x & 3;
    if ((int)blockIdx.x < shift_blocks) {
        // (the quads behind the last point repeat its work -- same values to the same places: a wave with few lanes in use is the
        // slow one, k_g1slp.hip, and here it would be the launch's longest)
        const int i = blockIdx.x * 16 + (threadIdx.x >> 2);
        pip_shift_point(i < n ? i : n - 1, in, pts32, jac, pre, n_max, beta, quad);
        return;
    }
    int i = ((int)blockIdx.x - shift_blocks) * 16 + (threadIdx.x >> 2);
    if (i >= n0 + n1) i = n0 + n1 - 1;
    const bool second = i >= n0;
    if (second) i -= n0;
    int* st = second ? status1 : status0;
    if (st[i] != 0) return;
    const G1Affine a = (second ? pts1 : pts0)[i];
    if (is_inf(a)) return;
    const bool ok = g1_in_subgroup_coop(affq_from_affine(a), beta, quad);
    if (!ok && quad == 0) st[i] = 2;
}
// counting sort of the 32 n items of a job by their byte.  Half-scalar entry e < 2n (k1 of scalar e, or k2 of scalar e - n)
// carries 16 items; slice s of PS_SLICES owns a contiguous range of entries.
__device__ __forceinline__ int ps_byte(const Half128& v, int p) { return (v.v[p >> 2] >> ((p & 3) * 8)) & 255; }
__global__ __launch_bounds__(256) void k_ps_hist(PipJob j0, PipJob j1, int n_max, const Half128* __restrict__ halves,
                                                 int* __restrict__ hist /*[2][PS_SLICES][256]*/) {
    __shared__ int h[PIP_B];
    const int slice = blockIdx.x, job = blockIdx.y, t = threadIdx.x;
    const int E = 2 * (job ? j1 : j0).n;
    const Half128* hv = halves + (size_t)job * 2 * n_max;
    h[t] = 0;
    __syncthreads();
    const int lo = (int)((long)E * slice / PS_SLICES), hi = (int)((long)E * (slice + 1) / PS_SLICES);
    for (int e = lo + t; e < hi; e += 256) {
        const Half128 v = hv[e];
#pragma unroll
        for (int p = 0; p < PS_P; p++) atomicAdd(&h[ps_byte(v, p)], 1);
    }
    __syncthreads();
    hist[((size_t)job * PS_SLICES + slice) * PIP_B + t] = h[t];
}
// hist[job][slice][b] <- first position of slice's items of bucket b in the sorted list; start[job][b] = first position of bucket b
__global__ __launch_bounds__(256) void k_ps_scan(int* __restrict__ hist, int* __restrict__ start /*[2][257]*/) {
    __shared__ int tot[PIP_B];
    const int job = blockIdx.x, b = threadIdx.x;
    int* h = hist + (size_t)job * PS_SLICES * PIP_B;
    int acc = 0;
    for (int s = 0; s < PS_SLICES; s++) {
        const int c = h[s * PIP_B + b];
        h[s * PIP_B + b] = acc;
        acc += c;
    }
    tot[b] = acc;
    __syncthreads();
    if (b == 0) {
        int a = 0;
        for (int j = 0; j < PIP_B; j++) { const int c = tot[j]; tot[j] = a; a += c; }
        start[job * (PIP_B + 1) + PIP_B] = a;
    }
    __syncthreads();
    const int base = tot[b];
    start[job * (PIP_B + 1) + b] = base;
    for (int s = 0; s < PS_SLICES; s++) h[s * PIP_B + b] += base;
}
__global__ __launch_bounds__(256) void k_ps_scatter(PipJob j0, PipJob j1, int n_max, const Half128* __restrict__ halves,
                                                    const int* __restrict__ hist, int* __restrict__ items /*[2][32 n_max]*/) {
    __shared__ int cur[PIP_B];
    const int slice = blockIdx.x, job = blockIdx.y, t = threadIdx.x;
    const int n = (job ? j1 : j0).n, E = 2 * n;
    const Half128* hv = halves + (size_t)job * 2 * n_max;
    cur[t] = hist[((size_t)job * PS_SLICES + slice) * PIP_B + t];
    __syncthreads();
    int* out = items + (size_t)job * 2 * PS_P * n_max;
    const int lo = (int)((long)E * slice / PS_SLICES), hi = (int)((long)E * (slice + 1) / PS_SLICES);
    for (int e = lo + t; e < hi; e += 256) {
        const Half128 v = hv[e];
        const int phi = e >= n, i = phi ? e - n : e;
#pragma unroll
        for (int p = 0; p < PS_P; p++) out[atomicAdd(&cur[ps_byte(v, p)], 1)] = (phi * PS_P + p) * n_max + i;
    }
}
// two waves per bucket: lane l adds the bucket's items l, l + 128, ... into an XYZZ sum (the next point is requested one
// addition ahead), then the 128 partial sums fold through LDS.  buckets: [job][256] JacQ
constexpr int PS_LANES = 128;  // 256 buckets x 2 jobs x 2 waves = one wave on every SIMD of the chip
__global__ __launch_bounds__(PS_LANES) void k_ps_buckets(const AffQ* __restrict__ pts32, const int* __restrict__ items,
                                                         const int* __restrict__ start, int n_max, JacQ* __restrict__ buckets) {
    __shared__ JacQ red[PS_LANES];
    const int b = blockIdx.x, job = blockIdx.y, l = threadIdx.x;
    XyzzQ acc = xyzz_inf();
    if (b) {
        const int* it = items + (size_t)job * 2 * PS_P * n_max;
        const int end = start[job * (PIP_B + 1) + b + 1];
        int p = start[job * (PIP_B + 1) + b] + l;
        AffQ cur;
        if (p < end) cur = pts32[it[p]];
        for (; p < end; p += PS_LANES) {
            AffQ nxt = cur;
            if (p + PS_LANES < end) nxt = pts32[it[p + PS_LANES]];
            acc = add_mixed(acc, cur);
            cur = nxt;
        }
    }
    red[l] = to_jacq(acc);
    coop_tree_fold<PS_LANES>(red, PS_LANES / 2, l);  // the idle lanes of each level join its additions (g1_coop.hpp)
    if (l == 0) buckets[(size_t)job * PIP_B + b] = red[0];
}
// points[dst_off + i] = src[i] (device gather of the 64 SRS points behind the proofs/commitments)
__global__ void k_copy_affine(const G1Affine* __restrict__ src, G1Affine* __restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Per-cell values of the vanishing polynomial for a batch of R recoveries (one thread per (blob, cell)).
// zp[r][0..deg_r] = coefficients of Z'_r(y) = prod over missing domain-order indices (y - omega_128^i)
// (vanishing_poly, polynomial/src/poly_coeff.rs:109-115, built on the host: <= 64 roots).
//   zeval[r][c] = Z'(omega_128^brp7(c))            (zero exactly on the missing cells)
//   zcinv[r][c] = 1 / Z'(7^64 * omega_128^brp7(c)) (never zero: the coset has no roots of Z, reed_solomon.rs:356-357)
// Z'_r(y) = prod over the missing cells of (y - omega_128^i), i the domain-order index (vanishing_poly,
// polynomial/src/poly_coeff.rs:109-115; construct_vanishing_poly_from_block_erasures, reed_solomon.rs:220-262).
// One wave per blob: lane k holds coefficient k of the monic running product (the leading 1 is implicit), and a factor
// (y - root) is one multiply-add with the neighbour's coefficient: z[k] <- z[k] * (-root) + z[k-1].  <= 64 steps.
// present: 128-bit mask per blob (bit i set = cell with domain index i is present).  zp: [R][65] Montgomery, deg: [R].
__global__ __launch_bounds__(64) void k_rec_vanishing_poly(const uint32_t* __restrict__ present, const Fr* __restrict__ w8192,
                                                           Fr* __restrict__ zp, int* __restrict__ deg, int R) {
    const int r = blockIdx.x, k = threadIdx.x;
    const Fr zero_ = zero<FrParams>(), one_ = one<FrParams>();
    Fr z = zero_;
    int d = 0;
    for (int i = 0; i < N_CELLS; i++) {
        if ((present[(size_t)r * 4 + (i >> 5)] >> (i & 31)) & 1) continue;  // wave-uniform
        const Fr nr = neg(w8192[64 * i]);
        const Fr cur = k < d ? z : (k == d ? one_ : zero_);
        Fr prev;
#pragma unroll
        for (int w = 0; w < 8; w++) prev.v[w] = __shfl_up(cur.v[w], 1);
        if (k == 0) prev = zero_;
        z = add(mul(cur, nr), prev);
        d++;
    }
    zp[(size_t)r * 65 + k] = k < d ? z : (k == d ? one_ : zero_);
    if (k == 0) {
        zp[(size_t)r * 65 + 64] = d == 64 ? one_ : zero_;
        deg[r] = d;
    }
}

__global__ void k_rec_vanishing(const Fr* __restrict__ zp, const int* __restrict__ deg, const Fr* __restrict__ w8192,
                                Fr seven64, Fr* __restrict__ zeval, Fr* __restrict__ zcinv, int R) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * N_CELLS) return;
    int r = idx >> 7, c = idx & 127;
    const Fr* z = zp + (size_t)r * 65;
    int d = deg[r];
    int bc = (int)(__brev((unsigned)c) >> 25);
    Fr x = w8192[64 * bc], xc = mul(seven64, x);
    Fr a = zero<FrParams>(), b = zero<FrParams>();
    for (int k = d; k >= 0; k--) {
        a = add(mul(a, x), z[k]);
        b = add(mul(b, xc), z[k]);
    }
    zeval[idx] = a;
    zcinv[idx] = inv_fast(b);  // binary extended GCD (inverse.hpp): a tenth of the instructions of b^(r-2), which was 0.47 ms of a single-blob recovery
}

// ------------------------------------------------------------------------------------------------
// Recovery (reed_solomon.rs:332-384).  Data lives in "cell order" = bit-reversed domain order
// (cosets.rs:177-179), which is exactly what the DIT network consumes and the DIF network produces,
// so the reference's three reverse_bit_order passes disappear.  Z(x) = Z'(x^64) takes one value per cell:
// zeval[c] = Z'(omega_128^brp7(c)), zcinv[c] = 1 / Z'(7^64 omega_128^brp7(c))  (host-computed, 128 each).
//
// k_rec_dit_half: half h of V (optionally multiplied cell-wise by fac[c]) -> DIT_4096 -> T
// grid = (R, 2), block = 1024, LDS 128 KiB.
__global__ __launch_bounds__(1024) void k_rec_dit_half(const Fr* __restrict__ V, const Fr* __restrict__ fac, Fr* __restrict__ T,
                                                      const Fr* __restrict__ w8192) {
    extern __shared__ uint32_t s[];
    const int r = blockIdx.x, h = blockIdx.y;
    const Fr* src = V + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        Fr x = src[e];
        if (fac) x = mul(x, fac[(size_t)r * N_CELLS + ((h * N_BLOB + e) >> 6)]);
        vl_store(s, e, x);
    }
    __syncthreads();
    v_dit_inverse4096(s, w8192);
    Fr* dst = T + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) dst[e] = vl_load(s, e);
}
// last DIT layer of the 8192-point inverse transform + 1/8192 + coset (un)shift by shift[i]  (domain.rs:129-142,214-223).
// final_pass: also check that coefficients 4096.. are zero (reed_solomon.rs:373-380) and emit only the low half to `coeffs`.
__global__ void k_rec_dit_last(const Fr* __restrict__ T, const Fr* __restrict__ shift, Fr n_inv, Fr* __restrict__ U,
                               Fr* __restrict__ coeffs, int* __restrict__ status, const Fr* __restrict__ w8192, int final_pass) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;  // r * 4096 + i
    int r = idx >> 12, i = idx & (N_BLOB - 1);
    const Fr* t = T + (size_t)r * N_EXT;
    Fr a = t[i], b = t[i + N_BLOB];
    Fr tw = i ? mul(b, w8192[N_EXT - i]) : b;
    Fr lo = mul(mul(add(a, tw), n_inv), shift[i]);
    Fr hi = mul(mul(sub(a, tw), n_inv), shift[i + N_BLOB]);
    if (final_pass) {
        if (!is_zero(hi)) atomicOr(&status[r], 4);
        coeffs[(size_t)r * N_BLOB + i] = lo;
    } else {
        U[(size_t)r * N_EXT + i] = lo;
        U[(size_t)r * N_EXT + i + N_BLOB] = hi;
    }
}
// first DIF layer + DIF_4096 on half h, then cell-wise multiply by fac (the inverse vanishing values): U -> V
__global__ __launch_bounds__(1024) void k_rec_dif_half(const Fr* __restrict__ U, const Fr* __restrict__ fac, Fr* __restrict__ V,
                                                      const Fr* __restrict__ w8192) {
    extern __shared__ uint32_t s[];
    const int r = blockIdx.x, h = blockIdx.y;
    const Fr* src = U + (size_t)r * N_EXT;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        Fr a = src[e], b = src[e + N_BLOB];
        Fr x = h ? sub(a, b) : add(a, b);
        if (h && e) x = mul(x, w8192[e]);
        vl_store(s, e, x);
    }
    __syncthreads();
    v_dif_forward4096(s, w8192);
    Fr* dst = V + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024)
        dst[e] = mul(vl_load(s, e), fac[(size_t)r * N_CELLS + ((h * N_BLOB + e) >> 6)]);
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_verify() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_cells_to_fr));
}
static Fr as_fr2(const Fr8& x) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }
void init_attributes_verify() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_dit_half), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_dif_half), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
}
void cells_to_fr(const uint8_t* cells, void* evals, const int* slot_of, int* status, const int* status_of, const int* src_of, int n,
                 hipStream_t st) {
    k_cells_to_fr<<<(n * CELL_LEN + 255) / 256, 256, 0, st>>>(cells, (Fr*)evals, slot_of, status, status_of, src_of, n);
}
void verify_scalars(const Fr8* pow_table24, int k0, const int* cell_idx, const void* w8192, void* rp_mont, void* s1, void* s2,
                    int n, hipStream_t st) {
    PowTable t;
    for (int i = 0; i < 24; i++) t.p[i] = as_fr2(pow_table24[i]);
    k_verify_scalars<<<(n + 255) / 256, 256, 0, st>>>(t, k0, cell_idx, (const Fr*)w8192, (Fr*)rp_mont, (Fr*)s1, (Fr*)s2, n);
}
void verify_weights(const void* rp_mont, const int* row, void* weights, int n, int m, hipStream_t st) {
    if (m > 0) k_verify_weights<<<m, 256, 0, st>>>((const Fr*)rp_mont, row, (Fr*)weights, n, m);
}
void interp(const void* evals, const int* cell_idx, const void* rp_mont, const void* w8192, const Fr8& inv64, void* partial,
            int nblocks, void* out_neg_canon, int n, hipStream_t st) {
    k_interp<<<nblocks, 64, 0, st>>>((const Fr*)evals, cell_idx, (const Fr*)rp_mont, (const Fr*)w8192, as_fr2(inv64), (Fr*)partial, n);
    k_interp_fold<<<1, 256, 0, st>>>((const Fr*)partial, nblocks, (Fr*)out_neg_canon);
}
void interp_cells(const void* evals, const int* cell_idx, const void* w8192, const Fr8& inv64, void* coef, int n, hipStream_t st) {
    k_interp_cells<<<n < 1024 ? n : 1024, 64, 0, st>>>((const Fr*)evals, cell_idx, (const Fr*)w8192, as_fr2(inv64), (Fr*)coef, n);
}
void interp_sum(const void* coef, const void* rp_mont, void* partial, int nblocks, void* out_neg_canon, int n, hipStream_t st) {
    k_interp_sum<<<nblocks, 256, 0, st>>>((const Fr*)coef, (const Fr*)rp_mont, (Fr*)partial, n);
    k_interp_fold<<<1, 256, 0, st>>>((const Fr*)partial, nblocks, (Fr*)out_neg_canon);
}
size_t pip_workspace_bytes(int n_max) {
    return (size_t)2 * n_max * SIZEOF_AFFQ + (size_t)2 * 2 * n_max * sizeof(Half128) + (size_t)2 * PIP_W * 2 * n_max * sizeof(int) +
           (size_t)2 * PIP_W * (PIP_B + 1) * sizeof(int) + (size_t)2 * PIP_W * PIP_B * SIZEOF_JACQ + (size_t)2 * PIP_W * SIZEOF_JACQ + 512;
}
void copy_affine(const void* src, void* dst, int n, hipStream_t st) {
    k_copy_affine<<<(n + 63) / 64, 64, 0, st>>>((const G1Affine*)src, (G1Affine*)dst, n);
}
// two bucket MSMs over a shared point array: job 0 uses points[0..n0) with sc0, job 1 points[0..n1) with sc1
void msm_pippenger2(const void* points, const void* sc0, int n0, const void* sc1, int n1, void* workspace, void* out_affine2,
                    const Fp12w& beta, hipStream_t st) {
    const int n_max = n1 > n0 ? n1 : n0;
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const Fq<1> bt = fq_from_fp(b384);
    char* p = (char*)workspace;
    AffQ* pts = (AffQ*)p; p += (size_t)2 * n_max * SIZEOF_AFFQ;
    Half128* halves = (Half128*)p; p += (size_t)2 * 2 * n_max * sizeof(Half128);
    int* idx = (int*)p; p += (size_t)2 * PIP_W * 2 * n_max * sizeof(int);
    int* start = (int*)p; p += (size_t)2 * PIP_W * (PIP_B + 1) * sizeof(int);
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    JacQ* buckets = (JacQ*)p; p += (size_t)2 * PIP_W * PIP_B * SIZEOF_JACQ;
    JacQ* wsum = (JacQ*)p;
    PipJob j0{(const Fr*)sc0, n0}, j1{(const Fr*)sc1, n1};
    k_pip_to_affq<<<(n_max + 63) / 64, 64, 0, st>>>((const G1Affine*)points, pts, n_max, n_max, bt);
    k_pip_glv_split<<<dim3((n_max + 63) / 64, 2), 64, 0, st>>>(j0, j1, n_max, halves);
    k_pip_sort<<<dim3(PIP_W, 2), 256, 0, st>>>(j0, j1, n_max, halves, idx, start);
    k_pip_buckets<<<dim3(PIP_W * PIP_B / 64, 2), 64, 0, st>>>(pts, idx, start, n_max, buckets);
    k_pip_window<<<dim3(PIP_W, 2), 256, 0, st>>>(buckets, wsum);
    k_pip_final<<<2, 64, 0, st>>>(wsum, (G1Affine*)out_affine2);
}
// byte-shifted form: workspace = [pts32 | jac | pre | halves | items | hist | start | buckets | wsum]
struct PsLayout {
    AffQ* pts32; JacQ* jac; Fq<2>* pre; Half128* halves; int* items; int* hist; int* start; JacQ* buckets; JacQ* wsum;
    size_t bytes;
};
static PsLayout ps_layout(void* workspace, int n_max) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const uintptr_t p = (uintptr_t)workspace;  // integer arithmetic: the size query passes no buffer
    PsLayout L;
    size_t o = 0;
    L.pts32 = (AffQ*)(p + o); o += up((size_t)2 * PS_P * n_max * SIZEOF_AFFQ);
    L.jac = (JacQ*)(p + o); o += up((size_t)(PS_P - 1) * n_max * SIZEOF_JACQ);
    L.pre = (Fq<2>*)(p + o); o += up((size_t)(PS_P - 1) * n_max * sizeof(Fq<2>));
    L.halves = (Half128*)(p + o); o += up((size_t)2 * 2 * n_max * sizeof(Half128));
    L.items = (int*)(p + o); o += up((size_t)2 * 2 * PS_P * n_max * sizeof(int));
    L.hist = (int*)(p + o); o += up((size_t)2 * PS_SLICES * PIP_B * sizeof(int));
    L.start = (int*)(p + o); o += up((size_t)2 * (PIP_B + 1) * sizeof(int));
    L.buckets = (JacQ*)(p + o); o += up((size_t)2 * PIP_B * SIZEOF_JACQ);
    L.wsum = (JacQ*)(p + o); o += up((size_t)2 * SIZEOF_JACQ);
    L.bytes = o;
    return L;
}
size_t pip_shift_workspace_bytes(int n_max) { return ps_layout(nullptr, n_max).bytes; }
void pip_shift_prepare(const void* points, int n_pts, int n_max, void* workspace, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const PsLayout L = ps_layout(workspace, n_max);
    k_pip_shift<<<(n_pts + 63) / 64, 64, 0, st>>>((const G1Affine*)points, L.pts32, L.jac, L.pre, n_pts, n_max, fq_from_fp(b384));
}
void pip_shift_prepare_and_subgroup(const void* points, int n_pts, int n_max, void* workspace, const void* pts0, int* status0, int n0,
                                    const void* pts1, int* status1, int n1, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const PsLayout L = ps_layout(workspace, n_max);
    // up to a few thousand points every wave of the quad form still has a SIMD to itself (4 lanes per point: 16 points per wave,
    // 1,024 SIMDs); beyond that the chip fills and one lane per point is less work (launch::coop_points_max)
    if (n_pts + n0 + n1 <= coop_points_max()) {
        const int sb = (n_pts + 15) / 16, tb = (n0 + n1 + 15) / 16;
        k_pip_shift_subgroup_coop<<<sb + tb, 64, 0, st>>>((const G1Affine*)points, L.pts32, L.jac, L.pre, n_pts, n_max, sb, (const G1Affine*)pts0,
                                                          status0, n0, (const G1Affine*)pts1, status1, n1, fq_from_fp(b384));
        return;
    }
    const int shift_blocks = (n_pts + 63) / 64, sub_blocks = (n0 + n1 + 63) / 64;
    k_pip_shift_subgroup<<<shift_blocks + sub_blocks, 64, 0, st>>>((const G1Affine*)points, L.pts32, L.jac, L.pre, n_pts, n_max, shift_blocks,
                                                                   (const G1Affine*)pts0, status0, n0, (const G1Affine*)pts1, status1, n1, fq_from_fp(b384));
}
void msm_pippenger2_shifted(const void* sc0, int n0, const void* sc1, int n1, int n_max, void* workspace, void* out_jacq2,
                            hipStream_t st) {
    const PsLayout L = ps_layout(workspace, n_max);
    PipJob j0{(const Fr*)sc0, n0}, j1{(const Fr*)sc1, n1};
    k_pip_glv_split<<<dim3((n_max + 63) / 64, 2), 64, 0, st>>>(j0, j1, n_max, L.halves);
    k_ps_hist<<<dim3(PS_SLICES, 2), 256, 0, st>>>(j0, j1, n_max, L.halves, L.hist);
    k_ps_scan<<<2, 256, 0, st>>>(L.hist, L.start);
    k_ps_scatter<<<dim3(PS_SLICES, 2), 256, 0, st>>>(j0, j1, n_max, L.halves, L.hist, L.items);
    k_ps_buckets<<<dim3(PIP_B, 2), PS_LANES, 0, st>>>(L.pts32, L.items, L.start, n_max, L.buckets);
    // the two sums leave in Jacobian form: the caller's host thread normalises them (one 4 us inversion each) instead of a
    // single GPU lane (0.17 ms)
    k_pip_window<<<dim3(1, 2), 256, 0, st>>>(L.buckets, (JacQ*)out_jacq2);
}
void rec_vanishing_poly(const uint32_t* present, const void* w8192, void* zp, int* deg, int R, hipStream_t st) {
    k_rec_vanishing_poly<<<R, 64, 0, st>>>(present, (const Fr*)w8192, (Fr*)zp, deg, R);
}
void rec_vanishing(const void* zp, const int* deg, const void* w8192, const Fr8& seven64, void* zeval, void* zcinv, int R,
                   hipStream_t st) {
    k_rec_vanishing<<<(R * N_CELLS + 127) / 128, 128, 0, st>>>((const Fr*)zp, deg, (const Fr*)w8192, as_fr2(seven64), (Fr*)zeval,
                                                              (Fr*)zcinv, R);
}
void rec_dit_half(int R, const void* V, const void* fac, void* T, const void* w8192, hipStream_t st) {
    k_rec_dit_half<<<dim3(R, 2), 1024, LDS_NTT, st>>>((const Fr*)V, (const Fr*)fac, (Fr*)T, (const Fr*)w8192);
}
void rec_dit_last(int R, const void* T, const void* shift, const Fr8& n_inv, void* U, void* coeffs, int* status,
                  const void* w8192, int final_pass, hipStream_t st) {
    k_rec_dit_last<<<R * N_BLOB / 256, 256, 0, st>>>((const Fr*)T, (const Fr*)shift, as_fr2(n_inv), (Fr*)U, (Fr*)coeffs, status,
                                                    (const Fr*)w8192, final_pass);
}
void rec_dif_half(int R, const void* U, const void* fac, void* V, const void* w8192, hipStream_t st) {
    k_rec_dif_half<<<dim3(R, 2), 1024, LDS_NTT, st>>>((const Fr*)U, (const Fr*)fac, (Fr*)V, (const Fr*)w8192);
}
}  // namespace launch
}  // namespace kzg
