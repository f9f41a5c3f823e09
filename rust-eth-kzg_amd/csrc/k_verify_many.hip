// Kernels of eth_kzg_amd_verify_cell_kzg_proof_batch_many: MANY independent verify_cell_kzg_proof_batch problems in one pass
// (the reference verifies concurrently from many threads on one context, bindings/node/src/lib.rs:92-299; here the
// problems ride side by side on the GPU).  Per problem b the equation of FK20Verifier::verify_multi_opening
// (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:129-260) needs two G1 sums,
//     A_b = sum_k r_b^k pi_k
//     B_b = sum_k r_b^k h_k^64 pi_k + sum_row w_row C_row - commit(sum_k r_b^k I_k)
// with its own Fiat-Shamir challenge r_b.  What does not depend on the challenge (point decoding, subgroup tests, cell
// decoding, per-cell interpolation) runs over the concatenation of all problems with the single-problem kernels of
// k_g1misc.hip / k_verify.hip; this file adds what is per problem:
//   k_vm_scalars      r_b^k, the two proof scalars, from a per-problem table of r_b^(2^i)
//   k_vm_weights      w_row = sum of r_b^k over the cells of that commitment (block per unique commitment, its problem's cells only)
//   k_vm_interp_sum   the 64 coefficients of sum_k r_b^k I_k per problem (block per problem), negated, canonical
//   k_vm_mul          ONE LANE PER (point, scalar) product: a verification of 128 cells is 257 independent scalar
//                     multiplications, a thousand of them fill the chip, so no bucket method and no sorting: GLV split, both
//                     halves in signed 4-bit fixed windows over ONE table of 1..8 P brought to a common Z (g1_mulc.hpp's
//                     isomorphic-curve trick): 128 doublings + 64 mixed additions, every lane in lockstep
//   k_vm_reduce       per problem: the two sums of its products (+ the fixed-base commitment of the interpolation polynomial,
//                     which comes from the commitment window table: the first 64 SRS points are its group 0)
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "g1_subgroup.hpp"
#include "g1_coop.hpp"
#include "launch.hpp"
#include "glv.hpp"

namespace kzg {

struct VmPow { Fr p[24]; };  // r^(2^i), Montgomery

__global__ void k_vm_scalars(const VmPow* __restrict__ tabs, const int* __restrict__ batch_of, const int* __restrict__ pos_in_batch,
                             const int* __restrict__ cell_idx, const Fr* __restrict__ w8192, Fr* __restrict__ rp_mont,
                             Fr* __restrict__ s1, Fr* __restrict__ s2, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const VmPow* tab = tabs + batch_of[k];
    const int e = pos_in_batch[k];
    Fr acc = one<FrParams>();
    for (int i = 0; i < 24; i++)
        if ((e >> i) & 1) acc = mul(acc, tab->p[i]);  // compute_powers (verifier.rs:333-343)
    rp_mont[k] = acc;
    s1[k] = from_mont(acc);
    // coset generator h_c = omega_8192^brp7(c) (cosets.rs:89-112); h_c^64 = omega_128^brp7(c) = w8192[64 * brp7(c)]
    const int bc = (int)(__brev((unsigned)cell_idx[k]) >> 25);
    s2[k] = from_mont(mul(acc, w8192[64 * bc]));
}

__device__ __forceinline__ Fr vm_block_sum(uint32_t (*part)[256], Fr acc, int t) {
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
    Fr a;
#pragma unroll
    for (int l = 0; l < 8; l++) a.v[l] = part[l][0];
    return a;
}
// weights[row] = sum_{k in the row's problem : row_k == row} r^k (verifier.rs:216-219), canonical; rows are global
// (problem's first row + local row), row_batch[row] = its problem, [cell_start[b], cell_start[b + 1]) that problem's cells
__global__ __launch_bounds__(256) void k_vm_weights(const Fr* __restrict__ rp_mont, const int* __restrict__ row,
                                                    const int* __restrict__ row_batch, const int* __restrict__ cell_start,
                                                    Fr* __restrict__ weights) {
    __shared__ uint32_t part[8][256];
    const int r = blockIdx.x, t = threadIdx.x, b = row_batch[r];
    const int lo = cell_start[b], hi = cell_start[b + 1];
    Fr acc = zero<FrParams>();
    for (int k = lo + t; k < hi; k += 256)
        if (row[k] == r) acc = add(acc, rp_mont[k]);
    acc = vm_block_sum(part, acc, t);
    if (t == 0) weights[r] = from_mont(acc);
}
// out[b][i] = -(sum_{k in b} r^k coef[k][i]), canonical: the scalars of the interpolation commitment (verifier.rs:348-384, :235)
__global__ __launch_bounds__(256) void k_vm_interp_sum(const Fr* __restrict__ coef, const Fr* __restrict__ rp_mont,
                                                       const int* __restrict__ cell_start, Fr* __restrict__ out_neg_canon) {
    __shared__ uint32_t s[4][8][64];
    const int b = blockIdx.x, q = threadIdx.x >> 6, i = threadIdx.x & 63;
    const int lo = cell_start[b], hi = cell_start[b + 1];
    Fr acc = zero<FrParams>();
    for (int k = lo + q; k < hi; k += 4) acc = add(acc, mul(coef[(size_t)k * 64 + i], rp_mont[k]));
#pragma unroll
    for (int l = 0; l < 8; l++) s[q][l][i] = acc.v[l];
    __syncthreads();
    if (q == 0) {
        for (int o = 1; o < 4; o++) {
            Fr x;
#pragma unroll
            for (int l = 0; l < 8; l++) x.v[l] = s[o][l][i];
            acc = add(acc, x);
        }
        out_neg_canon[(size_t)b * 64 + i] = from_mont(neg(acc));
    }
}

// signed Booth digit of 4-bit window w (< 32) of a 127-bit magnitude held in four words (bit 127 = sign, masked by the caller)
__device__ __forceinline__ int vm_booth4(const uint32_t m[4], int w) {
    uint32_t x;
    if (w == 0) x = (m[0] << 1) & 0x1fu;
    else {
        const int lo = 4 * w - 1, word = lo >> 5, sh = lo & 31;
        uint64_t two = m[word];
        if (word + 1 < 4) two |= (uint64_t)m[word + 1] << 32;
        x = (uint32_t)(two >> sh) & 0x1fu;
    }
    const int t = (int)((x + 1) >> 1);
    return (x >> 4) ? t - 16 : t;
}
// k P for a PER-LANE scalar k = s1 m1 + s2 m2 lambda (balanced GLV halves, glv.hpp): the table (j + 1) P, j < 8, is built with
// mixed additions (P is affine) and brought to one common Z without an inversion -- (X_j l_j^2, Y_j l_j^3), l_j = Z / z_j,
// are affine coordinates on the isomorphic curve y^2 = x^3 + 4 Z^6, whose group law (a = 0) and endomorphism are the same
// (g1_mulc.hpp) -- so the 64 additions of the main loop are mixed additions too.  All lanes run the same 32 windows:
// 4 doublings, one addition for k1's digit, one for k2's (phi of a table entry only swaps in beta x); a zero digit masks its lane.
__device__ __forceinline__ JacQ vm_step(const JacQ& t, const AffQ& p) { return add_mixed(t, p); }
__device__ __forceinline__ JacQ vm_step(const JacQ& t, const JacQ& p) { return add(t, p); }
__device__ __forceinline__ JacQ vm_lift(const AffQ& p) { return to_jacq(p); }
__device__ __forceinline__ JacQ vm_lift(const JacQ& p) { return p; }
template <class Base>  // AffQ (decoded input points) or JacQ (sums that are multiplied again: the folded pairing inputs)
__device__ JacQ vm_mul_by_scalar(const Base& P, const uint32_t split[8], const Fq<1>& beta) {
    constexpr int NT = 8;
    AffQ2 A[NT];
    Fq<2> bx[NT];
    Fq<ZB> zc;
    {
        JacQ T[NT];
        Fq<ZB> pre[NT];  // pre[j] = z_0 ... z_j
        T[0] = vm_lift(P);
        pre[0] = T[0].z;
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            T[j] = j == 1 ? dbl(T[0]) : vm_step(T[j - 1], P);
            pre[j] = relax<ZB>(mul(pre[j - 1], T[j].z));
        }
        zc = pre[NT - 1];
        Fq<ZB> suf = relax<ZB>(fq_one());  // z_(j+1) ... z_(NT-1)
#pragma unroll 1
        for (int j = NT - 1; j >= 0; j--) {
            const Fq<ZB> lam = j > 0 ? relax<ZB>(mul(pre[j > 0 ? j - 1 : 0], suf)) : suf;  // product of all the other z
            const Fq<2> l2 = sqr(lam);
            A[j].x = mul(T[j].x, l2);
            A[j].y = mul(T[j].y, mul(l2, lam));
            bx[j] = mul(A[j].x, beta);
            suf = relax<ZB>(mul(suf, T[j].z));
        }
    }
    uint32_t m1[4] = {split[0], split[1], split[2], split[3] & 0x7fffffffu};
    uint32_t m2[4] = {split[4], split[5], split[6], split[7] & 0x7fffffffu};
    const bool n1 = (split[3] >> 31) != 0, n2 = (split[7] >> 31) != 0;
    JacQ acc = jacq_inf();
#pragma unroll 1
    for (int w = 31; w >= 0; w--) {
        if (w != 31) {
#pragma unroll 1
            for (int s = 0; s < 4; s++) acc = dbl(acc);
        }
        const int d1 = vm_booth4(m1, w), d2 = vm_booth4(m2, w);
        if (d1 != 0) acc = add_mixed(acc, A[(d1 < 0 ? -d1 : d1) - 1], (d1 < 0) != n1);
        if (d2 != 0) {
            AffQ2 op = A[(d2 < 0 ? -d2 : d2) - 1];
            op.x = bx[(d2 < 0 ? -d2 : d2) - 1];
            acc = add_mixed(acc, op, (d2 < 0) != n2);
        }
    }
    acc.z = relax<ZB>(mul(acc.z, zc));  // back from the isomorphic curve (an identity P gives zc = 0: the identity)
    return acc;
}
// The same product by the four lanes of a quad (g1_coop.hpp), for the small passes whose waves have a SIMD each: the doublings
// and the mixed additions of the main loop and of the table are shared (3.5 and 5.5 multiplication times instead of 6.5 and
// 10.5), the table's common-Z step is repeated on every lane.  All four lanes hold the same P and the same scalar, so the
// digit tests are uniform in the quad.
__device__ __forceinline__ JacQ vm_step_coop(const JacQ& t, const AffQ& p, int quad) { return coop_add_mixed(t, p, false, quad); }
__device__ __forceinline__ JacQ vm_step_coop(const JacQ& t, const JacQ& p, int quad) { return coop_add(t, p, false, quad); }
template <class Base>
__device__ JacQ vm_mul_by_scalar_coop(const Base& P, const uint32_t split[8], const Fq<1>& beta, int quad) {
    constexpr int NT = 8;
    AffQ2 A[NT];
    Fq<2> bx[NT];
    Fq<ZB> zc;
    {
        JacQ T[NT];
        Fq<ZB> pre[NT];
        T[0] = vm_lift(P);
        pre[0] = T[0].z;
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            T[j] = j == 1 ? coop_dbl(T[0], quad) : vm_step_coop(T[j - 1], P, quad);
            pre[j] = relax<ZB>(mul(pre[j - 1], T[j].z));
        }
        zc = pre[NT - 1];
        Fq<ZB> suf = relax<ZB>(fq_one());
#pragma unroll 1
        for (int j = NT - 1; j >= 0; j--) {
            const Fq<ZB> lam = j > 0 ? relax<ZB>(mul(pre[j > 0 ? j - 1 : 0], suf)) : suf;
            const Fq<2> l2 = sqr(lam);
            A[j].x = mul(T[j].x, l2);
            A[j].y = mul(T[j].y, mul(l2, lam));
            bx[j] = mul(A[j].x, beta);
            suf = relax<ZB>(mul(suf, T[j].z));
        }
    }
    uint32_t m1[4] = {split[0], split[1], split[2], split[3] & 0x7fffffffu};
    uint32_t m2[4] = {split[4], split[5], split[6], split[7] & 0x7fffffffu};
    const bool n1 = (split[3] >> 31) != 0, n2 = (split[7] >> 31) != 0;
    JacQ acc = jacq_inf();
#pragma unroll 1
    for (int w = 31; w >= 0; w--) {
        if (w != 31) {
#pragma unroll 1
            for (int s = 0; s < 4; s++) acc = coop_dbl(acc, quad);
        }
        const int d1 = vm_booth4(m1, w), d2 = vm_booth4(m2, w);
        if (d1 != 0) acc = coop_add_mixed(acc, A[(d1 < 0 ? -d1 : d1) - 1], (d1 < 0) != n1, quad);
        if (d2 != 0) {
            AffQ2 op = A[(d2 < 0 ? -d2 : d2) - 1];
            op.x = bx[(d2 < 0 ? -d2 : d2) - 1];
            acc = coop_add_mixed(acc, op, (d2 < 0) != n2, quad);
        }
    }
    acc.z = relax<ZB>(mul(acc.z, zc));
    return acc;
}
// products e < n: s1[e] pi_e;  n <= e < 2n: s2[e - n] pi_(e - n);  2n <= e < 2n + m: w[e - 2n] C_(e - 2n)
// pts = [proofs n | commitments m] affine Montgomery-384 (decoded); scalars canonical
__global__ __launch_bounds__(64, 2) void k_vm_mul(const G1Affine* __restrict__ pts, const Fr* __restrict__ s1, const Fr* __restrict__ s2,
                                                  const Fr* __restrict__ wts, JacQ* __restrict__ prod, int n, int m, Fq<1> beta) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= 2 * n + m) return;
    const int pi = e < n ? e : e < 2 * n ? e - n : n + (e - 2 * n);
    const Fr k = e < n ? s1[e] : e < 2 * n ? s2[e - n] : wts[e - 2 * n];
    uint32_t split[8];
    glv_split_balanced(k, split);
    prod[e] = vm_mul_by_scalar(affq_from_affine(pts[pi]), split, beta);
}
// SMALL passes (a handful of problems: concurrent single calls combined, verify_many.hip) are a chain of latencies, so the chain
// is cut: ONE launch holds every scalar multiplication of the pass -- the 2 n + m products above and, as 64 more lanes per
// problem, the terms isc[b][j] SRS_j of the interpolation commitments (one more 255-bit multiplication each instead of a
// window-table MSM launch of its own: 25 % more lanes, nothing on a chip that is empty anyway) -- AND the subgroup tests of the
// decoded points in further blocks (their 126 dependent doublings run beside the multiplications' 128 instead of in front).
//   products: e < n: s1[e] pi_e | n <= e < 2n: s2 pi | 2n <= e < 2n + m: w C | 2n + m <= e < 2n + m + 64 B: isc[b][j] SRS_j
__global__ __launch_bounds__(64, 2) void k_vm_mul_small(const G1Affine* __restrict__ pts, const Fr* __restrict__ s1, const Fr* __restrict__ s2,
                                                        const Fr* __restrict__ wts, const Fr* __restrict__ isc, const G1Affine* __restrict__ srs,
                                                        JacQ* __restrict__ prod, int n, int m, int n_batches, int mul_blocks,
                                                        int* __restrict__ status /*[n + m]*/, Fq<1> beta) {
    if ((int)blockIdx.x >= mul_blocks) {  // subgroup tests of [proofs n | commitments m]: status 0 -> 0 / 2, others kept
        const int i = ((int)blockIdx.x - mul_blocks) * 64 + threadIdx.x;
        if (i >= n + m || status[i] != 0) return;
        const G1Affine a = pts[i];
        if (is_inf(a)) return;
        if (!g1_in_subgroup_q(affq_from_affine(a), beta)) status[i] = 2;
        return;
    }
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int base = 2 * n + m;
    if (e >= base + 64 * n_batches) return;
    G1Affine P;
    Fr k;
    if (e < base) {
        P = pts[e < n ? e : e < 2 * n ? e - n : n + (e - 2 * n)];
        k = e < n ? s1[e] : e < 2 * n ? s2[e - n] : wts[e - 2 * n];
    } else {
        P = srs[(e - base) & 63];
        k = isc[e - base];
    }
    uint32_t split[8];
    glv_split_balanced(k, split);
    prod[e] = vm_mul_by_scalar(affq_from_affine(P), split, beta);
}
// k_vm_mul_small with four lanes per product and per subgroup test (16 of either per block): what a pass of a few problems takes
__global__ __launch_bounds__(64, 2) void k_vm_mul_small_coop(const G1Affine* __restrict__ pts, const Fr* __restrict__ s1, const Fr* __restrict__ s2,
                                                             const Fr* __restrict__ wts, const Fr* __restrict__ isc, const G1Affine* __restrict__ srs,
                                                             JacQ* __restrict__ prod, int n, int m, int n_batches, int mul_blocks,
                                                             int* __restrict__ status /*[n + m]*/, Fq<1> beta) {
    const int quad = threadIdx.x & 3, sub = threadIdx.x >> 2;
    if ((int)blockIdx.x >= mul_blocks) {
        const int i = ((int)blockIdx.x - mul_blocks) * 16 + sub;
        if (i >= n + m || status[i] != 0) return;
        const G1Affine a = pts[i];
        if (is_inf(a)) return;
        if (!g1_in_subgroup_coop(affq_from_affine(a), beta, quad)) status[i] = 2;
        return;
    }
    const int e = blockIdx.x * 16 + sub;
    const int base = 2 * n + m;
    if (e >= base + 64 * n_batches) return;
    G1Affine P;
    Fr k;
    if (e < base) {
        P = pts[e < n ? e : e < 2 * n ? e - n : n + (e - 2 * n)];
        k = e < n ? s1[e] : e < 2 * n ? s2[e - n] : wts[e - 2 * n];
    } else {
        P = srs[(e - base) & 63];
        k = isc[e - base];
    }
    uint32_t split[8];
    glv_split_balanced(k, split);
    prod[e] = vm_mul_by_scalar_coop(affq_from_affine(P), split, beta, quad);
}
// two trees of 128 partial sums side by side (job 0 in red[0..128), job 1 in red[128..256)), each folded by its half of the
// block with four lanes per addition (g1_coop.hpp: the idle lanes of a tree level share its additions); red[128 * job] = the sums
__device__ __forceinline__ void vm_two_trees(JacQ* red, int t, int first_span = 64) {
    const int job = t >> 7, l = t & 127, quad = l & 3, slot = l >> 2;  // 32 additions per round and half
    JacQ* mine = red + 128 * job;
    __syncthreads();
#pragma unroll 1
    for (int span = first_span; span >= 1; span >>= 1) {
#pragma unroll 1
        for (int base = 0; base < span; base += 32) {
            const int a = base + slot;
            if (a < span) {
                const JacQ r = coop_add(mine[a], mine[a + span], false, quad);
                if (quad == 0) mine[a] = r;
            }
        }
        __syncthreads();
    }
}
// the sums of a small pass: as k_vm_reduce, with the 64 interpolation terms of the problem in place of icommit[b]
__global__ __launch_bounds__(256) void k_vm_reduce_small(const JacQ* __restrict__ prod, const int* __restrict__ cell_start,
                                                         const int* __restrict__ row_start, JacQ* __restrict__ out, int n, int m) {
    __shared__ JacQ red[256];
    const int b = blockIdx.x, t = threadIdx.x, job = t >> 7, l = t & 127;
    const int lo = cell_start[b], hi = cell_start[b + 1], rlo = row_start[b], rhi = row_start[b + 1];
    JacQ acc = jacq_inf();
    const JacQ* src = prod + (job ? n : 0);
    for (int k = lo + l; k < hi; k += 128) acc = add(acc, src[k]);
    if (job) {
        for (int r = rlo + l; r < rhi; r += 128) acc = add(acc, prod[2 * (size_t)n + r]);
        if (l < 64) acc = add(acc, prod[2 * (size_t)n + m + 64 * (size_t)b + l]);
    }
    red[t] = acc;
    vm_two_trees(red, t);
    if (l == 0) out[2 * (size_t)b + job] = red[128 * job];
}
// per problem b: out[2b] = sum of prod[cells of b], out[2b + 1] = sum of prod[n + cells of b] + sum of prod[2n + rows of b]
// + icommit[b]; threads 0-127 take the first sum, 128-255 the second, 7-level trees in LDS
__global__ __launch_bounds__(256) void k_vm_reduce(const JacQ* __restrict__ prod, const JacQ* __restrict__ icommit,
                                                   const int* __restrict__ cell_start, const int* __restrict__ row_start,
                                                   JacQ* __restrict__ out, int n) {
    __shared__ JacQ red[256];
    const int b = blockIdx.x, t = threadIdx.x, job = t >> 7, l = t & 127;
    const int lo = cell_start[b], hi = cell_start[b + 1], rlo = row_start[b], rhi = row_start[b + 1];
    JacQ acc = jacq_inf();
    const JacQ* src = prod + (job ? n : 0);
    for (int k = lo + l; k < hi; k += 128) acc = add(acc, src[k]);
    if (job) {
        for (int r = rlo + l; r < rhi; r += 128) acc = add(acc, prod[2 * (size_t)n + r]);
        if (l == 0) acc = add(acc, icommit[b]);
    }
    red[t] = acc;
    vm_two_trees(red, t);
    if (l == 0) out[2 * (size_t)b + job] = red[128 * job];
}

// Folding the problems' pairing checks into ONE: with weights rho_b (128 bits, derived by the host from ALL the problems'
// challenges, 0 for a problem that is excluded) the 2 x B sums become  S_j = sum_b rho_b out[b][j],  and
// e(S_0, [tau^64]_2) e(S_1, -[1]_2) = 1 holds for every choice of weights iff every problem's own equation holds (up to 2^-128
// for weights the prover cannot predict: the same argument as the powers of r inside one batch, verifier.rs:148-160).
// k_vm_fold_mul: lane = (problem, j): rho_b * out[b][j] (a 128-bit scalar: k1 only);  k_vm_fold_sum: one block, two trees.
__global__ __launch_bounds__(64, 2) void k_vm_fold_mul(const JacQ* __restrict__ sums, const uint32_t* __restrict__ rho /*[B][4]*/,
                                                       JacQ* __restrict__ prod, int n_batches, Fq<1> beta) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= 2 * n_batches) return;
    const int b = e >> 1;
    uint32_t split[8] = {rho[4 * b], rho[4 * b + 1], rho[4 * b + 2], rho[4 * b + 3] & 0x7fffffffu, 0, 0, 0, 0};
    prod[e] = vm_mul_by_scalar(sums[e], split, beta);
}
// the same with four lanes per product (g1_coop.hpp): the 2 B products of a pass are a few waves on an idle chip, and each is a
// 127-bit multiplication's chain of 124 doublings and ~32 additions
__global__ __launch_bounds__(64, 2) void k_vm_fold_mul_coop(const JacQ* __restrict__ sums, const uint32_t* __restrict__ rho /*[B][4]*/,
                                                            JacQ* __restrict__ prod, int n_batches, Fq<1> beta) {
    const int e = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (e >= 2 * n_batches) return;
    const int b = e >> 1;
    uint32_t split[8] = {rho[4 * b], rho[4 * b + 1], rho[4 * b + 2], rho[4 * b + 3] & 0x7fffffffu, 0, 0, 0, 0};
    prod[e] = vm_mul_by_scalar_coop(sums[e], split, beta, quad);
}
__global__ __launch_bounds__(256) void k_vm_fold_sum(const JacQ* __restrict__ prod, JacQ* __restrict__ out2, int n_batches) {
    __shared__ JacQ red[256];
    const int t = threadIdx.x, job = t >> 7, l = t & 127;
    JacQ acc = jacq_inf();
    for (int b = l; b < n_batches; b += 128) acc = add(acc, prod[2 * (size_t)b + job]);
    red[t] = acc;
    vm_two_trees(red, t);
    if (l == 0) out2[job] = red[128 * job];
}

// sums of the weighted per-problem pairs over RANGES of problems: out[r][j] = sum of prod[b][j], ranges[r][0] <= b < ranges[r][1]
// (the probes of the search for the wrong proofs of a pass whose folded check failed; the products rho_b * sum_b are resident)
__global__ __launch_bounds__(256) void k_vm_fold_ranges(const JacQ* __restrict__ prod, const int* __restrict__ ranges, JacQ* __restrict__ out) {
    __shared__ JacQ red[256];
    const int t = threadIdx.x, job = t >> 7, l = t & 127;
    const int lo = ranges[2 * blockIdx.x], hi = ranges[2 * blockIdx.x + 1];
    JacQ acc = jacq_inf();
    for (int b = lo + l; b < hi; b += 128) acc = add(acc, prod[2 * (size_t)b + job]);
    const int width = hi - lo < 128 ? hi - lo : 128;  // lanes that hold anything
#pragma unroll 1
    for (int span = 64; span >= 1; span >>= 1) {
        if (span >= width) continue;  // (block-uniform) nothing to fold at this distance
        red[t] = acc;
        __syncthreads();
        if (l < span) acc = add(acc, red[t + span]);
        __syncthreads();
    }
    if (l == 0) out[2 * (size_t)blockIdx.x + job] = acc;
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_verify_many() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_vm_scalars));
}
static Fr vm_fr(const Fr8& x) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }
void vm_scalars(const void* pow_tables /*[B][24] Fr, device*/, const int* batch_of, const int* pos_in_batch, const int* cell_idx,
                const void* w8192, void* rp_mont, void* s1, void* s2, int n, hipStream_t st) {
    if (n > 0) k_vm_scalars<<<(n + 255) / 256, 256, 0, st>>>((const VmPow*)pow_tables, batch_of, pos_in_batch, cell_idx, (const Fr*)w8192,
                                                               (Fr*)rp_mont, (Fr*)s1, (Fr*)s2, n);
}
void vm_weights(const void* rp_mont, const int* row, const int* row_batch, const int* cell_start, void* weights, int m, hipStream_t st) {
    if (m > 0) k_vm_weights<<<m, 256, 0, st>>>((const Fr*)rp_mont, row, row_batch, cell_start, (Fr*)weights);
}
void vm_interp_sum(const void* coef, const void* rp_mont, const int* cell_start, void* out_neg_canon, int n_batches, hipStream_t st) {
    if (n_batches > 0) k_vm_interp_sum<<<n_batches, 256, 0, st>>>((const Fr*)coef, (const Fr*)rp_mont, cell_start, (Fr*)out_neg_canon);
}
void vm_mul(const void* pts, const void* s1, const void* s2, const void* wts, void* prod, int n, int m, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const int total = 2 * n + m;
    if (total > 0)
        k_vm_mul<<<(total + 63) / 64, 64, 0, st>>>((const G1Affine*)pts, (const Fr*)s1, (const Fr*)s2, (const Fr*)wts, (JacQ*)prod, n, m, fq_from_fp(b384));
}
void vm_mul_small(const void* pts, const void* s1, const void* s2, const void* wts, const void* isc, const void* srs, void* prod, int n, int m,
                  int n_batches, int* status, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const int products = 2 * n + m + 64 * n_batches;
    if (products + n + m <= 2 * coop_points_max()) {  // <= 1,024 waves of 16 quads: every wave still has a SIMD to itself
        const int mb = (products + 15) / 16, sb = (n + m + 15) / 16;
        k_vm_mul_small_coop<<<mb + sb, 64, 0, st>>>((const G1Affine*)pts, (const Fr*)s1, (const Fr*)s2, (const Fr*)wts, (const Fr*)isc,
                                                    (const G1Affine*)srs, (JacQ*)prod, n, m, n_batches, mb, status, fq_from_fp(b384));
        return;
    }
    const int mul_blocks = (2 * n + m + 64 * n_batches + 63) / 64, sub_blocks = (n + m + 63) / 64;
    k_vm_mul_small<<<mul_blocks + sub_blocks, 64, 0, st>>>((const G1Affine*)pts, (const Fr*)s1, (const Fr*)s2, (const Fr*)wts, (const Fr*)isc,
                                                           (const G1Affine*)srs, (JacQ*)prod, n, m, n_batches, mul_blocks, status, fq_from_fp(b384));
}
void vm_reduce_small(const void* prod, const int* cell_start, const int* row_start, void* out, int n, int m, int n_batches, hipStream_t st) {
    if (n_batches > 0) k_vm_reduce_small<<<n_batches, 256, 0, st>>>((const JacQ*)prod, cell_start, row_start, (JacQ*)out, n, m);
}
void vm_fold(const void* sums, const uint32_t* rho, void* prod, void* out2, int n_batches, const Fp12w& beta, hipStream_t st) {
    if (n_batches <= 0) return;
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    if (2 * n_batches <= 2 * coop_points_max())
        k_vm_fold_mul_coop<<<(2 * n_batches + 15) / 16, 64, 0, st>>>((const JacQ*)sums, rho, (JacQ*)prod, n_batches, fq_from_fp(b384));
    else k_vm_fold_mul<<<(2 * n_batches + 63) / 64, 64, 0, st>>>((const JacQ*)sums, rho, (JacQ*)prod, n_batches, fq_from_fp(b384));
    k_vm_fold_sum<<<1, 256, 0, st>>>((const JacQ*)prod, (JacQ*)out2, n_batches);
}
void vm_fold_ranges(const void* prod, const int* ranges, void* out, int n_ranges, hipStream_t st) {
    if (n_ranges > 0) k_vm_fold_ranges<<<n_ranges, 256, 0, st>>>((const JacQ*)prod, ranges, (JacQ*)out);
}
void vm_reduce(const void* prod, const void* icommit, const int* cell_start, const int* row_start, void* out, int n, int n_batches,
               hipStream_t st) {
    if (n_batches > 0) k_vm_reduce<<<n_batches, 256, 0, st>>>((const JacQ*)prod, (const JacQ*)icommit, cell_start, row_start, (JacQ*)out, n);
}
}  // namespace launch
}  // namespace kzg
