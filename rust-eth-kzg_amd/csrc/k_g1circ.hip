// Small-batch form of stages E + F of compute_cells_and_kzg_proofs (BASELINE.json config 2: a single blob).
// Reference: Domain::ifft_g1_take_n followed by Domain::fft_g1 (crates/cryptography/polynomial/src/domain.rs:149-194)
// as called by compute_h_poly_commitments / compute_multi_opening_proofs (fk20/h_poly.rs:18-68, fk20/prover.rs:190-215).
//
// "inverse FFT, keep 64, forward FFT" is ONE fixed linear map on the 128 points u[j] that leave the MSM stage:
//     F[k] = sum_j c_(k-j) u[j],   c_d = sum_{t<64} w^(d t)        (the 1/128 is already folded into the MSM scalars)
// a circulant whose symbol vanishes for every even d != 0:  c_0 = 64, c_d = 2 / (1 - w^d) for odd d.  So each output
// is 64 products by PUBLIC scalars plus 64 u[k].  With a handful of blobs the radix-2 network (14 dependent scalar
// multiplications) leaves > 95 % of the SIMDs idle; here the dependent chain is as short as the group law allows:
//   k_g1_dbl_table : D[b][j][half][t] = 2^t u[j] (half 0) and phi(2^t u[j]) (half 1), t < T   -- ONE chain of T doublings
//   k_g1_circ_sum  : F[k] = sum over the ~5.6 k non-zero NAF digits of the GLV halves of all c_d of +-D[b][k-d][half][t],
//                    spread over 256 lanes (22 additions each) + an 8-level tree in LDS.
// ~45x the additions of the radix-2 network per blob, but on SIMDs that would otherwise idle; used for <= CIRC_MAX blobs.
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "g1_coop.hpp"
#include "curve30.hpp"
#include "g1_coop30.hpp"
#include "launch.hpp"

namespace kzg {
using launch::CIRC_LANES;

// The kernels below are written once for both point forms: JacS, the signed 13 x 30-bit field (launch::FMT_JACS: what the engine
// runs -- with it the single-blob path, too, is in the prover's ONE Fp representation from the MSM's fold to the proof bytes; round 6),
// and JacQ, the 14 x 29-bit field (ETH_KZG_AMD_ARENA_SIGNED=0, the cross-check).  A doubling of the signed form is the halved one:
// (X / 4, Y / 8, Z / 2) is the same point, so the table holds 2^t u[j] all the same.
template <class Pt> struct CircOps;
template <> struct CircOps<JacQ> {
    using Beta = Fq<1>;
    static __device__ __forceinline__ JacQ dbl1(const JacQ& p) { return dbl(p); }
    static __device__ __forceinline__ void set_phi_x(JacQ& q, const JacQ& p, const Beta& beta) { q.x = relax<XB>(mul(p.x, beta)); }
    static __device__ __forceinline__ JacQ coop_add4(const JacQ& p, const JacQ& q, bool negq, int quad) { return coop_add(p, q, negq, quad); }
    template <int NT> static __device__ __forceinline__ void fold(JacQ* red, int first_span, int tid) { coop_tree_fold<NT>(red, first_span, tid); }
    // coop_dbl with the product beta X in lane 3 of its first level
    static __device__ __forceinline__ JacQ coop_dbl_phi(const JacQ& p, int quad, const Beta& beta, JacQ& phi) {
        const bool l0 = quad == 0, l1 = quad == 1, l3 = quad == 3;
        const Fq<XB> bw = relax<XB>(beta);
        const Fq<XB> a1 = select(l0 || l3, p.x, p.y);
        const Fq<XB> b1 = select(l0, p.x, select(l1, p.y, select(l3, bw, relax<XB>(p.z))));
        const Fq<2> r1 = mul(a1, b1);
        const Fq<2> A = quad_bcast<0>(r1), B = quad_bcast<1>(r1), YZ = quad_bcast<2>(r1), bX = quad_bcast<3>(r1);
        phi = p;
        phi.x = relax<XB>(bX);
        const Fq<6> E = add(dbl(A), A);
        const Fq<2> r2 = mul(select(l0, p.x, relax<XB>(E)), select(l0, relax<XB>(B), relax<XB>(E)));
        const Fq<2> XY2 = quad_bcast<0>(r2), F = quad_bcast<1>(r2);
        const Fq<8> Dd = dbl2(XY2);
        auto x3 = sub2(F, Dd);
        auto y3 = mul_add(E, sub(Dd, x3), neg2(B), dbl2(B));
        JacQ r;
        r.x = relax<XB>(x3);
        r.y = relax<XB>(y3);
        r.z = dbl(YZ);
        return r;
    }
};
template <> struct CircOps<JacS> {
    using Beta = Fs<1, DC>;
    static __device__ __forceinline__ JacS dbl1(const JacS& p) { return dbl_half(p); }
    static __device__ __forceinline__ void set_phi_x(JacS& q, const JacS& p, const Beta& beta) { q.x = relax<4, DC>(mul(p.x, beta)); }
    static __device__ __forceinline__ JacS coop_add4(const JacS& p, const JacS& q, bool negq, int quad) { return coop4_add(p, q, negq, quad); }
    template <int NT> static __device__ __forceinline__ void fold(JacS* red, int first_span, int tid) { coop4_tree_fold<NT>(red, first_span, tid); }
    static __device__ __forceinline__ JacS coop_dbl_phi(const JacS& p, int quad, const Beta& beta, JacS& phi) {
        Fs<1, DC> bx;
        const JacS r = coop4_dbl_half_phi(p, quad, beta, bx);
        phi = p;
        phi.x = relax<4, DC>(bx);
        return r;
    }
};

// segs > 1: the MSM stage has also produced 2^(32 s) u[j] in lane s * n + b (k_fk20_scalars scales the scalars), so the
// chain of T doublings splits into `segs` independent chains of 128 / segs (the last takes the remainder): thread = (segment, blob, j).
template <class Pt>
__global__ __launch_bounds__(64) void k_g1_dbl_table(const Pt* __restrict__ X, int stride, int n, int segs, Pt* __restrict__ D,
                                                     int T, typename CircOps<Pt>::Beta beta) {
    const int tid = blockIdx.x * 64 + threadIdx.x;
    if (tid >= segs * n * N_CELLS) return;
    const int seg = tid / (n * N_CELLS), bj = tid - seg * n * N_CELLS;
    const int b = bj >> 7, j = bj & 127;
    const int seg_len = 128 / segs;  // 32 (four segments) or 64 (two)
    const int t0 = seg_len * seg;
    const int t1 = seg + 1 < segs ? t0 + seg_len : T;
    Pt p = X[(size_t)j * stride + seg * n + b];
    Pt* d0 = D + (size_t)bj * 2 * T;
    Pt* d1 = d0 + T;
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        d0[t] = p;
        Pt q = p;
        CircOps<Pt>::set_phi_x(q, p, beta);  // phi(X : Y : Z) = (beta X : Y : Z)
        d1[t] = q;
        p = CircOps<Pt>::dbl1(p);
    }
}

// term word: bits 0-6 d (column offset), 7-14 t, 15 half, 16 minus, 17 valid.  terms[i * CIRC_LANES + lane];
// row 0 holds valid, positive terms only (the host orders them so), so every lane starts from a table entry.
template <class Pt>
__global__ __launch_bounds__(CIRC_LANES) void k_g1_circ_sum(const Pt* __restrict__ D, int T, const uint32_t* __restrict__ terms,
                                                            int per_lane, Pt* __restrict__ X, int stride) {
    __shared__ Pt part[CIRC_LANES];
    const int k = blockIdx.x, b = blockIdx.y, l = threadIdx.x;
    const Pt* Db = D + (size_t)b * N_CELLS * 2 * T;
    auto entry = [&](uint32_t w) -> const Pt* {
        const int j = (k - (int)(w & 127)) & 127;
        return Db + ((size_t)(j * 2 + ((w >> 15) & 1))) * T + ((w >> 7) & 255);
    };
    Pt acc = *entry(terms[l]);
    constexpr int LEVELS = 8;  // log2(CIRC_LANES)
    static_assert(CIRC_LANES == 1 << LEVELS, "tree depth");
#pragma unroll 1
    for (int s = 0; s < per_lane - 1; s++) {
        const uint32_t w = terms[(size_t)(s + 1) * CIRC_LANES + l];
        if ((w >> 17) & 1) acc = add(acc, *entry(w), (w >> 16) & 1);
    }
    part[l] = acc;
    CircOps<Pt>::template fold<CIRC_LANES>(part, CIRC_LANES / 2, l);  // the tree's idle lanes share its additions
    if (l == 0) X[(size_t)(__brev((unsigned)k) >> 25) * stride + b] = part[0];  // proofs leave in bit-reversed order
}

// The same two kernels with FOUR lanes per chain (g1_coop.hpp, g1_coop30.hpp): a handful of blobs leaves the chip idle, and both
// kernels are dependent chains -- T / segs doublings, then ~30 general additions per lane of the sum.  The quad shares each doubling
// (the beta X of the phi image rides in a lane the doubling leaves idle) and each addition.
template <class Pt>
__global__ __launch_bounds__(64) void k_g1_dbl_table_coop(const Pt* __restrict__ X, int stride, int n, int segs, Pt* __restrict__ D,
                                                          int T, typename CircOps<Pt>::Beta beta) {
    const int tid = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (tid >= segs * n * N_CELLS) return;
    const int seg = tid / (n * N_CELLS), bj = tid - seg * n * N_CELLS;
    const int b = bj >> 7, j = bj & 127;
    const int seg_len = 128 / segs;
    const int t0 = seg_len * seg;
    const int t1 = seg + 1 < segs ? t0 + seg_len : T;
    Pt p = X[(size_t)j * stride + seg * n + b];
    Pt* d0 = D + (size_t)bj * 2 * T;
    Pt* d1 = d0 + T;
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        Pt phi;
        const Pt p2 = CircOps<Pt>::coop_dbl_phi(p, quad, beta, phi);
        if (quad == 0) d0[t] = p;
        if (quad == 1) d1[t] = phi;
        p = p2;
    }
}
// Two blocks per output (blockIdx.z), each 64 logical lanes of four threads = four waves, one per SIMD of a CU: a block of 128
// logical lanes would put two waves on every SIMD of its CU and run at half speed next to 128 idle CUs (measured: no gain).
// Logical lane L of half h takes the columns 128 h + L and 128 h + L + 64 of the term table's rows; the halves' sums meet in
// k_g1_circ_join.  part2: [blob][k][2].
template <class Pt>
__global__ __launch_bounds__(256) void k_g1_circ_sum_coop(const Pt* __restrict__ D, int T, const uint32_t* __restrict__ terms,
                                                          int per_lane, Pt* __restrict__ part2) {
    constexpr int LL = CIRC_LANES / 4, LEVELS = 6;
    static_assert(LL == 1 << LEVELS, "tree depth");
    __shared__ Pt part[LL];
    const int k = blockIdx.x, b = blockIdx.y, half = blockIdx.z, l = threadIdx.x >> 2, quad = threadIdx.x & 3;
    const Pt* Db = D + (size_t)b * N_CELLS * 2 * T;
    auto entry = [&](uint32_t w) -> const Pt* {
        const int j = (k - (int)(w & 127)) & 127;
        return Db + ((size_t)(j * 2 + ((w >> 15) & 1))) * T + ((w >> 7) & 255);
    };
    const int col0 = 2 * LL * half + l;
    Pt acc = *entry(terms[col0]);
    const int n_terms = 2 * per_lane;  // per logical lane
    const int steps = n_terms - 1 + LEVELS;
    // the next term's table entry is requested before the current addition starts (a memory latency per step otherwise)
    Pt nxt;
    bool nxt_act = false, nxt_minus = false;
    auto fetch = [&](int i) {
        const uint32_t w = terms[(size_t)(i >> 1) * CIRC_LANES + col0 + LL * (i & 1)];
        nxt_act = (w >> 17) & 1;
        nxt_minus = (w >> 16) & 1;
        if (nxt_act) nxt = *entry(w);
    };
    if (n_terms > 1) fetch(1);
#pragma unroll 1
    for (int s = 0; s < steps; s++) {
        Pt other;
        bool act, minus = false;
        if (s < n_terms - 1) {
            other = nxt;
            act = nxt_act;
            minus = nxt_minus;
            if (s + 2 < n_terms) fetch(s + 2);
        } else {
            // the tree: a wave takes part as a whole or not at all (a wave with few lanes in use is the slow one: k_g1slp.hip,
            // k_slp_mulc_s); the quads behind the last addition of a level repeat it and keep nothing
            const int span = (LL / 2) >> (s - (n_terms - 1));
            if (quad == 0) part[l] = acc;
            __syncthreads();
            const bool wave_in = (int)((threadIdx.x & ~63u) >> 2) < span, keep = l < span;
            const int ll = keep ? l : span - 1;
            Pt mine = acc;
            if (wave_in) {
                if (!keep) mine = part[ll];
                other = part[ll + span];
            }
            __syncthreads();
            if (wave_in) {
                const Pt r = CircOps<Pt>::coop_add4(mine, other, false, quad);
                if (keep) acc = r;
            }
            continue;
        }
        if (act) acc = CircOps<Pt>::coop_add4(acc, other, minus, quad);
    }
    if (threadIdx.x == 0) part2[((size_t)b * N_CELLS + k) * 2 + half] = acc;
}
template <class Pt>
__global__ __launch_bounds__(64) void k_g1_circ_join(const Pt* __restrict__ part2, Pt* __restrict__ X, int stride, int n) {
    const int o = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (o >= n * N_CELLS) return;
    const int b = o >> 7, k = o & 127;
    const Pt r = CircOps<Pt>::coop_add4(part2[(size_t)o * 2], part2[(size_t)o * 2 + 1], false, quad);
    if (quad == 0) X[(size_t)(__brev((unsigned)k) >> 25) * stride + b] = r;  // proofs leave in bit-reversed order
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1circ() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_g1_dbl_table<JacS>));
}
size_t g1_circ_table_bytes(int n, int T) { return (size_t)n * N_CELLS * 2 * (T + 1) * sizeof(JacQ); }  // + two partial sums per output (sized for the larger point form)
template <class Pt>
static void circ128(Pt* X, int stride, int n, int segs, Pt* D, int T, const uint32_t* terms, int per_lane, typename CircOps<Pt>::Beta bt, hipStream_t st) {
    // one blob: 1.47 -> 1.27 ms per call; from two blobs on the chip is busy enough for the quads' extra instructions to cost more than
    // the shorter chains save (2 blobs 1.74 -> 1.79 ms, 4 blobs 2.29 -> 2.49): the one-lane forms stay (and ETH_KZG_AMD_COOP_POINTS=0 forces them)
    if (n == 1 && coop_points_max() > 0) {
        k_g1_dbl_table_coop<Pt><<<(segs * n * N_CELLS + 15) / 16, 64, 0, st>>>(X, stride, n, segs, D, T, bt);
        // the halves' sums go behind the table (g1_circ_table_bytes reserves the room)
        Pt* part2 = D + (size_t)n * N_CELLS * 2 * T;
        k_g1_circ_sum_coop<Pt><<<dim3(N_CELLS, n, 2), CIRC_LANES, 0, st>>>(D, T, terms, per_lane, part2);
        k_g1_circ_join<Pt><<<(n * N_CELLS + 15) / 16, 64, 0, st>>>(part2, X, stride, n);
        return;
    }
    k_g1_dbl_table<Pt><<<(segs * n * N_CELLS + 63) / 64, 64, 0, st>>>(X, stride, n, segs, D, T, bt);
    k_g1_circ_sum<Pt><<<dim3(N_CELLS, n), CIRC_LANES, 0, st>>>(D, T, terms, per_lane, X, stride);
}
// X: [128][stride] MSM outputs (natural order) -> X: proofs (bit-reversed), for blobs 0 .. n-1; fmt: the point form of X (and of D)
void g1_circ128(void* X, int stride, int n, int segs, void* D, int T, const void* terms, int per_lane, const Fp12w& beta, hipStream_t st, int fmt) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    if (fmt == FMT_JACS) circ128<JacS>((JacS*)X, stride, n, segs, (JacS*)D, T, (const uint32_t*)terms, per_lane, fs_from_fp(b384), st);
    else circ128<JacQ>((JacQ*)X, stride, n, segs, (JacQ*)D, T, (const uint32_t*)terms, per_lane, fq_from_fp(b384), st);
}
}  // namespace launch
}  // namespace kzg
