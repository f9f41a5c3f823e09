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
#include "launch.hpp"

namespace kzg {
using launch::CIRC_LANES;

// segs > 1: the MSM stage has also produced 2^(32 s) u[j] in lane s * n + b (k_fk20_scalars scales the scalars), so the
// chain of T doublings splits into `segs` independent chains of 128 / segs (the last takes the remainder): thread = (segment, blob, j).
__global__ __launch_bounds__(64) void k_g1_dbl_table(const JacQ* __restrict__ X, int stride, int n, int segs, JacQ* __restrict__ D,
                                                     int T, Fq<1> beta) {
    const int tid = blockIdx.x * 64 + threadIdx.x;
    if (tid >= segs * n * N_CELLS) return;
    const int seg = tid / (n * N_CELLS), bj = tid - seg * n * N_CELLS;
    const int b = bj >> 7, j = bj & 127;
    const int seg_len = 128 / segs;  // 32 (four segments) or 64 (two)
    const int t0 = seg_len * seg;
    const int t1 = seg + 1 < segs ? t0 + seg_len : T;
    JacQ p = X[(size_t)j * stride + seg * n + b];
    JacQ* d0 = D + (size_t)bj * 2 * T;
    JacQ* d1 = d0 + T;
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        d0[t] = p;
        JacQ q = p;
        q.x = relax<XB>(mul(p.x, beta));  // phi(X : Y : Z) = (beta X : Y : Z)
        d1[t] = q;
        p = dbl(p);
    }
}

// term word: bits 0-6 d (column offset), 7-14 t, 15 half, 16 minus, 17 valid.  terms[i * CIRC_LANES + lane];
// row 0 holds valid, positive terms only (the host orders them so), so every lane starts from a table entry.
__global__ __launch_bounds__(CIRC_LANES) void k_g1_circ_sum(const JacQ* __restrict__ D, int T, const uint32_t* __restrict__ terms,
                                                            int per_lane, JacQ* __restrict__ X, int stride) {
    __shared__ JacQ part[CIRC_LANES];
    const int k = blockIdx.x, b = blockIdx.y, l = threadIdx.x;
    const JacQ* Db = D + (size_t)b * N_CELLS * 2 * T;
    auto entry = [&](uint32_t w) -> const JacQ* {
        const int j = (k - (int)(w & 127)) & 127;
        return Db + ((size_t)(j * 2 + ((w >> 15) & 1))) * T + ((w >> 7) & 255);
    };
    JacQ acc = *entry(terms[l]);
    constexpr int LEVELS = 8;  // log2(CIRC_LANES)
    static_assert(CIRC_LANES == 1 << LEVELS, "tree depth");
    const int steps = per_lane - 1 + LEVELS;
#pragma unroll 1
    for (int s = 0; s < steps; s++) {  // one inlined addition serves the accumulation and the tree
        JacQ other;
        bool act, minus = false;
        if (s < per_lane - 1) {
            const uint32_t w = terms[(size_t)(s + 1) * CIRC_LANES + l];
            act = (w >> 17) & 1;
            minus = (w >> 16) & 1;
            if (act) other = *entry(w);
        } else {
            const int span = (CIRC_LANES / 2) >> (s - (per_lane - 1));
            part[l] = acc;
            __syncthreads();
            act = l < span;
            if (act) other = part[l + span];
            __syncthreads();
        }
        if (act) acc = add(acc, other, minus);
    }
    if (l == 0) X[(size_t)(__brev((unsigned)k) >> 25) * stride + b] = acc;  // proofs leave in bit-reversed order
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1circ() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_g1_dbl_table));
}
size_t g1_circ_table_bytes(int n, int T) { return (size_t)n * N_CELLS * 2 * T * sizeof(JacQ); }
// X: [128][stride] MSM outputs (natural order) -> X: proofs (bit-reversed), for blobs 0 .. n-1
void g1_circ128(void* X, int stride, int n, int segs, void* D, int T, const void* terms, int per_lane, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const Fq<1> bt = fq_from_fp(b384);
    k_g1_dbl_table<<<(segs * n * N_CELLS + 63) / 64, 64, 0, st>>>((const JacQ*)X, stride, n, segs, (JacQ*)D, T, bt);
    k_g1_circ_sum<<<dim3(N_CELLS, n), CIRC_LANES, 0, st>>>((const JacQ*)D, T, (const uint32_t*)terms, per_lane, (JacQ*)X, stride);
}
}  // namespace launch
}  // namespace kzg
