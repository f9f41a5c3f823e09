// One-time construction of the fixed-base window tables (context set-up).
#include "kcommon.hpp"
#include "curve29.hpp"
#include "launch.hpp"

namespace kzg {

// Build the window table.  One thread per (base, window): Q = 2^(c*w) * P, entries d*Q for d = 1..2^(c-1),
// normalised to affine with one inversion per thread (Montgomery's trick over the thread's entries).
// bases: [n_groups][nb] affine.  scratch: one Fp per table entry (prefix products of the Z's).
template <int C>
__global__ __launch_bounds__(64) void k_build_table(const G1Affine* __restrict__ bases, AffQ* __restrict__ table,
                                                    G1Jac* __restrict__ scratch, int n_groups, int nb) {
    constexpr int W = (255 + C) / C;
    constexpr int T = 1 << (C - 1);
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n_groups * nb * W;
    if (t >= total) return;
    int w = (int)(t % W);
    long bi = t / W;
    int i = (int)(bi % nb), group = (int)(bi / nb);
    G1Affine P = bases[bi];
    AffQ* dst = table + ((((size_t)group * W + w) * nb + i) << (C - 1));
    G1Jac* scr = scratch + ((((size_t)group * W + w) * nb + i) << (C - 1));
    if (is_inf(P)) {
        for (int d = 0; d < T; d++) dst[d] = affq_from_affine(aff_inf());
        return;
    }
    G1Jac Q = to_jac(P);
    for (int k = 0; k < C * w; k++) Q = dbl(Q);
    G1Affine Qa = to_affine(Q);
    // pass 1: Jacobian multiples d*Q (never the identity: d < r, Q != O)
    G1Jac cur = to_jac(Qa);
    for (int d = 0; d < T; d++) {
        scr[d] = cur;
        cur = add_mixed(cur, Qa);
    }
    // pass 2: prefix products of the Z's, parked in the x slot of the destination entries
    Fp prod = one<FpParams>();
    for (int d = 0; d < T; d++) {
        *reinterpret_cast<Fp*>(&dst[d]) = prod;  // parked in the (larger) destination slot until the back sweep
        prod = mul(prod, scr[d].z);
    }
    Fp invp = inv_fast(prod);
    for (int d = T - 1; d >= 0; d--) {
        Fp zi = mul(invp, *reinterpret_cast<const Fp*>(&dst[d]));
        invp = mul(invp, scr[d].z);
        Fp zi2 = sqr(zi);
        G1Affine a;
        a.x = mul(scr[d].x, zi2);
        a.y = mul(scr[d].y, mul(zi2, zi));
        dst[d] = affq_from_affine(a);  // canonical, Montgomery-406, 14 x 29-bit limbs
    }
}

namespace launch {
size_t table_entries(int c, int n_groups, int nb) {
    int W = (255 + c) / c;
    return ((size_t)n_groups * nb * W) << (c - 1);
}
void build_table(int c, const void* bases, void* table, void* scratch, int n_groups, int nb, hipStream_t st) {
    int W = (255 + c) / c;
    long threads = (long)n_groups * nb * W;
    unsigned blocks = (unsigned)((threads + 63) / 64);
    if (c == 8) k_build_table<8><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
    else if (c == 12) k_build_table<12><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
    else if (c == 13) k_build_table<13><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
    else if (c == 14) k_build_table<14><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
    else if (c == 10) k_build_table<10><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
    else k_build_table<4><<<blocks, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)table, (G1Jac*)scratch, n_groups, nb);
}
}  // namespace launch
}  // namespace kzg
