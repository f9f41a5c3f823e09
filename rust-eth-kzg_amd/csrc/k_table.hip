// One-time construction of the fixed-base window tables (context set-up).
#include "kcommon.hpp"
#include "curve30.hpp"
#include "launch.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------------
// The builder of the GLV tables (2^(bits-1) entries per (base, window) is a multiple of 64): two kernels in the unsaturated field.
//  k_table_windows : thread per base: Q_w = 2^(first bit of window w) P and 64 Q_w for every window, normalised to affine with one inversion.
//  k_table_fill_packed : ONE WAVE per (base, window).  Lane l starts at (l + 1) Q and steps by 64 Q, so at step k the wave
//                    holds the 64 consecutive entries d = 64 k + l + 1: every load and store is a contiguous 7 KiB run
//                    (the thread-per-(base, window) builder above strides each lane through its own 0.9 MB region and is
//                    bound by address translation, not arithmetic).  Normalisation without a per-entry inversion: along a
//                    lane Z_(k+1) = Z_k * f_k with f_k = 2 H_k from the mixed addition, so 1 / Z_k = (1 / Z_(k+1)) * f_k:
//                    one binary-GCD inversion per lane per 2^(c-1) / 64 entries, one multiplication per entry on the way back.
//                    X, Y (2 x 56 B) and f_k (56 B) wait in a scratch of 168 B per entry of the chunk being built.
// p + q with the factor f = Z3 / Z1 (2 H in general; 2 Y1 when p == q and the sum is a doubling)
__device__ __forceinline__ JacQ add_mixed_f(const JacQ& p, const AffQ& q, Fq<260>& f) {
    Fq<2> z1z1 = sqr(p.z);
    Fq<2> u2 = mul(q.x, z1z1);
    Fq<2> s2p = mul(mul(q.y, p.z), z1z1);
    auto h = sub(u2, p.x);
    auto rr = dbl(signed_sub(false, s2p, p.y));
    Fq<2> hh = sqr(h);
    Fq<8> i = dbl2(hh);
    Fq<2> j = mul(h, i);
    Fq<2> v = mul(p.x, i);
    JacQ r;
    auto x3 = sub_sub2(sqr(rr), j, v);
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(v, x3), neg2(p.y), j));
    const Fq<2> zh = mul(p.z, h);
    r.z = dbl(zh);
    f = relax<260>(dbl(h));
    if (product_is_zero(zh)) {  // same x: the table only ever meets p == q here (d Q + 64 Q with d = 64)
        r = dbl(p);
        f = relax<260>(dbl(p.y));
    }
    return r;
}

// the windows of a GLV table have mixed widths (launch::glv_window_bits)
template <int C, int W>
__global__ void k_table_windows(const G1Affine* __restrict__ bases, AffQ* __restrict__ qw /*[n][2][W]*/, JacQ* __restrict__ tmp,
                                Fq<2>* __restrict__ pre, int n_bases) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bases) return;
    AffQ* out = qw + (size_t)b * 2 * W;
    const AffQ P = affq_from_affine(bases[b]);
    if (is_inf(P)) {
        for (int i = 0; i < 2 * W; i++) out[i] = P;
        return;
    }
    JacQ* J = tmp + (size_t)b * 2 * W;
    Fq<2>* pr = pre + (size_t)b * 2 * W;
    JacQ cur = to_jacq(P);
    for (int w = 0; w < W; w++) {
        J[w] = cur;
        const int bits = launch::glv_window_bits(C, w);
        for (int s = 0; s < bits; s++) {
            cur = dbl(cur);
            if (s == 5) J[W + w] = cur;  // 64 Q_w
        }
    }
    Fq<2> prod = relax<2>(fq_one());
    for (int i = 0; i < 2 * W; i++) { pr[i] = prod; prod = mul(prod, J[i].z); }
    Fq<2> inv = relax<2>(fq_inv(prod));
    for (int i = 2 * W - 1; i >= 0; i--) {
        const JacQ p = J[i];
        const Fq<2> zi = mul(inv, pr[i]);
        inv = mul(inv, p.z);
        const Fq<2> zi2 = sqr(zi);
        AffQ a;
        a.x = reduce_once(mul(p.x, zi2));
        a.y = reduce_once(mul(p.y, mul(zi2, zi)));
        out[i] = a;
    }
}

// GLV table (curve30.hpp: TabS): the scalars are split k = k1 + k2 lambda with |k1|, |k2| < 2^127 and phi is applied to the
// SUM of the k2 terms (phi is a homomorphism), so one table over W = 8 windows of c = 16 bits serves both halves: 16
// gathered additions per base instead of 19 at width 14 -- and half the memory per window, which is what lets the
// window be 16 bits wide at all.  Entries are packed canonical coordinates (2 x 48 B): at 128 B they would not fit in HBM.
// The walk runs in the 14 x 29-bit field; an entry is stored in the form the MSM kernels compute in -- the signed 13 x 30-bit
// field's Montgomery-390 value as exact centred digits (one product by 2^-16 and the digit recentring per coordinate).
// Same wave-per-(base, window) walk as k_table_fill; the un-normalised X, Y wait in a scratch (the 96-B entry cannot hold them).
template <int C, int W>
__global__ __launch_bounds__(64) void k_table_fill_packed(const AffQ* __restrict__ qw, void* const* __restrict__ blocks,
                                                          Fq<260>* __restrict__ scratch_f, Fq<XB>* __restrict__ scratch_xy,
                                                          int nb, int* __restrict__ err) {
    const int lane = threadIdx.x;
    const long blk = blockIdx.x;  // = (group * W + w) * nb + i : the table's own block order
    const int i = (int)(blk % nb), w = (int)((blk / nb) % W);
    const long group = blk / ((long)nb * W);
    const int T = 1 << (launch::glv_window_bits(C, w) - 1), K = T / 64;  // entries of this window per base (mixed widths)
    const long base = group * nb + i;
    const AffQ Q = qw[(size_t)base * 2 * W + w], S = qw[(size_t)base * 2 * W + W + w];
    // a group's lower WL = ceil(W / 2) windows and its upper W - WL windows are two blocks (k_msm_glv.inc: tab_window)
    constexpr int WL = (W + 1) / 2;
    const int upper = w >= WL ? 1 : 0;
    // inside a block: [window][base][digit]; the scratch follows the table's own order: [group][window][base][digit]
    const size_t in_block = launch::glv_entries_per_base(C, upper ? WL : 0, w) * (size_t)nb + (size_t)i * T;
    TabS* dst = reinterpret_cast<TabS*>(blocks[2 * group + upper]) + in_block;
    const size_t in_table = ((size_t)group * launch::glv_entries_per_base(C, 0, W) + launch::glv_entries_per_base(C, 0, w)) * (size_t)nb + (size_t)i * T;
    Fq<260>* scr = scratch_f + in_table;
    Fq<XB>* raw = scratch_xy + 2 * in_table;  // 2 per entry
    if (is_inf(Q)) {  // identity base (wave-uniform): an all-identity block
        TabS z;
        for (int t = 0; t < 24; t++) z.w[t] = 0;
        for (int k = 0; k < K; k++) dst[k * 64 + lane] = z;
        return;
    }
    JacQ cur = jacq_inf();
    const int n = lane + 1;
#pragma unroll 1
    for (int bit = 6; bit >= 0; bit--) {
        cur = dbl(cur);
        const JacQ t = add_mixed(cur, Q);
        const bool take = (n >> bit) & 1;
        cur.x = select(take, t.x, cur.x);
        cur.y = select(take, t.y, cur.y);
        cur.z = select(take, t.z, cur.z);
    }
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        raw[2 * (size_t)(k * 64 + lane)] = cur.x;
        raw[2 * (size_t)(k * 64 + lane) + 1] = cur.y;
        if (k + 1 < K) {
            Fq<260> f;
            cur = add_mixed_f(cur, S, f);
            scr[k * 64 + lane] = f;
        }
    }
    if (is_inf(cur)) atomicOr(err, 1);  // cannot happen for a base of prime order
    Fq<2> zinv = relax<2>(fq_inv(cur.z));
#pragma unroll 1
    for (int k = K - 1; k >= 0; k--) {
        const Fq<XB> X = raw[2 * (size_t)(k * 64 + lane)], Y = raw[2 * (size_t)(k * 64 + lane) + 1];
        const Fq<2> zi2 = sqr(zinv);
        const Fq<1> ax = reduce_once(mul(X, zi2)), ay = reduce_once(mul(Y, mul(zi2, zinv)));
        TabS e;
        tabs_pack_from_fq(e.w, ax);
        tabs_pack_from_fq(e.w + 12, ay);
        dst[k * 64 + lane] = e;
        if (k > 0) zinv = mul(zinv, scr[(k - 1) * 64 + lane]);
    }
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_table() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>((&k_table_windows<8, glv_windows(8)>)));
}
// GLV tables: W = glv_windows(c) windows of mixed widths (launch::glv_window_bits), packed 96-B entries; scratch = 168 B per entry of the chunk (56 for the
// Z factors, 112 for the waiting X, Y); side = the per-base window points (affine, Jacobian, prefix products)
size_t table_glv_entries(int c, int n_groups, int nb) { return (size_t)n_groups * nb * glv_entries_per_base(c, 0, glv_windows(c)); }
size_t table_glv_side_bytes(int c, int n_groups, int nb) {
    const size_t n = (size_t)n_groups * nb;
    return n * 2 * glv_windows(c) * (SIZEOF_AFFQ + SIZEOF_JACQ + 56) + 256;
}
template <int C>
static void table_glv_c(const void* bases, void* const* table, void* scratch, void* side, int n_groups, int nb, int* err, hipStream_t st) {
    constexpr int W = glv_windows(C);
    const size_t n = (size_t)n_groups * nb, entries = table_glv_entries(C, n_groups, nb);
    char* qw = (char*)side;
    char* tmp = qw + n * 2 * W * SIZEOF_AFFQ;
    char* pre = tmp + n * 2 * W * SIZEOF_JACQ;
    char* scr_f = (char*)scratch;
    char* scr_xy = scr_f + entries * 56;
    k_table_windows<C, W><<<((int)n + 63) / 64, 64, 0, st>>>((const G1Affine*)bases, (AffQ*)qw, (JacQ*)tmp, (Fq<2>*)pre, (int)n);
    k_table_fill_packed<C, W><<<(unsigned)(n * W), 64, 0, st>>>((const AffQ*)qw, table, (Fq<260>*)scr_f, (Fq<XB>*)scr_xy, nb, err);
}
bool build_table_glv(int c, const void* bases, void* const* table, void* scratch, void* side, int n_groups, int nb, int* err, hipStream_t st) {
    if (c == 16) table_glv_c<16>(bases, table, scratch, side, n_groups, nb, err, st);
    else if (c == 15) table_glv_c<15>(bases, table, scratch, side, n_groups, nb, err, st);
    else if (c == 14) table_glv_c<14>(bases, table, scratch, side, n_groups, nb, err, st);
    else if (c == 12) table_glv_c<12>(bases, table, scratch, side, n_groups, nb, err, st);
    else if (c == 8) table_glv_c<8>(bases, table, scratch, side, n_groups, nb, err, st);
    else return false;
    return true;
}
}  // namespace launch
}  // namespace kzg
