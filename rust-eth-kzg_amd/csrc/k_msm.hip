// Fixed-base MSM over window tables (stage D of compute_cells_and_kzg_proofs; commitment MSM).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "g1_coop.hpp"
#include "launch.hpp"
#include "glv.hpp"
#include <stdexcept>

namespace kzg {

// The running sums are kept in XYZZ coordinates (curve29.hpp: 6M + 2S + one fused product pair per gathered entry,
// 7 % cheaper than the Jacobian mixed addition) and converted to Jacobian once, for the fold.
using MsmAcc = XyzzQ;
__device__ __forceinline__ MsmAcc msm_acc_inf() { return xyzz_inf(); }
__device__ __forceinline__ JacQ msm_acc_to_jacq(const MsmAcc& a) { return to_jacq(a); }

// ------------------------------------------------------------------------------------------------
// Fixed-base MSM with window tables (replaces FixedBaseMSMPrecompWindow::msm,
// fixed_base_msm_window.rs:102-168, and its batched affine adder batch_addition.rs:142-232).
// The table holds, for every base P and every window w, the multiples d * 2^(c*w) * P, d = 1..2^(c-1),
// so an MSM is a pure sum of table entries selected by the signed Booth digits of the scalars
// (booth_encoding.rs:4-46): no doublings at run time.
//   table index inside a group's block: (((w * NB + i) << (c-1)) + (|d| - 1); the groups' blocks are reached through a device
//   array of pointers (launch::TabBlocks: blocks[group]) -- the table is allocated and published piece by piece -- and
//   every kernel takes a group range [g0, g0 + gcnt) of the n_groups MSMs per slice
// Thread (m, w) accumulates the NB entries of MSM m = (slice, group) for window w; the W partial
// sums of an MSM sit in adjacent lanes and are folded through LDS.
// scalars: [msm][NB] canonical Fr.  out[(perm(group)) * out_stride + slice] Jacobian.
// Arithmetic in the unsaturated 14 x 29-bit field (fp29.hpp): table entries are AffQ padded to one 128-B line (TabQ), running sums XyzzQ
// (224 B, registers only), folded and stored sums JacQ (168 B).
__device__ __forceinline__ int booth_digit(const uint32_t* sc, int w, int c) {
    // (c+1)-bit window starting one bit below c*w; window 0 is padded with a zero bit
    int lo = c * w - 1;
    uint32_t x;
    if (w == 0) x = sc[0] << 1;
    else {
        int word = lo >> 5, sh = lo & 31;
        uint64_t two = sc[word];
        if (word + 1 < 8) two |= (uint64_t)sc[word + 1] << 32;
        x = (uint32_t)(two >> sh);
    }
    x &= (1u << (c + 1)) - 1;
    int t = (int)((x + 1) >> 1);
    return (x >> c) ? t - (1 << c) : t;
}

template <int C>
__global__ __launch_bounds__(256, 2) void k_msm_fixed(const Fr* __restrict__ scalars, launch::TabBlocks table,
                                                   JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                   int out_stride, int brp_bits) {
    constexpr int W = (255 + C) / C;  // number of Booth windows
    constexpr int PER_BLOCK = 256 / W;
    __shared__ JacQ red[PER_BLOCK * W];
    const int tid = threadIdx.x;
    const int local = tid / W, w = tid % W;
    const long q_lin = (long)blockIdx.x * PER_BLOCK + local;  // linear index, GROUP-major: consecutive lanes / blocks are slices (blobs) of one group -- the same table rows
    const long m = (q_lin % n_slices) * (long)n_groups + table.g0 + q_lin / n_slices;  // MSM index = slice * n_groups + group (the scalars' layout)
    const long total = (long)table.gcnt * n_slices;
    const bool active = local < PER_BLOCK && q_lin < total;
    MsmAcc acc = msm_acc_inf();
    int group = 0, slice = 0;
    if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const Fr* sc = scalars + (size_t)m * nb;
        const TabQ* tb = reinterpret_cast<const TabQ*>(table.blocks[group]) + ((size_t)w * nb << (C - 1));
        for (int i = 0; i < nb; i++) {
            int d = booth_digit(sc[i].v, w, C);
            if (d != 0) {
                int ad = d < 0 ? -d : d;
                const AffQ p = tb[((size_t)i << (C - 1)) + (ad - 1)].a;
                acc = add_mixed(acc, p, d < 0);
            }
        }
    }
    if (local < PER_BLOCK) red[local * W + w] = msm_acc_to_jacq(acc);
    __syncthreads();
    // fold the W window sums of each MSM (W is not a power of two in general)
    for (int span = 1; span < W; span <<= 1) {
        if (local < PER_BLOCK && (w % (2 * span)) == 0 && w + span < W) {
            red[local * W + w] = add(red[local * W + w], red[local * W + w + span]);
        }
        __syncthreads();
    }
    if (active && w == 0) {
        int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = red[local * W];
    }
}

// Large-batch variant: a thread runs a CHUNK of the windows of one MSM -- windows [s * Wc, (s+1) * Wc) of all nb bases,
// Wc = ceil(W / S) -- into one running sum; S = 1, 2 or 4 adjacent lanes share an MSM and fold through LDS.
// The windowed kernel above (S = W) folds the W window sums of every MSM with a five-level tree in which most lanes
// idle: five full Jacobian additions per wave on top of 64 mixed ones (12 % of its instructions), plus a conversion
// per window sum.  With S = 4 the fold is two additions per 304 (1 %), with S = 1 there is none; the price is fewer,
// longer-running threads, so the engine picks the smallest S that still fills the chip (engine.hip: launch_msm).
// The next table entry is requested one addition ahead (128 B into registers), so the random gathers from the
// 160 GB table are in flight during the ~4.5 k instructions of the current addition.
struct TabLine { uint4 q[7]; };
__device__ __forceinline__ TabLine load_line(const TabQ* p) {
    TabLine l;
    const uint4* s = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 7; i++) l.q[i] = s[i];
    return l;
}
__device__ __forceinline__ AffQ line_to_affq(const TabLine& l) {
    AffQ a;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const uint32_t w[4] = {l.q[i].x, l.q[i].y, l.q[i].z, l.q[i].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = 4 * i + j;
            if (t < QL) a.x.v[t] = w[j];
            else a.y.v[t - QL] = w[j];
        }
    }
    return a;
}
__device__ __forceinline__ Fr shl1(const Fr& a) {
    Fr r;
#pragma unroll
    for (int k = 7; k > 0; k--) r.v[k] = (a.v[k] << 1) | (a.v[k - 1] >> 31);
    r.v[0] = a.v[0] << 1;
    return r;
}
// signed Booth digit of the c+1 low bits; then s >>= c
template <int C>
__device__ __forceinline__ int take_booth(Fr& s) {
    const uint32_t x = s.v[0] & ((1u << (C + 1)) - 1);
#pragma unroll
    for (int k = 0; k < 7; k++) s.v[k] = (s.v[k] >> C) | (s.v[k + 1] << (32 - C));
    s.v[7] >>= C;
    const int t = (int)((x + 1) >> 1);
    return (x >> C) ? t - (1 << C) : t;
}
template <int C>
__global__ __launch_bounds__(256, 2) void k_msm_fixed_chunked(const Fr* __restrict__ scalars, launch::TabBlocks table,
                                                              JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                              int out_stride, int brp_bits, int S) {
    constexpr int W = (255 + C) / C;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x;
    const int chunk = tid & (S - 1);                                      // S is a power of two
    const long q_lin = ((long)blockIdx.x * 256 + tid) / S;  // linear index, GROUP-major: consecutive lanes / blocks are slices (blobs) of one group -- the same table rows
    const long m = (q_lin % n_slices) * (long)n_groups + table.g0 + q_lin / n_slices;  // MSM index = slice * n_groups + group (the scalars' layout)
    const bool active = q_lin < (long)table.gcnt * n_slices;
    const int Wc = (W + S - 1) / S;
    const int w0 = chunk * Wc, nw = (w0 + Wc <= W ? Wc : W - w0);        // this thread's windows [w0, w0 + nw); nw may be <= 0
    int slice = 0, group = 0;
    MsmAcc acc = msm_acc_inf();
    if (active && nw > 0) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const Fr* sc = scalars + (size_t)m * nb;
        const TabQ* tb = reinterpret_cast<const TabQ*>(table.blocks[group]) + (((size_t)w0 * nb) << (C - 1));  // entry (w, i, a): tb[(((w - w0) * nb + i) << (C-1)) + a]
        // The scalar sits in registers shifted left by one bit (window 0 is padded with a zero bit,
        // booth_encoding.rs:4-46; r < 2^255 so nothing is lost); each window reads its c+1 low bits and shifts the
        // scalar right by c: static register indexing only.
        auto fetch = [&](int i) {
            Fr t = shl1(sc[i]);
            for (int k = 0; k < w0; k++) (void)take_booth<C>(t);
            return t;
        };
        Fr s = fetch(0);
        int i = 0, w = 0;
        int d = take_booth<C>(s);
        TabLine cur = load_line(tb + (d ? (d < 0 ? -d : d) - 1 : 0));
        const int total = nw * nb;
#pragma unroll 1
        for (int e = 0; e < total; e++) {
            // address of the next entry, requested before the current addition
            int w2 = w + 1, i2 = i;
            if (w2 == nw) {
                w2 = 0;
                i2 = i + 1;
                s = fetch(i2 < nb ? i2 : 0);
            }
            const int d2 = i2 < nb ? take_booth<C>(s) : 0;
            const int a2 = d2 ? (d2 < 0 ? -d2 : d2) - 1 : 0;
            const TabLine nxt = load_line(tb + ((((size_t)w2 * nb + (i2 < nb ? i2 : 0)) << (C - 1)) + a2));
            if (d != 0) acc = add_mixed(acc, line_to_affq(cur), d < 0);
            cur = nxt;
            d = d2;
            w = w2;
            i = i2;
        }
    } else if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
    }
    JacQ sum = msm_acc_to_jacq(acc);
    if (S > 1) {  // fold the S chunk sums of each MSM (adjacent lanes)
        for (int span = 1; span < S; span <<= 1) {
            red[tid] = sum;
            __syncthreads();
            if ((chunk & (2 * span - 1)) == 0) sum = add(sum, red[tid + span]);
            __syncthreads();
        }
    }
    if (active && chunk == 0) {
        const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = sum;
    }
}

// Small-batch variant (a handful of blobs): one block per MSM, the W * nb table entries of the sum dealt round-robin
// to all 256 lanes (5 additions each for W = 19, nb = 64) and folded by an 8-level tree in LDS.  ~30 % more field
// work than the kernel above, but the dependent chain drops from 64 + 5 additions to 5 + 8.
template <int C>
__global__ __launch_bounds__(256) void k_msm_fixed_flat(const Fr* __restrict__ scalars, launch::TabBlocks table,
                                                        JacQ* __restrict__ out, int n_groups, int nb, int out_stride,
                                                        int brp_bits) {
    constexpr int W = (255 + C) / C;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x;
    const int slice = (int)(blockIdx.x / table.gcnt), group = table.g0 + (int)(blockIdx.x % table.gcnt);
    const long m = (long)slice * n_groups + group;  // MSM index = slice * n_groups + group
    const TabQ* tb = reinterpret_cast<const TabQ*>(table.blocks[group]);
    const Fr* sc = scalars + (size_t)m * nb;
    MsmAcc xacc = msm_acc_inf();
    for (int e = tid; e < W * nb; e += 256) {
        const int w = e / nb, i = e - w * nb;
        const int d = booth_digit(sc[i].v, w, C);
        if (d != 0) {
            const int ad = d < 0 ? -d : d;
            const AffQ p = tb[(((size_t)w * nb + i) << (C - 1)) + (ad - 1)].a;
            xacc = add_mixed(xacc, p, d < 0);
        }
    }
    red[tid] = msm_acc_to_jacq(xacc);
    coop_tree_fold<256>(red, 128, tid);  // the tree's idle lanes share its additions (g1_coop.hpp)
    if (tid == 0) {
        const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = red[0];
    }
}

__global__ void k_glv_split(Fr* __restrict__ scalars, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = scalars[i];
    uint32_t out[8];
    glv_split_balanced(k, out);
    Fr o;
#pragma unroll
    for (int l = 0; l < 8; l++) o.v[l] = out[l];
    scalars[i] = o;
}
namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_msm() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_glv_split));
}
template <int C>
static void msm_flat_c(const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                       int brp_bits, hipStream_t st) {
    if (table.gcnt <= 0 || n_slices <= 0) return;
    k_msm_fixed_flat<C><<<(unsigned)(table.gcnt * n_slices), 256, 0, st>>>((const Fr*)scalars, table, (JacQ*)out,
                                                                          n_groups, nb, out_stride, brp_bits);
}
void msm_fixed_flat(int c, const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                    int brp_bits, hipStream_t st) {
    if (c != PLAIN_WIDTH) throw std::runtime_error("plain window tables exist at width 4 only");
    msm_flat_c<PLAIN_WIDTH>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
}
// k_msm_glv.inc (one translation unit per window width) holds the GLV kernels; the split of the scalars is shared
void glv_split(void* scalars, size_t n, hipStream_t st) {
    k_glv_split<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((Fr*)scalars, n);
}
#define GLV_DECL(w) void msm_glv_w##w(int, const void*, const TabBlocks&, void*, int, int, int, int, int, const Fp12w&, hipStream_t); void preload_k_msm_glv##w();
GLV_DECL(8) GLV_DECL(12) GLV_DECL(14) GLV_DECL(15) GLV_DECL(16)
#undef GLV_DECL
void preload_k_ntt(); void preload_k_g1fft(); void preload_k_g1circ(); void preload_k_g1misc(); void preload_k_verify();
void preload_k_verify_many(); void preload_k_4844(); void preload_k_g1slp(); void preload_k_table();
void preload_code_objects() {
    preload_k_msm(); preload_k_msm_glv8(); preload_k_msm_glv12(); preload_k_msm_glv14(); preload_k_msm_glv15(); preload_k_msm_glv16();
    preload_k_ntt(); preload_k_g1fft(); preload_k_g1circ(); preload_k_g1misc(); preload_k_verify(); preload_k_verify_many();
    preload_k_4844(); preload_k_g1slp(); preload_k_table();
}
bool glv_width_supported(int c) {
    for (int w : GLV_WIDTHS)
        if (w == c) return true;
    return false;
}
void msm_glv(int c, int mode, const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
             int brp_bits, const Fp12w& beta, hipStream_t st) {
    switch (c) {
        case 8: return msm_glv_w8(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st);
        case 12: return msm_glv_w12(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st);
        case 14: return msm_glv_w14(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st);
        case 15: return msm_glv_w15(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st);
        default: return msm_glv_w16(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st);
    }
}
template <int C>
static void msm_chunked_c(const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                          int brp_bits, int S, hipStream_t st) {
    const long threads = (long)table.gcnt * n_slices * S;
    if (threads <= 0) return;
    k_msm_fixed_chunked<C><<<(unsigned)((threads + 255) / 256), 256, 0, st>>>((const Fr*)scalars, table, (JacQ*)out,
                                                                               n_groups, n_slices, nb, out_stride, brp_bits, S);
}
void msm_fixed_chunked(int c, const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                       int brp_bits, int S, hipStream_t st) {
    if (c != PLAIN_WIDTH) throw std::runtime_error("plain window tables exist at width 4 only");
    msm_chunked_c<PLAIN_WIDTH>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
}
template <int C>
static void msm_c(const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                  int brp_bits, hipStream_t st) {
    constexpr int PB = 256 / ((255 + C) / C);
    long total = (long)table.gcnt * n_slices;
    if (total <= 0) return;
    k_msm_fixed<C><<<(unsigned)((total + PB - 1) / PB), 256, 0, st>>>((const Fr*)scalars, table, (JacQ*)out,
                                                                     n_groups, n_slices, nb, out_stride, brp_bits);
}
void msm_fixed(int c, const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
               int brp_bits, hipStream_t st) {
    if (c != PLAIN_WIDTH) throw std::runtime_error("plain window tables exist at width 4 only");
    msm_c<PLAIN_WIDTH>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
}
}  // namespace launch
}  // namespace kzg
