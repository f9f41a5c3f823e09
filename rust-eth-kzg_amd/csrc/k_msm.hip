// Fixed-base MSM over window tables (stage D of compute_cells_and_kzg_proofs; commitment MSM).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "launch.hpp"
#include "glv.hpp"

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
//   table index: (((group * W + w) * NB + i) << (c-1)) + (|d| - 1)
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
__global__ __launch_bounds__(256, 2) void k_msm_fixed(const Fr* __restrict__ scalars, const TabQ* __restrict__ table,
                                                   JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                   int out_stride, int brp_bits) {
    constexpr int W = (255 + C) / C;  // number of Booth windows
    constexpr int PER_BLOCK = 256 / W;
    __shared__ JacQ red[PER_BLOCK * W];
    const int tid = threadIdx.x;
    const int local = tid / W, w = tid % W;
    const long m = (long)blockIdx.x * PER_BLOCK + local;  // MSM index = slice * n_groups + group
    const long total = (long)n_groups * n_slices;
    const bool active = local < PER_BLOCK && m < total;
    MsmAcc acc = msm_acc_inf();
    int group = 0, slice = 0;
    if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const Fr* sc = scalars + (size_t)m * nb;
        const TabQ* tb = table + (((size_t)group * W + w) * nb << (C - 1));
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
__global__ __launch_bounds__(256, 2) void k_msm_fixed_chunked(const Fr* __restrict__ scalars, const TabQ* __restrict__ table,
                                                              JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                              int out_stride, int brp_bits, int S) {
    constexpr int W = (255 + C) / C;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x;
    const int chunk = tid & (S - 1);                                      // S is a power of two
    const long m = ((long)blockIdx.x * 256 + tid) / S;                    // MSM index = slice * n_groups + group
    const bool active = m < (long)n_groups * n_slices;
    const int Wc = (W + S - 1) / S;
    const int w0 = chunk * Wc, nw = (w0 + Wc <= W ? Wc : W - w0);        // this thread's windows [w0, w0 + nw); nw may be <= 0
    int slice = 0, group = 0;
    MsmAcc acc = msm_acc_inf();
    if (active && nw > 0) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const Fr* sc = scalars + (size_t)m * nb;
        const TabQ* tb = table + ((((size_t)group * W + w0) * nb) << (C - 1));  // entry (w, i, a): tb[(((w - w0) * nb + i) << (C-1)) + a]
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
__global__ __launch_bounds__(256) void k_msm_fixed_flat(const Fr* __restrict__ scalars, const TabQ* __restrict__ table,
                                                        JacQ* __restrict__ out, int n_groups, int nb, int out_stride,
                                                        int brp_bits) {
    constexpr int W = (255 + C) / C;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x;
    const long m = blockIdx.x;  // MSM index = slice * n_groups + group
    const int slice = (int)(m / n_groups), group = (int)(m % n_groups);
    const Fr* sc = scalars + (size_t)m * nb;
    MsmAcc xacc = msm_acc_inf();
    for (int e = tid; e < W * nb; e += 256) {
        const int w = e / nb, i = e - w * nb;
        const int d = booth_digit(sc[i].v, w, C);
        if (d != 0) {
            const int ad = d < 0 ? -d : d;
            const AffQ p = table[((((size_t)group * W + w) * nb + i) << (C - 1)) + (ad - 1)].a;
            xacc = add_mixed(xacc, p, d < 0);
        }
    }
    JacQ acc = msm_acc_to_jacq(xacc);
#pragma unroll 1
    for (int span = 128; span >= 1; span >>= 1) {
        red[tid] = acc;
        __syncthreads();
        if (tid < span) acc = add(acc, red[tid + span]);
        __syncthreads();
    }
    if (tid == 0) {
        const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// GLV form of the three kernels above, over the packed 8-window table of width 16 (k_table.hip: k_table_fill_packed).
// Scalars come balanced-split in place (k_glv_split): 32 bytes = |k1| (sign in bit 127) | |k2| (sign in bit 127).
// Joint window u < 16: u < 8 is window u of k1, u >= 8 window u - 8 of k2; both read the SAME table rows, and phi is
// applied once to the sum of the k2 terms (phi is a homomorphism: one multiplication of X by beta per partial sum
// instead of one per gathered entry).  16 gathered additions per base instead of 19.
struct GlvScalar { uint32_t h[2][4]; };
static_assert(sizeof(GlvScalar) == sizeof(Fr), "split in place");
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
struct PackLine { uint4 q[6]; };
__device__ __forceinline__ PackLine load_pack(const TabP* p) {
    PackLine l;
    const uint4* s = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 6; i++) l.q[i] = s[i];
    return l;
}
__device__ __forceinline__ AffQ pack_to_affq(const PackLine& l) {
    uint32_t w[24];
#pragma unroll
    for (int i = 0; i < 6; i++) { w[4 * i] = l.q[i].x; w[4 * i + 1] = l.q[i].y; w[4 * i + 2] = l.q[i].z; w[4 * i + 3] = l.q[i].w; }
    AffQ a;
    regroup_32_to_29(a.x.v, w);
    regroup_32_to_29(a.y.v, w + 12);
    return a;
}
// signed Booth digit of window w (< 8) of a 128-bit magnitude read from memory (sign bit 127 ignored); 16-bit windows
__device__ __forceinline__ int booth16_mem(const uint32_t* m, int w) {
    uint32_t x;
    if (w == 0) x = (m[0] << 1) & 0x1ffffu;
    else {
        const int lo = 16 * w - 1, word = lo >> 5, sh = lo & 31;
        uint64_t two = m[word];
        if (word + 1 < 4) two |= (uint64_t)(word + 1 == 3 ? (m[3] & 0x7fffffffu) : m[word + 1]) << 32;
        else two &= 0x7fffffffu;
        x = (uint32_t)(two >> sh) & 0x1ffffu;
    }
    const int t = (int)((x + 1) >> 1);
    return (x >> 16) ? t - 65536 : t;
}
__device__ __forceinline__ JacQ apply_phi(const JacQ& p, const Fq<1>& beta) {
    JacQ r = p;
    r.x = relax<XB>(mul(p.x, beta));
    return r;
}

// large batches: block = 64 MSMs x 4 chunks, chunk = wave index: k1 windows 0-3, k1 windows 4-7, k2 windows 0-3, k2 windows 4-7
__global__ __launch_bounds__(256, 2) void k_msm_glv_chunked(const GlvScalar* __restrict__ scalars, const TabP* __restrict__ table,
                                                            JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                            int out_stride, int brp_bits, Fq<1> beta) {
    constexpr int C = launch::GLV_C, W = launch::GLV_W;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x, chunk = tid >> 6, lane = tid & 63;
    const int half = chunk >> 1, w0 = (chunk & 1) * 4;
    const long m = (long)blockIdx.x * 64 + lane;  // MSM index = slice * n_groups + group
    const bool active = m < (long)n_groups * n_slices;
    int slice = 0, group = 0;
    MsmAcc acc = msm_acc_inf();
    if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const GlvScalar* sc = scalars + (size_t)m * nb;
        const TabP* tb = table + ((((size_t)group * W + w0) * nb) << (C - 1));  // entry (w, i, a): tb[(((w - w0) * nb + i) << 15) + a]
        // this chunk's 64 bits of the half, kept as a 65-bit shift register v = (bits << 1) | the bit below
        uint64_t v = 0;
        uint32_t vtop = 0, sneg = 0;
        auto fetch = [&](int i) {
            const uint32_t* h = sc[i].h[half];
            const uint32_t a0 = h[0], a1 = h[1], a2 = h[2], a3 = h[3];
            sneg = a3 >> 31;
            const uint64_t bits = (chunk & 1) ? (((uint64_t)(a3 & 0x7fffffffu) << 32) | a2) : (((uint64_t)a1 << 32) | a0);
            const uint32_t below = (chunk & 1) ? (a1 >> 31) : 0u;
            v = (bits << 1) | below;
            vtop = (uint32_t)(bits >> 63);
        };
        auto take = [&]() {  // signed digit of the low 17 bits, then shift by 16
            const uint32_t x = (uint32_t)v & 0x1ffffu;
            v = (v >> 16) | ((uint64_t)vtop << 48);
            vtop = 0;
            const int t = (int)((x + 1) >> 1);
            return (x >> 16) ? t - 65536 : t;
        };
        fetch(0);
        int i = 0, w = 0;
        int d = take();
        uint32_t dneg = sneg;
        PackLine cur = load_pack(tb + (d ? (d < 0 ? -d : d) - 1 : 0));
        const int total = 4 * nb;
#pragma unroll 1
        for (int e = 0; e < total; e++) {
            int w2 = w + 1, i2 = i;
            const uint32_t dneg_cur = dneg;
            if (w2 == 4) {
                w2 = 0;
                i2 = i + 1;
                fetch(i2 < nb ? i2 : 0);
            }
            const int d2 = i2 < nb ? take() : 0;
            dneg = sneg;
            const int a2 = d2 ? (d2 < 0 ? -d2 : d2) - 1 : 0;
            const PackLine nxt = load_pack(tb + ((((size_t)w2 * nb + (i2 < nb ? i2 : 0)) << (C - 1)) + a2));
            if (d != 0) acc = add_mixed(acc, pack_to_affq(cur), (d < 0) != (dneg_cur != 0));
            cur = nxt;
            d = d2;
            w = w2;
            i = i2;
        }
    }
    // fold: (chunk 0 + chunk 1) + phi(chunk 2 + chunk 3)
    JacQ sum = msm_acc_to_jacq(acc);
    red[tid] = sum;
    __syncthreads();
    if ((chunk & 1) == 0) sum = add(sum, red[tid + 64]);
    if (chunk == 2) sum = apply_phi(sum, beta);
    __syncthreads();
    if (chunk == 2) red[tid] = sum;
    __syncthreads();
    if (chunk == 0) {
        sum = add(sum, red[tid + 128]);
        if (active) {
            const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
            out[(size_t)pos * out_stride + slice] = sum;
        }
    }
}

// Batches that fill the chip for whole rounds: a LANE owns a whole MSM (S = 1) or one of its GLV halves (S = 2).
// S = 1: the lane first sums the 8 x nb entries selected by k2, applies phi to that running sum IN PLACE (phi acts on an
// XYZZ point as X <- beta X), and keeps adding the 8 x nb entries selected by k1 into the same accumulator:
// phi(sum k2 terms) + sum k1 terms with no fold at all -- no LDS, no barrier, no Jacobian addition, one conversion per
// 1024 gathered additions.  2048 blobs are 4096 such waves: exactly two rounds of the chip's 2-per-SIMD wave slots
// (the four-chunk kernel above needs 16384 waves in blocks of four that retire together, and two Jacobian additions,
// two conversions and three barriers per block).  S = 2: wave 0 of a block sums the k2 half and hands phi of it over
// through LDS, wave 1 sums the k1 half and adds: half as long a wave for batches that would leave S = 1's last round
// part empty (engine.hip: launch_msm picks by predicted rounds).
template <int S>
__global__ __launch_bounds__(64 * S, 2) void k_msm_glv_lane(const GlvScalar* __restrict__ scalars, const TabP* __restrict__ table,
                                                            JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                            int out_stride, int brp_bits, Fq<1> beta) {
    constexpr int C = launch::GLV_C, W = launch::GLV_W;
    static_assert(S == 1 || S == 2, "a lane owns an MSM or one GLV half of it");
    static_assert(C == 16 && W == 8, "digit extraction below is written for eight 16-bit windows per half");
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;  // part: 0 = starts with (S = 2: owns) the k2 half
    const long m = (long)blockIdx.x * 64 + lane;  // MSM index = slice * n_groups + group
    const bool active = m < (long)n_groups * n_slices;
    int slice = 0, group = 0;
    MsmAcc acc = msm_acc_inf();
    if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const GlvScalar* sc = scalars + (size_t)m * nb;
        const TabP* tb = table + ((((size_t)group * W) * nb) << (C - 1));  // entry (w, i, a): tb[((w * nb + i) << 15) + a]
        // the 128-bit magnitude of the current half scalar as a shift register; `below` = the bit under the window
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, below = 0, sneg = 0;
        auto fetch = [&](int half, int i) {
            const uint4 h = *reinterpret_cast<const uint4*>(sc[i].h[half]);
            r0 = h.x; r1 = h.y; r2 = h.z;
            sneg = h.w >> 31;
            r3 = h.w & 0x7fffffffu;
            below = 0;
        };
        auto take = [&]() {  // signed Booth digit of the low 16 bits + the bit below, then shift by 16
            const uint32_t x = ((r0 & 0xffffu) << 1) | below;
            below = (r0 >> 15) & 1u;
            r0 = __funnelshift_r(r0, r1, 16);
            r1 = __funnelshift_r(r1, r2, 16);
            r2 = __funnelshift_r(r2, r3, 16);
            r3 >>= 16;
            const int t = (int)((x + 1) >> 1);
            return (x >> 16) ? t - 65536 : t;
        };
        const int per_half = W * nb;
        const int total = (S == 1 ? 2 : 1) * per_half;
        const int first_half = S == 1 ? 1 : 1 - part;  // k2 first (S = 1); S = 2: part 0 -> k2, part 1 -> k1
        fetch(first_half, 0);
        int d = take();
        uint32_t dneg = sneg;
        PackLine cur = load_pack(tb + (d ? (d < 0 ? -d : d) - 1 : 0));
#pragma unroll 1
        for (int e = 0; e < total; e++) {
            // the next entry is requested before the current addition: its digit, sign and address
            const int e2 = e + 1;
            const uint32_t dneg_cur = dneg;
            const int idx2 = e2 >= per_half ? e2 - per_half : e2;  // position inside its half: i2 * W + w2
            const int w2 = idx2 & (W - 1), i2 = idx2 >> 3;
            const bool more = e2 < total;
            if (w2 == 0 && more) fetch(S == 1 ? (e2 >= per_half ? 0 : 1) : first_half, i2);
            const int d2 = more ? take() : 0;
            dneg = sneg;
            const int a2 = d2 ? (d2 < 0 ? -d2 : d2) - 1 : 0;
            const PackLine nxt = load_pack(tb + ((((size_t)w2 * nb + (more ? i2 : 0)) << (C - 1)) + a2));
            if (S == 1 && e == per_half) acc.x = relax<XB>(mul(acc.x, beta));  // phi of the k2 sum, in place; k1 terms follow
            if (d != 0) acc = add_mixed(acc, pack_to_affq(cur), (d < 0) != (dneg_cur != 0));
            cur = nxt;
            d = d2;
        }
    }
    if (S == 1) {
        if (active) {
            const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
            out[(size_t)pos * out_stride + slice] = msm_acc_to_jacq(acc);
        }
    } else {
        __shared__ JacQ red[64];
        if (part == 0) {
            acc.x = relax<XB>(mul(acc.x, beta));
            red[lane] = msm_acc_to_jacq(acc);
        }
        __syncthreads();
        if (part == 1 && active) {
            const JacQ sum = add(msm_acc_to_jacq(acc), red[lane]);
            const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
            out[(size_t)pos * out_stride + slice] = sum;
        }
    }
}

// medium batches: thread = (MSM, joint window u < 16); 16 MSMs per block
__global__ __launch_bounds__(256, 2) void k_msm_glv_windowed(const GlvScalar* __restrict__ scalars, const TabP* __restrict__ table,
                                                             JacQ* __restrict__ out, int n_groups, int n_slices, int nb,
                                                             int out_stride, int brp_bits, Fq<1> beta) {
    constexpr int C = launch::GLV_C, W = launch::GLV_W;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x, local = tid >> 4, u = tid & 15, half = u >> 3, w = u & 7;
    const long m = (long)blockIdx.x * 16 + local;
    const bool active = m < (long)n_groups * n_slices;
    MsmAcc acc = msm_acc_inf();
    int group = 0, slice = 0;
    if (active) {
        slice = (int)(m / n_groups);
        group = (int)(m % n_groups);
        const GlvScalar* sc = scalars + (size_t)m * nb;
        const TabP* tb = table + ((((size_t)group * W + w) * nb) << (C - 1));
        for (int i = 0; i < nb; i++) {
            const uint32_t* h = sc[i].h[half];
            const int d = booth16_mem(h, w);
            if (d != 0) {
                const bool neg = (d < 0) != ((h[3] >> 31) != 0);
                const AffQ p = pack_to_affq(load_pack(tb + (((size_t)i << (C - 1)) + ((d < 0 ? -d : d) - 1))));
                acc = add_mixed(acc, p, neg);
            }
        }
    }
    JacQ sum = msm_acc_to_jacq(acc);
    if (half) sum = apply_phi(sum, beta);
    red[tid] = sum;
    __syncthreads();
    for (int span = 1; span < 16; span <<= 1) {
        if ((u & (2 * span - 1)) == 0) red[tid] = add(red[tid], red[tid + span]);
        __syncthreads();
    }
    if (active && u == 0) {
        const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = red[tid];
    }
}

// a handful of blobs: one block per MSM; lanes 0-127 take the k1 terms, 128-255 the k2 terms (4 entries each), 8-level tree
__global__ __launch_bounds__(256) void k_msm_glv_flat(const GlvScalar* __restrict__ scalars, const TabP* __restrict__ table,
                                                      JacQ* __restrict__ out, int n_groups, int nb, int out_stride, int brp_bits,
                                                      Fq<1> beta) {
    constexpr int C = launch::GLV_C, W = launch::GLV_W;
    __shared__ JacQ red[256];
    const int tid = threadIdx.x, half = tid >> 7, l = tid & 127;
    const long m = blockIdx.x;
    const int slice = (int)(m / n_groups), group = (int)(m % n_groups);
    const GlvScalar* sc = scalars + (size_t)m * nb;
    MsmAcc xacc = msm_acc_inf();
    for (int e = l; e < W * nb; e += 128) {
        const int w = e / nb, i = e - w * nb;
        const uint32_t* h = sc[i].h[half];
        const int d = booth16_mem(h, w);
        if (d != 0) {
            const bool neg = (d < 0) != ((h[3] >> 31) != 0);
            const AffQ p = pack_to_affq(load_pack(table + (((((size_t)group * W + w) * nb + i) << (C - 1)) + ((d < 0 ? -d : d) - 1))));
            xacc = add_mixed(xacc, p, neg);
        }
    }
    JacQ acc = msm_acc_to_jacq(xacc);
    if (half) acc = apply_phi(acc, beta);
#pragma unroll 1
    for (int span = 128; span >= 1; span >>= 1) {
        red[tid] = acc;
        __syncthreads();
        if (tid < span) acc = add(acc, red[tid + span]);
        __syncthreads();
    }
    if (tid == 0) {
        const int pos = brp_bits ? (int)(__brev((unsigned)group) >> (32 - brp_bits)) : group;
        out[(size_t)pos * out_stride + slice] = acc;
    }
}

namespace launch {
template <int C>
static void msm_flat_c(const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                       int brp_bits, hipStream_t st) {
    k_msm_fixed_flat<C><<<(unsigned)(n_groups * n_slices), 256, 0, st>>>((const Fr*)scalars, (const TabQ*)table, (JacQ*)out,
                                                                        n_groups, nb, out_stride, brp_bits);
}
void msm_fixed_flat(int c, const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                    int brp_bits, hipStream_t st) {
    if (c == 8) msm_flat_c<8>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 12) msm_flat_c<12>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 13) msm_flat_c<13>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 14) msm_flat_c<14>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 10) msm_flat_c<10>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else msm_flat_c<4>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
}
// mode: 0 flat (one block per MSM), 1 windowed, 2 four chunks per MSM, 3 a lane per MSM, 4 a lane per GLV half.
// The scalars are split IN PLACE first (they feed nothing else).
void msm_glv16(int mode, void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
               int brp_bits, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const Fq<1> bq = fq_from_fp(b384);
    const size_t n = (size_t)n_groups * n_slices * nb;
    const long msms = (long)n_groups * n_slices;
    k_glv_split<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((Fr*)scalars, n);
    if (mode == 0)
        k_msm_glv_flat<<<(unsigned)msms, 256, 0, st>>>((const GlvScalar*)scalars, (const TabP*)table, (JacQ*)out, n_groups, nb, out_stride, brp_bits, bq);
    else if (mode == 1)
        k_msm_glv_windowed<<<(unsigned)((msms + 15) / 16), 256, 0, st>>>((const GlvScalar*)scalars, (const TabP*)table, (JacQ*)out, n_groups,
                                                                        n_slices, nb, out_stride, brp_bits, bq);
    else if (mode == 3)
        k_msm_glv_lane<1><<<(unsigned)((msms + 63) / 64), 64, 0, st>>>((const GlvScalar*)scalars, (const TabP*)table, (JacQ*)out, n_groups,
                                                                      n_slices, nb, out_stride, brp_bits, bq);
    else if (mode == 4)
        k_msm_glv_lane<2><<<(unsigned)((msms + 63) / 64), 128, 0, st>>>((const GlvScalar*)scalars, (const TabP*)table, (JacQ*)out, n_groups,
                                                                       n_slices, nb, out_stride, brp_bits, bq);
    else
        k_msm_glv_chunked<<<(unsigned)((msms + 63) / 64), 256, 0, st>>>((const GlvScalar*)scalars, (const TabP*)table, (JacQ*)out, n_groups,
                                                                       n_slices, nb, out_stride, brp_bits, bq);
}
template <int C>
static void msm_chunked_c(const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                          int brp_bits, int S, hipStream_t st) {
    const long threads = (long)n_groups * n_slices * S;
    k_msm_fixed_chunked<C><<<(unsigned)((threads + 255) / 256), 256, 0, st>>>((const Fr*)scalars, (const TabQ*)table, (JacQ*)out,
                                                                               n_groups, n_slices, nb, out_stride, brp_bits, S);
}
void msm_fixed_chunked(int c, const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                       int brp_bits, int S, hipStream_t st) {
    if (c == 8) msm_chunked_c<8>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
    else if (c == 12) msm_chunked_c<12>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
    else if (c == 13) msm_chunked_c<13>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
    else if (c == 14) msm_chunked_c<14>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
    else if (c == 10) msm_chunked_c<10>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
    else msm_chunked_c<4>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, S, st);
}
template <int C>
static void msm_c(const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
                  int brp_bits, hipStream_t st) {
    constexpr int PB = 256 / ((255 + C) / C);
    long total = (long)n_groups * n_slices;
    k_msm_fixed<C><<<(unsigned)((total + PB - 1) / PB), 256, 0, st>>>((const Fr*)scalars, (const TabQ*)table, (JacQ*)out,
                                                                     n_groups, n_slices, nb, out_stride, brp_bits);
}
void msm_fixed(int c, const void* scalars, const void* table, void* out, int n_groups, int n_slices, int nb, int out_stride,
               int brp_bits, hipStream_t st) {
    if (c == 8) msm_c<8>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 12) msm_c<12>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 13) msm_c<13>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 14) msm_c<14>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else if (c == 10) msm_c<10>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
    else msm_c<4>(scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, st);
}
}  // namespace launch
}  // namespace kzg
