// G1 FFT over 128 positions, batched over blobs (stages E and F of compute_cells_and_kzg_proofs;
// also the set-up FFTs of the SRS vectors).  Reference: fft_inplace<G1Projective> via
// Domain::{fft_g1, ifft_g1_take_n} (crates/cryptography/polynomial/src/domain.rs:149-194, fft.rs:46-177).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "launch.hpp"

namespace kzg {

// Data layout: X[pos * stride + lane], lane = blob index inside the batch (stride = batch padded to a
// multiple of 64), so all 64 lanes of a wave run the SAME butterfly and the twiddle is wave-uniform.
// `b * twiddle` (fft.rs:164-177) is a 255-bit scalar multiplication by a PUBLIC constant, so the scalar is
// recoded once on the host: GLV split k = k1 + k2*lambda (phi(x,y) = (beta x, y) = [lambda](x,y)), then each
// 128-bit half in width-w non-adjacent form (odd digits |d| < 2^(w-1), one non-zero digit in w+1 on average).
// The two digit streams share the <= 129 doublings; the additions take (2j+1) P from a small table built with
// one doubling and 2^(w-2) - 1 additions (phi of a table entry only swaps in beta x).  w = 5: ~43 additions + 8
// for the table, against ~66 for the joint sparse form and 255 doublings + ~85 additions for a plain NAF.
// Every digit test is a scalar branch on wave-uniform data: no lane divergence.
// tab[k][2][33] words = 2 x 132 signed bytes: digit t of half h is byte t of tab[k][h].
__device__ __forceinline__ JacQ mul_by_twiddle(const JacQ& p, const uint32_t* __restrict__ tab, const Fq<1>& beta, int k) {
    // k is wave-uniform; 0 -> identity map, 64 -> negation (omega_128^64 = -1)
    if (k == 0) return p;
    if (k == 64) return neg(p);
    constexpr int NT = 1 << (launch::TWIDDLE_WNAF_W - 2);  // odd multiples P, 3P, .., (2 NT - 1) P
    // The table is brought to ONE common Z = prod z_j without an inversion: (X_j l_j^2, Y_j l_j^3) with l_j = Z / z_j are
    // the affine coordinates of the same points on the isomorphic curve y^2 = x^3 + 4 Z^6.  The group law for a = 0
    // never looks at b, and phi(x, y) = (beta x, y) is an endomorphism of that curve too, so the whole multiplication
    // runs there with MIXED additions (6M + 3S + pair instead of 10M + 4S + pair) and one final Z <- Z * Z_common.
    AffQ2 A[NT];
    Fq<2> bx[NT];
    Fq<ZB> zc;
    {
        JacQ T[NT];
        Fq<ZB> pre[NT];  // pre[j] = z_0 ... z_j
        const JacQ p2 = dbl(p);
        T[0] = p;
        pre[0] = p.z;
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            T[j] = add(T[j - 1], p2);
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
    const uint32_t* row = tab + (size_t)k * (2 * launch::TWIDDLE_WORDS);
    JacQ acc = jacq_inf();
    bool started = false;
#pragma unroll 1
    for (int wd = launch::TWIDDLE_WORDS - 1; wd >= 0; wd--) {
        const uint32_t w1 = __builtin_amdgcn_readfirstlane(row[wd]);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(row[launch::TWIDDLE_WORDS + wd]);
        if (!started && (w1 | w2) == 0) continue;
#pragma unroll 1
        for (int q = 3; q >= 0; q--) {
            if (started) acc = dbl(acc);
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                const int d = (int)(int8_t)((h ? w2 : w1) >> (8 * q));
                if (d == 0) continue;
                const int idx = ((d < 0 ? -d : d) - 1) >> 1;
                AffQ2 op = A[idx];
                if (h) op.x = bx[idx];
                if (!started) {
                    acc.x = relax<XB>(op.x);
                    acc.y = d < 0 ? relax<XB>(neg(op.y)) : relax<XB>(op.y);
                    acc.z = relax<ZB>(fq_one());
                    started = true;
                } else acc = add_mixed(acc, op, d < 0);
            }
        }
    }
    acc.z = relax<ZB>(mul(acc.z, zc));  // back from the isomorphic curve
    return acc;
}

// One radix-2 layer = two small kernels, so that each heavy primitive (Jacobian add / double) is inlined once:
//   k_g1_twiddle_mul : X[i1] <- w^e * X[src]      (src = i1, or i0 for the (h || 0) first DIF layer)
//   k_g1_butterfly   : (X[i0], X[i1]) <- (X[i0] + X[i1], X[i0] - X[i1])   (difference optional)
// DIT layer (inverse FFT, natural out):      twiddle_mul on b, then butterfly.
// DIF layer (forward FFT, bit-reversed out): butterfly, then twiddle_mul on b.
// q = butterfly index; half = butterfly span; twiddle exponent = j * tw_step (128 - that when inverse).
// grid = (n_bfly = 64, stride/64), block = 64 (one wave = one butterfly x 64 blobs).
__global__ __launch_bounds__(64, 2) void k_g1_twiddle_mul(JacQ* __restrict__ X, int stride, int half, int tw_step, int inverse,
                                                          int from_a, const uint32_t* __restrict__ tw, Fq<1> beta) {
    // in-place layers launch only the butterflies whose twiddle is not 1 (j != 0): compact index c -> q = c + c / (half - 1) + 1
    const int c = blockIdx.x, lane = blockIdx.y * 64 + threadIdx.x;
    const int q = from_a ? c : c + c / (half - 1) + 1;
    const int j = q & (half - 1);
    const int i0 = ((q - j) << 1) + j, i1 = i0 + half;
    int e = (j * tw_step) & 127;
    if (inverse) e = (128 - e) & 127;
    const JacQ src = X[(size_t)(from_a ? i0 : i1) * stride + lane];
    X[(size_t)i1 * stride + lane] = mul_by_twiddle(src, tw, beta, e);
}
__global__ __launch_bounds__(64) void k_g1_butterfly(JacQ* __restrict__ X, int stride, int half, int want_diff) {
    const int q = blockIdx.x, lane = blockIdx.y * 64 + threadIdx.x;
    const int j = q & (half - 1);
    const int i0 = ((q - j) << 1) + j, i1 = i0 + half;
    JacQ* pa = X + (size_t)i0 * stride + lane;
    JacQ* pb = X + (size_t)i1 * stride + lane;
    const JacQ a = *pa;
    const JacQ b = *pb;
#pragma unroll 1
    for (int s = 0; s < 2; s++) {  // one inlined add serves both the sum and the difference
        if (s == 1 && !want_diff) break;
        JacQ r = add(a, b, s == 1);
        if (s == 0) *pa = r;
        else *pb = r;
    }
}

// ------------------------------------------------------------------------------------------------
// Latency mode for up to three 64-blob groups (BASELINE.json configs 2 and 4: a single blob / 64 blobs per GPU).
// The radix-2 network needs 14 sequential twiddle multiplications while 94 % of the SIMDs idle.  Here the
// 128-point transform is a two-stage Cooley-Tukey 8 x 16 evaluated directly: every (output, term) product
// w^e * x is its own wave, so a transform is two rounds of twiddle multiplications + two log-depth sums.
// ~8x the multiplications of radix 2, but they run on otherwise idle SIMDs: 4 rounds instead of 14.
//   n = 16 n1 + n2,  k = k1 + 8 k2:
//   stage 1:  A[k1][n2]   = sum_{n1 < 8}  w^(16 n1 k1)        x[16 n1 + n2]
//   stage 2:  X[k1+8 k2]  = sum_{n2 < 16} w^(n2 k1 + 8 n2 k2) A[k1][n2]
// prod[(o * R + t) * stride + lane];  terms: number of non-zero input terms of the stage (forward FFT of (h || 0): 4).
__global__ __launch_bounds__(64, 2) void k_g1_dft_products(const JacQ* __restrict__ in, JacQ* __restrict__ prod, int stride,
                                                           int stage, int terms, int inverse, const uint32_t* __restrict__ tw,
                                                           Fq<1> beta) {
    const int R = stage == 1 ? 8 : 16;
    const int o = blockIdx.x / terms, t = blockIdx.x % terms, lane = blockIdx.y * 64 + threadIdx.x;
    int src, e;
    if (stage == 1) {  // o = k1 * 16 + n2
        const int k1 = o >> 4, n2 = o & 15;
        src = 16 * t + n2;
        e = (16 * t * k1) & 127;
    } else {  // o = k1 + 8 k2, reads A[k1 * 16 + t]
        const int k1 = o & 7, k2 = o >> 3;
        src = k1 * 16 + t;
        e = (t * k1 + 8 * t * k2) & 127;
    }
    if (inverse) e = (128 - e) & 127;
    prod[((size_t)o * R + t) * stride + lane] = mul_by_twiddle(in[(size_t)src * stride + lane], tw, beta, e);
}
// out[perm(o)] = sum_{t < terms} prod[o][t]; block = 256 threads = 64 lanes x 4 partial sums
__global__ __launch_bounds__(256) void k_g1_dft_sum(const JacQ* __restrict__ prod, JacQ* __restrict__ out, int stride, int R,
                                                    int terms, int brp_out) {
    __shared__ JacQ part[4][64];
    const int o = blockIdx.x, lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int gl = blockIdx.y * 64 + lane;
    JacQ acc = jacq_inf();
    for (int t = quarter; t < terms; t += 4) acc = add(acc, prod[((size_t)o * R + t) * stride + gl]);
    part[quarter][lane] = acc;
    __syncthreads();
    for (int span = 2; span >= 1; span >>= 1) {
        if (quarter < span) part[quarter][lane] = add(part[quarter][lane], part[quarter + span][lane]);
        __syncthreads();
    }
    if (quarter == 0) {
        const int pos = brp_out ? (int)(__brev((unsigned)o) >> 25) : o;
        out[(size_t)pos * stride + gl] = part[0][lane];
    }
}

namespace launch {
// mode 0: DIT butterfly (a, b) -> (a + w b, a - w b);  mode 1: DIF butterfly (a, b) -> (a + b, (a - b) w);
// mode 2: DIF first layer with b == identity: b <- a w;  mode 3: DIT last layer keeping only a <- a + w b.
void g1_fft_layer(void* X, int stride, int half, int tw_step, int inverse, int mode, const void* tw, const Fp12w& beta,
                  hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const Fq<1> bt = fq_from_fp(b384);  // host-side conversion to the 14 x 29-bit Montgomery-406 form
    dim3 grid(64, stride / 64);
    dim3 grid_tw(64 - 64 / half, stride / 64);  // butterflies with a twiddle other than 1 (none when half == 1)
    JacQ* x = (JacQ*)X;
    const uint32_t* js = (const uint32_t*)tw;
    switch (mode) {
        case 0:
            if (grid_tw.x) k_g1_twiddle_mul<<<grid_tw, 64, 0, st>>>(x, stride, half, tw_step, inverse, 0, js, bt);
            k_g1_butterfly<<<grid, 64, 0, st>>>(x, stride, half, 1);
            break;
        case 1:
            k_g1_butterfly<<<grid, 64, 0, st>>>(x, stride, half, 1);
            if (grid_tw.x) k_g1_twiddle_mul<<<grid_tw, 64, 0, st>>>(x, stride, half, tw_step, inverse, 0, js, bt);
            break;
        case 2:
            k_g1_twiddle_mul<<<grid, 64, 0, st>>>(x, stride, half, tw_step, inverse, 1, js, bt);
            break;
        default:
            if (grid_tw.x) k_g1_twiddle_mul<<<grid_tw, 64, 0, st>>>(x, stride, half, tw_step, inverse, 0, js, bt);
            k_g1_butterfly<<<grid, 64, 0, st>>>(x, stride, half, 0);
            break;
    }
}
// One 128-point transform of a few 64-lane groups in latency mode.  X: natural-order input [128][stride];
// n_in: number of leading non-identity inputs (128, or 64 for (h || 0)); n_out: outputs wanted (128 or the first 64);
// tmpA [128][stride], prod [128*16][stride].  Output goes back to X (natural order, or bit-reversed if brp_out).
void g1_dft128_direct(void* X, void* tmpA, void* prod, int stride, int n_in, int n_out, int inverse, int brp_out,
                      const void* tw, const Fp12w& beta, hipStream_t st) {
    Fp b384;
    for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
    const Fq<1> bt = fq_from_fp(b384);
    const uint32_t* js = (const uint32_t*)tw;
    const int terms1 = n_in / 16;   // n1 < n_in / 16
    const int outs2 = n_out;        // k < n_out  <=>  k2 < n_out / 8
    const int groups = stride / 64;
    k_g1_dft_products<<<dim3(128 * terms1, groups), 64, 0, st>>>((const JacQ*)X, (JacQ*)prod, stride, 1, terms1, inverse, js, bt);
    k_g1_dft_sum<<<dim3(128, groups), 256, 0, st>>>((const JacQ*)prod, (JacQ*)tmpA, stride, 8, terms1, 0);
    k_g1_dft_products<<<dim3(outs2 * 16, groups), 64, 0, st>>>((const JacQ*)tmpA, (JacQ*)prod, stride, 2, 16, inverse, js, bt);
    k_g1_dft_sum<<<dim3(outs2, groups), 256, 0, st>>>((const JacQ*)prod, (JacQ*)X, stride, 16, 16, brp_out);
}
}  // namespace launch
}  // namespace kzg
