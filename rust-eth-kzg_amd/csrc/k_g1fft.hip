// G1 FFT over 128 positions, batched over blobs (stages E and F of compute_cells_and_kzg_proofs;
// also the set-up FFTs of the SRS vectors).  Reference: fft_inplace<G1Projective> via
// Domain::{fft_g1, ifft_g1_take_n} (crates/cryptography/polynomial/src/domain.rs:149-194, fft.rs:46-177).
#include "engine.hpp"
#include "kcommon.hpp"
#include "curve29.hpp"
#include "launch.hpp"
#include "g1_mulc.hpp"

namespace kzg {

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
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1fft() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_g1_butterfly));
}
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
