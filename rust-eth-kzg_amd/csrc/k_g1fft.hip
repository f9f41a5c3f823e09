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
}  // namespace launch
}  // namespace kzg
