// G1 FFT over 128 positions, batched over blobs (stages E and F of compute_cells_and_kzg_proofs;
// also the set-up FFTs of the SRS vectors).  Reference: fft_inplace<G1Projective> via
// Domain::{fft_g1, ifft_g1_take_n} (crates/cryptography/polynomial/src/domain.rs:149-194, fft.rs:46-177).
#include "kcommon.hpp"
#include "launch.hpp"

namespace kzg {

// Data layout: X[pos * stride + lane], lane = blob index inside the batch (stride = batch padded to a
// multiple of 64), so all 64 lanes of a wave run the SAME butterfly and the twiddle is wave-uniform.
// `b * twiddle` (fft.rs:164-177) is a 255-bit scalar multiplication by a public constant: its NAF digits
// are precomputed on the host (naf[k] for omega_128^k: 8 words non-zero mask, 8 words sign mask) and every
// digit test is a scalar branch -- no lane divergence.
__device__ __forceinline__ G1Jac mul_by_twiddle(const G1Jac& p, const uint32_t* __restrict__ naf, int k) {
    // k is wave-uniform; 0 -> identity map, 64 -> negation (omega_128^64 = -1)
    if (k == 0) return p;
    if (k == 64) return neg(p);
    uint32_t nz[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        nz[i] = __builtin_amdgcn_readfirstlane(naf[(size_t)k * 16 + i]);
        sg[i] = __builtin_amdgcn_readfirstlane(naf[(size_t)k * 16 + 8 + i]);
    }
    G1Jac np = neg(p);
    G1Jac acc = jac_inf();
    bool started = false;
#pragma unroll 1
    for (int wd = 7; wd >= 0; wd--) {
        uint32_t nzw = nz[wd], sgw = sg[wd];
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            if (started) acc = dbl(acc);
            if ((nzw >> bit) & 1) {
                bool minus = (sgw >> bit) & 1;
                if (!started) { acc = minus ? np : p; started = true; }
                else acc = add(acc, minus ? np : p);
            }
        }
    }
    return acc;
}

// One radix-2 layer.  MODE 0: DIT butterfly (a, b) -> (a + w b, a - w b)      [inverse FFT, natural out]
//                     MODE 1: DIF butterfly (a, b) -> (a + b, (a - b) w)      [forward FFT, bit-reversed out]
//                     MODE 2: DIF first layer with b == identity: (a, -) -> (a, a w)   (input h || 0)
//                     MODE 3: DIT last layer keeping only the first half: a <- a + w b
// q = butterfly index; half = butterfly span; twiddle exponent = j * tw_step (128 - that when inverse).
// grid = (n_bfly, stride/64), block = 64.
template <int MODE>
__global__ __launch_bounds__(64) void k_g1_fft_layer(G1Jac* __restrict__ X, int stride, int half, int tw_step, int inverse,
                                                     const uint32_t* __restrict__ naf) {
    const int q = blockIdx.x, lane = blockIdx.y * 64 + threadIdx.x;
    const int j = q & (half - 1);
    const int i0 = ((q - j) << 1) + j, i1 = i0 + half;
    int e = (j * tw_step) & 127;
    if (inverse) e = (128 - e) & 127;
    G1Jac* pa = X + (size_t)i0 * stride + lane;
    G1Jac* pb = X + (size_t)i1 * stride + lane;
    if (MODE == 0) {
        G1Jac a = *pa, t = mul_by_twiddle(*pb, naf, e);
        *pa = add(a, t);
        *pb = add(a, neg(t));
    } else if (MODE == 1) {
        G1Jac a = *pa, b = *pb;
        *pa = add(a, b);
        *pb = mul_by_twiddle(add(a, neg(b)), naf, e);
    } else if (MODE == 2) {
        *pb = mul_by_twiddle(*pa, naf, e);
    } else {
        G1Jac a = *pa, t = mul_by_twiddle(*pb, naf, e);
        *pa = add(a, t);
    }
}

namespace launch {
void g1_fft_layer(void* X, int stride, int half, int tw_step, int inverse, int mode, const void* naf, hipStream_t st) {
    dim3 grid(64, stride / 64);
    G1Jac* x = (G1Jac*)X;
    const uint32_t* nf = (const uint32_t*)naf;
    switch (mode) {
        case 0: k_g1_fft_layer<0><<<grid, 64, 0, st>>>(x, stride, half, tw_step, inverse, nf); break;
        case 1: k_g1_fft_layer<1><<<grid, 64, 0, st>>>(x, stride, half, tw_step, inverse, nf); break;
        case 2: k_g1_fft_layer<2><<<grid, 64, 0, st>>>(x, stride, half, tw_step, inverse, nf); break;
        default: k_g1_fft_layer<3><<<grid, 64, 0, st>>>(x, stride, half, tw_step, inverse, nf); break;
    }
}
}  // namespace launch
}  // namespace kzg
