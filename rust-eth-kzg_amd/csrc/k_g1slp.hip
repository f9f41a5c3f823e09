// Executor of the straight-line program that g1_linmap.hpp compiles for the FK20 proofs map (the two G1 transforms of
// compute_cells_and_kzg_proofs as one linear map; reference: fft_inplace<G1Projective> via Domain::{ifft_g1_take_n,
// fft_g1}, crates/cryptography/polynomial/src/domain.rs:149-194).
//
// Arena layout: A[slot * stride + lane], lane = blob index inside the batch (stride = batch padded to a multiple of 64):
// one wave = one operation x 64 blobs, so the operation descriptor and, for multiplications, the digits of the public
// constant are wave-uniform (scalar loads and scalar branches, no divergence).  Operations of one launch are mutually
// independent and never write a slot that the same launch reads (linmap::make_schedule).
// words: 4 per operation: dst slot, a slot, b (slot | number of doublings | constant id), flags (1 = subtract, 2 = doubling run).
#include "engine.hpp"
#include "g1_mulc.hpp"

namespace kzg {

__global__ __launch_bounds__(64, 2) void k_slp_mulc(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                    const uint32_t* __restrict__ naf, Fq<1> beta) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 64 + threadIdx.x;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta);
}
// additions, subtractions (flags & 1) and runs of doublings (flags & 2, b = count) of one step, one wave per operation
__global__ __launch_bounds__(64) void k_slp_add(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    const int lane = blockIdx.y * 64 + threadIdx.x;
    JacQ r = A[(size_t)a * stride + lane];
    if (fl & 2u) {
#pragma unroll 1
        for (uint32_t k = 0; k < b; k++) r = dbl(r);
    } else {
        r = add(r, A[(size_t)b * stride + lane], (fl & 1u) != 0);
    }
    A[(size_t)dst * stride + lane] = r;
}

namespace launch {
// kind: 3 multiplication by a constant, anything else the mixed addition / subtraction / doubling launch (linmap::OpKind)
void g1_slp_launch(int kind, void* arena, int stride, const uint32_t* words, int count, const void* naf, const Fp12w& beta,
                   hipStream_t st) {
    const dim3 grid((unsigned)count, (unsigned)(stride / 64));
    if (kind == 3) {
        Fp b384;
        for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
        k_slp_mulc<<<grid, 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf, fq_from_fp(b384));
    } else {
        k_slp_add<<<grid, 64, 0, st>>>((JacQ*)arena, stride, words);
    }
}
}  // namespace launch
}  // namespace kzg
