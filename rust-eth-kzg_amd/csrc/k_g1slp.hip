// Executor of the straight-line program that g1_linmap.hpp compiles for the FK20 proofs map (the two G1 transforms of
// compute_cells_and_kzg_proofs as one linear map; reference: fft_inplace<G1Projective> via Domain::{ifft_g1_take_n,
// fft_g1}, crates/cryptography/polynomial/src/domain.rs:149-194).
//
// Arena layout: A[slot * stride + lane], lane = blob index inside the batch (stride = batch padded to a multiple of 64):
// one wave = one operation x 64 blobs, so the operation descriptor and, for multiplications, the digits of the public
// constant are wave-uniform (scalar loads and scalar branches, no divergence).  Operations of one launch are mutually
// independent and never write a slot that the same launch reads (linmap::make_schedule).
// Point format of the arena (launch::FMT_*, chosen by the engine): signed 13 x 30-bit points (JacS) and the `_s` kernels below -- one
// field from the MSM's sums to the proofs' compression, with one, two or four lanes per blob -- unless ETH_KZG_AMD_ARENA_SIGNED=0 asks
// for the 14 x 29-bit points (JacQ) and kernels of rounds 2-5 (the cross-check).
// words: 4 per operation: dst slot, a slot, b (slot | number of doublings | constant id), flags (1 = subtract, 2 = doubling run, 4 = a + b to dst AND a - b to slot flags >> 16; bits 3-7 of an addition: doublings of operand a first).
#include "engine.hpp"
#include <stdexcept>
#include "g1_mulc.hpp"
#include "g1_mulc30.hpp"
#include "g1_coop.hpp"

namespace kzg {

static_assert(sizeof(JacS) == launch::SIZEOF_JACS && sizeof(JacQ) == launch::SIZEOF_JACQ, "the engine sizes and offsets the arena with these");

// (a lane per blob on a 14 x 29-bit arena -- ETH_KZG_AMD_ARENA_SIGNED=0: the
// multiplication runs in the signed 13 x 30-bit field, g1_mulc30.hpp; the point is converted on the way in and out)
// n_active (every lane-per-blob kernel below): the blobs that are really there.  The lanes behind them (the batch is padded to a
// multiple of 64) hold the identity, and an identity operand sends its WAVE through the exact slow path of every operation: such
// lanes leave at once instead (round 6: 288 blobs ran their map in 3.80 ms, 320 blobs in 3.33 ms).
__global__ __launch_bounds__(64, 2) void k_slp_mulc(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                    const uint32_t* __restrict__ naf, Fs<1, DC> beta, int n_active) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = (gridDim.y - 1 - blockIdx.y) * 64 + threadIdx.x;  // (the last lane group first: see k_slp_mulc_s)
    if (lane >= n_active) return;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded30(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta);
}
// ... and on an arena in the signed form itself (launch::FMT_JACS: what the engine runs): nothing is converted
__global__ __launch_bounds__(64, 2) void k_slp_mulc_s(JacS* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                      const uint32_t* __restrict__ naf, Fs<1, DC> beta, int n_active) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    // The lane groups are dealt LAST GROUP FIRST: the last one is the one that may be partly filled, and a wave with 16 or fewer
    // lanes in use runs this kernel about twice as long as a full one (measured, not explained: 2049 .. 2064 blobs spent 14.6 ms
    // here, 2072 .. 2112 blobs 13.2 ms, tools/sweep_partial_group.sh) -- at the end of the launch that is a tail, at its start it
    // is hidden under the other 32 groups.
    const int lane = (gridDim.y - 1 - blockIdx.y) * 64 + threadIdx.x;
    // A wave with only some of its lanes in use is the SLOWER wave on this part (measured twice: this kernel with <= 16 lanes of the
    // last group in use lasts about twice as long; k_g1_compress on 12 or 32 of 64 lanes 191 us, on all 64 lanes 150 us), so the
    // lanes behind the last blob do not leave: they repeat its work and store nothing.
    const JacS src = A[(size_t)a * stride + (lane < n_active ? lane : n_active - 1)];
    const JacS r = mul_by_recoded30(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta);
    if (lane < n_active) A[(size_t)dst * stride + lane] = r;
}
// the constant multiplications of a batch of <= 16 blobs: four lanes per blob (a wave = 16 blobs x one operation), the quad sharing
// the digit loop's doublings and mixed additions
__global__ __launch_bounds__(64, 2) void k_slp_mulc_coop(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                         const uint32_t* __restrict__ naf, Fq<1> beta, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (lane >= lanes) return;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded<4>(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta, quad);
}
// ... and of 17 .. 64 blobs (BASELINE config 5's and 4's per-GPU shares): two lanes per blob, a wave = 32 blobs x one operation
// (from 33 blobs on two waves per operation: the engine then picks a compilation of the map with <= 512 multiplications).
// These three kernels serve the 14 x 29-bit arena of ETH_KZG_AMD_ARENA_SIGNED=0 (the cross-check of the signed path: k_slp_*_s below).
__global__ __launch_bounds__(64, 2) void k_slp_mulc_coop2(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                          const uint32_t* __restrict__ naf, Fq<1> beta, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 32 + (threadIdx.x >> 1), half = threadIdx.x & 1;
    if (lane >= lanes) return;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded<2>(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta, half);
}
// one cheap operation of the program on one lane: flags & 2: a run of b doublings; otherwise an addition (flags & 1: subtraction;
// flags & 4: a + b to dst AND a - b to slot flags >> 16) whose FIRST operand is doubled (flags >> 3) & 31 times in registers
// before the second one is read -- the schedule folds a doubling run into its only consumer (g1_linmap.hpp: make_schedule)
__device__ __forceinline__ void slp_cheap_op(JacQ* __restrict__ A, int stride, int lane, uint32_t dst, uint32_t a, uint32_t b, uint32_t fl) {
    JacQ r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = dbl(r);
    bool degenerate = false;
    if (!(fl & 2u)) {
        if (fl & 4u) {  // the difference is stored before the sum is computed (curve29.hpp: add_sub_*)
            const AddSubShared sh = add_sub_prepare(r, A[(size_t)b * stride + lane]);
            degenerate = sh.degenerate;  // (an identity, a = +-b: both results are redone below; what is stored here is overwritten)
            A[(size_t)(fl >> 16) * stride + lane] = add_sub_finish(sh, true);
            r = add_sub_finish(sh, false);
        } else {
            r = add(r, A[(size_t)b * stride + lane], (fl & 1u) != 0);
        }
    }
    A[(size_t)dst * stride + lane] = r;
    // The exact slow path of a pair comes LAST, when nothing else is live (inside the branch above its operands cost the common
    // path 33 spilled registers): the operands are read and doubled again.  Rare: all-zero / constant / two-valued blobs.
    if (degenerate) {
        asm volatile("" ::: "memory");
        JacQ p2 = A[(size_t)a * stride + lane];
#pragma unroll 1
        for (uint32_t k = 0; k < runs; k++) p2 = dbl(p2);
        const JacQ q2 = A[(size_t)b * stride + lane];
        const JacQ d = add_slow(p2, q2, true);
        A[(size_t)dst * stride + lane] = add_slow(p2, q2, false);
        A[(size_t)(fl >> 16) * stride + lane] = d;
    }
}
// additions, subtractions and runs of doublings of one step, one wave per operation.  Blocks are dealt in blockIdx order, x
// fastest: x = lane group, y = operation, and the schedule lists a step's operations longest first (g1_linmap.hpp), so every
// group's long operations start first and the launch ends on short ones.
__global__ __launch_bounds__(64, 2) void k_slp_add(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words, int n_active) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    if ((int)(blockIdx.x * 64 + threadIdx.x) >= n_active) return;
    slp_cheap_op(A, stride, blockIdx.x * 64 + threadIdx.x, dst, a, b, fl);
}

// The same operations on an arena in the signed 13 x 30-bit form (launch::FMT_JACS; curve30.hpp): add-1998-cmo-2 with the
// subtractions fused into the reductions, the sum-and-difference pair with its shared part computed once (add_sub), doubling runs
// in the halved form (dbl_half: (X / 4, Y / 8, Z / 2) is the same point).  Degenerate operands -- an identity, a = +-b -- leave by
// add_slow inside add / add_sub (Z3 = Z1 Z2 H is a fresh product: zero iff its digits are).
__global__ __launch_bounds__(64, 2) void k_slp_add_s(JacS* __restrict__ A, int stride, const uint32_t* __restrict__ words, int n_active) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    const int lane_of_thread = blockIdx.x * 64 + threadIdx.x;
    const bool keep = lane_of_thread < n_active;  // (padding lanes repeat the last blob's work and store nothing: see k_slp_mulc_s)
    const int lane = keep ? lane_of_thread : n_active - 1;
    JacS r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = dbl_half(r);
    bool degenerate = false;
    if (!(fl & 2u)) {
        if (fl & 4u) {  // the difference is stored before the sum is computed
            const AddSubSharedS sh = add_sub_prepare(r, A[(size_t)b * stride + lane]);
            degenerate = sh.degenerate;  // (an identity, a = +-b: both results are redone below; what is stored here is overwritten)
            const JacS df = add_sub_finish(sh, true);
            if (keep) A[(size_t)(fl >> 16) * stride + lane] = df;
            r = add_sub_finish(sh, false);
        } else {
            r = add_unchecked(r, A[(size_t)b * stride + lane], (fl & 1u) != 0, degenerate);
        }
    }
    if (keep) A[(size_t)dst * stride + lane] = r;
    // the exact slow path comes LAST, when nothing else is live (kept inside the formulas it would hold both operands alive across
    // them): the operands are read and doubled again.  Rare: all-zero / constant / two-valued / sparse blobs.
    if (degenerate && keep) {
        asm volatile("" ::: "memory");
        JacS p2 = A[(size_t)a * stride + lane];
#pragma unroll 1
        for (uint32_t k = 0; k < runs; k++) p2 = dbl_half(p2);
        const JacS q2 = A[(size_t)b * stride + lane];
        if (fl & 4u) {
            const JacS d = add_slow(p2, q2, true);
            A[(size_t)dst * stride + lane] = add_slow(p2, q2, false);
            A[(size_t)(fl >> 16) * stride + lane] = d;
        } else {
            A[(size_t)dst * stride + lane] = add_slow(p2, q2, (fl & 1u) != 0);
        }
    }
}

// The cheap operations of ONE lane group (<= 64 blobs: BASELINE config 4's and 5's per-GPU shares) with four lanes per blob
// (g1_coop.hpp): a level is then a few hundred waves on an idle chip, each a single addition -- 16.5 multiplication times for one
// lane, 5.5 for a quad; a doubling run likewise 3.5 per doubling instead of 6.5.  The sum-and-difference pair is two
// quad additions (11 against the shared form's 20).  16 blobs per wave; every lane of a quad stores the same result.
__global__ __launch_bounds__(64, 2) void k_slp_add_coop(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    const int lane = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (lane >= lanes) return;
    JacQ r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = coop_dbl(r, quad);
    if (!(fl & 2u)) {
        const JacQ q = A[(size_t)b * stride + lane];
        if (fl & 4u) {
            const JacQ d = coop_add(r, q, true, quad);
            A[(size_t)(fl >> 16) * stride + lane] = d;
            r = coop_add(r, q, false, quad);
        } else {
            r = coop_add(r, q, (fl & 1u) != 0, quad);
        }
    }
    A[(size_t)dst * stride + lane] = r;
}

// ---- one lane group or less on an arena in the signed form (launch::FMT_JACS; round 6): the several-lanes-per-blob kernels of the
// 13 x 30-bit field (g1_coop30.hpp) -- with them the prover's points are in ONE Fp representation at every batch size above the
// circulant form's.  COOP = 4: <= 16 blobs, a wave = 16 blobs x one operation; COOP = 2: 17 .. 64 blobs, a wave = 32 blobs.
template <int COOP>
__global__ __launch_bounds__(64, 2) void k_slp_mulc_coop_s(JacS* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                           const uint32_t* __restrict__ naf, Fs<1, DC> beta, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane_of_thread = blockIdx.y * (64 / COOP) + (threadIdx.x / COOP), sub = threadIdx.x % COOP;
    const bool keep = lane_of_thread < lanes;  // (padding lanes repeat the last blob's work and store nothing: see k_slp_mulc_s)
    const int lane = keep ? lane_of_thread : lanes - 1;
    const JacS src = A[(size_t)a * stride + lane];
    const JacS r = mul_by_recoded30<COOP>(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta, sub);
    if (keep) A[(size_t)dst * stride + lane] = r;
}
// the cheap operations with four lanes per blob (k_slp_add_coop's schedule): a doubling run 3 reductions deep per doubling, an
// addition 5, the sum-and-difference pair 4 levels + its two fused pairs.  Every lane of a quad stores the same result.
__global__ __launch_bounds__(64, 2) void k_slp_add_coop_s(JacS* __restrict__ A, int stride, const uint32_t* __restrict__ words, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    const int lane_of_thread = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    const bool keep = lane_of_thread < lanes;
    const int lane = keep ? lane_of_thread : lanes - 1;
    JacS r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = coop4_dbl_half(r, quad);
    if (!(fl & 2u)) {
        const JacS q = A[(size_t)b * stride + lane];
        if (fl & 4u) {
            JacS d;
            coop4_add_sub(r, q, quad, r, d);
            if (keep) A[(size_t)(fl >> 16) * stride + lane] = d;
        } else {
            r = coop4_add(r, q, (fl & 1u) != 0, quad);
        }
    }
    if (keep) A[(size_t)dst * stride + lane] = r;
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1slp() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_slp_add));
}
// kind: 3 multiplication by a constant, anything else the mixed addition / subtraction / doubling launch (linmap::OpKind)
void g1_slp_launch(int kind, void* arena, int stride, const uint32_t* words, int count, const void* naf, const Fp12w& beta,
                   hipStream_t st, int lanes, int coop_lanes, int fmt, int n_active) {
    if (lanes <= 0) lanes = stride;  // (a sub-range of the lanes: arena already points at its first lane, stride stays the arena's)
    if (n_active <= 0 || n_active > lanes) n_active = lanes;
    const dim3 grid((unsigned)count, (unsigned)(lanes / 64));
    if (fmt == FMT_JACS) {  // everything in the signed field (the engine's format unless ETH_KZG_AMD_ARENA_SIGNED=0)
        const bool coop = coop_points_max() > 0;
        if (kind == 3) {
            Fp b384;
            for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
            const Fs<1, DC> bs = fs_from_fp(b384);
            // coop_lanes: the blobs that are really there when they are few enough for four lanes each (<= 16: one quad wave per
            // operation) or two (17 .. 64)
            if (coop_lanes > 16 && coop)
                k_slp_mulc_coop_s<2><<<dim3((unsigned)count, (unsigned)((coop_lanes + 31) / 32)), 64, 0, st>>>((JacS*)arena, stride, words, (const uint32_t*)naf, bs, coop_lanes);
            else if (coop_lanes > 0 && coop)
                k_slp_mulc_coop_s<4><<<dim3((unsigned)count, (unsigned)((coop_lanes + 15) / 16)), 64, 0, st>>>((JacS*)arena, stride, words, (const uint32_t*)naf, bs, coop_lanes);
            else k_slp_mulc_s<<<grid, 64, 0, st>>>((JacS*)arena, stride, words, (const uint32_t*)naf, bs, n_active);
        } else {
            // one lane group and few enough operations for every quad wave to have a SIMD of its own: four lanes per blob
            if (lanes == 64 && count * 4 <= 1024 && coop)
                k_slp_add_coop_s<<<dim3((unsigned)((n_active + 15) / 16), (unsigned)count), 64, 0, st>>>((JacS*)arena, stride, words, n_active);
            else k_slp_add_s<<<dim3((unsigned)(lanes / 64), (unsigned)count), 64, 0, st>>>((JacS*)arena, stride, words, n_active);
        }
        return;
    }
    if (kind == 3) {
        Fp b384;
        for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
        // coop_lanes: the blobs that are really there when they are few enough for four lanes each (<= 16: one quad wave per operation)
        // or two (<= 32: still one wave per operation)
        if (coop_lanes > 16 && coop_points_max() > 0) {
            const dim3 g2((unsigned)count, (unsigned)((coop_lanes + 31) / 32));
            k_slp_mulc_coop2<<<g2, 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf, fq_from_fp(b384), coop_lanes);
        }
        else if (coop_lanes > 0 && coop_points_max() > 0)
            k_slp_mulc_coop<<<dim3((unsigned)count, (unsigned)((coop_lanes + 15) / 16)), 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf,
                                                                                                     fq_from_fp(b384), coop_lanes);
        else k_slp_mulc<<<grid, 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf, fs_from_fp(b384), n_active);
    } else {
        // one lane group and few enough operations for every quad wave to have a SIMD of its own: four lanes per blob
        if (lanes == 64 && count * 4 <= 1024 && coop_points_max() > 0)
            k_slp_add_coop<<<dim3((unsigned)((n_active + 15) / 16), (unsigned)count), 64, 0, st>>>((JacQ*)arena, stride, words, n_active);
        else k_slp_add<<<dim3((unsigned)(lanes / 64), (unsigned)count), 64, 0, st>>>((JacQ*)arena, stride, words, n_active);
    }
}
}  // namespace launch
}  // namespace kzg
