// Kernels of verify_cell_kzg_proof_batch and recover_cells_and_kzg_proofs.
// Reference: FK20Verifier::verify_multi_opening (crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:129-260,
// compute_sum_interpolation_poly :348-384), g1_lincomb (crates/cryptography/bls12_381/src/lincomb.rs:7-30),
// ReedSolomon::recover_polynomial_coefficient (crates/cryptography/erasure_codes/src/reed_solomon.rs:332-384),
// recover_evaluations_in_domain_order (fk20/cosets.rs:141-198), deserialize_cells (serialization/src/lib.rs:107-114).
#include "engine.hpp"
#include "kcommon.hpp"
#include "launch.hpp"

namespace kzg {

// LDS-resident 4096-point NTT helpers (same network as k_ntt.hip)
__device__ __forceinline__ Fr vl_load(const uint32_t* s, int idx) {
    Fr r;
#pragma unroll
    for (int l = 0; l < 8; l++) r.v[l] = s[l * N_BLOB + idx];
    return r;
}
__device__ __forceinline__ void vl_store(uint32_t* s, int idx, const Fr& a) {
#pragma unroll
    for (int l = 0; l < 8; l++) s[l * N_BLOB + idx] = a.v[l];
}
__device__ __forceinline__ void v_dit_inverse4096(uint32_t* s, const Fr* __restrict__ w8192) {
    for (int half = 1; half < N_BLOB; half <<= 1) {
        const int tw_step = N_EXT / (2 * half);
        for (int q = threadIdx.x; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = vl_load(s, i0), b = vl_load(s, i1);
            Fr t = j ? mul(b, w8192[(N_EXT - j * tw_step) & (N_EXT - 1)]) : b;
            vl_store(s, i0, add(a, t));
            vl_store(s, i1, sub(a, t));
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void v_dif_forward4096(uint32_t* s, const Fr* __restrict__ w8192) {
    for (int half = N_BLOB / 2; half >= 1; half >>= 1) {
        const int tw_step = N_EXT / (2 * half);
        for (int q = threadIdx.x; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = vl_load(s, i0), b = vl_load(s, i1);
            Fr d = sub(a, b);
            vl_store(s, i0, add(a, b));
            vl_store(s, i1, j ? mul(d, w8192[j * tw_step]) : d);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// cells (n x 2048 B, big-endian) -> Fr Montgomery [slot][64]; status |= 1 on a non-canonical element.
// slot_of == nullptr: slot = cell number (verify).  Otherwise the cell lands at slot_of[k] (recover scatter,
// cosets.rs:170-175); evals must be zero-filled first.
__global__ void k_cells_to_fr(const uint8_t* __restrict__ cells, Fr* __restrict__ evals, const int* __restrict__ slot_of,
                              int* __restrict__ status, int n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * CELL_LEN) return;
    int k = idx >> 6, e = idx & 63;
    Fr x = load_fr_be(cells + (size_t)idx * 32);
    if (geq_mod<FrParams>(x.v)) atomicOr(status, 1);
    int slot = slot_of ? slot_of[k] : k;
    evals[(size_t)slot * CELL_LEN + e] = to_mont(x);
}

// r^k for k < n from the table r^(2^i) (compute_powers, verifier.rs:333-343), plus the scalars of the
// two proof MSMs: s1[k] = r^k, s2[k] = r^k * h_{idx_k}^64 (verifier.rs:186-201), both out of Montgomery form.
struct PowTable { Fr p[24]; };
__global__ void k_verify_scalars(PowTable tab, const int* __restrict__ cell_idx, const Fr* __restrict__ w8192,
                                 Fr* __restrict__ rp_mont, Fr* __restrict__ s1, Fr* __restrict__ s2, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr acc = one<FrParams>();
    for (int i = 0; i < 24; i++)
        if ((k >> i) & 1) acc = mul(acc, tab.p[i]);
    rp_mont[k] = acc;
    s1[k] = from_mont(acc);
    // coset generator h_c = omega_8192^brp7(c) (cosets.rs:89-112); h_c^64 = omega_128^brp7(c) = w8192[64 * brp7(c)]
    int bc = (int)(__brev((unsigned)cell_idx[k]) >> 25);
    s2[k] = from_mont(mul(acc, w8192[64 * bc]));
}
// weights[row] = sum_{k : row_k == row} r^k  (verifier.rs:216-219), canonical
__global__ void k_verify_weights(const Fr* __restrict__ rp_mont, const int* __restrict__ row, Fr* __restrict__ weights,
                                 int n, int m) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    Fr acc = zero<FrParams>();
    for (int k = 0; k < n; k++)
        if (row[k] == r) acc = add(acc, rp_mont[k]);
    weights[r] = from_mont(acc);
}

// compute_sum_interpolation_poly (verifier.rs:348-384): for every cell k, interpolate its 64 evaluations on the
// coset h_c <omega_64> (bit-reversed inside the cell, so the DIT network takes the cell as stored), un-shift by
// h_c^-i, scale by r^k and accumulate.  One wave per cell; partial[block][64] is folded by k_interp_fold.
__global__ __launch_bounds__(64) void k_interp(const Fr* __restrict__ evals, const int* __restrict__ cell_idx,
                                               const Fr* __restrict__ rp_mont, const Fr* __restrict__ w8192, Fr inv64,
                                               Fr* __restrict__ partial, int n) {
    __shared__ uint32_t s[8][64];
    const int i = threadIdx.x;
    Fr acc = zero<FrParams>();
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        Fr x = evals[(size_t)k * CELL_LEN + i];
#pragma unroll
        for (int l = 0; l < 8; l++) s[l][i] = x.v[l];
        __syncthreads();
        for (int half = 1; half < 64; half <<= 1) {  // DIT, inverse twiddles omega_64^-j = w8192[8192 - 128 j * (32/half)]
            Fr a, b;
            int q = i & 31;
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            if (i < 32) {
#pragma unroll
                for (int l = 0; l < 8; l++) { a.v[l] = s[l][i0]; b.v[l] = s[l][i1]; }
                int e = (N_EXT - j * (N_EXT / (2 * half))) & (N_EXT - 1);
                Fr t = j ? mul(b, w8192[e]) : b;
                Fr u = add(a, t), v = sub(a, t);
#pragma unroll
                for (int l = 0; l < 8; l++) { s[l][i0] = u.v[l]; s[l][i1] = v.v[l]; }
            }
            __syncthreads();
        }
        Fr c;
#pragma unroll
        for (int l = 0; l < 8; l++) c.v[l] = s[l][i];
        __syncthreads();
        int bc = (int)(__brev((unsigned)cell_idx[k]) >> 25);
        Fr hinv_i = w8192[(N_EXT - ((bc * i) & (N_EXT - 1))) & (N_EXT - 1)];  // (h_c^-1)^i, domain.rs:214-223
        c = mul(mul(c, inv64), mul(hinv_i, rp_mont[k]));
        acc = add(acc, c);
    }
    partial[(size_t)blockIdx.x * 64 + i] = acc;
}
__global__ __launch_bounds__(64) void k_interp_fold(const Fr* __restrict__ partial, int nblocks, Fr* __restrict__ out_neg_canon) {
    const int i = threadIdx.x;
    Fr acc = zero<FrParams>();
    for (int b = 0; b < nblocks; b++) acc = add(acc, partial[(size_t)b * 64 + i]);
    out_neg_canon[i] = from_mont(neg(acc));  // the interpolation commitment enters the pairing input with a minus sign
}

// Variable-base lincomb (g1_lincomb): every thread computes k_i * P_i by double-and-add, the block folds its
// 64 products through LDS, out[block] receives the block sum.  Zero scalars / identity points contribute O.
__global__ __launch_bounds__(64) void k_lincomb_partial(const G1Affine* __restrict__ pts, const Fr* __restrict__ sc_canon,
                                                        int n, G1Jac* __restrict__ out) {
    __shared__ G1Jac red[64];
    int i = blockIdx.x * 64 + threadIdx.x;
    G1Jac acc = jac_inf();
    if (i < n) {
        G1Affine p = pts[i];
        Fr k = sc_canon[i];
        if (!is_inf(p)) {
            for (int b = 254; b >= 0; b--) {
                acc = dbl(acc);
                if ((k.v[b >> 5] >> (b & 31)) & 1) acc = add_mixed(acc, p);
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int span = 32; span >= 1; span >>= 1) {
        if (threadIdx.x < span) red[threadIdx.x] = add(red[threadIdx.x], red[threadIdx.x + span]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// out_affine[0] = sum of parts[0..na) ; out_affine[1] = sum of parts[na..na+nb)
__global__ void k_lincomb_final(const G1Jac* __restrict__ parts, int na, int nb, G1Affine* __restrict__ out_affine) {
    int t = threadIdx.x;
    if (t >= 2) return;
    int lo = t == 0 ? 0 : na, hi = t == 0 ? na : na + nb;
    G1Jac acc = jac_inf();
    for (int i = lo; i < hi; i++) acc = add(acc, parts[i]);
    out_affine[t] = to_affine(acc);
}

// ------------------------------------------------------------------------------------------------
// Recovery (reed_solomon.rs:332-384).  Data lives in "cell order" = bit-reversed domain order
// (cosets.rs:177-179), which is exactly what the DIT network consumes and the DIF network produces,
// so the reference's three reverse_bit_order passes disappear.  Z(x) = Z'(x^64) takes one value per cell:
// zeval[c] = Z'(omega_128^brp7(c)), zcinv[c] = 1 / Z'(7^64 omega_128^brp7(c))  (host-computed, 128 each).
//
// k_rec_dit_half: half h of V (optionally multiplied cell-wise by fac[c]) -> DIT_4096 -> T
// grid = (R, 2), block = 1024, LDS 128 KiB.
__global__ __launch_bounds__(1024) void k_rec_dit_half(const Fr* __restrict__ V, const Fr* __restrict__ fac, Fr* __restrict__ T,
                                                      const Fr* __restrict__ w8192) {
    extern __shared__ uint32_t s[];
    const int r = blockIdx.x, h = blockIdx.y;
    const Fr* src = V + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        Fr x = src[e];
        if (fac) x = mul(x, fac[(size_t)r * N_CELLS + ((h * N_BLOB + e) >> 6)]);
        vl_store(s, e, x);
    }
    __syncthreads();
    v_dit_inverse4096(s, w8192);
    Fr* dst = T + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) dst[e] = vl_load(s, e);
}
// last DIT layer of the 8192-point inverse transform + 1/8192 + coset (un)shift by shift[i]  (domain.rs:129-142,214-223).
// final_pass: also check that coefficients 4096.. are zero (reed_solomon.rs:373-380) and emit only the low half to `coeffs`.
__global__ void k_rec_dit_last(const Fr* __restrict__ T, const Fr* __restrict__ shift, Fr n_inv, Fr* __restrict__ U,
                               Fr* __restrict__ coeffs, int* __restrict__ status, const Fr* __restrict__ w8192, int final_pass) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;  // r * 4096 + i
    int r = idx >> 12, i = idx & (N_BLOB - 1);
    const Fr* t = T + (size_t)r * N_EXT;
    Fr a = t[i], b = t[i + N_BLOB];
    Fr tw = i ? mul(b, w8192[N_EXT - i]) : b;
    Fr lo = mul(mul(add(a, tw), n_inv), shift[i]);
    Fr hi = mul(mul(sub(a, tw), n_inv), shift[i + N_BLOB]);
    if (final_pass) {
        if (!is_zero(hi)) atomicOr(&status[r], 4);
        coeffs[(size_t)r * N_BLOB + i] = lo;
    } else {
        U[(size_t)r * N_EXT + i] = lo;
        U[(size_t)r * N_EXT + i + N_BLOB] = hi;
    }
}
// first DIF layer + DIF_4096 on half h, then cell-wise multiply by fac (the inverse vanishing values): U -> V
__global__ __launch_bounds__(1024) void k_rec_dif_half(const Fr* __restrict__ U, const Fr* __restrict__ fac, Fr* __restrict__ V,
                                                      const Fr* __restrict__ w8192) {
    extern __shared__ uint32_t s[];
    const int r = blockIdx.x, h = blockIdx.y;
    const Fr* src = U + (size_t)r * N_EXT;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        Fr a = src[e], b = src[e + N_BLOB];
        Fr x = h ? sub(a, b) : add(a, b);
        if (h && e) x = mul(x, w8192[e]);
        vl_store(s, e, x);
    }
    __syncthreads();
    v_dif_forward4096(s, w8192);
    Fr* dst = V + (size_t)r * N_EXT + (size_t)h * N_BLOB;
    for (int e = threadIdx.x; e < N_BLOB; e += 1024)
        dst[e] = mul(vl_load(s, e), fac[(size_t)r * N_CELLS + ((h * N_BLOB + e) >> 6)]);
}

namespace launch {
static Fr as_fr2(const Fr8& x) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }
void init_attributes_verify() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_dit_half), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_dif_half), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
}
void cells_to_fr(const uint8_t* cells, void* evals, const int* slot_of, int* status, int n, hipStream_t st) {
    k_cells_to_fr<<<(n * CELL_LEN + 255) / 256, 256, 0, st>>>(cells, (Fr*)evals, slot_of, status, n);
}
void verify_scalars(const Fr8* pow_table24, const int* cell_idx, const void* w8192, void* rp_mont, void* s1, void* s2, int n,
                    hipStream_t st) {
    PowTable t;
    for (int i = 0; i < 24; i++) t.p[i] = as_fr2(pow_table24[i]);
    k_verify_scalars<<<(n + 255) / 256, 256, 0, st>>>(t, cell_idx, (const Fr*)w8192, (Fr*)rp_mont, (Fr*)s1, (Fr*)s2, n);
}
void verify_weights(const void* rp_mont, const int* row, void* weights, int n, int m, hipStream_t st) {
    k_verify_weights<<<(m + 63) / 64, 64, 0, st>>>((const Fr*)rp_mont, row, (Fr*)weights, n, m);
}
void interp(const void* evals, const int* cell_idx, const void* rp_mont, const void* w8192, const Fr8& inv64, void* partial,
            int nblocks, void* out_neg_canon, int n, hipStream_t st) {
    k_interp<<<nblocks, 64, 0, st>>>((const Fr*)evals, cell_idx, (const Fr*)rp_mont, (const Fr*)w8192, as_fr2(inv64), (Fr*)partial, n);
    k_interp_fold<<<1, 64, 0, st>>>((const Fr*)partial, nblocks, (Fr*)out_neg_canon);
}
void lincomb_partial(const void* pts, const void* sc, int n, void* out_parts, hipStream_t st) {
    if (n > 0) k_lincomb_partial<<<(n + 63) / 64, 64, 0, st>>>((const G1Affine*)pts, (const Fr*)sc, n, (G1Jac*)out_parts);
}
void lincomb_final(const void* parts, int na, int nb, void* out_affine2, hipStream_t st) {
    k_lincomb_final<<<1, 64, 0, st>>>((const G1Jac*)parts, na, nb, (G1Affine*)out_affine2);
}
void rec_dit_half(int R, const void* V, const void* fac, void* T, const void* w8192, hipStream_t st) {
    k_rec_dit_half<<<dim3(R, 2), 1024, LDS_NTT, st>>>((const Fr*)V, (const Fr*)fac, (Fr*)T, (const Fr*)w8192);
}
void rec_dit_last(int R, const void* T, const void* shift, const Fr8& n_inv, void* U, void* coeffs, int* status,
                  const void* w8192, int final_pass, hipStream_t st) {
    k_rec_dit_last<<<R * N_BLOB / 256, 256, 0, st>>>((const Fr*)T, (const Fr*)shift, as_fr2(n_inv), (Fr*)U, (Fr*)coeffs, status,
                                                    (const Fr*)w8192, final_pass);
}
void rec_dif_half(int R, const void* U, const void* fac, void* V, const void* w8192, hipStream_t st) {
    k_rec_dif_half<<<dim3(R, 2), 1024, LDS_NTT, st>>>((const Fr*)U, (const Fr*)fac, (Fr*)V, (const Fr*)w8192);
}
}  // namespace launch
}  // namespace kzg
