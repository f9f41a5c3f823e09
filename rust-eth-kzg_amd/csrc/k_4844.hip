// EIP-4844 single-point openings: quotient of the blob polynomial by (X - z) and its value at z.
// Reference: divide_by_linear (crates/cryptography/kzg_single_open/src/prover.rs:48-65, a sequential Ruffini
// recurrence) and PolyCoeff::eval (crates/cryptography/polynomial/src/poly_coeff.rs:59-65).
#include "engine.hpp"
#include "kcommon.hpp"
#include "launch.hpp"

namespace kzg {

// With H_k = sum_{j >= k} c_j z^(j-k) (so H_k = c_k + z H_{k+1}):  y = H_0 and quotient q_k = H_{k+1}.
// The recurrence is cut into 64 chunks of 64 coefficients: every lane runs the recurrence inside its chunk,
// lane 0 stitches the 64 chunk heads together with z^64, and every lane adds its carry-in times z^(distance).
// grid = n_blobs, block = 64.  quotient: [b][4096] canonical Fr (entry 4095 = 0) = the scalars of the MSM against
// the monomial SRS; y_out: [b] canonical.
__global__ __launch_bounds__(64) void k_quotient_by_linear(const Fr* __restrict__ coeffs, const Fr* __restrict__ z_mont,
                                                           Fr* __restrict__ quotient, Fr* __restrict__ y_out) {
    __shared__ Fr head[65];
    const int b = blockIdx.x, L = threadIdx.x;
    const Fr* c = coeffs + (size_t)b * N_BLOB;
    Fr* q = quotient + (size_t)b * N_BLOB;
    const Fr z = z_mont[b];
    // local recurrence, parked (Montgomery) in the output slots: slot k holds the chunk-local H_k
    Fr h = zero<FrParams>();
    for (int k = 64 * L + 63; k >= 64 * L; k--) {
        h = add(c[k], mul(z, h));
        q[k] = h;
    }
    head[L] = h;
    if (L == 0) head[64] = zero<FrParams>();
    __syncthreads();
    if (L == 0) {
        Fr z64 = z;
        for (int i = 0; i < 6; i++) z64 = sqr(z64);
        for (int l = 62; l >= 0; l--) head[l] = add(head[l], mul(z64, head[l + 1]));  // true H at every chunk boundary
    }
    __syncthreads();
    const Fr carry = head[L + 1];  // H_{64(L+1)}
    Fr pw = z;
    // second sweep, top-down: true H_k = local + z^(64L+64-k) * carry, kept in place (Montgomery) for now
    for (int k = 64 * L + 63; k >= 64 * L; k--) {
        q[k] = add(q[k], mul(pw, carry));
        pw = mul(pw, z);
    }
    __syncthreads();
    // shift by one and leave Montgomery form: q_k = H_{k+1}
    // (each lane only needs one value from its upper neighbour: read it before anyone overwrites)
    const Fr next_head = L < 63 ? q[64 * L + 64] : zero<FrParams>();
    const Fr y = q[0];
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        const int k = 64 * L + i;
        const Fr v = i < 63 ? q[k + 1] : next_head;  // q[k+1] of this lane's own chunk is still H_{k+1}: ascending order
        q[k] = from_mont(v);
    }
    if (L == 0) y_out[b] = from_mont(y);
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_4844() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_quotient_by_linear));
}
void quotient_by_linear(int n, const void* coeffs, const void* z_mont, void* quotient, void* y_out, hipStream_t st) {
    k_quotient_by_linear<<<n, 64, 0, st>>>((const Fr*)coeffs, (const Fr*)z_mont, (Fr*)quotient, (Fr*)y_out);
}
}  // namespace launch
}  // namespace kzg
