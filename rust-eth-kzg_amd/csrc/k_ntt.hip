// Fr kernels: blob decode + 4096-point NTT in LDS, cell encoding, FK20 circulant-column NTTs.
// Reference functions replaced: a1 deserialize, a2 bit reversal, a3 Fr NTT,
// a9 Toeplitz/circulant column FFTs, a11 cell serialisation.
#include "engine.hpp"
#include "kcommon.hpp"
#include "launch.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------------
// LDS-resident 4096-point NTT.  Layout: lds[limb * 4096 + index] (limb-major: a wave touching
// consecutive indices touches consecutive banks).
__device__ __forceinline__ Fr lds_load(const uint32_t* s, int idx) {
    Fr r;
#pragma unroll
    for (int l = 0; l < 8; l++) r.v[l] = s[l * N_BLOB + idx];
    return r;
}
__device__ __forceinline__ void lds_store(uint32_t* s, int idx, const Fr& a) {
#pragma unroll
    for (int l = 0; l < 8; l++) s[l * N_BLOB + idx] = a.v[l];
}

// w8192[k] = omega_8192^k (Montgomery), k < 8192.  omega_m^j = w8192[j * 8192/m].
// DIT (input in bit-reversed order -> natural order), inverse twiddles, in LDS, 1024 threads.
__device__ __forceinline__ void ntt4096_dit_inverse(uint32_t* s, const Fr* __restrict__ w8192) {
    const int tid = threadIdx.x;
    for (int half = 1; half < N_BLOB; half <<= 1) {
        const int tw_step = N_EXT / (2 * half);  // exponent step in units of omega_8192
        for (int q = tid; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = lds_load(s, i0), b = lds_load(s, i1);
            int e = (N_EXT - j * tw_step) & (N_EXT - 1);  // omega^-j
            Fr t = j ? mul(b, w8192[e]) : b;
            lds_store(s, i0, add(a, t));
            lds_store(s, i1, sub(a, t));
        }
        __syncthreads();
    }
}
// DIF (natural order -> bit-reversed order), forward twiddles.
__device__ __forceinline__ void ntt4096_dif_forward(uint32_t* s, const Fr* __restrict__ w8192) {
    const int tid = threadIdx.x;
    for (int half = N_BLOB / 2; half >= 1; half >>= 1) {
        const int tw_step = N_EXT / (2 * half);
        for (int q = tid; q < N_BLOB / 2; q += 1024) {
            int j = q & (half - 1);
            int i0 = ((q - j) << 1) + j, i1 = i0 + half;
            Fr a = lds_load(s, i0), b = lds_load(s, i1);
            Fr d = sub(a, b);
            lds_store(s, i0, add(a, b));
            lds_store(s, i1, j ? mul(d, w8192[j * tw_step]) : d);
        }
        __syncthreads();
    }
}

// Stage A+B of compute_cells_and_kzg_proofs: blob bytes -> monomial coefficients.
//   coeffs = IFFT_4096(bit_reverse(blob))   (fk20/prover.rs:177-180, domain.rs:199-211)
// The DIT network wants its input bit-reversed, i.e. exactly the blob as given: no permutation pass.
// grid = n_blobs, block = 1024, dynamic LDS = 128 KiB.  status[b] |= 1 if any element >= r.
// If canon_out != nullptr the coefficients are also written out of Montgomery form (MSM scalars).
__global__ __launch_bounds__(1024) void k_blob_to_coeffs(const uint8_t* __restrict__ blobs, Fr* __restrict__ coeffs,
                                                        Fr* __restrict__ canon_out, int* __restrict__ status,
                                                        const Fr* __restrict__ w8192, Fr n_inv) {
    extern __shared__ uint32_t s[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* blob = blobs + (size_t)b * BYTES_PER_BLOB;
    bool bad = false;
    for (int e = tid; e < N_BLOB; e += 1024) {
        Fr x = load_fr_be(blob + 32 * e);
        bad |= geq_mod<FrParams>(x.v);
        lds_store(s, e, to_mont(x));
    }
    if (bad) atomicOr(&status[b], 1);
    __syncthreads();
    ntt4096_dit_inverse(s, w8192);
    for (int e = tid; e < N_BLOB; e += 1024) {
        Fr c = mul(lds_load(s, e), n_inv);
        coeffs[(size_t)b * N_BLOB + e] = c;
        if (canon_out) canon_out[(size_t)b * N_BLOB + e] = from_mont(c);
    }
}

// Stage H+I: cells = bit_reverse(NTT_8192(coeffs || 0)) serialised big-endian
// (prover.rs:158-165, serialization/src/lib.rs:132-156).  X[2k+h] = NTT_4096(a_i * w8192^(i*h))[k];
// the DIF network leaves half h bit-reversed in place, which is exactly cells[h*4096 ...].
// grid = (n_blobs, 2), block = 1024, dynamic LDS = 128 KiB.
__global__ __launch_bounds__(1024) void k_coeffs_to_cells(const Fr* __restrict__ coeffs, uint8_t* __restrict__ cells,
                                                         const Fr* __restrict__ w8192) {
    extern __shared__ uint32_t s[];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    for (int e = tid; e < N_BLOB; e += 1024) {
        Fr c = coeffs[(size_t)b * N_BLOB + e];
        if (h && e) c = mul(c, w8192[e]);
        lds_store(s, e, c);
    }
    __syncthreads();
    ntt4096_dif_forward(s, w8192);
    uint8_t* out = cells + ((size_t)b * N_EXT + (size_t)h * N_BLOB) * 32;
    for (int e = tid; e < N_BLOB; e += 1024) store_fr_be(out + 32 * e, from_mont(lds_load(s, e)));
}

// Generic LDS NTT of NPTS (<= 8192/ (threads..)) is not needed: recovery reuses the 4096 kernels through
// the split X[2k+h] identity; see k_ntt8192_* in engine.hip.

// ------------------------------------------------------------------------------------------------
// Stage C: the 64 circulant-column NTT_128 of FK20 (h_poly.rs:36-56, toeplitz.rs:132-144,
// batch_toeplitz.rs:94-106).  For blob b and i < 64 the length-128 vector is
//   v[0] = a[4095-i];  v[1..64] = 0;  v[64+k] = a[64k-1-i], k = 1..63
// scalars[b][j][i] = NTT_128(v)[j] * 128^-1  (the 128^-1 of the later G1 inverse FFT, domain.rs:189-191,
// folded in here because everything downstream is linear), stored OUT of Montgomery form for the
// MSM's window extraction.  grid = n_blobs * 16, block = 256 (4 vectors per block, one per wave).
// segs > 1 (tiny batches only): copies of every scalar multiplied by 2^(128 seg / segs) are written as extra "blobs"
// seg * n + b, so that the MSM stage also delivers 2^32 u, 2^64 u, 2^96 u (four segments; 2^64 u for two) and the
// doubling chain of k_g1circ.hip splits into `segs` independent shorter chains.
struct SegShifts { Fr p[3]; };  // Montgomery forms of the segment shifts, p[seg - 1]
__global__ __launch_bounds__(256) void k_fk20_scalars(const Fr* __restrict__ coeffs, Fr* __restrict__ scalars,
                                                      const Fr* __restrict__ w8192, Fr inv128, int n, int segs, SegShifts sh) {
    __shared__ uint32_t s[4][8][128];
    const int b = blockIdx.x >> 4, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = ((blockIdx.x & 15) << 2) + wv;
    const Fr* a = coeffs + (size_t)b * N_BLOB;
    uint32_t(*sv)[128] = s[wv];
    {
        // element `lane` (0..63) and element 64+lane
        Fr lo = zero<FrParams>(), hi = zero<FrParams>();
        if (lane == 0) lo = mul(a[N_BLOB - 1 - i], inv128);
        else hi = mul(a[64 * lane - 1 - i], inv128);
#pragma unroll
        for (int l = 0; l < 8; l++) { sv[l][lane] = lo.v[l]; sv[l][64 + lane] = hi.v[l]; }
    }
    __syncthreads();
    // DIF: natural -> bit-reversed
    for (int half = 64; half >= 1; half >>= 1) {
        int j = lane & (half - 1);
        int i0 = ((lane - j) << 1) + j, i1 = i0 + half;
        Fr x, y;
#pragma unroll
        for (int l = 0; l < 8; l++) { x.v[l] = sv[l][i0]; y.v[l] = sv[l][i1]; }
        Fr d = sub(x, y), sum = add(x, y);
        if (j) d = mul(d, w8192[j * (N_EXT / (2 * half))]);
#pragma unroll
        for (int l = 0; l < 8; l++) { sv[l][i0] = sum.v[l]; sv[l][i1] = d.v[l]; }
        __syncthreads();
    }
    // position q holds NTT[brp7(q)]
    for (int q = lane; q < 128; q += 64) {
        Fr x;
#pragma unroll
        for (int l = 0; l < 8; l++) x.v[l] = sv[l][q];
        int j = __brev((unsigned)q) >> 25;
        scalars[((size_t)b * 128 + j) * 64 + i] = from_mont(x);
        for (int sg = 1; sg < segs; sg++)
            scalars[((size_t)(sg * n + b) * 128 + j) * 64 + i] = from_mont(mul(x, sh.p[sg - 1]));
    }
}

__global__ __launch_bounds__(1024) void k_test_ntt4096(const uint8_t* in, uint8_t* out, const Fr* w8192, Fr n_inv, int inverse_dit) {
    extern __shared__ uint32_t s[];
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) lds_store(s, e, to_mont(load_fr_be(in + 32 * e)));
    __syncthreads();
    if (inverse_dit) ntt4096_dit_inverse(s, w8192);
    else ntt4096_dif_forward(s, w8192);
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        Fr c = lds_load(s, e);
        if (inverse_dit) c = mul(c, n_inv);
        store_fr_be(out + 32 * e, from_mont(c));
    }
}
__global__ void k_test_scalars_be(const uint8_t* in, Fr* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = load_fr_be(in + 32 * i);
}
template <class F>
__global__ void k_test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NB = F::N * 4;
    F x, y;
    for (int k = 0; k < F::N; k++) {
        const uint8_t* pa = a + (size_t)i * NB + 4 * k;
        const uint8_t* pb = b + (size_t)i * NB + 4 * k;
        x.v[F::N - 1 - k] = ((uint32_t)pa[0] << 24) | ((uint32_t)pa[1] << 16) | ((uint32_t)pa[2] << 8) | pa[3];
        y.v[F::N - 1 - k] = ((uint32_t)pb[0] << 24) | ((uint32_t)pb[1] << 16) | ((uint32_t)pb[2] << 8) | pb[3];
    }
    F z = from_mont(mul(to_mont(x), to_mont(y)));
    for (int k = 0; k < F::N; k++) {
        uint32_t w = z.v[F::N - 1 - k];
        uint8_t* po = out + (size_t)i * NB + 4 * k;
        po[0] = (uint8_t)(w >> 24); po[1] = (uint8_t)(w >> 16); po[2] = (uint8_t)(w >> 8); po[3] = (uint8_t)w;
    }
}

namespace launch {
static Fr as_fr(const Fr8& x) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }

void init_attributes() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_blob_to_coeffs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_coeffs_to_cells), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_test_ntt4096), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT);
}
void blob_to_coeffs(int n, const uint8_t* blobs, void* coeffs, void* canon, int* status, const void* w8192, const Fr8& n_inv, hipStream_t st) {
    k_blob_to_coeffs<<<n, 1024, LDS_NTT, st>>>(blobs, (Fr*)coeffs, (Fr*)canon, status, (const Fr*)w8192, as_fr(n_inv));
}
void coeffs_to_cells(int n, const void* coeffs, uint8_t* cells, const void* w8192, hipStream_t st) {
    k_coeffs_to_cells<<<dim3(n, 2), 1024, LDS_NTT, st>>>((const Fr*)coeffs, cells, (const Fr*)w8192);
}
void fk20_scalars(int n, const void* coeffs, void* scalars, const void* w8192, const Fr8& inv128, int segs, const Fr8* seg_shifts,
                  hipStream_t st) {
    SegShifts sh;
    for (int i = 0; i < 3; i++) sh.p[i] = as_fr(seg_shifts[i]);
    k_fk20_scalars<<<n * 16, 256, 0, st>>>((const Fr*)coeffs, (Fr*)scalars, (const Fr*)w8192, as_fr(inv128), n, segs, sh);
}
void test_ntt4096(const uint8_t* in, uint8_t* out, const void* w8192, const Fr8& n_inv, int inverse_dit, hipStream_t st) {
    k_test_ntt4096<<<1, 1024, LDS_NTT, st>>>(in, out, (const Fr*)w8192, as_fr(n_inv), inverse_dit);
}
void test_scalars_be(const uint8_t* in, void* out, size_t n, hipStream_t st) {
    k_test_scalars_be<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, (Fr*)out, n);
}
void test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp, hipStream_t st) {
    if (is_fp) k_test_field_mul<Fp><<<(n + 63) / 64, 64, 0, st>>>(a, b, out, n);
    else k_test_field_mul<Fr><<<(n + 63) / 64, 64, 0, st>>>(a, b, out, n);
}
}  // namespace launch
}  // namespace kzg
