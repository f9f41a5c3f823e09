// Fr kernels: blob decode + 4096-point NTT in LDS, cell encoding, FK20 circulant-column NTTs.
// Reference functions replaced: a1 deserialize, a2 bit reversal, a3 Fr NTT,
// a9 Toeplitz/circulant column FFTs, a11 cell serialisation.
#include "engine.hpp"
#include "kcommon.hpp"
#include "fr29.hpp"
#include "fr29_ntt.hpp"
#include "glv.hpp"
#include "launch.hpp"

namespace kzg {

// ------------------------------------------------------------------------------------------------
// LDS-resident 4096-point NTT in the unsaturated 9 x 29-bit form (fr29.hpp).  Layout: lds[limb * 4096 + index] (limb-major: a
// wave touching consecutive indices touches consecutive banks); 9 limbs = 144 KiB of the CU's 160 KiB.
// Both networks are Cooley-Tukey -- multiply the second input by the twiddle, then add / subtract -- so a value's bound
// grows by 2 r per layer and nothing is reduced inside a transform (bounds at every call site).
__device__ __forceinline__ Fr29 lds_load(const uint32_t* s, int idx) {
    Fr29 r;
#pragma unroll
    for (int l = 0; l < RL; l++) r.v[l] = s[l * N_BLOB + idx];
    return r;
}
__device__ __forceinline__ void lds_store(uint32_t* s, int idx, const Fr29& a) {
#pragma unroll
    for (int l = 0; l < RL; l++) s[l * N_BLOB + idx] = a.v[l];
}

// w29[k] = omega_8192^k in the 9 x 29-bit Montgomery form (canonical), k < 8192.  omega_m^j = w29[j * 8192/m].
//
// RADIX-4 PASSES IN REGISTERS (round 5).  Two layers of the radix-2 network are one pass: a thread loads the four
// elements i0 + {0, h, 2h, 3h} of a unit, runs the two butterflies of the first layer and the two of the second on them in registers,
// and stores four results -- the same four products as before (in a prime field the fourth root of unity is a twiddle like any other:
// radix 4 saves no multiplication), but half the LDS round trips, half the index arithmetic, and the first layer's sums and
// differences stay un-swept (fr29.hpp, LAZY LIMBS: they are what the second layer reads).  Bit-identical to the radix-2 network.
// Thread tid owns unit tid (1024 units of four elements).  Which passes need a block barrier follows from who wrote a unit's
// elements in the pass before (worked out at the two functions): three barriers per transform instead of six.
struct LdsElems {  // the transform's 4096 elements in LDS, limb-major (lds_load / lds_store above)
    uint32_t* s;
    __device__ __forceinline__ Fr29 load(int i) const { return lds_load(s, i); }
    __device__ __forceinline__ void store(int i, const Fr29& v) const { lds_store(s, i, v); }
};
// Inverse transform: input in bit-reversed order -> natural order, inverse twiddles omega^-j, in LDS, 1024 threads.
// Bound: in < B  ->  out < B + 24.
// Pass h (layers half = h, then 2h): unit u = blk * h + j (j < h) holds the elements blk * 4h + j + {0, h, 2h, 3h}; they were written
// in pass h / 4 by the units (4 blk + k) * (h / 4) + (j mod h / 4), k < 4: the same 4-, 16-, 64-thread group for h = 4, 16, 64 (a wave's
// LDS operations execute in program order: no barrier), four different waves for h = 256 and h = 1024 (a block barrier before each).
// The last pass leaves unit j's results at j + 1024 k -- exactly what thread j reads afterwards: no barrier at the end.
__device__ __forceinline__ void dit_inverse_pass(uint32_t* s, const Fr29* __restrict__ w29, int h) {
    ntt4096_dit_inverse_unit(LdsElems{s}, w29, h, (int)threadIdx.x);
}
__device__ __forceinline__ void ntt4096_dit_inverse(uint32_t* s, const Fr29* __restrict__ w29) {  // the caller's barrier stands before
    dit_inverse_pass(s, w29, 1);
    __builtin_amdgcn_wave_barrier();
    dit_inverse_pass(s, w29, 4);
    __builtin_amdgcn_wave_barrier();
    dit_inverse_pass(s, w29, 16);
    __builtin_amdgcn_wave_barrier();
    dit_inverse_pass(s, w29, 64);
    __syncthreads();
    dit_inverse_pass(s, w29, 256);
    __syncthreads();
    dit_inverse_pass(s, w29, 1024);
    __builtin_amdgcn_wave_barrier();  // thread j reads j + 1024 k next: its own unit's results
}
// Forward transform: natural order -> bit-reversed order (position q holds X[brp(q)]), Cooley-Tukey butterflies with the
// twiddles taken in bit-reversed order: at stride `half` the block i = q / half uses omega_4096^(brp(i) * half).
// (Round 2 used Gentleman-Sande butterflies here, whose sum path doubles the bound per layer.)  Bound: in < B -> out < B + 24.
// Pass h (layers half = 2h, then h; log_m = log2(1024 / h) blocks of 4h elements): unit u = blk * h + j holds blk * 4h + j + {0, h, 2h, 3h}.
// The first pass (h = 1024) reads j + 1024 k: what thread j itself stored in the input stage (no barrier before it); the passes h = 256
// and h = 64 read other waves' results (a block barrier before each); h = 16, 4, 1 stay inside the 64-thread group that wrote their
// elements; the callers read across waves afterwards: a block barrier at the end.
__device__ __forceinline__ void ct_forward_pass(uint32_t* s, const Fr29* __restrict__ w29, int h, int log_m) {
    ntt4096_ct_forward_unit(LdsElems{s}, w29, h, log_m, (int)threadIdx.x);
}
__device__ __forceinline__ void ntt4096_ct_forward(uint32_t* s, const Fr29* __restrict__ w29) {  // input stored by thread e mod 1024
    __builtin_amdgcn_wave_barrier();
    ct_forward_pass(s, w29, 1024, 0);
    __syncthreads();
    ct_forward_pass(s, w29, 256, 2);
    __syncthreads();
    ct_forward_pass(s, w29, 64, 4);
    __builtin_amdgcn_wave_barrier();
    ct_forward_pass(s, w29, 16, 6);
    __builtin_amdgcn_wave_barrier();
    ct_forward_pass(s, w29, 4, 8);
    __builtin_amdgcn_wave_barrier();
    ct_forward_pass(s, w29, 1, 10);
    __syncthreads();
}
__device__ __forceinline__ Fr fr_words_of(const Fr29& canonical) {
    Fr r;
    fr29_to_words(r.v, canonical);
    return r;
}

// Stage A+B of compute_cells_and_kzg_proofs: blob bytes -> monomial coefficients.
//   coeffs = IFFT_4096(bit_reverse(blob))   (fk20/prover.rs:177-180, domain.rs:199-211)
// The inverse network wants its input bit-reversed, i.e. exactly the blob as given: no permutation pass.
// grid = n_blobs, block = 1024, dynamic LDS = 144 KiB.  status[b] |= 1 if any element >= r.
// Outputs stay in the engine's stored forms: coeffs = saturated Montgomery (8 x 32 bits, canonical), canon_out = the
// plain integers (MSM scalars), both through the final multiplication by n^-1 with the right constant.
// The blob's elements go through the inverse transform as the PLAIN integers they are (round 5): a product of a plain value with a
// twiddle in Montgomery form is plain again, so the conversion product the input stage had (x * 2^522 / 2^261) is gone, and the
// final product every element has anyway carries the 2^261 instead.
struct NttConsts {
    Fr29 ninv_to_sat;   // n^-1 * 2^256 * 2^261 mod r: mul(x, .) = x n^-1 2^256 = the saturated Montgomery form of the coefficient
    Fr29 ninv_plain;    // n^-1 * 2^261 mod r: mul(x, .) = x n^-1
    Fr29 one;           // 2^261 mod r: mul(x, .) = x (the test kernel's forward direction: reduction only)
};
__global__ __launch_bounds__(1024) void k_blob_to_coeffs(const uint8_t* __restrict__ blobs, Fr* __restrict__ coeffs,
                                                        Fr* __restrict__ canon_out, int* __restrict__ status,
                                                        const Fr29* __restrict__ w29, NttConsts K) {
    extern __shared__ uint32_t s[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* blob = blobs + (size_t)b * BYTES_PER_BLOB;
    bool bad = false;
    for (int e = tid; e < N_BLOB; e += 1024) {
        const Fr x = load_fr_be(blob + 32 * e);
        bad |= geq_mod<FrParams>(x.v);
        lds_store(s, e, fr29_from_plain(x));  // the integer itself: < 2^256 < 3r, normalised limbs
    }
    if (bad) atomicOr(&status[b], 1);
    __syncthreads();
    ntt4096_dit_inverse(s, w29);  // < 27 r
    for (int e = tid; e < N_BLOB; e += 1024) {
        const Fr29 x = lds_load(s, e);
        coeffs[(size_t)b * N_BLOB + e] = fr_words_of(fr29_reduce_once(fr29_mul(x, K.ninv_to_sat)));
        if (canon_out) canon_out[(size_t)b * N_BLOB + e] = fr_words_of(fr29_reduce_once(fr29_mul(x, K.ninv_plain)));
    }
}

// Stage H+I: cells = bit_reverse(NTT_8192(coeffs || 0)) serialised big-endian
// (prover.rs:158-165, serialization/src/lib.rs:132-156).  X[2k+h] = NTT_4096(a_i * w8192^(i*h))[k];
// the forward network leaves half h bit-reversed in place, which is exactly cells[h*4096 ...].
// grid = (n_blobs, 2), block = 1024, dynamic LDS = 144 KiB.
__global__ __launch_bounds__(1024) void k_coeffs_to_cells(const Fr* __restrict__ coeffs, uint8_t* __restrict__ cells,
                                                         const Fr29* __restrict__ w29) {
    extern __shared__ uint32_t s[];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    for (int e = tid; e < N_BLOB; e += 1024) {
        Fr29 c = fr29_from_fr_mont(coeffs[(size_t)b * N_BLOB + e]);  // < 32 r
        if (h && e) c = fr29_mul(c, w29[e]);                          // < 2 r
        lds_store(s, e, c);
    }
    __syncthreads();
    ntt4096_ct_forward(s, w29);  // < 56 r
    uint8_t* out = cells + ((size_t)b * N_EXT + (size_t)h * N_BLOB) * 32;
    const Fr29 one_plain = fr29_const(r29::ONE_PLAIN);
    for (int e = tid; e < N_BLOB; e += 1024)
        store_fr_be(out + 32 * e, fr_words_of(fr29_reduce_once(fr29_mul(lds_load(s, e), one_plain))));  // X / 2^261 = the value
}

// ------------------------------------------------------------------------------------------------
// Stage C: the 64 circulant-column NTT_128 of FK20 (h_poly.rs:36-56, toeplitz.rs:132-144,
// batch_toeplitz.rs:94-106).  For blob b and i < 64 the length-128 vector is
//   v[0] = a[4095-i];  v[1..64] = 0;  v[64+k] = a[64k-1-i], k = 1..63
// scalars[b][j][i] = NTT_128(v)[j] * 128^-1  (the 128^-1 of the later G1 inverse FFT, domain.rs:189-191,
// folded in here because everything downstream is linear), stored as balanced GLV halves (glv.hpp) for the MSM's window extraction.
// grid = n_blobs * 16, block = 256 (4 vectors per block, one per wave).
// segs > 1 (tiny batches only): copies of every scalar multiplied by 2^(128 seg / segs) are written as extra "blobs"
// seg * n + b, so that the MSM stage also delivers 2^32 u, 2^64 u, 2^96 u (four segments; 2^64 u for two) and the
// doubling chain of k_g1circ.hip splits into `segs` independent shorter chains.
struct SegShifts { Fr29 p[3]; };  // the segment shifts as plain integers times 2^261 (this form), p[seg - 1]
__global__ __launch_bounds__(256) void k_fk20_scalars(const Fr* __restrict__ coeffs, Fr* __restrict__ scalars,
                                                      const Fr29* __restrict__ w29, Fr29 scale /* 32 x (128^-1 or 1/2) as an integer: see the input stage */, int n,
                                                      int segs, SegShifts sh) {
    __shared__ uint32_t s[4][RL][128];
    const int b = blockIdx.x >> 4, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = ((blockIdx.x & 15) << 2) + wv;
    const Fr* a = coeffs + (size_t)b * N_BLOB;
    uint32_t(*sv)[128] = s[wv];
    {
        // element `lane` (0..63) and element 64+lane
        Fr29 lo, hi;
#pragma unroll
        for (int l = 0; l < RL; l++) lo.v[l] = hi.v[l] = 0;
        // The vectors go through the transform as PLAIN integers (round 5, as in k_blob_to_coeffs): the stored coefficient is the integer
        // Y = y 2^256, the product with the integer 32 s is Y 32 s / 2^261 = y s, a plain value times a Montgomery-form twiddle stays plain,
        // and the output needs a reduction below r instead of a Montgomery product by one.
        if (lane == 0) lo = fr29_mul(fr29_from_plain(a[N_BLOB - 1 - i]), scale);  // (< r) x (< r) -> < 2r
        else hi = fr29_mul(fr29_from_plain(a[64 * lane - 1 - i]), scale);
#pragma unroll
        for (int l = 0; l < RL; l++) { sv[l][lane] = lo.v[l]; sv[l][64 + lane] = hi.v[l]; }
    }
    __builtin_amdgcn_wave_barrier();
    // natural -> bit-reversed, Cooley-Tukey with bit-reversed twiddles (see ntt4096_ct_forward): < 2 + 14 = 16 r
    int log_m = 0;
    for (int half = 64; half >= 1; half >>= 1, log_m++) {
        const int j = lane & (half - 1), blk = lane / half;
        const int i0 = ((lane - j) << 1) + j, i1 = i0 + half;
        const int e = log_m ? (int)(__brev((unsigned)blk) >> (32 - log_m)) * half * (N_EXT / 128) : 0;  // omega_128 = omega_8192^64
        Fr29 x, y;
#pragma unroll
        for (int l = 0; l < RL; l++) { x.v[l] = sv[l][i0]; y.v[l] = sv[l][i1]; }
        const Fr29 t = e ? fr29_mul(y, w29[e]) : fr29_partial_reduce(y);
        Fr29 sum = fr29_add<false>(x, t), d = fr29_sub2r<false>(x, t);
        if ((log_m & 1) || half == 1) { fr29_normalise(sum); fr29_normalise(d); }  // every second layer and the last one sweep the carries (fr29.hpp, LAZY LIMBS)
#pragma unroll
        for (int l = 0; l < RL; l++) { sv[l][i0] = sum.v[l]; sv[l][i1] = d.v[l]; }
        __builtin_amdgcn_wave_barrier();  // a wave owns its whole vector: its LDS operations execute in program order, no block barrier
    }
    // position q holds NTT[brp7(q)].  The scalar leaves already split k = k1 + k2 lambda for the GLV window tables (what the separate
    // k_glv_split pass of round 2 did in place: 0.19 ms of reading and writing 0.5 GB at 2048 blobs)
    auto emit = [&](size_t at, const Fr29& x) {  // x: the plain value, < 16 r (or < 2r: a segment's product)
        Fr k = fr_words_of(fr29_reduce_once(fr29_partial_reduce(x)));
        uint32_t h[8];
        glv_split_balanced(k, h);
#pragma unroll
        for (int l = 0; l < 8; l++) k.v[l] = h[l];
        scalars[at] = k;
    };
    for (int q = lane; q < 128; q += 64) {
        Fr29 x;
#pragma unroll
        for (int l = 0; l < RL; l++) x.v[l] = sv[l][q];
        const int j = __brev((unsigned)q) >> 25;
        emit(((size_t)b * 128 + j) * 64 + i, x);
        for (int sg = 1; sg < segs; sg++)  // (x sh) is this form again
            emit(((size_t)(sg * n + b) * 128 + j) * 64 + i, fr29_mul(x, sh.p[sg - 1]));  // plain x times the Montgomery form of the shift: plain
    }
}

#ifdef KZG_TEST_HOOKS  // stage-level test kernels: compiled into libc_eth_kzg_hooks.so only (csrc/Makefile)
__global__ __launch_bounds__(1024) void k_test_ntt4096(const uint8_t* in, uint8_t* out, const Fr29* w29, NttConsts K, int inverse_dit) {
    extern __shared__ uint32_t s[];
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) lds_store(s, e, fr29_from_plain(load_fr_be(in + 32 * e)));  // plain values (see NttConsts)
    __syncthreads();
    if (inverse_dit) ntt4096_dit_inverse(s, w29);
    else ntt4096_ct_forward(s, w29);
    for (int e = threadIdx.x; e < N_BLOB; e += 1024) {
        const Fr29 c = lds_load(s, e);
        store_fr_be(out + 32 * e, fr_words_of(fr29_reduce_once(fr29_mul(c, inverse_dit ? K.ninv_plain : K.one))));
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

#endif

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_ntt() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_blob_to_coeffs));
}
static Fr as_fr(const Fr8& x) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }

constexpr size_t LDS_NTT29 = (size_t)N_BLOB * RL * 4;  // 144 KiB: one 4096-point transform of 9-limb elements resident in LDS
void init_attributes() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_blob_to_coeffs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT29);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_coeffs_to_cells), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT29);
#ifdef KZG_TEST_HOOKS
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_test_ntt4096), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NTT29);
#endif
}
// host-side constants of the 9 x 29-bit form from the engine's saturated Montgomery values (Y = y 2^256 mod r, canonical)
// X = y 2^261 mod r = 32 Y mod r: in Fr arithmetic, the Montgomery form of (y * 32) read as a plain integer
static Fr29 fr29_mont_of(const Fr& y_mont) {
    Fr c32 = zero<FrParams>();
    c32.v[0] = 32;
    return fr29_from_plain(mul(y_mont, to_mont(c32)));
}
static Fr29 fr29_mont_of(const Fr8& y_mont) { return fr29_mont_of(as_fr(y_mont)); }
// the INTEGER 32 s mod r for a value s given in the saturated Montgomery form: mul(Y, .) = y s for a stored Y = y 2^256 (k_fk20_scalars)
static Fr29 fr29_int32x_of(const Fr8& s_mont) {
    Fr c32 = zero<FrParams>();
    c32.v[0] = 32;
    return fr29_from_plain(from_mont(mul(as_fr(s_mont), to_mont(c32))));
}
static NttConsts ntt_consts(const Fr8& n_inv_mont) {
    NttConsts K;
    K.ninv_to_sat = fr29_mont_of(to_mont(as_fr(n_inv_mont)));     // (n^-1 2^256) 2^261: the Montgomery form of the INTEGER n^-1 2^256 mod r, times 32
    K.ninv_plain = fr29_mont_of(n_inv_mont);                      // n^-1 2^261
    K.one = fr29_mont_of(one<FrParams>());                        // 2^261 (one<>() is the Montgomery form of 1)
    return K;
}
void ntt_twiddles29(const void* w8192_mont_host /*Fr[8192]*/, void* out_host /*8192 x 9 words*/) {
    const Fr* w = (const Fr*)w8192_mont_host;
    Fr29* o = (Fr29*)out_host;
    for (int k = 0; k < N_EXT; k++) o[k] = fr29_mont_of(*(const Fr8*)&w[k]);
}
void blob_to_coeffs(int n, const uint8_t* blobs, void* coeffs, void* canon, int* status, const void* w29, const Fr8& n_inv, hipStream_t st) {
    k_blob_to_coeffs<<<n, 1024, LDS_NTT29, st>>>(blobs, (Fr*)coeffs, (Fr*)canon, status, (const Fr29*)w29, ntt_consts(n_inv));
}
void coeffs_to_cells(int n, const void* coeffs, uint8_t* cells, const void* w29, hipStream_t st) {
    k_coeffs_to_cells<<<dim3(n, 2), 1024, LDS_NTT29, st>>>((const Fr*)coeffs, cells, (const Fr29*)w29);
}
void fk20_scalars(int n, const void* coeffs, void* scalars, const void* w29, const Fr8& inv128, int segs, const Fr8* seg_shifts,
                  hipStream_t st) {
    SegShifts sh;
    for (int i = 0; i < 3; i++) sh.p[i] = fr29_mont_of(seg_shifts[i]);
    k_fk20_scalars<<<n * 16, 256, 0, st>>>((const Fr*)coeffs, (Fr*)scalars, (const Fr29*)w29, fr29_int32x_of(inv128), n, segs, sh);
}
#ifdef KZG_TEST_HOOKS
void test_ntt4096(const uint8_t* in, uint8_t* out, const void* w29, const Fr8& n_inv, int inverse_dit, hipStream_t st) {
    k_test_ntt4096<<<1, 1024, LDS_NTT29, st>>>(in, out, (const Fr29*)w29, ntt_consts(n_inv), inverse_dit);
}
void test_scalars_be(const uint8_t* in, void* out, size_t n, hipStream_t st) {
    k_test_scalars_be<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, (Fr*)out, n);
}
void test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp, hipStream_t st) {
    if (is_fp) k_test_field_mul<Fp><<<(n + 63) / 64, 64, 0, st>>>(a, b, out, n);
    else k_test_field_mul<Fr><<<(n + 63) / 64, 64, 0, st>>>(a, b, out, n);
}
#endif
}  // namespace launch
}  // namespace kzg
