// tools/ubench_fp30.hip -- multiplication / squaring / fused-pair rates of the signed 13 x 30-bit field (csrc/fp30.hpp)
// next to the unsigned 14 x 29-bit field (csrc/fp29.hpp, verbatim chains) in ONE binary on ONE GPU, at 2 and 4 waves per SIMD.
// VERDICT r4's kill criterion for the 13-digit form: less than +8 % multiplications/s at 2 waves/SIMD -> stop.
//   hipcc -O3 -fwrapv -std=c++17 --offload-arch=gfx950 -DFQ_ASM_MAC -I rust-eth-kzg_amd/csrc tools/ubench_fp30.hip -o tools/ubench_fp30
//   (-fwrapv: the biased-unsigned variant keeps S + 2^63 in an int64_t)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "fp29.hpp"
#include "fp30.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 512;

template <int OP>
__global__ __launch_bounds__(256, 2) void k29(uint32_t* out, uint32_t seed) {
    Fq<2> x, y;
    for (int i = 0; i < QL; i++) { x.v[i] = (threadIdx.x * 2654435761u + seed + i * 977u) & QMASK; y.v[i] = (x.v[i] ^ 0x9e3779bu) & QMASK; }
    x.v[QL - 1] &= 0xff; y.v[QL - 1] &= 0xff;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { x = mul(x, y); y = mul(y, x); }
        else if (OP == 1) { x = sqr(x); y = sqr(y); }
        else { x = mul_add(x, y, y, x); y = mul_add(y, x, x, y); }
    }
    uint32_t h = 0;
    for (int i = 0; i < QL; i++) h ^= x.v[i] ^ y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// OP: 0 mul C x C -> C, 1 mul C x U -> U, 2 sqr -> C, 3 mul_add four centred, 4 mul_add split (wide operands)
template <int OP>
__global__ __launch_bounds__(256, 2) void k30(uint32_t* out, uint32_t seed) {
    Fs<1, DC> x, y;
    for (int i = 0; i < SL; i++) { x.v[i] = (int32_t)((threadIdx.x * 2654435761u + seed + i * 977u) & (uint32_t)SMASK) - SHALF; y.v[i] = (x.v[i] ^ 0x1e3779b) % SHALF; }
    x.v[SL - 1] &= 0xff; y.v[SL - 1] &= 0xff;
    Fs<1, DU> xu, yu;
    for (int i = 0; i < SL; i++) { xu.v[i] = x.v[i] & SMASK; yu.v[i] = y.v[i] & SMASK; }
    xu.v[SL - 1] &= 0xff; yu.v[SL - 1] &= 0xff;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { x = mul(x, y); y = mul(y, x); }
        else if (OP == 1) { xu = mul<DU>(x, xu); yu = mul<DU>(y, yu); }
        else if (OP == 2) { x = sqr(x); y = sqr(y); }
        else if (OP == 3) { x = mul_add(x, y, y, x); y = mul_add(y, x, x, y); }
        else if (OP == 4) { xu = mul_add<DU>(x, xu, yu, y); yu = mul_add<DU>(y, yu, xu, x); }
    }
    uint32_t h = 0;
    for (int i = 0; i < SL; i++) h ^= x.v[i] ^ y.v[i] ^ xu.v[i] ^ yu.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// the fused forms keep their widened bounds, so they get their own loop bodies (results folded back with a memcpy-like relax)
template <int OP>
__global__ __launch_bounds__(256, 2) void k30f(uint32_t* out, uint32_t seed) {
    Fs<1, DC> x, y;
    for (int i = 0; i < SL; i++) { x.v[i] = (int32_t)((threadIdx.x * 2654435761u + seed + i * 977u) & (uint32_t)SMASK) - SHALF; y.v[i] = (x.v[i] ^ 0x1e3779b) % SHALF; }
    x.v[SL - 1] &= 0xff; y.v[SL - 1] &= 0xff;
    Fs<1, DU> xu, yu;
    for (int i = 0; i < SL; i++) { xu.v[i] = x.v[i] & SMASK; yu.v[i] = y.v[i] & SMASK; }
    xu.v[SL - 1] &= 0xff; yu.v[SL - 1] &= 0xff;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {  // product minus stored value, centred out
            auto t = mul_inj<-1>(x, xu, yu);
            auto u = mul_inj<-1>(y, yu, xu);
            for (int k = 0; k < SL; k++) { x.v[k] = t.v[k]; y.v[k] = u.v[k]; }
        } else if (OP == 1) {  // square minus two stored values, floor digits out
            auto t = sqr_inj2<-1, -2, DU>(x, xu, yu);
            auto u = sqr_inj2<-1, -2, DU>(y, yu, xu);
            for (int k = 0; k < SL; k++) { xu.v[k] = t.v[k]; yu.v[k] = u.v[k]; x.v[k] ^= t.v[k] & 1; y.v[k] ^= u.v[k] & 1; }
        } else {  // normalise of a lazy difference
            auto t = normalise(sub_lazy(xu, yu));
            auto u = normalise(sub_lazy(yu, xu));
            for (int k = 0; k < SL; k++) { xu.v[k] = t.v[k] & SMASK; yu.v[k] = (u.v[k] + k) & SMASK; }
        }
    }
    uint32_t h = 0;
    for (int i = 0; i < SL; i++) h ^= x.v[i] ^ y.v[i] ^ xu.v[i] ^ yu.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// BIASED-UNSIGNED column accumulators.  The 13 x 30-bit design already uses the signed 64-bit range (a column
// of a C x C product reaches +-1.17 * 2^62, of a C x W product +-1.92 * 2^62, fp30.hpp), so the only bias there is room for is the
// unsigned mid-point: a column is kept as U = S + 2^63 in [0, 2^64) (the multiply-add's 64-bit addition is the same instruction
// either way), shifted LOGICALLY, and given back what the shift took: U' = (U >> 30) + 2^63 - 2^33.  Both constants have zero low
// 32 bits, so the quotient digit, the extracted digits and the top digit are untouched.  Same results as mul() (checked per lane
// below), 26 more 64-bit additions per product.  (It cannot keep the upper bits quiet either -- U crosses 2^63 whenever S
// changes sign -- so what this measures is the price of the idea, with nothing on the other side of the scale.)
namespace biased {
using namespace kzg::q30core;
constexpr uint64_t BIAS = 1ull << 63, REBIAS = BIAS - (BIAS >> 30);
__device__ __forceinline__ void shift_b(int64_t& acc) { acc = (int64_t)(((uint64_t)acc >> 30) + REBIAS); }
template <int K>
__device__ __forceinline__ void lo_col_b(int64_t& acc, const ProdMul& pr, int32_t* m) {
    pr.template col<K>(acc);
    run_vp<K, K>(acc, m);
    m[K] = (int32_t)((uint32_t)acc * q30::N0Q) >> 2;
    run_vp<1, 0>(acc, m + K);
    shift_b(acc);
}
template <int K, int OUTF>
__device__ __forceinline__ void hi_col_b(int64_t& acc, const ProdMul& pr, const int32_t* m, int32_t* r) {
    constexpr int lo = K - SL + 1, n = SL - lo, J = K - SL;
    pr.template col<K>(acc);
    run_vp<n, K - lo>(acc, m + lo);
    if constexpr (J == SL - 1) r[J] = (int32_t)acc;
    else if constexpr (OUTF == DC) {
        acc = (int64_t)((uint64_t)acc + (uint64_t)SHALF);
        r[J] = ((int32_t)acc & SMASK) - SHALF;
        shift_b(acc);
    } else {
        r[J] = (int32_t)acc & SMASK;
        shift_b(acc);
    }
}
template <int OUTF, int... Ks>
__device__ __forceinline__ void mont_b(const ProdMul& pr, int32_t* r, std::integer_sequence<int, Ks...>) {
    int32_t m[SL];
    int64_t acc = (int64_t)BIAS;
    (lo_col_b<Ks>(acc, pr, m), ...);
    (hi_col_b<SL + Ks, OUTF>(acc, pr, m, r), ...);
}
template <int OUTF = DC, int A, int FA, int B, int FB>
__device__ __forceinline__ Fs<1, OUTF> mul_b(const Fs<A, FA>& a, const Fs<B, FB>& b) {
    Fs<1, OUTF> r;
    mont_b<OUTF>(ProdMul{a.v, b.v}, r.v, Seq{});
    return r;
}
}  // namespace biased
// OP 0: C x C -> centred, 1: C x U -> floor digits -- the loops of k30<0> / k30<1> on the biased accumulator
template <int OP>
__global__ __launch_bounds__(256, 2) void k30b(uint32_t* out, uint32_t seed) {
    Fs<1, DC> x, y;
    for (int i = 0; i < SL; i++) { x.v[i] = (int32_t)((threadIdx.x * 2654435761u + seed + i * 977u) & (uint32_t)SMASK) - SHALF; y.v[i] = (x.v[i] ^ 0x1e3779b) % SHALF; }
    x.v[SL - 1] &= 0xff; y.v[SL - 1] &= 0xff;
    Fs<1, DU> xu, yu;
    for (int i = 0; i < SL; i++) { xu.v[i] = x.v[i] & SMASK; yu.v[i] = y.v[i] & SMASK; }
    xu.v[SL - 1] &= 0xff; yu.v[SL - 1] &= 0xff;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { x = biased::mul_b(x, y); y = biased::mul_b(y, x); }
        else { xu = biased::mul_b<DU>(x, xu); yu = biased::mul_b<DU>(y, yu); }
    }
    uint32_t h = 0;
    for (int i = 0; i < SL; i++) h ^= x.v[i] ^ y.v[i] ^ xu.v[i] ^ yu.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// every lane: one product both ways on its own operands, digit by digit (the two loops above diverge after the first difference, so a
// hash of their end states says nothing about WHERE)
__global__ void k_check_biased(uint32_t* bad, uint32_t seed) {
    Fs<1, DC> x, y;
    for (int i = 0; i < SL; i++) { x.v[i] = (int32_t)(((blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed + i * 977u) & (uint32_t)SMASK) - SHALF; y.v[i] = (x.v[i] ^ 0x1e3779b) % SHALF; }
    x.v[SL - 1] &= 0xff; y.v[SL - 1] &= 0xff;
    Fs<1, DU> xu;
    for (int i = 0; i < SL; i++) xu.v[i] = y.v[i] & SMASK;
    xu.v[SL - 1] &= 0xff;
    const Fs<1, DC> a = mul(x, y), b = biased::mul_b(x, y);
    const Fs<1, DU> c = mul<DU>(x, xu), d = biased::mul_b<DU>(x, xu);
    bool differ = false;
    for (int i = 0; i < SL; i++) differ |= a.v[i] != b.v[i] || c.v[i] != d.v[i];
    if (differ) atomicAdd(bad, 1u);
}
static unsigned biased_mismatches(uint32_t* out) {
    CK(hipMemset(out, 0, 4));
    k_check_biased<<<64, 256>>>(out, 7u);
    unsigned bad = 0;
    CK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
    return bad;
}
static bool g_sustained = false;  // --sustained: every kernel runs back to back for 1.5 s and the rate of the last second counts --
                                  // the point kernels are POWER-limited on MI355X (1.27 kW, the clock settles near 2.1 of 2.4 GHz), so
                                  // a 2 ms launch measures the issue rate at full clock, not the rate the chip sustains
template <class K>
double run_sustained(const char* name, int blocks, K kern, uint32_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, 256>>>(out, 1u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, 1u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float one; CK(hipEventElapsedTime(&one, e0, e1));
    const int warm = (int)(500.0f / one) + 1, timed = (int)(1000.0f / one) + 1;
    for (int r = 0; r < warm; r++) kern<<<blocks, 256>>>(out, 1u);
    CK(hipEventRecord(e0));
    for (int r = 0; r < timed; r++) kern<<<blocks, 256>>>(out, 1u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double g = 2.0 * ITER * blocks * 256 * (double)timed / (ms * 1e-3) * 1e-9;
    printf("%-52s blocks=%5d  %8.3f ms/launch sustained (cold %7.3f)  %8.2f G op/s\n", name, blocks, ms / timed, one, g);
    return g;
}
template <class K>
double run(const char* name, int blocks, K kern, uint32_t* out) {
    if (g_sustained) return run_sustained(name, blocks, kern, out);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, 256>>>(out, 1u); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, 1u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double g = 2.0 * ITER * blocks * 256 / (best * 1e-3) * 1e-9;
    printf("%-52s blocks=%5d  %8.3f ms  %8.2f G op/s\n", name, blocks, best, g);
    return g;
}
int main(int argc, char** argv) {
    g_sustained = argc > 1 && !strcmp(argv[1], "--sustained");
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    uint32_t* out; CK(hipMalloc(&out, 1 << 26));
    for (int wps : {2, 4}) {
        int blocks = prop.multiProcessorCount * wps;
        printf("--- %d waves per SIMD requested ---\n", wps);
        const double m29 = run("fp29 mul (392+14 MACs)", blocks, k29<0>, out);
        const double s29 = run("fp29 sqr (301 MACs)", blocks, k29<1>, out);
        const double p29 = run("fp29 mul_add a*b+c*d (588 MACs)", blocks, k29<2>, out);
        const double m30 = run("fp30 mul C x C -> centred (338+13 MACs)", blocks, k30<0>, out);
        const double u30 = run("fp30 mul C x U -> floor digits", blocks, k30<1>, out);
        const double mb30 = run("fp30 mul C x C, BIASED accumulator", blocks, k30b<0>, out);
        const double ub30 = run("fp30 mul C x U, BIASED accumulator", blocks, k30b<1>, out);
        printf("biased / plain at %d waves/SIMD: C x C %.3f  C x U %.3f   (products that differ from mul(): %u of 16384)\n", wps, mb30 / m30, ub30 / u30,
               biased_mismatches(out));
        const double s30 = run("fp30 sqr -> centred (260 MACs)", blocks, k30<2>, out);
        const double p30 = run("fp30 mul_add, four centred operands (507 MACs)", blocks, k30<3>, out);
        const double q30 = run("fp30 mul_add, wide operands: split columns", blocks, k30<4>, out);
        const double i30 = run("fp30 mul_inj (a b - x), centred out", blocks, k30f<0>, out);
        const double j30 = run("fp30 sqr_inj2 (a^2 - x - 2 y), floor digits out", blocks, k30f<1>, out);
        run("fp30 normalise(lazy difference)", blocks, k30f<2>, out);
        printf("ratio fp30 / fp29 at %d waves/SIMD: mul C %.3f  mul U %.3f  sqr %.3f  mul_add C %.3f  mul_add split %.3f  mul_inj %.3f  sqr_inj2/sqr29 %.3f\n",
               wps, m30 / m29, u30 / m29, s30 / s29, p30 / p29, q30 / p29, i30 / m29, j30 / s29);
    }
    return 0;
}
