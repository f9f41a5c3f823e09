// tools/ubench.hip -- VALU integer-multiply microbenchmarks for gfx950 (MI355X).
// Establishes the integer-MAC roofline the MSM / G1-FFT kernels are measured against
// (the v_mad_u64_u32 rate is not in the hardware guides).
//   hipcc -O3 --offload-arch=gfx950 -I rust-eth-kzg_amd/csrc tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "field.hpp"
using namespace kzg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 4096;

__global__ void k_mad64(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint64_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
            "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
            "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
            "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
            : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
}
__global__ void k_mullohi(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %9\n"
            "v_mul_lo_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %9\n v_mul_lo_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mad24(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u32_u24 %0, %8, %9, %0\n v_mad_u32_u24 %1, %8, %9, %1\n v_mad_u32_u24 %2, %8, %9, %2\n v_mad_u32_u24 %3, %8, %9, %3\n"
            "v_mad_u32_u24 %4, %8, %9, %4\n v_mad_u32_u24 %5, %8, %9, %5\n v_mad_u32_u24 %6, %8, %9, %6\n v_mad_u32_u24 %7, %8, %9, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mulhi24(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %9\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %9\n"
            "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %9\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_fma64(uint32_t* out, uint32_t seed) {
    double a = 1.0 + (threadIdx.x + seed) * 1e-9, b = 1e-12;
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
            "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}
__global__ void k_add32(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n"
            "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <class F>
__global__ void k_check(const F* a, const F* b, F* o_cios, F* o_fips, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_cios[i] = mul_cios(a[i], b[i]);
    o_fips[i] = mul_fips(a[i], b[i]);
}
template <class F>
int check_field(const char* name) {
    const int n = 1 << 16;
    std::vector<F> a(n), b(n), c1(n), c2(n);
    uint64_t s = 0x243F6A8885A308D3ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < F::N; k++) { a[i].v[k] = rnd(); b[i].v[k] = rnd(); }
        a[i].v[F::N - 1] &= 0x0fffffffu; b[i].v[F::N - 1] &= 0x0fffffffu;  // < modulus
        if (i < 4) for (int k = 0; k < F::N; k++) a[i].v[k] = (i & 1) ? 0u : a[i].v[k];
    }
    F *da, *db, *d1, *d2;
    CK(hipMalloc(&da, n * sizeof(F))); CK(hipMalloc(&db, n * sizeof(F))); CK(hipMalloc(&d1, n * sizeof(F))); CK(hipMalloc(&d2, n * sizeof(F)));
    CK(hipMemcpy(da, a.data(), n * sizeof(F), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * sizeof(F), hipMemcpyHostToDevice));
    k_check<F><<<n / 256, 256>>>(da, db, d1, d2, n);
    CK(hipMemcpy(c1.data(), d1, n * sizeof(F), hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), d2, n * sizeof(F), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        F h = mul_cios(a[i], b[i]);
        if (!eq(h, c1[i]) || !eq(h, c2[i])) bad++;
    }
    printf("check %s: %d / %d mismatches (device cios, device fips vs host cios)\n", name, bad, n);
    return bad;
}

template <class F, int CHAINS, int VARIANT>
__global__ void k_fieldmul(uint32_t* out, uint32_t seed, int iters) {
    F a[CHAINS], b;
    for (int c = 0; c < CHAINS; c++)
        for (int i = 0; i < F::N; i++) a[c].v[i] = (threadIdx.x + 1) * 2654435761u + seed * (i + 1) + c;
    for (int i = 0; i < F::N; i++) b.v[i] = (threadIdx.x + 7) * 40503u + i;
    a[0].v[F::N - 1] &= 0x0fffffff; b.v[F::N - 1] &= 0x0fffffff;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) a[c] = VARIANT ? mul_fips(a[c], b) : mul_cios(a[c], b);
    }
    uint32_t x = 0;
    for (int c = 0; c < CHAINS; c++) for (int i = 0; i < F::N; i++) x ^= a[c].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <class K, class... A>
double run(const char* name, double ops_per_thread, int blocks, int threads, K kern, A... args) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, threads>>>(args...); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0)); kern<<<blocks, threads>>>(args...); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double total = ops_per_thread * (double)blocks * threads;
    double rate = total / (best * 1e-3);
    printf("%-34s blocks=%5d thr=%4d  %8.3f ms  %10.3f Gop/s\n", name, blocks, threads, best, rate * 1e-9);
    return rate;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    uint32_t* out; CK(hipMalloc(&out, 1 << 26));
    if (check_field<Fp>("Fp") | check_field<Fr>("Fr")) { printf("FIELD CHECK FAILED\n"); return 1; }
    int cus = prop.multiProcessorCount;
    for (int wpc : {4, 8, 16}) {  // waves per CU
        int blocks = cus * wpc / 4, thr = 256;
        printf("--- %d waves/CU ---\n", wpc);
        run("v_mad_u64_u32", 8.0 * ITER, blocks, thr, k_mad64, out, 1u);
        run("v_mul_lo_u32+v_mul_hi_u32", 8.0 * ITER, blocks, thr, k_mullohi, out, 1u);
        run("v_mad_u32_u24", 8.0 * ITER, blocks, thr, k_mad24, out, 1u);
        run("v_mul_hi_u32_u24", 8.0 * ITER, blocks, thr, k_mulhi24, out, 1u);
        run("v_fma_f64", 8.0 * ITER, blocks, thr, k_fma64, out, 1u);
        run("v_add_co/v_addc_co", 8.0 * ITER, blocks, thr, k_add32, out, 1u);
    }
    for (int wpc : {4, 8, 16, 32}) {
        int blocks = cus * wpc / 4, thr = 256;
        printf("--- field mul, %d waves/CU ---\n", wpc);
        run("Fp mul cios (1 chain)", 256.0, blocks, thr, k_fieldmul<Fp, 1, 0>, out, 1u, 256);
        run("Fp mul cios (2 chains)", 512.0, blocks, thr, k_fieldmul<Fp, 2, 0>, out, 1u, 256);
        run("Fp mul fips (1 chain)", 256.0, blocks, thr, k_fieldmul<Fp, 1, 1>, out, 1u, 256);
        run("Fp mul fips (2 chains)", 512.0, blocks, thr, k_fieldmul<Fp, 2, 1>, out, 1u, 256);
        run("Fr mul cios (1 chain)", 256.0, blocks, thr, k_fieldmul<Fr, 1, 0>, out, 1u, 256);
        run("Fr mul fips (1 chain)", 256.0, blocks, thr, k_fieldmul<Fr, 1, 1>, out, 1u, 256);
        run("Fr mul fips (2 chains)", 512.0, blocks, thr, k_fieldmul<Fr, 2, 1>, out, 1u, 256);
    }
    return 0;
}
