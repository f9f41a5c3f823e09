// tools/ubench_fp29.hip -- multiplication / squaring / fused-pair rate of the unsaturated field (csrc/fp29.hpp) on gfx950:
// the ceiling bench.py's roofline_valu divides by.  Build twice to compare the compiler-scheduled columns with the
// verbatim chains of fp29_mac.hpp:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I rust-eth-kzg_amd/csrc [-DFQ_ASM_MAC] tools/ubench_fp29.hip -o tools/ubench_fp29
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.hpp"
using namespace kzg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 512;

template <int OP>
__global__ __launch_bounds__(256, 2) void k_op(uint32_t* out, uint32_t seed) {
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
template <class K>
void run(const char* name, int blocks, K kern, uint32_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, 256>>>(out, 1u); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, 1u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-30s blocks=%5d  %8.3f ms  %8.2f G op/s\n", name, blocks, best, 2.0 * ITER * blocks * 256 / (best * 1e-3) * 1e-9);
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    uint32_t* out; CK(hipMalloc(&out, 1 << 26));
#ifdef FQ_ASM_MAC
    printf("fp29 with verbatim v_mad_u64_u32 chains (fp29_mac.hpp)\n");
#else
    printf("fp29 with compiler-scheduled columns\n");
#endif
    for (int wps : {2, 4}) {
        int blocks = prop.multiProcessorCount * wps;
        printf("--- %d waves per SIMD requested ---\n", wps);
        run("mul (392 MACs)", blocks, k_op<0>, out);
        run("sqr (301 MACs)", blocks, k_op<1>, out);
        run("mul_add a*b+c*d (588 MACs)", blocks, k_op<2>, out);
    }
    return 0;
}
