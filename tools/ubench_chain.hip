// tools/ubench_chain.hip -- how many independent v_mad_u64_u32 chains a wave needs to keep the integer pipe busy on gfx950.
// The field multiplication of fp29.hpp is a chain of dependent multiply-adds per column; this measures the issue rate of
// ILP = 1, 2, 4 interleaved dependent chains at 1, 2 and 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2048;

template <int ILP>
__global__ void k_chain(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint64_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    for (int i = 0; i < ITER; i++) {
        if (ILP == 1)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0\n"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0\n"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0\n"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0\n"
                         : "+v"(x0) : "v"(a), "v"(b) : "vcc");
        else if (ILP == 2)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n"
                         "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n"
                         "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n"
                         "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n"
                         : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %4, %1\n"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %5, %4, %3\n"
                         "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %4, %1\n"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %5, %4, %3\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 ^ x1 ^ x2 ^ x3);
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
    printf("%-28s blocks=%5d  %8.3f ms  %9.1f G lane-MAC/s\n", name, blocks, best, 8.0 * ITER * blocks * 256 / (best * 1e-3) * 1e-9);
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    uint32_t* out; CK(hipMalloc(&out, 1 << 26));
    int cus = prop.multiProcessorCount;
    for (int wps : {1, 2, 4}) {
        printf("--- %d wave(s) per SIMD ---\n", wps);
        run("1 dependent chain", cus * wps, k_chain<1>, out);
        run("2 interleaved chains", cus * wps, k_chain<2>, out);
        run("4 interleaved chains", cus * wps, k_chain<4>, out);
    }
    return 0;
}
