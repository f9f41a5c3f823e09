// Is a wave with only some of its lanes in use slower than a full one?  (Seen in k_slp_mulc_s and k_g1_compress, round 6.)
// One wave per SIMD (1024 blocks of 64 threads); lanes >= K leave at once; the others run a dependent chain of
//   mode 0: Montgomery products of the signed 13 x 30-bit field (v_mad_i64_i32),  mode 1: 32-bit multiply-adds (v_mad_u32_u24 /
//   v_mul_lo_u32 + add),  mode 2: the products with the table in scratch (a dynamically indexed private array, like g1_mulc30.hpp)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rust-eth-kzg_amd/csrc tools/ubench_partial_wave.hip -o /tmp/ubench_partial_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "fp30.hpp"
using namespace kzg;

__global__ __launch_bounds__(64, 2) void k_chain(int mode, int K, int iters, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
    const int lane = threadIdx.x;
    if (lane >= K) return;
    const int gid = blockIdx.x * 64 + lane;
    if (mode == 1) {
        uint32_t a = in[gid & 1023], b = in[(gid + 7) & 1023] | 1u;
        for (int i = 0; i < iters * 300; i++) a = a * b + (uint32_t)i;
        out[gid] = (int32_t)a;
        return;
    }
    Fs<1, DC> x, y;
    for (int i = 0; i < SL; i++) { x.v[i] = in[(gid + i) & 1023] & 0x0fffffff; y.v[i] = in[(gid + 3 * i + 1) & 1023] & 0x0fffffff; }
    if (mode == 0) {
        for (int i = 0; i < iters; i++) x = mul(x, y);
    } else {
        Fs<1, DC> T[8];
        for (int j = 0; j < 8; j++) { T[j] = x; x = mul(x, y); }
        for (int i = 0; i < iters; i++) {
            const int idx = __builtin_amdgcn_readfirstlane((i * 5 + (i >> 3)) & 7);
            x = mul(x, T[idx]);
        }
    }
    int32_t s = 0;
    for (int i = 0; i < SL; i++) s ^= x.v[i];
    out[gid] = s;
}

int main() {
    int32_t *in, *out;
    hipMalloc(&in, 1024 * 4);
    hipMalloc(&out, 4096 * 64 * 4);
    int32_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = (int32_t)(i * 2654435761u);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {1024, 2048})
        for (int mode = 0; mode < 3; mode++)
            for (int K : {64, 48, 32, 24, 17, 16, 8, 4, 1}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0);
                    k_chain<<<blocks, 64>>>(mode, K, 2000, in, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                printf("blocks=%d mode=%d lanes_in_use=%2d: %.3f ms\n", blocks, mode, K, best);
            }
    // (2) does the time of a chain depend on how many SIMDs are busy?  all 64 lanes in use, one wave per block
    for (int blocks : {16, 64, 128, 256, 512, 1024})
        for (int K : {64, 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                k_chain<<<blocks, 64>>>(0, K, 2000, in, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("waves=%4d mode=0 lanes_in_use=%2d: %.3f ms\n", blocks, K, best);
        }
    return 0;
}
