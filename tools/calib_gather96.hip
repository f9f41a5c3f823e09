// tools/calib_gather96.hip -- does a cache-policy bit make the L2 fetch LESS than whole 128-B lines for the MSM's table gathers?
// Every lane gathers random 96-byte entries (six 16-byte loads, entries at a 96-byte stride: 1.5 lines touched per entry on
// average) from a 24 GB table, once with plain loads and once with non-temporal ones (`nt`: global_load_dwordx4 ... nt).
// Run under `rocprofv3 --pmc FETCH_SIZE` and compare the kernels' counters (x 2 on gfx950, MI355X_MICROARCH.md) with the bytes
// printed here; the kernels' own times are printed too.
//   hipcc -O3 --offload-arch=gfx950 tools/calib_gather96.hip -o tools/calib_gather96
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int PER_LANE = 64;
constexpr uint64_t ENTRIES = 1ull << 28;  // x 96 B = 25.8 GB, far beyond the 256 MiB Infinity Cache

template <int MODE>
__device__ __forceinline__ uint4 ld(const uint4* p) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    if (MODE == 1) {
        const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
// E = entry bytes (= stride), four independent entries in flight per lane (bandwidth-bound rather than latency-bound)
template <int MODE, int E>
__global__ void k_gather(const uint4* __restrict__ table, uint32_t* __restrict__ out, uint64_t entries) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = gid * 0x9e3779b97f4a7c15ull + 12345;
    uint4 acc = {0, 0, 0, 0};
    for (int i = 0; i < PER_LANE; i += 4) {
        const uint4* p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            p[u] = table + ((s >> 20) & (entries - 1)) * (E / 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < E / 16; k++) { const uint4 v = ld<MODE>(p[u] + k); acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
template <int MODE, int E>
static void run(const uint4* table, uint32_t* out, int blocks, int thr, hipEvent_t a, hipEvent_t b) {
    const uint64_t entries = E == 96 ? ENTRIES : ENTRIES / 2;  // powers of two: 25.8 GB of 96-B entries, 17.2 GB of 128-B ones
    const double n = (double)blocks * thr * PER_LANE;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        k_gather<MODE, E><<<blocks, thr>>>(table, out, entries);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%3d-byte entries, %s: %.3f ms, %.2f TB/s of entries\n", E, MODE ? "nt loads   " : "plain loads", ms, n * E / ms / 1e9);
    }
}
int main() {
    const int blocks = 8192, thr = 256;
    uint4* table; uint32_t* out;
    CK(hipMalloc(&table, ENTRIES * 96)); CK(hipMalloc(&out, (size_t)blocks * thr * 4));
    CK(hipMemset(table, 1, ENTRIES * 96));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double n = (double)blocks * thr * PER_LANE;
    run<0, 96>(table, out, blocks, thr, a, b);
    run<1, 96>(table, out, blocks, thr, a, b);
    run<0, 128>(table, out, blocks, thr, a, b);
    run<1, 128>(table, out, blocks, thr, a, b);
    double lines = 0;  // an entry at byte offset 96 e: offsets cycle 0, 96, 64, 32 mod 128
    for (int r = 0; r < 4; r++) { const int o = (r * 96) % 128; lines += (o + 96 <= 128) ? 1 : 2; }
    lines /= 4;
    double sect = 0;   // 64-byte sectors touched
    for (int r = 0; r < 4; r++) { const int o = (r * 96) % 64; sect += (o + 96 + 63) / 64; }
    sect /= 4;
    printf("entries gathered per launch: %.0f\n 96-byte entries: algorithmic %.3f GB  whole 128-B lines %.3f GB (%.2f per entry)  64-B sectors %.3f GB (%.2f per entry)\n"
           "128-byte entries (aligned): algorithmic = one line = two sectors = %.3f GB\n",
           n, n * 96 / 1e9, n * lines * 128 / 1e9, lines, n * sect * 64 / 1e9, sect, n * 128 / 1e9);
    return 0;
}
