// tools/calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE for the MSM's access pattern on gfx950: every lane gathers
// random 112-byte table entries with seven 16-byte loads (MI355X_MICROARCH.md: FETCH_SIZE is exact only for calibrated
// patterns; wide coalesced reads are reported at half their bytes).  Prints the algorithmic bytes and the bytes of the
// distinct 128-B lines the gathers touch; run under `rocprofv3 --pmc FETCH_SIZE` and compare.
//   hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int PER_LANE = 64;
constexpr uint64_t ENTRIES = 1ull << 27;  // x 112 B = 15 GB, far beyond the 256 MiB Infinity Cache

__global__ void k_gather(const uint4* __restrict__ table, uint32_t* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = gid * 0x9e3779b97f4a7c15ull + 12345;
    uint4 acc = {0, 0, 0, 0};
    for (int i = 0; i < PER_LANE; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t e = (s >> 20) & (ENTRIES - 1);
        const uint4* p = table + e * 7;
#pragma unroll
        for (int k = 0; k < 7; k++) { const uint4 v = p[k]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
int main() {
    const int blocks = 4096, thr = 256;
    uint4* table; uint32_t* out;
    CK(hipMalloc(&table, ENTRIES * 112)); CK(hipMalloc(&out, (size_t)blocks * thr * 4));
    CK(hipMemset(table, 1, ENTRIES * 112));
    k_gather<<<blocks, thr>>>(table, out); CK(hipDeviceSynchronize());
    const double n = (double)blocks * thr * PER_LANE;
    // an entry at byte offset 112 e covers [o, o + 112): one 128-B line if (o % 128) <= 16, else two
    double lines = 0;
    for (int r = 0; r < 8; r++) lines += ((r * 112) % 128 <= 16) ? 1 : 2;  // offsets cycle with period 8 entries
    lines /= 8;
    printf("entries gathered: %.0f  algorithmic bytes: %.3f GB  128-B lines touched: %.3f GB (%.3f lines per entry)\n", n, n * 112 / 1e9,
           n * lines * 128 / 1e9, lines);
    return 0;
}
