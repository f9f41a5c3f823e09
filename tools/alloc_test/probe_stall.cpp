// experiment: which HIP calls of another thread wait while one thread allocates window-table pieces -- in particular right after
// another process has freed a few hundred GB (the driver wipes freed VRAM; an allocation that lands on memory still being wiped
// waits for it inside the driver, holding whatever runtime lock it took).
//   ./stall hog 240        allocate + touch N GB, free, exit            (run first, then at once:)
//   ./stall probe 200      thread A: N GB in 0.8 GB pieces; thread B: a cycle of small HIP operations, longest wait per kind
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(unsigned* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i * 1024] = 1; }
__global__ void k_small(unsigned* p) { p[threadIdx.x] += 1; }
__global__ void k_copy(const uint4* src, uint4* dst, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
int main(int argc, char** argv) {
    const size_t GB = 1ull << 30;
    const char* mode = argc > 1 ? argv[1] : "probe";
    const int n_gb = argc > 2 ? atoi(argv[2]) : 200;
    hipSetDevice(0);
    hipFree(0);
    if (!strcmp(mode, "hog")) {
        std::vector<void*> ps;
        for (int i = 0; i < n_gb; i++) { void* p; if (hipMalloc(&p, GB) != hipSuccess) break; ps.push_back(p); k_touch<<<(unsigned)(GB / 4096 / 256), 256>>>((unsigned*)p, GB / 4096); }
        hipDeviceSynchronize();
        printf("hog: %zu GB allocated and touched\n", ps.size());
        return 0;  // exit without freeing: the process teardown frees (and the driver wipes)
    }
    const double t0 = now();
    std::atomic<bool> done{false};
    std::thread A([&] {
        hipSetDevice(0);
        const size_t piece = 850ull << 20;
        double worst = 0, total = 0;
        int n = (int)((size_t)n_gb * GB / piece);
        for (int i = 0; i < n; i++) {
            void* p;
            const double a = now();
            if (hipMalloc(&p, piece) != hipSuccess) { printf("A: hipMalloc failed at piece %d\n", i); break; }
            const double d = now() - a;
            total += d;
            if (d > worst) { worst = d; printf("A: piece %d at %.2f s took %.1f ms\n", i, a - t0, d * 1e3); }
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        printf("A: %d pieces, %.2f s in hipMalloc, done at %.2f s\n", n, total, now() - t0);
        done = true;
    });
    // thread B
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    unsigned *d, *h;
    hipMalloc(&d, 1 << 20);
    hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    unsigned* h_dev = nullptr;
    hipHostGetDevicePointer((void**)&h_dev, h, 0);
    const char* names[] = {"kernel launch + stream sync", "memcpyAsync H2D (pinned) + sync", "memcpyAsync D2H (pinned) + sync", "event record + event sync",
                           "memsetAsync + sync", "kernel copy from pinned host (zero-copy) + sync", "kernel copy to pinned host + sync", "hipMalloc 1 MB + hipFree"};
    double worst[8] = {0}, at[8] = {0};
    long count = 0;
    // (the small hipMalloc runs on a thread of its own: it waits behind A's allocation, and a cycle stuck in it would never
    // find out what the other operations do meanwhile)
    std::thread Cth([&] {
        hipSetDevice(0);
        while (!done) {
            const double a = now();
            void* p;
            hipMalloc(&p, 1 << 20);
            hipFree(p);
            const double dt = now() - a;
            if (dt > worst[7]) { worst[7] = dt; at[7] = a - t0; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    });
    while (!done) {
        for (int k = 0; k < 7; k++) {
            const double a = now();
            switch (k) {
                case 0: k_small<<<1, 64, 0, s>>>(d); hipStreamSynchronize(s); break;
                case 1: hipMemcpyAsync(d, h, 131072, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); break;
                case 2: hipMemcpyAsync(h, d, 131072, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); break;
                case 3: hipEventRecord(ev, s); hipEventSynchronize(ev); break;
                case 4: hipMemsetAsync(d, 0, 4096, s); hipStreamSynchronize(s); break;
                case 5: k_copy<<<32, 256, 0, s>>>((const uint4*)h_dev, (uint4*)d, 8192); hipStreamSynchronize(s); break;
                case 6: k_copy<<<32, 256, 0, s>>>((const uint4*)d, (uint4*)h_dev, 8192); hipStreamSynchronize(s); break;
                case 7: { void* p; hipMalloc(&p, 1 << 20); hipFree(p); break; }
            }
            const double dt = now() - a;
            if (dt > worst[k]) { worst[k] = dt; at[k] = a - t0; }
        }
        count++;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    A.join();
    Cth.join();
    for (int k = 0; k < 8; k++) printf("B: %-50s longest %8.1f ms (at %.2f s)\n", names[k], worst[k] * 1e3, at[k]);
    printf("B: %ld cycles\n", count);
    return 0;
}
