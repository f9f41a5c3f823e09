// experiment: how long does hipMalloc of the window tables take, in one piece vs several pieces allocated from parallel threads?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t GB = 1ull << 30;
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    hipSetDevice(0);
    hipFree(0);
    double t0 = now();
    if (mode == 0) {
        void* p; hipError_t e = hipMalloc(&p, 200 * GB);
        printf("one piece of 200 GB: %.3f s (%s)\n", now() - t0, hipGetErrorString(e));
        t0 = now(); hipMemset(p, 0, 1 << 20); hipDeviceSynchronize(); printf("first touch %.3f s\n", now() - t0);
    } else {
        const int n = mode;
        std::vector<std::thread> th; std::vector<void*> ps(n);
        for (int i = 0; i < n; i++) th.emplace_back([&, i] { hipSetDevice(0); hipMalloc(&ps[i], 200 * GB / n); });
        for (auto& t : th) t.join();
        printf("%d pieces of %d GB from %d threads: %.3f s\n", n, 200 / n, n, now() - t0);
    }
    return 0;
}
