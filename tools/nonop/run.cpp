// Experiment: do the s_nop 0 that hipcc's hazard recogniser puts after every inline-asm statement cost anything, and are they
// needed?  Loads two code objects of tools/ubench_fp29.hip (as compiled; with the s_nop stripped), runs the same kernels,
// compares outputs and times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
int main(int argc, char** argv) {
    const char* files[2] = {argv[1], argv[2]};
    const char* kn[3] = {"_Z4k_opILi0EEvPjj", "_Z4k_opILi1EEvPjj", "_Z4k_opILi2EEvPjj"};
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 2;
    uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<uint32_t> res[2][3];
    for (int f = 0; f < 2; f++) {
        hipModule_t mod; CK(hipModuleLoad(&mod, files[f]));
        for (int k = 0; k < 3; k++) {
            hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, kn[k]));
            uint32_t seed = 1;
            void* args[] = {&out, &seed};
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int r = 0; r < 6; r++) {
                CK(hipEventRecord(e0));
                CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, 0, args, nullptr));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
            }
            res[f][k].resize((size_t)blocks * 256);
            CK(hipMemcpy(res[f][k].data(), out, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost));
            printf("%s  kernel %d: %8.3f ms\n", files[f], k, best);
        }
    }
    for (int k = 0; k < 3; k++) printf("kernel %d outputs identical: %d\n", k, (int)(res[0][k] == res[1][k]));
    return 0;
}
