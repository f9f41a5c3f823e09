// tools/ubench_issue.hip -- issue cost, in shader cycles per wave64 instruction per SIMD, of the integer VALU instructions
// the point kernels are made of, at 1 / 2 / 4 waves per SIMD, with INDEPENDENT operands (no VCC chain) and as a dependent
// chain, plus the MAC : other = 3 : 1 mix of the Montgomery multiplication (the round-1 ceiling was
// measured with VCC-chained v_add_co / v_addc_co only).  Cycles come from s_memtime around the loop of one wave per SIMD
// slot (the shader clock the SIMD actually ran at), lane-ops/s from HIP events over the whole launch.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048, UNROLL = 16;  // 16 instructions per asm block, 8 accumulators used round-robin

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL32(NAME, ASM16)                                                                                             \
    __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                                        \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = (a ^ 0x9e3779b9u) | 1u;                                       \
        uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int i = 0; i < ITER; i++) {                                                                                  \
            asm volatile(ASM16 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)           \
                         : "v"(a), "v"(b));                                                                               \
        }                                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                               \
        if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                          \
    }
#define S_(n) #n
#define OP_ADD(n) "v_add_u32 %" S_(n) ", %" S_(n) ", %8\n"
#define OP_AND(n) "v_and_b32 %" S_(n) ", %" S_(n) ", %9\n"
#define OP_SHR(n) "v_lshrrev_b32 %" S_(n) ", 3, %" S_(n) "\n"
#define OP_ALIGN(n) "v_alignbit_b32 %" S_(n) ", %8, %" S_(n) ", 29\n"
#define OP_MOV(n) "v_mov_b32 %" S_(n) ", %8\n"
#define OP_LSHLADD(n) "v_lshl_add_u32 %" S_(n) ", %" S_(n) ", 3, %8\n"
#define OP_ADD3(n) "v_add3_u32 %" S_(n) ", %" S_(n) ", %8, %9\n"
#define OP_ANDOR(n) "v_and_or_b32 %" S_(n) ", %" S_(n) ", %8, %9\n"
#define OP_MULLO(n) "v_mul_lo_u32 %" S_(n) ", %" S_(n) ", %9\n"
#define OP_MAD24(n) "v_mad_u32_u24 %" S_(n) ", %8, %9, %" S_(n) "\n"
#define OP_FMA32(n) "v_fma_f32 %" S_(n) ", %8, %9, %" S_(n) "\n"
#define OP_ADDF32(n) "v_add_f32 %" S_(n) ", %" S_(n) ", %8\n"
#define OP_SUBREV(n) "v_subrev_u32 %" S_(n) ", %8, %" S_(n) "\n"
#define OP_CNDMASK(n) "v_cndmask_b32 %" S_(n) ", %" S_(n) ", %8, vcc\n"
#define OP_ADDCO(n) "v_add_co_u32 %" S_(n) ", vcc, %" S_(n) ", %8\n"
KERNEL32(k_add, R8(OP_ADD))
KERNEL32(k_and, R8(OP_AND))
KERNEL32(k_shr, R8(OP_SHR))
KERNEL32(k_align, R8(OP_ALIGN))
KERNEL32(k_mov, R8(OP_MOV))
KERNEL32(k_lshladd, R8(OP_LSHLADD))
KERNEL32(k_add3, R8(OP_ADD3))
KERNEL32(k_andor, R8(OP_ANDOR))
KERNEL32(k_mullo, R8(OP_MULLO))
KERNEL32(k_mad24, R8(OP_MAD24))
KERNEL32(k_fma32, R8(OP_FMA32))
KERNEL32(k_addf32, R8(OP_ADDF32))
KERNEL32(k_subrev, R8(OP_SUBREV))
// one dependent chain of 16 additions (every instruction reads the previous result)
#define OP_ADDCHAIN(n) "v_add_u32 %0, %0, %8\n"
KERNEL32(k_add_chain, R8(OP_ADDCHAIN))

// 64-bit multiply-add: 8 independent accumulators / one dependent chain / the 3 : 1 mix with plain 32-bit instructions
#define KERNEL64(NAME, ASM16)                                                                                             \
    __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                                        \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = (a ^ 0x9e3779b9u) | 1u;                                       \
        uint64_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;                  \
        uint32_t y0 = a, y1 = b, y2 = a ^ b, y3 = a + b;                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int i = 0; i < ITER; i++) {                                                                                  \
            asm volatile(ASM16 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),          \
                                 "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3)                                                   \
                         : "v"(a), "v"(b) : "vcc");                                                                       \
        }                                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) ^ y0 ^ y1 ^ y2 ^ y3; \
        if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                          \
    }
#define M(n) "v_mad_u64_u32 %" S_(n) ", vcc, %12, %13, %" S_(n) "\n"
#define Y_AND(n) "v_and_b32 %" S_(n) ", %" S_(n) ", %13\n"
#define Y_SHR(n) "v_lshrrev_b32 %" S_(n) ", 3, %" S_(n) "\n"
#define Y_ADD(n) "v_add_u32 %" S_(n) ", %" S_(n) ", %12\n"
#define Y_ALIGN(n) "v_alignbit_b32 %" S_(n) ", %12, %" S_(n) ", 29\n"
KERNEL64(k_mad64, M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7))
KERNEL64(k_mad64_chain, M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0))
// 12 multiply-adds + 4 plain instructions: the ratio of the point kernels (3,542 of 4,709 instructions are multiply-adds)
KERNEL64(k_mix_3to1, M(0) M(1) M(2) Y_AND(8) M(3) M(4) M(5) Y_SHR(9) M(6) M(7) M(0) Y_ADD(10) M(1) M(2) M(3) Y_ALIGN(11))
// the same with the multiply-adds as ONE dependent chain (a column of the Montgomery product accumulates into one register pair)
KERNEL64(k_mix_3to1_chain, M(0) M(0) M(0) Y_AND(8) M(0) M(0) M(0) Y_SHR(9) M(0) M(0) M(0) Y_ADD(10) M(0) M(0) M(0) Y_ALIGN(11))
// 8 + 8
KERNEL64(k_mix_1to1, M(0) Y_AND(8) M(1) Y_SHR(9) M(2) Y_ADD(10) M(3) Y_ALIGN(11) M(4) Y_AND(8) M(5) Y_SHR(9) M(6) Y_ADD(10) M(7) Y_ALIGN(11))

// the 64-bit forms around the multiply-add columns: the column shift acc >>= 29 and its two-instruction 32-bit replacement
#define SHR64(n) "v_lshrrev_b64 %" S_(n) ", 29, %" S_(n) "\n"
#define LSHLADD64(n) "v_lshl_add_u64 %" S_(n) ", %" S_(n) ", 3, %" S_(n) "\n"
KERNEL64(k_shr64, SHR64(0) SHR64(1) SHR64(2) SHR64(3) SHR64(4) SHR64(5) SHR64(6) SHR64(7) SHR64(0) SHR64(1) SHR64(2) SHR64(3) SHR64(4) SHR64(5) SHR64(6) SHR64(7))
KERNEL64(k_lshladd64, LSHLADD64(0) LSHLADD64(1) LSHLADD64(2) LSHLADD64(3) LSHLADD64(4) LSHLADD64(5) LSHLADD64(6) LSHLADD64(7) LSHLADD64(0) LSHLADD64(1) LSHLADD64(2) LSHLADD64(3) LSHLADD64(4) LSHLADD64(5) LSHLADD64(6) LSHLADD64(7))
// one column of the Montgomery product as the kernels run it: 13 chained multiply-adds, the quotient digit (v_mul_lo + v_and), the shift
#define Y_MULLO(n) "v_mul_lo_u32 %" S_(n) ", %" S_(n) ", %13\n"
KERNEL64(k_column, M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) Y_MULLO(8) Y_AND(8) SHR64(0))
// the same column with s_nop 0 after the multiply-add run (what hipcc emits after an inline-asm statement)
KERNEL64(k_column_nop, M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) "s_nop 0\n" Y_MULLO(8) Y_AND(8) SHR64(0))

template <class K>
void run(const char* name, int cus, K kern, uint32_t* out, unsigned long long* d_cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks of 64 threads, cus * 4 * wps of them = one per SIMD slot
        const int blocks = cus * 4 * wps;
        kern<<<blocks, 64>>>(out, d_cyc, 1u); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0)); kern<<<blocks, 64>>>(out, d_cyc, 1u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
        const double instr = (double)ITER * UNROLL;
        // the SIMD issues wps such streams side by side: cycles per instruction per SIMD = wave cycles / instructions / wps
        printf("  | %d w/SIMD: %5.2f cyc/instr/SIMD  %6.2f T lane-ops/s", wps, mean / instr / wps, instr * 64.0 * blocks / (best * 1e-3) * 1e-12);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz   (%d instructions per wave per launch; s_memtime ticks = shader cycles)\n", prop.name,
           prop.multiProcessorCount, prop.clockRate / 1000, ITER * UNROLL);
    uint32_t* out; CK(hipMalloc(&out, 1 << 24));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 1 << 20));
    const int cus = prop.multiProcessorCount;
    run("v_add_u32", cus, k_add, out, cyc);
    run("v_subrev_u32", cus, k_subrev, out, cyc);
    run("v_and_b32", cus, k_and, out, cyc);
    run("v_lshrrev_b32", cus, k_shr, out, cyc);
    run("v_alignbit_b32", cus, k_align, out, cyc);
    run("v_mov_b32", cus, k_mov, out, cyc);
    run("v_lshl_add_u32", cus, k_lshladd, out, cyc);
    run("v_add3_u32", cus, k_add3, out, cyc);
    run("v_and_or_b32", cus, k_andor, out, cyc);
    run("v_mul_lo_u32", cus, k_mullo, out, cyc);
    run("v_mad_u32_u24", cus, k_mad24, out, cyc);
    run("v_fma_f32", cus, k_fma32, out, cyc);
    run("v_add_f32", cus, k_addf32, out, cyc);
    run("v_add_u32 dependent chain", cus, k_add_chain, out, cyc);
    run("v_mad_u64_u32 x8 independent", cus, k_mad64, out, cyc);
    run("v_mad_u64_u32 one chain", cus, k_mad64_chain, out, cyc);
    run("mix 12 mad64 : 4 plain", cus, k_mix_3to1, out, cyc);
    run("mix 12 mad64 (chain) : 4", cus, k_mix_3to1_chain, out, cyc);
    run("mix 8 mad64 : 8 plain", cus, k_mix_1to1, out, cyc);
    run("v_lshrrev_b64", cus, k_shr64, out, cyc);
    run("v_lshl_add_u64", cus, k_lshladd64, out, cyc);
    run("column 13 mad64+mul_lo+and+shr64", cus, k_column, out, cyc);
    run("column with s_nop 0 (15+1)", cus, k_column_nop, out, cyc);
    return 0;
}
