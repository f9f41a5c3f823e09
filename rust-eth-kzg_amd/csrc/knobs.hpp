// EVERY environment variable the library reads, in one place.  They are read ONCE per context (Knobs::from_env in the
// constructor of the context's engine; the auxiliary engines a context creates copy its values): no getenv on any call path.
// The reference gets by with one runtime knob (UsePrecomp::Yes{width}, crates/cryptography/bls12_381/src/fixed_base_msm.rs:41-49);
// here eight are for the host application and the rest are hooks the parity tests use to force a schedule.
//
//   for the host application
//   ETH_KZG_AMD_DEVICE=<n>          GPU ordinal of eth_kzg_das_context_new (default 0)
//   ETH_KZG_AMD_DEVICES=<a,b,..>|all the device LIST of eth_kzg_das_context_new: one engine per listed GPU behind the one context pointer; single
//                                   calls go to the least-loaded device, batched calls are cut into contiguous slices (c_api.cpp)
//   ETH_KZG_AMD_TABLE_GB=<gb>|max   HBM for the two window tables together (default 108 = the nine-window GLV tables: 71 GB for FK20,
//                                   35 GB for commitments; max = whatever the HBM holds: eight windows of 16 bits for FK20, 242 GB in all)
//   ETH_KZG_AMD_GLV_WINDOW=<w>      exactly this GLV table width for FK20 (16, 15, 14, 12, 8)
//   ETH_KZG_AMD_PROGRESSIVE=0       build the wide tables inside the constructor instead of behind it
//   ETH_KZG_AMD_HOST_THREADS=<n>    helper threads of the host-pointer paths (gather / scatter, hashing, pairings)
//   ETH_KZG_AMD_SERIAL_LANES=<n>    engine lanes for concurrent recovery / commitment / EIP-4844 calls (default 4)
//   ETH_KZG_AMD_TRACE=1             timings of context creation and of the verification steps on stderr
//
//   test hooks (tests/test_gpu_*.py force every schedule against the oracle)
//   ETH_KZG_AMD_MSM_CHUNKS=0|4      0: the windowed MSM kernel at every batch size, 4: four chunks per MSM
//   ETH_KZG_AMD_SLP_PROGRAM=<0..5>  one compilation of the G1 linear map at every batch size
//   ETH_KZG_AMD_PIP_SHIFT_MIN=<n>   smallest verification that uses byte-shifted point copies
//   ETH_KZG_AMD_VM_SEARCH=0         many-verification: re-check every problem of a failed pass instead of searching
//   ETH_KZG_AMD_VM_FOLD=0           many-verification: one pairing per problem instead of one folded check per pass
//   ETH_KZG_AMD_VERIFY_COMBINE=0    concurrent single verifications are not combined into passes
//   ETH_KZG_AMD_MSM_SPLIT=0         batches of <= 32 blobs: a lane per MSM window (rounds 2-5) instead of two (A/B runs, the tests' cross-check)
//   ETH_KZG_AMD_ARENA_SIGNED=0      the prover's, recovery's and the commitments' G1 points in the 14 x 29-bit form and kernels of rounds 2-5
//                                   instead of the signed 13 x 30-bit ones: the tests' cross-check of the two forms, A/B runs
//   ETH_KZG_AMD_DEVICE_BATCH_MAX=<n> a device-resident prover call is cut into sub-batches of at most n blobs (default 4096)
//   ETH_KZG_AMD_FAULT=constructor   the context's constructor throws after its last step but one (tests: try_new returns NULL + a
//                                   message, nothing leaks, the next context works)
//   ETH_KZG_AMD_COOP_POINTS=<n>     largest launch that takes the several-lanes-per-point kernels (0: never; process-wide, read
//                                   once by launch::coop_points_max: launch geometry is decided outside any context)
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace kzg {

struct Knobs {
    int device = 0;
    std::vector<int> devices;  // ETH_KZG_AMD_DEVICES: the device list of eth_kzg_das_context_new; {-1} = all GPUs of the node
    double table_budget_gb = 0;  // 0: the engine's default; < 0: what the HBM holds
    int glv_window = 0;
    bool progressive = true;
    int host_threads = 0;  // 0: chosen from the core count
    int serial_lanes = 0;  // 0: the engine's default
    bool trace = false;
    int msm_chunks = -1, slp_program = -1, pip_shift_min = 0;
    bool vm_search = true, vm_fold = true, verify_combine = true, arena_signed = true, msm_split = true;
    std::string fault;  // ETH_KZG_AMD_FAULT (a copy: the environment may change under a long-lived context)
    int device_batch_max = 0;

    static Knobs from_env() {
        Knobs k;
        auto num = [](const char* name, int lo, int hi, int& out) {
            if (const char* s = getenv(name)) { const int v = atoi(s); if (v >= lo && v <= hi) out = v; }
        };
        auto flag = [](const char* name, bool& out) {
            if (const char* s = getenv(name)) out = atoi(s) != 0;
        };
        num("ETH_KZG_AMD_DEVICE", 0, 4095, k.device);
        if (const char* s = getenv("ETH_KZG_AMD_DEVICES")) {
            if (!strcmp(s, "all") || !strcmp(s, "ALL")) k.devices.push_back(-1);
            else
                for (const char* p = s; *p;) {  // "0,1,2": ordinals, repeats allowed (two engines on one GPU: the one-GPU test of the fan-out)
                    char* end = nullptr;
                    const long v = strtol(p, &end, 10);
                    if (end == p) break;
                    if (v >= 0 && v <= 4095 && k.devices.size() < 64) k.devices.push_back((int)v);
                    p = *end == ',' ? end + 1 : end;
                    if (*end && *end != ',') break;
                }
        }
        if (const char* s = getenv("ETH_KZG_AMD_TABLE_GB")) {
            if (!strcmp(s, "max") || !strcmp(s, "MAX")) k.table_budget_gb = -1;
            else if (atof(s) > 0) k.table_budget_gb = atof(s);
        }
        num("ETH_KZG_AMD_GLV_WINDOW", 8, 16, k.glv_window);
        flag("ETH_KZG_AMD_PROGRESSIVE", k.progressive);
        num("ETH_KZG_AMD_HOST_THREADS", 1, 64, k.host_threads);
        num("ETH_KZG_AMD_SERIAL_LANES", 1, 16, k.serial_lanes);
        k.trace = getenv("ETH_KZG_AMD_TRACE") != nullptr;
        num("ETH_KZG_AMD_MSM_CHUNKS", 0, 4, k.msm_chunks);
        num("ETH_KZG_AMD_SLP_PROGRAM", 0, 15, k.slp_program);
        num("ETH_KZG_AMD_PIP_SHIFT_MIN", 1, 1 << 24, k.pip_shift_min);
        flag("ETH_KZG_AMD_VM_SEARCH", k.vm_search);
        flag("ETH_KZG_AMD_VM_FOLD", k.vm_fold);
        flag("ETH_KZG_AMD_VERIFY_COMBINE", k.verify_combine);
        flag("ETH_KZG_AMD_ARENA_SIGNED", k.arena_signed);
        flag("ETH_KZG_AMD_MSM_SPLIT", k.msm_split);
        if (const char* s = getenv("ETH_KZG_AMD_FAULT")) k.fault = s;
        num("ETH_KZG_AMD_DEVICE_BATCH_MAX", 64, 1 << 20, k.device_batch_max);
        return k;
    }
};

}  // namespace kzg
