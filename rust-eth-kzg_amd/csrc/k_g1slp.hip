// Executor of the straight-line program that g1_linmap.hpp compiles for the FK20 proofs map (the two G1 transforms of
// compute_cells_and_kzg_proofs as one linear map; reference: fft_inplace<G1Projective> via Domain::{ifft_g1_take_n,
// fft_g1}, crates/cryptography/polynomial/src/domain.rs:149-194).
//
// Arena layout: A[slot * stride + lane], lane = blob index inside the batch (stride = batch padded to a multiple of 64):
// one wave = one operation x 64 blobs, so the operation descriptor and, for multiplications, the digits of the public
// constant are wave-uniform (scalar loads and scalar branches, no divergence).  Operations of one launch are mutually
// independent and never write a slot that the same launch reads (linmap::make_schedule).
// words: 4 per operation: dst slot, a slot, b (slot | number of doublings | constant id), flags (1 = subtract, 2 = doubling run, 4 = a + b to dst AND a - b to slot flags >> 16; bits 3-7 of an addition: doublings of operand a first).
#include "engine.hpp"
#include "g1_mulc.hpp"
#include "g1_mulc30.hpp"
#include "g1_coop.hpp"

namespace kzg {

// (batches that fill the chip: the multiplication runs in the signed 13 x 30-bit field, g1_mulc30.hpp; the point is read from and
// written to the arena's 14 x 29-bit form)
__global__ __launch_bounds__(64, 2) void k_slp_mulc(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                    const uint32_t* __restrict__ naf, Fs<1, DC> beta) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 64 + threadIdx.x;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded30(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta);
}
// the constant multiplications of a batch of <= 16 blobs: four lanes per blob (a wave = 16 blobs x one operation), the quad sharing
// the digit loop's doublings and mixed additions
__global__ __launch_bounds__(64, 2) void k_slp_mulc_coop(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                         const uint32_t* __restrict__ naf, Fq<1> beta, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (lane >= lanes) return;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded<4>(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta, quad);
}
// ... and of 17 .. 64 blobs (BASELINE config 5's and 4's per-GPU shares): two lanes per blob, a wave = 32 blobs x one operation
// (from 33 blobs on two waves per operation: the engine then picks a compilation of the map with <= 512 multiplications)
__global__ __launch_bounds__(64, 2) void k_slp_mulc_coop2(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words,
                                                          const uint32_t* __restrict__ naf, Fq<1> beta, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.x * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   cid = __builtin_amdgcn_readfirstlane(w[2]);
    const int lane = blockIdx.y * 32 + (threadIdx.x >> 1), half = threadIdx.x & 1;
    if (lane >= lanes) return;
    const JacQ src = A[(size_t)a * stride + lane];
    A[(size_t)dst * stride + lane] = mul_by_recoded<2>(src, naf + (size_t)cid * (2 * launch::TWIDDLE_WORDS), beta, half);
}
// one cheap operation of the program on one lane: flags & 2: a run of b doublings; otherwise an addition (flags & 1: subtraction;
// flags & 4: a + b to dst AND a - b to slot flags >> 16) whose FIRST operand is doubled (flags >> 3) & 31 times in registers
// before the second one is read -- the schedule folds a doubling run into its only consumer (g1_linmap.hpp: make_schedule)
__device__ __forceinline__ void slp_cheap_op(JacQ* __restrict__ A, int stride, int lane, uint32_t dst, uint32_t a, uint32_t b, uint32_t fl) {
    JacQ r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = dbl(r);
    bool degenerate = false;
    if (!(fl & 2u)) {
        if (fl & 4u) {  // the difference is stored before the sum is computed (curve29.hpp: add_sub_*)
            const AddSubShared sh = add_sub_prepare(r, A[(size_t)b * stride + lane]);
            degenerate = sh.degenerate;  // (an identity, a = +-b: both results are redone below; what is stored here is overwritten)
            A[(size_t)(fl >> 16) * stride + lane] = add_sub_finish(sh, true);
            r = add_sub_finish(sh, false);
        } else {
            r = add(r, A[(size_t)b * stride + lane], (fl & 1u) != 0);
        }
    }
    A[(size_t)dst * stride + lane] = r;
    // The exact slow path of a pair comes LAST, when nothing else is live (inside the branch above its operands cost the common
    // path 33 spilled registers): the operands are read and doubled again.  Rare: all-zero / constant / two-valued blobs.
    if (degenerate) {
        asm volatile("" ::: "memory");
        JacQ p2 = A[(size_t)a * stride + lane];
#pragma unroll 1
        for (uint32_t k = 0; k < runs; k++) p2 = dbl(p2);
        const JacQ q2 = A[(size_t)b * stride + lane];
        const JacQ d = add_slow(p2, q2, true);
        A[(size_t)dst * stride + lane] = add_slow(p2, q2, false);
        A[(size_t)(fl >> 16) * stride + lane] = d;
    }
}
// additions, subtractions and runs of doublings of one step, one wave per operation.  Blocks are dealt in blockIdx order, x
// fastest: x = lane group, y = operation, and the schedule lists a step's operations longest first (g1_linmap.hpp), so every
// group's long operations start first and the launch ends on short ones.
__global__ __launch_bounds__(64, 2) void k_slp_add(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    slp_cheap_op(A, stride, blockIdx.x * 64 + threadIdx.x, dst, a, b, fl);
}

// The cheap operations of ONE lane group (<= 64 blobs: BASELINE config 4's and 5's per-GPU shares) with four lanes per blob
// (g1_coop.hpp): a level is then a few hundred waves on an idle chip, each a single addition -- 16.5 multiplication times for one
// lane, 5.5 for a quad; a doubling run likewise 3.5 per doubling instead of 6.5.  The sum-and-difference pair is two
// quad additions (11 against the shared form's 20).  16 blobs per wave; every lane of a quad stores the same result.
__global__ __launch_bounds__(64, 2) void k_slp_add_coop(JacQ* __restrict__ A, int stride, const uint32_t* __restrict__ words, int lanes) {
    const uint32_t* w = words + (size_t)blockIdx.y * 4;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(w[0]), a = __builtin_amdgcn_readfirstlane(w[1]),
                   b = __builtin_amdgcn_readfirstlane(w[2]), fl = __builtin_amdgcn_readfirstlane(w[3]);
    const int lane = blockIdx.x * 16 + (threadIdx.x >> 2), quad = threadIdx.x & 3;
    if (lane >= lanes) return;
    JacQ r = A[(size_t)a * stride + lane];
    const uint32_t runs = (fl & 2u) ? b : (fl >> 3) & 31u;
#pragma unroll 1
    for (uint32_t k = 0; k < runs; k++) r = coop_dbl(r, quad);
    if (!(fl & 2u)) {
        const JacQ q = A[(size_t)b * stride + lane];
        if (fl & 4u) {
            const JacQ d = coop_add(r, q, true, quad);
            A[(size_t)(fl >> 16) * stride + lane] = d;
            r = coop_add(r, q, false, quad);
        } else {
            r = coop_add(r, q, (fl & 1u) != 0, quad);
        }
    }
    A[(size_t)dst * stride + lane] = r;
}

// ---------------------------------------------------------------------------------------------------------------
// One launch for a whole PHASE of the program's cheap operations (the dependency levels in front of the constant
// multiplications -- 11 since the doubling runs are folded into their consumers, 21 before -- and the 14 behind them) instead
// of one launch per level.  A level-by-level sequence of launches leaves the
// chip part empty at every level boundary (2,800 waves on 2,048 wave slots: a second round 37 % full, 41 times over) and,
// for a single 64-blob lane group, pays a launch and a tail per level.  Lanes are blobs, so the dependency structure is
// PER LANE GROUP: level l + 1 of a group needs level l of THAT group only.  The walker hands out operations by ticket:
//   * the lane groups are dealt to S <= 8 shards (shard = blockIdx % S: blocks b and b + 8 were observed to share an XCD, so
//     a shard's counters and points mostly stay in one L2 -- speed only, nothing depends on it);
//   * a shard's operations are numbered level-major across its groups; a wave draws the next number with one atomic add,
//     waits until the previous level OF THAT GROUP is complete (a counter per (group, level)), runs the operation, makes
//     its result visible (agent-scope release) and bumps the counter of its own (group, level).
// Tickets are drawn in dependency order by waves that are already running, so every wait is for a wave that holds an
// earlier ticket and never waits on a later one: no deadlock, whatever the residency.  Every spin is bounded all the same
// (a stuck counter sets the error word and the wave leaves; the host turns that into a device error).
// Visibility follows MI355X_MICROARCH.md ("inter-workgroup visibility"): per-XCD L2s are not coherent and a CU's L1 is never
// refreshed by another CU's stores, so producer = plain stores, agent-scope release fence, s_waitcnt vmcnt(0) (kept in asm:
// the compiler may drop the wait it thinks redundant), relaxed agent-scope add; consumer = relaxed agent-scope poll, ONE
// agent-scope acquire, then plain loads.  A wave is its own workgroup here, so no barrier is involved.
struct SlpWalk {
    const uint32_t* words;      // 4 per operation, level-major (the schedule's own order)
    const int* level_first;     // [n_levels] first operation of the level (index into words / 4)
    const int* level_count;     // [n_levels]
    int n_levels, n_groups, n_shards;
    int spin_limit;             // polls before a waiting wave gives up (about a second)
    int debug;
    // counters (zeroed before the launch): ticket[shard] at stride 32 ints, then done[group][level] at stride 16 ints, then error
    int* sync;
};
__device__ __forceinline__ int* slp_ticket(const SlpWalk& w, int shard) { return w.sync + 32 * shard; }
__device__ __forceinline__ int* slp_done(const SlpWalk& w, int group, int level) {
    return w.sync + 32 * 8 + 16 * ((size_t)group * w.n_levels + level);
}
__device__ __forceinline__ int* slp_error(const SlpWalk& w) { return w.sync + 32 * 8 + 16 * (size_t)w.n_groups * w.n_levels; }

__global__ __launch_bounds__(64) void k_slp_walk(JacQ* __restrict__ A, int stride, SlpWalk w) {
    const int shard = blockIdx.x % w.n_shards;
    const int groups_here = (w.n_groups - shard + w.n_shards - 1) / w.n_shards;  // groups shard, shard + S, shard + 2 S, ...
    if (groups_here <= 0) return;
    int total = 0;
    for (int l = 0; l < w.n_levels; l++) total += w.level_count[l];
    const int tickets = total * groups_here;
    if (w.debug && blockIdx.x == 0 && threadIdx.x == 0) {
        int* dbg = slp_error(w) + 4;
        dbg[8] = total; dbg[9] = tickets; dbg[10] = w.n_levels; dbg[11] = w.level_count[0];
    }
    for (int guard = 0;; guard++) {
        if (guard > 200000) {  // no wave can have this many operations: the tables are corrupt
            if (threadIdx.x == 0) __hip_atomic_store(slp_error(w), 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        int t = 0;
        if (threadIdx.x == 0) t = __hip_atomic_fetch_add(slp_ticket(w, shard), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= tickets) return;
        // ticket -> (level, group of this shard, operation): level-major across the shard's groups
        int level = 0, base = 0;
        for (;; level++) {
            const int span = w.level_count[level] * groups_here;
            if (t < base + span) break;
            base += span;
        }
        const int cnt = w.level_count[level];
        const int gl = (t - base) / cnt, i = (t - base) - gl * cnt;
        const int group = shard + gl * w.n_shards;
        if (level > 0) {  // the previous level of this group must be complete
            const int need = w.level_count[level - 1];
            const int* flag = slp_done(w, group, level - 1);
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(16);
                // a stuck counter (or another wave's verdict that one is stuck) ends the walk: fail loudly, never hang the GPU
                if ((++spins & 1023) == 0 && (spins > w.spin_limit || __hip_atomic_load(slp_error(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (threadIdx.x == 0) {
                        __hip_atomic_store(slp_error(w), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        int* dbg = slp_error(w) + 4;  // first reporter: ticket, level, group, operation, counter seen, needed
                        if (__hip_atomic_fetch_add(slp_error(w) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                            dbg[0] = t; dbg[1] = level; dbg[2] = group; dbg[3] = i;
                            dbg[4] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dbg[5] = need; dbg[6] = tickets; dbg[7] = groups_here;
                        }
                    }
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const uint32_t* op = w.words + ((size_t)w.level_first[level] + i) * 4;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(op[0]), a = __builtin_amdgcn_readfirstlane(op[1]),
                       b = __builtin_amdgcn_readfirstlane(op[2]), fl = __builtin_amdgcn_readfirstlane(op[3]);
        const int lane = group * 64 + threadIdx.x;
        slp_cheap_op(A, stride, lane, dst, a, b, fl);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(slp_done(w, group, level), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_g1slp() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_slp_add));
}
// kind: 3 multiplication by a constant, anything else the mixed addition / subtraction / doubling launch (linmap::OpKind)
void g1_slp_launch(int kind, void* arena, int stride, const uint32_t* words, int count, const void* naf, const Fp12w& beta,
                   hipStream_t st, int lanes, int coop_lanes) {
    if (lanes <= 0) lanes = stride;  // (a sub-range of the lanes: arena already points at its first lane, stride stays the arena's)
    const dim3 grid((unsigned)count, (unsigned)(lanes / 64));
    if (kind == 3) {
        Fp b384;
        for (int i = 0; i < 12; i++) b384.v[i] = beta.v[i];
        // coop_lanes: the blobs that are really there when they are few enough for four lanes each (<= 16: one quad wave per operation)
        // or two (<= 32: still one wave per operation)
        if (coop_lanes > 16 && coop_points_max() > 0)
            k_slp_mulc_coop2<<<dim3((unsigned)count, (unsigned)((coop_lanes + 31) / 32)), 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf, fq_from_fp(b384), coop_lanes);
        else if (coop_lanes > 0 && coop_points_max() > 0)
            k_slp_mulc_coop<<<dim3((unsigned)count, (unsigned)((coop_lanes + 15) / 16)), 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf,
                                                                                                     fq_from_fp(b384), coop_lanes);
        else k_slp_mulc<<<grid, 64, 0, st>>>((JacQ*)arena, stride, words, (const uint32_t*)naf, fs_from_fp(b384));
    } else {
        // one lane group and few enough operations for every quad wave to have a SIMD of its own: four lanes per blob
        if (lanes == 64 && count * 4 <= 1024 && coop_points_max() > 0)
            k_slp_add_coop<<<dim3(4u, (unsigned)count), 64, 0, st>>>((JacQ*)arena, stride, words, lanes);
        else k_slp_add<<<dim3((unsigned)(lanes / 64), (unsigned)count), 64, 0, st>>>((JacQ*)arena, stride, words);
    }
}
size_t g1_slp_walk_sync_ints(int n_groups, int n_levels) { return 32 * 8 + 16 * (size_t)n_groups * n_levels + 32; }
// one phase of cheap operations in one launch; sync = g1_slp_walk_sync_ints ints (zeroed here); wave_slots = what the chip holds
void g1_slp_walk(void* arena, int stride, const uint32_t* words, const int* level_first, const int* level_count, int n_levels,
                 int max_level_count, int total_ops, int* sync, int wave_slots, hipStream_t st) {
    const int n_groups = stride / 64;
    SlpWalk w;
    w.words = words;
    w.level_first = level_first;
    w.level_count = level_count;
    w.n_levels = n_levels;
    w.n_groups = n_groups;
    w.n_shards = n_groups < 8 ? n_groups : 8;
    w.sync = sync;
    w.spin_limit = 1 << 20;
    w.debug = getenv("ETH_KZG_AMD_SLP_DEBUG") != nullptr;
    if (const char* e = getenv("ETH_KZG_AMD_SLP_SPIN_LIMIT")) w.spin_limit = atoi(e);
    (void)hipMemsetAsync(sync, 0, g1_slp_walk_sync_ints(n_groups, n_levels) * sizeof(int), st);
    // enough waves to keep every level's operations of every group in flight, at most what the chip holds at once
    long want = (long)max_level_count * n_groups;
    if (want > wave_slots) want = wave_slots;
    if (want > (long)total_ops * n_groups) want = (long)total_ops * n_groups;
    want = (want + w.n_shards - 1) / w.n_shards * w.n_shards;
    k_slp_walk<<<(unsigned)want, 64, 0, st>>>((JacQ*)arena, stride, w);
}
}  // namespace launch
}  // namespace kzg
