// Host-callable launchers of the HIP kernels (one translation unit per kernel family so that
// hipcc compiles them in parallel).  Pointers are device pointers; element types are noted.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace kzg {
struct Fr8;    // 8 x u32, Montgomery (engine.hpp)
struct Fp12w;  // 12 x u32, Montgomery
namespace launch {

void init_attributes();  // opt in to 128 KiB dynamic LDS for the NTT kernels
// Load every code object of the library now (k_msm.hip).  HIP loads a translation unit's code object when one of its kernels
// is first launched, and that load allocates device memory: a caller whose first recovery / first wide-table MSM happens while
// the table builder thread sits in a hipMalloc (seconds, when the driver is still wiping memory another process freed) would
// wait for it -- kernel launches, copies and events themselves do not (tools/alloc_test/probe_stall.cpp).
void preload_code_objects();

// k_ntt.hip
// The three Fr stages of the prover compute in the unsaturated 9 x 29-bit form (fr29.hpp): w29 = the twiddle table in that
// form (8192 x 9 words, built on the host by ntt_twiddles29 from the saturated table); inputs / outputs keep the engine's
// stored forms (saturated Montgomery coefficients, plain-integer scalars, big-endian bytes).
constexpr size_t SIZEOF_FR29 = 36;
void ntt_twiddles29(const void* w8192_mont_host /*Fr[8192]*/, void* out_host /*8192 x 36 B*/);
void blob_to_coeffs(int n, const uint8_t* blobs, void* coeffs /*Fr*/, void* canon /*Fr or null*/, int* status,
                    const void* w29, const Fr8& n_inv, hipStream_t st);
void coeffs_to_cells(int n, const void* coeffs, uint8_t* cells, const void* w29, hipStream_t st);
// every scalar leaves as its balanced GLV halves (what launch::glv_split would make of it)
void fk20_scalars(int n, const void* coeffs, void* scalars, const void* w29, const Fr8& inv128, int segs, const Fr8* seg_shifts,
                  hipStream_t st);
void test_ntt4096(const uint8_t* in, uint8_t* out, const void* w29, const Fr8& n_inv, int inverse_dit, hipStream_t st);
void test_scalars_be(const uint8_t* in, void* out, size_t n, hipStream_t st);
void test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp, hipStream_t st);

// k_msm.hip
// A window table as the kernels see it: a device array of block pointers -- two per group (blocks[2 group + upper]: the lower
// ceil(W / 2) windows, the upper rest; inside a block [window][base][digit]) -- and the range [g0, g0 + gcnt) of groups this launch covers (of the n_groups MSMs per slice the scalars hold).
// Tables are allocated in pieces and published group by group while they are built (engine.hip: SharedTable): a launch over
// the groups already built on the new table and one over the rest on the old table make one MSM stage.
struct TabBlocks {
    const void* const* blocks;
    int g0, gcnt;
};
// GLV tables (packed 96-B entries, W = glv_windows(c) windows of c bits over the 128-bit half scalars; k_msm_glv.inc, one
// translation unit per width): mode 0 flat, 1 windowed, 2 four chunks per MSM.  The scalars must be stored as balanced GLV
// halves (glv_split, or k_fk20_scalars' fused split).  Entries and sums are in the signed 13 x 30-bit field (curve30.hpp).
// A GLV table is NAMED by its nominal width c and has W = ceil(128 / c) windows -- but W windows of c bits cover W c >= 128 bits, and
// at widths that do not divide 128 the surplus doubles the table for nothing (nine windows of 15 bits = 135 bits: 116 GB; the same
// nine windows as two of 15 and seven of 14 bits = 128 bits: 58 GB, the same 18 gathered additions).  So the windows have MIXED
// widths (round 5): b = floor(128 / W) bits, the lowest 128 - W b of them one bit more.  c = 16 and c = 8 divide 128: unchanged.
constexpr int glv_windows(int c) { return (128 + c - 1) / c; }
constexpr int glv_lower_windows(int c) { return (glv_windows(c) + 1) / 2; }  // windows in the lower block of a group
constexpr int glv_base_bits(int c) { return 128 / glv_windows(c); }
constexpr int glv_wide_windows(int c) { return 128 - glv_windows(c) * glv_base_bits(c); }  // the lowest windows, one bit wider
constexpr int glv_window_bits(int c, int w) { return glv_base_bits(c) + (w < glv_wide_windows(c) ? 1 : 0); }
constexpr int glv_window_lo(int c, int w) { return w * glv_base_bits(c) + (w < glv_wide_windows(c) ? w : glv_wide_windows(c)); }  // its first bit
// entries per base of the windows [w0, w1): sum of 2^(bits - 1)
constexpr size_t glv_entries_per_base(int c, int w0, int w1) {
    size_t n = 0;
    for (int w = w0; w < w1; w++) n += (size_t)1 << (glv_window_bits(c, w) - 1);
    return n;
}
constexpr int GLV_WIDTHS[] = {16, 15, 14, 12, 8};  // widest first: the order the engine tries them in
bool glv_width_supported(int c);
void glv_split(void* scalars, size_t n, hipStream_t st);
// Point arrays between the G1 stages come in two formats (FMT_*): the signed 13 x 30-bit form (JacS, 156 B: the MSM sums, the linear
// map's arena, the circulant form, the commitment's fold and the input of the compression -- one field under the whole prover, recovery
// and commitment path) and the 14 x 29-bit form (JacQ, 168 B: set-up, verification, the stage hooks, ETH_KZG_AMD_ARENA_SIGNED=0).
constexpr int FMT_JACQ = 0, FMT_JACS = 1;
void msm_glv(int c, int mode, const void* scalars, const TabBlocks& table, void* out /*JacQ or JacS by out_fmt*/, int n_groups, int n_slices, int nb, int out_stride,
             int brp_bits, const Fp12w& beta, hipStream_t st, int out_fmt = FMT_JACQ);
// k_table.hip
#ifdef TABS_STRIDE_128
constexpr size_t SIZEOF_TABP = 128;   // EXPERIMENT: a GLV table entry on a line of its own (curve30.hpp)
#else
constexpr size_t SIZEOF_TABP = 96;    // a packed GLV table entry
#endif
size_t table_glv_entries(int c, int n_groups, int nb);
size_t table_glv_side_bytes(int c, int n_groups, int nb);
// scratch: 168 B per entry of the chunk; side: table_glv_side_bytes; false if the width is not built in.
// blocks: device array of 2 * n_groups block pointers of THIS chunk's groups (lower / upper windows of each)
bool build_table_glv(int c, const void* bases, void* const* blocks, void* scratch, void* side, int n_groups, int nb, int* err, hipStream_t st);

// k_g1fft.hip
void g1_fft_layer(void* X, int stride, int half, int tw_step, int inverse, int mode, const void* tw, const Fp12w& beta,
                  hipStream_t st);

// k_g1slp.hip: one launch of the straight-line program of the FK20 proofs map (g1_linmap.hpp); kind = linmap::OpKind
void g1_slp_launch(int kind, void* arena, int stride, const uint32_t* words, int count, const void* naf, const Fp12w& beta,
                   hipStream_t st, int lanes = 0 /* lanes to run (a multiple of 64, from the arena pointer on); 0: all `stride` of them */,
                   int coop_lanes = 0 /* > 0: the batch has this many blobs (<= 64): the constant multiplications take four (<= 16) or two lanes per blob */,
                   int fmt = FMT_JACQ /* the arena's point format; FMT_JACS: the kernels of the signed field */,
                   int n_active = 0 /* blobs that are really there (0: all `lanes`): the padding lanes behind them are skipped */);

// k_g1misc.hip
void g1_set_inf(void* X, size_t n, hipStream_t st, int fmt = FMT_JACQ);
int coop_points_max();  // largest launch (points) that takes the four-lanes-per-point kernels (k_g1misc.hip)
void spin(uint64_t wall_clock_ticks, hipStream_t st);  // one wave, resident for that many ticks of the constant-rate device clock
void g1_compress(const void* X, uint8_t* out, int n_pos, int stride, int n_slices, hipStream_t st, int fmt = FMT_JACQ);
void g1_sum_positions(void* X, int n_pos, int stride, int n_slices, hipStream_t st, int fmt = FMT_JACQ);
// subgroup_check: 0 none, 1 endomorphism test, 2 definitional [r]P == O
void g1_decompress(const uint8_t* in, void* out /*G1Affine*/, int* status, int n, int subgroup_check, const Fp12w& beta,
                   hipStream_t st);
// two segments in one launch, production checks (on curve + subgroup)
void g1_decompress2(const uint8_t* in0, void* out0, int* status0, int n0, const uint8_t* in1, void* out1, int* status1, int n1,
                    const Fp12w& beta, hipStream_t st);
// its two halves as separate launches: decode + on-curve test (status 0 / 1), then the subgroup test of the decoded points (0 -> 0 / 2)
void g1_decode2(const uint8_t* in0, void* out0, int* status0, int n0, const uint8_t* in1, void* out1, int* status1, int n1,
                const Fp12w& beta, hipStream_t st);
void g1_subgroup2(const void* pts0, int* status0, int n0, const void* pts1, int* status1, int n1, const Fp12w& beta, hipStream_t st);
void fk20_srs_vectors(const void* srs, void* X, hipStream_t st);
void fk20_gather_bases(const void* X, void* bases, hipStream_t st);
void test_load_points(const uint8_t* in, void* X, int n_lanes, int stride, hipStream_t st);
void test_recompress(const void* pts, uint8_t* out, int n, hipStream_t st);

// twiddle recoding shared by engine.hip (host) and k_g1fft.hip (device)
constexpr int TWIDDLE_WNAF_W = 5;   // window width of the non-adjacent form (table of 2^(w-2) odd multiples)
constexpr int TWIDDLE_WORDS = 33;   // 132 signed digit bytes per 128-bit half
// k_g1circ.hip
constexpr int CIRC_LANES = 256;
size_t g1_circ_table_bytes(int n, int T);
void g1_circ128(void* X, int stride, int n, int segs, void* D, int T, const void* terms, int per_lane, const Fp12w& beta, hipStream_t st, int fmt = FMT_JACQ);
// k_verify.hip
void init_attributes_verify();
// slot_of: destination cell slot per input cell (null = identity); status_of: status word per input cell (null = word 0)
void cells_to_fr(const uint8_t* cells, void* evals, const int* slot_of, int* status, const int* status_of, const int* src_of, int n,
                 hipStream_t st);
void rec_vanishing_poly(const uint32_t* present, const void* w8192, void* zp, int* deg, int R, hipStream_t st);
void rec_vanishing(const void* zp, const int* deg, const void* w8192, const Fr8& seven64, void* zeval, void* zcinv, int R,
                   hipStream_t st);
void verify_scalars(const Fr8* pow_table24, int k0, const int* cell_idx, const void* w8192, void* rp_mont, void* s1, void* s2,
                    int n, hipStream_t st);
void verify_weights(const void* rp_mont, const int* row, void* weights, int n, int m, hipStream_t st);
void interp(const void* evals, const int* cell_idx, const void* rp_mont, const void* w8192, const Fr8& inv64, void* partial,
            int nblocks, void* out_neg_canon, int n, hipStream_t st);
void interp_cells(const void* evals, const int* cell_idx, const void* w8192, const Fr8& inv64, void* coef, int n, hipStream_t st);
void interp_sum(const void* coef, const void* rp_mont, void* partial, int nblocks, void* out_neg_canon, int n, hipStream_t st);
// large verification batches: byte-shifted point copies built before the challenge is known (k_verify.hip)
size_t pip_shift_workspace_bytes(int n_max);
void pip_shift_prepare(const void* points, int n_pts, int n_max, void* workspace, const Fp12w& beta, hipStream_t st);
// the same and, in the same launch, the subgroup tests of two point segments (status 0 -> 0 / 2): one stream per verification
void pip_shift_prepare_and_subgroup(const void* points, int n_pts, int n_max, void* workspace, const void* pts0, int* status0, int n0,
                                    const void* pts1, int* status1, int n1, const Fp12w& beta, hipStream_t st);
void msm_pippenger2_shifted(const void* sc0, int n0, const void* sc1, int n1, int n_max, void* workspace, void* out_jacq2,
                            hipStream_t st);  // out: two JacQ (SIZEOF_JACQ each)
size_t pip_workspace_bytes(int n_max);
void copy_affine(const void* src, void* dst, int n, hipStream_t st);
void msm_pippenger2(const void* points, const void* sc0, int n0, const void* sc1, int n1, void* workspace, void* out_affine2,
                    const Fp12w& beta, hipStream_t st);
void rec_dit_half(int R, const void* V, const void* fac, void* T, const void* w8192, hipStream_t st);
void rec_dit_last(int R, const void* T, const void* shift, const Fr8& n_inv, void* U, void* coeffs, int* status,
                  const void* w8192, int final_pass, hipStream_t st);
void rec_dif_half(int R, const void* U, const void* fac, void* V, const void* w8192, hipStream_t st);

// k_verify_many.hip: many independent verifications in one pass (per-problem scalars, one lane per scalar multiplication)
void vm_scalars(const void* pow_tables /*[B][24] Fr*/, const int* batch_of, const int* pos_in_batch, const int* cell_idx, const void* w8192,
                void* rp_mont, void* s1, void* s2, int n, hipStream_t st);
void vm_weights(const void* rp_mont, const int* row, const int* row_batch, const int* cell_start, void* weights, int m, hipStream_t st);
void vm_interp_sum(const void* coef, const void* rp_mont, const int* cell_start, void* out_neg_canon, int n_batches, hipStream_t st);
// pts = [proofs n | commitments m] (G1Affine); prod[2n + m] JacQ: s1[e] pi_e | s2[e] pi_e | w[j] C_j
void vm_mul(const void* pts, const void* s1, const void* s2, const void* wts, void* prod, int n, int m, const Fp12w& beta, hipStream_t st);
// small passes: every scalar multiplication of the pass (incl. the 64 interpolation terms isc[b][j] SRS_j per problem, prod[2n + m + 64 b + j])
// and the subgroup tests of [proofs n | commitments m] (status: 0 -> 0 / 2) in ONE launch; then the per-problem sums
void vm_mul_small(const void* pts, const void* s1, const void* s2, const void* wts, const void* isc, const void* srs, void* prod, int n, int m,
                  int n_batches, int* status, const Fp12w& beta, hipStream_t st);
void vm_reduce_small(const void* prod, const int* cell_start, const int* row_start, void* out, int n, int m, int n_batches, hipStream_t st);
void vm_reduce(const void* prod, const void* icommit, const int* cell_start, const int* row_start, void* out /*[B][2] JacQ*/, int n,
               int n_batches, hipStream_t st);
// out2[j] = sum_b rho_b sums[b][j] (rho: 4 words per problem, 0 excludes it); prod: scratch of 2 B JacQ
void vm_fold(const void* sums, const uint32_t* rho, void* prod, void* out2, int n_batches, const Fp12w& beta, hipStream_t st);
// out[r][2] = the weighted pairs summed over problems ranges[r][0] .. ranges[r][1] - 1 (prod as vm_fold left it)
void vm_fold_ranges(const void* prod, const int* ranges /*[n_ranges][2], device*/, void* out /*[n_ranges][2] JacQ*/, int n_ranges, hipStream_t st);

// k_4844.hip
void quotient_by_linear(int n, const void* coeffs, const void* z_mont, void* quotient, void* y_out, hipStream_t st);

constexpr size_t SIZEOF_FR = 32, SIZEOF_G1AFFINE = 96, SIZEOF_G1JAC = 144;
constexpr size_t SIZEOF_JACS = 156;  // signed 13 x 30-bit Jacobian point (curve30.hpp)
constexpr size_t SIZEOF_AFFQ = 112, SIZEOF_JACQ = 168;  // unsaturated 14 x 29-bit forms (curve29.hpp): affine points, FFT arrays

}  // namespace launch
}  // namespace kzg
