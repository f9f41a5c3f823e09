// Fixed-base MSM over window tables (stage D of compute_cells_and_kzg_proofs; commitment MSM): the kernels live in k_msm_glv.inc
// (one translation unit per table width); this unit holds the scalar split they share and the dispatch by width.
// Rounds 1-4 also had MSM kernels over PLAIN tables here (windows over the full 255-bit scalar, 128-B entries, the 14-digit
// field): fall-back widths, the commitment table, use_precomp = false.  All three are GLV tables now.
#include "engine.hpp"
#include "kcommon.hpp"
#include "launch.hpp"
#include "glv.hpp"

namespace kzg {

// k = k1 + k2 lambda with |k1|, |k2| < 2^127 (glv.hpp), in place: the scalars of an MSM whose producer has not split them already
__global__ void k_glv_split(Fr* __restrict__ scalars, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = scalars[i];
    uint32_t out[8];
    glv_split_balanced(k, out);
    Fr o;
#pragma unroll
    for (int l = 0; l < 8; l++) o.v[l] = out[l];
    scalars[i] = o;
}
namespace launch {
// the code object of this translation unit is loaded now (HIP loads a code object on the first launch of one of its kernels, and
// that load is an allocation: it would wait behind a table piece the builder thread is allocating)
void preload_k_msm() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_glv_split));
}
// k_msm_glv.inc (one translation unit per window width) holds the GLV kernels; the split of the scalars is shared
void glv_split(void* scalars, size_t n, hipStream_t st) {
    k_glv_split<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((Fr*)scalars, n);
}
#define GLV_DECL(w) void msm_glv_w##w(int, const void*, const TabBlocks&, void*, int, int, int, int, int, const Fp12w&, hipStream_t, int); void preload_k_msm_glv##w();
GLV_DECL(8) GLV_DECL(12) GLV_DECL(14) GLV_DECL(15) GLV_DECL(16)
#undef GLV_DECL
void preload_k_ntt(); void preload_k_g1fft(); void preload_k_g1circ(); void preload_k_g1misc(); void preload_k_verify();
void preload_k_verify_many(); void preload_k_4844(); void preload_k_g1slp(); void preload_k_table();
void preload_code_objects() {
    preload_k_msm(); preload_k_msm_glv8(); preload_k_msm_glv12(); preload_k_msm_glv14(); preload_k_msm_glv15(); preload_k_msm_glv16();
    preload_k_ntt(); preload_k_g1fft(); preload_k_g1circ(); preload_k_g1misc(); preload_k_verify(); preload_k_verify_many();
    preload_k_4844(); preload_k_g1slp(); preload_k_table();
}
bool glv_width_supported(int c) {
    for (int w : GLV_WIDTHS)
        if (w == c) return true;
    return false;
}
void msm_glv(int c, int mode, const void* scalars, const TabBlocks& table, void* out, int n_groups, int n_slices, int nb, int out_stride,
             int brp_bits, const Fp12w& beta, hipStream_t st, int out_fmt) {
    switch (c) {
        case 8: return msm_glv_w8(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st, out_fmt);
        case 12: return msm_glv_w12(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st, out_fmt);
        case 14: return msm_glv_w14(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st, out_fmt);
        case 15: return msm_glv_w15(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st, out_fmt);
        default: return msm_glv_w16(mode, scalars, table, out, n_groups, n_slices, nb, out_stride, brp_bits, beta, st, out_fmt);
    }
}
}  // namespace launch
}  // namespace kzg
