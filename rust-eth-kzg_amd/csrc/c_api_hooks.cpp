// The C symbols of the stage-level test hooks (include/c_eth_kzg_test_hooks.h).  Linked into libc_eth_kzg_hooks.so only -- the
// library tests/ loads; the product library libc_eth_kzg.so neither defines nor exports them (csrc/Makefile).
#include "../../include/c_eth_kzg_test_hooks.h"
#include "c_ctx.hpp"

#include <cstdio>
#include <cstdlib>

namespace {
kzg::Engine* eng(const DASContext* ctx) {
    if (!ctx || !ctx->engine) {
        fprintf(stderr, "c_eth_kzg: context pointer is null\n");
        abort();
    }
    return ctx->engine;
}
}  // namespace

extern "C" {

int eth_kzg_amd_test_fr_ntt4096(const DASContext* ctx, const uint8_t* in, uint8_t* out, int inverse_dit) {
    return eng(ctx)->test_fr_ntt4096(in, out, inverse_dit);
}
int eth_kzg_amd_test_g1_fft128(const DASContext* ctx, const uint8_t* in, uint8_t* out, int n_lanes, int inverse) {
    return eng(ctx)->test_g1_fft128(in, out, n_lanes, inverse);
}
int eth_kzg_amd_test_fixed_msm(const DASContext* ctx, const uint8_t* scalars, int n_msm, uint8_t* out) {
    return eng(ctx)->test_fixed_msm(scalars, n_msm, out);
}
int eth_kzg_amd_test_g1_decompress(const DASContext* ctx, const uint8_t* in, int n, int subgroup_check, int32_t* status,
                                   uint8_t* out) {
    return eng(ctx)->test_g1_decompress(in, n, subgroup_check, status, out);
}
int eth_kzg_amd_test_field_mul(const DASContext* ctx, const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp) {
    return eng(ctx)->test_field_mul(a, b, out, n, is_fp);
}

}  // extern "C"
