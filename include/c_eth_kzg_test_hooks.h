/* Stage-level hooks of libc_eth_kzg for the kernel parity tests under tests/ -- NOT part of the drop-in ABI
 * (bindings/c of the reference has nothing like them).  Kept out of c_eth_kzg.h so that consumers never see them. */
#ifndef C_ETH_KZG_TEST_HOOKS_H
#define C_ETH_KZG_TEST_HOOKS_H
#include "c_eth_kzg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* canonical big-endian encodings; return 0 on success */
int eth_kzg_amd_test_fr_ntt4096(const DASContext *ctx, const uint8_t *in, uint8_t *out, int inverse_dit);
int eth_kzg_amd_test_g1_fft128(const DASContext *ctx, const uint8_t *in, uint8_t *out, int n_lanes, int inverse);
int eth_kzg_amd_test_fixed_msm(const DASContext *ctx, const uint8_t *scalars, int n_msm, uint8_t *out);
int eth_kzg_amd_test_g1_decompress(const DASContext *ctx, const uint8_t *in, int n, int subgroup_check, int32_t *status,
                                   uint8_t *out);
int eth_kzg_amd_test_field_mul(const DASContext *ctx, const uint8_t *a, const uint8_t *b, uint8_t *out, int n,
                               int is_fp);

#ifdef __cplusplus
}
#endif
#endif /* C_ETH_KZG_TEST_HOOKS_H */
