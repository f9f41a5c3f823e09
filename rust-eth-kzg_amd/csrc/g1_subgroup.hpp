// The G1 subgroup test in the unsaturated field, shared by the decoder (k_g1misc.hip) and the fused pre-challenge kernel of a
// verification (k_verify.hip: k_pip_shift_subgroup).  Reference: blst's endomorphism test behind G1Affine::from_compressed
// (crates/serialization/src/lib.rs:69-99 deserialize_compressed_g1).
#pragma once
#include "curve29.hpp"

namespace kzg {

// [|z|] P for a general Jacobian P (|z| = 0xd201000000010000: 63 doublings + 5 additions)
__device__ __forceinline__ JacQ mul_by_z_abs_q(const JacQ& p) {
    constexpr uint64_t Z = 0xd201000000010000ULL;
    JacQ acc = p;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        acc = dbl(acc);
        if ((Z >> i) & 1) acc = add(acc, p);
    }
    return acc;
}
// Scott's test for a curve point P != O: [z^2]P - P == phi(P) = (beta x, y)
__device__ __forceinline__ bool g1_in_subgroup_q(const AffQ& pa, const Fq<1>& beta) {
    const JacQ p = to_jacq(pa);
    const JacQ q = mul_by_z_abs_q(mul_by_z_abs_q(p));
    const JacQ r = add_mixed(q, pa, true);
    if (is_inf(r)) return false;
    const Fq<2> zz = sqr(r.z);
    if (!is_zero_slow(sub(r.x, mul(mul(pa.x, beta), zz)))) return false;
    if (!is_zero_slow(sub(r.y, mul(pa.y, mul(zz, r.z))))) return false;
    return true;
}
}  // namespace kzg
