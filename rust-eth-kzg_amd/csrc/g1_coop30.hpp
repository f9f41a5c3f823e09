// Two lanes per point operation in the signed 13 x 30-bit field (fp30.hpp / curve30.hpp): the chain forms of the constant
// multiplication -- halved doubling and mixed addition with a table entry on the common Z (g1_mulc30.hpp) -- for batches of
// 17 .. 64 blobs, where a constant multiplication of the G1 linear map is a dependent chain of 128 doublings and ~43 additions on a
// SIMD that has nothing else to do (BASELINE configs 4 and 5: the per-GPU shares).  Same idea as g1_coop.hpp's pair forms in the
// 14 x 29-bit field: the lanes 2b and 2b + 1 of a wave both hold blob b's operands, compute one level's independent products side by
// side (operands picked by lane, ONE multiplication issued), exchange them with pair-broadcast DPP moves (13 per field element) and
// run the linear steps and the last, fused reductions redundantly:
//     doubling      X^2 | Y^2  ->  X Y^2 | Y Z  ->  H^2 - 2 M  ->  H (M - X3) - B^2            4 reductions deep (7 alone):  1,406 multiply-adds (2,054)
//     mixed add     Z1^2 | y2 Z1  ->  x2 Z1Z1 - X1 | (y2 Z1) Z1Z1 - Y1  ->  H^2 | H Z1  ->  H^3 | X1 H^2  ->  X3  ->  Y3
//                                                                                      6 reductions deep (10 alone): 2,228 (3,497)
// against 1,624 and 2,233 of the 14-digit pair forms: 276 k instead of 304 k multiply-adds per multiplication chain.
// Formulas, bounds and exits are those of curve30.hpp: dbl_half and add_mixed(JacS, AffT) -- the exact slow path is taken by both lanes
// of a pair together (Z3 is broadcast before it is tested).
#pragma once
#include "curve30.hpp"

namespace kzg {

// both lanes of a pair (2k, 2k + 1) receive the value of the pair's lane J
template <int J, int B, int F>
__device__ __forceinline__ Fs<B, F> pair_bcast(const Fs<B, F>& a) {
    static_assert(J == 0 || J == 1, "lane of the pair");
    constexpr int CTRL = J == 0 ? 0xA0 : 0xF5;  // quad_perm [0,0,2,2] / [1,1,3,3]
    Fs<B, F> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], CTRL, 0xf, 0xf, true);
    return r;
}
template <int B, int F>
__device__ __forceinline__ Fs<B, F> select(bool c, const Fs<B, F>& a, const Fs<B, F>& b) {
    Fs<B, F> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// curve30.hpp: dbl_half by a pair; l0 = this lane is the pair's even lane.  Both lanes must be active and hold the same p.
__device__ __forceinline__ JacS coop2_dbl_half(const JacS& p, bool l0) {
    // level 1: lane 0 A = X^2, lane 1 B = Y^2
    const Fs<1, DC> r1 = sqr(select(l0, p.x, relax<4, DC>(p.y)));
    const Fs<1, DC> A = pair_bcast<0>(r1), B = pair_bcast<1>(r1);
    // level 2: lane 0 M = B X, lane 1 Z3 = Y Z
    const Fs<1, DC> r2 = mul(select(l0, B, p.y), select(l0, p.x, relax<4, DC>(p.z)));
    const Fs<1, DC> M = pair_bcast<0>(r2), Z3 = pair_bcast<1>(r2);
    const Fs<2, DC> H = half_of_triple(A);
    // levels 3 and 4 on both lanes (each needs the one before)
    JacS r;
    r.x = relax<4, DC>(sqr_inj<-2, DC>(H, M));                 // <= 3
    r.y = mul_add<DC>(H, sub_lazy(M, r.x), neg(B), B);         // H (M - X3) - B^2, one reduction
    r.z = Z3;
    return r;
}

// curve30.hpp: add_mixed(JacS, AffT) by a pair: p + q, or p - q when negq (wave-uniform)
__device__ __forceinline__ JacS coop2_add_mixed(const JacS& p, const AffT& q, bool negq, bool l0) {
    // level 1: lane 0 Z1^2, lane 1 T = (+-y2) Z1
    const Fs<1, DC> r1 = mul(select(l0, p.z, cneg(negq, q.y)), p.z);
    const Fs<1, DC> z1z1 = pair_bcast<0>(r1), t = pair_bcast<1>(r1);
    // level 2: lane 0 H = x2 Z1Z1 - X1, lane 1 rr = T Z1Z1 - Y1 (the subtrahend injected into the reduction)
    const Fs<5, DC> r2 = mul_inj<-1, DC>(select(l0, q.x, t), z1z1, select(l0, p.x, relax<4, DC>(p.y)));
    const Fs<5, DC> h = pair_bcast<0>(r2), rr = pair_bcast<1>(r2);  // (rr <= 2 in value; the type carries the lanes' common bound)
    // level 3: lane 0 HH = H^2, lane 1 Z3 = H Z1
    const Fs<1, DC> r3 = mul(h, select(l0, h, relax<5, DC>(p.z)));
    const Fs<1, DC> hh = pair_bcast<0>(r3), z3 = pair_bcast<1>(r3);
    // level 4: lane 0 HHH = H HH, lane 1 V = X1 HH
    const Fs<1, DC> r4 = mul(select(l0, h, relax<5, DC>(p.x)), hh);
    const Fs<1, DC> hhh = pair_bcast<0>(r4), v = pair_bcast<1>(r4);
    // levels 5 and 6 on both lanes
    JacS r;
    r.x = sqr_inj2<-1, -2, DC>(rr, hhh, v);                                          // <= 4
    r.y = mul_add<DC>(rr, sub_lazy(v, r.x), neg(p.y), hhh);                          // rr (V - X3) - Y1 HHH
    r.z = z3;
    if (__builtin_expect(product_is_zero(z3), 0)) r = add_mixed_slow(p, q, negq);    // identity accumulator, P + P, P - P: both lanes
    return r;
}

}  // namespace kzg
