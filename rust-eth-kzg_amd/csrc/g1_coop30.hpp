// Two (and, below, four) lanes per point operation in the signed 13 x 30-bit field (fp30.hpp / curve30.hpp): the chain forms of the constant
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

// ---- four lanes per point operation (<= 16 blobs in a 64-lane wave: lanes 4b .. 4b + 3 hold blob b; the folds of the one-block MSM) ----
// Round 6: the quad forms of g1_coop.hpp in the signed 13 x 30-bit field, so that the prover's path from the
// scalars to the proof bytes is in ONE Fp representation at every batch size above the circulant form's.  Four lanes make room for
// the square of rr (and of both rr of a sum-and-difference pair) next to the level that computes H^2, so X3 = rr^2 - HHH - 2 V is
// three additive steps instead of a fused squaring on every lane:
//     doubling      X^2 | Y^2 | Y Z  ->  X Y^2 | H^2  ->  H (M - X3) - B^2                                   3 reductions deep
//     mixed add     Z1^2 | y2 Z1  ->  H | rr (subtrahends injected)  ->  H^2 | H Z1 | rr^2  ->  H^3 | X1 H^2  ->  Y3     5 deep
//     addition      Z1^2 | Z2^2 | Y1 Z2 | Y2 Z1  ->  U1 | U2 | S1 | S2  ->  H^2 | Z1 Z2 | rr^2 (| rr'^2)  ->  H^3 | U1 H^2 | Z1 Z2 H  ->  Y3
// every lane of a quad (four consecutive lanes) receives lane J's value
template <int J, int B, int F>
__device__ __forceinline__ Fs<B, F> quad_bcast(const Fs<B, F>& a) {
    static_assert(J >= 0 && J < 4, "lane of the quad");
    Fs<B, F> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], J * 0x55, 0xf, 0xf, true);
    return r;
}
template <int B, int F>
__device__ __forceinline__ Fs<B, F> select4(int quad, const Fs<B, F>& a, const Fs<B, F>& b, const Fs<B, F>& c, const Fs<B, F>& d) {
    return select(quad == 0, a, select(quad == 1, b, select(quad == 2, c, d)));
}

// curve30.hpp: dbl_half by a quad; quad = lane & 3.  ALL FOUR lanes must be active and hold the same p.
__device__ __forceinline__ JacS coop4_dbl_half(const JacS& p, int quad) {
    const bool l0 = quad == 0, l1 = quad == 1;
    const Fs<4, DC> y4 = relax<4, DC>(p.y), z4 = relax<4, DC>(p.z);
    // level 1: lane 0 A = X^2, lane 1 B = Y^2, lanes 2 and 3 Z3 = Y Z
    const Fs<1, DC> r1 = mul(select(l0, p.x, y4), select(l0, p.x, select(l1, y4, z4)));
    const Fs<1, DC> A = quad_bcast<0>(r1), B = quad_bcast<1>(r1), Z3 = quad_bcast<2>(r1);
    const Fs<2, DC> H = half_of_triple(A);
    // level 2: lane 0 M = B X, the others H^2
    const Fs<1, DC> r2 = mul(select(l0, relax<4, DC>(B), relax<4, DC>(H)), select(l0, p.x, relax<4, DC>(H)));
    const Fs<1, DC> M = quad_bcast<0>(r2), HH = quad_bcast<1>(r2);
    // level 3 on every lane
    JacS r;
    r.x = relax<4, DC>(sub(HH, mul_small<2>(M)));              // H^2 - 2 M: <= 3
    r.y = mul_add<DC>(H, sub_lazy(M, r.x), neg(B), B);         // H (M - X3) - B^2, one reduction
    r.z = Z3;
    return r;
}

// the same with the product beta X -- the x of phi(p) = (beta X : Y : Z) -- in the lane that the first level leaves idle (the
// doubling table of the circulant form, k_g1circ.hip, stores p and phi(p) before every doubling)
__device__ __forceinline__ JacS coop4_dbl_half_phi(const JacS& p, int quad, const Fs<1, DC>& beta, Fs<1, DC>& beta_x) {
    const bool l0 = quad == 0, l1 = quad == 1, l3 = quad == 3;
    const Fs<4, DC> y4 = relax<4, DC>(p.y), z4 = relax<4, DC>(p.z);
    const Fs<1, DC> r1 = mul(select(l0 || l3, p.x, y4), select(l0, p.x, select(l1, y4, select(l3, relax<4, DC>(beta), z4))));
    const Fs<1, DC> A = quad_bcast<0>(r1), B = quad_bcast<1>(r1), Z3 = quad_bcast<2>(r1);
    beta_x = quad_bcast<3>(r1);
    const Fs<2, DC> H = half_of_triple(A);
    const Fs<1, DC> r2 = mul(select(l0, relax<4, DC>(B), relax<4, DC>(H)), select(l0, p.x, relax<4, DC>(H)));
    const Fs<1, DC> M = quad_bcast<0>(r2), HH = quad_bcast<1>(r2);
    JacS r;
    r.x = relax<4, DC>(sub(HH, mul_small<2>(M)));
    r.y = mul_add<DC>(H, sub_lazy(M, r.x), neg(B), B);
    r.z = Z3;
    return r;
}

// curve30.hpp: add_mixed(JacS, AffT) by a quad: p + q, or p - q when negq (wave-uniform)
__device__ __forceinline__ JacS coop4_add_mixed(const JacS& p, const AffT& q, bool negq, int quad) {
    const bool l0 = quad == 0, l1 = quad == 1;
    // level 1: lane 0 Z1^2, the others T = (+-y2) Z1
    const Fs<1, DC> r1 = mul(select(l0, p.z, cneg(negq, q.y)), p.z);
    const Fs<1, DC> z1z1 = quad_bcast<0>(r1), t = quad_bcast<1>(r1);
    // level 2: lane 0 H = x2 Z1Z1 - X1, the others rr = T Z1Z1 - Y1 (the subtrahend injected into the reduction)
    const Fs<5, DC> r2 = mul_inj<-1, DC>(select(l0, q.x, t), z1z1, select(l0, p.x, relax<4, DC>(p.y)));
    const Fs<5, DC> h = quad_bcast<0>(r2), rr = quad_bcast<1>(r2);  // (rr <= 2 in value; the type carries the lanes' common bound)
    // level 3: lane 0 HH = H^2, lane 1 Z3 = H Z1, lanes 2 and 3 rr^2
    const Fs<1, DC> r3 = mul(select(l0 || l1, h, rr), select(l0, h, select(l1, relax<5, DC>(p.z), rr)));
    const Fs<1, DC> hh = quad_bcast<0>(r3), z3 = quad_bcast<1>(r3), rr2 = quad_bcast<2>(r3);
    // level 4: lane 0 HHH = H HH, the others V = X1 HH
    const Fs<1, DC> r4 = mul(select(l0, h, relax<5, DC>(p.x)), hh);
    const Fs<1, DC> hhh = quad_bcast<0>(r4), v = quad_bcast<1>(r4);
    // level 5 on every lane
    JacS r;
    r.x = sub(sub(rr2, hhh), mul_small<2>(v));                                       // rr^2 - HHH - 2 V: <= 4
    r.y = mul_add<DC>(rr, sub_lazy(v, r.x), neg(p.y), hhh);                          // rr (V - X3) - Y1 HHH
    r.z = z3;
    if (__builtin_expect(product_is_zero(z3), 0)) r = add_mixed_slow(p, q, negq);    // identity accumulator, P + P, P - P: all four lanes
    return r;
}

// curve30.hpp: add by a quad: p + q, or p - q when negq; degenerate operands leave by add_slow on all four lanes
__device__ __forceinline__ JacS coop4_add(const JacS& p, const JacS& q, bool negq, int quad) {
    // level 1: Z1^2 | Z2^2 | Y1 Z2 | (+-Y2) Z1
    const Fs<1, DC> r1 = mul(select4(quad, p.z, q.z, p.y, cneg(negq, q.y)), select4(quad, p.z, q.z, q.z, p.z));
    const Fs<1, DC> z1z1 = quad_bcast<0>(r1), z2z2 = quad_bcast<1>(r1), a = quad_bcast<2>(r1), t = quad_bcast<3>(r1);
    // level 2: U1 = X1 Z2Z2 | U2 = X2 Z1Z1 | S1 = Y1 Z2 Z2Z2 | S2 = +-Y2 Z1 Z1Z1
    const Fs<1, DC> r2 = mul(select4(quad, p.x, q.x, relax<4, DC>(a), relax<4, DC>(t)), select4(quad, z2z2, z1z1, z2z2, z1z1));
    const Fs<1, DC> u1 = quad_bcast<0>(r2), u2 = quad_bcast<1>(r2), s1 = quad_bcast<2>(r2), s2 = quad_bcast<3>(r2);
    const Fs<2, DC> h = sub(u2, u1), rr = sub(s2, s1);
    // level 3: HH = H^2 | Z1 Z2 | rr^2 (lanes 2 and 3)
    const Fs<1, DC> r3 = mul(select4(quad, h, relax<2, DC>(p.z), rr, rr), select4(quad, h, relax<2, DC>(q.z), rr, rr));
    const Fs<1, DC> hh = quad_bcast<0>(r3), z1z2 = quad_bcast<1>(r3), rr2 = quad_bcast<2>(r3);
    // level 4: HHH = H HH | V = U1 HH | Z3 = Z1 Z2 H (lanes 2 and 3)
    const Fs<2, DC> u1w = relax<2, DC>(u1);
    const Fs<1, DC> r4 = mul(select4(quad, h, u1w, h, h), select4(quad, hh, hh, z1z2, z1z2));
    const Fs<1, DC> hhh = quad_bcast<0>(r4), v = quad_bcast<1>(r4), z3 = quad_bcast<2>(r4);
    // level 5 on every lane
    JacS r;
    r.x = sub(sub(rr2, hhh), mul_small<2>(v));                                       // rr^2 - HHH - 2 V: <= 4
    r.y = mul_add<DC>(rr, sub_lazy(v, r.x), neg(s1), hhh);                           // rr (V - X3) - S1 HHH
    r.z = z3;
    if (__builtin_expect(product_is_zero(z3), 0)) r = add_slow(p, q, negq);
    return r;
}
// p + q AND p - q (the sum-and-difference pairs of the G1 linear map): the fourth lane of level 3 squares the other rr, everything else
// is shared -- four levels and two fused pairs instead of eight levels and two pairs
__device__ __forceinline__ void coop4_add_sub(const JacS& p, const JacS& q, int quad, JacS& sum, JacS& diff) {
    const Fs<1, DC> r1 = mul(select4(quad, p.z, q.z, p.y, q.y), select4(quad, p.z, q.z, q.z, p.z));
    const Fs<1, DC> z1z1 = quad_bcast<0>(r1), z2z2 = quad_bcast<1>(r1), a = quad_bcast<2>(r1), t = quad_bcast<3>(r1);
    const Fs<1, DC> r2 = mul(select4(quad, p.x, q.x, relax<4, DC>(a), relax<4, DC>(t)), select4(quad, z2z2, z1z1, z2z2, z1z1));
    const Fs<1, DC> u1 = quad_bcast<0>(r2), u2 = quad_bcast<1>(r2), s1 = quad_bcast<2>(r2), s2 = quad_bcast<3>(r2);
    const Fs<2, DC> h = sub(u2, u1), rp = sub(s2, s1), rm = neg(add(s2, s1));     // S2 - S1 and -S2 - S1
    const Fs<1, DC> r3 = mul(select4(quad, h, relax<2, DC>(p.z), rp, rm), select4(quad, h, relax<2, DC>(q.z), rp, rm));
    const Fs<1, DC> hh = quad_bcast<0>(r3), z1z2 = quad_bcast<1>(r3), rp2 = quad_bcast<2>(r3), rm2 = quad_bcast<3>(r3);
    const Fs<2, DC> u1w = relax<2, DC>(u1);
    const Fs<1, DC> r4 = mul(select4(quad, h, u1w, h, h), select4(quad, hh, hh, z1z2, z1z2));
    const Fs<1, DC> hhh = quad_bcast<0>(r4), v = quad_bcast<1>(r4), z3 = quad_bcast<2>(r4);
    if (__builtin_expect(product_is_zero(z3), 0)) {  // an identity operand, p = +-q: the exact forms, all four lanes
        const JacS s_ = add_slow(p, q, false), d_ = add_slow(p, q, true);
        sum = s_;
        diff = d_;
        return;
    }
    const Fs<2, DC> v2 = mul_small<2>(v);
    const Fs<1, DC> ns1 = neg(s1);
    JacS d;
    d.x = sub(sub(rm2, hhh), v2);
    d.y = mul_add<DC>(rm, sub_lazy(v, d.x), ns1, hhh);
    d.z = z3;
    JacS s_;
    s_.x = sub(sub(rp2, hhh), v2);
    s_.y = mul_add<DC>(rp, sub_lazy(v, s_.x), ns1, hhh);
    s_.z = z3;
    sum = s_;
    diff = d;
}

// g1_coop.hpp: coop_tree_fold on points of the signed field: red[0 .. 2 * first_span) hold the lanes' partial sums, red[0] their
// total on return; four lanes per addition, NT / 4 additions per round
template <int NT>
__device__ __forceinline__ void coop4_tree_fold(JacS* red, int first_span, int tid) {
    const int quad = tid & 3, slot = tid >> 2;
    __syncthreads();
#pragma unroll 1
    for (int span = first_span; span >= 1; span >>= 1) {
#pragma unroll 1
        for (int base = 0; base < span; base += NT / 4) {  // a round reads red[a] and red[a + span] and writes red[a]: its own slots only
            const int a = base + slot;
            // A wave takes part as a whole or not at all: a wave with few lanes in use is the slow one (k_g1slp.hip: k_slp_mulc_s), and
            // the last levels have 16, 8 and 4.  The quads behind the last addition repeat it and store nothing; its owner is in their
            // wave, so they read its operands before it writes.
            if (base + ((tid & ~63) >> 2) < span) {
                const int aa = a < span ? a : span - 1;
                const JacS r = coop4_add(red[aa], red[aa + span], false, quad);
                if (quad == 0 && a < span) red[a] = r;
            }
        }
        __syncthreads();
    }
}

}  // namespace kzg
