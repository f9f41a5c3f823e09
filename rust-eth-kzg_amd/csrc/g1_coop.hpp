// Point doublings by the four lanes of a quad.  A chain of dependent doublings (the subgroup test's 126, the 120 behind the
// byte-shifted copies of a verification) is latency, not work, when the batch is a few hundred points: the waves have a SIMD
// each and the chip is idle.  The seven products of a doubling have a critical path of three,
//     X^2 | Y^2 | Y Z   ->   X (Y^2) | (3 X^2)^2   ->   E (D - X3) - 8 Y^4,
// so four lanes that all hold the point compute one level's products side by side (operands picked by lane, one multiplication
// issued), exchange them with quad-broadcast DPP moves (14 per field element) and repeat the cheap linear steps redundantly:
// 3.5 multiplication times per doubling instead of 6.5.  Every lane of the quad ends with the whole result, so the additions
// between doublings (rare in both chains) simply run four times over.  Same formulas and bounds as curve29.hpp: dbl().
#pragma once
#include "curve29.hpp"

namespace kzg {

template <class T> struct fq_bound;
template <int B> struct fq_bound<Fq<B>> { static constexpr int value = B; };

// every lane of a quad (four consecutive lanes) receives lane J's value
template <int J, int B>
__device__ __forceinline__ Fq<B> quad_bcast(const Fq<B>& a) {
    static_assert(J >= 0 && J < 4, "lane of the quad");
    Fq<B> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], J * 0x55, 0xf, 0xf, true);
    return r;
}
// q = lane & 3; ALL FOUR lanes of the quad must be active and hold the same p
__device__ __forceinline__ JacQ coop_dbl(const JacQ& p, int q) {
    const bool l0 = q == 0, l1 = q == 1;
    // level 1: lane 0 X^2, lane 1 Y^2, lanes 2 and 3 Y Z
    const Fq<XB> a1 = select(l0, p.x, p.y);
    const Fq<XB> b1 = select(l0, p.x, select(l1, p.y, relax<XB>(p.z)));
    const Fq<2> r1 = mul(a1, b1);
    const Fq<2> A = quad_bcast<0>(r1), B = quad_bcast<1>(r1), YZ = quad_bcast<2>(r1);
    // level 2: lane 0 X B, the others E^2
    const Fq<6> E = add(dbl(A), A);
    const Fq<XB> a2 = select(l0, p.x, relax<XB>(E));
    const Fq<XB> b2 = select(l0, relax<XB>(B), relax<XB>(E));
    const Fq<2> r2 = mul(a2, b2);
    const Fq<2> XY2 = quad_bcast<0>(r2), F = quad_bcast<1>(r2);
    const Fq<8> D = dbl2(XY2);
    // level 3, on every lane: one fused product pair
    JacQ r;
    auto x3 = sub2(F, D);
    auto y3 = mul_add(E, sub(D, x3), neg2(B), dbl2(B));
    r.x = relax<XB>(x3);
    r.y = relax<XB>(y3);
    r.z = dbl(YZ);
    return r;
}

// Mixed addition (curve29.hpp: add_mixed, madd-2007-bl with Z3 = 2 Z1 H) by the four lanes of a quad: its ten products in five
// levels --  Z1^2 | y2 Z1  ->  x2 Z1Z1 | (y2 Z1) Z1Z1  ->  H^2 | Z1 H | rr^2  ->  H I | X1 I  ->  the fused pair of Y3  -- instead of
// 10.5 multiplication times.  Aff: AffQ (canonical; the identity (0,0) is tested up front like add_mixed does) or AffQ2 (table
// points that are fresh products and never the identity).  Degenerate operands leave by the same exact slow path, on all four lanes.
__device__ __forceinline__ bool coop_affine_is_inf(const AffQ& q) { return affine_is_inf(q); }
__device__ __forceinline__ bool coop_affine_is_inf(const AffQ2&) { return false; }
template <class Aff>
__device__ __forceinline__ JacQ coop_add_mixed(const JacQ& p, const Aff& q, bool negq, int quad) {
    if (coop_affine_is_inf(q)) return p;
    const bool l0 = quad == 0, l1 = quad == 1;
    const Fq<2> qx = relax<2>(q.x), qy = relax<2>(q.y);
    // level 1: lane 0 Z1^2, the others y2 Z1
    const Fq<ZB> a1 = select(l0, p.z, relax<ZB>(qy));
    const Fq<2> r1 = mul(a1, p.z);
    const Fq<2> z1z1 = quad_bcast<0>(r1), t = quad_bcast<1>(r1);
    // level 2: lane 0 U2 = x2 Z1Z1, the others S2 = (y2 Z1) Z1Z1
    const Fq<2> r2 = mul(select(l0, qx, t), z1z1);
    const Fq<2> u2 = quad_bcast<0>(r2), s2p = quad_bcast<1>(r2);
    auto h = sub(u2, p.x);
    auto rr = dbl(signed_sub(negq, s2p, p.y));
    // level 3: lane 0 H^2, lane 1 Z1 H, lanes 2 and 3 rr^2
    constexpr int HB = fq_bound<decltype(h)>::value, RB = fq_bound<decltype(rr)>::value, WB = HB > RB ? HB : RB;
    const Fq<WB> hw = relax<WB>(h), rw = relax<WB>(rr);
    const Fq<WB> a3 = select(l0, hw, select(l1, relax<WB>(p.z), rw));
    const Fq<WB> b3 = select(l0 || l1, hw, rw);
    const Fq<2> r3 = mul(a3, b3);
    const Fq<2> hh = quad_bcast<0>(r3), zh = quad_bcast<1>(r3), rr2 = quad_bcast<2>(r3);
    const Fq<8> i = dbl2(hh);
    // level 4: lane 0 J = H I, the others V = X1 I
    constexpr int XH = HB > XB ? HB : XB;
    const Fq<2> r4 = mul(select(l0, relax<XH>(h), relax<XH>(p.x)), i);
    const Fq<2> j = quad_bcast<0>(r4), v = quad_bcast<1>(r4);
    // level 5, on every lane
    JacQ r;
    auto x3 = sub_sub2(rr2, j, v);
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(v, x3), neg2(p.y), j));
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_mixed_slow(p, q, negq);
    return r;
}

template <int B>
__device__ __forceinline__ Fq<B> select4(int quad, const Fq<B>& a, const Fq<B>& b, const Fq<B>& c, const Fq<B>& d) {
    return select(quad == 0, a, select(quad == 1, b, select(quad == 2, c, d)));
}
// General addition (curve29.hpp: add, add-2007-bl with Z3 = 2 Z1 Z2 H) by the four lanes of a quad.  Its fifteen products have
// more width than the doubling's: four levels of up to four independent ones and the fused pair of Y3,
//     Z1^2 | Z2^2 | Y1 Z2 | Y2 Z1  ->  U1 | U2 | S1 | S2  ->  (2H)^2 | rr^2 | Z1 Z2  ->  H I | U1 I | Z1 Z2 H  ->  Y3,
// 5.5 multiplication times instead of 16.5.  p + q, or p - q when negq; degenerate operands leave by add_slow on all four lanes.
__device__ __forceinline__ JacQ coop_add(const JacQ& p, const JacQ& q, bool negq, int quad) {
    const Fq<XB> pz = relax<XB>(p.z), qz = relax<XB>(q.z);
    const Fq<2> r1 = mul(select4(quad, pz, qz, p.y, q.y), select4(quad, pz, qz, qz, pz));
    const Fq<2> z1z1 = quad_bcast<0>(r1), z2z2 = quad_bcast<1>(r1), y1z2 = quad_bcast<2>(r1), y2z1 = quad_bcast<3>(r1);
    const Fq<2> r2 = mul(select4(quad, p.x, q.x, relax<XB>(y1z2), relax<XB>(y2z1)), select4(quad, z2z2, z1z1, z2z2, z1z1));
    const Fq<2> u1 = quad_bcast<0>(r2), u2 = quad_bcast<1>(r2), s1 = quad_bcast<2>(r2), s2p = quad_bcast<3>(r2);
    auto h = sub(u2, u1);
    auto rr = dbl(signed_sub(negq, s2p, s1));
    auto h2 = dbl(h);
    constexpr int HB = fq_bound<decltype(h2)>::value, RB = fq_bound<decltype(rr)>::value, WB = HB > RB ? HB : RB;
    const Fq<WB> h2w = relax<WB>(h2), rrw = relax<WB>(rr), pzw = relax<WB>(p.z), qzw = relax<WB>(q.z);
    const Fq<2> r3 = mul(select4(quad, h2w, rrw, pzw, pzw), select4(quad, h2w, rrw, qzw, qzw));
    const Fq<2> i = quad_bcast<0>(r3), rr2 = quad_bcast<1>(r3), z1z2 = quad_bcast<2>(r3);
    constexpr int H1 = fq_bound<decltype(h)>::value;
    const Fq<H1> hw = h, u1w = relax<H1>(u1), zzw = relax<H1>(z1z2), iw = relax<H1>(i);
    const Fq<2> r4 = mul(select4(quad, hw, u1w, zzw, zzw), select4(quad, iw, iw, hw, hw));
    const Fq<2> j = quad_bcast<0>(r4), v = quad_bcast<1>(r4), zh = quad_bcast<2>(r4);
    JacQ r;
    auto x3 = sub_sub2(rr2, j, v);
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(v, x3), neg2(s1), j));
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_slow(p, q, negq);
    return r;
}

// The tree that ends every block-wide sum: red[0 .. 2 * first_span) hold the lanes' partial sums, red[0] their total on return.
// A plain tree leaves more lanes idle with every level and still pays a full addition (16.5 multiplication times) per level;
// here the idle lanes join in: four lanes per addition, NT / 4 additions per round (the first level of a full block takes two
// rounds).  Called by all NT threads of the block; no extra waves, so it pays whatever the occupancy.
template <int NT>
__device__ __forceinline__ void coop_tree_fold(JacQ* red, int first_span, int tid) {
    const int quad = tid & 3, slot = tid >> 2;
    __syncthreads();
#pragma unroll 1
    for (int span = first_span; span >= 1; span >>= 1) {
#pragma unroll 1
        for (int base = 0; base < span; base += NT / 4) {  // a round reads red[a] and red[a + span] and writes red[a]: its own slots only
            const int a = base + slot;
            if (a < span) {
                const JacQ r = coop_add(red[a], red[a + span], false, quad);
                if (quad == 0) red[a] = r;
            }
        }
        __syncthreads();
    }
}

// ---- two lanes per point operation (a batch of 17 .. 32 blobs in a 64-lane wave: lanes 2b and 2b + 1 hold blob b) ----
// every lane of a pair receives the value of the pair's lane J
template <int J, int B>
__device__ __forceinline__ Fq<B> pair_bcast(const Fq<B>& a) {
    static_assert(J == 0 || J == 1, "lane of the pair");
    constexpr int CTRL = J == 0 ? 0xA0 : 0xF5;  // quad_perm [0,0,2,2] / [1,1,3,3]
    Fq<B> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xf, 0xf, true);
    return r;
}
// doubling by a pair: X^2 | Y^2  ->  X Y^2 | Y Z  ->  (3 X^2)^2 | (2 Y^2)^2  ->  E (D - X3) on both: 4 multiplication times (6.5 alone)
__device__ __forceinline__ JacQ coop2_dbl(const JacQ& p, int h) {
    const bool l0 = h == 0;
    const Fq<XB> a1 = select(l0, p.x, p.y);
    const Fq<2> r1 = mul(a1, a1);
    const Fq<2> A = pair_bcast<0>(r1), B = pair_bcast<1>(r1);
    const Fq<2> r2 = mul(select(l0, p.x, p.y), select(l0, relax<XB>(B), relax<XB>(p.z)));
    const Fq<2> XY2 = pair_bcast<0>(r2), YZ = pair_bcast<1>(r2);
    const Fq<6> E = add(dbl(A), A);
    const Fq<6> a3 = select(l0, E, relax<6>(dbl(B)));
    const Fq<2> r3 = mul(a3, a3);
    const Fq<2> F = pair_bcast<0>(r3), B4 = pair_bcast<1>(r3);  // E^2, 4 B^2
    const Fq<8> D = dbl2(XY2);
    JacQ r;
    auto x3 = sub2(F, D);
    r.x = relax<XB>(x3);
    r.y = relax<XB>(sub(mul(E, sub(D, x3)), dbl(B4)));  // E (D - X3) - 8 B^2
    r.z = dbl(YZ);
    return r;
}
// mixed addition by a pair: Z1^2 | y2 Z1 -> x2 Z1Z1 | (y2 Z1) Z1Z1 -> H^2 | rr^2 -> H I | X1 I -> fused Y3 | Z1 H: 5.5 (10.5 alone)
template <class Aff>
__device__ __forceinline__ JacQ coop2_add_mixed(const JacQ& p, const Aff& q, bool negq, int h2) {
    if (coop_affine_is_inf(q)) return p;
    const bool l0 = h2 == 0;
    const Fq<2> qx = relax<2>(q.x), qy = relax<2>(q.y);
    const Fq<2> r1 = mul(select(l0, p.z, relax<ZB>(qy)), p.z);
    const Fq<2> z1z1 = pair_bcast<0>(r1), t = pair_bcast<1>(r1);
    const Fq<2> r2 = mul(select(l0, qx, t), z1z1);
    const Fq<2> u2 = pair_bcast<0>(r2), s2p = pair_bcast<1>(r2);
    auto h = sub(u2, p.x);
    auto rr = dbl(signed_sub(negq, s2p, p.y));
    constexpr int HB = fq_bound<decltype(h)>::value, RB = fq_bound<decltype(rr)>::value, WB = HB > RB ? HB : RB;
    const Fq<WB> a3 = select(l0, relax<WB>(h), relax<WB>(rr));
    const Fq<2> r3 = mul(a3, a3);
    const Fq<2> hh = pair_bcast<0>(r3), rr2 = pair_bcast<1>(r3);
    const Fq<8> i = dbl2(hh);
    constexpr int XH = HB > XB ? HB : XB;
    const Fq<2> r4 = mul(select(l0, relax<XH>(h), relax<XH>(p.x)), i);
    const Fq<2> j = pair_bcast<0>(r4), v = pair_bcast<1>(r4);
    auto x3 = sub_sub2(rr2, j, v);
    // lane 0: the fused pair of Y3, rr (V - X3) - 2 Y1 J; lane 1: Z1 H as the same instruction stream with a zero second product
    auto vx = sub(v, x3);
    auto ny = neg2(p.y);
    constexpr int VB = fq_bound<decltype(vx)>::value, NB = fq_bound<decltype(ny)>::value;
    constexpr int AB5 = RB > ZB ? RB : ZB, BB5 = VB > HB ? VB : HB;
    const Fq<AB5> a5 = select(l0, relax<AB5>(rr), relax<AB5>(p.z));
    const Fq<BB5> b5 = select(l0, relax<BB5>(vx), relax<BB5>(h));
    const Fq<NB> c5 = select(l0, ny, relax<NB>(fq_zero()));
    const Fq<2> r5 = mul_add(a5, b5, c5, j);
    const Fq<2> y3 = pair_bcast<0>(r5), zh = pair_bcast<1>(r5);
    JacQ r;
    r.x = relax<XB>(x3);
    r.y = relax<XB>(y3);
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_mixed_slow(p, q, negq);
    return r;
}

// [|z|] P (g1_subgroup.hpp: mul_by_z_abs_q) with the doublings shared by the quad
__device__ __forceinline__ JacQ coop_mul_by_z_abs(const JacQ& p, int q) {
    constexpr uint64_t Z = 0xd201000000010000ULL;
    JacQ acc = p;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        acc = coop_dbl(acc, q);
        if ((Z >> i) & 1) acc = coop_add(acc, p, false, q);
    }
    return acc;
}
// Scott's test (g1_subgroup.hpp: g1_in_subgroup_q), same verdict on all four lanes
__device__ __forceinline__ bool g1_in_subgroup_coop(const AffQ& pa, const Fq<1>& beta, int q) {
    const JacQ p = to_jacq(pa);
    const JacQ z2p = coop_mul_by_z_abs(coop_mul_by_z_abs(p, q), q);
    const JacQ r = add_mixed(z2p, pa, true);
    if (is_inf(r)) return false;
    const Fq<2> zz = sqr(r.z);
    if (!is_zero_slow(sub(r.x, mul(mul(pa.x, beta), zz)))) return false;
    if (!is_zero_slow(sub(r.y, mul(pa.y, mul(zz, r.z))))) return false;
    return true;
}
}  // namespace kzg
