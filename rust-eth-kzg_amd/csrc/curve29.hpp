// G1 group law over the unsaturated field (fp29.hpp) for the hot kernels.
// Same formulas as curve.hpp (dbl-2009-l, add-2007-bl, madd-2007-bl) rearranged so that
//   * Z3 is always 2 * (a product): bound 4, so "Z == 0 mod p" is four 14-word compares (kept branch-free:
//     an early-out on the low limb looked cheaper but its extra control flow made hipcc spill 4x more VGPRs);
//   * NO exceptional-case test sits on the hot path: every degenerate addition (an identity operand,
//     P + P, P + (-P)) makes Z3 = 2 Z1 (Z2) H vanish, so one cheap test of Z3 after the fact routes those
//     rare cases to an exact slow path.  The constant / all-zero / two-valued fixture blobs reach it.
// Stored coordinates: x, y < 64 p, z < 4 p, limbs normalised.
#pragma once
#include "curve.hpp"
#include "fp29.hpp"

namespace kzg {

constexpr int XB = 64;  // bound of stored x / y coordinates (in units of p)
constexpr int ZB = 4;   // bound of stored z

struct AffQ {  // canonical coordinates; identity = (0, 0)
    Fq<1> x, y;
};
// A GLV window-table entry: both coordinates canonical (Montgomery-406 values < p), packed 12 x 32 bits each
// (k_table.hip: k_table_fill_packed).  Identity = all zero.
struct alignas(16) TabP {
    uint32_t w[24];
};
static_assert(sizeof(TabP) == 96, "packed table entries");
struct JacQ {  // identity <=> z == 0 mod p
    Fq<XB> x, y;
    Fq<ZB> z;
};

HD bool is_inf(const AffQ& p) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) d |= p.x.v[i] | p.y.v[i];
    return d == 0;
}
HD bool is_inf(const JacQ& p) { return is_zero(p.z); }
// "t == 0 mod p" for a fresh product t < 2p: only 0 and p qualify, so one look at the low limb settles almost every
// case and the 28-word comparison runs only when that limb matches (hot paths: once per addition)
HD bool product_is_zero(const Fq<2>& t) {
    const uint32_t l0 = t.v[0];
    if (l0 != 0 && l0 != q29::P[0]) return false;
    return is_zero(t);
}
HD bool affine_is_inf(const AffQ& q) { return (q.x.v[0] | q.y.v[0]) == 0 && is_inf(q); }
HD JacQ jacq_inf() {
    JacQ r;
    r.x = relax<XB>(fq_one());
    r.y = relax<XB>(fq_one());
    r.z = relax<ZB>(fq_zero());
    return r;
}
HD JacQ to_jacq(const AffQ& a) {
    if (is_inf(a)) return jacq_inf();
    JacQ r;
    r.x = relax<XB>(a.x);
    r.y = relax<XB>(a.y);
    r.z = relax<ZB>(fq_one());
    return r;
}
// Negating a stored coordinate would double its bound, so the hot paths never negate a point: add / add_mixed take
// a `negq` flag and negate the product S2 instead (bound 2 -> 4, no reduction).  neg() itself goes through one
// multiplication by R to return to a canonical value; it is used off the hot path only.
HD JacQ neg(const JacQ& p) {
    JacQ r = p;
    r.y = relax<XB>(canonical(neg(p.y)));
    return r;
}
template <int B>
HD Fq<B> select(bool c, const Fq<B>& a, const Fq<B>& b) {
    Fq<B> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// dbl-2009-l with D = 4 X Y^2 written as a product: 3M + 4S
HD JacQ dbl(const JacQ& p) {
    Fq<2> A = sqr(p.x), B = sqr(p.y);
    Fq<8> D = dbl2(mul(p.x, B));
    Fq<6> E = add(dbl(A), A);
    Fq<2> F = sqr(E);
    JacQ r;
    auto x3 = sub2(F, D);                                       // F - 2D < 2 + 32
    auto y3 = mul_add(E, sub(D, x3), neg2(B), dbl2(B));         // E(D - x3) - 8 B^2, one reduction: < 2
    r.x = relax<XB>(x3);
    r.y = relax<XB>(y3);
    r.z = dbl(mul(p.y, p.z));                                   // identity stays identity: z = 0 -> z3 = 0
    return r;
}

// exact slow paths ------------------------------------------------------------------------------------------
HD JacQ add_slow(const JacQ& p, const JacQ& q, bool negq);
HD JacQ add_mixed_slow(const JacQ& p, const AffQ& q, bool negq);

// add-2007-bl: 12M + 4S, no branches on the hot path except the final Z3 test
// (p + q, or p - q when negq)
HD JacQ add(const JacQ& p, const JacQ& q, bool negq = false) {
    Fq<2> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fq<2> u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    Fq<2> s1 = mul(mul(p.y, q.z), z2z2), s2p = mul(mul(q.y, p.z), z1z1);
    auto h = sub(u2, u1);                         // < 6
    auto rr = dbl(signed_sub(negq, s2p, s1));     // 2(+-S2 - S1) < 16
    Fq<2> i = sqr(dbl(h));
    Fq<2> j = mul(h, i);
    Fq<2> v = mul(u1, i);
    JacQ r;
    auto x3 = sub_sub2(sqr(rr), j, v);                            // rr^2 - J - 2V < 2 + 4 + 8
    auto y3 = mul_add(rr, sub(v, x3), neg2(s1), j);               // rr(V - X3) - 2 S1 J, one reduction: < 2
    r.x = relax<XB>(x3);
    r.y = relax<XB>(y3);
    const Fq<2> zh = mul(mul(p.z, q.z), h);  // Z3 / 2: zero test on the product (two candidates) instead of on its double (four)
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_slow(p, q, negq);
    return r;
}

// p + q AND p - q: everything of add-2007-bl except rr, X3 and Y3 is shared (the two results have the same Z3), so the
// second result costs one squaring and one fused product pair on top of the first's 12M + 4S.  In two steps so that a kernel
// can store one result before it computes the other (both at once would not fit 256 registers).  Degenerate operands (an
// identity, p = +-q) make the shared Z3 vanish: `degenerate` then sends BOTH results to the exact slow path.
struct AddSubShared {
    Fq<2> s1, s2p, j, v, zh;
    bool degenerate;
};
HD AddSubShared add_sub_prepare(const JacQ& p, const JacQ& q) {
    AddSubShared sh;
    Fq<2> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fq<2> u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    sh.s1 = mul(mul(p.y, q.z), z2z2);
    sh.s2p = mul(mul(q.y, p.z), z1z1);
    auto h = sub(u2, u1);
    Fq<2> i = sqr(dbl(h));
    sh.j = mul(h, i);
    sh.v = mul(u1, i);
    sh.zh = mul(mul(p.z, q.z), h);
    sh.degenerate = product_is_zero(sh.zh);
    return sh;
}
HD JacQ add_sub_finish(const AddSubShared& sh, bool negq) {  // p + q, or p - q when negq (not for degenerate operands)
    auto rr = dbl(signed_sub(negq, sh.s2p, sh.s1));
    auto x3 = sub_sub2(sqr(rr), sh.j, sh.v);
    JacQ r;
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(sh.v, x3), neg2(sh.s1), sh.j));
    r.z = dbl(sh.zh);
    return r;
}

// madd-2007-bl with Z3 = 2 Z1 H: 8M + 3S
HD JacQ add_mixed(const JacQ& p, const AffQ& q, bool negq = false) {
    if (affine_is_inf(q)) return p;  // an affine identity (0,0) does not make Z3 vanish: test it up front
    Fq<2> z1z1 = sqr(p.z);
    Fq<2> u2 = mul(q.x, z1z1);
    Fq<2> s2p = mul(mul(q.y, p.z), z1z1);
    auto h = sub(u2, p.x);                        // < 2 + 128
    auto rr = dbl(signed_sub(negq, s2p, p.y));    // 2(+-S2 - Y1) < 264
    Fq<2> hh = sqr(h);
    Fq<8> i = dbl2(hh);
    Fq<2> j = mul(h, i);
    Fq<2> v = mul(p.x, i);
    JacQ r;
    auto x3 = sub_sub2(sqr(rr), j, v);                            // rr^2 - J - 2V < 2 + 4 + 8
    auto y3 = mul_add(rr, sub(v, x3), neg2(p.y), j);              // rr(V - X3) - 2 Y1 J, one reduction: < 2
    r.x = relax<XB>(x3);
    r.y = relax<XB>(y3);
    const Fq<2> zh = mul(p.z, h);
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_mixed_slow(p, q, negq);
    return r;
}

HD JacQ add_slow(const JacQ& p, const JacQ& q, bool negq) {
    if (is_inf(p)) return negq ? neg(q) : q;
    if (is_inf(q)) return p;
    // Z3 = 2 Z1 Z2 H == 0 with Z1, Z2 != 0  =>  H == 0: same x.  Same y -> doubling, opposite y -> identity.
    Fq<2> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fq<2> s1 = mul(mul(p.y, q.z), z2z2), s2p = mul(mul(q.y, p.z), z1z1);
    Fq<4> s2 = select(negq, neg(s2p), relax<4>(s2p));
    if (is_zero_slow(sub(s2, s1))) return dbl(p);
    return jacq_inf();
}
HD JacQ add_mixed_slow(const JacQ& p, const AffQ& q, bool negq) {
    if (is_inf(q)) return p;
    if (is_inf(p)) {
        JacQ r = to_jacq(q);
        return negq ? neg(r) : r;
    }
    Fq<2> z1z1 = sqr(p.z);
    Fq<2> s2p = mul(mul(q.y, p.z), z1z1);
    Fq<4> s2 = select(negq, neg(s2p), relax<4>(s2p));
    if (is_zero_slow(sub(s2, p.y))) return dbl(p);
    return jacq_inf();
}

// Mixed addition with a table point whose coordinates are fresh products (< 2p, not canonical) and which is never the
// identity: the twiddle multiplication's table after it has been brought to a common Z (k_g1fft.hip).
struct AffQ2 {
    Fq<2> x, y;
};
HD JacQ add_mixed_slow(const JacQ& p, const AffQ2& q, bool negq);
HD JacQ add_mixed(const JacQ& p, const AffQ2& q, bool negq) {
    Fq<2> z1z1 = sqr(p.z);
    Fq<2> u2 = mul(q.x, z1z1);
    Fq<2> s2p = mul(mul(q.y, p.z), z1z1);
    auto h = sub(u2, p.x);
    auto rr = dbl(signed_sub(negq, s2p, p.y));
    Fq<2> hh = sqr(h);
    Fq<8> i = dbl2(hh);
    Fq<2> j = mul(h, i);
    Fq<2> v = mul(p.x, i);
    JacQ r;
    auto x3 = sub_sub2(sqr(rr), j, v);
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(v, x3), neg2(p.y), j));
    const Fq<2> zh = mul(p.z, h);
    r.z = dbl(zh);
    if (product_is_zero(zh)) return add_mixed_slow(p, q, negq);
    return r;
}
HD JacQ add_mixed_slow(const JacQ& p, const AffQ2& q, bool negq) {
    JacQ qj;
    qj.x = relax<XB>(q.x);
    qj.y = relax<XB>(q.y);
    qj.z = relax<ZB>(fq_one());
    return add_slow(p, qj, negq);  // identity accumulator, P + P or P - P: the general slow path decides
}

// XYZZ accumulator for long runs of mixed additions (the fixed-base MSM): (X, Y, ZZ, ZZZ) stands for the affine point
// (X / ZZ, Y / ZZZ), ZZ^3 = ZZZ^2; identity <=> ZZ == 0.  madd-2008-s costs 6M + 2S + one fused product pair against
// 6M + 3S + one fused pair for the Jacobian form above, and needs none of its doublings of intermediate values.
struct XyzzQ {
    Fq<XB> x, y;
    Fq<2> zz, zzz;
};
HD XyzzQ xyzz_inf() {
    XyzzQ r;
    r.x = relax<XB>(fq_one());
    r.y = relax<XB>(fq_one());
    r.zz = relax<2>(fq_zero());
    r.zzz = relax<2>(fq_zero());
    return r;
}
HD bool is_inf(const XyzzQ& p) { return is_zero(p.zz); }
HD JacQ to_jacq(const XyzzQ& p) {  // (X ZZ, Y ZZZ, ZZ): X ZZ / ZZ^2 = X / ZZ and Y ZZZ / ZZ^3 = Y / ZZZ
    JacQ r;
    r.x = relax<XB>(mul(p.x, p.zz));
    r.y = relax<XB>(mul(p.y, p.zzz));
    r.z = relax<ZB>(p.zz);
    return r;
}
HD XyzzQ add_mixed_slow(const XyzzQ& p, const AffQ& q, bool negq);
HD XyzzQ add_mixed(const XyzzQ& p, const AffQ& q, bool negq = false) {
    if (affine_is_inf(q)) return p;
    Fq<2> u2 = mul(q.x, p.zz), s2 = mul(q.y, p.zzz);
    auto pp_ = sub(u2, p.x);                      // P  < 2 + 128
    auto rr = signed_sub(negq, s2, p.y);          // R = +-S2 - Y1 < 4 + 128
    Fq<2> pp = sqr(pp_);
    Fq<2> ppp = mul(pp_, pp);
    Fq<2> qq = mul(p.x, pp);
    auto x3 = sub_sub2(sqr(rr), ppp, qq);         // R^2 - PPP - 2Q < 2 + 4 + 8
    XyzzQ r;
    r.x = relax<XB>(x3);
    r.y = relax<XB>(mul_add(rr, sub(qq, x3), neg(p.y), ppp));  // R(Q - X3) - Y1 PPP, one reduction
    r.zz = mul(p.zz, pp);
    r.zzz = mul(p.zzz, ppp);
    if (product_is_zero(r.zz)) return add_mixed_slow(p, q, negq);  // identity accumulator, or equal x: P + P / P - P
    return r;
}
HD XyzzQ add_mixed_slow(const XyzzQ& p, const AffQ& q, bool negq) {
    if (is_inf(q)) return p;
    JacQ j;
    if (is_inf(p)) {
        j = to_jacq(q);
        if (negq) j = neg(j);
    } else {
        Fq<2> s2p = mul(q.y, p.zzz);
        Fq<4> s2 = select(negq, neg(s2p), relax<4>(s2p));
        if (!is_zero_slow(sub(s2, p.y))) return xyzz_inf();  // opposite points
        j = dbl(to_jacq(q));                                 // equal points: 2 q
        if (negq) j = neg(j);
    }
    XyzzQ r;  // Jacobian (X, Y, Z) -> (X, Y, Z^2, Z^3)
    r.x = j.x;
    r.y = j.y;
    r.zz = sqr(j.z);
    r.zzz = mul(r.zz, j.z);
    return r;
}

HD Fq<1> reduce_once(const Fq<2>& t) {  // t < 2p -> canonical
    uint32_t d[QL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) {
        uint32_t x = t.v[i] - q29::P[i] - borrow;
        borrow = x >> 31;
        d[i] = x & QMASK;
    }
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = borrow ? t.v[i] : d[i];
    return r;
}
template <int B>
HD Fq<1> fq_inv(const Fq<B>& z) {  // through the saturated form, where the binary-GCD inversion lives
    return fq_from_fp(inv_fast(fp_from_fq(z)));
}

// conversions ------------------------------------------------------------------------------------------------
HD AffQ affq_from_affine(const G1Affine& a) {  // a canonical Montgomery-384; identity (0,0) maps to (0,0)
    AffQ r;
    r.x = fq_from_fp(a.x);
    r.y = fq_from_fp(a.y);
    return r;
}
HD JacQ jacq_from_jac(const G1Jac& p) {
    if (is_inf(p)) return jacq_inf();
    JacQ r;
    r.x = relax<XB>(fq_from_fp(p.x));
    r.y = relax<XB>(fq_from_fp(p.y));
    r.z = relax<ZB>(fq_from_fp(p.z));
    return r;
}
HD G1Jac jac_from_jacq(const JacQ& p) {
    if (is_inf(p)) return jac_inf();
    G1Jac r;
    r.x = fp_from_fq(p.x);
    r.y = fp_from_fq(p.y);
    r.z = fp_from_fq(p.z);
    return r;
}

}  // namespace kzg
