// G1 group law over the signed 13 x 30-bit field (fp30.hpp) for the GLV window MSM and the constant multiplications.
// What changes against curve29.hpp is not the formulas' shape but where reductions and carries happen:
//   * the XYZZ mixed addition (madd-2008-s) is nine reductions and NO separate additive step on a stored value:
//     P = U2 - X1, R = +-S2 - Y1 and X3 = R^2 - PPP - 2 Q come out of the reductions that form U2, S2 and R^2 (the
//     subtrahends' digits are injected into the upper columns), Q - X3 is used un-normalised, Y1 is negated digit-wise;
//   * only what a squaring or a second wide operand needs leaves a reduction with centred digits (P, R, PP, PPP);
//     X3, Y3, ZZ3, ZZZ3, Q leave as floor digits ("shift and mask" per column);
//   * values are signed, so a fresh product is zero mod p only if all its digits are zero: the exceptional-case test of an
//     addition (ZZ3 == 0: identity accumulator, P + P, P - P) is thirteen ORs.
// Table entries are canonical coordinates in Montgomery-390 form with EXACT centred digits, packed into the same 96 bytes as
// the 14 x 29-bit form's entries (TabS below).  Results go back to the 14 x 29-bit form (JacQ) at the kernels' boundary.
// Checked on the CPU against the saturated group law (tests/c/test_curve30.cpp).
#pragma once
#include "curve29.hpp"
#include "fp30.hpp"

namespace kzg {

// ---- the two unsaturated forms: 14 x 29 unsigned (Montgomery-406) <-> 13 x 30 signed (Montgomery-390) ------------------
HD void regroup_29_to_30(int32_t* out, const uint32_t* in) {  // normalised limbs, value < 2^390
#pragma unroll
    for (int i = 0; i < SL; i++) {
        const int bit = 30 * i, lo = bit / 29, sh = bit - 29 * lo;
        uint64_t acc = (uint64_t)in[lo] >> sh;
        int have = 29 - sh;
        if (lo + 1 < QL) { acc |= (uint64_t)in[lo + 1] << have; have += 29; }
        if (have < 30 && lo + 2 < QL) acc |= (uint64_t)in[lo + 2] << have;
        out[i] = (int32_t)((uint32_t)acc & (uint32_t)SMASK);
    }
}
HD void regroup_30_to_29(uint32_t* out, const int32_t* in) {  // non-negative floor digits, value < 2^386
#pragma unroll
    for (int i = 0; i < QL; i++) {
        const int bit = 29 * i, lo = bit / 30, sh = bit - 30 * lo;
        uint64_t acc = (uint64_t)(uint32_t)in[lo] >> sh;
        const int have = 30 - sh;
        if (lo + 1 < SL) acc |= (uint64_t)(uint32_t)in[lo + 1] << have;
        out[i] = i + 1 < QL ? ((uint32_t)acc & QMASK) : (uint32_t)acc;
    }
}
template <int OUTF = DC, int B>
HD Fs<1, OUTF> fs_from_fq(const Fq<B>& a) {  // same field element: x 2^406 -> x 2^390
    static_assert(B <= 64, "fs_from_fq: value below 64 p");
    Fs<B, DU> t;
    regroup_29_to_30(t.v, a.v);
    Fs<1, DC> c;
#pragma unroll
    for (int i = 0; i < SL; i++) c.v[i] = q30::C_FROM_29[i];
    return mul<OUTF>(c, t);
}
template <int B, int F>
HD Fq<2> fq_from_fs(const Fs<B, F>& a) {  // x 2^390 -> x 2^406, 0 < result < 2 p
    Fs<1, DC> c;
#pragma unroll
    for (int i = 0; i < SL; i++) c.v[i] = q30::C_TO_29[i];
    Fs<1, DU> t = mul<DU>(c, a);  // |t| < p
    uint32_t cy = 0;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) {  // t + p > 0: floor digits all non-negative
        const uint32_t s = (uint32_t)t.v[i] + (uint32_t)q30::PU[i] + cy;
        t.v[i] = (int32_t)(s & (uint32_t)SMASK);
        cy = s >> 30;
    }
    t.v[SL - 1] = t.v[SL - 1] + q30::PU[SL - 1] + (int32_t)cy;
    Fq<2> r;
    regroup_30_to_29(r.v, t.v);
    return r;
}

// ---- points -------------------------------------------------------------------------------------------------------------
struct AffS {  // canonical coordinates, exact centred digits; identity = (0, 0)
    Fs<1, DC> x, y;
};
// A GLV window-table entry: both coordinates canonical (Montgomery-390 values in [0, p)) as exact centred digits, packed
// 12 words each: word i = digit i's low 30 bits | two bits of the top digit (< 2^22) in bits 30-31.  Identity = all zero.
#ifdef TABS_STRIDE_128
// EXPERIMENT (tools/exp_variants.sh): one 128-byte line per entry instead of 96 B at 32-B alignment (1.5 lines on
// average).  Every table grows by a third: only widths <= 15 fit the HBM.
struct alignas(128) TabS {
    uint32_t w[24];
};
static_assert(sizeof(TabS) == 128, "one line per table entry");
#else
struct alignas(16) TabS {
    uint32_t w[24];
};
static_assert(sizeof(TabS) == 96, "packed table entries");
#endif
// c = a canonical value as non-negative floor digits
HD void tabs_pack_coord(uint32_t* w, const Fs<1, DU>& c) {
    int32_t d[SL];
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) {
        const int32_t u = c.v[i] + carry;  // <= 2^30
        carry = u >= SHALF ? 1 : 0;
        d[i] = u - (carry << 30);
    }
    d[SL - 1] = c.v[SL - 1] + carry;  // < 2^22
#pragma unroll
    for (int i = 0; i < SL - 1; i++) w[i] = ((uint32_t)d[i] & (uint32_t)SMASK) | ((((uint32_t)d[SL - 1] >> (2 * i)) & 3u) << 30);
}
HD Fs<1, DC> tabs_unpack_coord(const uint32_t* w) {
    Fs<1, DC> r;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) r.v[i] = (int32_t)(w[i] << 2) >> 2;
    uint32_t t = w[SL - 2] >> 30;
#pragma unroll
    for (int i = SL - 3; i >= 0; i--) t = (t << 2) | (w[i] >> 30);  // one v_alignbit_b32 each
    r.v[SL - 1] = (int32_t)t;
    return r;
}
HD AffS tabs_unpack(const uint32_t* w) {
    AffS a;
    a.x = tabs_unpack_coord(w);
    a.y = tabs_unpack_coord(w + 12);
    return a;
}
// a canonical 14 x 29-bit coordinate (Montgomery-406, value < p) -> the packed words of the same field element
HD void tabs_pack_from_fq(uint32_t* w, const Fq<1>& a) {
    tabs_pack_coord(w, canonical_of_product(fs_from_fq<DU>(a)));
}
HD bool affine_is_inf(const AffS& q) {
    int32_t d = 0;
#pragma unroll
    for (int i = 0; i < SL; i++) d |= q.x.v[i] | q.y.v[i];
    return d == 0;
}
HD AffS affs_from_affq(const AffQ& a) {  // (0, 0) stays (0, 0)
    AffS r;
    uint32_t w[12];
    tabs_pack_from_fq(w, a.x);
    r.x = tabs_unpack_coord(w);
    tabs_pack_from_fq(w, a.y);
    r.y = tabs_unpack_coord(w);
    return r;
}

// Jacobian point of the folds and the constant multiplication: |x| <= 4 p, |y|, |z| <= p, centred digits; identity <=> z == 0 mod p
struct JacS {
    Fs<4, DC> x;
    Fs<1, DC> y, z;
};
HD JacS jacs_inf() {
    JacS r;
    r.x = relax<4, DC>(fs_one());
    r.y = fs_one();
    r.z = fs_zero();
    return r;
}
HD bool is_inf(const JacS& p) { return is_zero_slow(p.z); }
HD JacS jacs_from_jacq(const JacQ& p) {
    JacS r;
    r.x = relax<4, DC>(fs_from_fq(p.x));
    r.y = fs_from_fq(p.y);
    r.z = fs_from_fq(p.z);
    return r;
}
HD JacQ jacq_from_jacs(const JacS& p) {  // z == 0 mod p arrives as z = p: the 14 x 29-bit form's is_zero sees it
    JacQ r;
    r.x = relax<XB>(fq_from_fs(p.x));
    r.y = relax<XB>(fq_from_fs(p.y));
    r.z = relax<ZB>(fq_from_fs(p.z));
    return r;
}
HD JacS to_jacs(const AffS& a) {
    if (affine_is_inf(a)) return jacs_inf();
    JacS r;
    r.x = relax<4, DC>(a.x);
    r.y = a.y;
    r.z = fs_one();
    return r;
}
HD JacS neg(const JacS& p) {
    JacS r = p;
    r.y = neg(p.y);
    return r;
}

// dbl-2009-l in signed form: A = X^2, B = Y^2, XB = X B (= D / 4), E = 3 A, X3 = E^2 - 8 XB, Y3 = E (4 XB - X3) - 8 B^2,
// Z3 = 2 Y Z (3M + 4S).  General use (folds, slow paths): X3, Y3 (<= 9 p) are brought back to the stored bounds by one product
// with R each; the constant multiplication has its own chain form (g1_mulc30.hpp).
HD JacS dbl(const JacS& p) {
    const Fs<1, DC> A = sqr(p.x), B = sqr(p.y);
    const Fs<1, DC> xb = mul(B, p.x);
    const Fs<3, DC> E = mul_small<3>(A);
    const auto x3 = sqr_inj<-8, DC>(E, xb);                         // <= 1 + 8
    const auto g = sub(mul_small<2>(mul_small<2>(xb)), x3);         // 4 XB - X3: <= 13
    const Fs<1, DU> bb = sqr<DU>(B);
    const auto y3 = mul_inj<-8, DC>(E, g, bb);                      // <= 9
    JacS r;
    r.x = relax<4, DC>(mul(fs_one(), x3));
    r.y = mul(fs_one(), y3);
    r.z = mul(add_lazy(p.y, p.y), p.z);                             // identity stays identity
    return r;
}

// add-1998-cmo-2 (12M + 4S, no doublings of intermediates): p + q, or p - q when negq.  No exceptional-case test on the way:
// an identity operand, P + P and P - P all make Z3 = Z1 Z2 H vanish (a fresh product: zero iff its digits are zero).
HD JacS add_slow(const JacS& p, const JacS& q, bool negq);
// the common path alone: `degenerate` tells the caller that the result is void and add_slow has to be asked (a caller that can
// re-read its operands does so then, instead of keeping both points alive across the whole formula: k_g1slp.hip)
HD JacS add_unchecked(const JacS& p, const JacS& q, bool negq, bool& degenerate) {
    const Fs<1, DC> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    const Fs<1, DU> u1 = mul<DU>(z2z2, p.x);
    const Fs<1, DU> s1 = mul<DU>(mul(p.y, q.z), z2z2);
    const auto h = mul_inj<-1, DC>(z1z1, q.x, u1);                                  // U2 - U1: <= 2
    const auto rr = mul_inj<-1, DC>(mul(cneg(negq, q.y), p.z), z1z1, s1);           // +-S2 - S1: <= 2
    JacS r;
    r.z = mul(mul(p.z, q.z), h);
    degenerate = product_is_zero(r.z);
    const Fs<1, DC> hh = sqr(h), hhh = mul(h, hh);
    const Fs<1, DU> v = mul<DU>(hh, u1);
    r.x = sqr_inj2<-1, -2, DC>(rr, hhh, v);                                         // rr^2 - HHH - 2 V: <= 4
    r.y = mul_add<DC>(rr, sub(v, r.x), neg(s1), hhh);                               // rr (V - X3) - S1 HHH
    return r;
}
HD JacS add(const JacS& p, const JacS& q, bool negq = false) {
    const Fs<1, DC> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    const Fs<1, DU> u1 = mul<DU>(z2z2, p.x);
    const Fs<1, DU> s1 = mul<DU>(mul(p.y, q.z), z2z2);
    const auto h = mul_inj<-1, DC>(z1z1, q.x, u1);                                  // U2 - U1: <= 2
    const auto rr = mul_inj<-1, DC>(mul(cneg(negq, q.y), p.z), z1z1, s1);           // +-S2 - S1: <= 2
    const Fs<1, DC> hh = sqr(h), hhh = mul(h, hh);
    const Fs<1, DU> v = mul<DU>(hh, u1);
    JacS r;
    r.x = sqr_inj2<-1, -2, DC>(rr, hhh, v);                                         // rr^2 - HHH - 2 V: <= 4
    r.y = mul_add<DC>(rr, sub(v, r.x), neg(s1), hhh);                               // rr (V - X3) - S1 HHH
    r.z = mul(mul(p.z, q.z), h);
    if (product_is_zero(r.z)) return add_slow(p, q, negq);
    return r;
}
HD JacS add_slow(const JacS& p, const JacS& q, bool negq) {
    if (is_inf(p)) return negq ? neg(q) : q;
    if (is_inf(q)) return p;
    // Z1, Z2 != 0 and H == 0: same x.  Same y -> doubling, opposite y -> identity.
    const Fs<1, DC> z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    const Fs<1, DU> s1 = mul<DU>(mul(p.y, q.z), z2z2);
    const auto rr = mul_inj<-1, DC>(mul(cneg(negq, q.y), p.z), z1z1, s1);
    if (is_zero_slow(rr)) return dbl(p);
    return jacs_inf();
}
// p + q AND p - q (the sum-and-difference pairs of the G1 linear map, k_g1slp.hip) with everything the two share computed once:
// Z1Z1, Z2Z2, U1, S1, H, HH, HHH, V, Z3 and S2's first product -- 10M + 3S -- then one reduction for +-S2 - S1, one square and one
// product pair per result (two separate additions: 24M + 8S + two pairs; this: 12M + 5S + two pairs).  In two steps so that the
// caller can store the first result before the second is computed (six shared values live instead of two whole points).
// Z3 = Z1 Z2 H is a fresh product: zero iff an operand is the identity or p = +-q -- `degenerate`: the caller takes both results
// from add_slow (the shared values are then meaningless).
struct AddSubSharedS {
    Fs<1, DC> t, z1z1, hhh, z3;  // y2 z1, z1^2, H^3, Z3
    Fs<1, DU> s1, v;             // S1, V = U1 H^2
    bool degenerate;
};
HD AddSubSharedS add_sub_prepare(const JacS& p, const JacS& q) {
    AddSubSharedS sh;
    const Fs<1, DC> z2z2 = sqr(q.z);
    sh.z1z1 = sqr(p.z);
    const Fs<1, DU> u1 = mul<DU>(z2z2, p.x);
    sh.s1 = mul<DU>(mul(p.y, q.z), z2z2);
    const auto h = mul_inj<-1, DC>(sh.z1z1, q.x, u1);                               // U2 - U1: <= 2
    sh.t = mul(q.y, p.z);
    const Fs<1, DC> hh = sqr(h);
    sh.hhh = mul(h, hh);
    sh.v = mul<DU>(hh, u1);
    sh.z3 = mul(mul(p.z, q.z), h);
    sh.degenerate = product_is_zero(sh.z3);
    return sh;
}
HD JacS add_sub_finish(const AddSubSharedS& sh, bool negq) {  // p + q, or p - q when negq (not for degenerate operands)
    const auto rr = mul_inj<-1, DC>(cneg(negq, sh.t), sh.z1z1, sh.s1);              // +-S2 - S1: <= 2
    JacS r;
    r.x = sqr_inj2<-1, -2, DC>(rr, sh.hhh, sh.v);                                    // <= 4
    r.y = mul_add<DC>(rr, sub(sh.v, r.x), neg(sh.s1), sh.hhh);                       // rr (V - X3) - S1 HHH
    r.z = sh.z3;
    return r;
}
// both at once (host tests; sum / diff may alias p or q)
HD void add_sub(const JacS& p, const JacS& q, JacS& sum, JacS& diff) {
    const AddSubSharedS sh = add_sub_prepare(p, q);
    if (sh.degenerate) {
        const JacS s_ = add_slow(p, q, false), d_ = add_slow(p, q, true);
        sum = s_;
        diff = d_;
        return;
    }
    sum = add_sub_finish(sh, false);
    diff = add_sub_finish(sh, true);
}
template <int B>
HD JacS apply_phi(const JacS& p, const Fs<B, DC>& beta) {
    JacS r = p;
    r.x = relax<4, DC>(mul(beta, p.x));
    return r;
}

// ---- XYZZ accumulator of the MSM (madd-2008-s): (X, Y, ZZ, ZZZ) stands for (X / ZZ, Y / ZZZ); identity <=> ZZ == 0 -----------
struct XyzzS {
    Fs<4, DU> x;
    Fs<1, DU> y, zz, zzz;
};
HD XyzzS xyzz30_inf() {
    XyzzS r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.x.v[i] = r.y.v[i] = r.zz.v[i] = r.zzz.v[i] = 0;
    return r;
}
HD bool is_inf(const XyzzS& p) { return product_is_zero(p.zz); }
HD JacS to_jacs(const XyzzS& p) {  // (X ZZ, Y ZZZ, ZZ)
    const Fs<1, DC> zz = normalise(p.zz), zzz = normalise(p.zzz);
    JacS r;
    r.x = relax<4, DC>(mul(zz, p.x));
    r.y = mul(zzz, p.y);
    r.z = zz;
    return r;
}
// The exact slow path is a real call on the device.  Inlined, whatever is live across it (the prefetched entry, the digit
// registers) is spilled at the top of EVERY addition (38 scratch stores on the hot path).  The call takes pointers to COPIES
// made inside the cold branch: handing it references to the caller's own values would give those an address, i.e. keep the
// accumulator in scratch memory for the whole loop (measured: 130 dwords stored per addition, MSM 38 -> 51 ms).
#if defined(__HIP_DEVICE_COMPILE__)
#define SLOW_PATH_FN __device__ __attribute__((noinline))
#else
#define SLOW_PATH_FN inline
#endif
HD XyzzS add_mixed_slow(const XyzzS& p, const AffS& q, bool negq);
// nine reductions: 338 * 6 + 260 * 2 + 507 multiply-adds + 9 * 13 for the Montgomery digits, 39 injected digits
HD XyzzS add_mixed(const XyzzS& p, const AffS& q, bool negq = false) {
    if (affine_is_inf(q)) return p;
    const auto P = mul_inj<-1, DC>(q.x, p.zz, p.x);                  // U2 - X1: <= 5
    const auto R = mul_inj<-1, DC>(cneg(negq, q.y), p.zzz, p.y);      // +-S2 - Y1: <= 2
    const Fs<1, DC> PP = sqr(P), PPP = mul(P, PP);
    const Fs<1, DU> Q = mul<DU>(PP, p.x);
    XyzzS r;
    r.x = sqr_inj2<-1, -2, DU>(R, PPP, Q);                            // R^2 - PPP - 2 Q: <= 4
    r.y = mul_add<DU>(R, sub_lazy(Q, r.x), neg(p.y), PPP);            // R (Q - X3) - Y1 PPP, one reduction, split columns
    r.zz = mul<DU>(PP, p.zz);
    r.zzz = mul<DU>(PPP, p.zzz);
    if (__builtin_expect(product_is_zero(r.zz), 0)) r = add_mixed_slow(p, q, negq);  // identity accumulator, or equal x: P + P / P - P
    return r;
}
HD XyzzS add_mixed_slow_impl(const XyzzS& p, const AffS& q, bool negq) {
    if (affine_is_inf(q)) return p;
    JacS j;
    if (is_inf(p)) {
        j = to_jacs(q);
        if (negq) j = neg(j);
    } else {
        const auto R = mul_inj<-1, DC>(cneg(negq, q.y), p.zzz, p.y);
        if (!is_zero_slow(R)) return xyzz30_inf();                    // opposite points
        j = dbl(to_jacs(q));                                          // equal points: 2 q
        if (negq) j = neg(j);
    }
    XyzzS r;  // Jacobian (X, Y, Z) -> (X, Y, Z^2, Z^3), floor digits
    r.x = relax<4, DU>(mul<DU>(fs_one(), j.x));
    r.y = mul<DU>(fs_one(), j.y);
    r.zz = sqr<DU>(j.z);
    r.zzz = mul<DU>(j.z, r.zz);
    return r;
}

// device: operands and result travel through a plain word buffer that exists only inside the cold branch (neither the
// accumulator nor the result may have their address taken: they would live in scratch memory for the whole loop)
SLOW_PATH_FN void add_mixed_slow_call(int32_t* buf, bool negq) {
    XyzzS p;
    AffS q;
#pragma unroll
    for (int i = 0; i < SL; i++) {
        p.x.v[i] = buf[i]; p.y.v[i] = buf[SL + i]; p.zz.v[i] = buf[2 * SL + i]; p.zzz.v[i] = buf[3 * SL + i];
        q.x.v[i] = buf[4 * SL + i]; q.y.v[i] = buf[5 * SL + i];
    }
    const XyzzS r = add_mixed_slow_impl(p, q, negq);
#pragma unroll
    for (int i = 0; i < SL; i++) { buf[i] = r.x.v[i]; buf[SL + i] = r.y.v[i]; buf[2 * SL + i] = r.zz.v[i]; buf[3 * SL + i] = r.zzz.v[i]; }
}
HD XyzzS add_mixed_slow(const XyzzS& p, const AffS& q, bool negq) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t buf[6 * SL];
#pragma unroll
    for (int i = 0; i < SL; i++) {
        buf[i] = p.x.v[i]; buf[SL + i] = p.y.v[i]; buf[2 * SL + i] = p.zz.v[i]; buf[3 * SL + i] = p.zzz.v[i];
        buf[4 * SL + i] = q.x.v[i]; buf[5 * SL + i] = q.y.v[i];
    }
    add_mixed_slow_call(buf, negq);
    XyzzS r;
#pragma unroll
    for (int i = 0; i < SL; i++) { r.x.v[i] = buf[i]; r.y.v[i] = buf[SL + i]; r.zz.v[i] = buf[2 * SL + i]; r.zzz.v[i] = buf[3 * SL + i]; }
    return r;
#else
    return add_mixed_slow_impl(p, q, negq);
#endif
}

// ---- the constant multiplication's chain (g1_mulc30.hpp): halved doubling and mixed addition on centred digits ------------------
// Doubling with the projective scaling lambda = 1/2 -- (X3 / 4, Y3 / 8, Z3 / 2) is the same point -- which removes every
// power of two from dbl-2009-l: with A = X^2, B = Y^2, M = X B and the slope H = 3 A / 2 (one halving mod p),
//     X3 = H^2 - 2 M,   Y3 = H (M - X3) - B^2,   Z3 = Y Z.
// Six reductions (2M + 3S + one product pair), M - X3 used un-normalised, -2 M injected: 2,054 multiply-adds against 2,317 of
// the 14 x 29-bit form's doubling.  The identity (Z = 0) stays the identity.
HD JacS dbl_half(const JacS& p) {
    const Fs<1, DC> A = sqr(p.x), B = sqr(p.y);
    const Fs<1, DC> M = mul(B, p.x);
    const Fs<2, DC> H = half_of_triple(A);
    JacS r;
    r.x = relax<4, DC>(sqr_inj<-2, DC>(H, M));                 // <= 3
    r.y = mul_add<DC>(H, sub_lazy(M, r.x), neg(B), B);         // H (M - X3) - B^2, one reduction
    r.z = mul(p.y, p.z);
    return r;
}
// an affine point whose coordinates are fresh products (the multiplication's table on the isomorphic curve, never the identity)
struct AffT {
    Fs<1, DC> x, y;
};
// add-1998-cmo-2 with Z2 = 1 (7M + 3S + one product pair, no doublings of intermediates): p + q, or p - q when negq
HD JacS add_mixed_slow(const JacS& p, const AffT& q, bool negq);
HD JacS add_mixed(const JacS& p, const AffT& q, bool negq) {
    const Fs<1, DU> z1z1 = sqr<DU>(p.z);
    const auto h = mul_inj<-1, DC>(q.x, z1z1, p.x);                                  // U2 - X1: <= 5
    const auto rr = mul_inj<-1, DC>(mul(cneg(negq, q.y), p.z), z1z1, p.y);           // +-S2 - Y1: <= 2
    const Fs<1, DC> hh = sqr(h), hhh = mul(h, hh);
    const Fs<1, DC> v = mul(hh, p.x);
    JacS r;
    r.x = sqr_inj2<-1, -2, DC>(rr, hhh, v);                                          // <= 4
    r.y = mul_add<DC>(rr, sub_lazy(v, r.x), neg(p.y), hhh);                          // rr (V - X3) - Y1 HHH
    r.z = mul(h, p.z);
    if (__builtin_expect(product_is_zero(r.z), 0)) r = add_mixed_slow(p, q, negq);   // identity accumulator, P + P, P - P
    return r;
}
HD JacS add_mixed_slow_impl(const JacS& p, const AffT& q, bool negq) {
    JacS qj;
    qj.x = relax<4, DC>(q.x);
    qj.y = q.y;
    qj.z = fs_one();
    return add_slow(p, qj, negq);
}
SLOW_PATH_FN void add_mixed_slow_call_jacs(int32_t* buf, bool negq) {
    JacS p;
    AffT q;
#pragma unroll
    for (int i = 0; i < SL; i++) {
        p.x.v[i] = buf[i]; p.y.v[i] = buf[SL + i]; p.z.v[i] = buf[2 * SL + i];
        q.x.v[i] = buf[3 * SL + i]; q.y.v[i] = buf[4 * SL + i];
    }
    const JacS r = add_mixed_slow_impl(p, q, negq);
#pragma unroll
    for (int i = 0; i < SL; i++) { buf[i] = r.x.v[i]; buf[SL + i] = r.y.v[i]; buf[2 * SL + i] = r.z.v[i]; }
}
HD JacS add_mixed_slow(const JacS& p, const AffT& q, bool negq) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t buf[5 * SL];  // copies made inside the cold branch (see the XYZZ slow path above)
#pragma unroll
    for (int i = 0; i < SL; i++) {
        buf[i] = p.x.v[i]; buf[SL + i] = p.y.v[i]; buf[2 * SL + i] = p.z.v[i];
        buf[3 * SL + i] = q.x.v[i]; buf[4 * SL + i] = q.y.v[i];
    }
    add_mixed_slow_call_jacs(buf, negq);
    JacS r;
#pragma unroll
    for (int i = 0; i < SL; i++) { r.x.v[i] = buf[i]; r.y.v[i] = buf[SL + i]; r.z.v[i] = buf[2 * SL + i]; }
    return r;
#else
    return add_mixed_slow_impl(p, q, negq);
#endif
}

}  // namespace kzg
