// Signed unsaturated Fp for the two kernels that are 91 % of a prover step (the GLV window MSM and the constant
// multiplications of the G1 linear map): 13 digits of 30 bits in i32 registers, digits centred in [-2^29, 2^29], Montgomery
// radix R = 2^390, one v_mad_i64_i32 per digit product.
//
// Why 13 signed digits instead of the 14 unsigned 29-bit ones of fp29.hpp: a Montgomery product costs 2 n^2 + n
// multiply-adds, 351 instead of 406 (squaring 260 instead of 301), and the multiply-adds are 80 % of the MSM's time.
// A signed 64-bit column holds 2^63: with centred digits a column of the product scan is at most 12 full a_i b_j terms
// (the top digit of a value below 32 p is tiny) and the m_i p_j terms, whose sum is bounded by sum |P_j| = 6.21 * 2^29
// -- BLS12-381's p happens to have small centred digits -- so in units of U = 2^58:
//      C x C  : 12 + 0.5 + 6.21          = 18.7 U
//      C x W  : 24 + 0.5 + 6.21          = 30.7 U   < 32 U = 2^63
//      C x C + C x C (one reduction)     = 31.2 U
// where C = centred digits (|d| <= 2^29 + 4) and W = wide digits (|d| <= 2^30 + 8: a product's unsigned floor digits U,
// or an un-normalised sum / difference of two values).  So ONE operand of every product may be wide, and
//   * a product leaves as floor digits (DU: "shift and mask" per column, as cheap as fp29's) unless a later squaring or a
//     product with another wide value needs it centred (DC: three more instructions per digit);
//   * a - b of two floor-digit values is used as it stands (no multiple of p, no carry sweep), negation is 13 subtractions;
//   * "product minus stored value" (U2 - X1, S2 - Y1, R^2 - PPP - 2 Q of the mixed addition) is ONE reduction: the
//     subtrahend's digits are injected into the upper columns as multiply-adds by the inline constants -1 / -2
//     (REDC(a b - x R) = a b / R - x).
// Values are signed: Fs<B, F> holds |value| <= B p.  A product of operands with A B <= 256 has |value| < 0.91 p
// (|a b| / R <= 256 * p * 0.0016, |m p| / R <= p / 2), so a fresh product is zero mod p only if it IS zero.
// Bounds and digit classes live in the type and every formula is checked at compile time, as in fp29.hpp.
// Checked on the CPU against the saturated field (tests/c/test_fp30.cpp) and against Python's integers (tests/test_host_units.py).
#pragma once
#include "field.hpp"
#include "fp30_consts.hpp"
#if defined(__HIP_DEVICE_COMPILE__)
#include "fp30_mac.hpp"
#endif
#include <utility>

namespace kzg {

constexpr int SL = 13;
constexpr int32_t SMASK = (1 << 30) - 1;
constexpr int32_t SHALF = 1 << 29;
// digit classes (digits 0..11; the top digit is whatever is left and stays below 2^26 for |value| <= 32 p)
constexpr int DC = 1;  // centred: |d| <= 2^29 + 4
constexpr int DU = 2;  // floor digits of a product: 0 <= d < 2^30 (the top digit carries the sign)
constexpr int DW = 3;  // wide: |d| <= 2^30 + 8 (difference of two DU values, sum / difference of two DC values)

template <int B, int F = DC>
struct Fs {
    static_assert(B >= 1 && B <= 256, "bound out of range");
    static_assert(F == DC || F == DU || F == DW, "digit class");
    int32_t v[SL];
};

template <int B2, int F2, int B, int F>
HD Fs<B2, F2> relax(const Fs<B, F>& a) {
    static_assert(B2 >= B, "relax can only widen a bound");
    static_assert(F2 == F || F2 == DW, "a digit class can only be widened");
    Fs<B2, F2> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = a.v[i];
    return r;
}

namespace q30core {
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_DEVICE_ASM 1
#else
#define FS_DEVICE_ASM 0
#endif
// acc += sum_{j < N} x[j] * y[-j]
template <int N>
HD void run_vv(int64_t& acc, const int32_t* x, const int32_t* y) {
    if constexpr (N > 0) {
#if FS_DEVICE_ASM
        MacS<N>::vv(acc, x, y);
#else
#pragma unroll
        for (int j = 0; j < N; j++) acc += (int64_t)x[j] * y[-j];
#endif
    }
}
// acc += sum_{j < N} m[j] * P[K - j]
template <int N, int K>
HD void run_vp(int64_t& acc, const int32_t* m) {
    if constexpr (N > 0) {
#if FS_DEVICE_ASM
        MacS<N>::template vp<K>(acc, m);
#else
#pragma unroll
        for (int j = 0; j < N; j++) acc += (int64_t)m[j] * q30::P[K - j];
#endif
    }
}
// the same followed by C1 * x (+ C2 * y): digits injected into an upper column
template <int N, int K, int C1>
HD void run_vp1(int64_t& acc, const int32_t* m, int32_t x) {
#if FS_DEVICE_ASM
    MacS<N>::template vp1<K, C1>(acc, m, x);
#else
#pragma unroll
    for (int j = 0; j < N; j++) acc += (int64_t)m[j] * q30::P[K - j];
    acc += (int64_t)C1 * x;
#endif
}
template <int N, int K, int C1, int C2>
HD void run_vp2(int64_t& acc, const int32_t* m, int32_t x, int32_t y) {
#if FS_DEVICE_ASM
    MacS<N>::template vp2<K, C1, C2>(acc, m, x, y);
#else
#pragma unroll
    for (int j = 0; j < N; j++) acc += (int64_t)m[j] * q30::P[K - j];
    acc += (int64_t)C1 * x + (int64_t)C2 * y;
#endif
}

// products of column K (0 <= K <= 24 carry products, K = 25 none): a * b
struct ProdMul {
    const int32_t *a, *b;
    template <int K>
    HD void col(int64_t& acc) const {
        constexpr int lo = K < SL ? 0 : K - SL + 1, hi = K < SL ? K : SL - 1;
        run_vv<hi - lo + 1>(acc, a + lo, b + (K - lo));
    }
};
// a * a with the cross terms taken once against the doubled operand a2 = 2 a
struct ProdSqr {
    const int32_t *a, *a2;
    template <int K>
    HD void col(int64_t& acc) const {
        constexpr int lo = K < SL ? 0 : K - SL + 1;
        constexpr int n = (K + 1) / 2 - lo;  // cross terms i < K - i, i >= lo
        run_vv<(n > 0 ? n : 0)>(acc, a2 + lo, a + (K - lo));
        if constexpr ((K & 1) == 0 && K / 2 < SL) run_vv<1>(acc, a + K / 2, a + K / 2);
    }
};
// a * b + c * d
struct ProdMul2 {
    const int32_t *a, *b, *c, *d;
    template <int K>
    HD void col(int64_t& acc) const {
        constexpr int lo = K < SL ? 0 : K - SL + 1, hi = K < SL ? K : SL - 1;
        run_vv<hi - lo + 1>(acc, a + lo, b + (K - lo));
        run_vv<hi - lo + 1>(acc, c + lo, d + (K - lo));
    }
};
// injected digits: result = REDC(products) + C1 * x (+ C2 * y)
struct Inj0 {
    static constexpr int N = 0;
};
template <int C1_>
struct Inj1 {
    static constexpr int N = 1, C1 = C1_;
    const int32_t* x;
};
template <int C1_, int C2_>
struct Inj2 {
    static constexpr int N = 2, C1 = C1_, C2 = C2_;
    const int32_t *x, *y;
};

template <int OUTF>
HD void extract(int64_t& acc, int32_t& r) {
    if constexpr (OUTF == DC) {  // centred digit, rounded carry
        const int64_t t = acc + SHALF;
        r = ((int32_t)t & SMASK) - SHALF;
        acc = t >> 30;
    } else {  // floor digit
        r = (int32_t)acc & SMASK;
        acc >>= 30;
    }
}
template <int K, class Prod>
HD void lo_col(int64_t& acc, const Prod& pr, int32_t* m) {
    pr.template col<K>(acc);
    run_vp<K, K>(acc, m);
    m[K] = (int32_t)((uint32_t)acc * q30::N0Q) >> 2;  // centred digit of -acc / p mod 2^30
    run_vp<1, 0>(acc, m + K);
    acc >>= 30;  // exact: the low 30 bits are zero
}
template <int K, int OUTF, class Prod, class Inj>
HD void hi_col(int64_t& acc, const Prod& pr, const int32_t* m, const Inj& inj, int32_t* r) {
    constexpr int lo = K - SL + 1, n = SL - lo, J = K - SL;  // emits digit J
    pr.template col<K>(acc);
    if constexpr (Inj::N == 0) run_vp<n, K - lo>(acc, m + lo);
    else if constexpr (Inj::N == 1) run_vp1<n, K - lo, Inj::C1>(acc, m + lo, inj.x[J]);
    else run_vp2<n, K - lo, Inj::C1, Inj::C2>(acc, m + lo, inj.x[J], inj.y[J]);
    if constexpr (J < SL - 1) extract<OUTF>(acc, r[J]);
    else r[J] = (int32_t)acc;
}
template <int OUTF, class Prod, class Inj, int... Ks>
HD void mont(const Prod& pr, const Inj& inj, int32_t* r, std::integer_sequence<int, Ks...>) {
    int32_t m[SL];
    int64_t acc = 0;
    (lo_col<Ks>(acc, pr, m), ...);
    (hi_col<SL + Ks, OUTF>(acc, pr, m, inj, r), ...);
}
using Seq = std::make_integer_sequence<int, SL>;

// a * b + c * d where the products' operands may be wide: the columns that could pass 2^63 in one accumulator (7 and more terms
// per product) keep a * b in a second accumulator whose low 30 bits are handed over before every shift.
template <int K>
constexpr bool split_col() {
    constexpr int terms = (K < SL ? K : 2 * SL - 2 - K) + 1;
    return terms >= 7;
}
template <int K>
HD void lo_col2w(int64_t& acc, int64_t& acc1, const ProdMul2& pr, int32_t* m) {
    constexpr int n = K + 1;
    if constexpr (split_col<K>()) {
        run_vv<n>(acc1, pr.a, pr.b + K);
        run_vv<n>(acc, pr.c, pr.d + K);
        run_vp<K, K>(acc, m);
        acc += (int32_t)acc1 & SMASK;
        acc1 >>= 30;
    } else {
        run_vv<n>(acc, pr.a, pr.b + K);
        run_vv<n>(acc, pr.c, pr.d + K);
        run_vp<K, K>(acc, m);
    }
    m[K] = (int32_t)((uint32_t)acc * q30::N0Q) >> 2;
    run_vp<1, 0>(acc, m + K);
    acc >>= 30;
}
template <int K, int OUTF>
HD void hi_col2w(int64_t& acc, int64_t& acc1, const ProdMul2& pr, const int32_t* m, int32_t* r) {
    constexpr int lo = K - SL + 1, n = SL - lo, J = K - SL;
    if constexpr (split_col<K>()) {
        run_vv<n>(acc1, pr.a + lo, pr.b + (K - lo));
        run_vv<n>(acc, pr.c + lo, pr.d + (K - lo));
        run_vp<n, K - lo>(acc, m + lo);
        acc += (int32_t)acc1 & SMASK;
        acc1 >>= 30;
    } else {
        run_vv<n>(acc, pr.a + lo, pr.b + (K - lo));
        run_vv<n>(acc, pr.c + lo, pr.d + (K - lo));
        run_vp<n, K - lo>(acc, m + lo);
    }
    if constexpr (J < SL - 1) extract<OUTF>(acc, r[J]);
    else r[J] = (int32_t)acc;
    if constexpr (split_col<K>() && !split_col<K + 1>()) acc += acc1;  // after the last split column the carry of a * b joins (< 2^34)
}
template <int OUTF, int... Ks>
HD void mont2w(const ProdMul2& pr, int32_t* r, std::integer_sequence<int, Ks...>) {
    int32_t m[SL];
    int64_t acc = 0, acc1 = 0;
    (lo_col2w<Ks>(acc, acc1, pr, m), ...);
    (hi_col2w<SL + Ks, OUTF>(acc, acc1, pr, m, r), ...);
}
}  // namespace q30core

// ---- products -------------------------------------------------------------------------------------------------
template <int A, int FA, int B, int FB>
constexpr bool fs_mul_ok() {
    // one operand centred; a wide operand's top digit must stay below 2^27: |value| <= 64 p (column 12: 24 + 0.75 + 6.21 U)
    return (FA == DC || FB == DC) && (long)A * B <= 256 && (FA == DC || A <= 64) && (FB == DC || B <= 64);
}
// a * b / R
template <int OUTF = DC, int A, int FA, int B, int FB>
HD Fs<1, OUTF> mul(const Fs<A, FA>& a, const Fs<B, FB>& b) {
    static_assert(OUTF == DC || OUTF == DU, "a product leaves centred or as floor digits");
    static_assert(fs_mul_ok<A, FA, B, FB>(), "mul: one operand must be centred, bounds A * B <= 256, wide operands <= 32 p");
    Fs<1, OUTF> r;
    q30core::mont<OUTF>(q30core::ProdMul{a.v, b.v}, q30core::Inj0{}, r.v, q30core::Seq{});
    return r;
}
template <int OUTF = DC, int A>
HD Fs<1, OUTF> sqr(const Fs<A, DC>& a) {
    static_assert((long)A * A <= 256, "sqr: operand bound too large");
    int32_t a2[SL];
#pragma unroll
    for (int i = 0; i < SL; i++) a2[i] = a.v[i] * 2;
    Fs<1, OUTF> r;
    q30core::mont<OUTF>(q30core::ProdSqr{a.v, a2}, q30core::Inj0{}, r.v, q30core::Seq{});
    return r;
}
// a * b / R + C1 * x   (C1 = +-1, +-2: "product minus a stored value" in ONE reduction; x of any digit class)
template <int C1, int OUTF = DC, int A, int FA, int B, int FB, int X, int FX>
HD Fs<1 + (C1 < 0 ? -C1 : C1) * X, OUTF> mul_inj(const Fs<A, FA>& a, const Fs<B, FB>& b, const Fs<X, FX>& x) {
    static_assert(fs_mul_ok<A, FA, B, FB>(), "mul_inj: operand classes / bounds");
    static_assert(C1 >= -16 && C1 <= 16 && C1 != 0 && X <= 32, "mul_inj: injected value (inline constant)");
    Fs<1 + (C1 < 0 ? -C1 : C1) * X, OUTF> r;
    q30core::mont<OUTF>(q30core::ProdMul{a.v, b.v}, q30core::Inj1<C1>{x.v}, r.v, q30core::Seq{});
    return r;
}
// a^2 / R + C1 * x + C2 * y
template <int C1, int C2, int OUTF = DC, int A, int X, int FX, int Y, int FY>
HD Fs<1 + (C1 < 0 ? -C1 : C1) * X + (C2 < 0 ? -C2 : C2) * Y, OUTF> sqr_inj2(const Fs<A, DC>& a, const Fs<X, FX>& x, const Fs<Y, FY>& y) {
    static_assert((long)A * A <= 256 && X <= 32 && Y <= 32 && C1 >= -16 && C1 <= 16 && C2 >= -16 && C2 <= 16, "sqr_inj2: bounds");
    int32_t a2[SL];
#pragma unroll
    for (int i = 0; i < SL; i++) a2[i] = a.v[i] * 2;
    Fs<1 + (C1 < 0 ? -C1 : C1) * X + (C2 < 0 ? -C2 : C2) * Y, OUTF> r;
    q30core::mont<OUTF>(q30core::ProdSqr{a.v, a2}, q30core::Inj2<C1, C2>{x.v, y.v}, r.v, q30core::Seq{});
    return r;
}
template <int C1, int OUTF = DC, int A, int X, int FX>
HD Fs<1 + (C1 < 0 ? -C1 : C1) * X, OUTF> sqr_inj(const Fs<A, DC>& a, const Fs<X, FX>& x) {
    static_assert((long)A * A <= 256 && X <= 32 && C1 >= -16 && C1 <= 16, "sqr_inj: bounds");
    int32_t a2[SL];
#pragma unroll
    for (int i = 0; i < SL; i++) a2[i] = a.v[i] * 2;
    Fs<1 + (C1 < 0 ? -C1 : C1) * X, OUTF> r;
    q30core::mont<OUTF>(q30core::ProdSqr{a.v, a2}, q30core::Inj1<C1>{x.v}, r.v, q30core::Seq{});
    return r;
}
// (a * b + c * d) / R with ONE reduction.  All four operands centred: one accumulator (31.2 U).  With a wide operand in
// each product (a or b, c or d): the split form (the middle 13 columns keep a * b apart).
template <int OUTF = DC, int A, int FA, int B, int FB, int C, int FC, int D, int FD>
HD Fs<1, OUTF> mul_add(const Fs<A, FA>& a, const Fs<B, FB>& b, const Fs<C, FC>& c, const Fs<D, FD>& d) {
    static_assert((long)A * B + (long)C * D <= 256, "mul_add: operand bounds too large");
    static_assert(fs_mul_ok<A, FA, B, FB>() && fs_mul_ok<C, FC, D, FD>(), "mul_add: operand classes");
    Fs<1, OUTF> r;
    if constexpr (FA == DC && FB == DC && FC == DC && FD == DC)
        q30core::mont<OUTF>(q30core::ProdMul2{a.v, b.v, c.v, d.v}, q30core::Inj0{}, r.v, q30core::Seq{});
    else
        q30core::mont2w<OUTF>(q30core::ProdMul2{a.v, b.v, c.v, d.v}, r.v, q30core::Seq{});
    return r;
}

// ---- additive steps ---------------------------------------------------------------------------------------------
// digits back to the centred class: one parallel step (every digit hands its rounded carry to the next); any i32 digits on entry
// (carry = floor((d + 2^29) / 2^30) = ((d >> 29) + 1) >> 1 needs no headroom; the digit's low 30 bits are right under wrap-around)
template <int B, int F>
HD Fs<B, DC> normalise(const Fs<B, F>& a) {
    Fs<B, DC> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) {
        const uint32_t t = (uint32_t)a.v[i] + (uint32_t)SHALF;
        r.v[i] = (int32_t)(t & (uint32_t)SMASK) - SHALF + c;
        c = ((a.v[i] >> 29) + 1) >> 1;
    }
    r.v[SL - 1] = a.v[SL - 1] + c;
    return r;
}
// un-normalised forms: usable as ONE operand of a product (the other centred), or as an injected value
template <int A, int B>
HD Fs<A + B, DW> sub_lazy(const Fs<A, DU>& a, const Fs<B, DU>& b) {  // floor digits: |a_i - b_i| < 2^30
    Fs<A + B, DW> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
template <int A, int B>
HD Fs<A + B, DW> sub_lazy(const Fs<A, DC>& a, const Fs<B, DC>& b) {
    Fs<A + B, DW> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
template <int A, int B>
HD Fs<A + B, DW> add_lazy(const Fs<A, DC>& a, const Fs<B, DC>& b) {
    Fs<A + B, DW> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <int A>
HD Fs<A, DW> neg(const Fs<A, DU>& a) {
    Fs<A, DW> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = -a.v[i];
    return r;
}
template <int A>
HD Fs<A, DW> neg(const Fs<A, DW>& a) {
    Fs<A, DW> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = -a.v[i];
    return r;
}
template <int A>
HD Fs<A, DC> neg(const Fs<A, DC>& a) {
    Fs<A, DC> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = -a.v[i];
    return r;
}
// (negate ? -a : a), per lane
template <int A, int F>
HD Fs<A, (F == DC ? DC : DW)> cneg(bool negate, const Fs<A, F>& a) {
    Fs<A, (F == DC ? DC : DW)> r;
    const int32_t s = negate ? -1 : 1;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = a.v[i] * s;
    return r;
}
// normalised forms (centred out); at most one operand wide, or a difference of two floor-digit values: the digit sums fit i32
template <int FA, int FB>
constexpr bool fs_add_ok() { return FA == DC || FB == DC; }
template <int A, int FA, int B, int FB>
HD Fs<A + B, DC> add(const Fs<A, FA>& a, const Fs<B, FB>& b) {
    static_assert(fs_add_ok<FA, FB>(), "add: one operand must be centred");
    Fs<A + B, DW> t;
#pragma unroll
    for (int i = 0; i < SL; i++) t.v[i] = a.v[i] + b.v[i];
    return normalise(t);
}
template <int A, int FA, int B, int FB>
HD Fs<A + B, DC> sub(const Fs<A, FA>& a, const Fs<B, FB>& b) {
    static_assert(fs_add_ok<FA, FB>() || (FA == DU && FB == DU), "sub: one operand centred, or both floor digits");
    Fs<A + B, DW> t;
#pragma unroll
    for (int i = 0; i < SL; i++) t.v[i] = a.v[i] - b.v[i];
    return normalise(t);
}
template <int K, int A>
HD Fs<K * A, DC> mul_small(const Fs<A, DC>& a) {  // K a for K = 2, 3 (3 * (2^29 + 4) < 2^31)
    static_assert(K == 2 || K == 3, "mul_small: 2 a or 3 a");
    Fs<K * A, DW> t;
#pragma unroll
    for (int i = 0; i < SL; i++) t.v[i] = a.v[i] * K;
    return normalise(t);
}

// 3 a / 2 mod p for a FRESH centred product a (exact digits in [-2^29, 2^29)): t = 3 a, plus p if t is odd (p is odd), then one
// bit to the right across the digits -- floor halves, the dropped bit of digit i + 1 enters digit i as 2^29 -- and the carry
// step.  |3 a_i + p_i| < 3.97 * 2^29 fits i32 (max |P_i| = 0.964 * 2^29).  The slope of the halved doubling (curve30.hpp).
HD Fs<2, DC> half_of_triple(const Fs<1, DC>& a) {
    int32_t t[SL];
    const int32_t odd = -(a.v[0] & 1);  // 3 a_0 is odd iff a_0 is
#pragma unroll
    for (int i = 0; i < SL; i++) t[i] = a.v[i] * 3 + (odd & q30::P[i]);
    Fs<2, DW> h;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) h.v[i] = (t[i] >> 1) + ((t[i + 1] & 1) << 29);
    h.v[SL - 1] = t[SL - 1] >> 1;
    return normalise(h);
}

// ---- zero tests, canonical form ---------------------------------------------------------------------------------------
// a FRESH product (|value| < p, exact digits): zero mod p <=> every digit is zero
template <int F>
HD bool product_is_zero(const Fs<1, F>& t) {
    int32_t d = 0;
#pragma unroll
    for (int i = 0; i < SL; i++) d |= t.v[i];
    return d == 0;
}
HD Fs<1, DC> fs_one() {
    Fs<1, DC> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = q30::ONE[i];
    return r;
}
HD Fs<1, DC> fs_zero() {
    Fs<1, DC> r;
#pragma unroll
    for (int i = 0; i < SL; i++) r.v[i] = 0;
    return r;
}
// a fresh product as floor digits (|value| < p) -> its representative in [0, p), all 13 digits non-negative
HD Fs<1, DU> canonical_of_product(Fs<1, DU> t) {
    const bool negative = t.v[SL - 1] < 0;  // the lower digits are >= 0 and below 2^360 in total
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < SL - 1; i++) {
        const uint32_t s = (uint32_t)t.v[i] + (negative ? (uint32_t)q30::PU[i] : 0u) + c;
        t.v[i] = (int32_t)(s & (uint32_t)SMASK);
        c = s >> 30;
    }
    t.v[SL - 1] = t.v[SL - 1] + (negative ? q30::PU[SL - 1] : 0) + (int32_t)c;
    return t;
}
// the representative in [0, p) of any value
template <int B, int F>
HD Fs<1, DU> canonical(const Fs<B, F>& a) {
    return canonical_of_product(mul<DU>(fs_one(), a));  // a * R / R: same value mod p, |t| < p
}
template <int B, int F>
HD bool is_zero_slow(const Fs<B, F>& a) {
    return product_is_zero(canonical(a));
}

// ---- conversions ------------------------------------------------------------------------------------------------------
// a non-negative integer below 2^390 given as n32 words of 32 bits -> 13 floor digits of 30 bits
template <int NW>
HD void regroup_32_to_30(int32_t* out, const uint32_t* in) {
#pragma unroll
    for (int i = 0; i < SL; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint64_t two = w < NW ? in[w] : 0u;
        if (w + 1 < NW) two |= (uint64_t)in[w + 1] << 32;
        out[i] = (int32_t)((uint32_t)(two >> s) & (uint32_t)SMASK);
    }
}
// 13 non-negative floor digits (value < 2^384) -> 12 words
HD void regroup_30_to_32(uint32_t* out, const int32_t* in) {
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int lo = (32 * w) / 30, sh = 32 * w - 30 * lo;
        uint64_t acc = (uint64_t)(uint32_t)in[lo] >> sh;
        int have = 30 - sh;
        if (lo + 1 < SL) { acc |= (uint64_t)(uint32_t)in[lo + 1] << have; have += 30; }
        if (have < 32 && lo + 2 < SL) acc |= (uint64_t)(uint32_t)in[lo + 2] << have;
        out[w] = (uint32_t)acc;
    }
}
// Fp (Montgomery-384, canonical) -> Fs (Montgomery-390), |value| < p
template <int OUTF = DC>
HD Fs<1, OUTF> fs_from_fp(const Fp& a) {
    Fs<1, DU> t;
    regroup_32_to_30<12>(t.v, a.v);
    Fs<1, DC> c;
#pragma unroll
    for (int i = 0; i < SL; i++) c.v[i] = q30::C_FROM_FP[i];
    return mul<OUTF>(c, t);
}
// Fs -> Fp (Montgomery-384, canonical)
template <int B, int F>
HD Fp fp_from_fs(const Fs<B, F>& a) {
    Fs<1, DC> c;
#pragma unroll
    for (int i = 0; i < SL; i++) c.v[i] = q30::C_TO_FP[i];
    const Fs<1, DU> m = canonical(mul<DC>(c, a));  // a * 2^384 / 2^390, then the representative in [0, p)
    Fp r;
    regroup_30_to_32(r.v, m.v);
    return r;
}

}  // namespace kzg
