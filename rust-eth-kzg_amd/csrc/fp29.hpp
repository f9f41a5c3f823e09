// Unsaturated Fp for the hot kernels (MSM accumulation, G1-FFT): 14 limbs of 29 bits in u32 registers,
// Montgomery radix R = 2^406.  Why: on gfx950 every integer VALU op (v_mad_u64_u32 included) issues at the
// same rate, so instruction count is the cost.  With 29-bit limbs a whole product-scanning column
// (14 a*b + 14 m*p products, each < 2^58) fits one 64-bit accumulator: 392 v_mad_u64_u32 and NO carry
// instructions per multiplication (the saturated 12 x 32-bit form needs 288 MACs + 288 carry folds), no final
// conditional subtraction (25 spare bits: values live in [0, B*p) with a small static bound B), and additions
// are 14 independent v_add_u32 plus one carry sweep.  Measured 75.8 G mul/s vs 57.8 G mul/s (tools/ubench, u29).
//
// Bounds are tracked in the type: Fq<B> holds a value < B*p with normalised limbs (< 2^29 except the top one).
// mul/sqr require A*B <= 2^24 and return Fq<2>; sub adds the multiple K*p, K = pow2 >= 2*bound(b).
#pragma once
#include "field.hpp"
#include "fp29_consts.hpp"
#include "fp29_mac.hpp"
#include <utility>


namespace kzg {

constexpr int QL = 14;
constexpr uint32_t QMASK = (1u << 29) - 1;

template <int B>
struct Fq {
    static_assert(B >= 1 && B <= (1 << 20), "bound out of range");
    uint32_t v[QL];
};

constexpr int pow2_at_least(int x) {
    int k = 2;
    while (k < x) k <<= 1;
    return k;
}
constexpr int log2_exact(int k) {
    int e = 0;
    while ((1 << e) < k) e++;
    return e;
}

template <int B2, int B>
HD Fq<B2> relax(const Fq<B>& a) {
    static_assert(B2 >= B, "relax can only widen a bound");
    Fq<B2> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i];
    return r;
}

// carry sweep: limbs 0..12 back below 2^29
template <int B>
HD void normalise(Fq<B>& a) {
#pragma unroll
    for (int i = 0; i < QL - 1; i++) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= QMASK;
    }
}

template <int A, int B>
HD Fq<A + B> add(const Fq<A>& a, const Fq<B>& b) {
    Fq<A + B> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] + b.v[i];
    normalise(r);
    return r;
}
template <int A>
HD Fq<2 * A> dbl(const Fq<A>& a) {
    Fq<2 * A> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] << 1;
    normalise(r);
    return r;
}
// a - b + K*p, K = 2^e >= 2 * bound(b)
template <int A, int B>
HD Fq<A + pow2_at_least(2 * B)> sub(const Fq<A>& a, const Fq<B>& b) {
    constexpr int K = pow2_at_least(2 * B);
    constexpr int E = log2_exact(K);
    static_assert(E >= 1 && E <= 12, "subtrahend bound too large");
    Fq<A + K> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] + q29::SUBK[E - 1][i] - b.v[i];
    normalise(r);
    return r;
}
template <int B>
HD Fq<pow2_at_least(2 * B)> neg(const Fq<B>& b) {
    constexpr int K = pow2_at_least(2 * B);
    constexpr int E = log2_exact(K);
    static_assert(E >= 1 && E <= 12, "bound too large");
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = q29::SUBK[E - 1][i] - b.v[i];
    normalise(r);
    return r;
}

// K*p - 2b, K = 2^e >= 4 * bound(b)
template <int B>
HD Fq<pow2_at_least(4 * B)> neg2(const Fq<B>& b) {
    constexpr int K = pow2_at_least(4 * B);
    constexpr int E = log2_exact(K);
    static_assert(E >= 1 && E <= 12, "bound too large");
    Fq<K> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = q29::SUBK[E - 1][i] - (b.v[i] << 1);
    normalise(r);
    return r;
}

// ---- fused forms: several additive steps, ONE carry sweep (limbs have 3 spare bits: lazy sums stay below 8 * 2^29) ----
// 4a
template <int A>
HD Fq<4 * A> dbl2(const Fq<A>& a) {
    Fq<4 * A> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] << 2;
    normalise(r);
    return r;
}
// a - b - 2c   (+ multiples of p)
template <int A, int B, int C>
HD Fq<A + pow2_at_least(2 * B) + pow2_at_least(4 * C)> sub_sub2(const Fq<A>& a, const Fq<B>& b, const Fq<C>& c) {
    constexpr int K1 = pow2_at_least(2 * B), K2 = pow2_at_least(4 * C);
    constexpr int E1 = log2_exact(K1), E2 = log2_exact(K2);
    static_assert(E1 >= 1 && E1 <= 12 && E2 >= 1 && E2 <= 12, "subtrahend bound too large");
    Fq<A + K1 + K2> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] + q29::SUBK[E1 - 1][i] - b.v[i] + q29::SUBK[E2 - 1][i] - (c.v[i] << 1);
    normalise(r);
    return r;
}
// a - 2c
template <int A, int C>
HD Fq<A + pow2_at_least(4 * C)> sub2(const Fq<A>& a, const Fq<C>& c) {
    constexpr int K = pow2_at_least(4 * C);
    constexpr int E = log2_exact(K);
    static_assert(E >= 1 && E <= 12, "subtrahend bound too large");
    Fq<A + K> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = a.v[i] + q29::SUBK[E - 1][i] - (c.v[i] << 1);
    normalise(r);
    return r;
}
// (negate ? -a : a) - b   (+ multiples of p); `negate` may be wave-uniform or per-lane
template <int A, int B>
HD Fq<pow2_at_least(2 * A) + pow2_at_least(2 * B)> signed_sub(bool negate, const Fq<A>& a, const Fq<B>& b) {
    constexpr int KA = pow2_at_least(2 * A), KB = pow2_at_least(2 * B);
    constexpr int EA = log2_exact(KA), EB = log2_exact(KB);
    static_assert(EA >= 1 && EA <= 12 && EB >= 1 && EB <= 12, "bound too large");
    Fq<KA + KB> r;
#pragma unroll
    for (int i = 0; i < QL; i++) {
        uint32_t t = negate ? q29::SUBK[EA - 1][i] - a.v[i] : a.v[i];
        r.v[i] = t + q29::SUBK[EB - 1][i] - b.v[i];
    }
    normalise(r);
    return r;
}
// ---- device form with verbatim multiply-add chains (fp29_mac.hpp), enabled per translation unit with -DFQ_ASM_MAC ----
#if defined(__HIP_DEVICE_COMPILE__) && defined(FQ_ASM_MAC)
namespace q29asm {
template <int K>  // low half: column K < QL computes m[K]
__device__ __forceinline__ void mul_lo(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m) {
    MacRun<K + 1>::vv(acc, a, b + K);
    if constexpr (K > 0) MacRun<K>::template vp<K>(acc, m);
    m[K] = ((uint32_t)acc * q29::N0) & QMASK;
    MacRun<1>::template vp<0>(acc, m + K);
    acc >>= 29;
}
template <int K>  // high half: column K in [QL, 2 QL) emits limb K - QL
__device__ __forceinline__ void mul_hi(uint64_t& acc, const uint32_t* a, const uint32_t* b, const uint32_t* m, uint32_t* r) {
    constexpr int lo = K - QL + 1, n = QL - lo;
    if constexpr (n > 0) {
        MacRun<n>::vv(acc, a + lo, b + (K - lo));
        MacRun<n>::template vp<K - lo>(acc, m + lo);
    }
    r[K - QL] = (uint32_t)acc & QMASK;
    acc >>= 29;
}
template <int K>
__device__ __forceinline__ void sqr_lo(uint64_t& acc, const uint32_t* a, const uint32_t* a2, uint32_t* m) {
    constexpr int n = (K + 1) / 2;  // cross terms i < K - i
    if constexpr (n > 0) MacRun<n>::vv(acc, a2, a + K);
    if constexpr ((K & 1) == 0) MacRun<1>::vv(acc, a + K / 2, a + K / 2);
    if constexpr (K > 0) MacRun<K>::template vp<K>(acc, m);
    m[K] = ((uint32_t)acc * q29::N0) & QMASK;
    MacRun<1>::template vp<0>(acc, m + K);
    acc >>= 29;
}
template <int K>
__device__ __forceinline__ void sqr_hi(uint64_t& acc, const uint32_t* a, const uint32_t* a2, const uint32_t* m, uint32_t* r) {
    constexpr int lo = K - QL + 1, n = (K + 1) / 2 - lo, nm = QL - lo;
    if constexpr (n > 0) MacRun<n>::vv(acc, a2 + lo, a + (K - lo));
    if constexpr ((K & 1) == 0 && K / 2 < QL) MacRun<1>::vv(acc, a + K / 2, a + K / 2);
    if constexpr (nm > 0) MacRun<nm>::template vp<K - lo>(acc, m + lo);
    r[K - QL] = (uint32_t)acc & QMASK;
    acc >>= 29;
}
template <int K>
__device__ __forceinline__ void mul2_lo(uint64_t& acc, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* m) {
    MacRun<K + 1>::vv(acc, a, b + K);
    MacRun<K + 1>::vv(acc, c, d + K);
    if constexpr (K > 0) MacRun<K>::template vp<K>(acc, m);
    m[K] = ((uint32_t)acc * q29::N0) & QMASK;
    MacRun<1>::template vp<0>(acc, m + K);
    acc >>= 29;
}
template <int K>
__device__ __forceinline__ void mul2_hi(uint64_t& acc, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, const uint32_t* m,
                                        uint32_t* r) {
    constexpr int lo = K - QL + 1, n = QL - lo;
    if constexpr (n > 0) {
        MacRun<n>::vv(acc, a + lo, b + (K - lo));
        MacRun<n>::vv(acc, c + lo, d + (K - lo));
        MacRun<n>::template vp<K - lo>(acc, m + lo);
    }
    r[K - QL] = (uint32_t)acc & QMASK;
    acc >>= 29;
}
template <int... Ks>
__device__ __forceinline__ void mul2_all(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* r,
                                         std::integer_sequence<int, Ks...>) {
    uint32_t m[QL];
    uint64_t acc = 0;
    (mul2_lo<Ks>(acc, a, b, c, d, m), ...);
    (mul2_hi<QL + Ks>(acc, a, b, c, d, m, r), ...);
}
template <int... Ks>
__device__ __forceinline__ void mul_all(const uint32_t* a, const uint32_t* b, uint32_t* r, std::integer_sequence<int, Ks...>) {
    uint32_t m[QL];
    uint64_t acc = 0;
    (mul_lo<Ks>(acc, a, b, m), ...);
    (mul_hi<QL + Ks>(acc, a, b, m, r), ...);
}
template <int... Ks>
__device__ __forceinline__ void sqr_all(const uint32_t* a, uint32_t* r, std::integer_sequence<int, Ks...>) {
    uint32_t m[QL], a2[QL];
#pragma unroll
    for (int i = 0; i < QL; i++) a2[i] = a[i] << 1;
    uint64_t acc = 0;
    (sqr_lo<Ks>(acc, a, a2, m), ...);
    (sqr_hi<QL + Ks>(acc, a, a2, m, r), ...);
}
}  // namespace q29asm
#endif

// Montgomery product scanning, single 64-bit accumulator per column.
template <int A, int B>
HD Fq<2> mul(const Fq<A>& a, const Fq<B>& b) {
    static_assert((long)A * B <= (1L << 24), "mul: operand bounds too large (result would exceed 2p)");
#if defined(__HIP_DEVICE_COMPILE__) && defined(FQ_ASM_MAC)
    Fq<2> ra;
    q29asm::mul_all(a.v, b.v, ra.v, std::make_integer_sequence<int, QL>{});
    return ra;
#endif
    uint32_t m[QL];
    Fq<2> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < QL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        m[k] = ((uint32_t)acc * q29::N0) & QMASK;
        acc += (uint64_t)m[k] * q29::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = QL; k < 2 * QL; k++) {
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        r.v[k - QL] = (uint32_t)acc & QMASK;
        acc >>= 29;
    }
    return r;
}
// squaring: cross products once with a doubled operand (2*a_i < 2^30 keeps every column below 2^64)
template <int A>
HD Fq<2> sqr(const Fq<A>& a) {
    static_assert((long)A * A <= (1L << 24), "sqr: operand bound too large");
#if defined(__HIP_DEVICE_COMPILE__) && defined(FQ_ASM_MAC)
    Fq<2> ra;
    q29asm::sqr_all(a.v, ra.v, std::make_integer_sequence<int, QL>{});
    return ra;
#endif
    uint32_t m[QL], a2[QL];
#pragma unroll
    for (int i = 0; i < QL; i++) a2[i] = a.v[i] << 1;
    Fq<2> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < QL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        m[k] = ((uint32_t)acc * q29::N0) & QMASK;
        acc += (uint64_t)m[k] * q29::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = QL; k < 2 * QL; k++) {
#pragma unroll
        for (int i = k - QL + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        r.v[k - QL] = (uint32_t)acc & QMASK;
        acc >>= 29;
    }
    return r;
}

// a*b + c*d with ONE Montgomery reduction: the two products share the columns (42 terms < 2^58 each still fit 64 bits),
// which saves the 196 m*p multiply-adds and the column bookkeeping of a second multiplication.
template <int A, int B, int C, int D>
HD Fq<2> mul_add(const Fq<A>& a, const Fq<B>& b, const Fq<C>& c, const Fq<D>& d) {
    static_assert((long)A * B + (long)C * D <= (1L << 24), "mul_add: operand bounds too large (result would exceed 2p)");
#if defined(__HIP_DEVICE_COMPILE__) && defined(FQ_ASM_MAC)
    Fq<2> ra;
    q29asm::mul2_all(a.v, b.v, c.v, d.v, ra.v, std::make_integer_sequence<int, QL>{});
    return ra;
#endif
    uint32_t m[QL];
    Fq<2> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < QL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        m[k] = ((uint32_t)acc * q29::N0) & QMASK;
        acc += (uint64_t)m[k] * q29::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = QL; k < 2 * QL; k++) {
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = k - QL + 1; i < QL; i++) acc += (uint64_t)m[i] * q29::P[k - i];
        r.v[k - QL] = (uint32_t)acc & QMASK;
        acc >>= 29;
    }
    return r;
}

// value == 0 mod p ?   (value < B*p with normalised limbs => it is one of 0, p, ..., (B-1)p)
template <int B>
HD bool is_zero(const Fq<B>& a) {
    static_assert(B <= 8, "is_zero: reduce first");
    bool z = false;
#pragma unroll
    for (int mlt = 0; mlt < B; mlt++) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < QL; i++) d |= a.v[i] ^ q29::MULP[mlt][i];
        z |= (d == 0);
    }
    return z;
}
// canonical representative in [0, p)
template <int B>
HD Fq<1> canonical(const Fq<B>& a) {
    Fq<1> one;
#pragma unroll
    for (int i = 0; i < QL; i++) one.v[i] = q29::ONE[i];
    Fq<2> t = mul(a, one);  // a * R * R^-1 = a mod p, < 2p
    // t - p if t >= p
    uint32_t d[QL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) {
        uint32_t x = t.v[i] - q29::P[i] - borrow;
        borrow = x >> 31;  // limbs < 2^30: a wrap-around shows in bit 31
        d[i] = x & QMASK;
    }
    // top limb carries no mask issue: values < 2p keep limb 13 tiny
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = borrow ? t.v[i] : d[i];
    return r;
}
template <int B>
HD bool is_zero_slow(const Fq<B>& a) {
    Fq<1> c = canonical(a);
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) d |= c.v[i];
    return d == 0;
}

HD Fq<1> fq_one() {
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = q29::ONE[i];
    return r;
}
HD Fq<1> fq_zero() {
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = 0;
    return r;
}

// ---- conversions to / from the saturated Montgomery-384 form (field.hpp) -------------------------------------
// regroup 12 x 32 bits (value < p) into 14 x 29 bits
HD void regroup_32_to_29(uint32_t* out, const uint32_t* in) {
#pragma unroll
    for (int i = 0; i < QL; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint64_t two = w < 12 ? in[w] : 0u;
        if (w + 1 < 12) two |= (uint64_t)in[w + 1] << 32;
        out[i] = (uint32_t)(two >> s) & QMASK;
    }
}
HD void regroup_29_to_32(uint32_t* out, const uint32_t* in) {  // value < 2^384
#pragma unroll
    for (int w = 0; w < 12; w++) {
        // bits [32w, 32w+32): from limbs floor(32w/29) ..
        const int lo = (32 * w) / 29, sh = 32 * w - 29 * lo;
        uint64_t acc = (uint64_t)in[lo] >> sh;
        int have = 29 - sh;
        if (lo + 1 < QL) { acc |= (uint64_t)in[lo + 1] << have; have += 29; }
        if (have < 32 && lo + 2 < QL) acc |= (uint64_t)in[lo + 2] << have;
        out[w] = (uint32_t)acc;
    }
}
// Fp (Montgomery-384, canonical) -> Fq<1> (Montgomery-406, canonical)
HD Fq<1> fq_from_fp(const Fp& a) {
    Fq<1> t, c;
    regroup_32_to_29(t.v, a.v);
#pragma unroll
    for (int i = 0; i < QL; i++) c.v[i] = q29::C_FROM_FP[i];
    Fq<2> m = mul(t, c);  // a * 2^428 * 2^-406 = a * 2^22
    // canonicalise: m < 2p
    uint32_t d[QL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) {
        uint32_t x = m.v[i] - q29::P[i] - borrow;
        borrow = x >> 31;
        d[i] = x & QMASK;
    }
    Fq<1> r;
#pragma unroll
    for (int i = 0; i < QL; i++) r.v[i] = borrow ? m.v[i] : d[i];
    return r;
}
// Fq<B> (Montgomery-406) -> Fp (Montgomery-384, canonical)
template <int B>
HD Fp fp_from_fq(const Fq<B>& a) {
    Fq<1> c;
#pragma unroll
    for (int i = 0; i < QL; i++) c.v[i] = q29::C_TO_FP[i];
    Fq<2> m = mul(a, c);  // a * 2^384 * 2^-406 = a * 2^-22
    uint32_t d[QL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) {
        uint32_t x = m.v[i] - q29::P[i] - borrow;
        borrow = x >> 31;
        d[i] = x & QMASK;
    }
#pragma unroll
    for (int i = 0; i < QL; i++) d[i] = borrow ? m.v[i] : d[i];
    Fp r;
    regroup_29_to_32(r.v, d);
    return r;
}

}  // namespace kzg
