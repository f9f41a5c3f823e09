// Unsaturated Fr for the NTT kernels (k_ntt.hip): 9 limbs of 29 bits in u32 registers, Montgomery radix 2^261.
// Why (two-class cost model): the saturated 8 x 32-bit multiplication needs 128 v_mad_u64_u32 AND 128
// carry folds (both three-operand class: ~4.4 cycles each); with 29-bit limbs a whole product-scanning column (9 a*b + 9 m*r
// products < 2^58 each) fits one 64-bit accumulator: 162 multiply-adds and no carry instructions.  r = 1 mod 2^32, so the
// Montgomery factor -r^-1 mod 2^29 is 2^29 - 1 (m = -t mod 2^29: a negation, not a multiplication) and r's limb 0 is 1
// (m * r_0 is an addition).
//
// Values live in [0, B * r) with a small bound B tracked BY THE CALLER (comments at every use): 2^261 ~ 70.6 r, a
// multiplication needs bound(a) * bound(b) <= 64 and returns a value < 2r with normalised limbs.  The NTT networks are
// all Cooley-Tukey (multiply, then add / subtract), so bounds grow by 2 per layer and no reduction is needed inside a
// transform: 12 layers from a value < 32 r end below 56 r.
//
// Relation to the saturated Montgomery form (radix 2^256) the rest of the engine stores: X29 = 32 * Y32 mod r.  So a stored
// element enters as (Y << 5) -- a re-grouping of bits, value < 32 r -- and leaves through the final multiplication every
// path has anyway (by n^-1, by 1 for canonical output, by 128^-1 ...), with the constant chosen for the wanted form.
#pragma once
#include "field.hpp"
#include "fr29_consts.hpp"

namespace kzg {

constexpr int RL = 9;
constexpr uint32_t RMASK = (1u << 29) - 1;

struct Fr29 {
    uint32_t v[RL];
};

// carry sweep: limbs 0..7 back below 2^29 (limb 8 takes what is left; the value stays below 2^261)
HD void fr29_normalise(Fr29& a) {
#pragma unroll
    for (int i = 0; i < RL - 1; i++) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= RMASK;
    }
}
// a + b (bounds add).  NORM = false leaves the carries where they are: every second layer of a transform does (LAZY LIMBS below).
template <bool NORM = true>
HD Fr29 fr29_add(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < RL; i++) r.v[i] = a.v[i] + b.v[i];
    if (NORM) fr29_normalise(r);
    return r;
}
// a - b + 2r for b < 2r with normalised limbs (every subtrahend in the transforms is a fresh product): bound(a) + 2
template <bool NORM = true>
HD Fr29 fr29_sub2r(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < RL; i++) r.v[i] = a.v[i] + r29::SUBK[0][i] - b.v[i];
    if (NORM) fr29_normalise(r);
    return r;
}
// LAZY LIMBS (round 5): the carry sweep of a sum or difference is 16 of a butterfly's ~310 instructions, twice per butterfly, and
// only every SECOND layer needs it.  With a, t normalised (limbs < 2^29; SUBK's limbs are in [2^30 - 2, 1.5 * 2^30)): a + t < 2^30
// and a + 2r - t < 2^31 per limb, stored as they are.  The next layer reads such values as a (then a + t < 2^31 + 2^29 and
// a + 2r - t < 3.5 * 2^30 < 2^32: no wrap, swept before the store) or as the product's first operand b: a column is
// 9 * 2^31 * 2^29 + 9 * 2^58 = 45 * 2^58 < 2^64, and fr29_partial_reduce carries in 64 bits whatever the limbs are.  The VALUE
// bounds (B r) are untouched.  tests/c/test_fr29.cpp runs the 4096-point network both ways and measures the limbs.
//
// Montgomery product a * b / 2^261 mod r, product scanning; bound(a) * bound(b) <= 64 -> result < 2r, limbs normalised.
// ONE of the operands may carry un-normalised limbs up to 2^31 (9 * 2^60 + 9 * 2^58 = 45 * 2^58 < 2^64), the other < 2^29.
HD Fr29 fr29_mul(const Fr29& a, const Fr29& b) {
    uint32_t m[RL];
    Fr29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < RL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * r29::P[k - i];
        m[k] = (0u - (uint32_t)acc) & RMASK;  // -t mod 2^29
        acc += m[k];                          // m * r_0, r_0 = 1: the low 29 bits cancel
        acc >>= 29;
    }
#pragma unroll
    for (int k = RL; k < 2 * RL - 1; k++) {
#pragma unroll
        for (int i = k - RL + 1; i < RL; i++) acc += (uint64_t)a.v[i] * b.v[k - i] + (uint64_t)m[i] * r29::P[k - i];
        r.v[k - RL] = (uint32_t)acc & RMASK;
        acc >>= 29;
    }
    r.v[RL - 1] = (uint32_t)acc;
    return r;
}
// value < 2^261 (any bound) -> the same residue below 2r, normalised limbs, WITHOUT a multiplication: the quotient is
// estimated from the top limb, q = (top * 565) >> 32 <= value / r (565 / 2^264 < 1 / r), and q r is subtracted with one
// running signed accumulator; the remainder stays below 1.04 r (checked over the whole range in tests/c/test_fr29.cpp).
// Used where a Cooley-Tukey butterfly's twiddle is 1: ~40 instructions instead of a 190-instruction product by one.
HD Fr29 fr29_partial_reduce(const Fr29& b) {
    const uint32_t q = (uint32_t)(((uint64_t)b.v[RL - 1] * 565u) >> 32);
    Fr29 r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < RL - 1; i++) {
        acc += (int64_t)b.v[i] - (int64_t)((uint64_t)q * r29::P[i]);
        r.v[i] = (uint32_t)acc & RMASK;
        acc >>= 29;  // arithmetic: carries the borrow
    }
    acc += (int64_t)b.v[RL - 1] - (int64_t)((uint64_t)q * r29::P[RL - 1]);
    r.v[RL - 1] = (uint32_t)acc;
    return r;
}
HD Fr29 fr29_const(const uint32_t (&c)[RL]) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < RL; i++) r.v[i] = c[i];
    return r;
}
// value < 2r, normalised limbs -> canonical (< r)
HD Fr29 fr29_reduce_once(const Fr29& t) {
    uint32_t d[RL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < RL; i++) {
        const uint32_t x = t.v[i] - r29::P[i] - borrow;
        borrow = x >> 31;
        d[i] = i < RL - 1 ? (x & RMASK) : x;
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < RL; i++) r.v[i] = borrow ? t.v[i] : d[i];
    return r;
}

// ---- re-grouping between 8 x 32-bit words and 9 x 29-bit limbs -------------------------------------------------------
// limbs of (y << SH) for a 256-bit y (SH = 0: the same integer; SH = 5: the saturated Montgomery form -> this one)
template <int SH>
HD Fr29 fr29_from_words(const uint32_t (&w)[8]) {
    static_assert(SH == 0 || SH == 5, "shift");
    Fr29 r;
#pragma unroll
    for (int i = 0; i < RL; i++) {
        const int lo = 29 * i - SH;  // bit position in y of this limb's bit 0 (negative for the shifted-in zeros)
        uint32_t x;
        if (lo < 0) x = w[0] << (-lo);
        else {
            const int word = lo >> 5, sh = lo & 31;
            x = word < 8 ? w[word] >> sh : 0u;
            if (sh > 3 && word + 1 < 8) x |= w[word + 1] << (32 - sh);
        }
        r.v[i] = i < RL - 1 ? (x & RMASK) : x;  // the top limb keeps every remaining bit (y < 2^256: at most 24 + SH of them)
    }
    return r;
}
// canonical limbs (value < 2^256) -> 8 words
HD void fr29_to_words(uint32_t (&w)[8], const Fr29& a) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int lo = 32 * k, limb = lo / 29, sh = lo % 29;
        uint32_t x = a.v[limb] >> sh;
        if (limb + 1 < RL) x |= a.v[limb + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && limb + 2 < RL) x |= a.v[limb + 2] << (58 - sh);
        w[k] = x;
    }
}
HD Fr29 fr29_from_fr_mont(const Fr& y) { return fr29_from_words<5>(y.v); }  // saturated Montgomery (canonical, < r) -> this form, value < 32 r
HD Fr29 fr29_from_plain(const Fr& y) { return fr29_from_words<0>(y.v); }   // the same integer

}  // namespace kzg
