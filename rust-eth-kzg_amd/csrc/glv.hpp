// GLV decomposition of a scalar on the device: k = k1 + k2 lambda (mod r), lambda = z^2 - 1 the eigenvalue of
// phi(x, y) = (beta x, y) on G1.  Used by the bucket MSMs of the verifier (k_verify.hip: unsigned halves) and by the GLV
// window tables of the prover (k_msm.hip: balanced signed halves, |k1|, |k2| < 2^127, so that eight 16-bit Booth windows
// cover a half without a carry out of the top window).
#pragma once
#include "field.hpp"

namespace kzg {

// r = k mod lambda (< lambda < 2^128), q = floor(k / lambda) (<= lambda + 1): one multiplication by floor(2^256 / lambda) and
// one correction.  k canonical (not Montgomery), < 2^255.
HD void glv_split_unsigned(const Fr& k, uint32_t r4[4], uint32_t q[4]) {
    constexpr uint32_t G[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x1u};  // floor(2^256 / lambda)
    constexpr uint32_t L[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};        // lambda
    {   // q = (k * G) >> 256 : column sums of the 8 x 5 product, keeping limbs 8..11 (q < 2^128)
        uint64_t carry = 0;
        for (int col = 0; col < 12; col++) {
            uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;
            for (int a = 0; a < 8; a++) {
                const int b = col - a;
                if (b < 0 || b > 4) continue;
                const uint64_t pr = (uint64_t)k.v[a] * G[b];
                lo += pr & 0xffffffffu;
                hi += pr >> 32;
            }
            hi += lo >> 32;
            if (col >= 8) q[col - 8] = (uint32_t)lo;
            carry = hi;
        }
    }
    uint32_t ql[5] = {0, 0, 0, 0, 0};  // q * lambda, 5 limbs
    {
        uint64_t carry = 0;
        for (int col = 0; col < 5; col++) {
            uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;
            for (int a = 0; a < 4; a++) {
                const int b = col - a;
                if (b < 0 || b > 3) continue;
                const uint64_t pr = (uint64_t)q[a] * L[b];
                lo += pr & 0xffffffffu;
                hi += pr >> 32;
            }
            hi += lo >> 32;
            ql[col] = (uint32_t)lo;
            carry = hi;
        }
    }
    uint32_t r[5];
    {
        uint32_t br = 0;
        for (int l = 0; l < 5; l++) {
            const uint64_t d = (uint64_t)k.v[l] - ql[l] - br;
            r[l] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
    }
    {   // if r >= lambda: r -= lambda, q += 1
        uint32_t t[5], br = 0;
        for (int l = 0; l < 5; l++) {
            const uint64_t d = (uint64_t)r[l] - (l < 4 ? L[l] : 0u) - br;
            t[l] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
        if (!br) {
            for (int l = 0; l < 5; l++) r[l] = t[l];
            uint32_t c = 1;
            for (int l = 0; l < 4; l++) { const uint64_t sum = (uint64_t)q[l] + c; q[l] = (uint32_t)sum; c = (uint32_t)(sum >> 32); }
        }
    }
    for (int l = 0; l < 4; l++) r4[l] = r[l];
}

// 128-bit helpers on 4 x u32
HD bool gt128(const uint32_t a[4], const uint32_t b[4]) {
    for (int l = 3; l >= 0; l--) {
        if (a[l] != b[l]) return a[l] > b[l];
    }
    return false;
}
HD void sub128(uint32_t out[4], const uint32_t a[4], const uint32_t b[4]) {
    uint32_t br = 0;
    for (int l = 0; l < 4; l++) {
        const uint64_t d = (uint64_t)a[l] - b[l] - br;
        out[l] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
}
HD void add128_small(uint32_t a[4], uint32_t c) {
    for (int l = 0; l < 4; l++) { const uint64_t s = (uint64_t)a[l] + c; a[l] = (uint32_t)s; c = (uint32_t)(s >> 32); }
}
HD bool is_zero128(const uint32_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

// Balanced form: k = s1 m1 + s2 m2 lambda with magnitudes m1, m2 <= (lambda + 1) / 2 + 1 < 2^127.  Output: 2 x 4 words,
// the sign in bit 127 of each half.  Steps (each keeps k1 + k2 lambda fixed mod r, using lambda^2 + lambda + 1 = r):
//   r > (lambda - 1) / 2   ->  k1 = r - lambda, k2 = q + 1
//   k2 > (lambda + 1) / 2  ->  k2 -= lambda + 1, k1 -= 1
HD void glv_split_balanced(const Fr& k, uint32_t out[8]) {
    constexpr uint32_t L[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};         // lambda
    constexpr uint32_t HL[4] = {0x7fffffffu, 0x00000000u, 0x8000d201u, 0x5622d200u};        // (lambda - 1) / 2
    constexpr uint32_t L1[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};        // lambda + 1
    constexpr uint32_t HL1[4] = {0x80000000u, 0x00000000u, 0x8000d201u, 0x5622d200u};       // (lambda + 1) / 2
    uint32_t r[4], q[4];
    glv_split_unsigned(k, r, q);
    bool neg1 = false, neg2 = false;
    if (gt128(r, HL)) {
        uint32_t t[4];
        sub128(t, L, r);
        for (int l = 0; l < 4; l++) r[l] = t[l];
        neg1 = true;
        add128_small(q, 1);
    }
    if (gt128(q, HL1)) {
        uint32_t t[4];
        sub128(t, L1, q);
        for (int l = 0; l < 4; l++) q[l] = t[l];
        neg2 = true;
        // k1 -= 1
        if (neg1) add128_small(r, 1);
        else if (is_zero128(r)) { r[0] = 1; neg1 = true; }
        else { const uint32_t one[4] = {1, 0, 0, 0}; uint32_t t2[4]; sub128(t2, r, one); for (int l = 0; l < 4; l++) r[l] = t2[l]; }
    }
    if (is_zero128(r)) neg1 = false;
    if (is_zero128(q)) neg2 = false;
    for (int l = 0; l < 4; l++) { out[l] = r[l]; out[4 + l] = q[l]; }
    out[3] |= neg1 ? 0x80000000u : 0u;
    out[7] |= neg2 ? 0x80000000u : 0u;
}

}  // namespace kzg
