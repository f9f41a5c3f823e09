// BLS12-381 G1 (y^2 = x^3 + 4 over Fp) group law for gfx950: one point per lane.
// Jacobian accumulators, affine table entries, exact handling of every exceptional case
// (identity operands, P + P, P + (-P)): the reference's fixture blobs (all-zero, constant,
// two-valued) drive those paths for real (SURVEY.md section 7 "hard parts").
//
// Replaces blstrs::G1Projective / G1Affine as used by the reference
// (crates/cryptography/polynomial/src/fft.rs:31-35,164-177;
//  crates/cryptography/bls12_381/src/fixed_base_msm_window.rs:125-165;
//  crates/cryptography/bls12_381/src/lib.rs:56-104).
#pragma once
#include "field.hpp"
#include "inverse.hpp"

namespace kzg {

struct G1Affine {  // identity is encoded as (0, 0), which is not on the curve
    Fp x, y;
};
struct G1Jac {  // identity <=> z == 0
    Fp x, y, z;
};

HD bool is_inf(const G1Affine& p) { return is_zero(p.x) && is_zero(p.y); }
HD bool is_inf(const G1Jac& p) { return is_zero(p.z); }
HD G1Jac jac_inf() {
    G1Jac r;
    r.x = one<FpParams>();
    r.y = one<FpParams>();
    r.z = zero<FpParams>();
    return r;
}
HD G1Affine aff_inf() {
    G1Affine r;
    r.x = zero<FpParams>();
    r.y = zero<FpParams>();
    return r;
}
HD G1Jac to_jac(const G1Affine& a) {
    if (is_inf(a)) return jac_inf();
    G1Jac r;
    r.x = a.x;
    r.y = a.y;
    r.z = one<FpParams>();
    return r;
}
HD G1Jac neg(const G1Jac& p) {
    G1Jac r = p;
    r.y = neg(p.y);
    return r;
}
HD G1Affine neg(const G1Affine& p) {
    G1Affine r = p;
    r.y = neg(p.y);
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S
HD G1Jac dbl(const G1Jac& p) {
    // identity: z = 0 gives z3 = 0, stays identity
    Fp A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    Fp t = sqr(add(p.x, B));
    t = sub(sub(t, A), C);
    Fp D = dbl(t);
    Fp E = add(dbl(A), A);
    Fp F = sqr(E);
    G1Jac r;
    r.x = sub(F, dbl(D));
    Fp C8 = dbl(dbl(dbl(C)));
    r.y = sub(mul(E, sub(D, r.x)), C8);
    r.z = dbl(mul(p.y, p.z));
    return r;
}

// add-2007-bl, complete via explicit branches
HD G1Jac add(const G1Jac& p, const G1Jac& q) {
    if (is_inf(p)) return q;
    if (is_inf(q)) return p;
    Fp z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fp u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    Fp s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    Fp h = sub(u2, u1), rr = sub(s2, s1);
    if (is_zero(h)) {
        if (is_zero(rr)) return dbl(p);
        return jac_inf();
    }
    rr = dbl(rr);
    Fp i = sqr(dbl(h));
    Fp j = mul(h, i);
    Fp v = mul(u1, i);
    G1Jac r;
    r.x = sub(sub(sqr(rr), j), dbl(v));
    r.y = sub(mul(rr, sub(v, r.x)), dbl(mul(s1, j)));
    r.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return r;
}

// madd-2007-bl (q affine, Z2 = 1): 7M + 4S, complete via explicit branches
HD G1Jac add_mixed(const G1Jac& p, const G1Affine& q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return to_jac(q);
    Fp z1z1 = sqr(p.z);
    Fp u2 = mul(q.x, z1z1);
    Fp s2 = mul(mul(q.y, p.z), z1z1);
    Fp h = sub(u2, p.x), rr = sub(s2, p.y);
    if (is_zero(h)) {
        if (is_zero(rr)) return dbl(p);
        return jac_inf();
    }
    Fp hh = sqr(h);
    Fp i = dbl(dbl(hh));
    Fp j = mul(h, i);
    rr = dbl(rr);
    Fp v = mul(p.x, i);
    G1Jac r;
    r.x = sub(sub(sqr(rr), j), dbl(v));
    r.y = sub(mul(rr, sub(v, r.x)), dbl(mul(p.y, j)));
    r.z = sub(sub(sqr(add(p.z, h)), z1z1), hh);
    return r;
}

HD bool eq(const G1Jac& p, const G1Jac& q) {
    bool pi = is_inf(p), qi = is_inf(q);
    if (pi || qi) return pi && qi;
    Fp z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    if (!eq(mul(p.x, z2z2), mul(q.x, z1z1))) return false;
    return eq(mul(mul(p.y, q.z), z2z2), mul(mul(q.y, p.z), z1z1));
}

HD G1Affine to_affine(const G1Jac& p) {
    if (is_inf(p)) return aff_inf();
    Fp zi = inv_fast(p.z), zi2 = sqr(zi);  // binary-GCD inversion (inverse.hpp)
    G1Affine r;
    r.x = mul(p.x, zi2);
    r.y = mul(p.y, mul(zi2, zi));
    return r;
}

// k * P with a canonical (non-Montgomery) little-endian scalar of NL u32 limbs; plain
// double-and-add, MSB first.  Used for one-time setup and as the slow reference path.
template <int NL>
HD G1Jac scalar_mul(const G1Jac& p, const uint32_t* k) {
    G1Jac acc = jac_inf();
    for (int i = 32 * NL - 1; i >= 0; i--) {
        acc = dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = add(acc, p);
    }
    return acc;
}

// canonical y > (p-1)/2 ?  (the "lexicographically largest" flag of the ZCash encoding)
HD bool fp_is_lex_largest(const Fp& y_mont) {
    constexpr uint32_t HALF[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                   0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    Fp y = from_mont(y_mont);
    uint32_t t[12];
    // y > HALF  <=>  HALF - y borrows
    return sub_limbs<12>(t, HALF, y.v) != 0;
}

// 48-byte ZCash compressed encoding (crates/serialization/src/lib.rs:84-86)
HD void g1_compress(uint8_t* out, const G1Affine& a) {
    if (is_inf(a)) {
        for (int i = 0; i < 48; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    Fp x = from_mont(a.x);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t w = x.v[11 - i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
    out[0] |= 0x80;
    if (fp_is_lex_largest(a.y)) out[0] |= 0x20;
}

// y = sqrt(x^3 + 4) via a^((p+1)/4); returns false if not a square
HD bool fp_sqrt(Fp& out, const Fp& a) {
    constexpr uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    Fp r = pow_fixed<FpParams, 12>(a, E);
    out = r;
    return eq(sqr(r), a);
}

// Decode 48 bytes; rc 0 ok, 1 bad encoding / x >= p / not on curve.  No subgroup check here.
HD int g1_decompress(G1Affine& out, const uint8_t* in) {
    uint8_t b0 = in[0];
    bool compressed = (b0 >> 7) & 1, infinity = (b0 >> 6) & 1, sign = (b0 >> 5) & 1;
    if (!compressed) return 1;
    Fp x;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t w = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | in[4 * i + 3];
        if (i == 0) w &= 0x1fffffffu;
        x.v[11 - i] = w;
    }
    if (infinity) {
        if (sign || !is_zero(x)) return 1;
        out = aff_inf();
        return 0;
    }
    if (geq_mod<FpParams>(x.v)) return 1;
    x = to_mont(x);
    Fp four = zero<FpParams>();
    four.v[0] = 4;
    four = to_mont(four);
    Fp y2 = add(mul(sqr(x), x), four), y;
    if (!fp_sqrt(y, y2)) return 1;
    if (fp_is_lex_largest(y) != sign) y = neg(y);
    out.x = x;
    out.y = y;
    return 0;
}

// [r]P == O (definitional subgroup membership; P affine, not identity => caller treats identity as member)
HD bool g1_in_subgroup(const G1Affine& a) {
    if (is_inf(a)) return true;
    G1Jac r = scalar_mul<8>(to_jac(a), FrParams::MOD);
    return is_inf(r);
}

// Endomorphism subgroup test (the method blst uses; M. Scott, "A note on group membership tests for G1, G2 and GT
// on BLS pairing-friendly curves", 2021): with phi(x, y) = (beta x, y) acting on G1 as [lambda], lambda = z^2 - 1,
// a curve point P lies in G1 iff [z^2]P - P == phi(P).  [z^2]P = [|z|]([|z|]P): 126 doublings + 10 additions
// (|z| = 0xd201000000010000 has weight 6) instead of the 255-bit [r]P.  The definitional test stays available
// (g1_in_subgroup) and the GPU tests check that both agree on points outside the subgroup.
HD G1Jac mul_by_z_abs(const G1Jac& p) {
    constexpr uint64_t Z = 0xd201000000010000ULL;
    G1Jac acc = p;
    for (int i = 62; i >= 0; i--) {
        acc = dbl(acc);
        if ((Z >> i) & 1) acc = add(acc, p);
    }
    return acc;
}
HD bool g1_in_subgroup_endo(const G1Affine& a, const Fp& beta) {
    if (is_inf(a)) return true;
    G1Jac p = to_jac(a);
    G1Jac q = mul_by_z_abs(mul_by_z_abs(p));  // [z^2]P
    G1Jac r = add(q, neg(p));                 // [z^2 - 1]P
    if (is_inf(r)) return false;
    Fp zz = sqr(r.z);
    if (!eq(r.x, mul(mul(a.x, beta), zz))) return false;
    return eq(r.y, mul(a.y, mul(zz, r.z)));
}

}  // namespace kzg
