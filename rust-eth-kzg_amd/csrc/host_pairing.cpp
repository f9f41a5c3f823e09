// See host_pairing.hpp.  Tower Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v);
// M-type twist E'(Fp2): y^2 = x^3 + 4(1+u), untwist (x, y) -> (x / w^2, y / w^3).
#include "host_pairing.hpp"
#include <mutex>

namespace kzg {
namespace pairing {

using FpP = FpParams;
static inline Fp fz() { return zero<FpP>(); }
static inline Fp fo() { return one<FpP>(); }

// ---------------- Fp2 ----------------
static inline Fp2 operator+(const Fp2& a, const Fp2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
static inline Fp2 operator-(const Fp2& a, const Fp2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
static inline Fp2 neg2(const Fp2& a) { return {neg(a.c0), neg(a.c1)}; }
static inline Fp2 conj(const Fp2& a) { return {a.c0, neg(a.c1)}; }
static inline Fp2 operator*(const Fp2& a, const Fp2& b) {  // Karatsuba, 3 Fp mul
    Fp t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    Fp t2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(t0, t1), sub(sub(t2, t0), t1)};
}
static inline Fp2 sqr2(const Fp2& a) {  // (a0+a1)(a0-a1), 2 a0 a1
    Fp t = mul(a.c0, a.c1);
    return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), add(t, t)};
}
static inline Fp2 scale(const Fp2& a, const Fp& s) { return {mul(a.c0, s), mul(a.c1, s)}; }
static inline Fp2 mul_xi(const Fp2& a) { return {sub(a.c0, a.c1), add(a.c0, a.c1)}; }  // * (1+u)
static inline Fp2 inv2(const Fp2& a) {
    Fp n = inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, n), neg(mul(a.c1, n))};
}
static inline bool is_zero2(const Fp2& a) { return is_zero(a.c0) && is_zero(a.c1); }
static inline bool eq2(const Fp2& a, const Fp2& b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
static inline Fp2 zero2() { return {fz(), fz()}; }
static inline Fp2 one2() { return {fo(), fz()}; }
static Fp2 pow2(const Fp2& a, const uint32_t* e, int nl) {
    Fp2 acc = one2();
    for (int i = 32 * nl - 1; i >= 0; i--) {
        acc = sqr2(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * a;
    }
    return acc;
}
static bool sqrt2(Fp2& out, const Fp2& a) {  // p = 3 mod 4 (Adj & Rodriguez-Henriquez, Alg. 9)
    static const uint32_t E1[12] = {0xffffeaaau, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                    0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};  // (p-3)/4
    static const uint32_t E2[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                    0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};  // (p-1)/2
    if (is_zero2(a)) { out = a; return true; }
    Fp2 a1 = pow2(a, E1, 12);
    Fp2 x0 = a1 * a;
    Fp2 alpha = a1 * x0;
    Fp2 a0 = conj(alpha) * alpha;
    Fp2 m1 = {neg(fo()), fz()};
    if (eq2(a0, m1)) return false;
    Fp2 x;
    if (eq2(alpha, m1)) x = {neg(x0.c1), x0.c0};  // u * x0
    else {
        Fp2 t = alpha;
        t.c0 = add(t.c0, fo());
        x = pow2(t, E2, 12) * x0;
    }
    if (!eq2(sqr2(x), a)) return false;
    out = x;
    return true;
}

// ---------------- Fp6 ----------------
static inline Fp6 operator+(const Fp6& a, const Fp6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
static inline Fp6 operator-(const Fp6& a, const Fp6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
static inline Fp6 neg6(const Fp6& a) { return {neg2(a.c0), neg2(a.c1), neg2(a.c2)}; }
static inline Fp6 operator*(const Fp6& a, const Fp6& b) {  // Karatsuba-style, 6 Fp2 mul
    Fp2 v0 = a.c0 * b.c0, v1 = a.c1 * b.c1, v2 = a.c2 * b.c2;
    Fp2 c0 = v0 + mul_xi((a.c1 + a.c2) * (b.c1 + b.c2) - v1 - v2);
    Fp2 c1 = (a.c0 + a.c1) * (b.c0 + b.c1) - v0 - v1 + mul_xi(v2);
    Fp2 c2 = (a.c0 + a.c2) * (b.c0 + b.c2) - v0 - v2 + v1;
    return {c0, c1, c2};
}
static inline Fp6 mul_v(const Fp6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
static Fp6 inv6(const Fp6& a) {
    Fp2 t0 = sqr2(a.c0) - mul_xi(a.c1 * a.c2);
    Fp2 t1 = mul_xi(sqr2(a.c2)) - a.c0 * a.c1;
    Fp2 t2 = sqr2(a.c1) - a.c0 * a.c2;
    Fp2 d = inv2(a.c0 * t0 + mul_xi(a.c2 * t1 + a.c1 * t2));
    return {t0 * d, t1 * d, t2 * d};
}

// ---------------- Fp12 ----------------
static inline Fp12 one12() { return {{one2(), zero2(), zero2()}, {zero2(), zero2(), zero2()}}; }
static inline Fp12 operator*(const Fp12& a, const Fp12& b) {  // 3 Fp6 mul
    Fp6 v0 = a.c0 * b.c0, v1 = a.c1 * b.c1;
    Fp6 c1 = (a.c0 + a.c1) * (b.c0 + b.c1) - v0 - v1;
    return {v0 + mul_v(v1), c1};
}
// a^2 by the complex method: (a0 + a1 w)^2 = (a0 + a1)(a0 + v a1) - t - v t + 2 t w, t = a0 a1: 2 Fp6 multiplications instead of 3
static inline Fp12 sqr12(const Fp12& a) {
    const Fp6 t = a.c0 * a.c1;
    const Fp6 c0 = (a.c0 + a.c1) * (a.c0 + mul_v(a.c1)) - t - mul_v(t);
    return {c0, t + t};
}
// (x0 + x1 v + x2 v^2)(A + B v): 5 Fp2 multiplications
static inline Fp6 mul6_by_01(const Fp6& x, const Fp2& A, const Fp2& B) {
    const Fp2 aa = x.c0 * A, bb = x.c1 * B;
    const Fp2 c0 = mul_xi((x.c1 + x.c2) * B - bb) + aa;
    const Fp2 c2 = (x.c0 + x.c2) * A - aa + bb;
    const Fp2 c1 = (x.c0 + x.c1) * (A + B) - aa - bb;
    return {c0, c1, c2};
}
// (x0 + x1 v + x2 v^2)(y v), y in Fp: (xi x2 y, x0 y, x1 y)
static inline Fp6 mul6_by_1fp(const Fp6& x, const Fp& y) { return {mul_xi(scale(x.c2, y)), scale(x.c0, y), scale(x.c1, y)}; }
// f * (A + B v + y v w): the value of a line at a G1 point has three of its six Fp2 coefficients, one of them in Fp.
// Karatsuba over w with the sparse factors: 5 + 5 Fp2 multiplications + 6 by an Fp, against 18 for a general product.
static inline Fp12 mul12_by_line(const Fp12& f, const Fp2& A, const Fp2& B, const Fp& y) {
    const Fp6 v0 = mul6_by_01(f.c0, A, B), v1 = mul6_by_1fp(f.c1, y);
    const Fp2 By = {add(B.c0, y), B.c1};
    const Fp6 c1 = mul6_by_01(f.c0 + f.c1, A, By) - v0 - v1;
    return {v0 + mul_v(v1), c1};
}
// a^2 for a in the cyclotomic subgroup (a^(p^6+1) = 1, where every value sits after the easy part of the final
// exponentiation): Granger-Scott squaring, three Fp4 squarings = 6 Fp2 multiplications instead of 18.
static inline Fp12 cyc_sqr12(const Fp12& a) {
    const Fp2 &r0 = a.c0.c0, &r4 = a.c0.c1, &r3 = a.c0.c2, &r2 = a.c1.c0, &r1 = a.c1.c1, &r5 = a.c1.c2;
    auto fp4_sqr = [](const Fp2& x, const Fp2& y, Fp2& s0, Fp2& s1) {  // (x + y s)^2, s^2 = xi
        const Fp2 t = x * y;
        s0 = (x + y) * (mul_xi(y) + x) - t - mul_xi(t);
        s1 = t + t;
    };
    Fp2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(r0, r1, t0, t1);
    fp4_sqr(r2, r3, t2, t3);
    fp4_sqr(r4, r5, t4, t5);
    auto three_minus_two = [](const Fp2& t, const Fp2& z) { Fp2 d = t - z; return d + d + t; };  // 3 t - 2 z
    auto three_plus_two = [](const Fp2& t, const Fp2& z) { Fp2 d = t + z; return d + d + t; };    // 3 t + 2 z
    Fp12 r;
    r.c0.c0 = three_minus_two(t0, r0);
    r.c1.c1 = three_plus_two(t1, r1);
    r.c1.c0 = three_plus_two(mul_xi(t5), r2);
    r.c0.c2 = three_minus_two(t4, r3);
    r.c0.c1 = three_minus_two(t2, r4);
    r.c1.c2 = three_plus_two(t3, r5);
    return r;
}
static inline Fp12 conj12(const Fp12& a) { return {a.c0, neg6(a.c1)}; }
static Fp12 inv12(const Fp12& a) {
    Fp6 t = inv6(a.c0 * a.c0 - mul_v(a.c1 * a.c1));
    return {a.c0 * t, neg6(a.c1 * t)};
}
static bool is_one12(const Fp12& a) {
    const Fp12 o = one12();
    const Fp2* x = &a.c0.c0;
    const Fp2* y = &o.c0.c0;
    for (int i = 0; i < 6; i++)
        if (!eq2(x[i], y[i])) return false;
    return true;
}

// Frobenius: v^p = xi^((p-1)/3) v, w^p = xi^((p-1)/6) w
static Fp2 G1C, G2C, G4C;  // xi^((p-1)/6), its square, its fourth power
static void init_once();
void init() {  // contexts are created from any thread, and the engines of a device list side by side (c_api.cpp): once, with a fence
    static std::once_flag once;
    std::call_once(once, init_once);
}
static void init_once() {
    static const uint32_t P16[12] = {0xfffff1c7u, 0x49aa7fffu, 0x72e35555u, 0x051caaaau, 0xd3c82906u, 0xe688231au,
                                     0x7deb831fu, 0xe613e1ebu, 0xb5e1f223u, 0x0c849bf3u, 0x5eeaa66fu, 0x045582fcu};  // (p-1)/6
    Fp2 xi = {fo(), fo()};
    G1C = pow2(xi, P16, 12);
    G2C = sqr2(G1C);
    G4C = sqr2(G2C);
}
static inline Fp6 frob6(const Fp6& a) { return {conj(a.c0), conj(a.c1) * G2C, conj(a.c2) * G4C}; }
static inline Fp12 frob12(const Fp12& a) {
    Fp6 c1 = frob6(a.c1);
    return {frob6(a.c0), {c1.c0 * G1C, c1.c1 * G1C, c1.c2 * G1C}};
}

// ---------------- G2 ----------------
static bool lex_largest2(const Fp2& y) { return is_zero(y.c1) ? fp_is_lex_largest(y.c0) : fp_is_lex_largest(y.c1); }
static bool fp_from_be48(Fp& out, const uint8_t* in, bool mask_flags) {
    Fp x;
    for (int i = 0; i < 12; i++) {
        uint32_t w = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | in[4 * i + 3];
        if (i == 0 && mask_flags) w &= 0x1fffffffu;
        x.v[11 - i] = w;
    }
    if (geq_mod<FpP>(x.v)) return false;
    out = to_mont(x);
    return true;
}
bool g2_decompress(G2Affine& out, const uint8_t in[96]) {
    bool compressed = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sign = (in[0] >> 5) & 1;
    if (!compressed) return false;
    Fp2 x;
    if (!fp_from_be48(x.c1, in, true) || !fp_from_be48(x.c0, in + 48, false)) return false;
    if (infinity) {
        if (sign || !is_zero2(x)) return false;
        out = {zero2(), zero2(), true};
        return true;
    }
    Fp four = zero<FpP>();
    four.v[0] = 4;
    four = to_mont(four);
    Fp2 y2 = sqr2(x) * x + Fp2{four, four}, y;
    if (!sqrt2(y, y2)) return false;
    if (lex_largest2(y) != sign) y = neg2(y);
    out = {x, y, false};
    return true;
}
G2Affine g2_neg(const G2Affine& q) { return {q.x, neg2(q.y), q.inf}; }

// |x| = 0xd201000000010000 (the BLS parameter is negative; see product_is_one)
static const uint64_t X_ABS = 0xd201000000010000ULL;

G2Prepared prepare(const G2Affine& q) {
    G2Prepared pr;
    pr.inf = q.inf;
    if (q.inf) return pr;
    Fp2 tx = q.x, ty = q.y;
    auto step = [&](bool dbl) {
        Fp2 lam;
        if (dbl) {
            Fp2 x2 = sqr2(tx);
            lam = (x2 + x2 + x2) * inv2(ty + ty);
        } else lam = (q.y - ty) * inv2(q.x - tx);
        pr.lines.push_back({lam * tx - ty, neg2(lam)});
        Fp2 x3 = sqr2(lam) - tx - (dbl ? tx : q.x);
        Fp2 y3 = lam * (tx - x3) - ty;
        tx = x3;
        ty = y3;
    };
    for (int i = 62; i >= 0; i--) {
        step(true);
        if ((X_ABS >> i) & 1) step(false);
    }
    return pr;
}

static Fp12 pow_x(const Fp12& f) {  // f^x with x negative: conj(f^|x|) inside the cyclotomic subgroup
    Fp12 acc = f;
    for (int i = 62; i >= 0; i--) {
        acc = cyc_sqr12(acc);
        if ((X_ABS >> i) & 1) acc = acc * f;
    }
    return conj12(acc);
}

static Fp12 miller_loop_n(const G1Affine* P, const G2Prepared* const* Q, int n) {
    init();
    Fp12 f = one12();
    size_t idx = 0;
    auto lines = [&]() {
        for (int k = 0; k < n; k++)
            if (!is_inf(P[k]) && !Q[k]->inf) {
                const Line& l = Q[k]->lines[idx];
                f = mul12_by_line(f, l.a, scale(l.b, P[k].x), P[k].y);
            }
        idx++;
    };
    for (int i = 62; i >= 0; i--) {
        f = sqr12(f);
        lines();
        if ((X_ABS >> i) & 1) lines();
    }
    return conj12(f);  // x < 0
}
Fp12 miller_loop(const G1Affine& P, const G2Prepared& Q) {
    const G2Prepared* q = &Q;
    return miller_loop_n(&P, &q, 1);
}
Fp12 fp12_mul(const Fp12& a, const Fp12& b) { return a * b; }
bool product_is_one(const G1Affine* P, const G2Prepared* const* Q, int n) { return final_exponentiation_is_one(miller_loop_n(P, Q, n)); }
bool final_exponentiation_is_one(const Fp12& f) {
    init();
    // easy part: f^((p^6-1)(p^2+1))
    Fp12 t = conj12(f) * inv12(f);
    t = frob12(frob12(t)) * t;
    // hard part, up to the harmless factor 3: 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3
    Fp12 a = pow_x(t) * conj12(t);            // t^(x-1)
    a = pow_x(a) * conj12(a);                 // t^((x-1)^2)
    Fp12 b = pow_x(a) * frob12(a);            // ^(x+p)
    Fp12 c = pow_x(pow_x(b)) * frob12(frob12(b)) * conj12(b);  // ^(x^2+p^2-1)
    Fp12 r = c * (t * t * t);
    return is_one12(r);
}

}  // namespace pairing
}  // namespace kzg
