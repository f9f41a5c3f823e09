// CPU check of csrc/glv.hpp (the balanced GLV split behind the GLV window table): for random and edge scalars,
// k1 + k2 * lambda == k (mod r) over Fr, |k1|, |k2| < 2^127, and the unsigned split r + q * lambda == k with r < lambda.
// Built and run by tests/test_host_units.py with hipcc's host pass (no kernel is launched).
#include "glv.hpp"
#include <cstdio>
using namespace kzg;

static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static Fr from_words(const uint32_t* w, int n) {
    Fr a = zero<FrParams>();
    for (int i = 0; i < n; i++) a.v[i] = w[i];
    return to_mont(a);
}
int main() {
    const uint32_t L[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};
    const Fr lam = from_words(L, 4);
    int bad = 0;
    if (!is_zero(add(add(sqr(lam), lam), one<FrParams>()))) { printf("lambda is not a cube root of unity\n"); return 1; }
    auto check = [&](const Fr& k_canon) {
        const Fr km = to_mont(k_canon);
        uint32_t r4[4], q[4];
        glv_split_unsigned(k_canon, r4, q);
        if (!eq(add(from_words(r4, 4), mul(from_words(q, 4), lam)), km) || !gt128(L, r4)) bad++;
        uint32_t out[8];
        glv_split_balanced(k_canon, out);
        const bool n1 = out[3] >> 31, n2 = out[7] >> 31;
        uint32_t m1[4] = {out[0], out[1], out[2], out[3] & 0x7fffffffu}, m2[4] = {out[4], out[5], out[6], out[7] & 0x7fffffffu};
        Fr k1 = from_words(m1, 4), k2 = from_words(m2, 4);
        if (n1) k1 = neg(k1);
        if (n2) k2 = neg(k2);
        if (!eq(add(k1, mul(k2, lam)), km)) bad++;
        // magnitudes <= (lambda + 1) / 2 + 1 (bit 127 is the sign bit, so < 2^127 holds by construction of m1 / m2: check the bound)
        const uint32_t HB[4] = {0x80000001u, 0x00000000u, 0x8000d201u, 0x5622d200u};  // (lambda + 1) / 2 + 1
        if (gt128(m1, HB) || gt128(m2, HB)) bad++;
    };
    for (int it = 0; it < 200000; it++) {
        Fr k;
        for (int i = 0; i < 8; i++) k.v[i] = rnd();
        k.v[7] &= 0x3fffffffu;  // < 2^254 < r
        if (it % 7 == 0) for (int i = 4; i < 8; i++) k.v[i] = 0;  // small scalars
        check(k);
    }
    // edges: 0, 1, r - 1, multiples of lambda and their neighbours, the branch points of the balancing steps
    Fr e = zero<FrParams>();
    check(e);
    e.v[0] = 1; check(e);
    Fr rm1;
    for (int i = 0; i < 8; i++) rm1.v[i] = FrParams::MOD[i];
    rm1.v[0] -= 1; check(rm1);
    const uint32_t halves[3][4] = {{0x7fffffffu, 0, 0x8000d201u, 0x5622d200u}, {0x80000000u, 0, 0x8000d201u, 0x5622d200u}, {0x80000001u, 0, 0x8000d201u, 0x5622d200u}};
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {  // k = halves[a] + halves[b] * lambda (+-1), as a canonical integer < r
            Fr km = add(from_words(halves[a], 4), mul(from_words(halves[b], 4), lam));
            for (int d = -1; d <= 1; d++) {
                Fr kk = d < 0 ? sub(km, one<FrParams>()) : d > 0 ? add(km, one<FrParams>()) : km;
                check(from_mont(kk));
            }
        }
    for (int m = 0; m < 4; m++) {
        uint32_t w[4] = {(uint32_t)m, 0, 0, 0};
        Fr km = mul(from_words(w, 4), lam);
        check(from_mont(km));
        check(from_mont(add(km, one<FrParams>())));
        check(from_mont(sub(km, one<FrParams>())));
    }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
