// CPU check of the signed 13 x 30-bit field (csrc/fp30.hpp) against the saturated Montgomery field of csrc/field.hpp:
// products at every operand class the kernels use (centred x centred, centred x floor digits, centred x lazy difference),
// squarings, the fused "product plus injected digits" forms, the one-reduction product pairs (one accumulator and the split
// form), the additive steps, canonical form and the conversions.  Worst-case digit patterns (every digit at its class bound,
// signs aligned) run through the same functions: built with -fsanitize=signed-integer-overflow a column that passed 2^63
// would abort (tests/test_sanitizers.py).  With an argument the program also dumps raw digit triples for an exact
// comparison with Python's integers (tests/test_host_units.py).
// Built and run with hipcc's host pass (no kernel is launched).
#include "fp30.hpp"
#include <cstdio>
#include <cstring>
using namespace kzg;

static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); }
static Fp random_fp() {
    Fp x;
    for (int i = 0; i < 12; i++) x.v[i] = rnd();
    x.v[11] &= 0x0fffffffu;  // < 2^380 < p
    return x;
}
static int bad = 0, checks = 0;
static void expect_eq(const Fp& got, const Fp& want, const char* what) {
    checks++;
    if (!eq(got, want)) { bad++; if (bad < 20) printf("MISMATCH %s\n", what); }
}
template <int B, int F>
static void expect_class(const Fs<B, F>& a, const char* what) {  // digits inside their class
    checks++;
    bool ok = true;
    for (int i = 0; i < SL - 1; i++) {
        const int64_t d = a.v[i];
        if (F == DC) ok &= d >= -(int64_t)SHALF - 4 && d <= (int64_t)SHALF + 4;
        else if (F == DU) ok &= d >= 0 && d <= SMASK;
        else ok &= d >= -(int64_t)(1 << 30) - 8 && d <= (int64_t)(1 << 30) + 8;
    }
    const int64_t top = a.v[SL - 1];
    ok &= top > -(int64_t)B * 2100000 - 8 && top < (int64_t)B * 2100000 + 8;  // p / 2^360 = 1.70 M
    if (!ok) { bad++; if (bad < 20) printf("CLASS %s\n", what); }
}
template <int B, int F>
static void dump(FILE* f, const char* tag, const Fs<B, F>& a) {
    fprintf(f, "%s", tag);
    for (int i = 0; i < SL; i++) fprintf(f, " %d", a.v[i]);
    fprintf(f, "\n");
}

int main(int argc, char** argv) {
    FILE* dumpf = argc > 1 ? fopen(argv[1], "w") : nullptr;
    for (int it = 0; it < 4000; it++) {
        const Fp A = random_fp(), B = random_fp(), C = random_fp(), D = random_fp();
        const Fs<1, DC> a = fs_from_fp(A), b = fs_from_fp(B), c = fs_from_fp(C), d = fs_from_fp(D);
        const Fs<1, DU> au = fs_from_fp<DU>(A), bu = fs_from_fp<DU>(B), cu = fs_from_fp<DU>(C), du = fs_from_fp<DU>(D);
        expect_class(a, "from_fp C"); expect_class(au, "from_fp U");
        expect_eq(fp_from_fs(a), A, "round trip C");
        expect_eq(fp_from_fs(au), A, "round trip U");
        const Fp AB = mul(A, B), CD = mul(C, D);
        // products
        expect_eq(fp_from_fs(mul(a, b)), AB, "mul CxC->C");
        expect_eq(fp_from_fs(mul<DU>(a, b)), AB, "mul CxC->U");
        expect_eq(fp_from_fs(mul(a, bu)), AB, "mul CxU->C");
        expect_eq(fp_from_fs(mul<DU>(au, b)), AB, "mul UxC->U");
        expect_class(mul(a, bu), "mul class C"); expect_class(mul<DU>(au, b), "mul class U");
        const auto w = sub_lazy(au, cu);  // A - C, wide digits
        expect_class(w, "lazy sub class");
        expect_eq(fp_from_fs(mul(w, b)), mul(sub(A, C), B), "mul WxC");
        expect_eq(fp_from_fs(mul<DU>(b, neg(w))), mul(sub(C, A), B), "mul Cx(-W)");
        expect_eq(fp_from_fs(sqr(a)), sqr(A), "sqr ->C");
        expect_eq(fp_from_fs(sqr<DU>(a)), sqr(A), "sqr ->U");
        // fused forms: a b - c, a b + 2 c, a^2 - c - 2 d
        expect_eq(fp_from_fs(mul_inj<-1>(a, bu, cu)), sub(AB, C), "mul_inj -1");
        expect_eq(fp_from_fs(mul_inj<2, DU>(a, b, c)), add(AB, add(C, C)), "mul_inj +2");
        expect_eq(fp_from_fs(mul_inj<1, DU>(au, b, w)), add(AB, sub(A, C)), "mul_inj wide");
        expect_eq(fp_from_fs(sqr_inj2<-1, -2, DU>(a, cu, du)), sub(sub(sqr(A), C), add(D, D)), "sqr_inj2");
        expect_eq(fp_from_fs(sqr_inj<-2>(a, d)), sub(sqr(A), add(D, D)), "sqr_inj");
        expect_class(mul_inj<-1>(a, bu, cu), "mul_inj class");
        expect_class(sqr_inj2<-1, -2, DU>(a, cu, du), "sqr_inj2 class");
        // one-reduction pairs
        expect_eq(fp_from_fs(mul_add(a, b, c, d)), add(AB, CD), "mul_add CCCC");
        expect_eq(fp_from_fs(mul_add<DU>(a, w, cu, d)), add(mul(A, sub(A, C)), CD), "mul_add split");
        expect_eq(fp_from_fs(mul_add(a, bu, neg(cu), d)), sub(AB, CD), "mul_add split neg");
        expect_class(mul_add<DU>(a, w, cu, d), "mul_add split class");
        // additive steps
        expect_eq(fp_from_fs(add(a, bu)), add(A, B), "add");
        expect_eq(fp_from_fs(sub(au, bu)), sub(A, B), "sub UU");
        expect_eq(fp_from_fs(sub(a, w)), sub(A, sub(A, C)), "sub CW");
        expect_class(sub(a, w), "sub class"); expect_class(add(a, bu), "add class");
        expect_eq(fp_from_fs(mul_small<3>(a)), add(A, add(A, A)), "3a");
        expect_class(mul_small<3>(a), "3a class");
        expect_eq(fp_from_fs(cneg(true, au)), neg(A), "cneg");
        expect_eq(fp_from_fs(cneg(false, a)), A, "cneg no");
        expect_eq(fp_from_fs(neg(a)), neg(A), "neg C");
        expect_eq(fp_from_fs(add_lazy(a, b)), add(A, B), "add_lazy");
        // a chain: bounds accumulate, products still exact
        {
            auto s1 = add(a, b);                    // 2
            auto s2 = sub(s1, cu);                  // 3
            auto s3 = mul_small<3>(s2);             // 9
            auto s4 = add(s3, s3);                  // 18
            Fp want = sub(add(A, B), C);
            want = add(want, add(want, want));
            want = add(want, want);
            expect_eq(fp_from_fs(mul(s4, d)), mul(want, D), "chain");
            expect_eq(fp_from_fs(sqr(relax<16, DC>(s3))), sqr(add(sub(add(A, B), C), add(sub(add(A, B), C), sub(add(A, B), C)))), "chain sqr");
        }
        // zero tests
        checks++;
        if (!product_is_zero(mul(a, fs_zero())) || product_is_zero(mul<DU>(a, b)) || !is_zero_slow(sub(au, au)) || !is_zero_slow(sub_lazy(a, a))) { bad++; printf("zero tests\n"); }
        if (dumpf && it < 200) {
            dump(dumpf, "a", a); dump(dumpf, "bu", bu); dump(dumpf, "w", w); dump(dumpf, "cu", cu); dump(dumpf, "du", du);
            dump(dumpf, "mulCU", mul(a, bu)); dump(dumpf, "mulWC_U", mul<DU>(w, a)); dump(dumpf, "sqr", sqr(a));
            dump(dumpf, "sqr_inj2", sqr_inj2<-1, -2, DU>(a, cu, du)); dump(dumpf, "mul_add_split", mul_add<DU>(a, w, cu, a));
            dump(dumpf, "canon_w", canonical(w));
        }
    }
    // worst-case digit patterns: every digit at its class bound, all products of a column with the same sign
    for (int pat = 0; pat < 64; pat++) {
        Fs<32, DC> c1, c2;
        Fs<32, DW> w1, w2;
        for (int i = 0; i < SL - 1; i++) {
            const int s1 = (pat & 1) ? -1 : 1, s2 = (pat & 2) ? -1 : 1, alt = (pat & 4) ? ((i & 1) ? -1 : 1) : 1;
            c1.v[i] = s1 * alt * (SHALF + 4);
            c2.v[i] = s2 * alt * (SHALF + 4);
            w1.v[i] = s2 * alt * ((1 << 30) + 8);
            w2.v[i] = s1 * ((pat & 8) ? alt : 1) * ((1 << 30) + 8);
        }
        const int top = ((pat & 16) ? -1 : 1) * ((1 << 26) - 1);
        c1.v[SL - 1] = top; c2.v[SL - 1] = (pat & 32) ? -top : top; w1.v[SL - 1] = top; w2.v[SL - 1] = -top;
        // the results' values are checked through the homomorphism: compare with the same product formed from canonical copies
        const Fp C1 = fp_from_fs(c1), C2 = fp_from_fs(c2), W1 = fp_from_fs(relax<32, DW>(normalise(w1))), W2 = fp_from_fs(relax<32, DW>(normalise(w2)));
        // bounds of the types are lied about on purpose (values up to 2^386): only the column sums matter here
        Fs<8, DC> c1s, c2s; Fs<8, DW> w1s, w2s;
        memcpy(&c1s, &c1, sizeof c1); memcpy(&c2s, &c2, sizeof c2); memcpy(&w1s, &w1, sizeof w1); memcpy(&w2s, &w2, sizeof w2);
        expect_eq(fp_from_fs(mul(c1s, w1s)), mul(C1, W1), "worst CxW");
        expect_eq(fp_from_fs(mul<DU>(w2s, c2s)), mul(W2, C2), "worst WxC");
        expect_eq(fp_from_fs(sqr(c1s)), sqr(C1), "worst sqr");
        expect_eq(fp_from_fs(mul_add(c1s, c2s, c2s, c1s)), add(mul(C1, C2), mul(C1, C2)), "worst mul_add CCCC");
        expect_eq(fp_from_fs(mul_add<DU>(c1s, w1s, w2s, c2s)), add(mul(C1, W1), mul(W2, C2)), "worst mul_add split");
        expect_eq(fp_from_fs(mul_inj<-2>(c1s, w1s, w2s)), sub(mul(C1, W1), add(W2, W2)), "worst mul_inj");
        expect_eq(fp_from_fs(sqr_inj2<-1, -2, DU>(c2s, w1s, w2s)), sub(sub(sqr(C2), W1), add(W2, W2)), "worst sqr_inj2");
    }
    if (dumpf) fclose(dumpf);
    printf("%d checks, %d mismatches\n", checks, bad);
    return bad != 0;
}
