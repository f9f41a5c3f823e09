// CPU check of the unsaturated group law (csrc/curve29.hpp, fp29.hpp incl. the fused mul_add) against the saturated
// reference formulas of csrc/curve.hpp, on random curve points and on every exceptional case.
// Built and run by tests/test_host_units.py with hipcc's host pass (no kernel is launched).
#include "curve29.hpp"
#include <cstdio>
using namespace kzg;

static uint64_t st = 0x243f6a8885a308d3ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); }

static G1Affine random_point() {  // on the curve y^2 = x^3 + 4 (not necessarily in the r-torsion: irrelevant to the formulas)
    for (;;) {
        Fp x;
        for (int i = 0; i < 12; i++) x.v[i] = rnd();
        x.v[11] &= 0x0fffffffu;
        Fp four = zero<FpParams>();
        four.v[0] = 4;
        Fp rhs = add(mul(sqr(x), x), to_mont(four)), y;
        if (fp_sqrt(y, rhs)) { G1Affine a; a.x = x; a.y = (rnd() & 1) ? y : neg(y); return a; }
    }
}
static int bad = 0, checks = 0;
static void expect(const JacQ& got, const G1Jac& want, const char* what) {
    checks++;
    if (!eq(jac_from_jacq(got), want)) { bad++; if (bad < 10) printf("MISMATCH %s\n", what); }
}
int main() {
    for (int it = 0; it < 300; it++) {
        G1Affine Pa = random_point(), Qa = random_point();
        G1Jac P = to_jac(Pa), Q = to_jac(Qa);
        for (int k = 0; k < (it % 5); k++) { P = dbl(P); Q = add(Q, P); }  // non-trivial Z
        JacQ p = jacq_from_jac(P), q = jacq_from_jac(Q);
        G1Affine Qaff = to_affine(Q);
        AffQ qa = affq_from_affine(Qaff);
        expect(dbl(p), dbl(P), "dbl");
        expect(add(p, q), add(P, Q), "add");
        expect(add(p, q, true), add(P, neg(Q)), "sub");
        expect(add_mixed(p, qa), add_mixed(P, Qaff), "madd");
        expect(add_mixed(p, qa, true), add_mixed(P, neg(Qaff)), "msub");
        // exceptional cases
        expect(add(p, p), dbl(P), "P+P");
        expect(add(p, p, true), jac_inf(), "P-P");
        expect(add(p, jacq_inf()), P, "P+O");
        expect(add(jacq_inf(), q, true), neg(Q), "O-Q");
        expect(add(jacq_inf(), jacq_inf()), jac_inf(), "O+O");
        expect(dbl(jacq_inf()), jac_inf(), "2O");
        JacQ qj = jacq_from_jac(to_jac(Qaff));
        expect(add_mixed(qj, qa), dbl(Q), "Q+Q mixed");
        expect(add_mixed(qj, qa, true), jac_inf(), "Q-Q mixed");
        expect(add_mixed(jacq_inf(), qa, true), neg(Q), "O-Q mixed");
        AffQ inf_a = affq_from_affine(aff_inf());
        expect(add_mixed(p, inf_a), P, "P+O mixed");
        // mixed addition with a non-canonical affine operand (the twiddle table on the isomorphic curve with Z = 1)
        {
            AffQ2 q2;
            q2.x = mul(qa.x, fq_one());  // same value, as a fresh product (< 2p)
            q2.y = mul(qa.y, fq_one());
            expect(add_mixed(p, q2, false), add_mixed(P, Qaff), "madd2");
            expect(add_mixed(p, q2, true), add_mixed(P, neg(Qaff)), "msub2");
            expect(add_mixed(qj, q2, false), dbl(Q), "Q+Q mixed2");
            expect(add_mixed(qj, q2, true), jac_inf(), "Q-Q mixed2");
            expect(add_mixed(jacq_inf(), q2, true), neg(Q), "O-Q mixed2");
        }
        // XYZZ accumulator (the MSM's): sums of +-q onto p and the exceptional cases
        {
            XyzzQ xa = xyzz_inf();
            xa = add_mixed(xa, affq_from_affine(to_affine(P)));           // O + P
            expect(to_jacq(xa), P, "xyzz O+P");
            expect(to_jacq(add_mixed(xa, qa)), add_mixed(P, Qaff), "xyzz madd");
            expect(to_jacq(add_mixed(xa, qa, true)), add_mixed(P, neg(Qaff)), "xyzz msub");
            XyzzQ xq = add_mixed(xyzz_inf(), qa, true);                   // O - Q
            expect(to_jacq(xq), neg(Q), "xyzz O-Q");
            expect(to_jacq(add_mixed(xq, qa)), jac_inf(), "xyzz -Q+Q");
            expect(to_jacq(add_mixed(xq, qa, true)), dbl(neg(Q)), "xyzz -Q-Q");
            expect(to_jacq(add_mixed(xa, inf_a)), P, "xyzz P+O");
            XyzzQ run = xa;
            G1Jac RUN = P;
            for (int k = 0; k < 30; k++) { run = add_mixed(run, qa, k % 3 == 0); RUN = add_mixed(RUN, (k % 3 == 0) ? neg(Qaff) : Qaff); }
            expect(to_jacq(run), RUN, "xyzz chain");
        }
        // chains keep the stored bounds: 40 steps of mixed operations
        JacQ acc = p;
        G1Jac ACC = P;
        for (int k = 0; k < 40; k++) {
            if (k % 3 == 0) { acc = dbl(acc); ACC = dbl(ACC); }
            else if (k % 3 == 1) { acc = add_mixed(acc, qa, k & 4); ACC = add_mixed(ACC, (k & 4) ? neg(Qaff) : Qaff); }
            else { acc = add(acc, q, k & 8); ACC = add(ACC, (k & 8) ? neg(Q) : Q); }
        }
        expect(acc, ACC, "chain");
    }
    printf("curve29: %d checks, %d mismatches\n", checks, bad);
    return bad ? 1 : 0;
}
