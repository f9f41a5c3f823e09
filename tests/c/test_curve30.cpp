// CPU check of the signed 13 x 30-bit group law (csrc/curve30.hpp: XYZZ mixed addition with fused subtractions, the general
// Jacobian addition and doubling of the folds, their exact slow paths, the packed table entry, the conversions to and from
// the 14 x 29-bit form) against the saturated reference formulas of csrc/curve.hpp, on random curve points and on every
// exceptional case.  Built and run by tests/test_host_units.py with hipcc's host pass (no kernel is launched).
#include "curve30.hpp"
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
static G1Jac jac_from_jacs(const JacS& p) {
    if (is_inf(p)) return jac_inf();
    G1Jac r;
    r.x = fp_from_fs(p.x);
    r.y = fp_from_fs(p.y);
    r.z = fp_from_fs(p.z);
    return r;
}
static int bad = 0, checks = 0;
static void expect(const JacS& got, const G1Jac& want, const char* what) {
    checks++;
    if (!eq(jac_from_jacs(got), want)) { bad++; if (bad < 10) printf("MISMATCH %s\n", what); }
}
static void expectq(const JacQ& got, const G1Jac& want, const char* what) {
    checks++;
    if (!eq(jac_from_jacq(got), want)) { bad++; if (bad < 10) printf("MISMATCH %s\n", what); }
}
int main() {
    for (int it = 0; it < 300; it++) {
        G1Affine Pa = random_point(), Qa = random_point();
        G1Jac P = to_jac(Pa), Q = to_jac(Qa);
        for (int k = 0; k < (it % 5); k++) { P = dbl(P); Q = add(Q, P); }  // non-trivial Z
        const JacQ pq = jacq_from_jac(P), qq = jacq_from_jac(Q);
        const JacS p = jacs_from_jacq(pq), q = jacs_from_jacq(qq);
        G1Affine Qaff = to_affine(Q), Paff = to_affine(P);
        const AffQ qa29 = affq_from_affine(Qaff);
        const AffS qa = affs_from_affq(qa29), pa = affs_from_affq(affq_from_affine(Paff));
        // conversions and the packed entry
        expect(p, P, "29 -> 30");
        expectq(jacq_from_jacs(p), P, "30 -> 29");
        {
            TabS e;
            tabs_pack_from_fq(e.w, qa29.x);
            tabs_pack_from_fq(e.w + 12, qa29.y);
            const AffS u = tabs_unpack(e.w);
            checks++;
            if (!eq(fp_from_fs(u.x), Qaff.x) || !eq(fp_from_fs(u.y), Qaff.y)) { bad++; printf("MISMATCH packed entry\n"); }
            for (int i = 0; i < SL - 1; i++)
                if (u.x.v[i] < -SHALF || u.x.v[i] >= SHALF || u.y.v[i] < -SHALF || u.y.v[i] >= SHALF) { bad++; printf("entry digit not centred\n"); }
        }
        expect(dbl(p), dbl(P), "dbl");
        expect(add(p, q), add(P, Q), "add");
        expect(add(p, q, true), add(P, neg(Q)), "sub");
        expect(add(p, p), dbl(P), "P+P");
        expect(add(p, p, true), jac_inf(), "P-P");
        expect(add(p, jacs_inf()), P, "P+O");
        expect(add(jacs_inf(), q, true), neg(Q), "O-Q");
        expect(add(jacs_inf(), jacs_inf()), jac_inf(), "O+O");
        expect(dbl(jacs_inf()), jac_inf(), "2O");
        // the sum-and-difference pair of the G1 linear map (k_slp_add_s): both results from one shared computation, aliasing the
        // first operand as the kernel does, and every exit of the slow path; then what the arena of a large batch goes through:
        // chains of pairs, additions, subtractions and halved doubling runs on points that stay in the signed form throughout
        {
            JacS s, d;
            add_sub(p, q, s, d);
            expect(s, add(P, Q), "add_sub: sum");
            expect(d, add(P, neg(Q)), "add_sub: difference");
            JacS a = p, dd;
            add_sub(a, q, a, dd);  // sum written over the first operand
            expect(a, add(P, Q), "add_sub aliased: sum");
            expect(dd, add(P, neg(Q)), "add_sub aliased: difference");
            a = p;
            add_sub(a, p, a, dd);  // P + P and P - P
            expect(a, dbl(P), "add_sub: P+P");
            expect(dd, jac_inf(), "add_sub: P-P");
            a = p;
            add_sub(a, neg(p), a, dd);  // P - P and P + P
            expect(a, jac_inf(), "add_sub: P+(-P)");
            expect(dd, dbl(P), "add_sub: P-(-P)");
            a = jacs_inf();
            add_sub(a, q, a, dd);
            expect(a, Q, "add_sub: O+Q");
            expect(dd, neg(Q), "add_sub: O-Q");
            a = p;
            add_sub(a, jacs_inf(), a, dd);
            expect(a, P, "add_sub: P+O");
            expect(dd, P, "add_sub: P-O");
            {   // the common path alone and its flag (k_slp_add_s re-reads its operands when the flag is up)
                bool deg = true;
                expect(add_unchecked(p, q, false, deg), add(P, Q), "add_unchecked");
                checks++;
                if (deg) { bad++; printf("add_unchecked flags a generic sum\n"); }
                expect(add_unchecked(p, q, true, deg), add(P, neg(Q)), "add_unchecked: difference");
                bool d1 = false, d2 = false, d3 = false, d4 = false;
                (void)add_unchecked(p, p, false, d1);
                (void)add_unchecked(p, p, true, d2);
                (void)add_unchecked(jacs_inf(), q, false, d3);
                (void)add_unchecked(p, jacs_inf(), true, d4);
                checks++;
                if (!(d1 && d2 && d3 && d4)) { bad++; printf("add_unchecked misses a degenerate case\n"); }
            }
            JacS x = p, y = q;
            G1Jac X = P, Y = Q;
            for (int k = 0; k < 12; k++) {
                JacS t;
                add_sub(x, y, x, t);                       // (x, y) <- (x + y, x - y) ...
                G1Jac T = add(X, neg(Y));
                X = add(X, Y);
                y = dbl_half(dbl_half(t));                 // ... the difference doubled twice in the halved form
                Y = dbl(dbl(T));
                if (k % 3 == 0) { x = add(x, y, true); X = add(X, neg(Y)); }
            }
            expect(x, X, "arena chain: x");
            expect(y, Y, "arena chain: y");
            checks++;
            const Fp zx = fp_from_fs(jacs_inf().z);
            bool zero = true;
            for (int i = 0; i < 12; i++) zero = zero && zx.v[i] == 0;
            if (!zero) { bad++; printf("identity's z is not canonical zero\n"); }
        }
        // XYZZ accumulator (the MSM's): sums of +-q onto p and the exceptional cases
        {
            XyzzS xa = xyzz30_inf();
            xa = add_mixed(xa, pa);                                        // O + P
            expect(to_jacs(xa), P, "xyzz O+P");
            expect(to_jacs(add_mixed(xa, qa)), add_mixed(P, Qaff), "xyzz madd");
            expect(to_jacs(add_mixed(xa, qa, true)), add_mixed(P, neg(Qaff)), "xyzz msub");
            XyzzS xq = add_mixed(xyzz30_inf(), qa, true);                  // O - Q
            expect(to_jacs(xq), neg(Q), "xyzz O-Q");
            expect(to_jacs(add_mixed(xq, qa)), jac_inf(), "xyzz -Q+Q");
            expect(to_jacs(add_mixed(xq, qa, true)), dbl(neg(Q)), "xyzz -Q-Q");
            AffS inf_a = affs_from_affq(affq_from_affine(aff_inf()));
            checks++;
            if (!affine_is_inf(inf_a)) { bad++; printf("affine identity\n"); }
            expect(to_jacs(add_mixed(xa, inf_a)), P, "xyzz P+O");
            XyzzS run = xa;
            G1Jac RUN = P;
            for (int k = 0; k < 40; k++) { run = add_mixed(run, k % 2 ? pa : qa, k % 3 == 0); RUN = add_mixed(RUN, (k % 3 == 0) ? neg(k % 2 ? Paff : Qaff) : (k % 2 ? Paff : Qaff)); }
            expect(to_jacs(run), RUN, "xyzz chain");
            // the fold of the chunked kernel: (c0 + c1) + phi(c2 + c3) shape, through the 14 x 29-bit form at the end
            const JacS s01 = add(to_jacs(run), to_jacs(xa));
            expectq(jacq_from_jacs(s01), add(RUN, P), "fold + convert");
        }
        // the constant multiplication's chain: halved doublings (lambda = 1/2 scaling: same projective point) and mixed additions
        // with a table entry whose coordinates are fresh products
        {
            AffT qt;
            qt.x = mul(qa.x, fs_one());
            qt.y = mul(qa.y, fs_one());
            expect(dbl_half(p), dbl(P), "dbl_half");
            expect(dbl_half(jacs_inf()), jac_inf(), "dbl_half O");
            expect(add_mixed(p, qt, false), add_mixed(P, Qaff), "madd T");
            expect(add_mixed(p, qt, true), add_mixed(P, neg(Qaff)), "msub T");
            const JacS qj = jacs_from_jacq(jacq_from_jac(to_jac(Qaff)));
            expect(add_mixed(qj, qt, false), dbl(Q), "Q+Q T");
            expect(add_mixed(qj, qt, true), jac_inf(), "Q-Q T");
            expect(add_mixed(jacs_inf(), qt, true), neg(Q), "O-Q T");
            JacS c = p;
            G1Jac C = P;
            for (int k = 0; k < 60; k++) {
                if (k % 4 != 3) { c = dbl_half(c); C = dbl(C); }
                else { c = add_mixed(c, qt, k & 8); C = add_mixed(C, (k & 8) ? neg(Qaff) : Qaff); }
            }
            expect(c, C, "mulc chain");
            expectq(jacq_from_jacs(c), C, "mulc chain -> 29");
        }
        // chains keep the stored bounds
        JacS acc = p;
        G1Jac ACC = P;
        for (int k = 0; k < 40; k++) {
            if (k % 3 == 0) { acc = dbl(acc); ACC = dbl(ACC); }
            else { acc = add(acc, q, k & 8); ACC = add(ACC, (k & 8) ? neg(Q) : Q); }
        }
        expect(acc, ACC, "chain");
    }
    printf("curve30: %d checks, %d mismatches\n", checks, bad);
    return bad ? 1 : 0;
}
