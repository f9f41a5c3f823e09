// CPU check of the host pairing (csrc/host_pairing.cpp): the sparse line multiplication, the complex squaring and the
// cyclotomic squaring against the general Fp12 product, and the pairing check itself on points of the trusted setup:
//   e(a [tau]_1, [1]_2) * e(-a [1]_1, [tau]_2) == 1   and   e(a [1]_1, [1]_2) * e(-a [1]_1, [tau]_2) != 1.
// Built and run by tests/test_host_units.py with the path of rust-eth-kzg_amd/data/trusted_setup_4096.bin.
#include <thread>
#include <vector>
#include "host_pairing.cpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace kzg;
using namespace kzg::pairing;

static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static Fp rfp() {  // any residue below 2^380 < p, read as a Montgomery form
    Fp a;
    for (int i = 0; i < 12; i++) a.v[i] = rnd();
    a.v[11] &= 0x0fffffffu;
    return a;
}
static Fp2 rfp2() { return {rfp(), rfp()}; }
static Fp6 rfp6() { return {rfp2(), rfp2(), rfp2()}; }
static Fp12 rfp12() { return {rfp6(), rfp6()}; }
// the value of a line at P as a general Fp12 element: l(P) * w^3 = a + (b * xP) v + yP (v w)
static Fp12 line_value(const Line& l, const G1Affine& P) {
    Fp12 r;
    r.c0 = {l.a, scale(l.b, P.x), zero2()};
    r.c1 = {zero2(), {P.y, fz()}, zero2()};
    return r;
}
static bool eq12(const Fp12& a, const Fp12& b) {
    const Fp2 *x = &a.c0.c0, *y = &b.c0.c0;
    for (int i = 0; i < 6; i++)
        if (!eq2(x[i], y[i])) return false;
    return true;
}

int main(int argc, char** argv) {
    {   // contexts are created from many threads, and the engines of a device list side by side: init() from eight threads at once
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++) th.emplace_back([] { init(); });
        for (auto& t : th) t.join();
    }
    init();
    int bad = 0;
    for (int it = 0; it < 200; it++) {
        const Fp12 f = rfp12();
        if (!eq12(sqr12(f), f * f)) bad++;
        Line l{rfp2(), rfp2()};
        G1Affine P;
        P.x = rfp();
        P.y = rfp();
        if (!eq12(mul12_by_line(f, l.a, scale(l.b, P.x), P.y), f * line_value(l, P))) bad++;
        // into the cyclotomic subgroup by the easy part of the final exponentiation
        Fp12 t = conj12(f) * inv12(f);
        t = frob12(frob12(t)) * t;
        if (!eq12(cyc_sqr12(t), t * t)) bad++;
        if (!eq12(t * conj12(t), one12())) bad++;  // the inverse of a cyclotomic element is its conjugate
    }
    printf("fp12 building blocks: %d mismatches\n", bad);
    if (argc < 2) return 2;
    FILE* fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    std::vector<uint8_t> srs(16 + 4096 * 48 + 65 * 96);
    if (fread(srs.data(), 1, srs.size(), fh) != srs.size() || memcmp(srs.data(), "KZGSRS01", 8)) return 2;
    fclose(fh);
    const uint8_t *g1 = srs.data() + 16, *g2 = g1 + 4096 * 48;
    G1Affine one1, tau1;
    G2Affine one2_, tau2;
    if (g1_decompress(one1, g1) || g1_decompress(tau1, g1 + 48) || !g2_decompress(one2_, g2) || !g2_decompress(tau2, g2 + 96)) return 3;
    const G2Prepared q_one = prepare(one2_), q_tau = prepare(tau2);
    const G2Prepared* q[2] = {&q_one, &q_tau};
    int pbad = 0;
    for (int it = 0; it < 6; it++) {
        uint32_t k[8];
        for (int i = 0; i < 8; i++) k[i] = it ? rnd() : (i == 0);
        k[7] &= 0x3fffffffu;
        const G1Affine a_tau = to_affine(scalar_mul<8>(to_jac(tau1), k)), a_one = to_affine(scalar_mul<8>(to_jac(one1), k));
        const G1Affine ok[2] = {a_tau, neg(a_one)}, no[2] = {a_one, neg(a_one)};
        if (!product_is_one(ok, q, 2)) pbad++;
        if (product_is_one(no, q, 2)) pbad++;
        // the check in pieces (one Miller loop per pair, as the engine runs them on two threads) decides the same
        if (!final_exponentiation_is_one(fp12_mul(miller_loop(ok[0], q_one), miller_loop(ok[1], q_tau)))) pbad++;
        if (final_exponentiation_is_one(fp12_mul(miller_loop(no[0], q_one), miller_loop(no[1], q_tau)))) pbad++;
    }
    // identities: e(O, Q) = 1
    const G1Affine id[2] = {aff_inf(), aff_inf()};
    if (!product_is_one(id, q, 2)) pbad++;
    printf("pairing checks: %d mismatches\n", pbad);
    return (bad | pbad) != 0;
}
