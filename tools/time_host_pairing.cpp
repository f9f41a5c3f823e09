// How long the host side of a verification's pairing check takes (csrc/host_pairing.cpp): two Miller loops + one final exponentiation.
//   g++ -O3 -march=native -std=c++17 -I rust-eth-kzg_amd/csrc tools/time_host_pairing.cpp -o /tmp/time_host_pairing && /tmp/time_host_pairing rust-eth-kzg_amd/data/trusted_setup_4096.bin
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "host_pairing.cpp"
using namespace kzg;
using namespace kzg::pairing;
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    init();
    FILE* fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    std::vector<uint8_t> srs(16 + 4096 * 48 + 65 * 96);
    if (fread(srs.data(), 1, srs.size(), fh) != srs.size()) return 2;
    fclose(fh);
    const uint8_t *g1 = srs.data() + 16, *g2 = g1 + 4096 * 48;
    G1Affine one1, tau1;
    G2Affine one2_, tau2;
    if (g1_decompress(one1, g1) || g1_decompress(tau1, g1 + 48) || !g2_decompress(one2_, g2) || !g2_decompress(tau2, g2 + 96)) return 3;
    const G2Prepared q_one = prepare(one2_), q_tau = prepare(tau2);
    const G2Prepared* q[2] = {&q_one, &q_tau};
    const G1Affine ok[2] = {tau1, neg(one1)};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int R = 200;
    int good = 0;
    auto t0 = now();
    for (int i = 0; i < R; i++) good += product_is_one(ok, q, 2);
    auto t1 = now();
    Fp12 f = miller_loop(ok[0], q_one);
    for (int i = 0; i < R; i++) f = miller_loop(ok[i & 1], i & 1 ? q_tau : q_one);
    auto t2 = now();
    for (int i = 0; i < R; i++) good += final_exponentiation_is_one(f);
    auto t3 = now();
    for (int i = 0; i < R; i++) { G2Prepared p = prepare(tau2); good += !p.inf; }
    auto t4 = now();
    printf("pairing check of two pairs: %.3f ms   one Miller loop: %.3f ms   final exponentiation: %.3f ms   prepare(G2): %.3f ms   (%d)\n",
           ms(t0, t1) / R, ms(t1, t2) / R, ms(t2, t3) / R, ms(t3, t4) / R, good);
    return 0;
}
