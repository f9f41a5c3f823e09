// Explorer for csrc/g1_linmap.hpp: every assignment of Hankel splits (2 / 4 / 8 per size) compiled into the FK20 proofs map,
// priced for LATENCY on a chip that is not full -- constant multiplications (one wave each), dependency levels, and the longest
// operation of every level -- next to the operation counts the throughput model uses.  Host only:
//   hipcc -O2 -std=c++17 -x hip --cuda-host-only -I rust-eth-kzg_amd/csrc tools/linmap_explore.cpp -o /tmp/linmap_explore
#include "g1_linmap.hpp"
#include <cstdio>
#include <cstdlib>
using namespace kzg;
using namespace kzg::linmap;

static Fr fr_pow(Fr b, const uint32_t* e, int nl) {
    Fr acc = one<FrParams>();
    for (int i = 32 * nl - 1; i >= 0; i--) { acc = sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, b); }
    return acc;
}
int main(int argc, char** argv) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
    e[0] -= 1;
    for (int s = 0; s < 7; s++) for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
    const Fr g = fr_pow(fr_small(7), e, 8);
    std::vector<Fr> w(128);
    w[0] = one<FrParams>();
    for (int i = 1; i < 128; i++) w[i] = mul(w[i - 1], g);
    const double NS_PER_INSTR = 4.4 / 2.155;  // one wave per SIMD, three-operand class
    const double GAP_US = argc > 1 ? atof(argv[1]) : 5.0;
    printf("splits(2,4,8,16,32) balanced | mulc add dbl | levels  cheap_us  (max-op instr per level)\n");
    for (int balanced = 0; balanced < 2; balanced++)
        for (int k32 : {2, 4, 8}) for (int k16 : {2, 4, 8}) for (int k8 : {2, 4, 8}) for (int k4 : {2, 4}) {
            Strategy S;
            S.tuned = false;
            S.hankel_split = {{2, 2}, {4, k4}, {8, k8}, {16, k16}, {32, k32}};
            S.balanced_lincomb = balanced != 0;
            Plan plan = build_fk20_proofs_plan(w, S);
            const Schedule sc = make_schedule(plan, /*fuse_add_sub=*/false);
            double cheap_us = 0;
            int levels = 0, mulc_launches = 0;
            std::string detail;
            for (auto& L : sc.launches) {
                if (L.kind == OP_MULC) { mulc_launches++; detail += " [M]"; continue; }
                double mx = 0;
                for (int i = 0; i < L.count; i++) {
                    const uint32_t* wd = &sc.words[(size_t)(L.first + i) * 4];
                    const double c = (wd[3] & 2u) ? wd[2] * COST_DBL : ((wd[3] >> 3) & 31u) * COST_DBL + COST_ADD;
                    if (c > mx) mx = c;
                }
                levels++;
                cheap_us += mx * NS_PER_INSTR * 1e-3 + GAP_US;
                char buf[32];
                snprintf(buf, sizeof buf, " %.0fk", mx / 1e3);
                detail += buf;
            }
            printf("%d %d %d %d %d  %d | %4ld %5ld %5ld | %2d+%dM %7.0f us |%s\n", 2, k4, k8, k16, k32, balanced, plan.count(OP_MULC),
                   plan.count(OP_ADD) + plan.count(OP_SUB), plan.doublings(), levels, mulc_launches, cheap_us, detail.c_str());
        }
    return 0;
}
