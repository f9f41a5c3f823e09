// CPU check of csrc/g1_linmap.hpp: the FK20 proofs map compiled to a straight-line program of point operations must
// equal its definition (IDFT_128, keep 64, DFT_128) when run over Fr, both as a plan and as the scheduled slot program
// the device executes.  Also checks the Toeplitz / Hankel building blocks on their own.
// Built and run by tests/test_host_units.py with hipcc's host pass (no kernel is launched).
#include "g1_linmap.hpp"
#include <cstdio>
using namespace kzg;
using namespace kzg::linmap;

static uint64_t st = 0x243f6a8885a308d3ull;
static Fr rnd_fr() {
    Fr a;
    for (int i = 0; i < 8; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a.v[i] = (uint32_t)(st >> 16); }
    a.v[7] &= 0x3fffffffu;
    return a;
}
static Fr fr_pow(Fr b, const uint32_t* e, int nl) {
    Fr acc = one<FrParams>();
    for (int i = 32 * nl - 1; i >= 0; i--) { acc = sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, b); }
    return acc;
}
int main() {
    int bad = 0;
    // omega_128 = 7^((r-1)/128)
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
    e[0] -= 1;
    for (int s = 0; s < 7; s++) for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
    const Fr g = fr_pow(fr_small(7), e, 8);
    std::vector<Fr> w(128);
    w[0] = one<FrParams>();
    for (int i = 1; i < 128; i++) w[i] = mul(w[i - 1], g);
    if (!eq(mul(w[127], g), one<FrParams>()) || eq(w[64], one<FrParams>())) { printf("bad root of unity\n"); return 1; }

    // building blocks: Hankel products of every size and split against the definition
    for (int k : {2, 4, 8}) {
        for (int n : {2, 4, 8, 16, 32}) {
            if (n % k) continue;
            Compiler C;
            C.tune(32);
            C.hankel_split[n] = k;
            Builder B(n);
            std::vector<Ref> x(n);
            for (int i = 0; i < n; i++) x[i] = B.input(i);
            std::vector<Fr> h(2 * n - 1), in(n);
            for (auto& v : h) v = rnd_fr();
            for (auto& v : in) v = rnd_fr();
            auto y = C.hankel(B, x, h);
            Plan p = B.take(y);
            auto got = run_over_fr(p, in);
            for (int i = 0; i < n; i++) {
                Fr acc = zero<FrParams>();
                for (int j = 0; j < n; j++) acc = add(acc, mul(h[i + j], in[j]));
                if (!eq(acc, got[i])) { bad++; if (bad < 5) printf("MISMATCH hankel n=%d k=%d i=%d\n", n, k, i); }
            }
            printf("hankel n=%2d split %d: %4ld mulc %5ld add %5ld dbl\n", n, k, p.count(OP_MULC), p.count(OP_ADD) + p.count(OP_SUB), p.doublings());
        }
    }
    // the full map, with and without the 8-way split
    for (int toom8 = 0; toom8 < 2; toom8++) {
        Plan p = build_fk20_proofs_plan(w, toom8 != 0, true);
        Schedule S = make_schedule(p);
        printf("fk20 plan (toom8=%d): %ld mulc, %ld add/sub, %ld doublings; schedule: %zu launches, %d slots, cost %.1f M instr (radix-2: %.1f M)\n",
               toom8, p.count(OP_MULC), p.count(OP_ADD) + p.count(OP_SUB), p.doublings(), S.launches.size(), S.n_slots,
               (p.count(OP_MULC) * COST_MULC + (p.count(OP_ADD) + p.count(OP_SUB)) * COST_ADD + p.doublings() * COST_DBL) / 1e6,
               (642 * COST_MULC + 2 * 7 * 128 * COST_ADD) / 1e6);
        int ml = 0;
        for (auto& L : S.launches) if (L.kind == OP_MULC) printf("  mulc launch %d: %d multiplications\n", ml++, L.count);
        for (int it = 0; it < 3; it++) {
            std::vector<Fr> in(128);
            for (auto& v : in) v = rnd_fr();
            if (it == 1) for (auto& v : in) v = zero<FrParams>();
            if (it == 2) for (int j = 0; j < 128; j++) in[j] = j == 5 ? one<FrParams>() : zero<FrParams>();
            auto want = fk20_proofs_map_by_definition(w, in);
            auto got = run_over_fr(p, in);
            auto got2 = run_schedule_over_fr(S, p.consts, 128, 128, in);
            for (int k = 0; k < 128; k++) {
                if (!eq(want[k], got[k])) { bad++; if (bad < 5) printf("MISMATCH plan out %d\n", k); }
                if (!eq(want[k], got2[k])) { bad++; if (bad < 5) printf("MISMATCH schedule out %d\n", k); }
            }
            // every form of the schedule the engine can pick: with / without the fused a + b, a - b pairs, with / without the
            // doubling runs folded into their consumers
            for (int form = 0; form < 4; form++) {
                const Schedule F = make_schedule(p, (form & 1) != 0, (form & 2) != 0);
                if (it == 0) printf("  schedule form pairs=%d runs=%d: %zu launches, %ld fused pairs, %ld folded runs, %d slots\n", form & 1, (form >> 1) & 1,
                                    F.launches.size(), F.fused_pairs, F.fused_runs, F.n_slots);
                const auto got3 = run_schedule_over_fr(F, p.consts, 128, 128, in);
                for (int k = 0; k < 128; k++)
                    if (!eq(want[k], got3[k])) { bad++; if (bad < 5) printf("MISMATCH schedule form %d out %d\n", form, k); }
            }
        }
    }
    // the depth-optimised compilations the engine uses for batches that do not fill the chip (engine.hip: slp_strategy): fixed
    // splits per size, small-integer sums as balanced trees.  Each against the definition, as a plan and as its slot program.
    const int fixed_splits[][4] = {{2, 2, 2, 2}, {4, 2, 4, 2}, {2, 2, 2, 4}, {4, 2, 4, 8}, {4, 4, 4, 4}, {4, 8, 8, 8}};
    for (auto& f : fixed_splits) {
        for (int balanced = 0; balanced < 2; balanced++) {
            Strategy st_;
            st_.tuned = false;
            st_.balanced_lincomb = balanced != 0;
            st_.hankel_split = {{2, 2}, {4, f[0]}, {8, f[1]}, {16, f[2]}, {32, f[3]}};
            Plan p = build_fk20_proofs_plan(w, st_);
            Schedule S = make_schedule(p, false);
            if (balanced) printf("fixed splits 4->%d 8->%d 16->%d 32->%d, balanced sums: %ld mulc, %ld add/sub, %ld doublings, %zu launches, %d slots\n", f[0], f[1], f[2], f[3],
                                 p.count(OP_MULC), p.count(OP_ADD) + p.count(OP_SUB), p.doublings(), S.launches.size(), S.n_slots);
            for (int it = 0; it < 2; it++) {
                std::vector<Fr> in(128);
                for (auto& v : in) v = rnd_fr();
                if (it == 1) for (int j = 0; j < 128; j++) in[j] = j == 77 ? one<FrParams>() : zero<FrParams>();
                auto want = fk20_proofs_map_by_definition(w, in);
                auto got = run_over_fr(p, in);
                auto got2 = run_schedule_over_fr(S, p.consts, 128, 128, in);
                for (int k = 0; k < 128; k++) {
                    if (!eq(want[k], got[k])) { bad++; if (bad < 5) printf("MISMATCH fixed plan out %d\n", k); }
                    if (!eq(want[k], got2[k])) { bad++; if (bad < 5) printf("MISMATCH fixed schedule out %d\n", k); }
                }
            }
        }
    }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
