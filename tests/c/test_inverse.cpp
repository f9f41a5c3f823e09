// CPU check of csrc/inverse.hpp (binary-GCD inversion) against Fermat's a^(p-2) from csrc/field.hpp.
// Built and run by tests/test_host_units.py:  hipcc -x hip --offload-arch=gfx950 (host pass only; no kernel is launched).
#include "inverse.hpp"
#include <cstdio>
#include <cstdlib>
using namespace kzg;

template <class P>
static int run(const char* name, int iters) {
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        Felt<P> a;
        for (int i = 0; i < P::N; i++) a.v[i] = next();
        if (it < 40) {  // small and structured values first
            for (int i = 0; i < P::N; i++) a.v[i] = 0;
            a.v[0] = it + 1;
            if (it >= 20) a.v[(it - 20) % P::N] = 0x80000000u >> (it % 7);
            if (it == 39) for (int i = 0; i < P::N; i++) a.v[i] = P::MOD[i] - (i == 0);  // m - 1
        }
        a.v[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;  // < 2^(BITS-1) < m
        if (is_zero(a)) continue;
        Felt<P> x = inv(a), y = inv_fast(a);
        if (!eq(x, y) || !eq(mul(a, y), one<P>())) bad++;
    }
    Felt<P> z = zero<P>();
    if (!is_zero(inv_fast(z))) bad++;
    printf("%s: %d cases, %d mismatches\n", name, iters, bad);
    return bad;
}
int main() {
    int bad = run<FpParams>("Fp", 3000) + run<FrParams>("Fr", 3000);
    return bad ? 1 : 0;
}
