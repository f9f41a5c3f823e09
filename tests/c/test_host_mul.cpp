// CPU check of the 64-bit host multiplication (csrc/field.hpp: mul_host64, used by the pairing check and set-up) against
// the 32-bit CIOS form, on 20000 random pairs per field plus (p-1)^2.  Built and run by tests/test_host_units.py.
#include "field.hpp"
#include <cstdio>
using namespace kzg;
template <class P> int run() {
    uint64_t st = 88172645463325252ull; int bad = 0;
    for (int it = 0; it < 20000; it++) {
        Felt<P> a, b;
        for (int i = 0; i < P::N; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a.v[i] = (uint32_t)st; st ^= st << 13; st ^= st >> 7; st ^= st << 17; b.v[i] = (uint32_t)(st >> 9); }
        a.v[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1; b.v[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;
        if (it == 0) { for (int i = 0; i < P::N; i++) { a.v[i] = P::MOD[i]; b.v[i] = P::MOD[i]; } a.v[0] -= 1; b.v[0] -= 1; }
        if (!eq(mul_host64(a, b), mul_cios(a, b))) bad++;
    }
    return bad;
}
int main() { int b = run<FpParams>() + run<FrParams>(); printf("mul_host64 vs mul_cios: %d mismatches\n", b); return b != 0; }
