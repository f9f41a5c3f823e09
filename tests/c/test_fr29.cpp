// CPU check of csrc/fr29.hpp (the unsaturated 9 x 29-bit Fr of the prover's NTT kernels) against the saturated 8 x 32-bit
// Montgomery arithmetic of csrc/field.hpp: products at every bound the kernels use, the re-grouping between the two limb
// layouts, the reduction to canonical form, and a Cooley-Tukey transform with bit-reversed twiddles (the network of
// ntt4096_ct_forward) against the definition of the DFT -- including the claim that 12 layers need no reduction.
// Built and run by tests/test_host_units.py with hipcc's host pass (no kernel is launched).
#include "fr29.hpp"
#include "fr29_ntt.hpp"
#include <cstdio>
#include <vector>
using namespace kzg;

static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint32_t rnd32() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static Fr rnd_fr() {  // canonical value < r (as a plain integer)
    Fr a;
    for (int i = 0; i < 8; i++) a.v[i] = rnd32();
    a.v[7] &= 0x3fffffffu;
    return a;
}
static Fr small(uint32_t v) { Fr a = zero<FrParams>(); a.v[0] = v; return a; }
// X form (canonical) of the field element whose saturated Montgomery form is y
static Fr29 x_of(const Fr& y_mont) { return fr29_from_plain(mul(y_mont, to_mont(small(32)))); }
static bool same(const Fr29& a, const Fr29& b) { for (int i = 0; i < RL; i++) if (a.v[i] != b.v[i]) return false; return true; }
static Fr29 canon(const Fr29& lt2r) { return fr29_reduce_once(lt2r); }
// k * a in the 29-bit form without reduction (value < k * bound(a) * r)
static Fr29 times(const Fr29& a, int k) { Fr29 r = a; for (int i = 1; i < k; i++) r = fr29_add(r, a); return r; }

int main() {
    int bad = 0;
    // re-grouping round trips and the shift-by-5 entry
    for (int it = 0; it < 20000; it++) {
        Fr x = rnd_fr();
        if (it == 0) x = zero<FrParams>();
        if (it == 1) { for (int i = 0; i < 8; i++) x.v[i] = FrParams::MOD[i]; x.v[0] -= 1; }  // r - 1
        Fr back;
        fr29_to_words(back.v, fr29_from_plain(x));
        if (!eq(back, x)) { bad++; if (bad < 5) printf("MISMATCH regroup\n"); }
        // (x << 5) = 32 x as an integer: compare with 32 additions
        Fr29 a = fr29_from_plain(x), s32 = times(a, 32), sh = fr29_from_fr_mont(x);
        if (!same(s32, sh)) { bad++; if (bad < 5) printf("MISMATCH shift-by-5 entry\n"); }
    }
    // products: (bound up to 56) x (canonical) against the saturated multiplication
    for (int it = 0; it < 20000; it++) {
        const Fr ya = to_mont(rnd_fr()), yb = to_mont(rnd_fr());
        const Fr yab = mul(ya, yb);
        const Fr29 want = x_of(yab);
        const Fr29 xb = x_of(yb);
        // entry form of a (value 32 Y < 32 r), then pushed up to 56 r by adding canonical copies (what 12 layers can do)
        Fr29 xa = fr29_from_fr_mont(ya);
        if (!same(canon(fr29_mul(xa, xb)), want)) { bad++; if (bad < 5) printf("MISMATCH mul bound 32\n"); }
        Fr29 xa1 = x_of(ya);  // canonical X form: same element
        if (!same(canon(fr29_mul(xa1, xb)), want)) { bad++; if (bad < 5) printf("MISMATCH mul canonical\n"); }
        // a + 24 r-ish: add 24 copies of the canonical form of zero-equivalent?  use a + 12 * (b + (r - b)) = a + 12 r' where every addend is < r
        Fr nb = neg(yb);
        Fr29 grown = xa;
        for (int k = 0; k < 12; k++) { grown = fr29_add(grown, x_of(yb)); grown = fr29_add(grown, x_of(nb)); }  // + 12 r exactly, value < 56 r
        if (!same(canon(fr29_mul(grown, xb)), want)) { bad++; if (bad < 5) printf("MISMATCH mul bound 56\n"); }
        // subtraction form: a - t + 2r with t a fresh product
        const Fr29 t = fr29_mul(xa1, xb);                       // a b in X form, < 2r
        const Fr29 d = fr29_sub2r(xa1, t);                      // a - ab + 2r
        const Fr29 dd = canon(fr29_mul(d, fr29_const(r29::ONE_PLAIN)));  // strip 2^261: the plain value of a - ab
        Fr plain;
        fr29_to_words(plain.v, dd);
        if (!eq(plain, from_mont(sub(ya, yab)))) { bad++; if (bad < 5) printf("MISMATCH sub2r\n"); }
        // canonical -> this form through R2
        Fr pa = from_mont(ya);
        if (!same(canon(fr29_mul(fr29_from_plain(pa), fr29_const(r29::R2))), xa1)) { bad++; if (bad < 5) printf("MISMATCH to_mont\n"); }
    }
    // partial reduction without a multiplication: same residue, below 2r, over the whole range a value can have
    for (int it = 0; it < 200000; it++) {
        Fr29 b;
        for (int i = 0; i < RL; i++) b.v[i] = rnd32() & RMASK;
        if (it < 64) { b = times(x_of(to_mont(rnd_fr())), it + 1); }                       // multiples of a canonical value: bounds 1 .. 64
        if (it == 64) for (int i = 0; i < RL; i++) b.v[i] = RMASK;                           // 2^261 - 1
        if (it == 65) for (int i = 0; i < RL; i++) b.v[i] = 0;
        const Fr29 red = fr29_partial_reduce(b);
        for (int i = 0; i < RL - 1; i++) if (red.v[i] > RMASK) { bad++; if (bad < 5) printf("MISMATCH partial_reduce limbs\n"); }
        // below 2r: one conditional subtraction must reach the canonical form; and the residue is the same (compare plain values)
        const Fr29 one = fr29_const(r29::ONE_PLAIN);
        const Fr29 c1 = canon(red);
        Fr29 chk = fr29_reduce_once(c1);
        if (!same(chk, c1)) { bad++; if (bad < 5) printf("MISMATCH partial_reduce not below 2r\n"); }
        if (!same(canon(fr29_mul(red, one)), canon(fr29_mul(b, one)))) { bad++; if (bad < 5) printf("MISMATCH partial_reduce residue\n"); }
    }
    // Cooley-Tukey natural -> bit-reversed with bit-reversed twiddles, n = 4096 (12 layers, entry bound 32, no reduction)
    {
        const int n = 4096, logn = 12;
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
        e[0] -= 1;
        for (int s = 0; s < logn; s++) for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
        Fr g = one<FrParams>(), base = to_mont(small(7));
        for (int i = 255; i >= 0; i--) { g = sqr(g); if ((e[i >> 5] >> (i & 31)) & 1) g = mul(g, base); }
        std::vector<Fr> w(n);
        w[0] = one<FrParams>();
        for (int i = 1; i < n; i++) w[i] = mul(w[i - 1], g);
        std::vector<Fr29> w29(n);
        for (int i = 0; i < n; i++) w29[i] = x_of(w[i]);
        std::vector<Fr> in(n);
        for (auto& v : in) v = to_mont(rnd_fr());
        std::vector<Fr29> x(n);
        for (int i = 0; i < n; i++) x[i] = fr29_from_fr_mont(in[i]);  // < 32 r
        int log_m = 0;
        for (int half = n / 2; half >= 1; half >>= 1, log_m++)
            for (int q = 0; q < n / 2; q++) {
                const int j = q & (half - 1), blk = q / half, i0 = ((q - j) << 1) + j, i1 = i0 + half;
                int br = 0;
                for (int b = 0; b < log_m; b++) br |= ((blk >> b) & 1) << (log_m - 1 - b);
                const Fr29 t = br ? fr29_mul(x[i1], w29[(size_t)br * half]) : fr29_partial_reduce(x[i1]);
                const Fr29 a = x[i0];
                x[i0] = fr29_add(a, t);
                x[i1] = fr29_sub2r(a, t);
            }
        // compare a sample of outputs with the definition
        for (int k : {0, 1, 2, 77, 2048, 4095}) {
            Fr acc = zero<FrParams>();
            for (int j = 0; j < n; j++) acc = add(acc, mul(in[j], w[(size_t)j * k % n]));
            int q = 0;
            for (int b = 0; b < logn; b++) q |= ((k >> b) & 1) << (logn - 1 - b);
            Fr got;
            fr29_to_words(got.v, canon(fr29_mul(x[q], fr29_const(r29::ONE_PLAIN))));
            if (!eq(got, from_mont(acc))) { bad++; if (bad < 8) printf("MISMATCH ntt output %d\n", k); }
        }
        // the top limb never ran out of room
        for (int i = 0; i < n; i++) if (x[i].v[RL - 1] >> 29) { bad++; if (bad < 8) printf("OVERFLOW top limb\n"); break; }
        // LAZY LIMBS (fr29.hpp): the same transform as the kernels run it -- the layers alternate, lazy first, the carries swept
        // only in every second layer -- must give the same residues in every position, and its limbs must stay inside the bounds the
        // header states (checked in 64 bits: a limb-wise sum that wrapped would still compare equal mod 2^32).  Twice: random
        // inputs, and inputs with every limb at its maximum.
        for (int worst = 0; worst < 2; worst++) {
            std::vector<Fr29> y(n), z(n);
            for (int i = 0; i < n; i++) {
                y[i] = fr29_from_fr_mont(in[i]);
                if (worst) { for (int l = 0; l < RL - 1; l++) y[i].v[l] = RMASK; y[i].v[RL - 1] = (32u << 24) - 1; }  // < 32 r, limbs full
                z[i] = y[i];
            }
            uint64_t max_lazy = 0, max_presweep = 0;
            log_m = 0;
            for (int half = n / 2; half >= 1; half >>= 1, log_m++) {
                const bool norm = log_m & 1;
                for (int q = 0; q < n / 2; q++) {
                    const int j = q & (half - 1), blk = q / half, i0 = ((q - j) << 1) + j, i1 = i0 + half;
                    int br = 0;
                    for (int b = 0; b < log_m; b++) br |= ((blk >> b) & 1) << (log_m - 1 - b);
                    const Fr29 tz = br ? fr29_mul(z[i1], w29[(size_t)br * half]) : fr29_partial_reduce(z[i1]);
                    const Fr29 az = z[i0];
                    z[i0] = fr29_add(az, tz);
                    z[i1] = fr29_sub2r(az, tz);
                    const Fr29 t = br ? fr29_mul(y[i1], w29[(size_t)br * half]) : fr29_partial_reduce(y[i1]);
                    const Fr29 a = y[i0];
                    for (int l = 0; l < RL - 1; l++) {
                        const uint64_t sum = (uint64_t)a.v[l] + t.v[l], dif = (uint64_t)a.v[l] + r29::SUBK[0][l] - t.v[l];
                        uint64_t& m = norm ? max_presweep : max_lazy;
                        if (sum > m) m = sum;
                        if (dif > m) m = dif;
                    }
                    if (norm) { y[i0] = fr29_add<true>(a, t); y[i1] = fr29_sub2r<true>(a, t); }
                    else { y[i0] = fr29_add<false>(a, t); y[i1] = fr29_sub2r<false>(a, t); }
                }
            }
            if (max_lazy >= (4ull << 29) || max_presweep >= (7ull << 29)) { bad++; printf("LAZY LIMBS out of bounds: %llx %llx\n", (unsigned long long)max_lazy, (unsigned long long)max_presweep); }
            const Fr29 one_p = fr29_const(r29::ONE_PLAIN);
            for (int i = 0; i < n; i++)
                if (!same(canon(fr29_mul(y[i], one_p)), canon(fr29_mul(z[i], one_p)))) { bad++; if (bad < 8) printf("MISMATCH lazy transform at %d (worst %d)\n", i, worst); }
            for (int i = 0; i < n; i++) for (int l = 0; l < RL - 1; l++) if (y[i].v[l] > RMASK) { bad++; if (bad < 8) printf("lazy transform: output limbs not swept\n"); i = n; break; }
            if (worst == 0) printf("lazy limbs: largest stored %.3f x 2^30, largest before a sweep %.3f x 2^30\n", max_lazy / 1073741824.0, max_presweep / 1073741824.0);
        }
    }
    // The radix-4 units the kernels run (csrc/fr29_ntt.hpp: one unit per thread and pass) on a vector: forward = the radix-2 network
    // above in every position (bit-identical limbs: the same operations in the same order), inverse (bit-reversed in, natural out,
    // omega^-j) against the definition; forward then inverse = n times the input.
    {
        const int n = 4096, logn = 12;
        struct VecElems {
            std::vector<Fr29>* v;
            Fr29 load(int i) const { return (*v)[i]; }
            void store(int i, const Fr29& x) const { (*v)[i] = x; }
        };
        // omega_8192 and its powers in this form
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = FrParams::MOD[i];
        e[0] -= 1;
        for (int s = 0; s < 13; s++) for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
        Fr g = one<FrParams>(), base = to_mont(small(7));
        for (int i = 255; i >= 0; i--) { g = sqr(g); if ((e[i >> 5] >> (i & 31)) & 1) g = mul(g, base); }
        std::vector<Fr> w(8192);
        w[0] = one<FrParams>();
        for (int i = 1; i < 8192; i++) w[i] = mul(w[i - 1], g);
        std::vector<Fr29> w29(8192);
        for (int i = 0; i < 8192; i++) w29[i] = x_of(w[i]);
        std::vector<Fr> in(n);
        for (auto& v : in) v = to_mont(rnd_fr());
        auto brp = [&](int k) { int q = 0; for (int b = 0; b < logn; b++) q |= ((k >> b) & 1) << (logn - 1 - b); return q; };
        // forward, radix 4
        std::vector<Fr29> y(n), z(n);
        for (int i = 0; i < n; i++) y[i] = z[i] = fr29_from_fr_mont(in[i]);
        {
            int log_m = 0;
            for (int h = 1024; h >= 1; h >>= 2, log_m += 2)
                for (int u = 0; u < 1024; u++) ntt4096_ct_forward_unit(VecElems{&y}, w29.data(), h, log_m, u);
        }
        // forward, radix 2 with the kernels' former layer schedule (lazy, swept, lazy, ...)
        {
            int log_m = 0;
            for (int half = n / 2; half >= 1; half >>= 1, log_m++)
                for (int q = 0; q < n / 2; q++) {
                    const int j = q & (half - 1), blk = q / half, i0 = ((q - j) << 1) + j, i1 = i0 + half;
                    int br = 0;
                    for (int b = 0; b < log_m; b++) br |= ((blk >> b) & 1) << (log_m - 1 - b);
                    const Fr29 t = fr29_twiddled(z[i1], w29.data(), br * half * 2);
                    const Fr29 a = z[i0];
                    if (log_m & 1) { z[i0] = fr29_add<true>(a, t); z[i1] = fr29_sub2r<true>(a, t); }
                    else { z[i0] = fr29_add<false>(a, t); z[i1] = fr29_sub2r<false>(a, t); }
                }
        }
        for (int i = 0; i < n; i++) if (!same(y[i], z[i])) { bad++; if (bad < 8) printf("MISMATCH radix-4 forward differs from radix-2 at %d\n", i); }
        for (int k : {0, 1, 2, 77, 2048, 4095}) {
            Fr acc = zero<FrParams>();
            for (int j = 0; j < n; j++) acc = add(acc, mul(in[j], w[(size_t)2 * ((size_t)j * k % n)]));
            Fr got;
            fr29_to_words(got.v, canon(fr29_mul(y[brp(k)], fr29_const(r29::ONE_PLAIN))));
            if (!eq(got, from_mont(acc))) { bad++; if (bad < 8) printf("MISMATCH radix-4 forward output %d\n", k); }
        }
        // inverse, radix 4, on the forward transform's output (bit-reversed, as the network wants it): n times the input
        for (int i = 0; i < n; i++) y[i] = fr29_partial_reduce(y[i]);  // (< 2r again: the kernels never chain two transforms without a product)
        for (int h = 1; h <= 1024; h <<= 2)
            for (int u = 0; u < 1024; u++) ntt4096_dit_inverse_unit(VecElems{&y}, w29.data(), h, u);
        const Fr n_mont = to_mont(small(4096));
        for (int i = 0; i < n; i++) {
            Fr got;
            fr29_to_words(got.v, canon(fr29_mul(y[i], fr29_const(r29::ONE_PLAIN))));
            if (!eq(got, from_mont(mul(in[i], n_mont)))) { bad++; if (bad < 8) printf("MISMATCH radix-4 inverse(forward(x)) != n x at %d\n", i); }
        }
        // and the inverse alone against its definition on a fresh bit-reversed input
        std::vector<Fr> in2(n);
        for (auto& v : in2) v = to_mont(rnd_fr());
        for (int i = 0; i < n; i++) y[i] = fr29_from_fr_mont(in2[i]);  // position i holds x[brp(i)]
        for (int h = 1; h <= 1024; h <<= 2)
            for (int u = 0; u < 1024; u++) ntt4096_dit_inverse_unit(VecElems{&y}, w29.data(), h, u);
        for (int k : {0, 1, 3, 64, 1023, 4095}) {
            Fr acc = zero<FrParams>();
            for (int j = 0; j < n; j++) acc = add(acc, mul(in2[brp(j)], w[(size_t)(8192 - 2 * ((size_t)j * k % n)) % 8192]));
            Fr got;
            fr29_to_words(got.v, canon(fr29_mul(y[k], fr29_const(r29::ONE_PLAIN))));
            if (!eq(got, from_mont(acc))) { bad++; if (bad < 8) printf("MISMATCH radix-4 inverse output %d\n", k); }
        }
    }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
