// The radix-4 units of the 4096-point transforms of k_ntt.hip, as host + device code: the kernels run them on LDS, one unit per
// thread and pass; tests/c/test_fr29.cpp runs the same functions on a vector against the definition of the DFT and against the
// radix-2 network.  Elems: load(i) / store(i, value) over the 4096 elements.
#pragma once
#include "fr29.hpp"

namespace kzg {

constexpr int NTT_N = 4096, NTT_W = 8192;  // elements; w29[k] = omega_8192^k (9 x 29-bit Montgomery form, canonical)

// w^e b, < 2r; e = 0: no product, the value is only brought below 2r
HD Fr29 fr29_twiddled(const Fr29& b, const Fr29* __restrict__ w29, int e) {
    return e ? fr29_mul(b, w29[e]) : fr29_partial_reduce(b);
}
// Inverse transform (bit-reversed in, natural out, twiddles omega^-j), pass h = the radix-2 layers half = h and 2h, h = 1, 4, ..., 1024:
// unit u = blk h + j (j < h) works on the elements blk 4h + j + {0, h, 2h, 3h}.
template <class Elems>
HD void ntt4096_dit_inverse_unit(const Elems& m, const Fr29* __restrict__ w29, int h, int u) {
    const int j = u & (h - 1), i0 = ((u - j) << 2) + j;
    const int step_a = NTT_W / (2 * h), step_b = step_a >> 1;  // exponent steps of the two layers in units of omega_8192
    const int ea = (NTT_W - j * step_a) & (NTT_W - 1), eb0 = (NTT_W - j * step_b) & (NTT_W - 1), eb1 = NTT_W - (j + h) * step_b;
    const Fr29 x0 = m.load(i0), x1 = m.load(i0 + h), x2 = m.load(i0 + 2 * h), x3 = m.load(i0 + 3 * h);
    const Fr29 ta = fr29_twiddled(x1, w29, ea), tc = fr29_twiddled(x3, w29, ea);
    // layer half = h: sums and differences stay un-swept (LAZY LIMBS, fr29.hpp): the second layer is what reads them
    const Fr29 a0 = fr29_add<false>(x0, ta), a1 = fr29_sub2r<false>(x0, ta), a2 = fr29_add<false>(x2, tc), a3 = fr29_sub2r<false>(x2, tc);
    const Fr29 tb0 = fr29_twiddled(a2, w29, eb0), tb1 = fr29_mul(a3, w29[eb1]);  // (j + h) step_b is never a multiple of 8192
    m.store(i0, fr29_add<true>(a0, tb0));
    m.store(i0 + 2 * h, fr29_sub2r<true>(a0, tb0));
    m.store(i0 + h, fr29_add<true>(a1, tb1));
    m.store(i0 + 3 * h, fr29_sub2r<true>(a1, tb1));
}
// Forward transform (natural in, bit-reversed out, Cooley-Tukey butterflies with bit-reversed twiddles), pass h = the layers half = 2h
// and h, h = 1024, 256, ..., 1; log_m = log2(1024 / h): there are 2^log_m blocks of 4h elements.
template <class Elems>
HD void ntt4096_ct_forward_unit(const Elems& m, const Fr29* __restrict__ w29, int h, int log_m, int u) {
    const int j = u & (h - 1), blk = u / h, i0 = ((u - j) << 2) + j;
    unsigned rb = 0;  // brp of the 4h-block's index over log_m bits
    for (int b = 0; b < log_m; b++) rb |= (((unsigned)blk >> b) & 1u) << (log_m - 1 - b);
    const int ea = (int)rb * (2 * h) * (NTT_W / NTT_N);  // layer half = 2h: block blk of 2^log_m
    // layer half = h: blocks 2 blk and 2 blk + 1 of 2^(log_m + 1): brp(2 blk) = brp(blk), brp(2 blk + 1) = brp(blk) + 2^log_m
    const int eb0 = (int)rb * h * (NTT_W / NTT_N), eb1 = (int)(rb + (1u << log_m)) * h * (NTT_W / NTT_N);
    const Fr29 x0 = m.load(i0), x1 = m.load(i0 + h), x2 = m.load(i0 + 2 * h), x3 = m.load(i0 + 3 * h);
    const Fr29 ta = fr29_twiddled(x2, w29, ea), tc = fr29_twiddled(x3, w29, ea);
    const Fr29 a0 = fr29_add<false>(x0, ta), a2 = fr29_sub2r<false>(x0, ta), a1 = fr29_add<false>(x1, tc), a3 = fr29_sub2r<false>(x1, tc);
    const Fr29 tb0 = fr29_twiddled(a1, w29, eb0), tb1 = fr29_mul(a3, w29[eb1]);  // block 2 blk + 1 is never block 0
    m.store(i0, fr29_add<true>(a0, tb0));
    m.store(i0 + h, fr29_sub2r<true>(a0, tb0));
    m.store(i0 + 2 * h, fr29_add<true>(a2, tb1));
    m.store(i0 + 3 * h, fr29_sub2r<true>(a2, tb1));
}

}  // namespace kzg
