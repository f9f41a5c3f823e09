// Multiplication of a G1 point by a PUBLIC, host-recoded constant in the signed 13 x 30-bit field: the form k_slp_mulc runs
// for batches that fill the chip (83 % of the G1 linear map's time).  Same algorithm as g1_mulc.hpp -- GLV halves in width-5
// NAF over the 8 odd multiples, the table built by co-Z additions and brought to ONE common Z (the isomorphic curve
// y^2 = x^3 + 4 Z^6) so that every addition of the digit loop is a mixed one -- with the chain written for fp30.hpp:
//   * doublings in the halved form (curve30.hpp: dbl_half: 2,054 multiply-adds instead of 2,317, no constant multiples),
//   * mixed additions with the subtractions fused into the reductions (3,497 instead of 3,941),
//   * the point is read from and written to the arena as it is when the arena holds the signed form (batches of more than one
//     lane group); an arena in the 14 x 29-bit form (JacQ) costs six products per multiplication for the way in and out.
// Reference work being replaced: the blst scalar multiplication behind `b * twiddle` of fft.rs:164-177.
#pragma once
#include "kcommon.hpp"
#include "curve30.hpp"
#include "g1_coop30.hpp"
#include "launch.hpp"

namespace kzg {

// COOP = 2 / 4: the lanes of a pair / quad hold the same p and share the doublings and mixed additions of the digit loop
// (g1_coop30.hpp; part = this lane's place in its pair / quad); the table is built on every lane.
template <int COOP = 0>
__device__ __forceinline__ JacS mul_by_recoded30(const JacS& p, const uint32_t* __restrict__ row, const Fs<1, DC>& beta, int part = 0) {
    static_assert(COOP == 0 || COOP == 2 || COOP == 4, "one lane, a pair or a quad per multiplication");
    constexpr int NT = 1 << (launch::TWIDDLE_WNAF_W - 2);  // odd multiples P, 3P, .., (2 NT - 1) P
    AffT A[NT];
    Fs<1, DC> bx[NT];
    Fs<1, DC> zc;
    {
        // the odd multiples by co-Z additions (Meloni's ZADDU), see g1_mulc.hpp; everything scaled by the doubling's lambda = 1/2:
        // 2P = (H^2 - 2 M, H (M - X2) - B^2, Y Z) and P on the same Z = (X Y^2, Y^4) = (M, B^2)
        Fs<4, DC> ex[NT];  // entry j on Z_j
        Fs<2, DC> ey[NT];
        Fs<8, DC> dl[NT];  // Z_j = Z_(j-1) dl[j], j >= 1
        const Fs<1, DC> a = sqr(p.x), b = sqr(p.y);
        const Fs<1, DC> m = mul(b, p.x), bb = sqr(b);
        const Fs<2, DC> h = half_of_triple(a);
        const auto x2 = sqr_inj<-2, DC>(h, m);                                  // <= 3
        Fs<4, DC> tx = relax<4, DC>(x2);                                         // 2P, kept on the newest Z
        Fs<2, DC> ty = mul_inj<-1, DC>(h, sub_lazy(m, x2), bb);                  // H (M - X2) - B^2: <= 2
        const Fs<1, DC> z0 = mul(p.y, p.z);
        ex[0] = relax<4, DC>(m);
        ey[0] = relax<2, DC>(bb);
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            const Fs<8, DC> dx = sub(tx, ex[j - 1]);
            const Fs<4, DC> dy = sub(ty, ey[j - 1]);
            const Fs<1, DC> cc = sqr(dx);
            const Fs<1, DC> w1 = mul(cc, tx), w2 = mul(cc, ex[j - 1]);
            const auto x3 = sqr_inj2<-1, -1, DC>(dy, w1, w2);                   // <= 3
            const Fs<1, DC> a1 = mul(ty, sub_lazy(w1, w2));
            ey[j] = mul_inj<-1, DC>(dy, sub_lazy(w1, x3), a1);                  // <= 2
            ex[j] = relax<4, DC>(x3);
            tx = relax<4, DC>(w1);
            ty = relax<2, DC>(a1);
            dl[j] = dx;
        }
        Fs<1, DC> lam = fs_one();  // Z_(NT-1) / Z_j
#pragma unroll 1
        for (int j = NT - 1; j >= 0; j--) {
            const Fs<1, DC> l2 = sqr(lam);
            A[j].x = mul(ex[j], l2);
            A[j].y = mul(ey[j], mul(l2, lam));
#ifndef MULC30_NO_BX
            bx[j] = mul(A[j].x, beta);
#endif
            if (j > 0) lam = mul(lam, dl[j]);
        }
        zc = mul(z0, lam);
    }
    JacS acc = jacs_inf();
    bool started = false;
#pragma unroll 1
    for (int wd = launch::TWIDDLE_WORDS - 1; wd >= 0; wd--) {
        const uint32_t w1 = __builtin_amdgcn_readfirstlane(row[wd]);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(row[launch::TWIDDLE_WORDS + wd]);
        if (!started && (w1 | w2) == 0) continue;
#pragma unroll 1
        for (int q = 3; q >= 0; q--) {
            if (started) {
                if constexpr (COOP == 4) acc = coop4_dbl_half(acc, part);
                else if constexpr (COOP == 2) acc = coop2_dbl_half(acc, part == 0);
                else acc = dbl_half(acc);
            }
#pragma unroll 1
            for (int hf = 0; hf < 2; hf++) {
                const int d = (int)(int8_t)((hf ? w2 : w1) >> (8 * q));
                if (d == 0) continue;
#ifdef MULC30_IDX0_EXPERIMENT  // TIMING ONLY (wrong results): every digit reads entry 0 -- what the table's read traffic costs
                const int idx = 0;
#else
                const int idx = ((d < 0 ? -d : d) - 1) >> 1;
#endif
                AffT op = A[idx];
#ifdef MULC30_NO_BX  // EXPERIMENT: beta x computed per use instead of a second x-table in scratch (a third less table, +351 multiply-adds per k2 digit)
                if (hf) op.x = mul(op.x, beta);
#else
                if (hf) op.x = bx[idx];
#endif
                if (!started) {
                    acc.x = relax<4, DC>(op.x);
                    acc.y = cneg(d < 0, op.y);
                    acc.z = fs_one();
                    started = true;
                } else if constexpr (COOP == 4) acc = coop4_add_mixed(acc, op, d < 0, part);
                else if constexpr (COOP == 2) acc = coop2_add_mixed(acc, op, d < 0, part == 0);
                else acc = add_mixed(acc, op, d < 0);
            }
        }
    }
    acc.z = mul(acc.z, zc);  // back from the isomorphic curve
    return acc;
}
// the same on an arena in the 14 x 29-bit form: six products for the way in and out
template <int COOP = 0>
__device__ __forceinline__ JacQ mul_by_recoded30(const JacQ& pq, const uint32_t* __restrict__ row, const Fs<1, DC>& beta, int part = 0) {
    return jacq_from_jacs(mul_by_recoded30<COOP>(jacs_from_jacq(pq), row, beta, part));
}

}  // namespace kzg
