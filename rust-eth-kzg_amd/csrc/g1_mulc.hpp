// Multiplication of a G1 point by a PUBLIC constant that the host has recoded (engine.hip: recode_glv_wnaf):
// shared by the radix-2 G1-FFT layers (k_g1fft.hip) and the straight-line executor of the FK20 proofs map (k_g1slp.hip).
#pragma once
#include "kcommon.hpp"
#include "curve29.hpp"
#include "g1_coop.hpp"
#include "launch.hpp"

#ifndef G1_MULC_COZ
#define G1_MULC_COZ 1  // 0: the table by general additions and a product tree over their Z (round 2's form, kept for A/B builds)
#endif

namespace kzg {

// Data layout: X[pos * stride + lane], lane = blob index inside the batch (stride = batch padded to a
// multiple of 64), so all 64 lanes of a wave run the SAME butterfly and the twiddle is wave-uniform.
// `b * twiddle` (fft.rs:164-177) is a 255-bit scalar multiplication by a PUBLIC constant, so the scalar is
// recoded once on the host: GLV split k = k1 + k2*lambda (phi(x,y) = (beta x, y) = [lambda](x,y)), then each
// 128-bit half in width-w non-adjacent form (odd digits |d| < 2^(w-1), one non-zero digit in w+1 on average).
// The two digit streams share the <= 129 doublings; the additions take (2j+1) P from a small table built with
// one doubling and 2^(w-2) - 1 additions (phi of a table entry only swaps in beta x).  w = 5: ~43 additions + 8
// for the table, against ~66 for the joint sparse form and 255 doublings + ~85 additions for a plain NAF.
// Every digit test is a scalar branch on wave-uniform data: no lane divergence.
// tab[k][2][33] words = 2 x 132 signed bytes: digit t of half h is byte t of tab[k][h].
// row: the 2 x TWIDDLE_WORDS digit words of the constant (wave-uniform address)
// COOP = 4: the four lanes of a quad hold the same p and share the doublings and mixed additions of the digit loop (g1_coop.hpp:
// 3.5 and 5.5 multiplication times instead of 6.5 and 10.5); COOP = 2: the two lanes of a pair (4 and 5.5); `quad` is the lane's
// index in its quad / pair.  The table is built on every lane.
template <int COOP = 0>
__device__ __forceinline__ JacQ mul_by_recoded(const JacQ& p, const uint32_t* __restrict__ row, const Fq<1>& beta, int quad = -1) {
    constexpr int NT = 1 << (launch::TWIDDLE_WNAF_W - 2);  // odd multiples P, 3P, .., (2 NT - 1) P
    // The table is brought to ONE common Z = prod z_j without an inversion: (X_j l_j^2, Y_j l_j^3) with l_j = Z / z_j are
    // the affine coordinates of the same points on the isomorphic curve y^2 = x^3 + 4 Z^6.  The group law for a = 0
    // never looks at b, and phi(x, y) = (beta x, y) is an endomorphism of that curve too, so the whole multiplication
    // runs there with MIXED additions (6M + 3S + pair instead of 10M + 4S + pair) and one final Z <- Z * Z_common.
    AffQ2 A[NT];
    Fq<2> bx[NT];
    Fq<ZB> zc;
#if G1_MULC_COZ
    {
        // The odd multiples by co-Z arithmetic (Meloni's ZADDU: an addition of two points on ONE Z costs 4M + 2S and hands back
        // its first operand on the sum's Z): the doubling already holds P on the Z of 2P -- (4 X Y^2, 8 Y^4, 2 Y Z) -- and every
        // step (2j+1)P = 2P + (2j-1)P leaves the sum and 2P on Z_j = Z_(j-1) (X_2P - X_(2j-1)P).  Entry j is then lifted to the
        // last Z by lambda_j = product of the later differences: 72M + 27S for the table against 153M + 40S for seven general
        // additions and a product tree over their Z.  X_2P = X_(2j-1)P needs (2j+1)P or (2j-3)P = O: never in the prime-order
        // group; an identity input (Z = 0) keeps Z = 0 through zc whatever the other coordinates hold.
        Fq<XB> ex[NT], ey[NT];  // entry j on Z_j
        Fq<256> dl[NT];         // Z_j = Z_(j-1) dl[j], j >= 1
        const Fq<2> a = sqr(p.x), b = sqr(p.y), c = sqr(b);
        const Fq<8> d = dbl2(mul(p.x, b));
        const Fq<6> e = add(dbl(a), a);
        const auto x2 = sub2(sqr(e), d);
        Fq<XB> tx = relax<XB>(x2);                                                 // 2P, kept on the newest Z
        Fq<XB> ty = relax<XB>(mul_add(e, sub(d, x2), neg2(b), dbl2(b)));
        const Fq<ZB> z0 = dbl(mul(p.y, p.z));
        ex[0] = relax<XB>(d);
        ey[0] = relax<XB>(dbl2(dbl(c)));
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            const auto dx = sub(tx, ex[j - 1]), dy = sub(ty, ey[j - 1]);
            const Fq<2> cc = sqr(dx);
            const Fq<2> w1 = mul(tx, cc), w2 = mul(ex[j - 1], cc);
            const auto x3 = sub(sub(sqr(dy), w1), w2);
            const Fq<2> a1 = mul(ty, sub(w1, w2));
            ex[j] = relax<XB>(x3);
            ey[j] = relax<XB>(sub(mul(dy, sub(w1, x3)), a1));
            tx = relax<XB>(w1);
            ty = relax<XB>(a1);
            dl[j] = relax<256>(dx);
        }
        Fq<256> lam = relax<256>(fq_one());  // Z_(NT-1) / Z_j
#pragma unroll 1
        for (int j = NT - 1; j >= 0; j--) {
            const Fq<2> l2 = sqr(lam);
            A[j].x = mul(ex[j], l2);
            A[j].y = mul(ey[j], mul(l2, lam));
            bx[j] = mul(A[j].x, beta);
            if (j > 0) lam = relax<256>(mul(lam, dl[j]));
        }
        zc = relax<ZB>(mul(z0, lam));
    }
#else
    {
        JacQ T[NT];
        Fq<ZB> pre[NT];  // pre[j] = z_0 ... z_j
        const JacQ p2 = dbl(p);
        T[0] = p;
        pre[0] = p.z;
#pragma unroll 1
        for (int j = 1; j < NT; j++) {
            T[j] = add(T[j - 1], p2);
            pre[j] = relax<ZB>(mul(pre[j - 1], T[j].z));
        }
        zc = pre[NT - 1];
        Fq<ZB> suf = relax<ZB>(fq_one());  // z_(j+1) ... z_(NT-1)
#pragma unroll 1
        for (int j = NT - 1; j >= 0; j--) {
            const Fq<ZB> lam = j > 0 ? relax<ZB>(mul(pre[j > 0 ? j - 1 : 0], suf)) : suf;  // product of all the other z
            const Fq<2> l2 = sqr(lam);
            A[j].x = mul(T[j].x, l2);
            A[j].y = mul(T[j].y, mul(l2, lam));
            bx[j] = mul(A[j].x, beta);
            suf = relax<ZB>(mul(suf, T[j].z));
        }
    }
#endif
    JacQ acc = jacq_inf();
    bool started = false;
#pragma unroll 1
    for (int wd = launch::TWIDDLE_WORDS - 1; wd >= 0; wd--) {
        const uint32_t w1 = __builtin_amdgcn_readfirstlane(row[wd]);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(row[launch::TWIDDLE_WORDS + wd]);
        if (!started && (w1 | w2) == 0) continue;
#pragma unroll 1
        for (int q = 3; q >= 0; q--) {
            if (started) {
                if constexpr (COOP == 4) acc = coop_dbl(acc, quad);
                else if constexpr (COOP == 2) acc = coop2_dbl(acc, quad);
                else acc = dbl(acc);
            }
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                const int d = (int)(int8_t)((h ? w2 : w1) >> (8 * q));
                if (d == 0) continue;
                const int idx = ((d < 0 ? -d : d) - 1) >> 1;
                AffQ2 op = A[idx];
                if (h) op.x = bx[idx];
                if (!started) {
                    acc.x = relax<XB>(op.x);
                    acc.y = d < 0 ? relax<XB>(neg(op.y)) : relax<XB>(op.y);
                    acc.z = relax<ZB>(fq_one());
                    started = true;
                } else if constexpr (COOP == 4) acc = coop_add_mixed(acc, op, d < 0, quad);
                else if constexpr (COOP == 2) acc = coop2_add_mixed(acc, op, d < 0, quad);
                else acc = add_mixed(acc, op, d < 0);
            }
        }
    }
    acc.z = relax<ZB>(mul(acc.z, zc));  // back from the isomorphic curve
    return acc;
}

// twiddle k of the 128-point transforms: tab[k] is the recoding of omega_128^k
__device__ __forceinline__ JacQ mul_by_twiddle(const JacQ& p, const uint32_t* __restrict__ tab, const Fq<1>& beta, int k) {
    // k is wave-uniform; 0 -> identity map, 64 -> negation (omega_128^64 = -1)
    if (k == 0) return p;
    if (k == 64) return neg(p);
    return mul_by_recoded(p, tab + (size_t)k * (2 * launch::TWIDDLE_WORDS), beta);
}

}  // namespace kzg
