// Host-side optimal-ate pairing check for the cell-proof batch verifier.
// Replaces blstrs' multi_miller_loop / final_exponentiation / G2Prepared as used by the reference
// (crates/cryptography/bls12_381/src/lib.rs:45-50; crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:88-90,251-259).
// Two pairings per verification with two FIXED G2 points, so (like G2Prepared) the line coefficients
// of each G2 point are computed once at context creation; a check is then a Miller loop over the
// stored lines plus one final exponentiation -- a few ms of host time, constant in the batch size.
#pragma once
#include <cstdint>
#include <vector>
#include "curve.hpp"

namespace kzg {
namespace pairing {

struct Fp2 { Fp c0, c1; };                 // c0 + c1 u, u^2 = -1
struct Fp6 { Fp2 c0, c1, c2; };            // over Fp2, v^3 = 1 + u
struct Fp12 { Fp6 c0, c1; };               // over Fp6, w^2 = v
struct G2Affine { Fp2 x, y; bool inf; };
struct Line { Fp2 a, b; };                 // l(P) * w^3 = a + (b * xP) v + yP (v w)
struct G2Prepared { std::vector<Line> lines; bool inf = true; };

void init();                                               // Frobenius constants (idempotent)
bool g2_decompress(G2Affine& out, const uint8_t in[96]);   // ZCash encoding, on-curve check only
G2Affine g2_neg(const G2Affine& q);
G2Prepared prepare(const G2Affine& q);
// prod_i e(P_i, Q_i) == 1 ?   P_i affine (Montgomery coordinates, identity = (0,0)).
bool product_is_one(const G1Affine* P, const G2Prepared* const* Q, int n);
// The same check in pieces, for a caller that runs the Miller loops of different pairs on different threads: the loops are
// independent (each 63 squarings + 68 line products of its own), their values multiply, one final exponentiation decides.
Fp12 miller_loop(const G1Affine& P, const G2Prepared& Q);    // conjugated already (the BLS parameter is negative)
Fp12 fp12_mul(const Fp12& a, const Fp12& b);
bool final_exponentiation_is_one(const Fp12& f);             // f = product of miller_loop values

}  // namespace pairing
}  // namespace kzg
