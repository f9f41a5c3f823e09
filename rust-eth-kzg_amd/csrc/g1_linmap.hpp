// The two G1 transforms of FK20 (h = first 64 outputs of IFFT_128(y), proofs = FFT_128(h || 0)) as ONE fixed linear
// map over G1 with public coefficients, compiled on the host into a straight-line program of point operations
// (add / sub / repeated doubling / multiplication by a constant) that k_g1slp.hip runs batched over blobs.
// Replaces, for batches beyond the small-batch circulant kernel, what the reference does with two calls of
// fft_inplace<G1Projective> (crates/cryptography/polynomial/src/domain.rs:149-194, fft.rs:46-177; called from
// kzg_multi_open/src/fk20/batch_toeplitz.rs:117-133 and fk20/prover.rs:214-222).
//
// Why not the radix-2 network: on this GPU a multiplication of a point by a 255-bit constant costs ~690 k VALU
// instructions (GLV + width-5 NAF, k_g1fft.hip) and a point addition ~8 k, a ratio of ~84.  The radix-2 pair of
// transforms needs 642 constant multiplications per blob in 14 dependent rounds.  The same map is
//     proofs[2a]   = y[2a]   + sum_b c1[(a-b) mod 64] y[2b+1]
//     proofs[2a+1] = y[2a+1] + sum_b c2[(a-b) mod 64] y[2b]          (y pre-scaled by 1/2 in the MSM scalars)
// i.e. two cyclic convolutions of length 64 with FIXED kernels c1[d] = m(2d-1), c2[d] = m(2d+1),
// m(e) = (1/32) / (1 - w^e), w = omega_128.  A cyclic convolution with a fixed kernel splits for free (additions only)
// along X^64 - 1 = (X-1)(X+1)(X^2+1)...(X^32+1); each factor X^n + 1 is a Toeplitz matrix-vector product with fixed
// matrix, which Karatsuba / Toom-Cook evaluate with 3 / 7 / 15 sub-products per 2 / 4 / 8-way split.  In the
// transposed (Toeplitz) form of Toom-Cook the interpolation matrix lands on the FIXED operand (free: Fr arithmetic at
// set-up) and the variable points only see the evaluation matrix and its transpose, whose entries are small powers
// of two: additions and a few doublings.  A factor may also be split once more over Fr (X^n - z = (X^(n/2) - s)
// (X^(n/2) + s), s^2 = z) at the price of n constant multiplications.  A cost model picks per size.  Result: ~450
// constant multiplications per blob instead of 642, in 2-3 dependent rounds instead of 14 (which also fills the chip
// at medium batch sizes).
//
// The plan is verified at construction: it is executed over Fr (points replaced by random scalars) and compared with
// the definition of the map (an inverse DFT, truncation, a forward DFT).
#pragma once
#include "field.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace kzg {
namespace linmap {

enum OpKind : uint8_t { OP_ADD = 0, OP_SUB = 1, OP_DBL = 2, OP_MULC = 3 };
struct Op {
    OpKind kind;
    int dst;  // value id (SSA)
    int a;    // first operand (value id)
    int b;    // ADD/SUB: second operand; DBL: number of doublings; MULC: constant id
};
// a signed reference to a value: (id, negative?).  id < 0: the identity (a term that is absent)
struct Ref {
    int id = -1;
    bool neg = false;
    bool zero() const { return id < 0; }
};

struct Plan {
    int n_in = 0;
    int n_values = 0;           // inputs are values 0 .. n_in-1
    std::vector<Op> ops;        // in dependency order
    std::vector<Ref> outputs;   // one per output
    std::vector<Fr> consts;     // MULC constants, Montgomery form
    long count(OpKind k) const {
        long n = 0;
        for (auto& o : ops) n += o.kind == k;
        return n;
    }
    long doublings() const {
        long n = 0;
        for (auto& o : ops) n += o.kind == OP_DBL ? o.b : 0;
        return n;
    }
};

// relative costs in VALU instructions (measured shapes of k_g1fft.hip / curve29.hpp)
constexpr double COST_MULC = 690e3, COST_ADD = 8.3e3, COST_DBL = 3.1e3;

inline Fr fr_small(int64_t v) {
    Fr a = zero<FrParams>();
    uint64_t m = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
    a.v[0] = (uint32_t)m;
    a.v[1] = (uint32_t)(m >> 32);
    a = to_mont(a);
    return v < 0 ? neg(a) : a;
}

class Builder {
public:
    explicit Builder(int n_in) {
        plan_.n_in = n_in;
        plan_.n_values = n_in;
    }
    Ref input(int i) const { return Ref{i, false}; }
    static Ref negate(Ref r) { return r.zero() ? r : Ref{r.id, !r.neg}; }

    Ref add(Ref a, Ref b) {
        if (a.zero()) return b;
        if (b.zero()) return a;
        if (a.id == b.id) return a.neg == b.neg ? dbl(a, 1) : Ref{};
        if (!a.neg && !b.neg) return Ref{emit2(OP_ADD, a.id, b.id, true), false};
        if (a.neg && b.neg) return Ref{emit2(OP_ADD, a.id, b.id, true), true};
        if (!a.neg) return Ref{emit2(OP_SUB, a.id, b.id, false), false};  // a - b
        return Ref{emit2(OP_SUB, b.id, a.id, false), false};              // b - a
    }
    Ref sub(Ref a, Ref b) { return add(a, negate(b)); }
    Ref dbl(Ref a, int t) {
        if (a.zero() || t == 0) return a;
        // fold chains: 2^t (2^s v) = 2^(t+s) v only if the inner value has no other use -- keep it simple, no folding
        auto key = std::make_tuple((int)OP_DBL, a.id, t);
        auto it = memo_.find(key);
        if (it != memo_.end()) return Ref{it->second, a.neg};
        int id = plan_.n_values++;
        plan_.ops.push_back(Op{OP_DBL, id, a.id, t});
        memo_[key] = id;
        return Ref{id, a.neg};
    }
    Ref mulc(Ref a, const Fr& c) {  // c in Montgomery form
        if (a.zero() || is_zero(c)) return Ref{};
        if (eq(c, one<FrParams>())) return a;
        if (eq(c, neg(one<FrParams>()))) return negate(a);
        int cid = (int)plan_.consts.size();
        plan_.consts.push_back(c);
        int id = plan_.n_values++;
        plan_.ops.push_back(Op{OP_MULC, id, a.id, cid});
        return Ref{id, a.neg};
    }
    // sum_i coef_i * x_i with small integer coefficients: bit planes from the top, doublings merged
    Ref lincomb(const std::vector<std::pair<Ref, int64_t>>& terms) {
        std::vector<std::pair<Ref, uint64_t>> t;
        int top = -1;
        for (auto& pr : terms) {
            if (pr.first.zero() || pr.second == 0) continue;
            Ref r = pr.second < 0 ? negate(pr.first) : pr.first;
            uint64_t m = pr.second < 0 ? (uint64_t)(-pr.second) : (uint64_t)pr.second;
            t.emplace_back(r, m);
            for (int b = 63; b >= 0; b--)
                if ((m >> b) & 1) { if (b > top) top = b; break; }
        }
        Ref acc;
        int pending = 0;
        for (int b = top; b >= 0; b--) {
            if (!acc.zero()) pending++;
            if (b == top) pending = 0;
            for (auto& pr : t)
                if ((pr.second >> b) & 1) {
                    if (pending) { acc = dbl(acc, pending); pending = 0; }
                    acc = add(acc, pr.first);
                }
        }
        if (pending) acc = dbl(acc, pending);
        return acc;
    }
    // the same sum as a balanced tree when every coefficient is +-2^e: the terms sorted by exponent, halves summed relative to
    // their own lowest exponent, joined by ONE run of doublings and one addition per tree level -- depth log2(terms) instead
    // of one dependent (run + addition) per term (the Horner form above).  For batches that leave the chip part empty, where
    // a dependency level lasts as long as its longest operation.  Other coefficients fall back to the bit-plane form.
    Ref lincomb_balanced(const std::vector<std::pair<Ref, int64_t>>& terms) {
        struct T { Ref r; int e; };
        std::vector<T> t;
        for (auto& pr : terms) {
            if (pr.first.zero() || pr.second == 0) continue;
            const uint64_t m = pr.second < 0 ? (uint64_t)(-pr.second) : (uint64_t)pr.second;
            if (m & (m - 1)) return lincomb(terms);
            int e = 0;
            while (!((m >> e) & 1)) e++;
            t.push_back(T{pr.second < 0 ? negate(pr.first) : pr.first, e});
        }
        if (t.empty()) return Ref{};
        std::stable_sort(t.begin(), t.end(), [](const T& a, const T& b) { return a.e < b.e; });
        // sum of t[lo..hi) divided by 2^(t[lo].e)
        std::function<Ref(int, int)> rec = [&](int lo, int hi) -> Ref {
            if (hi - lo == 1) return t[lo].r;
            const int mid = (lo + hi) / 2;
            const Ref a = rec(lo, mid), b = rec(mid, hi);
            return add(dbl(b, t[mid].e - t[lo].e), a);
        };
        return dbl(rec(0, (int)t.size()), t[0].e);
    }
    Plan take(const std::vector<Ref>& outputs) {
        plan_.outputs = outputs;
        return std::move(plan_);
    }
    const Plan& plan() const { return plan_; }
    double cost() const { return plan_.count(OP_MULC) * COST_MULC + (plan_.count(OP_ADD) + plan_.count(OP_SUB)) * COST_ADD + plan_.doublings() * COST_DBL; }

private:
    int emit2(OpKind k, int a, int b, bool commutative) {
        if (commutative && a > b) std::swap(a, b);
        auto key = std::make_tuple((int)k, a, b);
        auto it = memo_.find(key);
        if (it != memo_.end()) return it->second;
        int id = plan_.n_values++;
        plan_.ops.push_back(Op{k, id, a, b});
        memo_[key] = id;
        return id;
    }
    Plan plan_;
    std::map<std::tuple<int, int, int>, int> memo_;
};

// ---------------------------------------------------------------------------------------------------------------
// Toom-Cook evaluation points for a k-way split, projective (a : b): E[r][J] = a^J b^(k-1-J).
struct EvalPoint { int a, b; };
inline std::vector<EvalPoint> toom_points(int k) {
    switch (k) {
        case 2: return {{0, 1}, {1, 0}, {1, 1}};
        case 3: return {{0, 1}, {1, 0}, {1, 1}, {-1, 1}, {2, 1}};
        case 4: return {{0, 1}, {1, 0}, {1, 1}, {-1, 1}, {2, 1}, {-2, 1}, {1, 2}};
        case 8: return {{0, 1}, {1, 0}, {1, 1}, {-1, 1}, {2, 1}, {-2, 1}, {1, 2}, {-1, 2}, {4, 1}, {-4, 1}, {1, 4}, {-1, 4}, {8, 1}, {-8, 1}, {1, 8}};
        default: throw std::runtime_error("toom_points: unsupported split");
    }
}
inline int64_t ipow(int64_t b, int e) {
    int64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}
// U = (E2^T)^-1 with E2[r][s] = a^s b^(2k-2-s), r, s < 2k-1: U[r][D] multiplies Hankel block D for evaluation point r
inline std::vector<std::vector<Fr>> toom_fixed_side(int k) {
    const auto pts = toom_points(k);
    const int R = 2 * k - 1;
    // solve E2^T U = I  <=>  for each column D of U: E2^T u = e_D.  Gauss-Jordan on [E2^T | I].
    std::vector<std::vector<Fr>> M(R, std::vector<Fr>(2 * R, zero<FrParams>()));
    for (int s = 0; s < R; s++) {
        for (int r = 0; r < R; r++) M[s][r] = fr_small(ipow(pts[r].a, s) * ipow(pts[r].b, 2 * k - 2 - s));
        M[s][R + s] = one<FrParams>();
    }
    for (int col = 0; col < R; col++) {
        int piv = -1;
        for (int row = col; row < R; row++)
            if (!is_zero(M[row][col])) { piv = row; break; }
        if (piv < 0) throw std::runtime_error("toom: singular evaluation matrix");
        std::swap(M[piv], M[col]);
        const Fr iv = inv(M[col][col]);
        for (int j = 0; j < 2 * R; j++) M[col][j] = mul(M[col][j], iv);
        for (int row = 0; row < R; row++) {
            if (row == col || is_zero(M[row][col])) continue;
            const Fr f = M[row][col];
            for (int j = 0; j < 2 * R; j++) M[row][j] = sub(M[row][j], mul(f, M[col][j]));
        }
    }
    std::vector<std::vector<Fr>> U(R, std::vector<Fr>(R));
    for (int r = 0; r < R; r++)
        for (int D = 0; D < R; D++) U[r][D] = M[r][R + D];  // (E2^T)^-1 [r][D]
    return U;
}

// ---------------------------------------------------------------------------------------------------------------
class Compiler {
public:
    // strategy tables filled by tune(): split used for a Hankel product of size n; whether a twisted-cyclic product of
    // size n is split over Fr first
    std::map<int, int> hankel_split;
    std::map<int, bool> root_split;
    bool allow_toom8 = true;
    bool balanced_lincomb = false;  // evaluation / interpolation sums as balanced trees (Builder::lincomb_balanced)
    Ref sum(Builder& B, const std::vector<std::pair<Ref, int64_t>>& t) const { return balanced_lincomb ? B.lincomb_balanced(t) : B.lincomb(t); }

    // y_i = sum_j h[i + j] x[j],  i, j < n;  h has 2n - 1 entries
    std::vector<Ref> hankel(Builder& B, const std::vector<Ref>& x, const std::vector<Fr>& h) {
        const int n = (int)x.size();
        if ((int)h.size() != 2 * n - 1) throw std::runtime_error("hankel: size mismatch");
        if (n == 1) return {B.mulc(x[0], h[0])};
        const int k = hankel_split.count(n) ? hankel_split[n] : 2;
        if (n % k) throw std::runtime_error("hankel: split does not divide the size");
        const int m = n / k, R = 2 * k - 1;
        const auto pts = toom_points(k);
        const auto& U = fixed_side(k);
        // evaluation of the block vector: X_r[i] = sum_J E[r][J] x[J m + i]; +-pairs share their even / odd parts
        std::vector<std::vector<Ref>> X(R, std::vector<Ref>(m));
        std::vector<int> partner(R, -1);
        for (int r = 0; r < R; r++)
            for (int q = 0; q < R; q++)
                if (q != r && pts[q].a == -pts[r].a && pts[q].b == pts[r].b && pts[r].a > 0) { partner[r] = q; partner[q] = r; }
        for (int r = 0; r < R; r++) {
            if (partner[r] >= 0 && pts[r].a < 0) continue;  // produced with its positive partner
            for (int i = 0; i < m; i++) {
                if (partner[r] < 0) {
                    std::vector<std::pair<Ref, int64_t>> t;
                    for (int J = 0; J < k; J++) t.emplace_back(x[J * m + i], ipow(pts[r].a, J) * ipow(pts[r].b, k - 1 - J));
                    X[r][i] = sum(B, t);
                } else {
                    std::vector<std::pair<Ref, int64_t>> ev, od;
                    for (int J = 0; J < k; J++) (J & 1 ? od : ev).emplace_back(x[J * m + i], ipow(pts[r].a, J) * ipow(pts[r].b, k - 1 - J));
                    const Ref e = sum(B, ev), o = sum(B, od);
                    X[r][i] = B.add(e, o);
                    X[partner[r]][i] = B.sub(e, o);
                }
            }
        }
        // sub-products with the fixed side h_r = sum_D U[r][D] h^(D), h^(D) = h[D m .. D m + 2m - 2]
        std::vector<std::vector<Ref>> Z(R);
        for (int r = 0; r < R; r++) {
            std::vector<Fr> hr(2 * m - 1, zero<FrParams>());
            for (int D = 0; D < R; D++) {
                if (is_zero(U[r][D])) continue;
                for (int s = 0; s < 2 * m - 1; s++) hr[s] = add(hr[s], mul(U[r][D], h[D * m + s]));
            }
            Z[r] = hankel(B, X[r], hr);
        }
        // y_I[i] = sum_r E[r][I] Z_r[i]; +-pairs enter through their sum (I even) or difference (I odd)
        std::vector<Ref> y(n);
        for (int i = 0; i < m; i++) {
            std::vector<Ref> S(R), Dm(R);
            for (int r = 0; r < R; r++)
                if (partner[r] >= 0 && pts[r].a > 0) {
                    S[r] = B.add(Z[r][i], Z[partner[r]][i]);
                    Dm[r] = B.sub(Z[r][i], Z[partner[r]][i]);
                }
            for (int I = 0; I < k; I++) {
                std::vector<std::pair<Ref, int64_t>> t;
                for (int r = 0; r < R; r++) {
                    if (partner[r] >= 0 && pts[r].a < 0) continue;
                    const int64_t coef = ipow(pts[r].a, I) * ipow(pts[r].b, k - 1 - I);
                    if (partner[r] < 0) t.emplace_back(Z[r][i], coef);
                    else t.emplace_back((I & 1) ? Dm[r] : S[r], coef);
                }
                y[I * m + i] = sum(B, t);
            }
        }
        return y;
    }

    // y = c * x mod (X^n - z): y_i = sum_{j <= i} c[i-j] x[j] + z sum_{j > i} c[n+i-j] x[j].   sqrt_of: returns true and a
    // square root of its argument if one is known (powers of the 2-adic root of unity)
    template <class SqrtFn>
    std::vector<Ref> twisted_cyclic(Builder& B, const std::vector<Ref>& x, const std::vector<Fr>& c, const Fr& z, SqrtFn&& sqrt_of) {
        const int n = (int)x.size();
        Fr s;
        if (n >= 2 && root_split.count(n) && root_split[n] && sqrt_of(z, s)) {
            const int hn = n / 2;
            const Fr half = inv(fr_small(2)), s_inv = inv(s);
            std::vector<Ref> x1(hn), x2(hn);
            std::vector<Fr> c1(hn), c2(hn);
            for (int i = 0; i < hn; i++) {
                const Ref t = B.mulc(x[hn + i], s);
                x1[i] = B.add(x[i], t);   // x mod (X^hn - s)
                x2[i] = B.sub(x[i], t);   // x mod (X^hn + s)
                const Fr sc = mul(s, c[hn + i]);
                c1[i] = mul(half, add(c[i], sc));
                c2[i] = mul(half, sub(c[i], sc));
            }
            const auto y1 = twisted_cyclic(B, x1, c1, s, sqrt_of);
            const auto y2 = twisted_cyclic(B, x2, c2, neg(s), sqrt_of);
            std::vector<Ref> y(n);
            for (int i = 0; i < hn; i++) {
                y[i] = B.add(y1[i], y2[i]);                       // (y1 + y2) / 2, the half sits in the kernels
                y[hn + i] = B.mulc(B.sub(y1[i], y2[i]), s_inv);   // (y1 - y2) / (2 s)
            }
            return y;
        }
        // Toeplitz t[d] = c[d] (d >= 0), z c[n + d] (d < 0)  ->  Hankel on the reversed input: h[u] = t[u - (n-1)]
        std::vector<Fr> h(2 * n - 1);
        for (int u = 0; u < 2 * n - 1; u++) {
            const int d = u - (n - 1);
            h[u] = d >= 0 ? c[d] : mul(z, c[n + d]);
        }
        std::vector<Ref> xr(n);
        for (int j = 0; j < n; j++) xr[j] = x[n - 1 - j];
        return hankel(B, xr, h);
    }

    // y = c (*) x mod (X^n - 1), n a power of two: free splits down to X - 1, one twisted product per factor X^m + 1
    template <class SqrtFn>
    std::vector<Ref> cyclic(Builder& B, const std::vector<Ref>& x, const std::vector<Fr>& c, SqrtFn&& sqrt_of) {
        const int n = (int)x.size();
        if (n == 1) return {B.mulc(x[0], c[0])};
        const int hn = n / 2;
        const Fr half = inv(fr_small(2));
        std::vector<Ref> xp(hn), xm(hn);
        std::vector<Fr> cp(hn), cm(hn);
        for (int i = 0; i < hn; i++) {
            xp[i] = B.add(x[i], x[hn + i]);
            xm[i] = B.sub(x[i], x[hn + i]);
            cp[i] = mul(half, add(c[i], c[hn + i]));
            cm[i] = mul(half, sub(c[i], c[hn + i]));
        }
        const auto yp = cyclic(B, xp, cp, sqrt_of);
        const auto ym = twisted_cyclic(B, xm, cm, neg(one<FrParams>()), sqrt_of);
        std::vector<Ref> y(n);
        for (int i = 0; i < hn; i++) {
            y[i] = B.add(yp[i], ym[i]);
            y[hn + i] = B.sub(yp[i], ym[i]);
        }
        return y;
    }

    // choose the splits by building each candidate with generic (random) fixed operands and pricing its operations
    void tune(int max_n) {
        uint64_t st = 0x9e3779b97f4a7c15ull;
        auto rnd_fr = [&]() {
            Fr a;
            for (int i = 0; i < 8; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a.v[i] = (uint32_t)(st >> 16); }
            a.v[7] &= 0x3fffffffu;
            return a;
        };
        auto no_sqrt = [](const Fr&, Fr&) { return false; };
        auto any_sqrt = [&](const Fr&, Fr& s) { s = rnd_fr(); return true; };  // pricing only: the value is irrelevant
        for (int n = 2; n <= max_n; n *= 2) {
            double best = 0;
            int best_k = 0;
            for (int k : {2, 4, 8}) {
                if (n % k || (k == 8 && !allow_toom8)) continue;
                hankel_split[n] = k;
                Builder B(n);
                std::vector<Ref> x(n);
                for (int i = 0; i < n; i++) x[i] = B.input(i);
                std::vector<Fr> h(2 * n - 1);
                for (auto& v : h) v = rnd_fr();
                hankel(B, x, h);
                if (!best_k || B.cost() < best) { best = B.cost(); best_k = k; }
            }
            hankel_split[n] = best_k;
            // root split or straight Toeplitz for a twisted cyclic product of this size?
            double cost[2];
            for (int split = 0; split < 2; split++) {
                root_split[n] = split != 0;
                Builder B(n);
                std::vector<Ref> x(n);
                for (int i = 0; i < n; i++) x[i] = B.input(i);
                std::vector<Fr> c(n);
                for (auto& v : c) v = rnd_fr();
                if (split) twisted_cyclic(B, x, c, rnd_fr(), any_sqrt);
                else twisted_cyclic(B, x, c, rnd_fr(), no_sqrt);
                cost[split] = B.cost();
            }
            root_split[n] = cost[1] < cost[0];
        }
    }

private:
    const std::vector<std::vector<Fr>>& fixed_side(int k) {
        auto it = fixed_.find(k);
        if (it == fixed_.end()) it = fixed_.emplace(k, toom_fixed_side(k)).first;
        return it->second;
    }
    std::map<int, std::vector<std::vector<Fr>>> fixed_;
};

// execute a plan over Fr (points replaced by scalars): the self-check of every plan
inline std::vector<Fr> run_over_fr(const Plan& p, const std::vector<Fr>& in) {
    std::vector<Fr> v(p.n_values, zero<FrParams>());
    for (int i = 0; i < p.n_in; i++) v[i] = in[i];
    for (auto& o : p.ops) {
        switch (o.kind) {
            case OP_ADD: v[o.dst] = add(v[o.a], v[o.b]); break;
            case OP_SUB: v[o.dst] = sub(v[o.a], v[o.b]); break;
            case OP_DBL: { Fr t = v[o.a]; for (int k = 0; k < o.b; k++) t = add(t, t); v[o.dst] = t; break; }
            case OP_MULC: v[o.dst] = mul(v[o.a], p.consts[o.b]); break;
        }
    }
    std::vector<Fr> out(p.outputs.size());
    for (size_t i = 0; i < out.size(); i++) {
        const Ref r = p.outputs[i];
        out[i] = r.zero() ? zero<FrParams>() : (r.neg ? neg(v[r.id]) : v[r.id]);
    }
    return out;
}

// How a plan is built: the split of a Hankel product per size, whether a twisted-cyclic product is first split over Fr, and the
// shape of the small-integer sums.  tuned = false: the tables below are used as given (sizes without an entry: 2-way, no root
// split); tuned = true: Compiler::tune fills them by operation count (the throughput optimum for batches that fill the chip).
struct Strategy {
    bool tuned = true, allow_toom8 = true;
    std::map<int, int> hankel_split;
    std::map<int, bool> root_split;
    bool balanced_lincomb = false;
};
// The FK20 map: in[j] = y_j / 2 (Fourier index j < 128, natural order), out[k] = proof at FFT index k (natural order).
// w128[e] = omega_128^e in Montgomery form, e < 128.
inline Plan build_fk20_proofs_plan(const std::vector<Fr>& w128, const Strategy& strat, bool verbose = false) {
    if (w128.size() != 128) throw std::runtime_error("w128: need the 128 powers");
    auto sqrt_of = [&](const Fr& z, Fr& s) {  // z = w^e with e even  ->  s = w^(e/2)
        for (int e = 0; e < 128; e += 2)
            if (eq(z, w128[e])) { s = w128[e / 2]; return true; }
        return false;
    };
    Compiler C;
    C.allow_toom8 = strat.allow_toom8;
    C.balanced_lincomb = strat.balanced_lincomb;
    if (strat.tuned) C.tune(32);
    else { C.hankel_split = strat.hankel_split; C.root_split = strat.root_split; }
    Builder B(128);
    const Fr one_ = one<FrParams>(), inv32 = inv(fr_small(32));
    auto m_of = [&](int e) {  // (1/32) / (1 - w^e), e odd
        return mul(inv32, inv(sub(one_, w128[((e % 128) + 128) % 128])));
    };
    std::vector<Ref> even(64), odd(64);
    for (int a = 0; a < 64; a++) { even[a] = B.input(2 * a); odd[a] = B.input(2 * a + 1); }
    std::vector<Fr> c1(64), c2(64);
    for (int d = 0; d < 64; d++) { c1[d] = m_of(2 * d - 1); c2[d] = m_of(2 * d + 1); }
    const auto t1 = C.cyclic(B, odd, c1, sqrt_of);   // -> even outputs
    const auto t2 = C.cyclic(B, even, c2, sqrt_of);  // -> odd outputs
    std::vector<Ref> out(128);
    for (int a = 0; a < 64; a++) {
        out[2 * a] = B.add(even[a], t1[a]);
        out[2 * a + 1] = B.add(odd[a], t2[a]);
    }
    // outputs must be plain (positive) values produced by an operation, so that the executor can pin their slots
    for (auto& r : out)
        if (r.zero() || r.neg || r.id < 128) throw std::runtime_error("fk20 plan: degenerate output");
    Plan p = B.take(out);
    if (verbose) {
        fprintf(stderr, "[linmap] hankel splits:");
        for (auto& kv : C.hankel_split) fprintf(stderr, " %d->%d", kv.first, kv.second);
        fprintf(stderr, "  root splits:");
        for (auto& kv : C.root_split) fprintf(stderr, " %d:%d", kv.first, (int)kv.second);
        fprintf(stderr, "\n[linmap] %ld constant multiplications, %ld additions, %ld doublings\n", p.count(OP_MULC),
                p.count(OP_ADD) + p.count(OP_SUB), p.doublings());
    }
    return p;
}
inline Plan build_fk20_proofs_plan(const std::vector<Fr>& w128, bool allow_toom8 = true, bool verbose = false) {
    Strategy s;
    s.allow_toom8 = allow_toom8;
    return build_fk20_proofs_plan(w128, s, verbose);
}

// the definition the plan is checked against: h = first 64 of IDFT_128(2 in) (with 1/128), out = DFT_128(h || 0)
inline std::vector<Fr> fk20_proofs_map_by_definition(const std::vector<Fr>& w128, const std::vector<Fr>& in) {
    const Fr inv64 = inv(fr_small(64));
    std::vector<Fr> h(64), out(128);
    for (int n = 0; n < 64; n++) {
        Fr acc = zero<FrParams>();
        for (int j = 0; j < 128; j++) acc = add(acc, mul(in[j], w128[(128 - (j * n) % 128) % 128]));
        h[n] = mul(acc, inv64);  // 2 / 128
    }
    for (int k = 0; k < 128; k++) {
        Fr acc = zero<FrParams>();
        for (int n = 0; n < 64; n++) acc = add(acc, mul(h[n], w128[(k * n) % 128]));
        out[k] = acc;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Scheduling for the device: launches in order; every launch holds operations of one kind that are mutually
// independent.  All constant multiplications of equal "multiplication depth" share ONE launch (they are the expensive,
// chip-filling part); the cheap operations between them are levelled by dependency.
struct Launch {
    OpKind kind;     // OP_MULC, or OP_ADD for a mixed launch of additions, subtractions and doubling runs
    int first, count;  // range in Schedule::words (4 words per operation)
};
struct Schedule {
    int n_slots = 0;             // arena slots; inputs occupy 0 .. n_in-1, outputs n_in .. n_in+n_out-1
    std::vector<uint32_t> words;  // per operation: dst slot, a slot, b (slot | doublings | constant id), flags (1 = subtract, 2 = doubling run, 4 = a + b to dst and a - b to slot flags >> 16; bits 3-7: doublings of operand a first)
    std::vector<Launch> launches;
    long mulc_total = 0;
    long fused_pairs = 0;  // (a + b, a - b) pairs emitted as one operation
    long fused_runs = 0;   // doubling runs folded into the operation that consumes them
};
inline Schedule make_schedule(const Plan& p, bool fuse_add_sub = true, bool fuse_runs = true) {
    const int n_out = (int)p.outputs.size();
    // dead-code elimination
    std::vector<char> live(p.n_values, 0);
    for (auto& r : p.outputs) live[r.id] = 1;
    for (int i = (int)p.ops.size() - 1; i >= 0; i--) {
        const Op& o = p.ops[i];
        if (!live[o.dst]) continue;
        live[o.a] = 1;
        if (o.kind == OP_ADD || o.kind == OP_SUB) live[o.b] = 1;
    }
    // A run of doublings whose only consumer is ONE addition / subtraction (or one a + b / a - b pair that becomes one operation)
    // is folded into that consumer: the device doubles the operand in registers before it adds.  1544 of the 1552 runs of the
    // FK20 map qualify: no launch level, no wave, no store and reload of their own (41 -> 25 cheap launches).
    std::vector<int> run_of(p.n_values, 0);  // value -> number of doublings folded into its consumer (its source is ops[def].a)
    std::vector<int> run_src(p.n_values, -1);
    if (fuse_runs) {
        std::vector<std::vector<int>> users(p.n_values);
        std::vector<char> is_output(p.n_values, 0);
        for (auto& r : p.outputs) is_output[r.id] = 1;
        for (int i = 0; i < (int)p.ops.size(); i++) {
            const Op& o = p.ops[i];
            if (!live[o.dst]) continue;
            users[o.a].push_back(i);
            if (o.kind == OP_ADD || o.kind == OP_SUB) users[o.b].push_back(i);
        }
        for (auto& o : p.ops) {
            if (!live[o.dst] || o.kind != OP_DBL || is_output[o.dst] || o.b > 31) continue;
            const auto& u = users[o.dst];
            auto two = [&](int k) { return p.ops[k].kind == OP_ADD || p.ops[k].kind == OP_SUB; };
            // the device doubles the FIRST operand only: the run must be the minuend of a subtraction (either side of an addition
            // will do: it is emitted first), and the other operand must not be a folded run itself
            auto fits = [&](const Op& c) {
                const int other = c.a == o.dst ? c.b : c.a;
                return c.a != c.b && !run_of[other] && (c.kind == OP_ADD || c.a == o.dst);
            };
            bool ok = false;
            if (u.size() == 1 && two(u[0]) && fits(p.ops[u[0]])) ok = true;
            if (fuse_add_sub && u.size() == 2 && two(u[0]) && two(u[1]) && p.ops[u[0]].kind != p.ops[u[1]].kind) {
                const Op &x = p.ops[u[0]], &y = p.ops[u[1]];
                if (std::min(x.a, x.b) == std::min(y.a, y.b) && std::max(x.a, x.b) == std::max(y.a, y.b) && fits(x) && fits(y)) ok = true;
            }
            if (!ok) continue;
            run_of[o.dst] = o.b;
            run_src[o.dst] = o.a;
        }
    }
    auto eff = [&](int v) { return run_of[v] ? run_src[v] : v; };  // the value an operand is read from
    // multiplication depth and level inside the phase
    std::vector<int> md(p.n_values, 0), lvl(p.n_values, 0);
    int max_md = 0;
    std::vector<int> max_lvl;
    for (auto& o : p.ops) {
        if (!live[o.dst]) continue;
        const bool two = o.kind == OP_ADD || o.kind == OP_SUB;
        if (run_of[o.dst]) {  // folded run: no operation of its own, it sits where its source sits
            md[o.dst] = md[o.a];
            lvl[o.dst] = lvl[o.a];
            continue;
        }
        if (o.kind == OP_MULC) {
            md[o.dst] = md[o.a] + 1;
            lvl[o.dst] = 0;
        } else {
            md[o.dst] = two ? std::max(md[o.a], md[o.b]) : md[o.a];
            int l = 0;
            if (md[o.a] == md[o.dst]) l = std::max(l, lvl[o.a] + 1);
            if (two && md[o.b] == md[o.dst]) l = std::max(l, lvl[o.b] + 1);
            lvl[o.dst] = std::max(l, 1);  // level 0 of a phase is its multiplication launch (inputs sit at (0, 0))
        }
        max_md = std::max(max_md, md[o.dst]);
        if ((int)max_lvl.size() <= md[o.dst]) max_lvl.resize(md[o.dst] + 1, 0);
        max_lvl[md[o.dst]] = std::max(max_lvl[md[o.dst]], lvl[o.dst]);
    }
    // launch order: (phase, level, kind); a time step per (phase, level) -- kinds within a step are independent
    std::vector<std::vector<int>> step_ops;  // op indices per step
    std::map<std::pair<int, int>, int> step_of;
    for (int ph = 0; ph <= max_md; ph++)
        for (int l = 0; l <= (ph < (int)max_lvl.size() ? max_lvl[ph] : 0); l++) {
            step_of[{ph, l}] = (int)step_ops.size();
            step_ops.emplace_back();
        }
    std::vector<int> def_step(p.n_values, -1), last_use(p.n_values, -1);
    for (int i = 0; i < (int)p.ops.size(); i++) {
        const Op& o = p.ops[i];
        if (!live[o.dst] || run_of[o.dst]) continue;
        const int s = step_of[{md[o.dst], lvl[o.dst]}];
        step_ops[s].push_back(i);
        def_step[o.dst] = s;
        last_use[eff(o.a)] = std::max(last_use[eff(o.a)], s);
        if (o.kind == OP_ADD || o.kind == OP_SUB) last_use[eff(o.b)] = std::max(last_use[eff(o.b)], s);
    }
    // slots: inputs and outputs pinned, temporaries from a free list; a slot freed at step s is reusable from step s+1
    Schedule S;
    std::vector<int> slot(p.n_values, -1);
    for (int i = 0; i < p.n_in; i++) slot[i] = i;
    for (int k = 0; k < n_out; k++) {
        if (slot[p.outputs[k].id] >= 0) throw std::runtime_error("schedule: an output value is shared or is an input");
        slot[p.outputs[k].id] = p.n_in + k;
    }
    int next_slot = p.n_in + n_out;
    std::vector<int> free_list;
    std::vector<std::vector<int>> expire(step_ops.size() + 1);
    for (int v = p.n_in; v < p.n_values; v++)
        if (live[v] && slot[v] < 0 && last_use[v] >= 0) expire[last_use[v]].push_back(v);
    for (int s = 0; s < (int)step_ops.size(); s++) {
        for (int i : step_ops[s]) {
            const int v = p.ops[i].dst;
            if (slot[v] >= 0) continue;
            if (!free_list.empty()) { slot[v] = free_list.back(); free_list.pop_back(); }
            else slot[v] = next_slot++;
        }
        for (int v : expire[s]) free_list.push_back(slot[v]);
        // emit the launches of this step: the constant multiplications, then ONE launch for the additions, subtractions
        // and doubling runs together (they are independent of each other within a step; half as many small launches)
        // a + b and a - b of the same two values are ONE operation on the device (curve29.hpp: add_sub_* share all but the last
        // squaring and product pair): the subtraction carries the pair, the addition is dropped from the list
        std::map<std::pair<int, int>, int> add_of;  // unordered operand pair -> index of the step's addition
        for (int i : step_ops[s])
            if (p.ops[i].kind == OP_ADD) add_of[{std::min(p.ops[i].a, p.ops[i].b), std::max(p.ops[i].a, p.ops[i].b)}] = i;
        std::map<int, int> partner;  // subtraction -> its addition
        std::vector<char> fused_away(p.ops.size(), 0);
        if (fuse_add_sub)
            for (int i : step_ops[s]) {
                if (p.ops[i].kind != OP_SUB) continue;
                auto it = add_of.find({std::min(p.ops[i].a, p.ops[i].b), std::max(p.ops[i].a, p.ops[i].b)});
                if (it == add_of.end() || fused_away[it->second]) continue;
                partner[i] = it->second;
                fused_away[it->second] = 1;
            }
        // the cheap operations of a step longest first (a run of k doublings costs k COST_DBL, an addition COST_ADD, a fused pair
        // 1.15 of that): the device deals them out in this order, so the launch ends on short operations instead of a straggling
        // doubling run
        std::vector<int> ordered = step_ops[s];
        auto op_cost = [&](int i) {
            const Op& o = p.ops[i];
            if (o.kind == OP_DBL) return o.b * COST_DBL;
            if (o.kind == OP_MULC) return COST_MULC;
            return (run_of[o.a] + run_of[o.b]) * COST_DBL + (partner.count(i) ? 1.15 * COST_ADD : COST_ADD);  // at most one of the two is a folded run
        };
        std::stable_sort(ordered.begin(), ordered.end(), [&](int x, int y) { return op_cost(x) > op_cost(y); });
        for (int pass = 0; pass < 2; pass++) {
            Launch L{pass == 0 ? OP_MULC : OP_ADD, (int)S.words.size() / 4, 0};
            for (int i : ordered) {
                const Op& o = p.ops[i];
                const bool two = o.kind == OP_ADD || o.kind == OP_SUB;
                if ((pass == 0) != (o.kind == OP_MULC)) continue;
                if (fused_away[i]) continue;
                // an addition / subtraction with a folded run: that operand first, its doublings in bits 3-7 of the flags
                int first = o.a, second = two ? o.b : 0;
                if (two && run_of[o.b]) std::swap(first, second);  // additions only (see the folding rule above)
                const uint32_t shifts = two ? (uint32_t)run_of[first] << 3 : 0u;
                if (two && run_of[first]) S.fused_runs++;
                if (partner.count(i)) {  // words: slot of a + b, a, b, 4 | shifts | slot of a - b << 16
                    S.words.push_back((uint32_t)slot[p.ops[partner[i]].dst]);
                    S.words.push_back((uint32_t)slot[eff(first)]);
                    S.words.push_back((uint32_t)slot[eff(second)]);
                    S.words.push_back(4u | shifts | ((uint32_t)slot[o.dst] << 16));
                    L.count++;
                    S.fused_pairs++;
                    continue;
                }
                S.words.push_back((uint32_t)slot[o.dst]);
                S.words.push_back((uint32_t)slot[two ? eff(first) : o.a]);
                S.words.push_back(two ? (uint32_t)slot[eff(second)] : (uint32_t)o.b);
                S.words.push_back((o.kind == OP_SUB ? 1u : o.kind == OP_DBL ? 2u : 0u) | shifts);
                L.count++;
            }
            if (L.count) S.launches.push_back(L);
            if (pass == 0) S.mulc_total += L.count;
        }
    }
    S.n_slots = next_slot;
    return S;
}

// execute a schedule over Fr exactly as the device does (slot arena, launches in order; within a launch all reads
// happen before all writes): checks the slot allocation and the launch order, not only the algebra
inline std::vector<Fr> run_schedule_over_fr(const Schedule& S, const std::vector<Fr>& consts, int n_in, int n_out, const std::vector<Fr>& in) {
    std::vector<Fr> arena(S.n_slots, zero<FrParams>());
    for (int i = 0; i < n_in; i++) arena[i] = in[i];
    for (auto& L : S.launches) {
        std::vector<Fr> res(L.count), res2(L.count);
        for (int i = 0; i < L.count; i++) {
            const uint32_t* w = &S.words[(size_t)(L.first + i) * 4];
            const Fr a = arena[w[1]];
            if (L.kind == OP_MULC) res[i] = mul(a, consts[w[2]]);
            else if (w[3] & 2u) { Fr t = a; for (uint32_t k = 0; k < w[2]; k++) t = add(t, t); res[i] = t; }
            else {
                Fr x = a, y = arena[w[2]];
                for (uint32_t k = 0; k < ((w[3] >> 3) & 31u); k++) x = add(x, x);
                if (w[3] & 4u) { res[i] = add(x, y); res2[i] = sub(x, y); }
                else res[i] = (w[3] & 1u) ? sub(x, y) : add(x, y);
            }
        }
        for (int i = 0; i < L.count; i++) {
            const uint32_t* w = &S.words[(size_t)(L.first + i) * 4];
            arena[w[0]] = res[i];
            if (L.kind != OP_MULC && (w[3] & 4u)) arena[w[3] >> 16] = res2[i];
        }
    }
    return std::vector<Fr>(arena.begin() + n_in, arena.begin() + n_in + n_out);
}

}  // namespace linmap
}  // namespace kzg
