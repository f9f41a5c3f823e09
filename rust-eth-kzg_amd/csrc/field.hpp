// BLS12-381 prime fields for gfx950: Fp (381 bit, 12 x u32) and Fr (255 bit, 8 x u32),
// Montgomery form.  Written for the CDNA4 VALU: one field element per lane, limbs in VGPRs,
// every 32x32->64 multiply-add a v_mad_u64_u32.  The same code compiles for the host
// (hipcc host pass) where it backs SRS loading and the pairing check.
//
// Replaces what the reference gets from blstrs::{Fp, Scalar}
// (reference: crates/cryptography/bls12_381/src/lib.rs:23-42, used by
//  batch_addition.rs:14-39 and polynomial/src/fft.rs:164-177).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __host__ __device__ __forceinline__

namespace kzg {

// ---------------------------------------------------------------------------------------------
// Modulus descriptors (little-endian u32 limbs).  constexpr so that the compiler materialises
// them as scalar literals (SGPR moves), never as memory loads.
struct FpParams {
    static constexpr int N = 12;
    static constexpr int BITS = 381;
    static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr uint32_t N0 = 0xfffcfffdu;  // -p^-1 mod 2^32
    // R mod p, R^2 mod p (R = 2^384)
    static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                         0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                        0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};
struct FrParams {
    static constexpr int N = 8;
    static constexpr int BITS = 255;
    static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t N0 = 0xffffffffu;  // -r^-1 mod 2^32
    static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                        0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                       0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

template <class P>
struct Felt {
    static constexpr int N = P::N;
    uint32_t v[N];
};
using Fp = Felt<FpParams>;
using Fr = Felt<FrParams>;

// ---------------------------------------------------------------------------------------------
template <class P>
HD bool is_zero(const Felt<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.v[i];
    return o == 0;
}
template <class P>
HD bool eq(const Felt<P>& a, const Felt<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
template <class P>
HD Felt<P> zero() {
    Felt<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = 0;
    return r;
}
template <class P>
HD Felt<P> one() {
    Felt<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = P::ONE[i];
    return r;
}

// r = a - b, returns borrow.  __builtin_subc/addc lower to v_sub_co/v_subb_co carry chains on gfx950.
template <int N>
HD uint32_t sub_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        unsigned bo;
        r[i] = __builtin_subc(a[i], b[i], br, &bo);
        br = bo;
    }
    return br;
}
template <int N>
HD uint32_t add_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        unsigned co;
        r[i] = __builtin_addc(a[i], b[i], c, &co);
        c = co;
    }
    return c;
}
// conditional final subtraction: r = (a >= mod) ? a - mod : a   (a < 2*mod)
template <class P>
HD void reduce_once(Felt<P>& a) {
    uint32_t t[P::N];
    uint32_t br = sub_limbs<P::N>(t, a.v, P::MOD);
#pragma unroll
    for (int i = 0; i < P::N; i++) a.v[i] = br ? a.v[i] : t[i];
}
template <class P>
HD Felt<P> add(const Felt<P>& a, const Felt<P>& b) {
    Felt<P> r;
    add_limbs<P::N>(r.v, a.v, b.v);  // no overflow: 2*mod < 2^(32N)
    reduce_once(r);
    return r;
}
template <class P>
HD Felt<P> sub(const Felt<P>& a, const Felt<P>& b) {
    Felt<P> r;
    uint32_t br = sub_limbs<P::N>(r.v, a.v, b.v);
    uint32_t t[P::N];
    add_limbs<P::N>(t, r.v, P::MOD);
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = br ? t[i] : r.v[i];
    return r;
}
template <class P>
HD Felt<P> neg(const Felt<P>& a) {
    Felt<P> r;
    sub_limbs<P::N>(r.v, P::MOD, a.v);
    bool z = is_zero(a);
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = z ? 0u : r.v[i];
    return r;
}
template <class P>
HD Felt<P> dbl(const Felt<P>& a) { return add(a, a); }

// Montgomery multiplication, CIOS, 32-bit limbs (portable form; used on the host and as the
// cross-check for the device form below).
template <class P>
HD Felt<P> mul_cios(const Felt<P>& a, const Felt<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        t[N + 1] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * P::N0;
        c = (uint64_t)m * P::MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    Felt<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    reduce_once(r);  // t < 2*mod and t[N] == 0 for these moduli
    return r;
}

// 96-bit column accumulator for product scanning: (ex : lo) += x * y.
// Device form: one v_mad_u64_u32 (carry-out in VCC) + one v_addc_co_u32 that folds the carry.
struct Acc96 {
    uint64_t lo;
    uint32_t ex;
};
HD void mac(Acc96& A, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(A.lo), "+v"(A.ex)
        : "v"(x), "v"(y)
        : "vcc");
#else
    uint64_t p = (uint64_t)x * y;
    A.lo += p;
    A.ex += (A.lo < p);
#endif
}
// same with a wave-uniform (scalar) second factor: the modulus limbs live in SGPRs
HD void mac_s(Acc96& A, uint32_t x, uint32_t y_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(A.lo), "+v"(A.ex)
        : "v"(x), "s"(y_uniform)
        : "vcc");
#else
    mac(A, x, y_uniform);
#endif
}
HD void acc_shift(Acc96& A) {
    A.lo = (A.lo >> 32) | ((uint64_t)A.ex << 32);
    A.ex = 0;
}

// Montgomery multiplication, finely-integrated product scanning (FIPS): column k accumulates
// sum a[i]*b[k-i] + m[i]*p[k-i]; no row carry chains, 2*N^2 v_mad_u64_u32 in total.
template <class P>
HD Felt<P> mul_fips(const Felt<P>& a, const Felt<P>& b) {
    constexpr int N = P::N;
    uint32_t m[N];
    Felt<P> r;
    Acc96 A{0, 0};
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < k; i++) {
            mac(A, a.v[i], b.v[k - i]);
            mac_s(A, m[i], P::MOD[k - i]);
        }
        mac(A, a.v[k], b.v[0]);
        m[k] = (uint32_t)A.lo * P::N0;
        mac_s(A, m[k], P::MOD[0]);
        acc_shift(A);
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            mac(A, a.v[i], b.v[k - i]);
            mac_s(A, m[i], P::MOD[k - i]);
        }
        r.v[k - N] = (uint32_t)A.lo;
        acc_shift(A);
    }
    reduce_once(r);  // top word is zero for these moduli (4p < 2^384, 2r < 2^256)
    return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host form: the same limbs read as N/2 64-bit words (little-endian host), CIOS with 128-bit products: ~3x fewer multiply
// instructions than the 32-bit form above.  Backs the pairing check, SRS set-up and the transcript scalars.
template <class P>
inline Felt<P> mul_host64(const Felt<P>& a, const Felt<P>& b) {
    constexpr int N = P::N / 2;
    static_assert(P::N % 2 == 0, "even limb count");
    typedef unsigned __int128 u128;
    uint64_t A[N], B[N], M[N], t[N + 2];
    for (int i = 0; i < N; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
        M[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    uint64_t inv = M[0];  // Newton iteration for M[0]^-1 mod 2^64 (correct to 3 bits, doubling each round)
    for (int i = 0; i < 6; i++) inv *= 2 - M[0] * inv;
    const uint64_t n0 = 0 - inv;
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
        u128 c = 0;
        for (int j = 0; j < N; j++) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[N];
        t[N] = (uint64_t)c;
        t[N + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * n0;
        c = (u128)m * M[0] + t[0];
        c >>= 64;
        for (int j = 1; j < N; j++) {
            c += (u128)m * M[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[N];
        t[N - 1] = (uint64_t)c;
        t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    Felt<P> r;
    for (int i = 0; i < N; i++) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
    reduce_once(r);
    return r;
}
#endif

template <class P>
HD Felt<P> mul(const Felt<P>& a, const Felt<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mul_fips(a, b);
#else
    return mul_host64(a, b);
#endif
}
template <class P>
HD Felt<P> sqr(const Felt<P>& a) { return mul(a, a); }

template <class P>
HD Felt<P> to_mont(const Felt<P>& a) {
    Felt<P> r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) r2.v[i] = P::R2[i];
    return mul(a, r2);
}
template <class P>
HD Felt<P> from_mont(const Felt<P>& a) {
    Felt<P> o = zero<P>();
    o.v[0] = 1;
    return mul(a, o);
}
// a >= mod ? (canonical, non-Montgomery limbs)
template <class P>
HD bool geq_mod(const uint32_t* a) {
    uint32_t t[P::N];
    return sub_limbs<P::N>(t, a, P::MOD) == 0;
}

// square-and-multiply with a fixed public exponent (little-endian u32 limbs), MSB first
template <class P, int EL>
HD Felt<P> pow_fixed(const Felt<P>& a, const uint32_t (&e)[EL]) {
    Felt<P> acc = one<P>();
    bool started = false;
    for (int i = 32 * EL - 1; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = mul(acc, a);
            started = true;
        }
    }
    return acc;
}

// Fermat inversion a^(mod-2)
template <class P>
HD Felt<P> inv(const Felt<P>& a) {
    uint32_t e[P::N], two[P::N];
#pragma unroll
    for (int i = 0; i < P::N; i++) two[i] = i == 0 ? 2u : 0u;
    sub_limbs<P::N>(e, P::MOD, two);  // mod - 2 (r ends in ...00000001: the borrow must propagate)
    Felt<P> acc = one<P>();
    bool started = false;
    for (int i = 32 * P::N - 1; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

}  // namespace kzg
