#include "sha256.hpp"
#include <cstring>
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

namespace kzg {

static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void blocks_portable(uint32_t h[8], const uint8_t* p, size_t nblocks) {
    for (; nblocks; nblocks--, p += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
}

#if defined(__x86_64__)
__attribute__((target("sha,sse4.1,ssse3"))) static void blocks_shani(uint32_t state[8], const uint8_t* data, size_t nblocks) {
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&state[0]);
    __m128i st1 = _mm_loadu_si128((const __m128i*)&state[4]);
    tmp = _mm_shuffle_epi32(tmp, 0xB1);          // CDAB
    st1 = _mm_shuffle_epi32(st1, 0x1B);          // EFGH
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);  // ABEF
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);       // CDGH
    for (; nblocks; nblocks--, data += 64) {
        __m128i abef = st0, cdgh = st1, msg, m0, m1, m2, m3;
        m0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 0)), MASK);
        m1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16)), MASK);
        m2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 32)), MASK);
        m3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 48)), MASK);
#define RND4(M, KI)                                                              \
    msg = _mm_add_epi32(M, _mm_loadu_si128((const __m128i*)&K[KI]));             \
    st1 = _mm_sha256rnds2_epu32(st1, st0, msg);                                  \
    msg = _mm_shuffle_epi32(msg, 0x0E);                                          \
    st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
#define SCHED(A, B, C, D) /* A = next schedule word group from A,B,C,D */         \
    A = _mm_sha256msg1_epu32(A, B);                                              \
    A = _mm_add_epi32(A, _mm_alignr_epi8(D, C, 4));                              \
    A = _mm_sha256msg2_epu32(A, D);
        RND4(m0, 0) RND4(m1, 4) RND4(m2, 8) RND4(m3, 12)
        for (int k = 16; k < 64; k += 16) {
            SCHED(m0, m1, m2, m3) RND4(m0, k)
            SCHED(m1, m2, m3, m0) RND4(m1, k + 4)
            SCHED(m2, m3, m0, m1) RND4(m2, k + 8)
            SCHED(m3, m0, m1, m2) RND4(m3, k + 12)
        }
#undef RND4
#undef SCHED
        st0 = _mm_add_epi32(st0, abef);
        st1 = _mm_add_epi32(st1, cdgh);
    }
    tmp = _mm_shuffle_epi32(st0, 0x1B);        // FEBA
    st1 = _mm_shuffle_epi32(st1, 0xB1);        // DCHG
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);     // DCBA
    st1 = _mm_alignr_epi8(st1, tmp, 8);        // ABEF -> HGFE
    _mm_storeu_si128((__m128i*)&state[0], st0);
    _mm_storeu_si128((__m128i*)&state[4], st1);
}
static bool detect_shani() {
    unsigned a, b, c, d;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
    bool sha = (b >> 29) & 1;
    if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
    bool sse41 = (c >> 19) & 1, ssse3 = (c >> 9) & 1;
    return sha && sse41 && ssse3;
}
#else
static bool detect_shani() { return false; }
static void blocks_shani(uint32_t*, const uint8_t*, size_t) {}
#endif

static bool self_test_shani() {
    // the accelerated path must agree with the portable one before it is trusted
    uint8_t blk[128];
    for (int i = 0; i < 128; i++) blk[i] = (uint8_t)(i * 7 + 3);
    uint32_t a[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}, b[8];
    memcpy(b, a, sizeof a);
    blocks_portable(a, blk, 2);
    blocks_shani(b, blk, 2);
    return memcmp(a, b, sizeof a) == 0;
}
bool sha256_uses_shani() {  // called from many host threads at once (the transcript hashes of concurrent verifications): a function-local static
    static const bool shani = detect_shani() && self_test_shani();  // is initialised once, thread-safely
    return shani;
}
static void blocks(uint32_t h[8], const uint8_t* p, size_t n) {
    if (sha256_uses_shani()) blocks_shani(h, p, n);
    else blocks_portable(h, p, n);
}

Sha256::Sha256() : buf_len(0), total(0) {
    static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, IV, sizeof IV);
}
void Sha256::update(const uint8_t* data, size_t len) {
    total += len;
    if (buf_len) {
        size_t take = 64 - buf_len < len ? 64 - buf_len : len;
        memcpy(buf + buf_len, data, take);
        buf_len += take; data += take; len -= take;
        if (buf_len == 64) { blocks(h, buf, 1); buf_len = 0; }
    }
    if (len >= 64) {
        size_t n = len / 64;
        blocks(h, data, n);
        data += n * 64; len -= n * 64;
    }
    if (len) { memcpy(buf, data, len); buf_len = len; }
}
void Sha256::finish(uint8_t out[32]) {
    uint64_t bits = total * 8;
    uint8_t pad[72] = {0x80};
    size_t padlen = (buf_len < 56 ? 56 : 120) - buf_len;
    uint8_t lenb[8];
    for (int k = 0; k < 8; k++) lenb[k] = (uint8_t)(bits >> (56 - 8 * k));
    update(pad, padlen);
    update(lenb, 8);
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
}

}  // namespace kzg
