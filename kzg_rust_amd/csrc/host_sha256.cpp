// host_sha256.cpp -- see host_sha256.h.  Host-only translation unit (no device code).
#include "host_sha256.h"

#include <string.h>
#include <vector>

#if defined(__x86_64__) || defined(__i386__)
#include <cpuid.h>
#include <immintrin.h>
#define KZG_HOST_X86 1
#endif

namespace kzg_host {
namespace {

const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
    0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
    0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
    0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
    0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
    0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
const uint32_t H0[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline uint32_t load_be(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void store_be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

// FIPS 180-4 section 6.2.2, one block at a time
void compress_portable(uint32_t st[8], const uint8_t *p, size_t nblocks) {
    for (; nblocks; nblocks--, p += 64) {
        uint32_t w[64];
        for (int t = 0; t < 16; t++) w[t] = load_be(p + 4 * t);
        for (int t = 16; t < 64; t++) {
            const uint32_t s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
            const uint32_t s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
            w[t] = w[t - 16] + s0 + w[t - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int t = 0; t < 64; t++) {
            const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[t] + w[t];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
    }
}

#if defined(KZG_HOST_X86)
// SHA extensions: sha256rnds2 performs two rounds on the state halves (A,B,E,F) / (C,D,G,H); sha256msg1 / msg2 the two halves of
// the message schedule  W[t] = s1(W[t-2]) + W[t-7] + s0(W[t-15]) + W[t-16]  four words at a time:
//     W_G = msg2( msg1(W_{G-4}, W_{G-3}) + alignr(W_{G-1}, W_{G-2}, 4), W_{G-1} )      (W_G = words 4G .. 4G+3)
// LANES independent messages are walked in lockstep (their instructions interleave: the round instructions of one message are a
// dependent chain of ~4-cycle operations).
template <int LANES> __attribute__((target("sha,sse4.1,ssse3")))
void compress_shani(uint32_t (*st)[8], const uint8_t *const *ptr, size_t nblocks) {
    const __m128i BSWAP = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i s0[LANES], s1[LANES];
    for (int l = 0; l < LANES; l++) {
        __m128i t = _mm_loadu_si128((const __m128i *)&st[l][0]);           // a b c d
        __m128i u = _mm_loadu_si128((const __m128i *)&st[l][4]);           // e f g h
        t = _mm_shuffle_epi32(t, 0xB1);                                    // b a d c
        u = _mm_shuffle_epi32(u, 0x1B);                                    // h g f e
        s0[l] = _mm_alignr_epi8(t, u, 8);                                  // f e b a  = (A,B,E,F) as the instruction wants it
        s1[l] = _mm_blend_epi16(u, t, 0xF0);                               // h g d c  = (C,D,G,H)
    }
    const uint8_t *p[LANES];
    for (int l = 0; l < LANES; l++) p[l] = ptr[l];
    for (size_t b = 0; b < nblocks; b++) {
        __m128i w[LANES][4], a0[LANES], a1[LANES];
        for (int l = 0; l < LANES; l++) {
            a0[l] = s0[l]; a1[l] = s1[l];
            for (int q = 0; q < 4; q++) w[l][q] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p[l] + 16 * q)), BSWAP);
            p[l] += 64;
        }
#pragma unroll
        for (int G = 0; G < 16; G++) {
            const __m128i k = _mm_loadu_si128((const __m128i *)&K256[4 * G]);
            for (int l = 0; l < LANES; l++) {
                if (G >= 4) {
                    __m128i x = _mm_sha256msg1_epu32(w[l][G & 3], w[l][(G + 1) & 3]);
                    x = _mm_add_epi32(x, _mm_alignr_epi8(w[l][(G + 3) & 3], w[l][(G + 2) & 3], 4));
                    w[l][G & 3] = _mm_sha256msg2_epu32(x, w[l][(G + 3) & 3]);
                }
                __m128i m = _mm_add_epi32(w[l][G & 3], k);
                s1[l] = _mm_sha256rnds2_epu32(s1[l], s0[l], m);
                m = _mm_shuffle_epi32(m, 0x0E);
                s0[l] = _mm_sha256rnds2_epu32(s0[l], s1[l], m);
            }
        }
        for (int l = 0; l < LANES; l++) { s0[l] = _mm_add_epi32(s0[l], a0[l]); s1[l] = _mm_add_epi32(s1[l], a1[l]); }
    }
    for (int l = 0; l < LANES; l++) {
        const __m128i t = _mm_shuffle_epi32(s0[l], 0x1B);                  // a b e f
        const __m128i u = _mm_shuffle_epi32(s1[l], 0xB1);                  // g h c d  (lanes: d c h g -> shuffled)
        _mm_storeu_si128((__m128i *)&st[l][0], _mm_blend_epi16(t, u, 0xF0));        // a b c d
        _mm_storeu_si128((__m128i *)&st[l][4], _mm_alignr_epi8(u, t, 8));           // e f g h
    }
}
#endif

bool detect_shani() {
#if defined(KZG_HOST_X86)
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
    const bool ssse3 = (c >> 9) & 1, sse41 = (c >> 19) & 1;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
    return ssse3 && sse41 && ((b >> 29) & 1);
#else
    return false;
#endif
}

// LANES messages of nblocks blocks each
template <int LANES> void compress_n(uint32_t (*st)[8], const uint8_t *const *ptr, size_t nblocks, bool shani) {
#if defined(KZG_HOST_X86)
    if (shani) { compress_shani<LANES>(st, ptr, nblocks); return; }
#endif
    (void)shani;
    for (int l = 0; l < LANES; l++) compress_portable(st[l], ptr[l], nblocks);
}

bool use_shani(int impl) { return impl != SHA256_PORTABLE && sha256_have_shani(); }

// LANES challenge transcripts at once: head block (header | blob[0, 32)), the blob read in place at offset 32, tail (blob's last 32
// bytes | commitment | padding) in two blocks.  blob_bytes a multiple of 64.
template <int LANES> void challenge_n(uint8_t *out, const uint8_t *const *blob, size_t blob_bytes, const uint8_t *const *cm, uint64_t n_fe, bool shani) {
    uint32_t st[LANES][8];
    uint8_t head[LANES][64], tail[LANES][128];
    const uint8_t *hp[LANES], *mp[LANES], *tp[LANES];
    const uint64_t bits = (uint64_t)(32 + blob_bytes + 48) * 8;
    for (int l = 0; l < LANES; l++) {
        memcpy(st[l], H0, sizeof H0);
        memcpy(head[l], "FSBLOBVERIFY_V1_", 16);                           // FIAT_SHAMIR_PROTOCOL_DOMAIN (consts.rs:22)
        memset(head[l] + 16, 0, 8);                                        // u64be(0)
        for (int k = 0; k < 8; k++) head[l][24 + k] = (uint8_t)(n_fe >> (56 - 8 * k));      // u64be(FIELD_ELEMENTS_PER_BLOB)
        memcpy(head[l] + 32, blob[l], 32);
        memcpy(tail[l], blob[l] + blob_bytes - 32, 32);
        memcpy(tail[l] + 32, cm[l], 48);
        tail[l][80] = 0x80;
        memset(tail[l] + 81, 0, 128 - 81 - 8);
        for (int k = 0; k < 8; k++) tail[l][120 + k] = (uint8_t)(bits >> (56 - 8 * k));
        hp[l] = head[l]; mp[l] = blob[l] + 32; tp[l] = tail[l];
    }
    compress_n<LANES>(st, hp, 1, shani);
    compress_n<LANES>(st, mp, blob_bytes / 64 - 1, shani);
    compress_n<LANES>(st, tp, 2, shani);
    for (int l = 0; l < LANES; l++) for (int k = 0; k < 8; k++) store_be(out + 32 * l + 4 * k, st[l][k]);
}

}  // namespace

bool sha256_have_shani() { static const bool have = detect_shani(); return have; }

bool sha256(uint8_t out[32], const uint8_t *msg, size_t len, int impl) {
    const bool shani = use_shani(impl);
    uint32_t st[1][8];
    memcpy(st[0], H0, sizeof H0);
    const size_t full = len / 64;
    const uint8_t *p[1] = {msg};
    if (full) compress_n<1>(st, p, full, shani);
    uint8_t tail[128];
    const size_t rem = len - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    const size_t tb = rem + 9 <= 64 ? 1 : 2;
    const uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[64 * tb - 8 + k] = (uint8_t)(bits >> (56 - 8 * k));
    p[0] = tail;
    compress_n<1>(st, p, tb, shani);
    for (int k = 0; k < 8; k++) store_be(out + 4 * k, st[0][k]);
    return impl != SHA256_SHANI || shani;
}

void challenge_digests(uint8_t *out, const uint8_t *blobs, size_t blob_bytes, const uint8_t *commitments, size_t count, uint64_t n_fe, int impl) {
    const bool shani = use_shani(impl);
    if (blob_bytes < 64 || blob_bytes % 64) {                                  // (no preset has such blobs; kept total for the generic form)
        std::vector<uint8_t> m(32 + blob_bytes + 48);
        for (size_t i = 0; i < count; i++) {
            memcpy(m.data(), "FSBLOBVERIFY_V1_", 16); memset(m.data() + 16, 0, 8);
            for (int k = 0; k < 8; k++) m[24 + k] = (uint8_t)(n_fe >> (56 - 8 * k));
            memcpy(m.data() + 32, blobs + blob_bytes * i, blob_bytes);
            memcpy(m.data() + 32 + blob_bytes, commitments + 48 * i, 48);
            sha256(out + 32 * i, m.data(), m.size(), impl);
        }
        return;
    }
    size_t i = 0;
    for (; i + 2 <= count; i += 2) {
        const uint8_t *b[2] = {blobs + blob_bytes * i, blobs + blob_bytes * (i + 1)}, *c[2] = {commitments + 48 * i, commitments + 48 * (i + 1)};
        challenge_n<2>(out + 32 * i, b, blob_bytes, c, n_fe, shani);
    }
    if (i < count) {
        const uint8_t *b[1] = {blobs + blob_bytes * i}, *c[1] = {commitments + 48 * i};
        challenge_n<1>(out + 32 * i, b, blob_bytes, c, n_fe, shani);
    }
}

}  // namespace kzg_host
