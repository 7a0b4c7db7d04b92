// sha256.h -- FIPS 180-4 SHA-256 as a device function (one message per lane).
// Reference counterpart: blst_sha256 as used by compute_challenge (src/kzg.rs:298-339) and
// compute_r_powers (src/utils.rs:426-474).  The message is fed as big-endian 32-bit words, 16 per block;
// callers assemble the words (domain separator, lengths, blob bytes, points) themselves so nothing is
// staged in memory.  Host+device.
#pragma once
#include <stdint.h>
#include "field.h"

namespace kzg {

struct Sha256 { uint32_t h[8]; };

KZG_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

KZG_HD void sha256_init(Sha256 &s) {
    s.h[0] = 0x6a09e667u; s.h[1] = 0xbb67ae85u; s.h[2] = 0x3c6ef372u; s.h[3] = 0xa54ff53au;
    s.h[4] = 0x510e527fu; s.h[5] = 0x9b05688cu; s.h[6] = 0x1f83d9abu; s.h[7] = 0x5be0cd19u;
}

// one compression; w[16] = the block as big-endian words (clobbered: it is the rolling schedule)
KZG_HD_NOINLINE void sha256_block(Sha256 &s, uint32_t *w) {
    const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + (g ^ (e & (f ^ g))) + K[i] + w[i & 15];
        const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) | (c & (a | b)));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}

KZG_HD uint32_t load_be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// Generic (slow, byte-wise) one-shot hash: used for the short r-transcript and by the host unit tests.
KZG_HD void sha256_bytes(uint8_t out[32], const uint8_t *msg, uint64_t len) {
    Sha256 s; sha256_init(s);
    uint32_t w[16];
    uint64_t off = 0;
    for (; off + 64 <= len; off += 64) {
        for (int i = 0; i < 16; i++) w[i] = load_be32(msg + off + 4 * i);
        sha256_block(s, w);
    }
    uint8_t tail[128];
    const uint32_t rem = (uint32_t)(len - off);
    for (uint32_t i = 0; i < 128; i++) tail[i] = i < rem ? msg[off + i] : 0;
    tail[rem] = 0x80;
    const uint32_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (uint32_t o = 0; o < tl; o += 64) {
        for (int i = 0; i < 16; i++) w[i] = load_be32(tail + o + 4 * i);
        sha256_block(s, w);
    }
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(s.h[i] >> 24); out[4 * i + 1] = (uint8_t)(s.h[i] >> 16); out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
            out[4 * i + 3] = (uint8_t)s.h[i]; }
}

// digest words (big-endian word order: h[0] is most significant) -> 8 little-endian words of the 256-bit integer
KZG_HD void sha256_digest_to_words(uint32_t w[8], const Sha256 &s) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = s.h[7 - i];
}

}  // namespace kzg
