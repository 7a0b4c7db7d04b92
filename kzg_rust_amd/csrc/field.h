// field.h -- Fp (381-bit) and Fr (255-bit) Montgomery arithmetic for gfx950, unsaturated 29-bit limbs.
//
// Why 29-bit limbs: measured on MI355X (profiles/r01_valu_issue_rates.txt) v_mad_u64_u32 issues at
// the same rate as any other VALU op, so the cost of a field product is its instruction COUNT.
// With limbs < 2^29 every column sum of a Montgomery product (<= 28 terms < 2^58 each) fits a 64-bit
// accumulator, so each limb product is exactly ONE v_mad_u64_u32 with no carry handling: an Fp product
// is 2*14*14 = 392 mads + ~100 shift/mask ops, versus ~1350 instructions for saturated 32-bit limbs
// (288 mads + carry/zero-extension traffic).  Fp: 14 limbs (R = 2^406), Fr: 9 limbs (R = 2^261).
//
// Every value is kept CANONICAL (limbs < 2^29, value < modulus) so equality is limb equality.
// Reference counterpart: the blst_fp / blst_fr types behind src/utils.rs, src/kzg.rs;
// blst's 64-bit-limb R = 2^384 / 2^256 layout is deliberately not reproduced.
//
// Everything is KZG_HD (host+device): the same source is unit-tested with g++ on the CPU build box
// (tests/native/hd_probe.cpp) before it runs on a GPU.  The product ships only device instantiations.
#pragma once
#include <stdint.h>
#include "consts_gen.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_HD __host__ __device__ __forceinline__
#define KZG_HD_NOINLINE inline __host__ __device__ __attribute__((noinline))
#else
#define KZG_HD inline __attribute__((always_inline))
#define KZG_HD_NOINLINE inline __attribute__((noinline))
#endif

namespace kzg {

constexpr int LB = KZG_LIMB_BITS;                 // 29
constexpr uint32_t LMASK = (1u << LB) - 1u;
constexpr int NFP = KZG_FP_LIMBS;                 // 14
constexpr int NFR = KZG_FR_LIMBS;                 // 9

struct Fp { uint32_t l[NFP]; };
struct Fr { uint32_t l[NFR]; };

// ---------------------------------------------------------------------------------- limb helpers
// r = a - b over N 29-bit limbs (mod 2^(29N)); returns 1 if a < b
template <int N> KZG_HD uint32_t ul_sub(uint32_t *r, const uint32_t *a, const uint32_t *b) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t d = a[i] - b[i] - borrow;
        borrow = d >> 31;
        r[i] = d & LMASK;
    }
    return borrow;
}
template <int N> KZG_HD bool ul_is_zero(const uint32_t *a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a[i];
    return t == 0;
}
template <int N> KZG_HD bool ul_eq(const uint32_t *a, const uint32_t *b) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a[i] ^ b[i];
    return t == 0;
}
// a + b mod m  (a, b canonical)
template <int N> KZG_HD void mod_add(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *m) {
    // d = a + b - m with a signed carry chain; if it went negative add m back
    uint32_t d[N];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        int32_t t = (int32_t)(a[i] + b[i] - m[i]) + c;
        c = t >> LB;
        d[i] = (uint32_t)t & LMASK;
    }
    const uint32_t neg = (uint32_t)c;      // 0 or 0xffffffff
    uint32_t cc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t t = d[i] + (m[i] & neg) + cc;
        cc = t >> LB;
        r[i] = t & LMASK;
    }
}
// a - b mod m
template <int N> KZG_HD void mod_sub(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *m) {
    uint32_t d[N];
    const uint32_t neg = 0u - ul_sub<N>(d, a, b);
    uint32_t cc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t t = d[i] + (m[i] & neg) + cc;
        cc = t >> LB;
        r[i] = t & LMASK;
    }
}
// Montgomery product r = a*b/2^(29N) mod m.  a, b: limbs < 2^29, a*b < m*2^(29N).  inv = -m^-1 mod 2^29.
// Column accumulators: acc[j] collects a[j']*b[i] and q_i*m[j'] for the current weight; after each outer
// step the (now zero mod 2^29) lowest column is shifted out.  No carries until the final sweep.
template <int N> KZG_HD void mont_mul(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *m, const uint32_t inv) {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t bi = b[i];
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] += (uint64_t)a[j] * bi;
        const uint32_t q = ((uint32_t)acc[0] * inv) & LMASK;
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] += (uint64_t)q * m[j];
        const uint64_t carry = acc[0] >> LB;
#pragma unroll
        for (int j = 0; j < N - 1; j++) acc[j] = acc[j + 1];
        acc[N - 1] = 0;
        acc[0] += carry;
    }
    uint32_t t[N];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        c += acc[j];
        t[j] = (uint32_t)c & LMASK;
        c >>= LB;
    }
    // t < m + a*b/R; one conditional subtraction makes it canonical
    uint32_t s[N];
    const uint32_t br = ul_sub<N>(s, t, m);
#pragma unroll
    for (int j = 0; j < N; j++) r[j] = br ? t[j] : s[j];
}
// Montgomery square: the N(N+1)/2 distinct limb products (cross terms against a pre-doubled copy) go to 2N column
// accumulators, then the same word-by-word reduction.  N(N+1)/2 + N^2 limb products instead of 2 N^2 (Fp: 301 vs 392).
// Columns stay below 2^64: at most N products of < 2^59 plus N of < 2^58 each, N <= 14.
template <int N, bool LAZY = false> KZG_HD void mont_sqr(uint32_t *r, const uint32_t *a, const uint32_t *m, const uint32_t inv) {
    static_assert(N <= 14, "column accumulators sized for at most 14 limbs");
    uint64_t acc[2 * N];
    uint32_t a2[N];
#pragma unroll
    for (int j = 0; j < N; j++) a2[j] = a[j] << 1;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        acc[2 * i] += (uint64_t)a[i] * a[i];
#pragma unroll
        for (int j = i + 1; j < N; j++) acc[i + j] += (uint64_t)a2[i] * a[j];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t q = ((uint32_t)acc[i] * inv) & LMASK;
#pragma unroll
        for (int j = 0; j < N; j++) acc[i + j] += (uint64_t)q * m[j];
        acc[i + 1] += acc[i] >> LB;
    }
    uint32_t t[N];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        c += acc[N + j];
        t[j] = (LAZY && j == N - 1) ? (uint32_t)c : ((uint32_t)c & LMASK);
        c >>= LB;
    }
    if (LAZY) {
#pragma unroll
        for (int j = 0; j < N; j++) r[j] = t[j];
        return;
    }
    uint32_t s[N];
    const uint32_t br = ul_sub<N>(s, t, m);
#pragma unroll
    for (int j = 0; j < N; j++) r[j] = br ? t[j] : s[j];
}
// Same product WITHOUT the final conditional subtraction: result < m (1 + a*b / (m 2^(29N))), limbs normalised.
// For operands below ~2.6 m the result stays below 1.1 m, which is all a following product (or a bounded number of
// additions) needs; the last operation of a chain uses mont_mul, which canonicalises.  Saves ~45 of ~316 (Fr)
// instructions per product on throughput kernels.
template <int N> KZG_HD void mont_mul_lazy(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *m, const uint32_t inv) {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t bi = b[i];
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] += (uint64_t)a[j] * bi;
        const uint32_t q = ((uint32_t)acc[0] * inv) & LMASK;
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] += (uint64_t)q * m[j];
        const uint64_t carry = acc[0] >> LB;
#pragma unroll
        for (int j = 0; j < N - 1; j++) acc[j] = acc[j + 1];
        acc[N - 1] = 0;
        acc[0] += carry;
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        c += acc[j];
        r[j] = j < N - 1 ? ((uint32_t)c & LMASK) : (uint32_t)c;     // the top limb keeps any excess
        c >>= LB;
    }
}
// r = (a*b + c*d) / R, lazily: two products share one Montgomery reduction (2N^2 + N^2 limb products instead of 4N^2).
// Operands as for mont_mul_lazy with (a*b + c*d) < ~5 m^2 (N = 9) / < 2^6 m * 2^6 m in all (N = 14).  Column sums: an accumulator collects at most
// N rounds of three limb products (a_j b_i, c_j d_i, q m_j), each below 2^58 for limbs below 2^29 (the top limb of a lazy value is smaller still),
// plus carries below 2^36: 3 * 14 * 2^58 < 2^63.4 -- inside 64 bits up to N = 14 PROVIDED every limb below the top one is normalised (< 2^29),
// which is what fp_sub_lz / fp_add_lz / the lazy products leave.
template <int N> KZG_HD void mont_mul2_lazy(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d,
                                            const uint32_t *m, const uint32_t inv) {
    static_assert(N <= 14, "column accumulators: 3 N limb products of < 2^58 must stay below 2^64");
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; j++) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t bi = b[i], di = d[i];
#pragma unroll
        for (int j = 0; j < N; j++) { acc[j] += (uint64_t)a[j] * bi; acc[j] += (uint64_t)c[j] * di; }
        const uint32_t q = ((uint32_t)acc[0] * inv) & LMASK;
#pragma unroll
        for (int j = 0; j < N; j++) acc[j] += (uint64_t)q * m[j];
        const uint64_t carry = acc[0] >> LB;
#pragma unroll
        for (int j = 0; j < N - 1; j++) acc[j] = acc[j + 1];
        acc[N - 1] = 0;
        acc[0] += carry;
    }
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        cy += acc[j];
        r[j] = j < N - 1 ? ((uint32_t)cy & LMASK) : (uint32_t)cy;
        cy >>= LB;
    }
}
// 32-bit word array (little-endian words, NW of them) -> N 29-bit limbs
template <int N, int NW> KZG_HD void words_to_limbs(uint32_t *l, const uint32_t *w) {
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int bit = LB * k, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = wi < NW ? w[wi] : 0u;
        uint32_t hi = (wi + 1) < NW ? w[wi + 1] : 0u;
        uint32_t v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        l[k] = v & LMASK;
    }
}
// N 29-bit limbs -> NW 32-bit words (value must fit)
template <int N, int NW> KZG_HD void limbs_to_words(uint32_t *w, const uint32_t *l) {
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int bit = 32 * i, k = bit / LB, sh = bit % LB;     // word i starts inside limb k at offset sh
        uint32_t v = k < N ? (l[k] >> sh) : 0u;
        if ((k + 1) < N) v |= l[k + 1] << (LB - sh);             // LB - sh in [1,29]
        if ((k + 2) < N && (2 * LB - sh) < 32) v |= l[k + 2] << (2 * LB - sh);
        w[i] = v;
    }
}

// ---------------------------------------------------------------------------------- Fp
#define KZG_FP_CONSTS const uint32_t FP_MOD[NFP] = FP_MOD_INIT;
KZG_HD Fp fp_zero() { Fp r; for (int i = 0; i < NFP; i++) r.l[i] = 0; return r; }
KZG_HD Fp fp_one() { const uint32_t c[NFP] = FP_ONE_INIT; Fp r; for (int i = 0; i < NFP; i++) r.l[i] = c[i]; return r; }
KZG_HD void fp_add(Fp &r, const Fp &a, const Fp &b) { KZG_FP_CONSTS mod_add<NFP>(r.l, a.l, b.l, FP_MOD); }
KZG_HD void fp_sub(Fp &r, const Fp &a, const Fp &b) { KZG_FP_CONSTS mod_sub<NFP>(r.l, a.l, b.l, FP_MOD); }
KZG_HD void fp_dbl(Fp &r, const Fp &a) { KZG_FP_CONSTS mod_add<NFP>(r.l, a.l, a.l, FP_MOD); }
KZG_HD void fp_neg(Fp &r, const Fp &a) { Fp z = fp_zero(); fp_sub(r, z, a); }
KZG_HD bool fp_is_zero(const Fp &a) { return ul_is_zero<NFP>(a.l); }
KZG_HD bool fp_eq(const Fp &a, const Fp &b) { return ul_eq<NFP>(a.l, b.l); }
#if defined(KZG_FP_MUL_CALL) && defined(__HIP_DEVICE_COMPILE__)
// Throughput kernels: ONE out-of-line Fp product per kernel image, operands and result in VGPRs (two 16-lane vector
// arguments = v0..v31, result v0..v15).  A fully inlined G1 formula is ~7k instructions (11 products): several of them
// plus their callers overflow the 64 KB instruction cache that neighbouring CUs share, and with 8 waves per CU at
// different program counters instruction fetch becomes the bottleneck.  With the call the hot loop is ~1.5k instructions.
typedef uint32_t fp_vec __attribute__((ext_vector_type(16)));
__device__ __attribute__((noinline)) inline fp_vec fp_mul_vec(fp_vec a, fp_vec b) {
    KZG_FP_CONSTS
    uint32_t x[NFP], y[NFP], r[NFP];
#pragma unroll
    for (int i = 0; i < NFP; i++) { x[i] = a[i]; y[i] = b[i]; }
    mont_mul<NFP>(r, x, y, FP_MOD, FP_INVW);
    fp_vec o;
#pragma unroll
    for (int i = 0; i < NFP; i++) o[i] = r[i];
    o[14] = 0; o[15] = 0;
    return o;
}
KZG_HD void fp_mul(Fp &r, const Fp &a, const Fp &b) {
    fp_vec x, y;
#pragma unroll
    for (int i = 0; i < NFP; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
    x[14] = 0; x[15] = 0; y[14] = 0; y[15] = 0;
    const fp_vec o = fp_mul_vec(x, y);
#pragma unroll
    for (int i = 0; i < NFP; i++) r.l[i] = o[i];
}
#elif defined(KZG_FP_MUL_NOINLINE)
KZG_HD_NOINLINE void fp_mul(Fp &r, const Fp &a, const Fp &b) { KZG_FP_CONSTS mont_mul<NFP>(r.l, a.l, b.l, FP_MOD, FP_INVW); }
#else
KZG_HD void fp_mul(Fp &r, const Fp &a, const Fp &b) { KZG_FP_CONSTS mont_mul<NFP>(r.l, a.l, b.l, FP_MOD, FP_INVW); }
#endif
#if defined(KZG_FP_MUL_CALL) && defined(__HIP_DEVICE_COMPILE__)
KZG_HD void fp_sqr(Fp &r, const Fp &a) { fp_mul(r, a, a); }
#elif defined(KZG_FP_MUL_NOINLINE)
KZG_HD_NOINLINE void fp_sqr(Fp &r, const Fp &a) { KZG_FP_CONSTS mont_sqr<NFP>(r.l, a.l, FP_MOD, FP_INVW); }
#else
KZG_HD void fp_sqr(Fp &r, const Fp &a) { KZG_FP_CONSTS mont_sqr<NFP>(r.l, a.l, FP_MOD, FP_INVW); }
#endif
// ---- lazy (unreduced) Fp: R = 2^406 leaves 25 bits above p, so values may run up to a few dozen p between reductions.
// Products of operands a < 2^6 p, b < 2^6 p come out below p (1 + 2^-13) without the final conditional subtraction; sums are
// plain limb additions with carry normalisation; differences add a multiple of p first.  Used by the accumulation loops of the
// MSM / bucket kernels (g1x_add_mixed_lazy), which spend ~15 % of a canonical addition in those subtractions and selects.
KZG_HD void fp_mul_lz(Fp &r, const Fp &a, const Fp &b) { KZG_FP_CONSTS mont_mul_lazy<NFP>(r.l, a.l, b.l, FP_MOD, FP_INVW); }
KZG_HD void fp_sqr_lz(Fp &r, const Fp &a) { KZG_FP_CONSTS mont_sqr<NFP, true>(r.l, a.l, FP_MOD, FP_INVW); }
// r = (a b + c d) / R, lazily, ONE Montgomery reduction for the two products (588 limb products instead of 784).  Operands normalised lazy values
// with a b + c d < 2^12 p^2 (e.g. a < 6p, b < 10p, c < 4p, d < 2p in the point additions); result < p (1 + 2^-13).
KZG_HD void fp_mul2_lz(Fp &r, const Fp &a, const Fp &b, const Fp &c, const Fp &d) { KZG_FP_CONSTS mont_mul2_lazy<NFP>(r.l, a.l, b.l, c.l, d.l, FP_MOD,
        FP_INVW); }
// r = a + (kp - b), kp a multiple of p above b: limbs normalised (signed carries), the top limb keeps the excess
KZG_HD void fp_sub_lz(Fp &r, const Fp &a, const Fp &b, const uint32_t *kp) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) {
        const int32_t t = (int32_t)a.l[i] + (int32_t)kp[i] - (int32_t)b.l[i] + c;
        if (i < NFP - 1) { c = t >> LB; r.l[i] = (uint32_t)t & LMASK; }
        else r.l[i] = (uint32_t)t;
    }
}
// r = a + b + b2 (b2 optional second addend), normalised
KZG_HD void fp_add_lz(Fp &r, const Fp &a, const Fp &b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) { const uint32_t t = a.l[i] + b.l[i] + c; if (i < NFP - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t; }
}
// value < 64 p -> canonical
KZG_HD void fp_canon64(Fp &r, const Fp &a) {
    const uint32_t m32[NFP] = FP_MOD32_INIT, m16[NFP] = FP_MOD16_INIT, m8[NFP] = FP_MOD8_INIT, m4[NFP] = FP_MOD4_INIT, m2[NFP] = FP_MOD2_INIT,
            m1[NFP] = FP_MOD_INIT;
    uint32_t v[NFP], s[NFP];
#pragma unroll
    for (int i = 0; i < NFP; i++) v[i] = a.l[i];
    const uint32_t *ms[6] = {m32, m16, m8, m4, m2, m1};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t br = ul_sub<NFP>(s, v, ms[k]);
#pragma unroll
        for (int i = 0; i < NFP; i++) v[i] = br ? v[i] : s[i];
    }
#pragma unroll
    for (int i = 0; i < NFP; i++) r.l[i] = v[i];
}
KZG_HD void fp_canon16(Fp &r, const Fp &a) { fp_canon64(r, a); }
// could the lazy value v in [0, 64p) be a multiple of p?  exact filter on the low limb: v = j p  =>  j = v0 * p^-1 mod 2^29
KZG_HD bool fp_maybe_zero_lz(const Fp &v) { return ((v.l[0] * FP_PINVW) & LMASK) < 64u; }
KZG_HD void fp_select(Fp &r, bool take_b, const Fp &a, const Fp &b) {
#pragma unroll
    for (int i = 0; i < NFP; i++) r.l[i] = take_b ? b.l[i] : a.l[i];
}
// canonical integer value (not Montgomery) as limbs
KZG_HD void fp_from_mont(uint32_t out[NFP], const Fp &a) {
    KZG_FP_CONSTS
    uint32_t one[NFP];
#pragma unroll
    for (int i = 0; i < NFP; i++) one[i] = i == 0 ? 1u : 0u;
    mont_mul<NFP>(out, a.l, one, FP_MOD, FP_INVW);
}
KZG_HD void fp_to_mont(Fp &r, const uint32_t in[NFP]) {
    KZG_FP_CONSTS
    const uint32_t R2[NFP] = FP_R2_INIT;
    mont_mul<NFP>(r.l, in, R2, FP_MOD, FP_INVW);
}
// a^e, e given as 12 plain 32-bit words (<= 384 bits), square-and-multiply MSB first.
// Rolled loop on purpose: a single fp_mul body in the instruction stream.
KZG_HD void fp_pow(Fp &r, const Fp &a, const uint32_t *e) {
    Fp acc = fp_one();
    bool started = false;
    for (int i = 383; i >= 0; i--) {
        if (started) fp_sqr(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            if (started) fp_mul(acc, acc, a); else { acc = a; started = true; }
        }
    }
    r = acc;
}
// a^(p-2): kept as the independent cross-check of the divstep inversion (modinv.h) that fp_inv uses
KZG_HD void fp_inv_fermat(Fp &r, const Fp &a) { const uint32_t e[12] = FP_EXP_INV_INIT; fp_pow(r, a, e); }
// sqrt for p = 3 mod 4: a^((p+1)/4); false if a is not a square
KZG_HD bool fp_sqrt(Fp &r, const Fp &a) {
    // a^((p+1)/4), p = 3 mod 4.  Left-to-right over the 379-bit exponent with a sliding window of four bits over the odd powers
    // a, a^3, .., a^15: 375 squarings + 78 + 8 products (two-bit window: 377 + 142 + 2; bit by bit: 378 + 228), on lazy products (no
    // reduction below p until the end).  The exponent is a constant: every branch below is uniform across a wave.
    const uint32_t e[12] = FP_EXP_SQRT_INIT;
    Fp tab[8], a2, acc;
    fp_sqr_lz(a2, a);
    tab[0] = a;
#pragma unroll
    for (int k = 1; k < 8; k++) fp_mul_lz(tab[k], tab[k - 1], a2);
    auto bit = [&](int i) -> uint32_t { return (e[i >> 5] >> (i & 31)) & 1u; };
    bool started = false;
    int i = 383;
    while (i >= 0 && !bit(i)) i--;
    while (i >= 0) {
        if (!bit(i)) { fp_sqr_lz(acc, acc); i--; continue; }
        int l = i + 1 < 4 ? i + 1 : 4;
        while (!bit(i - l + 1)) l--;                              // the window ends in a set bit
        uint32_t v = 0;
        for (int k = 0; k < l; k++) v = (v << 1) | bit(i - k);
        if (started) for (int k = 0; k < l; k++) fp_sqr_lz(acc, acc);
        switch (v >> 1) {                                         // constant table index in every arm: the table stays in registers
            case 0: if (started) fp_mul_lz(acc, acc, tab[0]); else acc = tab[0]; break;
            case 1: if (started) fp_mul_lz(acc, acc, tab[1]); else acc = tab[1]; break;
            case 2: if (started) fp_mul_lz(acc, acc, tab[2]); else acc = tab[2]; break;
            case 3: if (started) fp_mul_lz(acc, acc, tab[3]); else acc = tab[3]; break;
            case 4: if (started) fp_mul_lz(acc, acc, tab[4]); else acc = tab[4]; break;
            case 5: if (started) fp_mul_lz(acc, acc, tab[5]); else acc = tab[5]; break;
            case 6: if (started) fp_mul_lz(acc, acc, tab[6]); else acc = tab[6]; break;
            default: if (started) fp_mul_lz(acc, acc, tab[7]); else acc = tab[7]; break;
        }
        started = true;
        i -= l;
    }
    Fp s, chk; fp_canon64(s, acc); fp_sqr(chk, s);
    r = s;
    return fp_eq(chk, a);
}
// ZCash sign bit: canonical value > (p-1)/2
KZG_HD bool fp_is_lex_largest(const Fp &a) {
    const uint32_t half[NFP] = FP_HALF_INIT;
    uint32_t v[NFP], t[NFP];
    fp_from_mont(v, a);
    return ul_sub<NFP>(t, half, v) != 0;    // half - v borrows  <=>  v > half
}
// 48 big-endian bytes -> Fp.  Returns false if the integer is >= p.  mask_top3 clears the 3 flag bits.
KZG_HD bool fp_from_be48(Fp &r, const uint8_t *in, bool mask_top3) {
    KZG_FP_CONSTS
    uint32_t w[12], v[NFP], t[NFP];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint8_t *p = in + 4 * (11 - i);
        w[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
    if (mask_top3) w[11] &= 0x1fffffffu;
    words_to_limbs<NFP, 12>(v, w);
    bool ok = ul_sub<NFP>(t, v, FP_MOD) != 0;     // v < p
    fp_to_mont(r, v);
    return ok;
}
KZG_HD void fp_to_be48(uint8_t *out, const Fp &a) {
    uint32_t v[NFP], w[12];
    fp_from_mont(v, a);
    limbs_to_words<NFP, 12>(w, v);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint8_t *p = out + 4 * (11 - i);
        p[0] = (uint8_t)(w[i] >> 24); p[1] = (uint8_t)(w[i] >> 16); p[2] = (uint8_t)(w[i] >> 8); p[3] = (uint8_t)w[i];
    }
}

// ---------------------------------------------------------------------------------- Fr
#define KZG_FR_CONSTS const uint32_t FR_MOD[NFR] = FR_MOD_INIT;
KZG_HD Fr fr_zero() { Fr r; for (int i = 0; i < NFR; i++) r.l[i] = 0; return r; }
KZG_HD Fr fr_one() { const uint32_t c[NFR] = FR_ONE_INIT; Fr r; for (int i = 0; i < NFR; i++) r.l[i] = c[i]; return r; }
KZG_HD void fr_add(Fr &r, const Fr &a, const Fr &b) { KZG_FR_CONSTS mod_add<NFR>(r.l, a.l, b.l, FR_MOD); }
KZG_HD void fr_sub(Fr &r, const Fr &a, const Fr &b) { KZG_FR_CONSTS mod_sub<NFR>(r.l, a.l, b.l, FR_MOD); }
KZG_HD void fr_mul(Fr &r, const Fr &a, const Fr &b) { KZG_FR_CONSTS mont_mul<NFR>(r.l, a.l, b.l, FR_MOD, FR_INVW); }
// (a squaring of its own like Fp's: 45 + 81 limb products instead of 162 -- the twelve z powers per blob of the challenge kernel, the r-power ladders)
KZG_HD void fr_sqr(Fr &r, const Fr &a) { KZG_FR_CONSTS mont_sqr<NFR>(r.l, a.l, FR_MOD, FR_INVW); }
// lazy product: operands < ~2.6 r, result < 1.1 r, not canonical (see mont_mul_lazy)
KZG_HD void fr_mul_lazy(Fr &r, const Fr &a, const Fr &b) { KZG_FR_CONSTS mont_mul_lazy<NFR>(r.l, a.l, b.l, FR_MOD, FR_INVW); }
// r = a*b + c*d with one reduction (lazy; result < 1.1 r for a*b + c*d < ~5 r^2)
KZG_HD void fr_mul2_lazy(Fr &r, const Fr &a, const Fr &b, const Fr &c, const Fr &d) { KZG_FR_CONSTS mont_mul2_lazy<NFR>(r.l, a.l, b.l, c.l, d.l, FR_MOD,
        FR_INVW); }
// lazy sum: plain limb addition with carry normalisation, no reduction (value grows; keep chains short)
KZG_HD void fr_add_lazy(Fr &r, const Fr &a, const Fr &b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFR; i++) { uint32_t t = a.l[i] + b.l[i] + c; c = i < NFR - 1 ? t >> LB : 0u; r.l[i] = i < NFR - 1 ? (t & LMASK) : t; }
}
KZG_HD bool fr_is_zero(const Fr &a) { return ul_is_zero<NFR>(a.l); }
KZG_HD bool fr_eq(const Fr &a, const Fr &b) { return ul_eq<NFR>(a.l, b.l); }
KZG_HD void fr_select(Fr &r, bool take_b, const Fr &a, const Fr &b) {
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = take_b ? b.l[i] : a.l[i];
}
// canonical integer as 8 plain 32-bit words
KZG_HD void fr_to_words(uint32_t w[8], const Fr &a) {
    KZG_FR_CONSTS
    uint32_t one[NFR], v[NFR];
#pragma unroll
    for (int i = 0; i < NFR; i++) one[i] = i == 0 ? 1u : 0u;
    mont_mul<NFR>(v, a.l, one, FR_MOD, FR_INVW);
    limbs_to_words<NFR, 8>(w, v);
}
// any 256-bit integer (8 words) -> Montgomery residue mod r.  The product by R^2 reduces it:
// v < 2^256 < 2^261 and R2 < r  =>  v*R2 < r*2^261, within mont_mul's contract.
// (hash_to_bls_field, utils.rs:250-258, relies on exactly this reduction in blst_fr_from_scalar.)
KZG_HD void fr_from_words(Fr &r, const uint32_t w[8]) {
    KZG_FR_CONSTS
    const uint32_t R2[NFR] = FR_R2_INIT;
    uint32_t v[NFR];
    words_to_limbs<NFR, 8>(v, w);
    mont_mul<NFR>(r.l, v, R2, FR_MOD, FR_INVW);
}
// value (8 words) < r ?
KZG_HD bool fr_words_canonical(const uint32_t w[8]) {
    KZG_FR_CONSTS
    uint32_t v[NFR], t[NFR];
    words_to_limbs<NFR, 8>(v, w);
    return ul_sub<NFR>(t, v, FR_MOD) != 0;
}
KZG_HD void fr_inv_fermat(Fr &r, const Fr &a) {
    const uint32_t e[8] = FR_EXP_INV_INIT;
    Fr acc = a;   // bit 254 of r-2 is set
    for (int i = 253; i >= 0; i--) {
        fr_sqr(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) fr_mul(acc, acc, a);
    }
    r = acc;
}
// 32 big-endian bytes <-> 8 little-endian words
KZG_HD void be32_to_words(uint32_t w[8], const uint8_t *in) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *p = in + 4 * (7 - i);
        w[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
}
KZG_HD void words_to_be32(uint8_t *out, const uint32_t w[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t *p = out + 4 * (7 - i);
        p[0] = (uint8_t)(w[i] >> 24); p[1] = (uint8_t)(w[i] >> 16); p[2] = (uint8_t)(w[i] >> 8); p[3] = (uint8_t)w[i];
    }
}
// bytes_to_bls_field (utils.rs:262-275): false if the value is >= r
KZG_HD bool fr_from_be32_checked(Fr &r, const uint8_t *in) {
    uint32_t w[8]; be32_to_words(w, in);
    bool ok = fr_words_canonical(w);
    fr_from_words(r, w);
    return ok;
}
KZG_HD void fr_to_be32(uint8_t *out, const Fr &a) { uint32_t w[8]; fr_to_words(w, a); words_to_be32(out, w); }

}  // namespace kzg

#include "modinv.h"   // fp_inv / fr_inv: batched-divstep inversion (defined after the arithmetic it builds on)
