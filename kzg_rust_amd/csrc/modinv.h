// modinv.h -- modular inversion by batched half-delta divsteps (Bernstein-Yang "safegcd" in the form popularised by
// libsecp256k1's modinv32: signed 30-bit limbs, 30 divsteps per 2x2 transition matrix).  Host+device.
//
// Why: Fermat inversion is ~440 DEPENDENT Montgomery products (~265k instructions for Fp), and inversions sit on
// latency-critical single-lane paths here (Jacobian -> affine before compressing a commitment / proof, the two
// pairing inputs, the Fp6 inverse inside the final exponentiation).  30 rounds of (30 divsteps on the low limbs +
// two small matrix applications) cost ~25k instructions: ~10x shorter chain.  Branch-free (constant time), so all
// lanes of a wave run the same instruction stream.
//
// divsteps:  (zeta, f, g) -> zeta<0 and g odd ? (-zeta-2, g, (g+... )/2) ...  with zeta = -(delta + 1/2), starting at -1.
// Number of divsteps that provably suffices for M-bit moduli: ceil((45907 M + 26313) / 19929): 879 for 381 bits,
// 589 for 255 bits -> 30 resp. 20 rounds of 30.
#pragma once
// (included from the bottom of field.h)

namespace kzg {

template <int NL> struct Signed30 { int32_t v[NL]; };   // value = sum v[i] 2^(30 i); v[0..NL-2] in [0,2^30) when normalised

struct Trans2x2 { int32_t u, v, q, r; };

// 30 divsteps on the low 30 bits of f, g: returns the new zeta and the transition matrix t with
// t * [f, g] = 2^30 * [f', g'].
KZG_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans2x2 &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);            // all ones if zeta < 0
        uint32_t mask2 = 0u - (g & 1u);                  // all ones if g odd
        uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
        g += x & mask2; q += y & mask2; r += z & mask2;
        c1 &= mask2;                                     // zeta < 0 and g odd
        zeta = (zeta ^ (int32_t)c1) - 1;                 // -zeta-2 or zeta-1
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}

// [d, e] <- t * [d, e] / 2^30 mod m, keeping both in (-2m, m)
template <int NL> KZG_HD void update_de_30(Signed30<NL> &d, Signed30<NL> &e, const Trans2x2 &t, const int32_t *m, uint32_t m_inv30) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei;
    int64_t ce = (int64_t)q * di + (int64_t)r * ei;
    md -= (int32_t)((m_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((m_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)m[0] * md;
    ce += (int64_t)m[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        di = d.v[i]; ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei;
        ce += (int64_t)q * di + (int64_t)r * ei;
        cd += (int64_t)m[i] * md;
        ce += (int64_t)m[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[NL - 1] = (int32_t)cd;
    e.v[NL - 1] = (int32_t)ce;
}
// [f, g] <- t * [f, g] / 2^30 (exact)
template <int NL> KZG_HD void update_fg_30(Signed30<NL> &f, Signed30<NL> &g, const Trans2x2 &t) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi;
    int64_t cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        fi = f.v[i]; gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[NL - 1] = (int32_t)cf;
    g.v[NL - 1] = (int32_t)cg;
}
// r in (-2m, m) -> [0, m), negated first if sign < 0
template <int NL> KZG_HD void normalize_30(Signed30<NL> &r, int32_t sign, const int32_t *m) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    int32_t cond_add = r.v[NL - 1] >> 31;
    const int32_t cond_negate = sign >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t x = r.v[i] + (m[i] & cond_add);
        r.v[i] = (x ^ cond_negate) - cond_negate;
    }
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    cond_add = r.v[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] += m[i] & cond_add;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
}
// x <- x^-1 mod m (x in [0, m); 0 -> 0)
template <int NL, int ROUNDS> KZG_HD void modinv30(Signed30<NL> &x, const int32_t *m, uint32_t m_inv30) {
    Signed30<NL> d, e, f, g = x;
#pragma unroll
    for (int i = 0; i < NL; i++) { d.v[i] = 0; e.v[i] = i == 0 ? 1 : 0; f.v[i] = m[i]; }
    int32_t zeta = -1;
    for (int i = 0; i < ROUNDS; ++i) {
        Trans2x2 t;
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de_30<NL>(d, e, t, m, m_inv30);
        update_fg_30<NL>(f, g, t);
    }
    normalize_30<NL>(d, f.v[NL - 1], m);
    x = d;
}

// 32-bit words (NW little-endian words) <-> NL signed 30-bit limbs (non-negative values)
template <int NL, int NW> KZG_HD void words_to_s30(int32_t *l, const uint32_t *w) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int bit = 30 * k, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = wi < NW ? w[wi] : 0u, hi = (wi + 1) < NW ? w[wi + 1] : 0u;
        uint32_t v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        l[k] = (int32_t)(v & 0x3fffffffu);
    }
}
template <int NL, int NW> KZG_HD void s30_to_words(uint32_t *w, const int32_t *l) {
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int bit = 32 * i, k = bit / 30, sh = bit % 30;
        uint32_t v = k < NL ? ((uint32_t)l[k] >> sh) : 0u;
        if ((k + 1) < NL) v |= (uint32_t)l[k + 1] << (30 - sh);
        if ((k + 2) < NL && (60 - sh) < 32) v |= (uint32_t)l[k + 2] << (60 - sh);
        w[i] = v;
    }
}

// ---- Fp: 13 limbs, 30 rounds (900 >= 879 divsteps)
KZG_HD void fp_inv(Fp &r, const Fp &a) {
    const uint32_t mw[12] = FP_MOD_WORDS_INIT;
    const uint32_t r3[NFP] = FP_R3_INIT;
    const uint32_t FP_MOD[NFP] = FP_MOD_INIT;
    int32_t m[13];
    words_to_s30<13, 12>(m, mw);
    uint32_t w[12];
    limbs_to_words<NFP, 12>(w, a.l);                // the Montgomery residue aR as a plain integer
    Signed30<13> x;
    words_to_s30<13, 12>(x.v, w);
    modinv30<13, 30>(x, m, FP_MOD_INV30);            // (aR)^-1
    s30_to_words<13, 12>(w, x.v);
    uint32_t v[NFP];
    words_to_limbs<NFP, 12>(v, w);
    mont_mul<NFP>(r.l, v, r3, FP_MOD, FP_INVW);      // a^-1 R^-1 * R^3 / R = a^-1 R
}
// ---- Fr: 9 limbs, 20 rounds (600 >= 589 divsteps)
KZG_HD void fr_inv(Fr &r, const Fr &a) {
    const uint32_t mw[8] = FR_MOD_WORDS_INIT;
    const uint32_t r3[NFR] = FR_R3_INIT;
    const uint32_t FR_MOD[NFR] = FR_MOD_INIT;
    int32_t m[9];
    words_to_s30<9, 8>(m, mw);
    uint32_t w[8];
    limbs_to_words<NFR, 8>(w, a.l);
    Signed30<9> x;
    words_to_s30<9, 8>(x.v, w);
    modinv30<9, 20>(x, m, FR_MOD_INV30);
    s30_to_words<9, 8>(w, x.v);
    uint32_t v[NFR];
    words_to_limbs<NFR, 8>(v, w);
    mont_mul<NFR>(r.l, v, r3, FR_MOD, FR_INVW);
}

}  // namespace kzg
