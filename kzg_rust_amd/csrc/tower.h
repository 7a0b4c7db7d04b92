// tower.h -- Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v) for BLS12-381.
// Stands in for the blst internals behind blst_fp12_mul / blst_final_exp / blst_miller_loop
// (reference call sites: src/utils.rs:206-212).  Host+device (see field.h).
//
// KZG_MID marks the mid-level routines (Fp2 product and up).  In the single-lane pairing
// translation unit they are real functions (one body each in the instruction stream, so the whole
// pairing stays inside the instruction cache); define KZG_MID_INLINE to force-inline them instead.
#pragma once
#include "field.h"

#if defined(KZG_MID_INLINE)
#define KZG_MID KZG_HD
#else
#define KZG_MID KZG_HD_NOINLINE
#endif

namespace kzg {

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

// ------------------------------------------------------------------ Fp2
KZG_HD Fp2 fp2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
KZG_HD Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
KZG_HD void fp2_add(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); }
KZG_HD void fp2_sub(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_sub(r.c0, a.c0, b.c0); fp_sub(r.c1, a.c1, b.c1); }
KZG_HD void fp2_dbl(Fp2 &r, const Fp2 &a) { fp_dbl(r.c0, a.c0); fp_dbl(r.c1, a.c1); }
KZG_HD void fp2_neg(Fp2 &r, const Fp2 &a) { fp_neg(r.c0, a.c0); fp_neg(r.c1, a.c1); }
KZG_HD void fp2_conj(Fp2 &r, const Fp2 &a) { r.c0 = a.c0; fp_neg(r.c1, a.c1); }
KZG_HD bool fp2_is_zero(const Fp2 &a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
KZG_HD bool fp2_eq(const Fp2 &a, const Fp2 &b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
// (a0 + a1 u)(b0 + b1 u) with three base-field products
KZG_MID void fp2_mul(Fp2 &r, const Fp2 &a, const Fp2 &b) {
    Fp t0, t1, sa, sb, m;
    fp_mul(t0, a.c0, b.c0);
    fp_mul(t1, a.c1, b.c1);
    fp_add(sa, a.c0, a.c1);
    fp_add(sb, b.c0, b.c1);
    fp_mul(m, sa, sb);
    fp_sub(r.c0, t0, t1);
    fp_sub(m, m, t0);
    fp_sub(r.c1, m, t1);
}
// (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u
KZG_MID void fp2_sqr(Fp2 &r, const Fp2 &a) {
    Fp s, d, m;
    fp_add(s, a.c0, a.c1);
    fp_sub(d, a.c0, a.c1);
    fp_mul(m, a.c0, a.c1);
    fp_mul(r.c0, s, d);
    fp_dbl(r.c1, m);
}
KZG_HD void fp2_mul_fp(Fp2 &r, const Fp2 &a, const Fp &s) { fp_mul(r.c0, a.c0, s); fp_mul(r.c1, a.c1, s); }
// multiply by xi = 1 + u
KZG_HD void fp2_mul_xi(Fp2 &r, const Fp2 &a) {
    Fp t0, t1;
    fp_sub(t0, a.c0, a.c1);
    fp_add(t1, a.c0, a.c1);
    r.c0 = t0; r.c1 = t1;
}
KZG_MID void fp2_inv(Fp2 &r, const Fp2 &a) {
    Fp n, t;
    fp_sqr(n, a.c0); fp_sqr(t, a.c1); fp_add(n, n, t);
    fp_inv(n, n);
    fp_mul(r.c0, a.c0, n);
    fp_mul(t, a.c1, n); fp_neg(r.c1, t);
}
KZG_HD void fp2_pow(Fp2 &r, const Fp2 &a, const uint32_t *e) {
    Fp2 acc = fp2_one();
    for (int i = 383; i >= 0; i--) {
        fp2_sqr(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) fp2_mul(acc, acc, a);
    }
    r = acc;
}
// Square root in Fp2 for p = 3 mod 4 (Adj & Rodriguez-Henriquez, eprint 2012/685, Alg. 9)
KZG_HD bool fp2_sqrt(Fp2 &r, const Fp2 &a) {
    if (fp2_is_zero(a)) { r = a; return true; }
    const uint32_t e34[12] = FP_EXP_P34_INIT;
    const uint32_t e12[12] = FP_EXP_P12_INIT;
    Fp2 a1, alpha, x0, res, chk, neg1;
    fp2_pow(a1, a, e34);
    fp2_sqr(alpha, a1); fp2_mul(alpha, alpha, a);
    fp2_mul(x0, a1, a);
    neg1 = fp2_zero(); Fp one = fp_one(); fp_neg(neg1.c0, one);
    if (fp2_eq(alpha, neg1)) {
        fp_neg(res.c0, x0.c1); res.c1 = x0.c0;
    } else {
        Fp2 b;
        fp_add(alpha.c0, alpha.c0, one);
        fp2_pow(b, alpha, e12);
        fp2_mul(res, b, x0);
    }
    fp2_sqr(chk, res);
    r = res;
    return fp2_eq(chk, a);
}
// ZCash sign of an Fp2 element: compare c1 first, then c0
KZG_HD bool fp2_is_lex_largest(const Fp2 &a) {
    return fp_is_zero(a.c1) ? fp_is_lex_largest(a.c0) : fp_is_lex_largest(a.c1);
}

// ------------------------------------------------------------------ Fp6
KZG_HD void fp6_add(Fp6 &r, const Fp6 &a, const Fp6 &b) { fp2_add(r.c0, a.c0, b.c0); fp2_add(r.c1, a.c1, b.c1); fp2_add(r.c2, a.c2, b.c2); }
KZG_HD void fp6_sub(Fp6 &r, const Fp6 &a, const Fp6 &b) { fp2_sub(r.c0, a.c0, b.c0); fp2_sub(r.c1, a.c1, b.c1); fp2_sub(r.c2, a.c2, b.c2); }
KZG_HD void fp6_neg(Fp6 &r, const Fp6 &a) { fp2_neg(r.c0, a.c0); fp2_neg(r.c1, a.c1); fp2_neg(r.c2, a.c2); }
KZG_HD Fp6 fp6_zero() { Fp6 r; r.c0 = fp2_zero(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
// Karatsuba over Fp2: 6 Fp2 products
KZG_MID void fp6_mul(Fp6 &r, const Fp6 &a, const Fp6 &b) {
    Fp2 v0, v1, v2, s, t, m, o0, o1, o2;
    fp2_mul(v0, a.c0, b.c0); fp2_mul(v1, a.c1, b.c1); fp2_mul(v2, a.c2, b.c2);
    fp2_add(s, a.c1, a.c2); fp2_add(t, b.c1, b.c2); fp2_mul(m, s, t);
    fp2_sub(m, m, v1); fp2_sub(m, m, v2); fp2_mul_xi(m, m); fp2_add(o0, v0, m);
    fp2_add(s, a.c0, a.c1); fp2_add(t, b.c0, b.c1); fp2_mul(m, s, t);
    fp2_sub(m, m, v0); fp2_sub(m, m, v1); fp2_mul_xi(s, v2); fp2_add(o1, m, s);
    fp2_add(s, a.c0, a.c2); fp2_add(t, b.c0, b.c2); fp2_mul(m, s, t);
    fp2_sub(m, m, v0); fp2_sub(m, m, v2); fp2_add(o2, m, v1);
    r.c0 = o0; r.c1 = o1; r.c2 = o2;
}
// multiply by v: (c0, c1, c2) -> (xi c2, c0, c1)
KZG_HD void fp6_mul_v(Fp6 &r, const Fp6 &a) {
    Fp2 t; fp2_mul_xi(t, a.c2);
    r.c2 = a.c1; r.c1 = a.c0; r.c0 = t;
}
// a * (b0 + b1 v)
KZG_MID void fp6_mul_by_01(Fp6 &r, const Fp6 &a, const Fp2 &b0, const Fp2 &b1) {
    Fp2 p00, p11, p21, t, o0, o1, o2;
    fp2_mul(p00, a.c0, b0); fp2_mul(p11, a.c1, b1); fp2_mul(p21, a.c2, b1);
    fp2_mul_xi(t, p21); fp2_add(o0, p00, t);
    fp2_mul(t, a.c0, b1); fp2_mul(o1, a.c1, b0); fp2_add(o1, o1, t);
    fp2_mul(o2, a.c2, b0); fp2_add(o2, o2, p11);
    r.c0 = o0; r.c1 = o1; r.c2 = o2;
}
// a * (b1 v)
KZG_MID void fp6_mul_by_1(Fp6 &r, const Fp6 &a, const Fp2 &b1) {
    Fp2 o0, o1, o2;
    fp2_mul(o0, a.c2, b1); fp2_mul_xi(o0, o0);
    fp2_mul(o1, a.c0, b1);
    fp2_mul(o2, a.c1, b1);
    r.c0 = o0; r.c1 = o1; r.c2 = o2;
}
KZG_MID void fp6_inv(Fp6 &r, const Fp6 &a) {
    Fp2 t0, t1, t2, m, d;
    fp2_sqr(t0, a.c0); fp2_mul(m, a.c1, a.c2); fp2_mul_xi(m, m); fp2_sub(t0, t0, m);
    fp2_sqr(t1, a.c2); fp2_mul_xi(t1, t1); fp2_mul(m, a.c0, a.c1); fp2_sub(t1, t1, m);
    fp2_sqr(t2, a.c1); fp2_mul(m, a.c0, a.c2); fp2_sub(t2, t2, m);
    fp2_mul(d, a.c2, t1); fp2_mul(m, a.c1, t2); fp2_add(d, d, m); fp2_mul_xi(d, d);
    fp2_mul(m, a.c0, t0); fp2_add(d, d, m);
    fp2_inv(d, d);
    fp2_mul(r.c0, t0, d); fp2_mul(r.c1, t1, d); fp2_mul(r.c2, t2, d);
}
KZG_HD void fp2_load_const(Fp2 &r, const uint32_t *c0, const uint32_t *c1) {
    for (int i = 0; i < NFP; i++) { r.c0.l[i] = c0[i]; r.c1.l[i] = c1[i]; }
}
// Frobenius x -> x^p on Fp6: conj coefficients, scale by xi^((p-1)/3), xi^(2(p-1)/3)
KZG_MID void fp6_frob(Fp6 &r, const Fp6 &a) {
    const uint32_t v1c0[NFP] = FROB_V1_C0_INIT, v1c1[NFP] = FROB_V1_C1_INIT;
    const uint32_t v2c0[NFP] = FROB_V2_C0_INIT, v2c1[NFP] = FROB_V2_C1_INIT;
    Fp2 k1, k2, t;
    fp2_load_const(k1, v1c0, v1c1); fp2_load_const(k2, v2c0, v2c1);
    fp2_conj(r.c0, a.c0);
    fp2_conj(t, a.c1); fp2_mul(r.c1, t, k1);
    fp2_conj(t, a.c2); fp2_mul(r.c2, t, k2);
}

// ------------------------------------------------------------------ Fp12
KZG_HD Fp12 fp12_one() { Fp12 r; r.c0 = fp6_zero(); r.c1 = fp6_zero(); r.c0.c0.c0 = fp_one(); return r; }
KZG_HD bool fp12_is_one(const Fp12 &a) {
    Fp one = fp_one();
    bool ok = fp_eq(a.c0.c0.c0, one) && fp_is_zero(a.c0.c0.c1);
    ok = ok && fp2_is_zero(a.c0.c1) && fp2_is_zero(a.c0.c2);
    ok = ok && fp2_is_zero(a.c1.c0) && fp2_is_zero(a.c1.c1) && fp2_is_zero(a.c1.c2);
    return ok;
}
KZG_MID void fp12_mul(Fp12 &r, const Fp12 &a, const Fp12 &b) {
    Fp6 t0, t1, s, t, m;
    fp6_mul(t0, a.c0, b.c0);
    fp6_mul(t1, a.c1, b.c1);
    fp6_add(s, a.c0, a.c1); fp6_add(t, b.c0, b.c1);
    fp6_mul(m, s, t);
    fp6_sub(m, m, t0); fp6_sub(r.c1, m, t1);
    fp6_mul_v(t1, t1); fp6_add(r.c0, t0, t1);
}
// (a0 + a1 w)^2 = (a0+a1)(a0 + v a1) - a0a1 - v a0a1  +  2 a0a1 w
KZG_MID void fp12_sqr(Fp12 &r, const Fp12 &a) {
    Fp6 ab, s, t, m, vab;
    fp6_mul(ab, a.c0, a.c1);
    fp6_add(s, a.c0, a.c1);
    fp6_mul_v(t, a.c1); fp6_add(t, t, a.c0);
    fp6_mul(m, s, t);
    fp6_mul_v(vab, ab);
    fp6_sub(m, m, ab); fp6_sub(r.c0, m, vab);
    fp6_add(r.c1, ab, ab);
}
KZG_HD void fp12_conj(Fp12 &r, const Fp12 &a) { r.c0 = a.c0; fp6_neg(r.c1, a.c1); }
KZG_MID void fp12_inv(Fp12 &r, const Fp12 &a) {
    Fp6 t0, t1;
    fp6_mul(t0, a.c0, a.c0);
    fp6_mul(t1, a.c1, a.c1); fp6_mul_v(t1, t1);
    fp6_sub(t0, t0, t1);
    fp6_inv(t0, t0);
    fp6_mul(r.c0, a.c0, t0);
    fp6_mul(t1, a.c1, t0); fp6_neg(r.c1, t1);
}
KZG_MID void fp12_frob(Fp12 &r, const Fp12 &a) {
    const uint32_t wc0[NFP] = FROB_W_C0_INIT, wc1[NFP] = FROB_W_C1_INIT;
    Fp2 kw; fp2_load_const(kw, wc0, wc1);
    Fp6 t;
    fp6_frob(r.c0, a.c0);
    fp6_frob(t, a.c1);
    fp2_mul(r.c1.c0, t.c0, kw); fp2_mul(r.c1.c1, t.c1, kw); fp2_mul(r.c1.c2, t.c2, kw);
}
// f *= (l0 + l1 v + l4 v w): the sparse element a Miller-loop line evaluates to (see pairing.h)
KZG_MID void fp12_mul_by_014(Fp12 &f, const Fp2 &l0, const Fp2 &l1, const Fp2 &l4) {
    Fp6 t0, t1, s, m; Fp2 l14;
    fp6_mul_by_01(t0, f.c0, l0, l1);
    fp6_mul_by_1(t1, f.c1, l4);
    fp6_add(s, f.c0, f.c1);
    fp2_add(l14, l1, l4);
    fp6_mul_by_01(m, s, l0, l14);
    fp6_sub(m, m, t0); fp6_sub(f.c1, m, t1);
    fp6_mul_v(t1, t1); fp6_add(f.c0, t0, t1);
}

}  // namespace kzg
