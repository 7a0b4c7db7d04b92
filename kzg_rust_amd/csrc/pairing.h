// pairing.h -- G2 (E': y^2 = x^3 + 4(1+u) over Fp2), optimal-ate Miller loop with PRECOMPUTED line
// coefficients, and the final-exponentiation "== 1" test.  Host+device (see field.h).
//
// Reference counterpart: pairings_verify (src/utils.rs:189-214), which calls blst_miller_loop twice,
// blst_fp12_mul, blst_final_exp and blst_fp12_is_one.  In every use the reference makes of it on the
// verify path the two G2 arguments are constants of the trusted setup ([tau]G2 and the G2 generator:
// kzg.rs:625) -- so the engine walks each G2 point ONCE at load_trusted_setup, stores the 68 line
// coefficient triples of its Miller loop, and a pairing check is then only Fp12 squarings and sparse
// products.  The boolean is the same as the reference's (it is a property of the group elements).
//
// Line through T (tangent) or T,Q (chord), evaluated at P = (xP, yP), after scaling by w^3 and an
// Fp2 factor (both are killed by the final exponentiation):
//     l = c0 + (c1 * xP) v + (c4 * yP) v w
//   tangent at Jacobian T=(X,Y,Z):  c0 = 3X^3 - 2Y^2,  c1 = -3X^2 Z^2,  c4 = Z3 Z^2   (Z3 = 2YZ)
//   chord T,Q (Q affine):           c0 = R xQ - yQ Z3, c1 = -R,         c4 = Z3        (H = xQ Z^2 - X, R = yQ Z^3 - Y, Z3 = Z H)
#pragma once
#include "tower.h"
#include "g1.h"

namespace kzg {

struct G2Affine { Fp2 x, y; };      // (0,0) = infinity
struct G2Jac { Fp2 x, y, z; };
struct LineCoeff { Fp2 c0, c1, c4; };
constexpr int N_LINES = 68;          // 63 doublings + 5 additions for |x| = 0xd201000000010000

KZG_HD bool g2a_is_inf(const G2Affine &a) { return fp2_is_zero(a.x) && fp2_is_zero(a.y); }

// 96 compressed bytes (x.c1 with flags || x.c0) -> affine.  0 ok, 1 bad encoding, 2 not on curve
// (blst_p2_uncompress, kzg.rs:877).
KZG_HD int g2_decompress(G2Affine &r, const uint8_t *in) {
    const uint8_t b0 = in[0];
    if (!(b0 & 0x80)) return 1;
    if (b0 & 0x40) {
        uint32_t acc = b0 & 0x3f;
        for (int i = 1; i < 96; i++) acc |= in[i];
        if (acc) return 1;
        r.x = fp2_zero(); r.y = fp2_zero();
        return 0;
    }
    Fp2 x, y, y2, b;
    if (!fp_from_be48(x.c1, in, true)) return 1;
    if (!fp_from_be48(x.c0, in + 48, false)) return 1;
    const uint32_t b4[NFP] = FP_B_INIT;
    for (int i = 0; i < NFP; i++) { b.c0.l[i] = b4[i]; b.c1.l[i] = b4[i]; }   // 4(1+u)
    fp2_sqr(y2, x); fp2_mul(y2, y2, x); fp2_add(y2, y2, b);
    if (!fp2_sqrt(y, y2)) return 2;
    bool want_large = (b0 & 0x20) != 0;
    if (fp2_is_lex_largest(y) != want_large) fp2_neg(y, y);
    r.x = x; r.y = y;
    return 0;
}

// Walk the Miller loop of Q once and record the line coefficients (used at load_trusted_setup).
KZG_HD void precompute_lines(LineCoeff *lines, const G2Affine &q) {
    Fp2 X = q.x, Y = q.y, Z = fp2_one();
    int n = 0;
    for (int i = 62; i >= 0; i--) {
        {   // tangent + doubling
            Fp2 A, B, C, D, E, F, Zsq, t, X3, Y3, Z3;
            fp2_sqr(A, X); fp2_sqr(B, Y); fp2_sqr(C, B); fp2_sqr(Zsq, Z);
            fp2_add(t, X, B); fp2_sqr(t, t); fp2_sub(t, t, A); fp2_sub(t, t, C); fp2_dbl(D, t);
            fp2_dbl(E, A); fp2_add(E, E, A);
            fp2_sqr(F, E);
            fp2_sub(X3, F, D); fp2_sub(X3, X3, D);
            fp2_mul(Z3, Y, Z); fp2_dbl(Z3, Z3);
            fp2_sub(t, D, X3); fp2_mul(Y3, E, t);
            fp2_dbl(C, C); fp2_dbl(C, C); fp2_dbl(C, C); fp2_sub(Y3, Y3, C);
            LineCoeff &L = lines[n++];
            fp2_mul(L.c0, E, X); fp2_sub(L.c0, L.c0, B); fp2_sub(L.c0, L.c0, B);
            fp2_mul(t, E, Zsq); fp2_neg(L.c1, t);
            fp2_mul(L.c4, Z3, Zsq);
            X = X3; Y = Y3; Z = Z3;
        }
        if ((BLS_X_ABS >> i) & 1) {   // chord + addition of Q
            Fp2 Zsq, U2, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
            fp2_sqr(Zsq, Z); fp2_mul(U2, q.x, Zsq);
            fp2_mul(S2, q.y, Z); fp2_mul(S2, S2, Zsq);
            fp2_sub(H, U2, X); fp2_sub(R, S2, Y);
            fp2_sqr(HH, H); fp2_mul(HHH, H, HH); fp2_mul(V, X, HH);
            fp2_sqr(X3, R); fp2_sub(X3, X3, HHH); fp2_sub(X3, X3, V); fp2_sub(X3, X3, V);
            fp2_sub(t, V, X3); fp2_mul(Y3, R, t); fp2_mul(t, Y, HHH); fp2_sub(Y3, Y3, t);
            fp2_mul(Z3, Z, H);
            LineCoeff &L = lines[n++];
            fp2_mul(L.c0, R, q.x); fp2_mul(t, q.y, Z3); fp2_sub(L.c0, L.c0, t);
            fp2_neg(L.c1, R);
            L.c4 = Z3;
            X = X3; Y = Y3; Z = Z3;
        }
    }
}

// f = conj( prod over the loop of  f^2 * l_1(P1) * l_2(P2) )  == ML(Q1,P1) * ML(Q2,P2).
// A point at infinity on either side contributes 1 (its lines are skipped), like blst_miller_loop on an
// infinite input.  P1/P2 are affine G1; lines1/lines2 come from precompute_lines(Q1/Q2).
KZG_HD void miller_loop_pair(Fp12 &f, const LineCoeff *lines1, const G1Affine &p1, const LineCoeff *lines2, const G1Affine &p2) {
    const bool use1 = !g1a_is_inf(p1), use2 = !g1a_is_inf(p2);
    f = fp12_one();
    int n = 0;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(f, f);
        const int steps = 1 + (int)((BLS_X_ABS >> i) & 1);
        for (int s = 0; s < steps; s++, n++) {
            Fp2 c1, c4;
            if (use1) {
                fp2_mul_fp(c1, lines1[n].c1, p1.x); fp2_mul_fp(c4, lines1[n].c4, p1.y);
                fp12_mul_by_014(f, lines1[n].c0, c1, c4);
            }
            if (use2) {
                fp2_mul_fp(c1, lines2[n].c1, p2.x); fp2_mul_fp(c4, lines2[n].c4, p2.y);
                fp12_mul_by_014(f, lines2[n].c0, c1, c4);
            }
        }
    }
    Fp12 c; fp12_conj(c, f); f = c;   // x < 0
}

// a^x for a in the cyclotomic subgroup (x < 0: conjugate = inverse there)
KZG_HD void cyc_exp_x(Fp12 &r, const Fp12 &a) {
    Fp12 acc = a;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(acc, acc);
        if ((BLS_X_ABS >> i) & 1) fp12_mul(acc, acc, a);
    }
    fp12_conj(r, acc);
}

// f^((p^12-1)/r) == 1 ?   Easy part (p^6-1)(p^2+1), then the hard part raised to the 3rd power,
// 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3  (gcd(3, r) = 1, so the "== 1" verdict is unchanged).
KZG_HD bool final_exp_is_one(const Fp12 &fin) {
    Fp12 f, t, a, b, c, d;
    fp12_conj(t, fin); fp12_inv(f, fin); fp12_mul(f, t, f);
    fp12_frob(t, f); fp12_frob(t, t); fp12_mul(f, t, f);
    cyc_exp_x(a, f); fp12_conj(t, f); fp12_mul(a, a, t);
    cyc_exp_x(b, a); fp12_conj(t, a); fp12_mul(a, b, t);
    cyc_exp_x(b, a); fp12_frob(t, a); fp12_mul(b, b, t);
    cyc_exp_x(c, b); cyc_exp_x(c, c);
    fp12_frob(t, b); fp12_frob(t, t); fp12_mul(c, c, t);
    fp12_conj(t, b); fp12_mul(c, c, t);
    fp12_sqr(d, f); fp12_mul(d, d, f);
    fp12_mul(c, c, d);
    return fp12_is_one(c);
}

}  // namespace kzg
