// g1.h -- BLS12-381 G1 (E: y^2 = x^3 + 4 over Fp): Jacobian group law, ZCash (de)compression,
// subgroup check, scalar multiplication.  Host+device (see field.h).
// Reference counterparts (all blst calls; blst itself is not in /root/reference):
//   blst_p1_add_or_double / blst_p1_cneg   utils.rs:166-167, 339      -> g1_add, g1_add_mixed, g1_neg
//   blst_p1_mult (nbits = 256)             utils.rs:126-140           -> g1_mul_words
//   blst_p1_uncompress / blst_p1_in_g1     utils.rs:282-310           -> g1_decompress, g1_in_subgroup
//   blst_p1_compress                       utils.rs:221-227           -> g1_compress_affine
// Affine points use (0,0) for the point at infinity ((0,0) is not on the curve); Jacobian uses z = 0.
#pragma once
#include "field.h"

#if defined(KZG_MID_INLINE)
#define KZG_G1_MID KZG_HD
#else
#define KZG_G1_MID KZG_HD_NOINLINE
#endif

namespace kzg {

struct G1Affine { Fp x, y; };
struct G1Jac { Fp x, y, z; };

KZG_HD G1Jac g1_inf() { G1Jac r; r.x = fp_zero(); r.y = fp_zero(); r.z = fp_zero(); return r; }
KZG_HD G1Affine g1a_inf() { G1Affine r; r.x = fp_zero(); r.y = fp_zero(); return r; }
KZG_HD bool g1_is_inf(const G1Jac &a) { return fp_is_zero(a.z); }
KZG_HD bool g1a_is_inf(const G1Affine &a) { return fp_is_zero(a.x) && fp_is_zero(a.y); }
KZG_HD void g1_neg(G1Jac &r, const G1Jac &a) { r.x = a.x; r.z = a.z; fp_neg(r.y, a.y); }
KZG_HD void g1a_neg(G1Affine &r, const G1Affine &a) { r.x = a.x; fp_neg(r.y, a.y); }
KZG_HD void g1_from_affine(G1Jac &r, const G1Affine &a) {
    bool inf = g1a_is_inf(a);
    r.x = a.x; r.y = a.y;
    Fp one = fp_one(), zero = fp_zero();
    fp_select(r.z, inf, one, zero);
}

// 2P, a = 0 (dbl-2009-l): 2M + 5S.  P = infinity stays infinity (z3 = 2 y z = 0).
KZG_G1_MID void g1_dbl(G1Jac &r, const G1Jac &p) {
    Fp A, B, C, D, E, F, t;
    fp_sqr(A, p.x);
    fp_sqr(B, p.y);
    fp_sqr(C, B);
    fp_add(t, p.x, B); fp_sqr(t, t); fp_sub(t, t, A); fp_sub(t, t, C); fp_dbl(D, t);
    fp_dbl(E, A); fp_add(E, E, A);
    fp_sqr(F, E);
    fp_mul(t, p.y, p.z);          // before x,y are overwritten (r may alias p)
    Fp X3, Y3;
    fp_sub(X3, F, D); fp_sub(X3, X3, D);
    fp_sub(D, D, X3); fp_mul(Y3, E, D);
    fp_dbl(C, C); fp_dbl(C, C); fp_dbl(C, C);
    fp_sub(r.y, Y3, C);
    r.x = X3;
    fp_dbl(r.z, t);
}

// Complete Jacobian + affine addition (add-or-double semantics, infinity-aware).
KZG_G1_MID void g1_add_mixed(G1Jac &r, const G1Jac &a, const G1Affine &b) {
    if (g1a_is_inf(b)) { r = a; return; }
    if (g1_is_inf(a)) { g1_from_affine(r, b); return; }
    Fp Z1Z1, U2, S2, H, R, HH, HHH, V, t;
    fp_sqr(Z1Z1, a.z);
    fp_mul(U2, b.x, Z1Z1);
    fp_mul(S2, b.y, a.z); fp_mul(S2, S2, Z1Z1);
    fp_sub(H, U2, a.x);
    fp_sub(R, S2, a.y);
    if (fp_is_zero(H)) {
        if (fp_is_zero(R)) { G1Jac d = a; g1_dbl(r, d); } else { r = g1_inf(); }
        return;
    }
    fp_sqr(HH, H); fp_mul(HHH, H, HH); fp_mul(V, a.x, HH);
    Fp X3, Y3;
    fp_sqr(X3, R); fp_sub(X3, X3, HHH); fp_sub(X3, X3, V); fp_sub(X3, X3, V);
    fp_sub(t, V, X3); fp_mul(Y3, R, t);
    fp_mul(t, a.y, HHH); fp_sub(Y3, Y3, t);
    fp_mul(r.z, a.z, H);
    r.x = X3; r.y = Y3;
}

// Complete Jacobian + Jacobian addition.
KZG_G1_MID void g1_add(G1Jac &r, const G1Jac &a, const G1Jac &b) {
    if (g1_is_inf(a)) { r = b; return; }
    if (g1_is_inf(b)) { r = a; return; }
    Fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, HH, HHH, V, t;
    fp_sqr(Z1Z1, a.z); fp_sqr(Z2Z2, b.z);
    fp_mul(U1, a.x, Z2Z2); fp_mul(U2, b.x, Z1Z1);
    fp_mul(S1, a.y, b.z); fp_mul(S1, S1, Z2Z2);
    fp_mul(S2, b.y, a.z); fp_mul(S2, S2, Z1Z1);
    fp_sub(H, U2, U1);
    fp_sub(R, S2, S1);
    if (fp_is_zero(H)) {
        if (fp_is_zero(R)) { G1Jac d = a; g1_dbl(r, d); } else { r = g1_inf(); }
        return;
    }
    fp_sqr(HH, H); fp_mul(HHH, H, HH); fp_mul(V, U1, HH);
    Fp X3, Y3, Z3;
    fp_sqr(X3, R); fp_sub(X3, X3, HHH); fp_sub(X3, X3, V); fp_sub(X3, X3, V);
    fp_sub(t, V, X3); fp_mul(Y3, R, t);
    fp_mul(t, S1, HHH); fp_sub(Y3, Y3, t);
    fp_mul(Z3, a.z, b.z); fp_mul(Z3, Z3, H);
    r.x = X3; r.y = Y3; r.z = Z3;
}

// Extended Jacobian ("XYZZ") accumulator: x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2; infinity: ZZ = 0.  Its mixed addition is
// 8M + 2S (madd-2008-s) against 8M + 3S for the Jacobian form above, at the price of one more coordinate: the form the
// accumulation loops use (wide-table MSM, bucket lincomb); everything else stays Jacobian.
struct G1X { Fp x, y, zz, zzz; };
KZG_HD G1X g1x_inf() { G1X r; r.x = fp_zero(); r.y = fp_zero(); r.zz = fp_zero(); r.zzz = fp_zero(); return r; }
KZG_HD bool g1x_is_inf(const G1X &a) { return fp_is_zero(a.zz); }
// Complete: handles a or b at infinity and b = +-a (doubling of the affine point, mdbl-2008-s-1 with a = 0).
KZG_G1_MID void g1x_add_mixed(G1X &r, const G1X &a, const G1Affine &b) {
    if (g1a_is_inf(b)) { r = a; return; }
    if (g1x_is_inf(a)) { r.x = b.x; r.y = b.y; r.zz = fp_one(); r.zzz = fp_one(); return; }
    Fp U2, S2, P, R, PP, PPP, Q, t;
    fp_mul(U2, b.x, a.zz);
    fp_mul(S2, b.y, a.zzz);
    fp_sub(P, U2, a.x);
    fp_sub(R, S2, a.y);
    if (fp_is_zero(P)) {
        if (!fp_is_zero(R)) { r = g1x_inf(); return; }
        Fp U, V, W, S, M;
        fp_dbl(U, b.y); fp_sqr(V, U); fp_mul(W, U, V); fp_mul(S, b.x, V);
        fp_sqr(M, b.x); fp_dbl(t, M); fp_add(M, M, t);
        Fp X3, Y3;
        fp_sqr(X3, M); fp_sub(X3, X3, S); fp_sub(X3, X3, S);
        fp_sub(t, S, X3); fp_mul(Y3, M, t); fp_mul(t, W, b.y); fp_sub(Y3, Y3, t);
        r.x = X3; r.y = Y3; r.zz = V; r.zzz = W;
        return;
    }
    fp_sqr(PP, P); fp_mul(PPP, P, PP); fp_mul(Q, a.x, PP);
    Fp X3, Y3;
    fp_sqr(X3, R); fp_sub(X3, X3, PPP); fp_sub(X3, X3, Q); fp_sub(X3, X3, Q);
    fp_sub(t, Q, X3); fp_mul(Y3, R, t);
    fp_mul(t, a.y, PPP); fp_sub(Y3, Y3, t);
    fp_mul(r.zz, a.zz, PP);
    fp_mul(r.zzz, a.zzz, PPP);
    r.x = X3; r.y = Y3;
}
// The same addition on LAZY coordinates, for the accumulation loops: no conditional subtraction after any product, sum or
// difference.  Invariant of the accumulator (in and out): X < 8p, Y < 4p, ZZ, ZZZ < 2p (limbs normalised); b canonical.
//   U2, S2 < 2p;  P = U2 + 8p - X in (0, 10p);  R = S2 + 4p - Y in (0, 6p);  PP, PPP, Q < 2p
//   X3 = R^2 + (2p - PPP) + 2 (2p - Q) in (0, 8p);  Y3 = R (Q + 8p - X3) + (2p - Y PPP) in (0, 4p);  ZZ3, ZZZ3 < 2p
// `started` replaces the ZZ == 0 test (a lazy zero need not be all-zero limbs).  Anything unusual -- b at infinity, or P
// possibly = 0 mod p (exact low-limb filter, one chance in 2^25 for a random P) -- goes through the canonical routine above.
KZG_G1_MID void g1x_add_mixed_lazy(G1X &acc, bool &started, const G1Affine &b) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m4[NFP] = FP_MOD4_INIT, m8[NFP] = FP_MOD8_INIT;
    if (!started) {
        if (g1a_is_inf(b)) return;
        acc.x = b.x; acc.y = b.y; acc.zz = fp_one(); acc.zzz = fp_one(); started = true;
        return;
    }
    Fp U2, S2, P, R;
    fp_mul_lz(U2, b.x, acc.zz);
    fp_mul_lz(S2, b.y, acc.zzz);
    fp_sub_lz(P, U2, acc.x, m8);
    fp_sub_lz(R, S2, acc.y, m4);
    if (fp_maybe_zero_lz(P) || g1a_is_inf(b)) {                  // rare: redo canonically (doubling / inverse / no-op)
        G1X c; fp_canon16(c.x, acc.x); fp_canon16(c.y, acc.y); fp_canon16(c.zz, acc.zz); fp_canon16(c.zzz, acc.zzz);
        g1x_add_mixed(c, c, b);                                   // complete; c is canonical, hence within the invariant
        acc = c;
        started = !g1x_is_inf(c);
        return;
    }
    Fp PP, PPP, Q, t, u;
    fp_sqr_lz(PP, P); fp_mul_lz(PPP, P, PP); fp_mul_lz(Q, acc.x, PP);
    Fp X3, Y3;
    fp_sqr_lz(X3, R);
    fp_sub_lz(t, X3, PPP, m2);                                    // R^2 + 2p - PPP        in (0, 4p)
    fp_sub_lz(u, t, Q, m2);                                       //  ... + 2p - Q         in (0, 6p)
    fp_sub_lz(X3, u, Q, m2);                                      //  ... + 2p - Q         in (0, 8p)
    fp_sub_lz(t, Q, X3, m8);                                      // Q + 8p - X3           in (0, 10p)
#if defined(KZG_G1_ADD_MUL2)
    // R (Q - X3) - Y PPP as two products under ONE Montgomery reduction (588 limb products instead of 784: -5 % of the addition).  Only where
    // the registers allow: the four operands are live together -- the fixed-base MSM (k_msm_wide.hip: +2.8 % commitments) takes it, the bucket
    // kernel of the batch linear combination went from 256 to 262 VGPRs with it and from 15.3 to 21.2 ms per 8192 batches.
    { const Fp z = fp_zero(); fp_sub_lz(u, z, acc.y, m4); }       // 4p - Y                in (0, 4p]
    fp_mul2_lz(Y3, R, t, u, PPP);                                 // < 2p
#else
    fp_mul_lz(Y3, R, t);
    fp_mul_lz(t, acc.y, PPP);
    fp_sub_lz(Y3, Y3, t, m2);                                     // in (0, 4p)
#endif
    fp_mul_lz(acc.zz, acc.zz, PP);
    fp_mul_lz(acc.zzz, acc.zzz, PPP);
    acc.x = X3; acc.y = Y3;
}
// lazy accumulator -> canonical XYZZ (infinity if nothing was added)
KZG_G1_MID void g1x_from_lazy(G1X &r, const G1X &acc, bool started) {
    if (!started) { r = g1x_inf(); return; }
    fp_canon16(r.x, acc.x); fp_canon16(r.y, acc.y); fp_canon16(r.zz, acc.zz); fp_canon16(r.zzz, acc.zzz);
}

// -> Jacobian without an inversion: take Z' = ZZ, then Z'^2 = ZZ^2 gives X' = X * ZZ and Z'^3 = ZZ^3 = ZZZ^2 gives Y' = Y * ZZZ.
KZG_G1_MID void g1x_to_jac(G1Jac &r, const G1X &a) {
    if (g1x_is_inf(a)) { r = g1_inf(); return; }
    fp_mul(r.x, a.x, a.zz);
    fp_mul(r.y, a.y, a.zzz);
    r.z = a.zz;
}

// XYZZ + XYZZ on lazy coordinates (add-2008-s, 12M + 2S), both operands within the accumulator invariant above (X < 8p, Y < 4p,
// ZZ, ZZZ < 2p) and the result within it again:
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1 < 2p;  P = U2 + 2p - U1, R = S2 + 2p - S1 in (0, 4p);  PP, PPP, Q < 2p
//   X3 = R^2 + (2p - PPP) + 2 (2p - Q) in (0, 8p);  Y3 = R (Q + 8p - X3) + (2p - S1 PPP) in (0, 4p);  ZZ3, ZZZ3 < 2p
// An operand at infinity (ZZ = 0 mod p) or P possibly = 0 mod p (exact low-limb filters) goes through the canonical Jacobian addition.
KZG_G1_MID void g1x_add_lazy2(G1X &r, const G1X &a, const G1X &b) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT;
    Fp U1, U2, S1, S2, ZZ12, ZZZ12, P, R;
    fp_mul_lz(U1, a.x, b.zz); fp_mul_lz(U2, b.x, a.zz);
    fp_mul_lz(S1, a.y, b.zzz); fp_mul_lz(S2, b.y, a.zzz);
    fp_mul_lz(ZZ12, a.zz, b.zz); fp_mul_lz(ZZZ12, a.zzz, b.zzz);     // (the operands are dead from here on)
    fp_sub_lz(P, U2, U1, m2);
    fp_sub_lz(R, S2, S1, m2);
    if (fp_maybe_zero_lz(P) || fp_maybe_zero_lz(a.zz) || fp_maybe_zero_lz(b.zz)) {      // rare: redo canonically
        G1X ca, cb; g1x_from_lazy(ca, a, true); g1x_from_lazy(cb, b, true);
        G1Jac ja, jb, jr; g1x_to_jac(ja, ca); g1x_to_jac(jb, cb);
        g1_add(jr, ja, jb);
        if (g1_is_inf(jr)) { r = g1x_inf(); return; }
        r.x = jr.x; r.y = jr.y; fp_sqr(r.zz, jr.z); fp_mul(r.zzz, r.zz, jr.z);
        return;
    }
    Fp PP, PPP, Q, t, u, X3, Y3;
    fp_sqr_lz(PP, P); fp_mul_lz(PPP, P, PP); fp_mul_lz(Q, U1, PP);
    fp_sqr_lz(X3, R);
    fp_sub_lz(t, X3, PPP, m2);                                    // in (0, 4p)
    fp_sub_lz(u, t, Q, m2);                                       // in (0, 6p)
    fp_sub_lz(X3, u, Q, m2);                                      // in (0, 8p)
    fp_sub_lz(t, Q, X3, m8);                                      // in (0, 10p)
    fp_mul_lz(Y3, R, t);
    fp_mul_lz(t, S1, PPP);
    fp_sub_lz(Y3, Y3, t, m2);                                     // in (0, 4p)
    fp_mul_lz(r.zz, ZZ12, PP);
    fp_mul_lz(r.zzz, ZZZ12, PPP);
    r.x = X3; r.y = Y3;
}


// A G1 argument of the pairing check without the inversion of an affine conversion: (X Z, Y, Z^3) of the Jacobian point.  The line
// functions are evaluated at x = X / Z^2, y = Y / Z^3 SCALED by Z^3 (l0 Z^3 + l_x (X Z) + l_y Y): a factor in Fp per line, which the
// final exponentiation kills (p - 1 divides (p^12 - 1) / r).  az = 0: the point at infinity (the pair contributes 1).
struct PairPt { Fp ax, ay, az; };
KZG_G1_MID void pairpt_from_jac(PairPt &r, const G1Jac &a, bool negate) {
    if (g1_is_inf(a)) { r.ax = fp_zero(); r.ay = fp_zero(); r.az = fp_zero(); return; }
    Fp z2;
    fp_mul(r.ax, a.x, a.z);
    fp_sqr(z2, a.z); fp_mul(r.az, z2, a.z);
    r.ay = a.y;
    if (negate) fp_neg(r.ay, r.ay);
}
KZG_HD void pairpt_from_affine(PairPt &r, const G1Affine &a) {
    if (g1a_is_inf(a)) { r.ax = fp_zero(); r.ay = fp_zero(); r.az = fp_zero(); return; }
    r.ax = a.x; r.ay = a.y; r.az = fp_one();
}
KZG_G1_MID void g1_to_affine(G1Affine &r, const G1Jac &a) {
    if (g1_is_inf(a)) { r = g1a_inf(); return; }
    Fp zi, zi2, zi3;
    fp_inv(zi, a.z); fp_sqr(zi2, zi); fp_mul(zi3, zi2, zi);
    fp_mul(r.x, a.x, zi2); fp_mul(r.y, a.y, zi3);
}
// x = (X Z) / Z^3, y = Y / Z^3
KZG_G1_MID void pairpt_to_affine(G1Affine &r, const PairPt &a) {
    if (fp_is_zero(a.az)) { r = g1a_inf(); return; }
    Fp zi; fp_inv(zi, a.az);
    fp_mul(r.x, a.ax, zi); fp_mul(r.y, a.ay, zi);
}

// [k]P, k given as 8 little-endian 32-bit words, processing `nwords` words MSB first (double-and-add).
KZG_HD void g1_mul_words(G1Jac &r, const G1Affine &p, const uint32_t *k, int nwords) {
    G1Jac acc = g1_inf();
    for (int w = nwords - 1; w >= 0; w--) {
        uint32_t x = k[w];
        for (int b = 0; b < 32; b++) {
            g1_dbl(acc, acc);
            if (x >> 31) g1_add_mixed(acc, acc, p);
            x <<= 1;
        }
    }
    r = acc;
}

// [|x|]P for the BLS parameter |x| = 0xd201000000010000 (64 bits, Hamming weight 6); P Jacobian.
// ---- lazy (unreduced) Jacobian doubling / addition for the dependent chains (subgroup test, Horner, window ladders): no
// reduction below p after any product, sum or difference; every routine states the bounds it needs and restores
// (in units of p, limbs normalised).  Multiples of p are added before differences; products tolerate operands up to 64p.
//   g1_dbl_lazy : in  X, Y, Z < 32p                      out X < 26p, Y < 18p, Z < 4p     (infinity stays Z = 0 mod p)
//   g1_add_lazy : in  acc X, Y, Z < 32p, b canonical     out X < 8p,  Y < 4p,  Z < 2p
//                 acc possibly at infinity / b = +-acc are caught by exact low-limb filters and redone canonically.
KZG_G1_MID void g1_dbl_lazy(G1Jac &r, const G1Jac &p) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT, m16[NFP] = FP_MOD16_INIT, m32[NFP] = FP_MOD32_INIT;
    Fp A, B, C, D, E, F, t, u;
    fp_sqr_lz(A, p.x);
    fp_sqr_lz(B, p.y);
    fp_sqr_lz(C, B);
    fp_add_lz(t, p.x, B); fp_sqr_lz(t, t);                       // (X + B)^2                      < 2p
    fp_sub_lz(u, t, A, m2); fp_sub_lz(t, u, C, m2);              //  ... - A - C + 4p              in (0, 6p)
    fp_add_lz(D, t, t);                                          // D                              < 12p
    fp_add_lz(E, A, A); fp_add_lz(E, E, A);                      // E = 3A                         < 6p
    fp_sqr_lz(F, E);
    fp_mul_lz(u, p.y, p.z);                                      // before x, y are overwritten (r may alias p)
    Fp X3, Y3;
    fp_sub_lz(t, F, D, m16); fp_sub_lz(X3, t, D, m8);            // F - 2D + 24p                   in (0, 26p)
    fp_sub_lz(t, D, X3, m32);                                    // D - X3 + 32p                   in (6p, 44p)
    fp_mul_lz(Y3, E, t);
    fp_add_lz(C, C, C); fp_add_lz(C, C, C); fp_add_lz(C, C, C);  // 8C                             < 16p
    fp_sub_lz(r.y, Y3, C, m16);                                  //                                in (0, 18p)
    r.x = X3;
    fp_add_lz(r.z, u, u);                                        //                                < 4p
}
KZG_G1_MID void g1_add_lazy(G1Jac &r, const G1Jac &a, const G1Jac &b) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT;
    Fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, R;
    fp_sqr_lz(Z1Z1, a.z); fp_sqr_lz(Z2Z2, b.z);
    fp_mul_lz(U1, a.x, Z2Z2); fp_mul_lz(U2, b.x, Z1Z1);
    fp_mul_lz(S1, a.y, b.z); fp_mul_lz(S1, S1, Z2Z2);
    fp_mul_lz(S2, b.y, a.z); fp_mul_lz(S2, S2, Z1Z1);
    fp_sub_lz(H, U2, U1, m2);                                    // in (0, 4p)
    fp_sub_lz(R, S2, S1, m2);
    if (fp_maybe_zero_lz(H) || fp_maybe_zero_lz(a.z) || g1_is_inf(b)) {          // rare: the complete canonical addition
        G1Jac c; fp_canon64(c.x, a.x); fp_canon64(c.y, a.y); fp_canon64(c.z, a.z);
        g1_add(r, c, b);
        return;
    }
    Fp HH, HHH, V, t, u;
    fp_sqr_lz(HH, H); fp_mul_lz(HHH, H, HH); fp_mul_lz(V, U1, HH);
    Fp X3, Y3, Z3;
    fp_sqr_lz(X3, R);
    fp_sub_lz(t, X3, HHH, m2); fp_sub_lz(u, t, V, m2); fp_sub_lz(X3, u, V, m2);        // in (0, 8p)
    fp_sub_lz(t, V, X3, m8);                                                            // in (0, 10p)
    fp_mul_lz(Y3, R, t);
    fp_mul_lz(t, S1, HHH); fp_sub_lz(Y3, Y3, t, m2);                                    // in (0, 4p)
    fp_mul_lz(Z3, a.z, b.z); fp_mul_lz(Z3, Z3, H);
    r.x = X3; r.y = Y3; r.z = Z3;
}
// The same with BOTH operands lazy (b within the bounds g1_add_lazy leaves: X < 8p, Y < 4p, Z < 2p): b at infinity is then caught
// by the low-limb filter too, and the rare path canonicalises both operands.
KZG_G1_MID void g1_add_lazy2(G1Jac &r, const G1Jac &a, const G1Jac &b) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT;
    Fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, R;
    fp_sqr_lz(Z1Z1, a.z); fp_sqr_lz(Z2Z2, b.z);
    fp_mul_lz(U1, a.x, Z2Z2); fp_mul_lz(U2, b.x, Z1Z1);
    fp_mul_lz(S1, a.y, b.z); fp_mul_lz(S1, S1, Z2Z2);
    fp_mul_lz(S2, b.y, a.z); fp_mul_lz(S2, S2, Z1Z1);
    fp_sub_lz(H, U2, U1, m2);                                    // in (0, 4p)
    fp_sub_lz(R, S2, S1, m2);
    if (fp_maybe_zero_lz(H) || fp_maybe_zero_lz(a.z) || fp_maybe_zero_lz(b.z)) {      // rare: the complete canonical addition
        G1Jac c, d; fp_canon64(c.x, a.x); fp_canon64(c.y, a.y); fp_canon64(c.z, a.z); fp_canon64(d.x, b.x); fp_canon64(d.y, b.y); fp_canon64(d.z, b.z);
        g1_add(r, c, d);
        return;
    }
    Fp HH, HHH, V, t, u;
    fp_sqr_lz(HH, H); fp_mul_lz(HHH, H, HH); fp_mul_lz(V, U1, HH);
    Fp X3, Y3, Z3;
    fp_sqr_lz(X3, R);
    fp_sub_lz(t, X3, HHH, m2); fp_sub_lz(u, t, V, m2); fp_sub_lz(X3, u, V, m2);        // in (0, 8p)
    fp_sub_lz(t, V, X3, m8);                                                            // in (0, 10p)
    fp_mul_lz(Y3, R, t);
    fp_mul_lz(t, S1, HHH); fp_sub_lz(Y3, Y3, t, m2);                                    // in (0, 4p)
    fp_mul_lz(Z3, a.z, b.z); fp_mul_lz(Z3, Z3, H);
    r.x = X3; r.y = Y3; r.z = Z3;
}
KZG_HD void g1_canon_lazy(G1Jac &r, const G1Jac &a) { fp_canon64(r.x, a.x); fp_canon64(r.y, a.y); fp_canon64(r.z, a.z); }

// [|x|] P for the BLS parameter |x| = 0xd201000000010000 (weight 6), lazy chain, canonical result.  p canonical.
KZG_HD void g1_mul_x_abs(G1Jac &r, const G1Jac &p) {
    G1Jac acc = p;                       // top bit
    for (int i = 62; i >= 0; i--) {
        g1_dbl_lazy(acc, acc);
        if ((BLS_X_ABS >> i) & 1) g1_add_lazy(acc, acc, p);
    }
    g1_canon_lazy(r, acc);
}
KZG_HD bool g1_in_subgroup(const G1Affine &p) {
    if (g1a_is_inf(p)) return true;
    G1Jac pj, t;
    g1_from_affine(pj, p);
    g1_mul_x_abs(t, pj);
    g1_mul_x_abs(t, t);                  // [x^2]P
    if (g1_is_inf(t)) return false;
    const uint32_t bc[NFP] = FP_BETA_INIT;
    Fp beta; for (int i = 0; i < NFP; i++) beta.l[i] = bc[i];
    Fp z2, z3, lhs, rhs;
    fp_sqr(z2, t.z); fp_mul(z3, z2, t.z);
    fp_mul(lhs, p.x, beta); fp_mul(lhs, lhs, z2);      // beta x Z^2 == X
    if (!fp_eq(lhs, t.x)) return false;
    fp_mul(lhs, p.y, z3); fp_neg(rhs, t.y);            // y Z^3 == -Y
    return fp_eq(lhs, rhs);
}
// Reference form of the same predicate, kept for the unit tests: [r]P == infinity.
KZG_HD bool g1_in_subgroup_naive(const G1Affine &p) {
    const uint32_t rw[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    G1Jac t;
    g1_mul_words(t, p, rw, 8);
    return g1_is_inf(t);
}

// GLV split of a scalar against the endomorphism phi(x,y) = (beta x, y) = [-x^2]:  k = a + b * x^2  (a = k mod x^2,
// b = k div x^2, both < 2^128 for k < r), hence  [k]P = [a]P + [b](-phi(P)),  -phi(P) = (beta x, -y).
// Restoring division by the 128-bit constant x^2 = 0xac45a4010001a4020000000100000000.
KZG_HD void glv_split(uint32_t a[4], uint32_t b[4], const uint32_t k[8]) {
    const uint32_t X2[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;      // remainder, up to 129 bits
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    for (int w = 7; w >= 0; w--) {
        uint32_t x = k[w];
        for (int i = 0; i < 32; i++) {
            r4 = (r4 << 1) | (r3 >> 31); r3 = (r3 << 1) | (r2 >> 31); r2 = (r2 << 1) | (r1 >> 31); r1 = (r1 << 1) | (r0 >> 31);
            r0 = (r0 << 1) | (x >> 31); x <<= 1;
            // t = r - X2
            uint64_t d = (uint64_t)r0 - X2[0]; uint32_t t0 = (uint32_t)d; uint32_t br = (uint32_t)(d >> 63);
            d = (uint64_t)r1 - X2[1] - br; uint32_t t1 = (uint32_t)d; br = (uint32_t)(d >> 63);
            d = (uint64_t)r2 - X2[2] - br; uint32_t t2 = (uint32_t)d; br = (uint32_t)(d >> 63);
            d = (uint64_t)r3 - X2[3] - br; uint32_t t3 = (uint32_t)d; br = (uint32_t)(d >> 63);
            const bool ge = r4 != 0 || br == 0;
            r0 = ge ? t0 : r0; r1 = ge ? t1 : r1; r2 = ge ? t2 : r2; r3 = ge ? t3 : r3; r4 = ge ? 0u : r4;
            q3 = (q3 << 1) | (q2 >> 31); q2 = (q2 << 1) | (q1 >> 31); q1 = (q1 << 1) | (q0 >> 31); q0 = (q0 << 1) | (ge ? 1u : 0u);
        }
    }
    a[0] = r0; a[1] = r1; a[2] = r2; a[3] = r3;
    b[0] = q0; b[1] = q1; b[2] = q2; b[3] = q3;
}
// The same split by Barrett division (~150 instructions instead of the ~6400 of the bit-serial form above: the fixed-base MSM splits
// every one of its 4096 scalars): for k < 2^255, with MU = floor(2^256 / x^2) (129 bits), b^ = ((k >> 127) MU) >> 129 is b or b - 1
// (checked over random and edge values against the restoring division, tests/test_device_math_host.py), so one conditional
// correction -- two are made -- finishes it.  k >= 2^255 (never a canonical scalar) is NOT supported.
KZG_HD void glv_split_fast(uint32_t a[4], uint32_t b[4], const uint32_t k[8]) {
    const uint32_t MU[4] = {0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u};      // low 128 bits of MU; bit 128 is set
    const uint32_t X2[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    uint32_t k1[4], t[9];
#pragma unroll
    for (int i = 0; i < 4; i++) k1[i] = (k[3 + i] >> 31) | (k[4 + i] << 1);            // k >> 127
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { c += (uint64_t)k1[i] * MU[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
        t[i + 4] = (uint32_t)c;
    }
    {   // + k1 * 2^128 (the top bit of MU)
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { c += (uint64_t)t[4 + i] + k1[i]; t[4 + i] = (uint32_t)c; c >>= 32; }
        t[8] = (uint32_t)c;
    }
    uint32_t q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = (t[4 + i] >> 1) | (t[5 + i] << 31);             // >> 129
    // r = k - q x^2 on 160 bits (the true remainder is below 3 x^2 < 2^130)
    uint32_t p5[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { if (i + j < 5) { c += (uint64_t)q[i] * X2[j] + p5[i + j]; p5[i + j] = (uint32_t)c; c >>= 32; } }
        if (i + 4 < 5) p5[i + 4] = (uint32_t)c;
    }
    uint32_t r5[5];
    {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)k[i] - p5[i] - br; r5[i] = (uint32_t)d; br = (uint32_t)(d >> 63); }
    }
#pragma unroll
    for (int round = 0; round < 2; round++) {
        uint32_t s5[5], br = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)r5[i] - (i < 4 ? X2[i] : 0u) - br; s5[i] = (uint32_t)d; br = (uint32_t)(d >> 63); }
        const bool ge = br == 0;
        uint32_t c = ge ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 5; i++) r5[i] = ge ? s5[i] : r5[i];
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t v = q[i] + c; c = v < c ? 1u : 0u; q[i] = v; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = r5[i]; b[i] = q[i]; }
}
// -phi(P) = (beta x, -y)
KZG_HD void g1a_neg_phi(G1Affine &r, const G1Affine &p) {
    const uint32_t bc[NFP] = FP_BETA_INIT;
    Fp beta; for (int i = 0; i < NFP; i++) beta.l[i] = bc[i];
    fp_mul(r.x, p.x, beta);
    fp_neg(r.y, p.y);
    if (g1a_is_inf(p)) r = p;
}

// [k]P for a 128-bit k with signed 4-bit windows: digits in [-8, 8], table {1..8}P built once per lane (7 additions),
// then 32 x (4 doublings + 1 addition).  Plain double-and-add pays dbl + add on EVERY bit (in a 64-lane wave some lane
// always has its bit set): 128 x (dbl + madd) = 1.4M instructions per lane; this is ~0.9M.  The table lives in global
// memory ([entry][word][lane]: coalesced per wave) because 64 lanes x 8 Jacobian points do not fit anything smaller.
constexpr int W4_ENTRIES = 8;
KZG_HD void w4_store(uint32_t *tab, int e, int lane, const G1Jac &v) {
#pragma unroll
    for (int i = 0; i < NFP; i++) { tab[((e * 3 + 0) * NFP + i) * 64 + lane] = v.x.l[i]; tab[((e * 3 + 1) * NFP + i) * 64 + lane] = v.y.l[i];
            tab[((e * 3 + 2) * NFP + i) * 64 + lane] = v.z.l[i]; }
}
KZG_HD void w4_load(G1Jac &v, const uint32_t *tab, int e, int lane) {
#pragma unroll
    for (int i = 0; i < NFP; i++) { v.x.l[i] = tab[((e * 3 + 0) * NFP + i) * 64 + lane]; v.y.l[i] = tab[((e * 3 + 1) * NFP + i) * 64 + lane];
            v.z.l[i] = tab[((e * 3 + 2) * NFP + i) * 64 + lane]; }
}
KZG_HD void g1_mul128_w4(G1Jac &r, const G1Affine &p, const uint32_t k[4], uint32_t *tab, int lane) {
    G1Jac t; g1_from_affine(t, p);
    w4_store(tab, 0, lane, t);
    for (int e = 1; e < W4_ENTRIES; e++) { g1_add_mixed(t, t, p); w4_store(tab, e, lane, t); }     // (e+1) P
    // signed recoding: k + 0x888...8 (33 nibbles) then nibble - 8 -> digits d_0..d_32 in [-8, 7], d_32 in {0, 1} - handled as nibble 32 unbiased
    uint32_t e4[5];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { c += (uint64_t)k[i] + 0x88888888u; e4[i] = (uint32_t)c; c >>= 32; }
    e4[4] = (uint32_t)c;                                   // 0 or 1: the 33rd digit (unbiased)
    G1Jac acc = g1_inf();
    if (e4[4]) w4_load(acc, tab, 0, lane);                 // top digit 1 -> P
    // lazy chain (no reductions until the end); one doubling body and one addition body in the instruction stream
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int step = 127; step >= 0; step--) {
        g1_dbl_lazy(acc, acc);
        if ((step & 3) == 0) {
            const int nib = step >> 2;
            const int d = (int)((e4[nib >> 3] >> (4 * (nib & 7))) & 15u) - 8;
            const int mag = d < 0 ? -d : d;
            G1Jac q = g1_inf();
            if (mag) { w4_load(q, tab, mag - 1, lane); if (d < 0) fp_neg(q.y, q.y); }
            g1_add_lazy(acc, acc, q);
        }
    }
    g1_canon_lazy(r, acc);
}

// ZCash compressed encoding of an affine point ((0,0) = infinity)
KZG_HD void g1_compress_affine(uint8_t *out, const G1Affine &p) {
    if (g1a_is_inf(p)) {
        for (int i = 0; i < 48; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fp_to_be48(out, p.x);
    out[0] |= (uint8_t)(0x80 | (fp_is_lex_largest(p.y) ? 0x20 : 0));
}

// Flags and x of a compressed encoding (blst_p1_uncompress rules: compression bit required; infinity must be exactly
// 0xc0 00..00; x < p).  Returns 0 ok (x set, or inf), 1 bad encoding.
KZG_HD int g1_parse_compressed(Fp &x, bool &inf, bool &y_large, const uint8_t *in) {
    const uint8_t b0 = in[0];
    inf = false; y_large = (b0 & 0x20) != 0;
    if (!(b0 & 0x80)) return 1;
    if (b0 & 0x40) {
        uint32_t acc = b0 & 0x3f;
        for (int i = 1; i < 48; i++) acc |= in[i];
        if (acc) return 1;
        inf = true;
        return 0;
    }
    return fp_from_be48(x, in, true) ? 0 : 1;
}
// x^3 + 4
KZG_HD void g1_curve_rhs(Fp &r, const Fp &x) {
    const uint32_t b4[NFP] = FP_B_INIT;
    Fp four; for (int i = 0; i < NFP; i++) four.l[i] = b4[i];
    fp_sqr(r, x); fp_mul(r, r, x); fp_add(r, r, four);
}
// Decode 48 compressed bytes.  Returns 0 ok, 1 bad encoding, 2 not on curve (x^3 + 4 must be a square).
KZG_HD int g1_decompress(G1Affine &r, const uint8_t *in) {
    Fp x, y, y2;
    bool inf, want_large;
    if (g1_parse_compressed(x, inf, want_large, in)) return 1;
    if (inf) { r = g1a_inf(); return 0; }
    g1_curve_rhs(y2, x);
    if (!fp_sqrt(y, y2)) return 2;
    if (fp_is_lex_largest(y) != want_large) fp_neg(y, y);
    r.x = x; r.y = y;
    return 0;
}

}  // namespace kzg
