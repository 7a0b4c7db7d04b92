// eval_core.h -- the arithmetic core of k_eval (k_verify.hip): evaluation of a blob polynomial given by its 4096 values at
// the bit-reversed roots of unity (reference src/kzg.rs:346-389) as a radix-4 tree over the domain.  Host + device: the same
// source is driven lane by lane on the device and serially by tests/native/hd_probe.cpp.
//
//   y = (1/N) sum_i p_i w_i prod_{j != i} (z - w_j)          (no inversion: prod_j (z - w_j) = z^N - 1)
//
// In bit-reversal order positions 4k..4k+3 hold  w, -w, iw, -iw  (w = w_{4k}, i = w^(N/4)): the four roots of x^4 = rho,
// rho = w^4.  For such a group
//     D = prod_j (z - w_j) = z^4 - rho
//     N = sum_j p_j w_j prod_{l != j} (z - w_l) = sum_j p_j w_j (z^3 + w_j z^2 + w_j^2 z + w_j^3)
//       = rho ( F_0 + t F_3 + t^2 F_2 + t^3 F_1 ),     t = z / w,   F_m = sum_j p_j c_j^m,  c = (1, -1, i, -i)
// (a 4-point DFT of the values: one product by i).  Write N = rho h.  Round 3, second form: the same step repeats one level up.
// Four neighbouring groups 4m..4m+3 have rho_k = rho' c_k (again the four roots of X^4 = rho'^4, now in X = z^4), and the sum of
// their fractions N_k / D_k over the common denominator prod_k (Z - rho_k) = Z^4 - rho'^4, Z = z^4, has the numerator
//     sum_k rho_k h_k prod_{l != k} (Z - rho_l) = sum_k h_k (rho_k Z^3 + rho_k^2 Z^2 + rho_k^3 Z + rho_k^4)
//       = rho'^4 ( G_0 + T G_3 + T^2 G_2 + T^3 G_1 ),   T = Z / rho',   G_m = sum_k h_k c_k^m
// -- the SAME node function applied to the four h of the level below, with z^4 for z.  Six levels take 4096 values to one h, the
// last "rho" is w_0^4096 = 1 and the last denominator is z^4096 - 1 = prod_j (z - w_j): y = h_top / N.  No denominator is ever
// formed.  A node costs five products (i D', T, three Horner steps) for four children: 1365 nodes x 5 = 6825 products per blob,
// against 7.5 per group + the merge of the lanes' pairs (~7900) for the running (P, S, H) fold this replaces, 14 per group for
// the value-at-a-time form, and ~3 x 4096 + a 4096-long batch inversion in the reference.  z inside the domain needs no special
// case: all of this is polynomial identity (for z = w_m the reference's branch kzg.rs:360-362 returns p_m, and so does this).
//
// Domains: T and i are Montgomery residues; the values enter as plain integers, so every h and y are plain.
// All products are lazy (mont_mul_lazy: result < r (1 + a b / (70.7 r^2)), not reduced below r) and the sums are plain limb
// additions, so a node's output is about four times its inputs' bound: in units of r, with inputs < c,
//     A, B < 2c;  C, D' = a - b + KC r < c + KC;  i D' < 1 + (c + KC)/70.7;  F_0 < 4c;  F_2 = A - B + KF2 r < 2c + KF2;
//     F_1 < c + KC + 2.3;  F_3 = C - i D' + 3 r < c + KC + 3;   h = ((F_1 T + F_2) T + F_3) T + F_0 < 4c + 1.1 + F-terms / 70
// KC >= c and KF2 >= 2c keep the differences positive.  Levels 1-3 (in a lane / across lanes through LDS): c = 1 -> 5.2 -> 22.1 ->
// 90.1; one lazy product by R (Montgomery one) brings that back to < 2.3, then levels 4-6: 2.3 -> 10.4 -> 42.9 -> 174, and the
// product by 1/N and one canonical product finish.  The top limb holds the excess: 175 r < 2^262.4, top limb < 2^30.4; column
// sums of the products stay below 2^64 (9 x 2^59.4 + 9 x 2^58).
#pragma once
#include "field.h"

namespace kzg {

// inverse "roots" of the tree's nodes, one flat table: level l (1..5) node j holds  w_(4^l j) ^ -(4^(l-1))  (Montgomery), w_idx the
// domain in bit-reversal order; level 6 is the single node with rho' = 1.
constexpr int EVAL_ZPOWERS = 5;        // z^4, z^16, z^64, z^256, z^1024: the z of levels 2..6, squared up once per blob by the challenge kernel
constexpr int EVAL_WAVES = 4;          // blobs (waves) per workgroup of k_eval: the last three levels of the four trees run together on wave 0
constexpr int EVAL_TAB_L1 = 0, EVAL_TAB_L2 = 1024, EVAL_TAB_L3 = 1280, EVAL_TAB_L4 = 1344, EVAL_TAB_L5 = 1360, EVAL_TAB_ENTRIES = 1364;

// K r as normalised limbs (the top limb keeps the excess), at compile time
template <int K> struct FrMultiple {
    uint32_t l[NFR];
    constexpr FrMultiple() : l{} {
        const uint32_t m[NFR] = FR_MOD_INIT;
        uint64_t c = 0;
        for (int i = 0; i < NFR; i++) { c += (uint64_t)m[i] * (uint32_t)K; l[i] = i < NFR - 1 ? (uint32_t)(c & LMASK) : (uint32_t)c; c >>= LB; }
    }
};
// r = a - b + K r' (r' the modulus): lazy values with b <= K r'; limbs carry-normalised, the top limb keeps the excess.
template <int K> KZG_HD void fr_sub_bias(Fr &r, const Fr &a, const Fr &b) {
    constexpr FrMultiple<K> kr;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFR; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)kr.l[i] + c;       // limbs < 2^29 (top: < 2^30.4 each side)
        if (i < NFR - 1) { c = t >> LB; r.l[i] = (uint32_t)t & LMASK; }
        else r.l[i] = (uint32_t)t;
    }
}

// Carry-free forms for values that only feed a product or a later normalising sum: the 64-bit columns of mont_mul_lazy take one operand
// with limbs up to 2^31 against a normalised one (9 x 2^60 + 9 x 2^58 + carries < 2^64).
KZG_HD void fr_add_raw(Fr &r, const Fr &a, const Fr &b) {                    // limbs add up, no carries
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = a.l[i] + b.l[i];
}
// K r with 2^29 lent to every limb below the top one (and taken back from the limb above): a - b + that is limb-wise positive for normalised b
template <int K> struct FrMultipleRaw {
    uint32_t l[NFR];
    constexpr FrMultipleRaw() : l{} {
        const FrMultiple<K> n;
        for (int i = 0; i < NFR; i++) l[i] = n.l[i] + (i < NFR - 1 ? (1u << LB) : 0u) - (i > 0 ? 1u : 0u);
    }
};
// r = a - b + K r', limb by limb without carries: a, b normalised, K r' > b; limbs of the result in [1, 3 * 2^29), top limb >= 0 because
// (K r' - b) / 2^232 >= 1 whenever K exceeds b's bound by 2^-22
template <int K> KZG_HD void fr_sub_bias_raw(Fr &r, const Fr &a, const Fr &b) {
    constexpr FrMultipleRaw<K> kr;
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = a.l[i] - b.l[i] + kr.l[i];
}
// r = a + b with the carries swept (b may be a raw sum with limbs up to 2^31)
KZG_HD void fr_add_sweep(Fr &r, const Fr &a, const Fr &b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFR; i++) {
        const uint32_t t = a.l[i] + b.l[i] + c;                               // < 2^29 + 2^31 + 2^3
        if (i < NFR - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t;
    }
}

// One node of the tree: four children (plain lazy values < KC r with normalised limbs, in bit-reversal order: roots w, -w, iw, -iw of
// the node's X^4 = w^4), T = (z^(4^(l-1))) / w and imag = w^(N/4), both Montgomery and normalised.  h = F_0 + T F_3 + T^2 F_2 + T^3 F_1,
// normalised.  Sums that only feed a product (A, B, D', F_1, the Horner partial sums) or the final sweep (F_0) are carry-free.
template <int KC, int KF2> KZG_HD void eval_node4(Fr &h, const Fr &c0, const Fr &c1, const Fr &c2, const Fr &c3, const Fr &T, const Fr &imag) {
    Fr A, B, C, Dd, iD, F0, F1, F2, F3;
    fr_add_raw(A, c0, c1); fr_add_raw(B, c2, c3);                             // limbs < 2^30
    fr_sub_bias<KC>(C, c0, c1);                                               // normalised: it enters F_3's normalising difference
    fr_sub_bias_raw<KC>(Dd, c2, c3);                                          // limbs < 3 * 2^29
    fr_mul_lazy(iD, Dd, imag);
    fr_add_raw(F0, A, B);                                                     // limbs < 2^31
    fr_sub_bias<KF2>(F2, A, B);                                               // normalised (int32 range: |A_i - B_i| < 2^30)
    fr_add_raw(F1, C, iD);                                                    // limbs < 2^30
    fr_sub_bias<3>(F3, C, iD);
    fr_mul_lazy(h, F1, T); fr_add_raw(h, h, F2);
    fr_mul_lazy(h, h, T); fr_add_raw(h, h, F3);
    fr_mul_lazy(h, h, T); fr_add_sweep(h, h, F0);
}
// The six levels with the bias constants their input bounds call for (header comment).  Level 1 takes the blob's values: < r when
// the blob is valid, anything below 2^256 = 2.21 r otherwise (flagged by the caller; KC = 3 keeps even those differences positive).
KZG_HD void eval_level1(Fr &h, const uint32_t pw[4][8], const Fr &T, const Fr &imag) {
    Fr p0, p1, p2, p3;
    words_to_limbs<NFR, 8>(p0.l, pw[0]); words_to_limbs<NFR, 8>(p1.l, pw[1]);
    words_to_limbs<NFR, 8>(p2.l, pw[2]); words_to_limbs<NFR, 8>(p3.l, pw[3]);
    eval_node4<3, 5>(h, p0, p1, p2, p3, T, imag);                              // < 5.2 r
}
KZG_HD void eval_level2(Fr &h, const Fr c[4], const Fr &T, const Fr &imag) { eval_node4<6, 11>(h, c[0], c[1], c[2], c[3], T, imag); }     // < 22.1 r
// level 3 ends with the lazy product by R that brings the bound back down: < 90.1 r -> < 2.3 r
KZG_HD void eval_level3(Fr &h, const Fr c[4], const Fr &T, const Fr &imag) {
    Fr t; eval_node4<23, 45>(t, c[0], c[1], c[2], c[3], T, imag);
    fr_mul_lazy(h, t, fr_one());
}
KZG_HD void eval_level4(Fr &h, const Fr c[4], const Fr &T, const Fr &imag) { eval_node4<3, 5>(h, c[0], c[1], c[2], c[3], T, imag); }       // < 10.4 r
KZG_HD void eval_level5(Fr &h, const Fr c[4], const Fr &T, const Fr &imag) { eval_node4<11, 21>(h, c[0], c[1], c[2], c[3], T, imag); }     // < 42.9 r
// level 6: T = z^1024 (rho' = 1); y = h / 4096 as the canonical plain integer
KZG_HD void eval_level6(Fr &y, const Fr c[4], const Fr &z1024, const Fr &imag) {
    const uint32_t inv4096[NFR] = FR_INV4096_INIT;
    Fr k4096; for (int i = 0; i < NFR; i++) k4096.l[i] = inv4096[i];
    Fr h, t;
    eval_node4<44, 87>(h, c[0], c[1], c[2], c[3], z1024, imag);                // < 174 r
    fr_mul_lazy(t, h, k4096);                                                  // < 3.5 r, plain
    fr_mul(y, t, fr_one());                                                    // canonical: the chain of lazy products ends here
}

}  // namespace kzg
