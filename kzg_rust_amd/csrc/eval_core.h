// eval_core.h -- the arithmetic core of k_eval (k_verify.hip): evaluation of a blob polynomial given by its 4096 values at
// the bit-reversed roots of unity (reference src/kzg.rs:346-389), four domain points at a time.  Host + device: the same
// source is driven lane by lane on the device and serially by tests/native/hd_probe.cpp.
//
//   y = (1/N) sum_i p_i w_i prod_{j != i} (z - w_j)          (no inversion: prod_j (z - w_j) = z^N - 1)
//
// In bit-reversal order positions 4k..4k+3 hold  w, -w, iw, -iw  (w = w_{4k}, i = w^(N/4)): the four roots of x^4 = rho,
// rho = w^4.  For such a group
//     D = prod_j (z - w_j) = z^4 - rho
//     N = sum_j p_j w_j prod_{l != j} (z - w_l) = sum_j p_j w_j (z^3 + w_j z^2 + w_j^2 z + w_j^3)
//       = rho ( F_0 + t F_3 + t^2 F_2 + t^3 F_1 ),     t = z / w,   F_m = sum_j p_j c_j^m,  c = (1, -1, i, -i)
// (a 4-point DFT of the values: one product by i).  With h = F_0 + t F_3 + t^2 F_2 + t^3 F_1 the numerator is N = rho h, and
// because rho = z^4 - D the factor rho never has to be multiplied in per group:
//     sum_k rho_k h_k prod_{l != k} D_l  =  z^4 * sum_k h_k prod_{l != k} D_l  -  (sum_k h_k) * prod_l D_l
// so a running triple  P = prod D,  S = sum_k h_k prod_{l != k} D_l,  H = sum_k h_k  folds a group in as
//     S <- S D + h P,   P <- P D,   H <- H + h        (eval_fold_group4)
// and  z^4 S - H P  is formed once per lane at the end (eval_fold_finish).  7.5 product-equivalents per 4 values (i D', t, three
// Horner steps, the two-product update of S with one reduction, P D; round 2 multiplied rho in per group: 8.5) against 14 for the
// value-at-a-time form and ~3 x 4 + a 4096-long batch inversion in the reference.  z inside the domain needs no special case: all
// of this is polynomial identity.
// Domains: z, t, D, rho, P are Montgomery residues; the values enter as plain integers, so F, N, S and y are plain.
// All intermediate products are lazy (mont_mul_lazy: not reduced below r); bounds are noted where they matter.
#pragma once
#include "field.h"

namespace kzg {

struct EvalGroupTab { Fr inv_root, rho; };      // per group k: w_{4k}^-1 and w_{4k}^4, Montgomery

// r = a - b + 2r' with r' the modulus: a, b < 2^256-ish lazy values with b < 2r'; limbs carry-normalised, result > 0.
KZG_HD void fr_sub_lazy(Fr &r, const Fr &a, const Fr &b) {
    KZG_FR_CONSTS
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFR; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)(2u * FR_MOD[i]) + c;     // 2 * limb < 2^30
        if (i < NFR - 1) { c = t >> LB; r.l[i] = (uint32_t)t & LMASK; }
        else r.l[i] = (uint32_t)t;
    }
}

// Fold the four values at positions 4k..4k+3 (plain 256-bit integers, 8 little-endian words each) into (P, S, H).
// z4 = z^4; imag = w^(N/4).  first: (P, S, H) are set instead of updated.  H is a plain lazy sum (< 5.1 r per group): at most
// EVAL_MAX_GROUPS_PER_FOLD groups per triple keep it inside the 9-limb representation (the top limb holds the excess).
constexpr int EVAL_MAX_GROUPS_PER_FOLD = 64;                      // 64 x 5.1 r < 2^264
KZG_HD void eval_fold_group4(Fr &P, Fr &S, Fr &H, bool first, const uint32_t pw[4][8], const Fr &z, const Fr &z4, const EvalGroupTab &g, const Fr &imag) {
    Fr p0, p1, p2, p3;
    words_to_limbs<NFR, 8>(p0.l, pw[0]); words_to_limbs<NFR, 8>(p1.l, pw[1]);
    words_to_limbs<NFR, 8>(p2.l, pw[2]); words_to_limbs<NFR, 8>(p3.l, pw[3]);
    Fr A, B, C, Dd, iD, F0, F1, F2, F3;
    fr_add_lazy(A, p0, p1); fr_add_lazy(B, p2, p3);              // < 2r each (values are < r when the blob is valid)
    fr_sub_lazy(C, p0, p1); fr_sub_lazy(Dd, p2, p3);             // p - p' + 2r in (r, 3r)
    fr_mul_lazy(iD, Dd, imag);                                   // < 1.1 r
    fr_add_lazy(F0, A, B);                                       // < 4r
    fr_sub_lazy(F2, A, B);                                       // < 4r
    fr_add_lazy(F1, C, iD);                                      // < 4.1r
    fr_sub_lazy(F3, C, iD);                                      // < 5r
    Fr t, h, D;
    fr_mul_lazy(t, z, g.inv_root);                               // z / w
    fr_mul_lazy(h, F1, t); fr_add_lazy(h, h, F2);                // < 5.1r
    fr_mul_lazy(h, h, t); fr_add_lazy(h, h, F3);                 // < 6.1r
    fr_mul_lazy(h, h, t); fr_add_lazy(h, h, F0);                 // < 5.1r
    fr_sub(D, z4, g.rho);                                        // canonical
    if (first) { P = D; S = h; H = h; }
    else { fr_mul2_lazy(S, S, D, h, P); fr_mul_lazy(P, P, D); fr_add_lazy(H, H, h); }      // S D + h P < 6.7 r^2: S < 1.1 r again
}
// The triple of a lane -> its pair (P, S) with S = sum_k N_k prod_{l != k} D_l:  S = z^4 S'' - H P  (+ a multiple of r to stay positive).
// H < 64 x 5.1 r and P < 1.1 r give H P / R < 5.2 r; the result is lazy (< 7 r) and goes into a canonical product next.
KZG_HD void eval_fold_finish(Fr &S, const Fr &P, const Fr &H, const Fr &z4) {
    KZG_FR_CONSTS
    Fr A, B;
    fr_mul_lazy(A, S, z4);                                       // < 1.1 r
    fr_mul_lazy(B, H, P);                                        // < 6.2 r
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFR; i++) {                              // A - B + 7 r
        const int64_t t = (int64_t)A.l[i] - (int64_t)B.l[i] + 7 * (int64_t)FR_MOD[i] + c;
        if (i < NFR - 1) { c = (int32_t)(t >> LB); S.l[i] = (uint32_t)t & LMASK; }
        else S.l[i] = (uint32_t)t;
    }
}

}  // namespace kzg
