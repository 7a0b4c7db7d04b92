// pairing_lanes.h -- the hard part of the final exponentiation with TWELVE lanes per pairing check, five checks per wave (round 4).
//
// Why: the wave-cooperative kernel (pairing_coop.h) gives a whole 64-lane wave to one check, because a lone check is a latency problem: every
// limb product on its own lane, every lane reducing its own partial, a limb-parallel combination.  With thousands of batches per launch set the
// pairing is a THROUGHPUT problem (8.3 ms of a 120 ms step), and there that mapping is wasteful: a cyclotomic squaring keeps 36 of 64 lanes
// busy in phase 1 and does 36 Montgomery reductions for 12 output coefficients (~730 wave-instructions per squaring per check, x 316 squarings
// = half of the kernel).  Here lane k of a group of twelve owns output coefficient k: it accumulates the <= 3 products of its row of the
// Granger-Scott squaring unreduced, reduces ONCE, and the rows are the simplified ones below (30 products per squaring instead of 36) --
// ~1500 wave-instructions per squaring for FIVE checks.  Full products (40 per check) are not faster this way and are kept simple.
//
// Representation as in pairing_coop.h: Fp12 = Fp[w] / (w^12 - 2 w^6 + 2), coefficient k of w^k.  Invariant of every coefficient between
// operations: 0 <= value <= 2p, limbs normalised (< 2^29) below the top one.
//
// Cyclotomic squaring, from the rows of build_coop_schedules (pairing_coop.h) with (a, b, c, d) = (a_t, a_{t+6}, a_{t+3}, a_{t+9}), t = 0, 1, 2:
//     row A   3 P1 - 6 P2 - 12 P4       = 3 [ a^2 - 2 b^2 - 4 d (c + d) ]          out = 3 S - 2 a_k
//     row B   6 P2 + 3 P3 + 6 P4        = 3 [ 2 b (a + b) + c (c + 4 d) + 2 d^2 ]  out = 3 S - 2 a_k
//     row C   6 (P5 - P6 - P7 - P8)     = 6 [ a c - 2 b d ]                        out = 6 S + 2 a_k
//     row D   6 (P7 + P8)               = 6 [ (a + 2 b) d + b c ]                  out = 6 S + 2 a_k
//     row D'  -12 (P7 + P8)                                                        out = 2 a_k - 12 S
//     row C'  6 (P5 - P6 + P7 + P8)     = 6 [ a (c + 2 d) + 2 b (c + d) ]          out = 6 S + 2 a_k
// (a_k: the INPUT coefficient of the output index k; output k -> (t, row): 0 (0,A) 6 (0,B) 3 (0,C) 9 (0,D) 2 (1,A) 8 (1,B) 5 (1,C) 11 (1,D)
// 4 (2,A) 10 (2,B) 1 (2,D') 7 (2,C')).  Checked against the generic square and the cooperative squaring by tests/test_device_math_host.py.
#pragma once
#include "pairing_coop.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define L12_LANES(k) for (int k = (int)((threadIdx.x & 63) % 12), l12_once_ = ((threadIdx.x & 63) < 60); l12_once_; l12_once_ = 0)
#define L12_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#define L12_LOCALS 1          // a value a lane keeps across a synchronisation: one per lane on the device ...
#define L12_AT(k) 0
#else
#define L12_LANES(k) for (int k = 0; k < 12; k++)
#define L12_SYNC() ((void)0)
#define L12_LOCALS 12         // ... and one per emulated lane in the host build (the lanes run one after the other there)
#define L12_AT(k) (k)
#endif

namespace kzg {

constexpr int L12_BATCHES = 5;                         // checks per wave (lanes 60 .. 63 idle)
struct L12Mem { Fp12W s[4]; };                         // per check: slots F, T0, T1, T2 of the hard part's program

// r = ma * a + mb * b, limbs normalised (ma, mb <= 4; operands within the invariant)
KZG_HD void l12_lin(Fp &r, const Fp &a, uint32_t ma, const Fp &b, uint32_t mb) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) { const uint32_t t = ma * a.l[i] + mb * b.l[i] + c; if (i < NFP - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t; }
}
// v <= 32 p  ->  v < 2 p (v - 16p, - 8p, - 4p, - 2p where that stays non-negative); limbs normalised in and out
KZG_HD void l12_below_2p(Fp &v) {
    const uint32_t m16[NFP] = FP_MOD16_INIT, m8[NFP] = FP_MOD8_INIT, m4[NFP] = FP_MOD4_INIT, m2[NFP] = FP_MOD2_INIT;
    const uint32_t *ms[4] = {m16, m8, m4, m2};
    uint32_t x[NFP], s[NFP];
    {   // the top limb may carry excess: fold it down so that limb-wise subtraction compares values
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < NFP; i++) { const uint32_t t = v.l[i] + c; if (i < NFP - 1) { c = t >> LB; x[i] = t & LMASK; } else x[i] = t; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t br = ul_sub<NFP>(s, x, ms[k]);
#pragma unroll
        for (int i = 0; i < NFP; i++) x[i] = br ? x[i] : s[i];
    }
#pragma unroll
    for (int i = 0; i < NFP; i++) v.l[i] = x[i];
}
// the row of output coefficient k: t = index of the Fp4 element, row type, sign of the 2 a_k term
struct L12Row { uint8_t t, row; };
enum : uint8_t { L12_A = 0, L12_B, L12_C, L12_D, L12_DP, L12_CP };
KZG_HD L12Row l12_row(int k) {
    // k:            0      1       2      3      4      5      6      7       8      9      10     11
    const uint8_t T[12] = {0, 2, 1, 0, 2, 1, 0, 2, 1, 0, 2, 1};
    const uint8_t Rw[12] = {L12_A, L12_DP, L12_A, L12_C, L12_A, L12_C, L12_B, L12_CP, L12_B, L12_D, L12_B, L12_D};
    L12Row r; r.t = T[k]; r.row = Rw[k];
    return r;
}
// dst = a^2 for a in the cyclotomic subgroup.  dst may alias a (every lane has read its operands before any lane writes).
KZG_HD void l12_cyc_sqr(Fp12W &dst, const Fp12W &a) {
    Fp outv[L12_LOCALS];
    L12_LANES(k) {
        const uint32_t m4[NFP] = FP_MOD4_INIT, m8[NFP] = FP_MOD8_INIT;
        const L12Row rw = l12_row(k);
        const Fp A = a.c[rw.t], B = a.c[rw.t + 6], Cc = a.c[rw.t + 3], D = a.c[rw.t + 9], ak = a.c[k];
        const Fp z = fp_zero();
        Fp x0, y0, x1, y1, x2, y2;                                  // S = x0 y0 + x1 y1 + x2 y2
        bool three = true;
        switch (rw.row) {
            case L12_A:  x0 = A; y0 = A; l12_lin(x1, B, 2, z, 0); fp_sub_lz(x1, z, x1, m4); y1 = B;                 // a a + (4p - 2b) b + (8p - 4d)(c + d)
                         l12_lin(x2, D, 4, z, 0); fp_sub_lz(x2, z, x2, m8); l12_lin(y2, Cc, 1, D, 1); break;
            case L12_B:  l12_lin(x0, B, 2, z, 0); l12_lin(y0, A, 1, B, 1); x1 = Cc; l12_lin(y1, Cc, 1, D, 4);       // 2b (a + b) + c (c + 4d) + 2d d
                         l12_lin(x2, D, 2, z, 0); y2 = D; break;
            case L12_C:  x0 = A; y0 = Cc; l12_lin(x1, B, 2, z, 0); fp_sub_lz(x1, z, x1, m4); y1 = D; three = false; break;      // a c + (4p - 2b) d
            case L12_D: case L12_DP: l12_lin(x0, A, 1, B, 2); y0 = D; x1 = B; y1 = Cc; three = false; break;                    // (a + 2b) d + b c
            default:     x0 = A; l12_lin(y0, Cc, 1, D, 2); l12_lin(x1, B, 2, z, 0); l12_lin(y1, Cc, 1, D, 1); three = false; break;   // a (c + 2d) + 2b (c + d)
        }
        uint64_t acc[2 * NFP];
        wide_zero(acc);
        wide_mac(acc, x0.l, y0.l);
        wide_mac(acc, x1.l, y1.l);
        if (three) wide_mac(acc, x2.l, y2.l);
        Fp S; wide_reduce(S, acc);                                  // < p (1 + 2^-5), limbs normalised
        Fp s3, o, t2;
        l12_lin(s3, S, 3, z, 0);                                    // 3 S
        l12_lin(t2, ak, 2, z, 0);                                   // 2 a_k   <= 4p
        if (rw.row == L12_A || rw.row == L12_B) fp_sub_lz(o, s3, t2, m4);                              // 3 S - 2 a_k + 4p          <= 7.1 p
        // 2 a_k - 12 S + 16p  <= 20 p
        else if (rw.row == L12_DP) { const uint32_t m16[NFP] = FP_MOD16_INIT; Fp s12; l12_lin(s12, s3, 4, z, 0); fp_sub_lz(o, t2, s12, m16); }
        else { Fp s6; l12_lin(s6, s3, 2, z, 0); fp_add_lz(o, s6, t2); }                                // 6 S + 2 a_k               <= 10.2 p
        l12_below_2p(o);
        outv[L12_AT(k)] = o;
    }
    L12_SYNC();
    L12_LANES(k) { dst.c[k] = outv[L12_AT(k)]; }
    L12_SYNC();
}
// ---- full products without an exchange (round 6; the exchange form it replaced: 23 lockstep limb products, two reductions, one exchange row).
// c = a * b with the fold applied to the OPERAND: coefficient k of (w^j a) after w^12 = 2 w^6 - 2
// (so w^18 = 2 w^12 - 4 w^6 ... = 2 w^(s-12) - 4 w^(s-18) for s >= 18) is a combination of at most TWO coefficients of a with small weights,
//     k <= 5:   j <= k:  a_{k-j}                                   j > k:  -2 a_{k+12-j}   (and -4 a_{k+18-j} when j >= k + 7)
//     k >= 6:   j <= k:  a_{k-j}  (+ 2 a_{k+6-j} when j >= k - 5)  j > k:   2 a_{k+6-j}    (+ 2 a_{k+12-j} when k <= 10),
// so lane k forms  c_k = sum_j b_j A^(j)_k  with ONE accumulator and ONE Montgomery reduction: 12 limb products for a full product (6 for a line or an
// even-only operand) instead of 23 in lockstep + two reductions + an exchange row, and nothing to fold afterwards.
struct L12Fold { int ix, iy, wx, wy; };
KZG_HD L12Fold l12_fold(int k, int j) {
    L12Fold f; f.wy = 0;
    if (k <= 5) {
        if (j <= k) { f.ix = k - j; f.wx = 1; } else { f.ix = k + 12 - j; f.wx = -2; }
        f.iy = f.ix;
        if (j >= k + 7) { f.iy = k + 18 - j; f.wy = -4; }
    } else {
        if (j <= k) { f.ix = k - j; f.wx = 1; f.iy = f.ix; if (j >= k - 5) { f.iy = k + 6 - j; f.wy = 2; } }
        else { f.ix = k + 6 - j; f.wx = 2; f.iy = f.ix; if (k <= 10) { f.iy = k + 12 - j; f.wy = 2; } }
    }
    return f;
}
// A = wx X + wy Y (+ 16 p when a weight is negative: X, Y <= 2p, so the value stays in (0, 22 p]); limbs normalised below the top one
KZG_HD void l12_fold_operand(Fp &A, const Fp12W &a, int k, int j) {
    const uint32_t m16[NFP] = FP_MOD16_INIT;
    const L12Fold f = l12_fold(k, j);
    const Fp &X = a.c[f.ix], &Y = a.c[f.iy];
    const bool neg = f.wx < 0;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) {
        const int64_t t = (int64_t)(neg ? m16[i] : 0u) + (int64_t)f.wx * (int64_t)(int32_t)X.l[i] + (int64_t)f.wy * (int64_t)(int32_t)Y.l[i] + c;
        if (i < NFP - 1) { c = t >> LB; A.l[i] = (uint32_t)t & LMASK; } else A.l[i] = (uint32_t)t;
    }
}
// dst = a * b over the coefficients of b in jmask (FULL_MASK; EVEN_MASK; LINE_MASK for an evaluated line, whose other coefficients are never read).
// Coefficients of a, b within [0, 2p]; result below p (1 + 2^-13), limbs normalised.  dst may alias a or b.
KZG_HD void l12_mulf(Fp12W &dst, const Fp12W &a, const Fp12W &b, uint32_t jmask) {
    Fp outv[L12_LOCALS];
    L12_LANES(k) {
        uint64_t acc[2 * NFP];
        wide_zero(acc);
        int n = 0;
#pragma unroll 1
        for (int j = 0; j < 12; j++) {
            if (!((jmask >> j) & 1u)) continue;                     // (uniform)
            Fp A; l12_fold_operand(A, a, k, j);
            wide_mac(acc, A.l, b.c[j].l);
            if (++n == 3) { wide_carry(acc); n = 0; }               // three products of normalised limbs between sweeps keep the columns below 2^64
        }
        wide_carry(acc);
        Fp r; wide_reduce(r, acc);                                  // T <= 12 * 22p * 2p < 2^20 p^2
        outv[L12_AT(k)] = r;
    }
    L12_SYNC();
    L12_LANES(k) { dst.c[k] = outv[L12_AT(k)]; }
    L12_SYNC();
}
KZG_HD void l12_copy(Fp12W &dst, const Fp12W &a) {
    Fp v[L12_LOCALS];
    L12_LANES(k) { v[L12_AT(k)] = a.c[k]; }
    L12_SYNC();
    L12_LANES(k) { dst.c[k] = v[L12_AT(k)]; }
    L12_SYNC();
}
// conjugation (w -> -w): odd coefficients 2p - a_k
KZG_HD void l12_conj(Fp12W &dst, const Fp12W &a) {
    Fp v[L12_LOCALS];
    L12_LANES(k) {
        const uint32_t m2[NFP] = FP_MOD2_INIT;
        Fp t = a.c[k];
        if (k & 1) { const Fp z = fp_zero(); fp_sub_lz(t, z, t, m2); }
        v[L12_AT(k)] = t;
    }
    L12_SYNC();
    L12_LANES(k) { dst.c[k] = v[L12_AT(k)]; }
    L12_SYNC();
}
// Frobenius, power 1 (tables as coop_frob: k < 6: a_k A_k - 2 a_{k+6} B_k; k >= 6: a_k A_k + a_{k-6} B_k) and power 2 (a_k A2_k): two products
// (one) under one reduction.  dst may alias a.
KZG_HD void l12_frob(Fp12W &dst, const Fp12W &a, const Fp *tabA, const Fp *tabB) {
    Fp v[L12_LOCALS];
    L12_LANES(k) {
        const uint32_t m4[NFP] = FP_MOD4_INIT;
        const Fp z = fp_zero();
        Fp x1;
        if (tabB) { if (k < 6) { l12_lin(x1, a.c[k + 6], 2, z, 0); fp_sub_lz(x1, z, x1, m4); } else x1 = a.c[k - 6]; }
        uint64_t acc[2 * NFP];
        wide_zero(acc);
        wide_mac(acc, a.c[k].l, tabA[k].l);
        if (tabB) wide_mac(acc, x1.l, tabB[k].l);
        Fp r; wide_reduce(r, acc);
        v[L12_AT(k)] = r;
    }
    L12_SYNC();
    L12_LANES(k) { dst.c[k] = v[L12_AT(k)]; }
    L12_SYNC();
}
// per lane: is coefficient k that of the element 1?  (the caller combines the twelve answers)
KZG_HD bool l12_coeff_is_one(const Fp12W &a, int k) {
    const Fp want = k == 0 ? fp_one() : fp_zero();
    Fp c; fp_norm_lz(c, a.c[k]); fp_canon64(c, c);
    return fp_eq(c, want);
}

// The hard part's program is the tail of build_pairing_program (pairing_coop.h) from `hard_start` on; it only uses the slots F, T0, T1, T2 and
// the operations below.  m.s[slot] as coop_slot.
KZG_HD void l12_run(L12Mem &m, const CoopInsn *prog, int pc0, int pc1, const FrobTables &ft) {
    for (int pc = pc0; pc < pc1; pc++) {
        const CoopInsn in = prog[pc];
        Fp12W &dst = m.s[in.dst];
        const Fp12W &a = m.s[in.a];
        switch (in.op) {
            case OP_CYC_SQR: l12_cyc_sqr(dst, a); break;
            case OP_MUL: l12_mulf(dst, a, m.s[in.b], FULL_MASK); break;
            case OP_CONJ: l12_conj(dst, a); break;
            case OP_FROB1: case OP_FROB2: l12_frob(dst, a, in.op == OP_FROB1 ? ft.a1 : ft.a2, in.op == OP_FROB1 ? ft.b1 : nullptr); break;
            default: l12_copy(dst, a); break;
        }
    }
}

}  // namespace kzg
