// pairing_coop.h -- wave-cooperative pairing check: one 64-lane wavefront per pairing product.
//
// Why: on one lane the pairing check of verify_kzg_proof_batch (reference src/utils.rs:189-214) is a single chain of
// ~20k dependent Fp products; a lone gfx950 wave issues one instruction every ~5 cycles, so that chain costs ~75 ms
// and bounds the latency of a whole 64-blob batch (profiles/r01).  Here the 12 Fp coefficients of an Fp12 element are
// spread over the lanes of a wave and every Fp12 operation becomes two short phases:
//
//   representation   Fp12 = Fp[w]/(w^12 - 2 w^6 + 2)  (w^2 = v, w^6 = 1 + u): a = sum_{k<12} a_k w^k, a_k in Fp.
//                    Tower coefficient (x0 + x1 u) v^j w^i sits at m = 2j + i:  a_m += x0 - x1,  a_{m+6} += x1.
//   product          phase 1: lane s (0..22) computes the convolution term d_s = sum_{i+j=s} a_i b_j.  The <= 12 limb
//                    products are accumulated UNREDUCED in 64-bit columns (29-bit limbs leave room for 4 products
//                    between carry sweeps) and Montgomery-reduced once.
//                    phase 2: lane k (0..11) folds w^12 = 2 w^6 - 2:
//                        k<=4: d_k - 2 d_{k+12} - 4 d_{k+18}   k=5: d_5 - 2 d_17
//                        6<=k<=10: d_k + 2 d_{k+6} + 2 d_{k+12}  k=11: d_11 + 2 d_17
//   square           same with d_s = 2 sum_{i<j} a_i a_j + a_{s/2}^2          (<= 6 limb products per lane)
//   line product     a Miller-loop line  c0 + c1' v + c4' v w  has only w^{0,2,3,6,8,9} -> <= 6 limb products per lane
//   Frobenius        a_k -> a_k gamma^k folded back, two constant products per lane; conjugation flips odd k.
//
// Operands live in a CoopMem block (LDS on the device); lanes only communicate through it, with a barrier between
// phases.  The same source runs on the host for unit tests: COOP_LANES loops over the 64 lanes sequentially there.
#pragma once
#include "pairing.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define COOP_LANES(lane) for (int lane = (int)threadIdx.x, coop_once_ = 1; coop_once_; coop_once_ = 0)
#define COOP_SYNC() __syncthreads()
#else
#define COOP_LANES(lane) for (int lane = 0; lane < 64; lane++)
#define COOP_SYNC() ((void)0)
#endif

namespace kzg {

struct Fp12W { Fp c[12]; };                    // coefficients of w^0 .. w^11
struct LineW { Fp l0, l6, l2, l8, l3, l9; };   // a precomputed line in the w basis; l2,l8 await *xP, l3,l9 await *yP

constexpr uint32_t LINE_MASK = (1u << 0) | (1u << 2) | (1u << 3) | (1u << 6) | (1u << 8) | (1u << 9);
constexpr uint32_t EVEN_MASK = 0x555u;
constexpr uint32_t FULL_MASK = 0xfffu;

struct CoopMem {
    Fp12W f, t0, t1, t2, t3, t4;
    Fp12W line[2];
    Fp d[23];
    Fp px[2], py[2];
    int flag;
};

// Tower line triple (pairing.h LineCoeff) -> w basis
KZG_HD void line_to_w(LineW &o, const LineCoeff &l) {
    fp_sub(o.l0, l.c0.c0, l.c0.c1); o.l6 = l.c0.c1;
    fp_sub(o.l2, l.c1.c0, l.c1.c1); o.l8 = l.c1.c1;
    fp_sub(o.l3, l.c4.c0, l.c4.c1); o.l9 = l.c4.c1;
}

// ---------------------------------------------------------------------------------- wide (unreduced) arithmetic
KZG_HD void wide_zero(uint64_t *acc) {
#pragma unroll
    for (int i = 0; i < 2 * NFP; i++) acc[i] = 0;
}
// acc[i+j] += a[i] * b[j]   (196 v_mad_u64_u32)
KZG_HD void wide_mac(uint64_t *acc, const uint32_t *a, const uint32_t *b) {
#pragma unroll
    for (int i = 0; i < NFP; i++) {
#pragma unroll
        for (int j = 0; j < NFP; j++) acc[i + j] += (uint64_t)a[i] * b[j];
    }
}
KZG_HD void wide_carry(uint64_t *acc) {
#pragma unroll
    for (int i = 0; i < 2 * NFP - 1; i++) { acc[i + 1] += acc[i] >> LB; acc[i] &= LMASK; }
}
KZG_HD void wide_double(uint64_t *acc) {
#pragma unroll
    for (int i = 0; i < 2 * NFP; i++) acc[i] <<= 1;
}
// Montgomery reduction of a carried accumulator holding T < 16 p^2:  r = T / 2^406 mod p, canonical.
KZG_HD void wide_reduce(Fp &r, uint64_t *acc) {
    const uint32_t m[NFP] = FP_MOD_INIT;
#pragma unroll
    for (int i = 0; i < NFP; i++) {
        const uint32_t q = ((uint32_t)acc[i] * FP_INVW) & LMASK;
#pragma unroll
        for (int j = 0; j < NFP; j++) acc[i + j] += (uint64_t)q * m[j];
        acc[i + 1] += acc[i] >> LB;
    }
    uint32_t t[NFP], s[NFP];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NFP; j++) { c += acc[NFP + j]; t[j] = (uint32_t)c & LMASK; c >>= LB; }
    const uint32_t br = ul_sub<NFP>(s, t, m);
#pragma unroll
    for (int j = 0; j < NFP; j++) r.l[j] = br ? t[j] : s[j];
}

// ---------------------------------------------------------------------------------- cooperative Fp12 operations
// phase 2 of every product: fold the 23 convolution terms with w^12 = 2 w^6 - 2
KZG_HD void coop_fold(Fp12W &dst, const Fp *d, int k) {
    Fp r, t;
    if (k <= 4) {
        fp_dbl(t, d[k + 18]); fp_add(t, t, d[k + 12]); fp_dbl(t, t);      // 4 d_{k+18} + 2 d_{k+12}
        fp_sub(r, d[k], t);
    } else if (k == 5) {
        fp_dbl(t, d[17]); fp_sub(r, d[5], t);
    } else if (k <= 10) {
        fp_add(t, d[k + 6], d[k + 12]); fp_dbl(t, t); fp_add(r, d[k], t);
    } else {
        fp_dbl(t, d[17]); fp_add(r, d[11], t);
    }
    dst.c[k] = r;
}

// dst = a * b, b having non-zero coefficients only where bmask has a bit set.  dst may alias a or b.
KZG_HD void coop_mul(CoopMem &m, Fp12W &dst, const Fp12W &a, const Fp12W &b, uint32_t bmask) {
    COOP_LANES(lane) {
        if (lane < 23) {
            uint64_t acc[2 * NFP];
            wide_zero(acc);
            int pending = 0;
            for (int j = 0; j < 12; j++) {
                const int i = lane - j;
                if (!((bmask >> j) & 1u) || i < 0 || i > 11) continue;
                wide_mac(acc, a.c[i].l, b.c[j].l);
                if (++pending == 4) { wide_carry(acc); pending = 0; }
            }
            wide_carry(acc);
            wide_reduce(m.d[lane], acc);
        }
    }
    COOP_SYNC();
    COOP_LANES(lane) { if (lane < 12) coop_fold(dst, m.d, lane); }
    COOP_SYNC();
}

// dst = a^2.  dst may alias a.
KZG_HD void coop_sqr(CoopMem &m, Fp12W &dst, const Fp12W &a) {
    COOP_LANES(lane) {
        if (lane < 23) {
            uint64_t acc[2 * NFP];
            wide_zero(acc);
            int pending = 0;
            const int lo = lane > 11 ? lane - 11 : 0;
            for (int i = lo; 2 * i < lane; i++) {            // pairs i < j = lane - i
                wide_mac(acc, a.c[i].l, a.c[lane - i].l);
                if (++pending == 4) { wide_carry(acc); pending = 0; }
            }
            wide_carry(acc);
            wide_double(acc);
            if (!(lane & 1)) wide_mac(acc, a.c[lane >> 1].l, a.c[lane >> 1].l);
            wide_carry(acc);
            wide_reduce(m.d[lane], acc);
        }
    }
    COOP_SYNC();
    COOP_LANES(lane) { if (lane < 12) coop_fold(dst, m.d, lane); }
    COOP_SYNC();
}

KZG_HD void coop_copy(Fp12W &dst, const Fp12W &a) {
    COOP_LANES(lane) { if (lane < 12) dst.c[lane] = a.c[lane]; }
    COOP_SYNC();
}
KZG_HD void coop_set_one(Fp12W &dst) {
    COOP_LANES(lane) { if (lane < 12) dst.c[lane] = lane == 0 ? fp_one() : fp_zero(); }
    COOP_SYNC();
}
// conjugation = p^6-power Frobenius: w -> -w
KZG_HD void coop_conj(Fp12W &dst, const Fp12W &a) {
    COOP_LANES(lane) {
        if (lane < 12) { Fp t = a.c[lane]; if (lane & 1) fp_neg(t, t); dst.c[lane] = t; }
    }
    COOP_SYNC();
}
// Frobenius (power 1 or 2).  (sum a_k w^k)^(p^e) = sum a_k g^k w^k with g = xi^((p^e-1)/6) = g0 + g1 u; folding
// u = w^6 - 1 and w^12 = 2w^6 - 2:   k<6:  c_k = a_k (g0-g1)_k - 2 a_{k+6} (g1)_{k+6}
//                                     k>=6: c_k = a_k (g0+g1)_k +   a_{k-6} (g1)_{k-6}
// tabA[k] = (g0-g1)_k for k<6, (g0+g1)_k for k>=6;  tabB[k] = (g1)_{k+6} for k<6 (to be doubled and subtracted),
// (g1)_{k-6} for k>=6.  dst must not alias a.
KZG_HD void coop_frob(Fp12W &dst, const Fp12W &a, const Fp *tabA, const Fp *tabB) {
    COOP_LANES(lane) {
        if (lane < 12) {
            Fp x, y;
            fp_mul(x, a.c[lane], tabA[lane]);
            if (lane < 6) { fp_mul(y, a.c[lane + 6], tabB[lane]); fp_dbl(y, y); fp_sub(x, x, y); }
            else { fp_mul(y, a.c[lane - 6], tabB[lane]); fp_add(x, x, y); }
            dst.c[lane] = x;
        }
    }
    COOP_SYNC();
}
KZG_HD bool coop_is_one(CoopMem &m, const Fp12W &a) {
    COOP_LANES(lane) { if (lane == 0) m.flag = 1; }
    COOP_SYNC();
    COOP_LANES(lane) {
        if (lane < 12) {
            const Fp want = lane == 0 ? fp_one() : fp_zero();
            if (!fp_eq(a.c[lane], want)) m.flag = 0;
        }
    }
    COOP_SYNC();
    return m.flag != 0;
}
// dst = a^-1 = conj(a) * N^-1 with N = a * conj(a) in Fp6 (even powers of w only).  The Fp6 inversion is a short
// single-lane tower computation.  Uses t3, t4 of the block.  dst may alias a.
KZG_HD void coop_inv(CoopMem &m, Fp12W &dst, const Fp12W &a) {
    coop_conj(m.t3, a);
    coop_mul(m, m.t4, a, m.t3, FULL_MASK);            // N: odd coefficients are zero
    COOP_LANES(lane) {
        if (lane == 0) {
            Fp6 n, ni;
            Fp2 *nc[3] = {&n.c0, &n.c1, &n.c2};
            for (int j = 0; j < 3; j++) {              // (x0 + x1 u) v^j  <-  w^(2j): x0 - x1, w^(2j+6): x1
                nc[j]->c1 = m.t4.c[2 * j + 6];
                fp_add(nc[j]->c0, m.t4.c[2 * j], m.t4.c[2 * j + 6]);
            }
            fp6_inv(ni, n);
            const Fp2 *ic[3] = {&ni.c0, &ni.c1, &ni.c2};
            for (int k = 0; k < 12; k++) m.t4.c[k] = fp_zero();
            for (int j = 0; j < 3; j++) {
                fp_sub(m.t4.c[2 * j], ic[j]->c0, ic[j]->c1);
                m.t4.c[2 * j + 6] = ic[j]->c1;
            }
        }
    }
    COOP_SYNC();
    coop_mul(m, dst, m.t3, m.t4, EVEN_MASK);
}

struct FrobTables { Fp a1[12], b1[12], a2[12]; };     // power-1 tables and the power-2 table (its g1 part is zero)

// p^2-power Frobenius: gamma = xi^((p^2-1)/6) is a 6th root of unity in Fp, so it is a plain coefficient scaling.
KZG_HD void coop_frob2(Fp12W &dst, const Fp12W &a, const Fp *tab) {
    COOP_LANES(lane) {
        if (lane < 12) { Fp x; fp_mul(x, a.c[lane], tab[lane]); dst.c[lane] = x; }
    }
    COOP_SYNC();
}

// a^x for a in the cyclotomic subgroup (x < 0 -> conjugate).  Uses t3 as scratch.  dst must not alias a.
KZG_HD void coop_cyc_exp_x(CoopMem &m, Fp12W &dst, const Fp12W &a) {
    coop_copy(dst, a);
    for (int i = 62; i >= 0; i--) {
        coop_sqr(m, dst, dst);
        if ((BLS_X_ABS >> i) & 1) coop_mul(m, dst, dst, a, FULL_MASK);
    }
    coop_conj(dst, dst);
}

// The whole check  ML(Q1, P1) * ML(Q2, P2) -> final exponentiation -> == 1, for precomputed line tables of Q1, Q2.
// p1 / p2 = (0,0) (infinity) makes that side contribute 1.
KZG_HD bool coop_pairing_check(CoopMem &m, const LineW *lines1, const G1Affine &p1, const LineW *lines2, const G1Affine &p2,
                               const FrobTables &ft) {
    const bool use1 = !g1a_is_inf(p1), use2 = !g1a_is_inf(p2);
    COOP_LANES(lane) {
        if (lane == 0) { m.px[0] = p1.x; m.py[0] = p1.y; m.px[1] = p2.x; m.py[1] = p2.y; }
        if (lane < 24) { m.line[lane / 12].c[lane % 12] = fp_zero(); }
    }
    COOP_SYNC();
    coop_set_one(m.f);
    int n = 0;
    for (int i = 62; i >= 0; i--) {
        coop_sqr(m, m.f, m.f);
        const int steps = 1 + (int)((BLS_X_ABS >> i) & 1);
        for (int s = 0; s < steps; s++, n++) {
            COOP_LANES(lane) {                          // evaluate both lines at their points: 8 products on 8 lanes
                if (lane < 12) {
                    const int q = lane / 6, e = lane % 6;
                    const LineW &L = q == 0 ? lines1[n] : lines2[n];
                    Fp v;
                    switch (e) {
                        case 0: m.line[q].c[0] = L.l0; break;
                        case 1: m.line[q].c[6] = L.l6; break;
                        case 2: fp_mul(v, L.l2, m.px[q]); m.line[q].c[2] = v; break;
                        case 3: fp_mul(v, L.l8, m.px[q]); m.line[q].c[8] = v; break;
                        case 4: fp_mul(v, L.l3, m.py[q]); m.line[q].c[3] = v; break;
                        default: fp_mul(v, L.l9, m.py[q]); m.line[q].c[9] = v; break;
                    }
                }
            }
            COOP_SYNC();
            if (use1) coop_mul(m, m.f, m.f, m.line[0], LINE_MASK);
            if (use2) coop_mul(m, m.f, m.f, m.line[1], LINE_MASK);
        }
    }
    coop_conj(m.f, m.f);                                // x < 0
    // final exponentiation, same chain as pairing.h final_exp_is_one
    coop_conj(m.t0, m.f); coop_inv(m, m.t1, m.f); coop_mul(m, m.f, m.t0, m.t1, FULL_MASK);          // ^(p^6-1)
    coop_frob2(m.t0, m.f, ft.a2); coop_mul(m, m.f, m.t0, m.f, FULL_MASK);                          // ^(p^2+1)
    // hard part: f^((x-1)^2 (x+p)(x^2+p^2-1)) * f^3
    coop_cyc_exp_x(m, m.t1, m.f); coop_conj(m.t0, m.f); coop_mul(m, m.t1, m.t1, m.t0, FULL_MASK);   // a = f^(x-1)
    coop_cyc_exp_x(m, m.t2, m.t1); coop_conj(m.t0, m.t1); coop_mul(m, m.t1, m.t2, m.t0, FULL_MASK); // a = a^(x-1)
    coop_cyc_exp_x(m, m.t2, m.t1); coop_frob(m.t0, m.t1, ft.a1, ft.b1); coop_mul(m, m.t2, m.t2, m.t0, FULL_MASK);   // b = a^(x+p)
    coop_cyc_exp_x(m, m.t1, m.t2); coop_cyc_exp_x(m, m.t0, m.t1);                                    // b^(x^2) -> t0
    coop_frob2(m.t1, m.t2, ft.a2); coop_mul(m, m.t0, m.t0, m.t1, FULL_MASK);                        // * b^(p^2)
    coop_conj(m.t1, m.t2); coop_mul(m, m.t0, m.t0, m.t1, FULL_MASK);                                 // * b^-1
    coop_sqr(m, m.t1, m.f); coop_mul(m, m.t1, m.t1, m.f, FULL_MASK);                                 // f^3
    coop_mul(m, m.t0, m.t0, m.t1, FULL_MASK);
    return coop_is_one(m, m.t0);
}

}  // namespace kzg
