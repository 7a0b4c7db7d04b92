// pairing_coop.h -- wave-cooperative pairing check: one 64-lane wavefront per pairing product.
//
// Why: on one lane the pairing check of verify_kzg_proof_batch (reference src/utils.rs:189-214) is a single chain of
// ~20k dependent Fp products; a lone gfx950 wave issues one instruction every ~5 cycles, so that chain costs ~75 ms
// and bounds the latency of a whole 64-blob batch (profiles/r01).  Here the 12 Fp coefficients of an Fp12 element are
// spread over the lanes of a wave and every Fp12 operation becomes two short phases:
//
//   representation   Fp12 = Fp[w]/(w^12 - 2 w^6 + 2)  (w^2 = v, w^6 = 1 + u): a = sum_{k<12} a_k w^k, a_k in Fp.
//                    Tower coefficient (x0 + x1 u) v^j w^i sits at m = 2j + i:  a_m += x0 - x1,  a_{m+6} += x1.
//   product          phase 1: a lane takes <= 3 limb products a_i b_j of ONE convolution index s = i + j, accumulates them
//                    UNREDUCED in 64-bit columns (29-bit limbs leave room for 4 products between carry sweeps) and
//                    Montgomery-reduces its own partial sum once (the reduction is linear, so the sum of reduced partials is
//                    the reduced sum) -> 14 limbs per lane in LDS.
//                    phase 2: lane k (0..11) forms coefficient k directly as a small-integer combination of those partials:
//                    the partials of d_k and the fold w^12 = 2 w^6 - 2 in one sweep
//                        k<=5: d_k - 2 d_{k+12} - 4 d_{k+18}       k>=6: d_k + 2 d_{k+6} + 2 d_{k+12}
//   square           same with d_s = 2 sum_{i<j} a_i a_j + a_{s/2}^2          (<= 2 limb products per lane)
//   line product     a Miller-loop line  c0 + c1' v + c4' v w  has only w^{0,2,3,6,8,9} -> 72 limb products, <= 2 per lane
//   cyclotomic sq.   Granger-Scott squaring for the final exponentiation's hard part (elements of the cyclotomic subgroup):
//                    36 lanes, ONE limb product each, then the same combination phase
//   Frobenius        a_k -> a_k gamma^k folded back, two constant products per lane; conjugation flips odd k.
//
// Operands live in a CoopMem block (LDS on the device); lanes only communicate through it, with a barrier between
// phases.  The same source runs on the host for unit tests: COOP_LANES loops over the 64 lanes sequentially there.
#pragma once
#include <initializer_list>
#include "pairing.h"

#if defined(__HIP_DEVICE_COMPILE__)
// One WAVE per Fp12 computation (a workgroup may hold several, each with its own CoopMem): the phases are separated by a
// wave-local fence -- LDS operations of one wave complete in order, so waiting for them is all a "barrier" has to do.
#define COOP_LANES(lane) for (int lane = (int)(threadIdx.x & 63), coop_once_ = 1; coop_once_; coop_once_ = 0)
#define COOP_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
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

constexpr int COOP_MAX_TERMS = 10;
// terms per coefficient the combination walks: product / square <= 9; line product, cyclotomic square <= 6
constexpr int COOP_T_FULL = 9, COOP_T_SPARSE = 6;
// Phase 2 of a product type (the "combination"): coefficient k = sum_t coef[k][t] * partial[lane[k][t]] + kp[k] * p.  The multiple
// of p keeps the value positive -- one more p than the negative terms need, so that the value is at least p and the top limb of the
// limb-parallel form below can never go negative; partials are < 1.04 p, so every coefficient stays below 32 p: the lazy bound of an
// Fp12W between operations.
struct CoopComb {
    uint32_t term[12][COOP_MAX_TERMS];   // (lane << 8) | (coef & 0xff); unused entries are 0 (coefficient 0 times the partial of lane 0)
    uint8_t nt[12], kp[12];
    uint32_t nmax;                       // the largest nt: the uniform trip count of the combination
};
// Work schedule of one product type.  Phase 1: which limb products lane l accumulates (all for the same convolution index s),
// packed into one word: bits 0-1 the number of (i, j) pairs (<= 3; 0: idle), bit 2 "the partial enters doubled" (cross terms
// a_i a_j, i < j, of a square), then i and j of pair k in bits 4 + 8k .. 7 + 8k and 8 + 8k .. 11 + 8k.
struct CoopSched {
    uint8_t s[64];            // convolution index this lane works for (255: idle); construction and tests only
    uint32_t prog[64];
    CoopComb comb;
};
// Granger-Scott squaring: lane l multiplies  (U[a0] + fa U[a1]) * (U[b0] + fb U[b1])  with fa, fb in {0, 1, 2} and
// U = (a_0 .. a_11, 2, 0); prog[l]: bit 0 active, a0 bits 1-4, a1 5-8, fa 9-10, b0 11-14, b1 15-18, fb 19-20.
struct CoopCycSched {
    uint32_t prog[64];
    CoopComb comb;
};
struct CoopScheds { CoopSched mul, sqr, line; CoopCycSched cyc; };
static_assert(sizeof(CoopScheds) % 4 == 0, "copied to LDS word by word");
struct CoopMem {
    Fp12W f, t0, t1, t2, t3, t4;
    Fp12W line[2];
    Fp cyc_consts[2];         // 2 (Montgomery form) and 0: entries 12, 13 of the cyclotomic squaring's operand table; must follow line[]
    Fp red[64];               // the reduced partial of every lane (phase 1 -> phase 2)
    Fp px[2], py[2], pz[2];   // the two G1 arguments as (X Z, Y, Z^3) (g1.h: PairPt)
    uint32_t pmod[16];        // the limbs of p (the combination indexes them by lane)
    CoopScheds sc;
    int flag;
};

inline void coop_comb_clear(CoopComb &cb) {
    for (int k = 0; k < 12; k++) { cb.nt[k] = 0; cb.kp[k] = 0; for (int t = 0; t < COOP_MAX_TERMS; t++) cb.term[k][t] = 0; }
    cb.nmax = 0;
}
inline bool coop_comb_term(CoopComb &cb, int k, int lane, int coef) {
    const int n = cb.nt[k];
    if (n >= COOP_MAX_TERMS) return false;
    cb.term[k][n] = ((uint32_t)lane << 8) | ((uint32_t)coef & 0xffu);
    cb.nt[k] = (uint8_t)(n + 1);
    return true;
}
// multiples of p and the bound check: partials are < 1.04 p; coefficient < 31 p
inline bool coop_comb_finish(CoopComb &cb) {
    bool ok = true;
    cb.nmax = 0;
    for (int k = 0; k < 12; k++) {
        int neg = 0, pos = 0;
        for (int t = 0; t < cb.nt[k]; t++) { const int c = (int)(int8_t)(cb.term[k][t] & 0xffu); if (c < 0) neg -= c; else pos += c; }
        const int K = (neg * 105 + 99) / 100 + 1;
        cb.kp[k] = (uint8_t)K;
        if (pos * 105 + K * 100 >= 3100) ok = false;
        if (cb.nt[k] > cb.nmax) cb.nmax = cb.nt[k];
    }
    return ok;
}

// Phase-2 tables of a product schedule from its lane -> convolution index map, with the bound check.
inline bool coop_finish_sched(CoopSched &sc) {
    bool ok = true;
    coop_comb_clear(sc.comb);
    for (int k = 0; k < 12; k++) {
        for (int l = 0; l < 64; l++) {
            const int s = sc.s[l];
            if (s == 255) continue;
            int coef = 0;
            if (s == k) coef = 1;
            else if (k <= 5) coef = s == k + 12 ? -2 : s == k + 18 ? -4 : 0;
            else coef = (s == k + 12 || s == k + 6) ? 2 : 0;
            if (!coef) continue;
            ok = coop_comb_term(sc.comb, k, l, coef) && ok;
        }
    }
    return coop_comb_finish(sc.comb) && ok;
}
// Schedules: convolution index s of a full product has c_s = min(s, 22 - s) + 1 limb products; it gets ceil(c_s / 3)
// lanes (56 lanes in all).  A square has floor(c_s / 2) cross products (2 per lane, doubled) plus a_{s/2}^2 on a lane of
// its own for even s.  A line product only has the 72 products with j in {0, 2, 3, 6, 8, 9}: 2 per lane.
inline bool build_coop_schedules(CoopScheds &out) {
    CoopSched &mul = out.mul, &sqr = out.sqr, &lin = out.line;
    for (CoopSched *sc : {&mul, &sqr, &lin})
        for (int l = 0; l < 64; l++) { sc->s[l] = 255; sc->prog[l] = 0; }
    // pair k of lane l (k = the lane's current count)
    auto add_pair = [](CoopSched &sc, int l, int i, int j, bool dbl) {
        const int k = (int)(sc.prog[l] & 3u);
        sc.prog[l] = (sc.prog[l] & ~3u) | (uint32_t)(k + 1) | (dbl ? 4u : 0u) | ((uint32_t)i << (4 + 8 * k)) | ((uint32_t)j << (8 + 8 * k));
    };
    bool ok = true;
    int lane = 0;
    for (int s = 0; s < 23; s++) {
        int k = 0;
        for (int i = 0; i < 12; i++) {
            const int j = s - i;
            if (j < 0 || j > 11) continue;
            if (k == 3) { lane++; k = 0; }
            if (lane < 64) { mul.s[lane] = (uint8_t)s; add_pair(mul, lane, i, j, false); }
            k++;
        }
        lane++;
    }
    ok = ok && lane <= 64;
    lane = 0;
    for (int s = 0; s < 23; s++) {
        int k = 0; bool any = false;
        for (int i = 0; i < 12; i++) {
            const int j = s - i;
            if (j <= i || j > 11) continue;
            if (k == 2) { lane++; k = 0; }
            if (lane < 64) { sqr.s[lane] = (uint8_t)s; add_pair(sqr, lane, i, j, true); }
            k++;
            any = true;
        }
        if (any) lane++;
        if (!(s & 1)) {
            if (lane < 64) { sqr.s[lane] = (uint8_t)s; add_pair(sqr, lane, s >> 1, s >> 1, false); }
            lane++;
        }
    }
    ok = ok && lane <= 64;
    lane = 0;
    for (int s = 0; s < 23; s++) {
        int k = 0; bool any = false;
        for (int i = 0; i < 12; i++) {
            const int j = s - i;
            if (j < 0 || j > 11 || !((LINE_MASK >> j) & 1u)) continue;
            if (k == 2) { lane++; k = 0; }
            if (lane < 64) { lin.s[lane] = (uint8_t)s; add_pair(lin, lane, i, j, false); }
            k++;
            any = true;
        }
        if (any) lane++;
    }
    ok = ok && lane <= 64;
    ok = coop_finish_sched(mul) && ok;
    ok = coop_finish_sched(sqr) && ok;
    ok = coop_finish_sched(lin) && ok;
    // Granger-Scott squaring of g = sum g_i w^i, g_i = x0 + x1 u in Fp2 with x0 = a_i + a_{i+6}, x1 = a_{i+6}.  For the three
    // Fp4 elements (g_t, g_{t+3}), t = 0, 1, 2, with (x, y) = (g_t, g_{t+3}) and a = a_t, b = a_{t+6}, c = a_{t+3}, d = a_{t+9}:
    //   P1 = (a + 2b) a   [x0^2 - x1^2]   P2 = (a + b) b  [x0 x1]   P3 = (c + 2d) c  [y0^2 - y1^2]   P4 = (c + d) d  [y0 y1]
    //   P5 = (a + b)(c + d) [x0 y0]       P6 = b d [x1 y1]          P7 = (a + b) d [x0 y1]           P8 = b (c + d) [x1 y0]
    // and E_k = 2 a_k.  g^2 = (3 A^2 - 2 conj A) + (3 s C^2 + 2 conj B) w + (3 B^2 - 2 conj C) w^2 over Fp4 = Fp2[s], s = w^3, gives
    //   t = 0:  a_0' = 3P1 - 6P2 - 12P4 - E0,  a_6' = 6P2 + 3P3 + 6P4 - E6,  a_3' = 6P5 - 6P6 - 6P7 - 6P8 + E3,  a_9' = 6P7 + 6P8 + E9
    //   t = 1:  the same four rows for (a_2', a_8', a_5', a_11')
    //   t = 2:  a_4', a_10' as the first two rows;  a_1' = -12P7 - 12P8 + E1,  a_7' = 6P5 - 6P6 + 6P7 + 6P8 + E7
    // (checked against the generic square by tests/test_device_math_host.py).
    CoopCycSched &cy = out.cyc;
    for (int l = 0; l < 64; l++) cy.prog[l] = 0;
    auto prod = [&](int l, int a0, int a1, int fa, int b0, int b1, int fb) {
        cy.prog[l] = 1u | ((uint32_t)a0 << 1) | ((uint32_t)a1 << 5) | ((uint32_t)fa << 9) | ((uint32_t)b0 << 11) | ((uint32_t)b1 << 15) | ((uint32_t)fb << 19);
    };
    for (int t = 0; t < 3; t++) {
        const int a = t, b = t + 6, c = t + 3, d = t + 9, l = 8 * t;
        prod(l + 0, a, b, 2, a, 13, 0);     // P1
        prod(l + 1, a, b, 1, b, 13, 0);     // P2
        prod(l + 2, c, d, 2, c, 13, 0);     // P3
        prod(l + 3, c, d, 1, d, 13, 0);     // P4
        prod(l + 4, a, b, 1, c, d, 1);      // P5
        prod(l + 5, b, 13, 0, d, 13, 0);    // P6
        prod(l + 6, a, b, 1, d, 13, 0);     // P7
        prod(l + 7, b, 13, 0, c, d, 1);     // P8
    }
    for (int k = 0; k < 12; k++) prod(24 + k, k, 13, 0, 12, 13, 0);       // E_k = 2 a_k
    coop_comb_clear(cy.comb);
    auto term = [&](int k, int l, int coef) { ok = coop_comb_term(cy.comb, k, l, coef) && ok; };
    const int lo_of[3] = {0, 2, 4}, hi_of[3] = {6, 8, 10};
    for (int t = 0; t < 3; t++) {
        const int l = 8 * t, lo = lo_of[t], hi = hi_of[t];
        term(lo, l + 0, 3); term(lo, l + 1, -6); term(lo, l + 3, -12); term(lo, 24 + lo, -1);
        term(hi, l + 1, 6); term(hi, l + 2, 3); term(hi, l + 3, 6); term(hi, 24 + hi, -1);
    }
    for (int t = 0; t < 2; t++) {
        const int l = 8 * t, k3 = t == 0 ? 3 : 5, k9 = t == 0 ? 9 : 11;
        term(k3, l + 4, 6); term(k3, l + 5, -6); term(k3, l + 6, -6); term(k3, l + 7, -6); term(k3, 24 + k3, 1);
        term(k9, l + 6, 6); term(k9, l + 7, 6); term(k9, 24 + k9, 1);
    }
    term(1, 16 + 6, -12); term(1, 16 + 7, -12); term(1, 24 + 1, 1);
    term(7, 16 + 4, 6); term(7, 16 + 5, -6); term(7, 16 + 6, 6); term(7, 16 + 7, 6); term(7, 24 + 7, 1);
    ok = coop_comb_finish(cy.comb) && ok;
    ok = ok && mul.comb.nmax <= (uint32_t)COOP_T_FULL && sqr.comb.nmax <= (uint32_t)COOP_T_FULL && lin.comb.nmax <= (uint32_t)COOP_T_SPARSE &&
            cy.comb.nmax <= (uint32_t)COOP_T_SPARSE;
    return ok;
}

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
// Montgomery reduction of a carried accumulator holding T < 2^20 p^2:  r = T / 2^406 mod p, LAZY: r < p (1 + 2^-5), not
// reduced below p (the coefficients of an Fp12W stay unreduced, < 32p, between operations; see coop_fold).
KZG_HD void wide_reduce(Fp &r, uint64_t *acc) {
    const uint32_t m[NFP] = FP_MOD_INIT;
#pragma unroll
    for (int i = 0; i < NFP; i++) {
        const uint32_t q = ((uint32_t)acc[i] * FP_INVW) & LMASK;
#pragma unroll
        for (int j = 0; j < NFP; j++) acc[i + j] += (uint64_t)q * m[j];
        acc[i + 1] += acc[i] >> LB;
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NFP; j++) { c += acc[NFP + j]; r.l[j] = j < NFP - 1 ? ((uint32_t)c & LMASK) : (uint32_t)c; c >>= LB; }
}

// ---------------------------------------------------------------------------------- cooperative Fp12 operations
// Phase 2 of every product, LIMB-parallel: lane (k, j) forms limb j of coefficient k,
//     acc_j = bias_j + sum_t coef_t * red[lane_t].l[j]        (one v_mad_i64_i32 per term),
// where the bias is K p written so that every limb below the top one is large -- K p_j + 2^35 at limb j < 13, the 2^35 taken back
// as 2^6 from limb j + 1 -- which keeps acc_j >= 0 whatever the negative terms (< 26 * 2^29 in all).  One carry step to the
// neighbouring lane (a DPP row shift on the device) then leaves limbs in [0, 2^29 + 2^7): "almost normalised", which is all the
// 64-bit product columns of the next operation need (14 * (2^29 + 2^7)^2 < 2^62); the top limb keeps the excess, and is >= 12
// because the value is >= p (CoopComb).  Rows of 16 lanes hold one coefficient (limbs 0..13, two idle lanes), four coefficients per
// pass, three passes: ~4 instructions per term instead of a 14-limb sweep per term on 12 lanes.
// T: compile-time number of terms walked (>= the schedule's nmax; unused entries have coefficient 0): a fixed-length, fully unrolled
// body lets all table entries and then all partial limbs be fetched back to back -- a loop pays two dependent LDS round trips per term.
template <int T> KZG_HD void coop_combine_all(CoopMem &m, const CoopComb &cb, Fp12W &dst) {
    COOP_LANES(lane) {
        const int kq = lane >> 4, jr = lane & 15, j = jr < NFP ? jr : NFP - 1;
        uint32_t e[3][T];
        int32_t v[3][T];
        int32_t kp[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            kp[q] = (int32_t)cb.kp[4 * q + kq];
#pragma unroll
            for (int t = 0; t < T; t++) e[q][t] = cb.term[4 * q + kq][t];
        }
        const int64_t pj = (int64_t)m.pmod[j];
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int t = 0; t < T; t++) v[q][t] = (int32_t)m.red[e[q][t] >> 8].l[j];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + kq;
            int64_t acc = (int64_t)kp[q] * pj + (j < NFP - 1 ? ((int64_t)1 << 35) : 0) - (j > 0 ? 64 : 0);
#pragma unroll
            for (int t = 0; t < T; t++) acc += (int64_t)(int32_t)(int8_t)(e[q][t] & 0xffu) * (int64_t)v[q][t];       // 32 x 32 -> 64 signed: one v_mad_i64_i32
            const uint32_t hi = (uint32_t)(acc >> LB);                                    // < 2^7 below the top limb
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
#else
            uint32_t hp = 0;
            if (jr > 0 && jr < NFP) {                                                     // the neighbouring lane's carry, recomputed
                int64_t a2 = (int64_t)kp[q] * (int64_t)m.pmod[jr - 1] + ((int64_t)1 << 35) - (jr - 1 > 0 ? 64 : 0);
                for (int t = 0; t < T; t++) a2 += (int64_t)(int32_t)(int8_t)(e[q][t] & 0xffu) * (int64_t)(int32_t)m.red[e[q][t] >> 8].l[jr - 1];
                hp = (uint32_t)(a2 >> LB);
            }
#endif
            const uint32_t out = (jr < NFP - 1 ? ((uint32_t)acc & LMASK) : (uint32_t)acc) + (jr > 0 ? hp : 0u);
            if (jr < NFP) dst.c[k].l[jr] = out;
        }
    }
    COOP_SYNC();
}
// limbs of a lazy coefficient -> normalised (below 2^29 except the top one); in front of routines that assume normalised limbs
KZG_HD void fp_norm_lz(Fp &r, const Fp &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) { const uint32_t t = a.l[i] + c; if (i < NFP - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t; }
}

// dst = a * b (sc = mul / line schedule; for the line schedule b has non-zero coefficients only at w^{0,2,3,6,8,9}) or, with
// sc = the square schedule and b = a, dst = a^2.  bmask: coefficients of b known to be zero are skipped.  dst may alias a or b.
// bs (or null): the line operand as its six non-zero coefficients (l0, l6, l2, l8, l3, l9 -- the layout of the evaluations made ahead of
// the loop, coop_eval_lines_item) instead of an Fp12W slot: coefficient j of b is bs[slot(j)].
KZG_HD void coop_product(CoopMem &m, const CoopSched &sc, Fp12W &dst, const Fp12W &a, const Fp12W &b, uint32_t bmask, const Fp *bs = nullptr) {
    COOP_LANES(lane) {
        const uint32_t pg = sc.prog[lane];
        const int np = (int)(pg & 3u);
        if (np) {
            uint64_t acc[2 * NFP];
            wide_zero(acc);
            for (int k = 0; k < np; k++) {
                const int i = (int)((pg >> (4 + 8 * k)) & 15u), j = (int)((pg >> (8 + 8 * k)) & 15u);
                const uint32_t *bj = bs ? bs[(0x5301004200ull >> (4 * j)) & 7u].l : b.c[j].l;       // j = 0, 2, 3, 6, 8, 9 -> slot 0, 2, 4, 1, 3, 5
                if ((bmask >> j) & 1u) wide_mac(acc, a.c[i].l, bj);
            }
            // no carry sweep in front of the reduction: with limbs < 2^29 + 2^7 three products leave every column below 42 * 2^58, a doubled
            // pair of cross products below 56 * 2^58, and the reduction adds at most 14 * 2^58 + 2^35: < 2^64 either way
            if (pg & 4u) wide_double(acc);
            Fp r; wide_reduce(r, acc);
            m.red[lane] = r;
        }
    }
    COOP_SYNC();
    if (&sc == &m.sc.line) coop_combine_all<COOP_T_SPARSE>(m, sc.comb, dst); else coop_combine_all<COOP_T_FULL>(m, sc.comb, dst);
}
KZG_HD void coop_mul(CoopMem &m, Fp12W &dst, const Fp12W &a, const Fp12W &b, uint32_t bmask) { coop_product(m, m.sc.mul, dst, a, b, bmask); }
KZG_HD void coop_sqr(CoopMem &m, Fp12W &dst, const Fp12W &a) { coop_product(m, m.sc.sqr, dst, a, a, FULL_MASK); }

// dst = a^2 for a in the cyclotomic subgroup (a^(p^6+1) = 1 and a^(p^4-p^2+1) = 1): Granger-Scott.  dst may alias a.
KZG_HD void coop_cyc_sqr(CoopMem &m, Fp12W &dst, const Fp12W &a) {
    const CoopCycSched &sc = m.sc.cyc;
    COOP_LANES(lane) {
        const uint32_t pg = sc.prog[lane];
        if (pg & 1u) {
            const Fp *U = a.c;                                   // entries 12, 13 come from m.cyc_consts
            auto get = [&](int idx) -> const Fp & { return idx < 12 ? U[idx] : m.cyc_consts[idx - 12]; };
            // o = U[i0] + f U[i1], f in {0, 1, 2}: limbs normalised below the top one
            auto operand = [&](Fp &o, int i0, int i1, uint32_t f) {
                const Fp &x = get(i0), &y = get(i1);
                const uint32_t sh = f >> 1, keep = f ? 0xffffffffu : 0u;
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < NFP; i++) {
                    const uint32_t t = x.l[i] + ((y.l[i] << sh) & keep) + c;
                    if (i < NFP - 1) { c = t >> LB; o.l[i] = t & LMASK; } else o.l[i] = t;
                }
            };
            Fp x, y;
            operand(x, (int)((pg >> 1) & 15u), (int)((pg >> 5) & 15u), (pg >> 9) & 3u);
            operand(y, (int)((pg >> 11) & 15u), (int)((pg >> 15) & 15u), (pg >> 19) & 3u);
            uint64_t acc[2 * NFP];
            wide_zero(acc);
            wide_mac(acc, x.l, y.l);
            Fp r; wide_reduce(r, acc);
            m.red[lane] = r;
        }
    }
    COOP_SYNC();
    coop_combine_all<COOP_T_SPARSE>(m, sc.comb, dst);
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
        if (lane < 12) {                                         // odd k: 32p - a_k (coefficients are lazy, < 32p)
            const uint32_t m32[NFP] = FP_MOD32_INIT;
            Fp t = a.c[lane];
            if (lane & 1) { const Fp z = fp_zero(); fp_sub_lz(t, z, t, m32); }
            dst.c[lane] = t;
        }
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
            const uint32_t m4[NFP] = FP_MOD4_INIT;
            fp_mul_lz(x, a.c[lane], tabA[lane]);
            if (lane < 6) { fp_mul_lz(y, a.c[lane + 6], tabB[lane]); fp_add_lz(y, y, y); fp_sub_lz(x, x, y, m4); }
            else { fp_mul_lz(y, a.c[lane - 6], tabB[lane]); fp_add_lz(x, x, y); }
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
            Fp c; fp_norm_lz(c, a.c[lane]); fp_canon64(c, c);
            if (!fp_eq(c, want)) m.flag = 0;
        }
    }
    COOP_SYNC();
    return m.flag != 0;
}
// In-place inverse of an Fp6 element stored in the w basis (even powers of w only; the odd coefficients are zero).  With
// N = a * conj(a) in Fp6 this gives a^-1 = conj(a) * N^-1.  Round 4: cooperative -- the ~40 base-field products of the tower formulas
//     A = a0^2 - xi a1 a2,  B = xi a2^2 - a0 a1,  C = a1^2 - a0 a2,  D = a0 A + xi (a2 B + a1 C),  a^-1 = (A, B, C) / D      (xi = 1 + u)
// are five stages of independent products, one per lane (schoolbook Fp2 products: x0 y0, x1 y1, x0 y1, x1 y0 on four lanes), with the
// sums in between on a few lanes; only the one Fp inversion (divsteps, ~25k instructions) stays on a single lane.  Operands and results
// live in LDS (V: 24 scratch values in two free Fp12 slots; m.red: the products), nothing on the private stack -- the single-lane tower
// version kept two Fp6 and a dozen Fp2 temporaries in scratch memory and took 0.19 of the 1.38 ms of a lone pairing check.
// red[l] = V[ia[l]] * V[ib[l]] for l < n
KZG_HD void coop_fp_products(CoopMem &m, const Fp *V, int n, const uint8_t *ia, const uint8_t *ib) {
    COOP_LANES(lane) { if (lane < n) fp_mul(m.red[lane], V[ia[lane]], V[ib[lane]]); }
    COOP_SYNC();
}
// dst[l] = sum of +red[t - 1] / -red[-t - 1] over row l of tab (0 ends a row; an empty row gives 0) for l < n; canonical in, canonical out
constexpr int COOP_LIN_W = 10;
KZG_HD void coop_fp_lincomb(CoopMem &m, Fp *dst, int n, const int8_t (*tab)[COOP_LIN_W]) {
    COOP_LANES(lane) {
        if (lane < n) {
            Fp acc = fp_zero();
            for (int i = 0; i < COOP_LIN_W; i++) {
                const int t = tab[lane][i];
                if (!t) break;
                if (t > 0) fp_add(acc, acc, m.red[t - 1]); else fp_sub(acc, acc, m.red[-t - 1]);
            }
            dst[lane] = acc;
        }
    }
    COOP_SYNC();
}
// V: 24 Fp of scratch (two Fp12 slots the program does not hold anything in at this point)
KZG_HD void coop_fp6_inv(CoopMem &m, Fp12W &x, Fp *V) {
#include "fp6inv_tables.inc"
    COOP_LANES(lane) {      // lazy -> canonical: (lo_0, hi_0, lo_1, hi_1, lo_2, hi_2) = coefficients 0, 6, 2, 8, 4, 10
        if (lane < 6) { Fp c; fp_norm_lz(c, x.c[2 * (lane >> 1) + 6 * (lane & 1)]); fp_canon64(c, c); m.red[lane] = c; }
    }
    COOP_SYNC();
    coop_fp_lincomb(m, V, 6, l0);
    coop_fp_products(m, V, 24, s1a, s1b);
    coop_fp_lincomb(m, V + 6, 6, l2);
    coop_fp_products(m, V, 12, s3a, s3b);
    coop_fp_lincomb(m, V + 12, 2, l4);
    coop_fp_products(m, V, 2, s5a, s5b);
    COOP_LANES(lane) {
        if (lane == 0) { Fp n; fp_add(n, m.red[0], m.red[1]); fp_inv(n, n); V[14] = n; }
    }
    COOP_SYNC();
    coop_fp_products(m, V, 2, s7a, s7b);
    coop_fp_lincomb(m, V + 15, 2, l7);
    coop_fp_products(m, V, 12, s8a, s8b);
    coop_fp_lincomb(m, x.c, 12, l9);
}

struct FrobTables { Fp a1[12], b1[12], a2[12]; };     // power-1 tables and the power-2 table (its g1 part is zero)

// p^2-power Frobenius: gamma = xi^((p^2-1)/6) is a 6th root of unity in Fp, so it is a plain coefficient scaling.
KZG_HD void coop_frob2(Fp12W &dst, const Fp12W &a, const Fp *tab) {
    COOP_LANES(lane) {
        if (lane < 12) { Fp x; fp_mul_lz(x, a.c[lane], tab[lane]); dst.c[lane] = x; }
    }
    COOP_SYNC();
}

// ---------------------------------------------------------------------------------- the pairing as a program
// Every call site of coop_mul / coop_sqr is a ~4k-instruction inlined body, and the pairing check has dozens of
// them; spelled out as straight-line code the kernel was >0.5 MB of instructions (far beyond the 64 KB instruction
// cache).  Instead the check is a flat list of instructions over 8 Fp12 slots, built once by build_pairing_program()
// on the host, and the kernel is a small interpreter with ONE body per opcode.
enum : uint8_t { OP_SET_ONE, OP_SQR, OP_MUL, OP_MUL_LINE0, OP_MUL_LINE1, OP_MUL_EVEN, OP_LINE_EVAL, OP_CONJ, OP_FROB1, OP_FROB2, OP_FP6INV, OP_COPY,
        OP_CYC_SQR };
struct CoopInsn { uint8_t op, dst, a, b; };
enum : uint8_t { S_F = 0, S_T0 = 1, S_T1 = 2, S_T2 = 3, S_T3 = 4, S_T4 = 5, S_L0 = 6, S_L1 = 7 };
constexpr int COOP_PROGRAM_MAX = 1024;
// line steps evaluated at once inside the Miller loop: 5 steps x 2 pairs x 6 coefficients = 60 lanes, 60 Fp = the slots t0 .. t4
constexpr int COOP_LINE_CHUNK = 5;

// hard_start (or null): receives the index of the first instruction of the hard part of the final exponentiation -- from there on the program only
// uses the slots F, T0, T1, T2 and the operations COPY, CYC_SQR, MUL, CONJ, FROB1, FROB2 (pairing_lanes.h runs that tail twelve lanes per check)
inline int build_pairing_program(CoopInsn *p, int *hard_start = nullptr) {
    int n = 0;
    auto emit = [&](uint8_t op, uint8_t d, uint8_t a, uint8_t b) { p[n].op = op; p[n].dst = d; p[n].a = a; p[n].b = b; n++; };
    auto cyc_exp_x = [&](uint8_t d, uint8_t a) {           // d = a^x (x < 0), a in the cyclotomic subgroup: square-and-multiply over |x|, then conjugate
        emit(OP_COPY, d, a, 0);
        for (int i = 62; i >= 0; i--) { emit(OP_CYC_SQR, d, d, 0); if ((BLS_X_ABS >> i) & 1) emit(OP_MUL, d, d, a); }
        emit(OP_CONJ, d, d, 0);
    };
    // Miller loops of both pairs, sharing the squarings (utils.rs:206-209)
    emit(OP_SET_ONE, S_F, 0, 0);
    int line = 0;
    for (int i = 62; i >= 0; i--) {
        emit(OP_SQR, S_F, S_F, 0);
        const int steps = 1 + (int)((BLS_X_ABS >> i) & 1);
        for (int s = 0; s < steps; s++, line++) {
            emit(OP_LINE_EVAL, 0, (uint8_t)line, 0);
            emit(OP_MUL_LINE0, S_F, S_F, S_L0);
            emit(OP_MUL_LINE1, S_F, S_F, S_L1);
        }
    }
    emit(OP_CONJ, S_F, S_F, 0);                                                       // x < 0
    // final exponentiation (utils.rs:210): easy part, then 3 * hard part = (x-1)^2 (x+p)(x^2+p^2-1) + 3
    emit(OP_CONJ, S_T0, S_F, 0);                                                      // f^-1 = conj(f) * N^-1, N = f conj(f) in Fp6
    emit(OP_MUL, S_T4, S_F, S_T0); emit(OP_FP6INV, S_T4, S_T4, 0); emit(OP_MUL_EVEN, S_T1, S_T0, S_T4);
    emit(OP_MUL, S_F, S_T0, S_T1);                                                    // ^(p^6-1)
    emit(OP_FROB2, S_T0, S_F, 0); emit(OP_MUL, S_F, S_T0, S_F);                               // ^(p^2+1)
    if (hard_start) *hard_start = n;
    cyc_exp_x(S_T1, S_F); emit(OP_CONJ, S_T0, S_F, 0); emit(OP_MUL, S_T1, S_T1, S_T0);        // a = f^(x-1)
    cyc_exp_x(S_T2, S_T1); emit(OP_CONJ, S_T0, S_T1, 0); emit(OP_MUL, S_T1, S_T2, S_T0);      // a = a^(x-1)
    cyc_exp_x(S_T2, S_T1); emit(OP_FROB1, S_T0, S_T1, 0); emit(OP_MUL, S_T2, S_T2, S_T0);     // b = a^(x+p)
    cyc_exp_x(S_T1, S_T2); cyc_exp_x(S_T0, S_T1);                                             // b^(x^2)
    emit(OP_FROB2, S_T1, S_T2, 0); emit(OP_MUL, S_T0, S_T0, S_T1);                            // * b^(p^2)
    emit(OP_CONJ, S_T1, S_T2, 0); emit(OP_MUL, S_T0, S_T0, S_T1);                             // * b^-1
    emit(OP_CYC_SQR, S_T1, S_F, 0); emit(OP_MUL, S_T1, S_T1, S_F);                            // f^3
    emit(OP_MUL, S_T0, S_T0, S_T1);                                                           // verdict: slot T0 == 1 ?
    return n;
}

// the slots t0 .. t4 as one row of 60 field elements (the Miller loop parks its line evaluations there; f, t0 .. t4 are laid out contiguously)
KZG_HD Fp *coop_line_buf(CoopMem &m) { return reinterpret_cast<Fp *>(&m.t0); }
KZG_HD Fp12W &coop_slot(CoopMem &m, int s) {
    Fp12W *base = &m.f;                          // f, t0..t4, line[0], line[1] are laid out contiguously
    return base[s];
}

// number of instructions of the Miller-loop part of the program (everything before the final exponentiation's first CONJ):
// SET_ONE, then per bit SQR and per line LINE_EVAL + two line products; the conjugation for x < 0 belongs to the tail
constexpr int COOP_MILLER_INSNS = 1 + 63 + 3 * N_LINES;

// The Miller loops of a lone check on SEVERAL waves per pair.  f = prod_i l_i^(2^(e_i)) is a product over the lines, so a wave may own a segment of the
// iterations: it starts from f = 1 at its first iteration, multiplies its segment's lines in, and only squares from the end of the segment to the end
// of the loop; the product of all the waves' values is the loop's.  With K segments per pair the dependent chain is the 63 squarings plus ONE segment's
// line products instead of all 68.  The split (host side): segment boundaries that equalise  (lines of the segment) x c_line + (squarings from the
// segment's start to the end) x c_sqr  over the segments, with the measured costs of the two operations (profiles/r04/pairing_op_breakdown.txt).
constexpr int MILLER_SPLIT_MAX = 4;
struct MillerSplit { int k; int pc_start[MILLER_SPLIT_MAX]; int pc_lines_end[MILLER_SPLIT_MAX]; };
inline MillerSplit miller_split(const CoopInsn *prog, int k, double c_sqr = 2.3, double c_line = 2.1) {
    int sq[64], steps[64], n_it = 0;
    for (int pc = 0; pc < 1 + 63 + 3 * N_LINES; pc++) {
        if (prog[pc].op == OP_SQR) { sq[n_it] = pc; steps[n_it] = 0; n_it++; }
        else if (prog[pc].op == OP_MUL_LINE0 && n_it > 0) steps[n_it - 1]++;
    }
    MillerSplit sp; sp.k = k < 1 ? 1 : k > MILLER_SPLIT_MAX ? MILLER_SPLIT_MAX : k;
    int bound[MILLER_SPLIT_MAX + 1];
    auto fits = [&](double T) {                                  // segments from the END of the loop backwards, each as long as T allows
        int b = n_it;
        bound[sp.k] = n_it;
        for (int j = sp.k - 1; j >= 0; j--) {
            double lines = 0;
            while (b > 0 && lines + steps[b - 1] * c_line + c_sqr * (n_it - (b - 1)) <= T) { b--; lines += steps[b] * c_line; }
            bound[j] = b;
        }
        return b == 0;
    };
    double lo = 0, hi = n_it * (c_sqr + 2 * c_line) + 1;
    for (int it = 0; it < 40; it++) { const double mid = 0.5 * (lo + hi); if (fits(mid)) hi = mid; else lo = mid; }
    fits(hi);
    bound[0] = 0;
    for (int j = 0; j < MILLER_SPLIT_MAX; j++) { sp.pc_start[j] = 0; sp.pc_lines_end[j] = 0x7fffffff; }
    for (int j = 0; j < sp.k; j++) {
        const int end_pc = 1 + 63 + 3 * N_LINES;
        sp.pc_start[j] = j == 0 ? 0 : bound[j] < n_it ? sq[bound[j]] : end_pc;       // (segment 0 starts at SET_ONE; an empty segment runs nothing)
        sp.pc_lines_end[j] = j == sp.k - 1 || bound[j + 1] >= n_it ? 0x7fffffff : sq[bound[j + 1]];
    }
    return sp;
}

// Interpreter, in three pieces so that the two Miller loops of a check can also run on two waves (k_pairing.hip):
//   coop_init     schedules and the two G1 points into the wave's CoopMem
//   coop_run      instructions [pc0, pc1); use1 / use2 = false skips that pair's line products (a pair at infinity contributes 1)
//   coop_is_one   the verdict (slot T0 == 1)
KZG_HD void coop_init(CoopMem &m, const CoopScheds *scheds, const PairPt &p1, const PairPt &p2) {
    COOP_LANES(lane) {
        {   // bring the schedules next to the data (word-wise copy, 64 lanes)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(scheds);
            uint32_t *dstw = reinterpret_cast<uint32_t *>(&m.sc);
            for (int o = lane; o < (int)(sizeof(CoopScheds) / 4); o += 64) dstw[o] = src[o];
        }
        if (lane == 0) {
            m.px[0] = p1.ax; m.py[0] = p1.ay; m.pz[0] = p1.az; m.px[1] = p2.ax; m.py[1] = p2.ay; m.pz[1] = p2.az;
            Fp two = fp_one(); fp_add(two, two, two);
            m.cyc_consts[0] = two; m.cyc_consts[1] = fp_zero();
        }
        if (lane < 24) { m.line[lane / 12].c[lane % 12] = fp_zero(); }
        if (lane < 16) { const uint32_t pm[NFP] = FP_MOD_INIT; uint32_t v = 0; for (int i = 0; i < NFP; i++) if (i == lane) v = pm[i]; m.pmod[lane] = v; }
    }
    COOP_SYNC();
}
// pre (or null): the line evaluations of both pairs made ahead of the loop, [pair][line][l0, l6, l2, l8, l3, l9 at the point] (coop_eval_lines)
// line_pc_end: line products are made only by instructions below it (a wave that owns a SEGMENT of a Miller loop multiplies its segment's lines
// in and only squares from there on: k_pairing_coop_split)
KZG_HD void coop_run(CoopMem &m, const CoopInsn *prog, int pc0, int pc1, const LineW *lines1, const LineW *lines2, bool use1, bool use2, const FrobTables &ft,
                     const Fp *pre = nullptr, int line_pc_end = 0x7fffffff) {
    if (pc1 <= pc0) return;
    CoopInsn nxt = prog[pc0];
    int cur_line = 0;                                             // with pre: the line the next line products take (set by OP_LINE_EVAL)
    for (int pc = pc0; pc < pc1; pc++) {
        const CoopInsn in = nxt;
        if (pc + 1 < pc1) nxt = prog[pc + 1];                    // fetched a whole operation ahead of its use
        Fp12W &dst = coop_slot(m, in.dst);
        const Fp12W &a = coop_slot(m, in.a);
        if (in.op == OP_MUL || in.op == OP_MUL_LINE0 || in.op == OP_MUL_LINE1 || in.op == OP_MUL_EVEN) {   // one body for every product
            const bool skip = (in.op == OP_MUL_LINE0 && !use1) || (in.op == OP_MUL_LINE1 && !use2) || (pc >= line_pc_end && (in.op == OP_MUL_LINE0 ||
                    in.op == OP_MUL_LINE1));
            const uint32_t mask = in.op == OP_MUL ? FULL_MASK : in.op == OP_MUL_EVEN ? EVEN_MASK : LINE_MASK;
            const Fp *bs = mask != LINE_MASK ? nullptr                                     // the line as its six evaluated coefficients:
                           : pre ? pre + ((in.op == OP_MUL_LINE1 ? N_LINES : 0) + cur_line) * 6      // all made ahead of the loop (two-wave kernel)
                                 // or five steps at a time (OP_LINE_EVAL)
                                 : coop_line_buf(m) + ((cur_line % COOP_LINE_CHUNK) * 2 + (in.op == OP_MUL_LINE1 ? 1 : 0)) * 6;
            if (!skip) coop_product(m, mask == LINE_MASK ? m.sc.line : m.sc.mul, dst, a, coop_slot(m, in.b), mask, bs);
            continue;
        }
        switch (in.op) {
            case OP_SET_ONE: coop_set_one(dst); break;
            case OP_SQR: coop_sqr(m, dst, a); break;
            case OP_CYC_SQR: coop_cyc_sqr(m, dst, a); break;
            case OP_LINE_EVAL: {
                const int n = in.a;
                cur_line = n;
                if (pre) break;                             // evaluated ahead of the loop: the line products read them in place
                if (n % COOP_LINE_CHUNK) break;             // this line is in the chunk evaluated COOP_LINE_CHUNK - 1 .. 1 steps ago
                // Evaluate the lines of the next five steps of BOTH pairs at their points in one go, scaled by Z^3: 60 products on 60 lanes,
                // ONE product body (operands picked per lane).  Line by line it was 12 lanes of 64 at work 68 times per check; now 14 times.
                // The evaluations park in the slots t0 .. t4, which the program only uses after the Miller loop ([step][pair][6 coefficients]).
                COOP_LANES(lane) {
                    const int c = lane / 12, q = (lane % 12) / 6, e = lane % 6;
                    if (c < COOP_LINE_CHUNK && n + c < N_LINES) {
                        const LineW &L = q == 0 ? lines1[n + c] : lines2[n + c];
                        const Fp *coef = &L.l0 + e;                                         // l0, l6, l2, l8, l3, l9
                        const Fp *arg = e < 2 ? &m.pz[q] : e < 4 ? &m.px[q] : &m.py[q];     // * Z^3, Z^3, X Z, X Z, Y, Y
                        fp_mul(coop_line_buf(m)[lane], *coef, *arg);                        // straight into LDS: no temporary on the private stack
                    }
                }
                COOP_SYNC();
                break;
            }
            case OP_CONJ: coop_conj(dst, a); break;
            case OP_FROB1: coop_frob(dst, a, ft.a1, ft.b1); break;
            case OP_FROB2: coop_frob2(dst, a, ft.a2); break;
            case OP_FP6INV: coop_fp6_inv(m, dst, m.t2.c); break;        // (t2, t3: free at this point of the program -- build_pairing_program)
            default: coop_copy(dst, a); break;
        }
    }
}
// All 2 x 68 line evaluations at once, ahead of the Miller loop (they depend on the points only; inside the loop each one is a
// dependent ~650-instruction step): item (q, n, e) = coefficient e of line n of pair q times its argument, items dealt to `nthreads`
// threads.  out: [2][N_LINES][6].
// pts: the two G1 arguments where they lie in memory (an address computed per lane into a by-value copy would put the copy on the private stack)
KZG_HD void coop_eval_lines_item(Fp *out, int item, const LineW *lines1, const LineW *lines2, const PairPt *pts) {
    const int q = item / (N_LINES * 6), n = (item / 6) % N_LINES, e = item % 6;
    const LineW &L = q == 0 ? lines1[n] : lines2[n];
    const PairPt &P = pts[q];
    const Fp *coef = &L.l0 + e;                                         // l0, l6, l2, l8, l3, l9
    const Fp *arg = e < 2 ? &P.az : e < 4 ? &P.ax : &P.ay;              // * Z^3, Z^3, X Z, X Z, Y, Y
    fp_mul(out[item], *coef, *arg);
}
// p1 / p2 = (0,0) (infinity) makes that pair contribute 1.
KZG_HD bool coop_pairing_check(CoopMem &m, const CoopInsn *prog, int n_insn, const CoopScheds *scheds, const LineW *lines1, const G1Affine &p1,
                               const LineW *lines2, const G1Affine &p2, const FrobTables &ft) {
    PairPt q1, q2; pairpt_from_affine(q1, p1); pairpt_from_affine(q2, p2);
    coop_init(m, scheds, q1, q2);
    coop_run(m, prog, 0, n_insn, lines1, lines2, !g1a_is_inf(p1), !g1a_is_inf(p2), ft);
    return coop_is_one(m, m.t0);
}

}  // namespace kzg
