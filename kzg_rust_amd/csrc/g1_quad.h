// g1_quad.h -- G1 doubling and addition walked by FOUR lanes (a DPP quad) per point: the latency forms of the verify path's small calls
// (k_ps_shift, k_ps_buckets, k_ps_weights in k_g1.hip).  Device only.  Unit test on the GPU: tests/native/quad_ops_test.hip.
#pragma once
#include "g1.h"

namespace kzg {

// The chain itself is walked by FOUR lanes per point (a DPP quad).  A lone wave issues one instruction per ~5 cycles whatever it is,
// so a doubling costs its instruction count: 5 squarings + 2 products in a row on one lane (dbl-2009-l).  Its products come in three
// dependent stages, and within a stage they are independent:
//     stage 1:  A = X^2,  B = Y^2,  S = (Y + Z)^2,  ZZ = Z^2          (four lanes, one squaring each;  Z3 = S - B - ZZ = 2 Y Z)
//     stage 2:  F = (3A)^2,  C = B^2,  G = (X + B)^2                   (three lanes;  D = 2 (G - A - C),  X3 = F - 2D)
//     stage 3:  E (D - X3)                                             (every lane for itself;  Y3 = E (D - X3) - 8C)
// so the quad runs ONE squaring body per stage, each lane on its own operand (picked by v_cndmask), and the results are broadcast with
// quad_perm DPP moves: 2 squarings + 1 product deep instead of 5 + 2.  All four lanes carry the same (X, Y, Z) and do the cheap linear
// steps redundantly.  Lazy bounds as in g1_dbl_lazy (g1.h): in X, Y, Z < 32p, out X < 26p, Y < 18p, Z < 6p.
template <int K> __device__ __forceinline__ Fp fp_quad_bcast(const Fp &v) {
    Fp r;
#pragma unroll
    for (int i = 0; i < NFP; i++) {
        uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55 /* quad_perm: [K, K, K, K] */, 0xf, 0xf, false);
        // keep the move a move: hipcc 7.2's DPP-combine pass folded these broadcasts into the subtractions that consume them
        // (v_subrev_u32_dpp ...) and lane 0 of every quad came out with a wrong Y3 in g1_add_quad (tests/native/quad_ops_test.hip)
        asm volatile("" : "+v"(x));
        r.l[i] = x;
    }
    return r;
}
__device__ __forceinline__ void fp_mul3_lz(Fp &r, const Fp &a) {        // 3a, limbs normalised
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) { const uint32_t t = 3u * a.l[i] + c; if (i < NFP - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t; }
}
__device__ __forceinline__ void fp_mul8_lz(Fp &r, const Fp &a) {        // 8a, limbs normalised (a's limbs below the top one < 2^29)
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NFP; i++) { const uint32_t t = (a.l[i] << 3) + c; if (i < NFP - 1) { c = t >> LB; r.l[i] = t & LMASK; } else r.l[i] = t; }
}
__device__ __forceinline__ void g1_dbl_quad(G1Jac &p, int role) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT, m16[NFP] = FP_MOD16_INIT, m32[NFP] = FP_MOD32_INIT;
    Fp u, r, t, v;
    fp_add_lz(t, p.y, p.z);                                       // Y + Z                           < 64p
    fp_select(u, role == 1, p.x, p.y); fp_select(u, role == 2, u, t); fp_select(u, role == 3, u, p.z);
    fp_sqr_lz(r, u);
    const Fp A = fp_quad_bcast<0>(r), B = fp_quad_bcast<1>(r), S = fp_quad_bcast<2>(r), ZZ = fp_quad_bcast<3>(r);
    Fp Z3; fp_sub_lz(t, S, B, m2); fp_sub_lz(Z3, t, ZZ, m2);     // S - B - ZZ + 4p                 in (0, 6p)
    Fp E; fp_mul3_lz(E, A);                                       // E = 3A                          < 4p
    fp_add_lz(t, p.x, B);                                         // X + B                           < 34p
    fp_select(u, role == 1, E, B); fp_select(u, role == 2, u, t);
    fp_sqr_lz(r, u);
    const Fp F = fp_quad_bcast<0>(r), C = fp_quad_bcast<1>(r), G = fp_quad_bcast<2>(r);
    fp_sub_lz(t, G, A, m2); fp_sub_lz(v, t, C, m2);               // G - A - C + 4p                  in (0, 6p)
    Fp D; fp_add_lz(D, v, v);                                     // D                               < 12p
    Fp X3; fp_sub_lz(t, F, D, m16); fp_sub_lz(X3, t, D, m8);      // F - 2D + 24p                    in (0, 26p)
    fp_sub_lz(t, D, X3, m32);                                     // D - X3 + 32p                    in (6p, 44p)
    Fp Y3; fp_mul_lz(Y3, E, t);
    fp_mul8_lz(v, C);                                             // 8C                              < 9p
    fp_sub_lz(p.y, Y3, v, m16);                                   //                                 in (0, 18p)
    p.x = X3; p.z = Z3;
}
// Jacobian + Jacobian addition by the same quad: its 16 products come in five dependent stages of <= 4 independent ones,
//     1:  Z1Z1 = Z1^2,  Z2Z2 = Z2^2,  T1 = Y1 Z2,  T2 = Y2 Z1          2:  U1 = X1 Z2Z2,  U2 = X2 Z1Z1,  S1 = T1 Z2Z2,  S2 = T2 Z1Z1
//     3:  HH = H^2,  RR = R^2,  ZZ = Z1 Z2        4:  HHH = H HH,  V = U1 HH,  Z3 = ZZ H        5:  R (V - X3),  S1 HHH
// so an addition is five products deep instead of sixteen.  Lazy bounds as g1_add_lazy2 (g1.h): a, b with X, Y, Z < 32p; out X < 8p,
// Y < 4p, Z < 2p.  An operand at infinity or P = +-Q (exact low-limb filters) goes through the canonical complete addition, on all four
// lanes alike, AFTER the cooperative stages (the first form returned early from inside them and gave a wrong Y3 on lane 0 of every quad
// with hipcc 7.2, although the same stages without the branch were right: tests/native/quad_ops_test.hip).
__device__ __forceinline__ void g1_add_quad(G1Jac &r, const G1Jac &a, const G1Jac &b, int role) {
    const uint32_t m2[NFP] = FP_MOD2_INIT, m8[NFP] = FP_MOD8_INIT;
    Fp u, v, w, t;
    fp_select(u, role == 1, a.z, b.z); fp_select(u, role == 2, u, a.y); fp_select(u, role == 3, u, b.y);
    fp_select(v, role == 1 || role == 2, a.z, b.z);
    fp_mul_lz(w, u, v);
    const Fp Z1Z1 = fp_quad_bcast<0>(w), Z2Z2 = fp_quad_bcast<1>(w), T1 = fp_quad_bcast<2>(w), T2 = fp_quad_bcast<3>(w);
    fp_select(u, role == 1, a.x, b.x); fp_select(u, role == 2, u, T1); fp_select(u, role == 3, u, T2);
    fp_select(v, role == 1 || role == 3, Z2Z2, Z1Z1);
    fp_mul_lz(w, u, v);
    const Fp U1 = fp_quad_bcast<0>(w), U2 = fp_quad_bcast<1>(w), S1 = fp_quad_bcast<2>(w), S2 = fp_quad_bcast<3>(w);
    Fp H, R;
    fp_sub_lz(H, U2, U1, m2);                                     // in (0, 4p)
    fp_sub_lz(R, S2, S1, m2);
    // rare: an operand at infinity or P = +-Q.  The cooperative stages below still run (on values that are then discarded): no DPP move
    // ever executes under a divergent branch, and the complete canonical addition replaces the result at the end.
    const bool rare = fp_maybe_zero_lz(H) || fp_maybe_zero_lz(a.z) || fp_maybe_zero_lz(b.z);
    G1Jac ca, cb;
    if (rare) { g1_canon_lazy(ca, a); g1_canon_lazy(cb, b); }    // (before r is written: r may alias a or b)
    fp_select(u, role == 1, H, R); fp_select(u, role >= 2, u, a.z);
    fp_select(v, role == 1, H, R); fp_select(v, role >= 2, v, b.z);
    fp_mul_lz(w, u, v);
    const Fp HH = fp_quad_bcast<0>(w), RR = fp_quad_bcast<1>(w), ZZ = fp_quad_bcast<2>(w);
    fp_select(u, role == 1, H, U1); fp_select(u, role >= 2, u, ZZ);
    fp_select(v, role >= 2, HH, H);
    fp_mul_lz(w, u, v);
    const Fp HHH = fp_quad_bcast<0>(w), V = fp_quad_bcast<1>(w), Z3 = fp_quad_bcast<2>(w);
    Fp X3;
    fp_sub_lz(t, RR, HHH, m2); fp_sub_lz(u, t, V, m2); fp_sub_lz(X3, u, V, m2);        // in (0, 8p)
    fp_sub_lz(t, V, X3, m8);                                                            // in (0, 10p)
    fp_select(u, role == 0, S1, R);
    fp_select(v, role == 0, HHH, t);
    fp_mul_lz(w, u, v);
    const Fp A = fp_quad_bcast<0>(w), B = fp_quad_bcast<1>(w);
    fp_sub_lz(r.y, A, B, m2);                                                           // in (0, 4p)
    r.x = X3; r.z = Z3;
    if (rare) g1_add(r, ca, cb);
}
__device__ __forceinline__ G1Jac g1_shfl_down_w(const G1Jac &v, int delta) {
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_down(v.x.l[i], delta, 64); r.y.l[i] = __shfl_down(v.y.l[i], delta, 64); r.z.l[i] = __shfl_down(v.z.l[i],
            delta, 64); }
    return r;
}

}  // namespace kzg
