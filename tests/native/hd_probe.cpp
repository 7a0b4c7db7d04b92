// Host build (g++) of the DEVICE math headers under kzg_rust_amd/csrc, exposed through a tiny C ABI so
// that tests/test_device_math_host.py can compare them with the CPU oracle on the build box (no GPU).
// Test infrastructure only: this file is never part of libkzg355.so.
// the fixed-base MSM's form of the lazy mixed addition (two products under one reduction); the other form is the batch linear combination's, checked
// on the GPU
#define KZG_G1_ADD_MUL2 1
#include "../../kzg_rust_amd/csrc/field.h"
#include "../../kzg_rust_amd/csrc/tower.h"
#include "../../kzg_rust_amd/csrc/g1.h"
#include "../../kzg_rust_amd/csrc/pairing.h"
#include "../../kzg_rust_amd/csrc/sha256.h"
#include "../../kzg_rust_amd/csrc/pairing_coop.h"
#include "../../kzg_rust_amd/csrc/pairing_lanes.h"
#include "../../kzg_rust_amd/csrc/eval_core.h"
#include "../../kzg_rust_amd/csrc/quot_core.h"
#include <vector>
#include <cstring>
using namespace kzg;
extern "C" {
int hd_fp_op(int op, uint8_t *out, const uint8_t *a, const uint8_t *b) {
    Fp x, y, r;
    if (!fp_from_be48(x, a, false) || !fp_from_be48(y, b, false)) return 1;
    switch (op) {
        case 0: fp_add(r, x, y); break; case 1: fp_sub(r, x, y); break; case 2: fp_mul(r, x, y); break;
        case 3: fp_inv_fermat(r, x); break; case 4: if (!fp_sqrt(r, x)) return 2; break;
        case 5: fp_neg(r, x); break; case 6: fp_dbl(r, x); break;
        case 7: { out[0] = fp_is_lex_largest(x); return 0; }
        case 8: fp_inv(r, x); break;
        case 9: fp_sqr(r, x); break;
        default: return 1;
    }
    fp_to_be48(out, r); return 0;
}
int hd_fr_op(int op, uint8_t *out, const uint8_t *a, const uint8_t *b) {
    Fr x, y, r; uint32_t w[8];
    be32_to_words(w, a); fr_from_words(x, w); be32_to_words(w, b); fr_from_words(y, w);
    switch (op) {
        case 0: fr_add(r, x, y); break; case 1: fr_sub(r, x, y); break; case 2: fr_mul(r, x, y); break;
        case 3: fr_inv_fermat(r, x); break;
        case 4: { be32_to_words(w, a); out[0] = fr_words_canonical(w); return 0; }
        case 5: fr_inv(r, x); break;
        case 6: fr_sqr(r, x); break;
        default: return 1;
    }
    fr_to_be32(out, r); return 0;
}
// out = a*b + c*d mod r through the lazy two-product reduction, then canonicalised by one ordinary product with 1
int hd_fr_mul2(uint8_t *out, const uint8_t *a, const uint8_t *b, const uint8_t *c, const uint8_t *d) {
    Fr x[4], r; uint32_t w[8]; const uint8_t *in[4] = {a, b, c, d};
    for (int i = 0; i < 4; i++) { be32_to_words(w, in[i]); fr_from_words(x[i], w); }
    fr_mul2_lazy(r, x[0], x[1], x[2], x[3]);
    Fr l; fr_mul_lazy(l, r, fr_one());          // a lazy value fed to a lazy product ...
    fr_mul(r, l, fr_one());                     // ... and the canonical product that ends the chain
    fr_to_be32(out, r); return 0;
}
// y = p(z) through eval_core.h's binary tree, level of groups by level of groups over arrays (the device deals the same groups to the lanes of a
// wave and moves the children through LDS; the node arithmetic and its lazy bounds are this code; the sum of the values is folded per "lane" of 64
// values as on the device).  blob: 4096 x 32 big-endian bytes, z: 32 bytes.
int hd_eval_poly(uint8_t *out32, const uint8_t *blob, const uint8_t *z_be) {
    constexpr int N_FE = 4096;
    static Fr roots[N_FE]; static EvalGroup tab[EVAL_TAB_GROUPS]; static bool ready = false;
    if (!ready) {
        const uint32_t rootc[NFR] = FR_ROOT4096_INIT;
        Fr base; for (int k = 0; k < NFR; k++) base.l[k] = rootc[k];
        Fr acc = fr_one();
        for (int i = 0; i < N_FE; i++) {
            uint32_t rev = 0; for (int b = 0; b < 12; b++) rev |= ((i >> b) & 1u) << (11 - b);
            roots[rev] = acc; fr_mul(acc, acc, base);
        }
        for (int level = 1; level <= 6; level++)
            for (int m = 0; m < (N_FE >> (2 * level)); m++) tab[eval_tab_first(level) + m] = EvalGroup{roots[4 * m], roots[4 * m + 2], roots[2 * m]};
        ready = true;
    }
    uint32_t w[8]; be32_to_words(w, z_be);
    Fr zp[13]; fr_from_words(zp[0], w);
    for (int k = 1; k <= 12; k++) fr_sqr(zp[k], zp[k - 1]);                               // z^(2^k)
    static Fr h[6][1024];
    Fr lane_sum[64], Sp = fr_zero();
    for (int l = 0; l < 64; l++) lane_sum[l] = fr_zero();
    for (int g = 0; g < 1024; g++) {
        uint32_t pw[4][8];
        for (int e = 0; e < 4; e++) { be32_to_words(pw[e], blob + 32 * (4 * g + e)); if (!fr_words_canonical(pw[e])) return 1; }
        eval_group_leaves(h[0][g], lane_sum[g & 63], pw, zp[0], zp[1], tab[EVAL_TAB_L1 + g]);      // lane g & 63 takes group g in step g >> 6
    }
    for (int l = 0; l < 64; l++) { eval_fold(lane_sum[l]); fr_add_lazy(Sp, Sp, lane_sum[l]); }
    for (int level = 2; level <= 6; level++)
        for (int m = 0; m < (N_FE >> (2 * level)); m++)
            eval_group(h[level - 1][m], h[level - 2] + 4 * m, zp[2 * level - 2], zp[2 * level - 1], tab[eval_tab_first(level) + m]);
    Fr y; eval_finish(y, h[5][0], Sp, zp[0], zp[12]);
    limbs_to_words<NFR, 8>(w, y.l);
    for (int i = 0; i < 8; i++) { const uint32_t v = w[7 - i]; out32[4 * i] = v >> 24; out32[4 * i + 1] = v >> 16; out32[4 * i + 2] = v >> 8;
            out32[4 * i + 3] = v; }
    return 0;
}
// y and the quotient q_i = (p_i - y) / (w_i - z) through quot_core.h, the way k_quotient_tree<lg> deals the tree to its lanes: 4096 >> lg "lanes", each
// with the path from the root to its node, the subtree below it in groups of four leaves, pass 1 / the sums (folded per lane, per wave of 64 lanes,
// per blob) / pass 2 with the inverses parked as 8 words in between.  out_q: 4096 x 32 big-endian bytes; out_y: 32.  2: z inside the domain
// (the device hands those blobs to k_quotient_scan).
int hd_quotient(uint8_t *out_q, uint8_t *out_y, const uint8_t *blob, const uint8_t *z_be, int lg) {
    constexpr int N_FE = 4096;
    if (lg != 2 && lg != 4 && lg != 6) return 3;
    static Fr roots[N_FE]; static bool ready = false;
    if (!ready) {
        const uint32_t rootc[NFR] = FR_ROOT4096_INIT;
        Fr base; for (int k = 0; k < NFR; k++) base.l[k] = rootc[k];
        Fr acc = fr_one();
        for (int i = 0; i < N_FE; i++) {
            uint32_t rev = 0; for (int b = 0; b < 12; b++) rev |= ((i >> b) & 1u) << (11 - b);
            roots[rev] = acc; fr_mul(acc, acc, base);
        }
        ready = true;
    }
    uint32_t w[8]; be32_to_words(w, z_be);
    Fr z; fr_from_words(z, w);
    QuotPrep pp;
    if (quot_prep(pp, z)) return 2;
    const int lanes = N_FE >> lg, L = lg - 2, D0 = 12 - lg, groups = 1 << L;
    std::vector<uint32_t> stash(8 * N_FE);
    std::vector<Fr> su(lanes), sp(lanes);
    for (int t = 0; t < lanes; t++) {
        Fr inv0; quot_path(inv0, pp, roots, D0, t);
        Fr Su = fr_zero(), Sp = fr_zero();
        Fr c[5][2];
        for (int g = 0; g < groups; g++) {
            for (int l = 0; l < L; l++)
                if ((g & ((1 << (L - l)) - 1)) == 0) {
                    const Fr src = l == 0 ? inv0 : c[l - 1][(g >> (L - l)) & 1];
                    const int a = (t << l) + (g >> (L - l));
                    quot_children(c[l][0], c[l][1], src, pp.zsq[11 - (D0 + l)], roots[2 * a]);
                }
            const Fr inv10 = L == 0 ? inv0 : c[L - 1][g & 1];
            const int a10 = (t << L) + g;
            uint32_t pw[4][8];
            for (int j = 0; j < 4; j++) { be32_to_words(pw[j], blob + 32 * (4 * a10 + j)); if (!fr_words_canonical(pw[j])) return 1; }
            Fr inv12[4];
            quot_group_pass1(inv12, Su, Sp, pw, inv10, a10, pp, roots);
            for (int j = 0; j < 4; j++) limbs_to_words<NFR, 8>(&stash[8 * (4 * a10 + j)], inv12[j].l);
        }
        quot_fold(Su); quot_fold(Sp);
        su[t] = Su; sp[t] = Sp;
    }
    Fr Su = fr_zero(), Sp = fr_zero();
    for (int w0 = 0; w0 < lanes; w0 += 64) {                       // a wave's 64 lanes, then the fold, then the waves
        Fr a = fr_zero(), b = fr_zero();
        for (int t = w0; t < w0 + 64 && t < lanes; t++) { fr_add_lazy(a, a, su[t]); fr_add_lazy(b, b, sp[t]); }
        quot_fold(a); quot_fold(b);
        fr_add_lazy(Su, Su, a); fr_add_lazy(Sp, Sp, b);
    }
    Fr y; quot_y(y, Su, Sp, pp);
    limbs_to_words<NFR, 8>(w, y.l); words_to_be32(out_y, w);
    for (int e = 0; e < N_FE; e++) {
        uint32_t pw[8], qw[8];
        be32_to_words(pw, blob + 32 * e);
        quot_leaf_pass2(qw, pw, &stash[8 * e], y);
        words_to_be32(out_q + 32 * e, qw);
    }
    return 0;
}
// 0 ok / 1 bad encoding / 2 not on curve / 3 not in subgroup ; out = recompressed point
int hd_g1_validate(uint8_t *out, const uint8_t *in, int check_subgroup) {
    G1Affine p; int rc = g1_decompress(p, in);
    if (rc) return rc;
    if (check_subgroup && !g1a_is_inf(p)) {
        const bool a = g1_in_subgroup(p), b = g1_in_subgroup_naive(p);
        if (a != b) return 99;          // endomorphism test and [r]P test must agree
        if (!a) return 3;
    }
    g1_compress_affine(out, p); return 0;
}
// out = [k]P + Q (Q optional), compressed; k = 32 big-endian bytes
int hd_g1_mul_add(uint8_t *out, const uint8_t *p, const uint8_t *k_be, const uint8_t *q) {
    G1Affine pa, qa, ra; G1Jac r; uint32_t w[8];
    if (g1_decompress(pa, p)) return 1;
    be32_to_words(w, k_be);
    g1_mul_words(r, pa, w, 8);
    if (q) { if (g1_decompress(qa, q)) return 1; g1_add_mixed(r, r, qa); }
    g1_to_affine(ra, r); g1_compress_affine(out, ra); return 0;
}
// out = sum of n compressed points accumulated in the extended-Jacobian (XYZZ) form the MSM / bucket kernels use
int hd_g1x_sum(uint8_t *out, const uint8_t *pts48, int n) {
    G1X acc = g1x_inf();
    for (int i = 0; i < n; i++) { G1Affine p; if (g1_decompress(p, pts48 + 48 * i)) return 1; g1x_add_mixed(acc, acc, p); }
    G1Jac j; g1x_to_jac(j, acc);
    G1Affine a; g1_to_affine(a, j); g1_compress_affine(out, a); return 0;
}
// the same through the lazy (unreduced) addition of the accumulation loops
int hd_g1x_sum_lazy(uint8_t *out, const uint8_t *pts48, int n) {
    G1X acc = g1x_inf(); bool started = false;
    for (int i = 0; i < n; i++) { G1Affine p; if (g1_decompress(p, pts48 + 48 * i)) return 1; g1x_add_mixed_lazy(acc, started, p); }
    G1X c; g1x_from_lazy(c, acc, started);
    G1Jac j; g1x_to_jac(j, c);
    G1Affine a; g1_to_affine(a, j); g1_compress_affine(out, a); return 0;
}
// Horner sum_i 16^i P_i (i = n-1 .. 0) with the lazy doubling / addition chain (lazy = 1) or the canonical one; the points
// enter as Jacobian with a non-trivial z (doubled once and halved back is overkill: lifted by z = x-coordinate of P_0)
int hd_horner(uint8_t *out, const uint8_t *pts48, int n, int lazy) {
    std::vector<G1Jac> v(n);
    for (int i = 0; i < n; i++) {
        G1Affine p; if (g1_decompress(p, pts48 + 48 * i)) return 1;
        g1_from_affine(v[i], p);
        if (!g1a_is_inf(p)) {                                  // (x, y, 1) -> (x l^2, y l^3, l), l = 5 + i
            Fp l = fp_one(), t; for (int k = 0; k < 4 + i; k++) fp_add(l, l, fp_one());
            fp_sqr(t, l); fp_mul(v[i].x, v[i].x, t); fp_mul(t, t, l); fp_mul(v[i].y, v[i].y, t); v[i].z = l;
        }
    }
    G1Jac acc = v[n - 1];
    for (int w = n - 2; w >= 0; w--) {
        if (lazy) { for (int k = 0; k < 4; k++) g1_dbl_lazy(acc, acc); g1_add_lazy(acc, acc, v[w]); }
        else { for (int k = 0; k < 4; k++) g1_dbl(acc, acc); g1_add(acc, acc, v[w]); }
    }
    if (lazy) g1_canon_lazy(acc, acc);
    G1Affine a; g1_to_affine(a, acc); g1_compress_affine(out, a); return 0;
}
// out = P + Q using the Jacobian+Jacobian routine (both lifted with a non-trivial z)
int hd_g1_add_jac(uint8_t *out, const uint8_t *p, const uint8_t *q) {
    G1Affine pa, qa, ra; G1Jac pj, qj, r;
    if (g1_decompress(pa, p) || g1_decompress(qa, q)) return 1;
    g1_from_affine(pj, pa); g1_from_affine(qj, qa);
    // re-randomise z: (x z^2, y z^3, z) with z = 5 (Montgomery form of some element: use pa.x+1 if nonzero)
    Fp z = fp_one(); fp_add(z, z, z); fp_add(z, z, fp_one());
    if (!g1_is_inf(pj)) { Fp z2, z3; fp_sqr(z2, z); fp_mul(z3, z2, z); fp_mul(pj.x, pj.x, z2); fp_mul(pj.y, pj.y, z3); pj.z = z; }
    g1_add(r, pj, qj);
    g1_to_affine(ra, r); g1_compress_affine(out, ra); return 0;
}
// [k]P through the GLV split: [k mod x^2]P + [k div x^2](-phi P)
// both GLV splits of a 256-bit big-endian k: out = a (16 bytes, little-endian words) | b | a_fast | b_fast
int hd_glv_splits(uint8_t *out, const uint8_t *k_be) {
    uint32_t k[8], a[4], b[4], af[4], bf[4];
    be32_to_words(k, k_be);
    glv_split(a, b, k);
    glv_split_fast(af, bf, k);
    memcpy(out, a, 16); memcpy(out + 16, b, 16); memcpy(out + 32, af, 16); memcpy(out + 48, bf, 16);
    return 0;
}
int hd_glv_mul(uint8_t *out, const uint8_t *p, const uint8_t *k_be) {
    G1Affine pa, qa, ra; G1Jac r1, r2; uint32_t w[8], a[4], b[4];
    if (g1_decompress(pa, p)) return 1;
    be32_to_words(w, k_be);
    glv_split(a, b, w);
    g1a_neg_phi(qa, pa);
    g1_mul_words(r1, pa, a, 4); g1_mul_words(r2, qa, b, 4);
    g1_add(r1, r1, r2);
    g1_to_affine(ra, r1); g1_compress_affine(out, ra); return 0;
}
// [k]P for a 128-bit k through the signed 4-bit window routine used by k_lincomb_terms (lane 5 of a 64-lane table)
int hd_w4_mul(uint8_t *out, const uint8_t *p, const uint8_t *k_be16) {
    G1Affine pa, ra; G1Jac r; uint32_t k[4];
    if (g1_decompress(pa, p)) return 1;
    for (int i = 0; i < 4; i++) k[i] = ((uint32_t)k_be16[4 * (3 - i)] << 24) | ((uint32_t)k_be16[4 * (3 - i) + 1] << 16) | ((uint32_t)k_be16[4 * (3 - i) +
            2] << 8) | k_be16[4 * (3 - i) + 3];
    std::vector<uint32_t> tab(W4_ENTRIES * 3 * NFP * 64);
    g1_mul128_w4(r, pa, k, tab.data(), 5);
    g1_to_affine(ra, r); g1_compress_affine(out, ra); return 0;
}
// [2^k]P computed as k_ps_shift does -- the doubling chain started from x ALONE on the curve Y^2 = X^3 + 4 s^3 (s = x^3 + 4), from
// (s x, s^2, 1), with the y of the decompressed point multiplied into Z only at the end -- against the plain chain from (x, y, 1).
// out = the first, ref = the second, both compressed.
int hd_shift_from_x(uint8_t *out, uint8_t *ref, const uint8_t *p48, int k) {
    Fp x, s; bool inf, large;
    if (g1_parse_compressed(x, inf, large, p48)) return 1;
    G1Affine pa; if (g1_decompress(pa, p48)) return 2;
    G1Jac a;
    g1_curve_rhs(s, x);
    fp_mul(a.x, s, x); fp_sqr(a.y, s); a.z = fp_one();
    if (inf) a = g1_inf();
    for (int i = 0; i < k; i++) g1_dbl_lazy(a, a);
    g1_canon_lazy(a, a);
    Fp zz; fp_mul(zz, a.z, pa.y); a.z = zz;
    G1Jac b; g1_from_affine(b, pa);
    for (int i = 0; i < k; i++) g1_dbl(b, b);
    G1Affine ra, rb; g1_to_affine(ra, a); g1_to_affine(rb, b);
    g1_compress_affine(out, ra); g1_compress_affine(ref, rb);
    return 0;
}
int hd_g2_decompress(const uint8_t *in) { G2Affine q; return g2_decompress(q, in); }
// e(p1,q1) == e(p2,q2) via precomputed lines: ML(q1,-p1) * ML(q2,p2)
int hd_pairings_verify(int *ok, const uint8_t *p1, const uint8_t *q1, const uint8_t *p2, const uint8_t *q2) {
    G1Affine a, b; G2Affine qa, qb;
    if (g1_decompress(a, p1) || g1_decompress(b, p2) || g2_decompress(qa, q1) || g2_decompress(qb, q2)) return 1;
    std::vector<LineCoeff> l1(N_LINES), l2(N_LINES);
    Fp12 f;
    G1Affine an; g1a_neg(an, a); if (g1a_is_inf(a)) an = a;
    if (g2a_is_inf(qa)) an = g1a_inf(); else precompute_lines(l1.data(), qa);
    if (g2a_is_inf(qb)) b = g1a_inf(); else precompute_lines(l2.data(), qb);
    miller_loop_pair(f, l1.data(), an, l2.data(), b);
    *ok = final_exp_is_one(f) ? 1 : 0;
    return 0;
}
// same check through the wave-cooperative pairing (host emulation of the 64 lanes)
int hd_pairings_verify_coop(int *ok, const uint8_t *p1, const uint8_t *q1, const uint8_t *p2, const uint8_t *q2) {
    G1Affine a, b; G2Affine qa, qb;
    if (g1_decompress(a, p1) || g1_decompress(b, p2) || g2_decompress(qa, q1) || g2_decompress(qb, q2)) return 1;
    std::vector<LineCoeff> l1(N_LINES), l2(N_LINES);
    std::vector<LineW> w1(N_LINES), w2(N_LINES);
    G1Affine an; g1a_neg(an, a); if (g1a_is_inf(a)) an = a;
    if (g2a_is_inf(qa)) an = g1a_inf(); else precompute_lines(l1.data(), qa);
    if (g2a_is_inf(qb)) b = g1a_inf(); else precompute_lines(l2.data(), qb);
    for (int i = 0; i < N_LINES; i++) { line_to_w(w1[i], l1[i]); line_to_w(w2[i], l2[i]); }
    static const uint32_t A1[12][NFP] = FROBW_A1_INIT, B1[12][NFP] = FROBW_B1_INIT, A2[12][NFP] = FROBW_A2_INIT;
    FrobTables ft;
    for (int k = 0; k < 12; k++) for (int i = 0; i < NFP; i++) { ft.a1[k].l[i] = A1[k][i]; ft.b1[k].l[i] = B1[k][i]; ft.a2[k].l[i] = A2[k][i]; }
    CoopMem *mem = new CoopMem();
    static CoopInsn prog[COOP_PROGRAM_MAX];
    const int n_insn = build_pairing_program(prog);
    if (n_insn > COOP_PROGRAM_MAX) return 2;
    static CoopScheds sc;
    if (!build_coop_schedules(sc)) return 3;
    *ok = coop_pairing_check(*mem, prog, n_insn, &sc, w1.data(), an, w2.data(), b, ft) ? 1 : 0;
    delete mem;
    return 0;
}
// The same check the way the kernels run it since the G1 arguments became projective: the points lifted to Jacobian coordinates with
// z = zsel + 2, turned into (X Z, Y, Z^3) (pairpt_from_jac), the Miller loops as two separate runs of the program's prefix (one pair
// each, as the two waves of k_pairing_coop2 do), one product, then the rest of the program.
int hd_pairings_verify_coop_proj(int *ok, const uint8_t *p1, const uint8_t *q1, const uint8_t *p2, const uint8_t *q2, int zsel) {
    G1Affine a, b; G2Affine qa, qb;
    if (g1_decompress(a, p1) || g1_decompress(b, p2) || g2_decompress(qa, q1) || g2_decompress(qb, q2)) return 1;
    std::vector<LineCoeff> l1(N_LINES), l2(N_LINES);
    std::vector<LineW> w1(N_LINES), w2(N_LINES);
    if (!g2a_is_inf(qa)) precompute_lines(l1.data(), qa);
    if (!g2a_is_inf(qb)) precompute_lines(l2.data(), qb);
    for (int i = 0; i < N_LINES; i++) { line_to_w(w1[i], l1[i]); line_to_w(w2[i], l2[i]); }
    auto lift = [&](PairPt &o, const G1Affine &p, bool negate) {
        G1Jac j; g1_from_affine(j, p);
        if (!g1_is_inf(j)) {
            Fp z = fp_one(); for (int k = 0; k < zsel + 1; k++) fp_add(z, z, fp_one());
            Fp z2, z3; fp_sqr(z2, z); fp_mul(z3, z2, z); fp_mul(j.x, j.x, z2); fp_mul(j.y, j.y, z3); j.z = z;
        }
        pairpt_from_jac(o, j, negate);
    };
    PairPt pa, pb; lift(pa, a, true); lift(pb, b, false);
    const bool use1 = !fp_is_zero(pa.az) && !g2a_is_inf(qa), use2 = !fp_is_zero(pb.az) && !g2a_is_inf(qb);
    static const uint32_t A1[12][NFP] = FROBW_A1_INIT, B1[12][NFP] = FROBW_B1_INIT, A2[12][NFP] = FROBW_A2_INIT;
    FrobTables ft;
    for (int k = 0; k < 12; k++) for (int i = 0; i < NFP; i++) { ft.a1[k].l[i] = A1[k][i]; ft.b1[k].l[i] = B1[k][i]; ft.a2[k].l[i] = A2[k][i]; }
    CoopMem *m0 = new CoopMem(), *m1 = new CoopMem();
    static CoopInsn prog[COOP_PROGRAM_MAX];
    const int n_insn = build_pairing_program(prog);
    if (n_insn > COOP_PROGRAM_MAX) return 2;
    static CoopScheds sc;
    if (!build_coop_schedules(sc)) return 3;
    coop_init(*m0, &sc, pa, pb); coop_init(*m1, &sc, pa, pb);
    // the line evaluations made ahead of the loops, as k_pairing_coop2 does (odd zsel: inside the loop, the single-wave kernel's way)
    std::vector<Fp> pre(2 * N_LINES * 6);
    const PairPt pab[2] = {pa, pb};
    for (int item = 0; item < 2 * N_LINES * 6; item++) coop_eval_lines_item(pre.data(), item, w1.data(), w2.data(), pab);
    const Fp *prep = (zsel & 1) ? nullptr : pre.data();
    coop_run(*m0, prog, 0, COOP_MILLER_INSNS, w1.data(), w2.data(), use1, false, ft, prep);
    coop_run(*m1, prog, 0, COOP_MILLER_INSNS, w1.data(), w2.data(), false, use2, ft, prep);
    for (int k = 0; k < 12; k++) m0->t0.c[k] = m1->f.c[k];
    coop_product(*m0, m0->sc.mul, m0->f, m0->f, m0->t0, FULL_MASK);
    coop_run(*m0, prog, COOP_MILLER_INSNS, n_insn, w1.data(), w2.data(), false, false, ft);
    *ok = coop_is_one(*m0, m0->t0) ? 1 : 0;
    delete m0; delete m1;
    return 0;
}
// The Miller loops in K segments per pair on 2 K "waves" (k_pairing_coop_split: miller_split's boundaries, every wave from f = 1 at its segment's first
// iteration, lines only inside the segment, squarings to the end of the loop), the tree of products over the partial values, the rest of the program.
int hd_pairings_verify_coop_segments(int *ok, const uint8_t *p1, const uint8_t *q1, const uint8_t *p2, const uint8_t *q2, int K) {
    G1Affine a, b; G2Affine qa, qb;
    if (K < 1 || K > MILLER_SPLIT_MAX) return 4;
    if (g1_decompress(a, p1) || g1_decompress(b, p2) || g2_decompress(qa, q1) || g2_decompress(qb, q2)) return 1;
    std::vector<LineCoeff> l1(N_LINES), l2(N_LINES);
    std::vector<LineW> w1(N_LINES), w2(N_LINES);
    if (!g2a_is_inf(qa)) precompute_lines(l1.data(), qa);
    if (!g2a_is_inf(qb)) precompute_lines(l2.data(), qb);
    for (int i = 0; i < N_LINES; i++) { line_to_w(w1[i], l1[i]); line_to_w(w2[i], l2[i]); }
    G1Affine an = a; if (!g1a_is_inf(an)) fp_neg(an.y, an.y);
    PairPt pa, pb; pairpt_from_affine(pa, an); pairpt_from_affine(pb, b);
    const bool use1 = !fp_is_zero(pa.az) && !g2a_is_inf(qa), use2 = !fp_is_zero(pb.az) && !g2a_is_inf(qb);
    static const uint32_t A1[12][NFP] = FROBW_A1_INIT, B1[12][NFP] = FROBW_B1_INIT, A2[12][NFP] = FROBW_A2_INIT;
    FrobTables ft;
    for (int k = 0; k < 12; k++) for (int i = 0; i < NFP; i++) { ft.a1[k].l[i] = A1[k][i]; ft.b1[k].l[i] = B1[k][i]; ft.a2[k].l[i] = A2[k][i]; }
    static CoopInsn prog[COOP_PROGRAM_MAX];
    const int n_insn = build_pairing_program(prog);
    if (n_insn > COOP_PROGRAM_MAX) return 2;
    static CoopScheds sc;
    if (!build_coop_schedules(sc)) return 3;
    const MillerSplit sp = miller_split(prog, K);
    // every iteration must belong to exactly one segment
    for (int j = 0; j + 1 < K; j++) if (sp.pc_lines_end[j] != sp.pc_start[j + 1]) return 5;
    if (sp.pc_start[0] != 0 || sp.pc_lines_end[K - 1] < COOP_MILLER_INSNS) return 6;
    std::vector<Fp> pre(2 * N_LINES * 6);
    const PairPt pab[2] = {pa, pb};
    for (int item = 0; item < 2 * N_LINES * 6; item++) coop_eval_lines_item(pre.data(), item, w1.data(), w2.data(), pab);
    std::vector<CoopMem *> m(2 * K);
    for (int w = 0; w < 2 * K; w++) {
        m[w] = new CoopMem();
        coop_init(*m[w], &sc, pa, pb);
        const int pair = w & 1, seg = w >> 1;
        if (seg > 0) coop_set_one(m[w]->f);
        coop_run(*m[w], prog, sp.pc_start[seg], COOP_MILLER_INSNS, w1.data(), w2.data(), pair == 0 && use1, pair == 1 && use2, ft, pre.data(),
                sp.pc_lines_end[seg]);
    }
    for (int stride = 1; stride < 2 * K; stride <<= 1)
        for (int w = 0; w + stride < 2 * K; w += 2 * stride) {
            for (int k = 0; k < 12; k++) m[w]->t0.c[k] = m[w + stride]->f.c[k];
            coop_product(*m[w], m[w]->sc.mul, m[w]->f, m[w]->f, m[w]->t0, FULL_MASK);
        }
    coop_run(*m[0], prog, COOP_MILLER_INSNS, n_insn, w1.data(), w2.data(), false, false, ft);
    *ok = coop_is_one(*m[0], m[0]->t0) ? 1 : 0;
    for (auto *x : m) delete x;
    return 0;
}
// The pairing check with the hard part of the final exponentiation run twelve lanes per check (pairing_lanes.h) behind the cooperative Miller loops
// and easy part.  Returns 0 and the verdict; 4 if any coefficient of the hard part's result differs (as a field element) from the cooperative run's.
int hd_pairings_verify_lanes12(int *ok, const uint8_t *p1, const uint8_t *q1, const uint8_t *p2, const uint8_t *q2) {
    G1Affine a, b; G2Affine qa, qb;
    if (g1_decompress(a, p1) || g1_decompress(b, p2) || g2_decompress(qa, q1) || g2_decompress(qb, q2)) return 1;
    std::vector<LineCoeff> l1(N_LINES), l2(N_LINES);
    std::vector<LineW> w1(N_LINES), w2(N_LINES);
    if (!g2a_is_inf(qa)) precompute_lines(l1.data(), qa);
    if (!g2a_is_inf(qb)) precompute_lines(l2.data(), qb);
    for (int i = 0; i < N_LINES; i++) { line_to_w(w1[i], l1[i]); line_to_w(w2[i], l2[i]); }
    PairPt pa, pb; pairpt_from_affine(pa, a); pairpt_from_affine(pb, b);
    { Fp ny; fp_neg(ny, pa.ay); pa.ay = ny; }                         // e(a, qa) e(b, qb) == 1 with the first point negated, as the other probes do
    const bool use1 = !fp_is_zero(pa.az) && !g2a_is_inf(qa), use2 = !fp_is_zero(pb.az) && !g2a_is_inf(qb);
    static const uint32_t A1[12][NFP] = FROBW_A1_INIT, B1[12][NFP] = FROBW_B1_INIT, A2[12][NFP] = FROBW_A2_INIT;
    FrobTables ft;
    for (int k = 0; k < 12; k++) for (int i = 0; i < NFP; i++) { ft.a1[k].l[i] = A1[k][i]; ft.b1[k].l[i] = B1[k][i]; ft.a2[k].l[i] = A2[k][i]; }
    static CoopInsn prog[COOP_PROGRAM_MAX];
    int hard = 0;
    const int n_insn = build_pairing_program(prog, &hard);
    static CoopScheds sc;
    if (!build_coop_schedules(sc) || hard <= 0 || hard >= n_insn) return 3;
    CoopMem *m = new CoopMem();
    coop_init(*m, &sc, pa, pb);
    coop_run(*m, prog, 0, hard, w1.data(), w2.data(), use1, use2, ft);
    L12Mem *lm = new L12Mem();
    for (int k = 0; k < 12; k++) { Fp c; fp_norm_lz(c, m->f.c[k]); fp_canon64(c, c); lm->s[0].c[k] = c; }      // the hand-over: canonical coefficients
    l12_run(*lm, prog, hard, n_insn, ft);
    coop_run(*m, prog, hard, n_insn, w1.data(), w2.data(), use1, use2, ft);
    int rc = 0;
    bool one = true;
    for (int k = 0; k < 12; k++) {
        Fp x, y; fp_norm_lz(x, m->t0.c[k]); fp_canon64(x, x); fp_norm_lz(y, lm->s[1].c[k]); fp_canon64(y, y);
        if (!fp_eq(x, y)) rc = 4;
        one = one && l12_coeff_is_one(lm->s[1], k);
    }
    *ok = one ? 1 : 0;
    if (rc == 0 && one != coop_is_one(*m, m->t0)) rc = 5;
    delete m; delete lm;
    return rc;
}
// W = sum_b b * B_b over 16 buckets the way k_lc_wsum does it: every bucket sum B_b = P_{2(b-1)} + P_{2(b-1)+1} accumulated in lazy
// extended-Jacobian coordinates (g1x_add_mixed_lazy), then acc += B_b; W += acc for b = 16 .. 1 with the lazy XYZZ + XYZZ addition.
int hd_weighted_bucket_sum(uint8_t *out, const uint8_t *pts48 /* 32 points */) {
    G1X B[16];
    for (int b = 0; b < 16; b++) {
        G1X acc = g1x_inf(); bool started = false;
        for (int k = 0; k < 2; k++) { G1Affine p; if (g1_decompress(p, pts48 + 48 * (2 * b + k))) return 1; g1x_add_mixed_lazy(acc, started, p); }
        if (!started) acc = g1x_inf();
        B[b] = acc;                                              // raw, as the bucket kernel parks it (all-zero = infinity)
    }
    G1X acc = B[15], sum = acc;
    for (int b = 14; b >= 0; b--) { g1x_add_lazy2(acc, acc, B[b]); g1x_add_lazy2(sum, sum, acc); }
    G1X c; g1x_from_lazy(c, sum, true);
    G1Jac j; g1x_to_jac(j, c);
    G1Affine a; g1_to_affine(a, j); g1_compress_affine(out, a); return 0;
}
void hd_sha256(uint8_t *out, const uint8_t *msg, uint64_t len) { sha256_bytes(out, msg, len); }
}
