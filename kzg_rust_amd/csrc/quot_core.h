// quot_core.h -- the arithmetic core of the quotient kernel (k_prove.hip): compute_kzg_proof_impl's y = p(z) and
//     q_i = (p_i - y) / (w_i - z)                                    (reference src/kzg.rs:461-490)
// for z outside the domain, WITHOUT the reference's three 4096-long batch inversions (kzg.rs:368, 484, 508) and without the
// "product of all the others" scan of round 1-4's kernel (ten field products per element).  Host + device: the same source is driven
// lane by lane on the device and serially by tests/native/hd_probe.cpp.
//
// The inverses come DOWN a binary tree over the domain.  In bit-reversal order the first 2^d roots are the 2^d-th roots of unity, and
// node (d, a), a < 2^d, stands for the factor  z^(N / 2^d) - w_a  (w_a = roots[a]); its two children are the square roots of w_a:
//     roots[2a] = s,  roots[2a + 1] = -s,   (z^k - s)(z^k + s) = z^(2k) - w_a,   k = N / 2^(d+1)
// so with inv(d, a) = 1 / (z^(2k) - w_a):
//     inv(d + 1, 2a) = (z^k + s) inv(d, a),     inv(d + 1, 2a + 1) = (z^k - s) inv(d, a)          -- one product per node,
// the root is inv(0, 0) = 1 / (z^N - 1) (ONE inversion per blob, taken by k_quotient_prep with a lane per blob) and the 4096 leaves are
// 1 / (z - w_i): 8190 products per blob = 2 per element, no batch inversion, no second table (the sums z^k +- s are additions).
//
// y needs no pass of its own:  w_i / (z - w_i) = z / (z - w_i) - 1, so with u_i = p_i / (z - w_i)
//     y = (z^N - 1) / N * sum_i p_i w_i / (z - w_i) = c_N (z sum_i u_i - sum_i p_i),    c_N = (z^N - 1) / N           (kzg.rs:346-389)
// and q_i = (y - p_i) / (z - w_i): pass 1 walks the tree, forms u_i (one product) and parks 1 / (z - w_i) in the output slot of q_i;
// after the blob's two sums, pass 2 reads it back: one product per element.  Per lane of a 256-thread workgroup (16 leaves): 8 (path
// from the root, repeated by every lane) + 30 (its subtree) + 16 (u) + 16 (q) = 70 products = 4.4 per element.
//
// Domains: z, its powers, the roots, every inverse, W and c_N are Montgomery residues; the blob's values enter as plain integers, so
// u, the sums, y and q are plain -- q leaves as the canonical 32-byte big-endian integer the fixed-base MSM reads like a blob.
// Lazy products throughout (mont_mul_lazy: result < r (1 + a b / (70.6 r^2))); bounds in units of r:
//   z^k, roots < 1 (canonical);  s+ = z^k + s < 2;  s- = z^k - s + r < 2;  every inverse < 1.1;
//   p < 2.21 (any 256-bit value: a non-canonical element is flagged by the caller, the arithmetic stays in range);  u < 1.1;
//   per-lane sums over at most 64 leaves: S_u < 71, S_p < 142 (top limb < 2^30.2: inside mont_mul_lazy's columns);  one lazy product by
//   R mod r (Montgomery one) brings a sum below 1 + 142 / 70.6 = 3.1 before it is added across lanes (64 x 3.1 = 198 < 280) and again
//   before it is added across waves;  y - p + 3 r < 4;  (y - p + 3 r) inv < 1 + 4.4 / 70.6: ONE conditional subtraction makes q canonical.
// z inside the domain (z^N = 1, kzg.rs:494-523) has no root inverse: k_quotient_prep lists such blobs and the scan kernel of rounds 1-4
// (k_quotient_scan, which needs no inversion there) takes them.
#pragma once
#include "eval_core.h"

namespace kzg {

struct QuotPrep {
    Fr zsq[12];      // z^(2^k), k = 0 .. 11 (Montgomery, canonical)
    Fr W;            // 1 / (z^4096 - 1) (Montgomery); zero when z is inside the domain
    Fr cN;           // (z^4096 - 1) / 4096 (Montgomery)
};

// per blob (one lane each): the powers, the root inverse; true when z is inside the domain
KZG_HD bool quot_prep(QuotPrep &o, const Fr &z) {
    const uint32_t inv4096[NFR] = FR_INV4096_INIT;
    Fr k4096; for (int i = 0; i < NFR; i++) k4096.l[i] = inv4096[i];
    Fr p = z;
    for (int k = 0; k < 12; k++) { o.zsq[k] = p; fr_sqr(p, p); }
    Fr D; fr_sub(D, p, fr_one());
    const bool inside = fr_is_zero(D);
    fr_mul(o.cN, D, k4096);
    if (inside) o.W = fr_zero();
    else fr_inv(o.W, D);
    return inside;
}

// inv(dn, a) from the root: dn products (every lane of a blob's workgroup repeats the levels it shares with its neighbours)
KZG_HD void quot_path(Fr &inv, const QuotPrep &pp, const Fr *roots, int dn, int a) {
    inv = pp.W;
    for (int d = 1; d <= dn; d++) {
        Fr s; fr_add_lazy(s, pp.zsq[12 - d], roots[a >> (dn - d)]);       // z^k + roots[a_d]: roots[odd] is the negative square root
        fr_mul_lazy(inv, s, inv);
    }
}
// the two children of a node: inv the node's inverse, zk = z^k of the children's level, s = roots[2 a]
KZG_HD void quot_children(Fr &c0, Fr &c1, const Fr &inv, const Fr &zk, const Fr &s) {
    Fr sp, sm;
    fr_add_lazy(sp, zk, s);
    fr_sub_bias<1>(sm, zk, s);
    fr_mul_lazy(c0, sp, inv);
    fr_mul_lazy(c1, sm, inv);
}
// pass 1 over the four leaves 4 a10 .. 4 a10 + 3 of the depth-10 node a10 (128 bytes of the blob): their inverses (lazy, < 2^256) and
// the running sums S_u += p / (z - w), S_p += p.  pw: the values as 8 little-endian words each.
KZG_HD void quot_group_pass1(Fr inv12[4], Fr &Su, Fr &Sp, const uint32_t pw[4][8], const Fr &inv10, int a10, const QuotPrep &pp, const Fr *roots) {
    Fr i11[2];
    quot_children(i11[0], i11[1], inv10, pp.zsq[1], roots[2 * a10]);
#pragma unroll
    for (int h = 0; h < 2; h++) quot_children(inv12[2 * h], inv12[2 * h + 1], i11[h], pp.zsq[0], roots[2 * (2 * a10 + h)]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        Fr p, u;
        words_to_limbs<NFR, 8>(p.l, pw[j]);
        fr_mul_lazy(u, p, inv12[j]);
        fr_add_lazy(Su, Su, u);
        fr_add_lazy(Sp, Sp, p);
    }
}
// a sum back below ~3 r (its value mod r unchanged): the lazy product by R mod r
KZG_HD void quot_fold(Fr &s) { Fr t; fr_mul_lazy(t, s, fr_one()); s = t; }
// y = c_N (z S_u - S_p) as the canonical plain integer; S_u, S_p folded sums of the whole blob.
// PRECONDITION (what makes the bias of 2 r below sufficient): S_p < R = 2^261 ~ 68.6 r.  A lazy product a b / R comes out below a b / R + r, so
// spn = S_p (R mod r) / R < S_p r / R + r < 2 r exactly when S_p < R.  The callers fold every partial sum once before they add them up (quot_fold:
// each < ~3 r; 16 waves of partial sums: < 50 r), which is inside that bound with room to spare; a caller that adds up more than 22 folded sums
// must fold again first (or take a bias of 4).
KZG_HD void quot_y(Fr &y, const Fr &Su, const Fr &Sp, const QuotPrep &pp) {
    Fr t1, t2, spn;
    fr_mul_lazy(t1, Su, pp.zsq[0]);
    fr_mul_lazy(spn, Sp, fr_one());                               // < 2 r (see above)
    fr_sub_bias<2>(t2, t1, spn);
    fr_mul(y, t2, pp.cN);
}
// pass 2, one leaf: q = (y - p) / (z - w) as 8 little-endian words of the canonical integer
KZG_HD void quot_leaf_pass2(uint32_t qw[8], const uint32_t pw[8], const uint32_t invw[8], const Fr &y) {
    Fr p, inv, d, q;
    words_to_limbs<NFR, 8>(p.l, pw);
    words_to_limbs<NFR, 8>(inv.l, invw);
    fr_sub_bias<3>(d, y, p);
    fr_mul(q, d, inv);
    limbs_to_words<NFR, 8>(qw, q.l);
}

}  // namespace kzg
