// eval_core.h -- the arithmetic core of k_eval (k_verify.hip): evaluation of a blob polynomial given by its 4096 values at
// the bit-reversed roots of unity (reference src/kzg.rs:346-389) as a binary tree over the domain with no inversion and no special case.
// Host + device: the same source is driven lane by lane on the device and serially by tests/native/hd_probe.cpp.
//
// With w / (z - w) = z / (z - w) - 1 and prod_j (z - w_j) = z^N - 1 the reference's formula becomes
//     y = (z^N - 1) / N * sum_i p_i w_i / (z - w_i) = ( z * Ntop - (z^N - 1) * sum_i p_i ) / N,     Ntop = sum_i p_i prod_{j != i} (z - w_j)
// a polynomial identity in z: for z = w_m the second term vanishes and Ntop = p_m N / w_m, so y = p_m -- what the reference's branch
// kzg.rs:360-362 returns; nothing is ever divided.  Ntop is the numerator of sum_i p_i / (z - w_i) over the common denominator, built up
// the tree of quot_core.h: in bit-reversal order the first 2^d roots are the 2^d-th roots of unity, node (d, a), a < 2^d, owns the
// factor z^(N / 2^d) - roots[a], and its children's factors are z^k -+ s with s = roots[2a], k = N / 2^(d+1).  The fractions
// n0 / (z^k - s) + n1 / (z^k + s) of the two children add up to
//     n = n0 (z^k + s) + n1 (z^k - s) = z^k (n0 + n1) + s (n0 - n1)
// over the node's factor: TWO limb products that share ONE Montgomery reduction (mont_mul2_lazy), z^k a per-blob value (uniform over the
// wave that owns the blob), s a constant of the tree.  4095 nodes per blob: 2 x 4095 limb-product passes + 4095 reductions, against
// 5 + 5 per four values (6825 of each per blob) in the radix-4 Horner form of rounds 3-5, and a third of its additions.
// The leaves are the values themselves (plain integers), sum_i p_i rides along as a carry-swept sum per lane.
//
// The device deals the tree in groups of three nodes: a group takes four neighbouring children (128 bytes of the blob at level 1) to
// their grandparent, so six levels of groups -- 1024, 256, 64, 16, 4, 1 groups -- cover the twelve levels of nodes.  Level l uses
// z^(4^(l-1)) for its two lower nodes and its square for the upper one, and three roots per group (EvalGroup; one flat table).
//
// Domains: z^k and the roots are Montgomery residues; the values enter as plain integers, so every n and y are plain.
// Bounds in units of r (lazy products: result < 1 + (a b + c d) / 70.6): leaves < 2.21 (any 256-bit value -- a non-canonical one is flagged
// by the caller, the arithmetic stays in range); a node with children < c: A = n0 + n1 < 2c (limbs < 2^30, no carries), B = n0 - n1 + K r
// < c + K (K = 3 at the leaves, 2 above; carry-free, limbs < 3 * 2^29), z^k, s < 1: n < 1 + (3c + K) / 70.6 = 1.14 from the leaves and
// 1.08 from then on, at every level.  Column sums of the two products: nine rounds of 2^30 2^29 + 3 * 2^29 2^29 + 2^29 2^29 = 27 * 2^59 <
// 2^63.8.  sum p: a lane's 64 values add up to < 142 r (top limb < 2^30.1); one lazy product by R mod r brings that below 3.1 before the
// sixty-four lanes are added (< 200 r), and the last two-product step takes z Ntop + (1 - z^N) sum p < 201 r^2 to < 3.9 r; the product
// by 1 / N is canonical.
#pragma once
#include "field.h"

namespace kzg {

// the three roots of a group: level l (1..6), group m (< 4^(6-l)) joins the nodes 4m .. 4m+3 of depth 14 - 2l: sa = roots[4m] and sb = roots[4m + 2]
// for the two lower nodes, st = roots[2m] for the upper one (roots: the domain in bit-reversal order, Montgomery).  One flat table, level after level.
struct EvalGroup { Fr sa, sb, st; };
constexpr int EVAL_ZPOWERS = 12;       // z^2, z^4, ..., z^4096 (entry k - 1 holds z^(2^k)): squared up once per blob by the challenge kernel
constexpr int EVAL_WAVES = 4;          // blobs (waves) per workgroup of k_eval: the last three levels of the four trees run together on wave 0
constexpr int EVAL_TAB_L1 = 0, EVAL_TAB_L2 = 1024, EVAL_TAB_L3 = 1280, EVAL_TAB_L4 = 1344, EVAL_TAB_L5 = 1360, EVAL_TAB_L6 = 1364, EVAL_TAB_GROUPS = 1365;
// The device table keeps the 27 limbs of a group as seven 16-byte pieces (the last limb slot is padding) with the pieces of 64 consecutive groups side by
// side: piece q of group e at (e / 64) * 7 * 64 + q * 64 + (e % 64), so that the 64 lanes of a wave that take 64 consecutive groups read 1 KiB per load.
struct EvalPiece { uint32_t w[4]; };
constexpr int EVAL_TAB_PIECES = ((EVAL_TAB_GROUPS + 63) / 64) * 7 * 64;
KZG_HD constexpr int eval_tab_piece(int e, int q) { return (e / 64) * 7 * 64 + q * 64 + (e % 64); }
KZG_HD void eval_group_pack(EvalPiece p[7], const EvalGroup &g) {
    for (int k = 0; k < 28; k++) p[k / 4].w[k % 4] = k < NFR ? g.sa.l[k] : k < 2 * NFR ? g.sb.l[k - NFR] : k < 3 * NFR ? g.st.l[k - 2 * NFR] : 0u;
}
KZG_HD void eval_group_unpack(EvalGroup &g, const EvalPiece p[7]) {
#pragma unroll
    for (int k = 0; k < 27; k++) (k < NFR ? g.sa.l[k] : k < 2 * NFR ? g.sb.l[k - NFR] : g.st.l[k - 2 * NFR]) = p[k / 4].w[k % 4];
}
KZG_HD constexpr int eval_tab_first(int level) {
    return level == 1 ? EVAL_TAB_L1 : level == 2 ? EVAL_TAB_L2 : level == 3 ? EVAL_TAB_L3 : level == 4 ? EVAL_TAB_L4 : level == 5 ? EVAL_TAB_L5 : EVAL_TAB_L6;
}

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

// One node: n = z^k (n0 + n1) + s (n0 - n1), children < K r with normalised limbs (header comment)
template <int K> KZG_HD void eval_node2(Fr &n, const Fr &n0, const Fr &n1, const Fr &zk, const Fr &s) {
    Fr A, B;
    fr_add_raw(A, n0, n1);
    fr_sub_bias_raw<K>(B, n0, n1);
    fr_mul2_lazy(n, A, zk, B, s);
}
// A group above the leaves: four children to their grandparent.  zk = z^k of the two lower nodes, zk2 its square.
KZG_HD void eval_group(Fr &n, const Fr c[4], const Fr &zk, const Fr &zk2, const EvalGroup &g) {
    Fr na, nb;
    eval_node2<2>(na, c[0], c[1], zk, g.sa);
    eval_node2<2>(nb, c[2], c[3], zk, g.sb);
    eval_node2<2>(n, na, nb, zk2, g.st);
}
// A group of four values (8 little-endian words each): the same with K = 3, and Sp += their sum (carries swept: Sp's limbs stay below 2^29,
// its top limb keeps the excess)
KZG_HD void eval_group_leaves(Fr &n, Fr &Sp, const uint32_t pw[4][8], const Fr &z, const Fr &z2, const EvalGroup &g) {
    Fr p0, p1, p2, p3, A, B, na, nb, S;
    words_to_limbs<NFR, 8>(p0.l, pw[0]); words_to_limbs<NFR, 8>(p1.l, pw[1]);
    words_to_limbs<NFR, 8>(p2.l, pw[2]); words_to_limbs<NFR, 8>(p3.l, pw[3]);
    fr_add_raw(A, p0, p1); fr_sub_bias_raw<3>(B, p0, p1);
    fr_mul2_lazy(na, A, z, B, g.sa);
    fr_add_raw(S, p2, p3); fr_sub_bias_raw<3>(B, p2, p3);
    fr_mul2_lazy(nb, S, z, B, g.sb);
    fr_add_raw(S, S, A);                                                       // limbs < 2^31
    fr_add_sweep(Sp, Sp, S);
    eval_node2<2>(n, na, nb, z2, g.st);
}
// a lane's sum of values back below 3.1 r (its value mod r unchanged): the lazy product by R mod r
KZG_HD void eval_fold(Fr &s) { Fr t; fr_mul_lazy(t, s, fr_one()); s = t; }
// y = (z Ntop - (z^N - 1) Sp) / N as the canonical plain integer; Sp the blob's folded sum (< 200 r), zN = z^4096 (canonical)
KZG_HD void eval_finish(Fr &y, const Fr &ntop, const Fr &Sp, const Fr &z, const Fr &zN) {
    const uint32_t inv4096[NFR] = FR_INV4096_INIT;
    Fr k4096; for (int i = 0; i < NFR; i++) k4096.l[i] = inv4096[i];
    Fr negD, t;
    fr_sub(negD, fr_one(), zN);                                                // 1 - z^N mod r, canonical
    fr_mul2_lazy(t, ntop, z, Sp, negD);                                        // < 3.9 r, plain
    fr_mul(y, t, k4096);                                                       // canonical: the chain of lazy products ends here
}

}  // namespace kzg
