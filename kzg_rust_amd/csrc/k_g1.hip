// k_g1.hip -- the G1-only kernels of the verify path: point validation (utils.rs:282-310) and the random linear
// combinations of verify_kzg_proof_batch (kzg.rs:601-622).  Split from k_verify.hip so that this translation unit can
// use its own inlining policy: these kernels are single dependent chains per lane (a lone wave issues one instruction
// every ~5 cycles), so call / scratch overhead is pure latency -- the G1 formulas and the Fp product are force-inlined.
#if !defined(KZG_G1_TU_NOINLINE)
#define KZG_MID_INLINE 1
#else
#define KZG_FP_MUL_NOINLINE 1
#endif
#include "kernels.h"
#include "g1_quad.h"

namespace kzg {

// ------------------------------------------------------------------------------------------------ points
// thread j < n_total: commitment j ; j >= n_total: proof j - n_total.
__device__ __forceinline__ void validate_points_body(const uint8_t *commitments, const uint8_t *proofs, int n_total, int n_per_group, G1Affine *pts, int *err,
        int stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * n_total) return;
    const bool is_proof = j >= n_total;
    if (is_proof && !proofs) return;
    const int i = is_proof ? j - n_total : j;
    const uint8_t *src = (is_proof ? proofs : commitments) + (size_t)stride * i;     // 48: packed arrays; 160: fields of records
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = src[k];
    G1Affine p;
    int rc = g1_decompress(p, b);
    if (rc == 0 && !g1a_is_inf(p) && !g1_in_subgroup(p)) rc = 3;     // infinity is accepted (utils.rs:298-301)
    const int g = i / n_per_group, k = i % n_per_group;
    if (rc != 0) { atomicOr(&err[g], ERR_BAD_POINT); p = g1a_inf(); }
    if (pts) pts[(size_t)g * 2 * n_per_group + (is_proof ? n_per_group + k : k)] = p;
}
__global__ void __launch_bounds__(256, 2) k_validate_points(const uint8_t *commitments, const uint8_t *proofs, int n_total, int n_per_group,
                                                         G1Affine *pts, int *err, int stride) {
    validate_points_body(commitments, proofs, n_total, n_per_group, pts, err, stride);
}

// The two halves of k_validate_points as kernels of their own, for the single-proof entry point: the linear combination only needs
// the decompressed points, the subgroup test only feeds the error word -- so the test runs on the side stream beside the rest of the
// chain (1.2 of verify_kzg_proof's 6.1 ms).
__global__ void __launch_bounds__(64) k_decompress_points(const uint8_t *commitments, const uint8_t *proofs, int n_total, int n_per_group,
                                                          G1Affine *pts, int *err, int stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * n_total) return;
    const bool is_proof = j >= n_total;
    if (is_proof && !proofs) return;                              // commitments only (compute_blob_kzg_proof's challenge step, kzg.rs:321)
    const int i = is_proof ? j - n_total : j;
    const uint8_t *src = (is_proof ? proofs : commitments) + (size_t)stride * i;
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = src[k];
    G1Affine p;
    const int g = i / n_per_group, k = i % n_per_group;
    if (g1_decompress(p, b) != 0) { atomicOr(&err[g], ERR_BAD_POINT); p = g1a_inf(); }
    pts[(size_t)g * 2 * n_per_group + (is_proof ? n_per_group + k : k)] = p;
}
__global__ void __launch_bounds__(64) k_subgroup_points(const G1Affine *pts, int n_points, int n_per_group, int *err, int commitments_only) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_points) return;
    if (commitments_only && (j % (2 * n_per_group)) >= n_per_group) return;       // the proof slots of the layout were never written
    const G1Affine p = pts[j];
    if (!g1a_is_inf(p) && !g1_in_subgroup(p)) atomicOr(&err[j / (2 * n_per_group)], ERR_BAD_POINT);      // infinity is accepted (utils.rs:298-301)
}

// The subgroup test by DPP quads (g1_quad.h), for few points: the two [|x|] ladders are 126 doublings + 10 additions in a row, 2 + 1 and
// 5 products deep here instead of 7 and 16 (1.0 -> 0.67 ms for a lone point: it is the critical path of compute_blob_kzg_proof once the
// challenge is hashed on the host).  Same predicate as g1_in_subgroup (g1.h): phi(P) == -[x^2]P.
__global__ void __launch_bounds__(256) k_subgroup_points_quad(const G1Affine *pts, int n_points, int n_per_group, int *err, int commitments_only) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, role = tid & 3;
    const bool live = (tid >> 2) < n_points;
    const int j = live ? (tid >> 2) : n_points - 1;               // idle quads redo the last point (every lane takes part in the DPP moves)
    const bool skip = !live || (commitments_only && (j % (2 * n_per_group)) >= n_per_group);      // (the proof slots of the layout were never written)
    G1Affine p = pts[skip ? 0 : j];
    if (skip) p = g1a_inf();
    G1Jac base; g1_from_affine(base, p);
    G1Jac t = base;
    // one loop, one inlined instance of each quad routine: two passes of the 63-step ladder for |x| = 0xd201000000010000
#pragma unroll 1
    for (int step = 0; step < 126; step++) {
        const int i = 62 - (step % 63);
        if (step == 63) { g1_canon_lazy(t, t); base = t; }       // second ladder: [|x|] of the first one's result
        g1_dbl_quad(t, role);
        if ((BLS_X_ABS >> i) & 1) g1_add_quad(t, t, base, role);  // (a constant exponent: the branch is uniform)
    }
    g1_canon_lazy(t, t);
    bool ok = true;
    if (!g1a_is_inf(p)) {
        ok = !g1_is_inf(t);
        const uint32_t bc[NFP] = FP_BETA_INIT;
        Fp beta; for (int q = 0; q < NFP; q++) beta.l[q] = bc[q];
        Fp z2, z3, lhs, rhs;
        fp_sqr(z2, t.z); fp_mul(z3, z2, t.z);
        fp_mul(lhs, p.x, beta); fp_mul(lhs, lhs, z2);             // beta x Z^2 == X
        ok = ok && fp_eq(lhs, t.x);
        fp_mul(lhs, p.y, z3); fp_neg(rhs, t.y);                   // y Z^3 == -Y
        ok = ok && fp_eq(lhs, rhs);
    }
    if (!skip && role == 0 && !ok) atomicOr(&err[j / (2 * n_per_group)], ERR_BAD_POINT);      // infinity is accepted (utils.rs:298-301)
}

// The same ladder started from x ALONE, beside the square root of the decoding instead of behind it (the trick of k_ps_shift: with s = x^3 + 4 = y^2 the
// curve E'': Y^2 = X^3 + 4 s^3 is isomorphic to E, (s x, s^2) is the image of P = (x, y), and neither the doubling nor the addition formulas of a = 0
// curves involve the constant -- so [x^2] of that point is walked on E'' unchanged, and a Jacobian point (X, Y, Z) of E'' IS (X, Y, y Z) on E).  For
// compute_blob_kzg_proof the commitment's validation is the critical path once the challenge is hashed on the host: 0.45 ms of square root, THEN 0.65 ms
// of ladder.  k_subgroup_ladder_from_x_quad leaves [x^2]P'' per point; k_subgroup_finish, behind both, applies y and tests phi(P) == -[x^2]P.
__global__ void __launch_bounds__(256) k_subgroup_ladder_from_x_quad(const uint8_t *cbytes, int stride, int n, G1Jac *T) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, role = tid & 3;
    const bool live = (tid >> 2) < n;
    const int i = live ? (tid >> 2) : n - 1;                      // idle quads redo the last point (every lane takes part in the DPP moves)
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = cbytes[(size_t)stride * i + k];
    Fp x = fp_zero(), sv;
    bool inf = false, large;
    if (g1_parse_compressed(x, inf, large, b)) inf = true;       // (a bad encoding is an error of the decoding kernel; here it is the point at infinity)
    g1_curve_rhs(sv, x);
    G1Jac base;
    fp_mul(base.x, sv, x); fp_sqr(base.y, sv); base.z = fp_one();
    if (inf) base = g1_inf();
    G1Jac t = base;
    // one loop, one inlined instance of each quad routine: two passes of the 63-step ladder for |x| = 0xd201000000010000
#pragma unroll 1
    for (int step = 0; step < 126; step++) {
        const int bit = 62 - (step % 63);
        if (step == 63) { g1_canon_lazy(t, t); base = t; }       // second ladder: [|x|] of the first one's result
        g1_dbl_quad(t, role);
        if ((BLS_X_ABS >> bit) & 1) g1_add_quad(t, t, base, role);   // (a constant exponent: the branch is uniform)
    }
    g1_canon_lazy(t, t);
    if (live && role == 0) T[i] = t;
}
// pts: the decoded points in the [group][commitments | proofs] layout with one commitment per group (slot 2 i); T: the ladder's results on E''
__global__ void __launch_bounds__(64) k_subgroup_finish(const G1Affine *pts, const G1Jac *T, int n, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = pts[2 * (size_t)i];
    if (g1a_is_inf(p)) return;                                    // infinity is accepted (utils.rs:298-301); a point that failed to decode was flagged there
    G1Jac t = T[i];
    Fp tz; fp_mul(tz, t.z, p.y);                                  // (X, Y, Z) on E'' = (X, Y, y Z) on E
    bool ok = !fp_is_zero(tz);
    const uint32_t bc[NFP] = FP_BETA_INIT;
    Fp beta; for (int q = 0; q < NFP; q++) beta.l[q] = bc[q];
    Fp z2, z3, lhs, rhs;
    fp_sqr(z2, tz); fp_mul(z3, z2, tz);
    fp_mul(lhs, p.x, beta); fp_mul(lhs, lhs, z2);                 // beta x Z^2 == X
    ok = ok && fp_eq(lhs, t.x);
    fp_mul(lhs, p.y, z3); fp_neg(rhs, t.y);                       // y Z^3 == -Y
    ok = ok && fp_eq(lhs, rhs);
    if (!ok) atomicOr(&err[i], ERR_BAD_POINT);
}

// Decompress the C_i / proof_i fields of gathered records (already validated by their owner rank).
__global__ void __launch_bounds__(64) k_points_from_records(const uint8_t *records, int n_total, int n_per_group, G1Affine *pts, int *err) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * n_total) return;
    const bool is_proof = j >= n_total;
    const int i = is_proof ? j - n_total : j;
    const uint8_t *src = records + (size_t)RECORD_BYTES * i + (is_proof ? 112 : 0);
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = src[k];
    G1Affine p;
    const int g = i / n_per_group, k = i % n_per_group;
    if (g1_decompress(p, b) != 0) { atomicOr(&err[g], ERR_BAD_POINT); p = g1a_inf(); }
    pts[(size_t)g * 2 * n_per_group + (is_proof ? n_per_group + k : k)] = p;
}

// ------------------------------------------------------------------------------------------------ lincomb
// The 3n + 1 scalar multiplications of one batch
//   class 0:  a_i * proof_i                                   -> proof_lincomb         (kzg.rs:601)
//   class 1:  b_i * proof_i,  a_i * C_i,  c * (-G)            -> rhs                   (kzg.rs:603-622)
// are each split with the GLV endomorphism into two 128-bit halves  [k]P = [k mod x^2]P + [k div x^2](-phi P)
// placed on two different lanes: 2(3n+1) items per batch, each a 128-step double-and-add chain.
//
// Latency matters here (a lone wave issues one instruction per ~5 cycles; two waves sharing a SIMD each drop to one per
// ~9), so the items are dealt to ONE-WAVE workgroups that the dispatcher spreads over different CUs:
//   k_lincomb_terms   one 64-lane workgroup per 64 items: scalar multiplication, then a butterfly sum per class over
//                     the wave (shuffles, no LDS) -> 2 partial points per wave
//   k_lincomb_finish  one workgroup per batch: lane c sums the partials of class c and converts to affine (the two
//                     inversions run side by side) -> (-proof_lincomb, rhs) for the pairing.
__device__ __forceinline__ G1Jac g1_shfl_down8(const G1Jac &v, int delta) {      // within segments of 8 lanes
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_down(v.x.l[i], delta, 8); r.y.l[i] = __shfl_down(v.y.l[i], delta, 8); r.z.l[i] = __shfl_down(v.z.l[i],
            delta, 8); }
    return r;
}
__device__ __forceinline__ G1Jac g1_shfl_xor(const G1Jac &v, int mask) {
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_xor(v.x.l[i], mask, 64); r.y.l[i] = __shfl_xor(v.y.l[i], mask, 64); r.z.l[i] = __shfl_xor(v.z.l[i], mask,
            64); }
    return r;
}
__host__ __device__ inline int lincomb_waves_per_group(int n) { return (2 * (3 * n + 1) + 63) / 64; }

__global__ void __launch_bounds__(64) k_lincomb_terms(const G1Affine *pts, const uint32_t *scal_a, const uint32_t *scal_b,
                                                       const uint32_t *scal_c, int n, G1Jac *partials, uint32_t *wtabs) {
    const int wpg = lincomb_waves_per_group(n);
    const int g = blockIdx.x / wpg, wv = blockIdx.x % wpg, lane = threadIdx.x;
    uint32_t *wtab = wtabs + (size_t)blockIdx.x * (W4_ENTRIES * 3 * NFP * 64);
    const int item = wv * 64 + lane;
    const G1Affine *gp = pts + (size_t)g * 2 * n;        // [0,n) commitments, [n,2n) proofs
    G1Jac m = g1_inf();
    int cls = -1;
    if (item < 2 * (3 * n + 1)) {
        const int t = item >> 1, half = item & 1;
        G1Affine p; uint32_t k[8];
        if (t < n) { p = gp[n + t]; for (int q = 0; q < 8; q++) k[q] = scal_a[8 * ((size_t)g * n + t) + q]; cls = 0; }
        else if (t < 2 * n) { p = gp[n + (t - n)]; for (int q = 0; q < 8; q++) k[q] = scal_b[8 * ((size_t)g * n + (t - n)) + q]; cls = 1; }
        else if (t < 3 * n) { p = gp[t - 2 * n]; for (int q = 0; q < 8; q++) k[q] = scal_a[8 * ((size_t)g * n + (t - 2 * n)) + q]; cls = 1; }
        else {
            const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
            for (int q = 0; q < NFP; q++) { p.x.l[q] = gx[q]; p.y.l[q] = gy[q]; }
            fp_neg(p.y, p.y);
            for (int q = 0; q < 8; q++) k[q] = scal_c[8 * (size_t)g + q];
            cls = 1;
        }
        uint32_t ka[4], kb[4];
        glv_split(ka, kb, k);
        if (half) { G1Affine q; g1a_neg_phi(q, p); p = q; }
        g1_mul128_w4(m, p, half ? kb : ka, wtab, lane);
    }
    for (int c = 0; c < 2; c++) {
        G1Jac v = g1_inf();
        if (__ballot(cls == c) != 0ull) {                   // wave-uniform: skip a class nobody in this wave holds
            if (cls == c) v = m;
#pragma unroll 1
            for (int off = 1; off < 64; off <<= 1) { G1Jac o = g1_shfl_xor(v, off); g1_add(v, v, o); }
        }
        if (lane == 0) partials[((size_t)g * wpg + wv) * 2 + c] = v;
    }
}

__global__ void __launch_bounds__(64) k_lincomb_finish(const G1Jac *partials, int n, PairPt *pair_pts) {
    const int wpg = lincomb_waves_per_group(n);
    const int g = blockIdx.x, c = threadIdx.x;
    if (c >= 2) return;
    G1Jac acc = g1_inf();
    for (int wv = 0; wv < wpg; wv++) { G1Jac v = partials[((size_t)g * wpg + wv) * 2 + c]; g1_add(acc, acc, v); }
    PairPt a; pairpt_from_jac(a, acc, c == 0);                // pairings_verify negates its first G1 argument (utils.rs:198-201)
    pair_pts[2 * (size_t)g + c] = a;
}

// ------------------------------------------------------------------------------------------------ lincomb, one record per check
// MANY independent single-proof checks (verify_kzg_proof kzg.rs:409-426, verify_blob_kzg_proof kzg.rs:547-569; the *_many entry points): with one
// record per "batch" the equation has r^0 = 1, so proof_lincomb IS the proof and rhs = C + [z] proof + [y](-G) -- two scalar multiplications, GLV-
// split into four 128-bit ladders on four neighbouring lanes: 16 checks per wave.  (k_lincomb_terms with n = 1 walks two more ladders for the
// scalars a_0 = 1 and keeps 8 of its 64 lanes busy.)  scal_b[g] = z, scal_c[g] = y as k_rpowers leaves them for n = 1.
__global__ void __launch_bounds__(64) k_lincomb_single(const G1Affine *pts, const uint32_t *scal_b, const uint32_t *scal_c, int groups, PairPt *pair_pts,
                                                        uint32_t *wtabs) {
    const int lane = threadIdx.x, q = lane & 3;
    const int g_raw = blockIdx.x * 16 + (lane >> 2);
    const bool live = g_raw < groups;
    const int g = live ? g_raw : groups - 1;                      // idle quads redo the last check (every lane takes part in the shuffles)
    uint32_t *wtab = wtabs + (size_t)blockIdx.x * (W4_ENTRIES * 3 * NFP * 64);
    const G1Affine *gp = pts + 2 * (size_t)g;                     // [0] commitment, [1] proof
    G1Affine p; uint32_t k[8];
    if (q < 2) { p = gp[1]; for (int w = 0; w < 8; w++) k[w] = scal_b[8 * (size_t)g + w]; }      // [z] proof       (kzg.rs:415-418 moved to the G1 side)
    else {                                                                                        // [y] (-G)        (kzg.rs:421)
        const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
        for (int w = 0; w < NFP; w++) { p.x.l[w] = gx[w]; p.y.l[w] = gy[w]; }
        fp_neg(p.y, p.y);
        for (int w = 0; w < 8; w++) k[w] = scal_c[8 * (size_t)g + w];
    }
    uint32_t ka[4], kb[4];
    glv_split_fast(ka, kb, k);                                    // (z, y leave k_rpowers reduced below r < 2^255)
    if (q & 1) { G1Affine t; g1a_neg_phi(t, p); p = t; }
    G1Jac m;
    g1_mul128_w4(m, p, (q & 1) ? kb : ka, wtab, lane);
    // the four ladders of a check, then its commitment: one loop, one inlined instance of the complete addition
    G1Jac c; g1_from_affine(c, gp[0]);
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        G1Jac o = g1_shfl_xor(m, it == 0 ? 1 : 2);
        if (it == 2) o = c;
        g1_add(m, m, o);
    }
    if (!live) return;
    if (q == 0) { PairPt a; pairpt_from_jac(a, m, false); pair_pts[2 * (size_t)g + 1] = a; }
    if (q == 1) {                                                 // pairings_verify negates its first G1 argument (utils.rs:198-201)
        G1Jac pj; g1_from_affine(pj, gp[1]);
        PairPt a; pairpt_from_jac(a, pj, true); pair_pts[2 * (size_t)g] = a;
    }
}

// ------------------------------------------------------------------------------------------------ lincomb, bucket form
// Throughput form of the same two sums: a variable-base Pippenger MSM per batch ("the batched G1 MSM" of
// verify_kzg_proof_batch).  Independent scalar multiplications cost ~1830 field-product equivalents per lane whatever the
// batch; the bucket method shares the doublings:
//   k_lc_prep     one lane per term: GLV split, signed 5-bit recoding of both halves (25 digits in [-16, 15] and an unsigned top
//                 digit), the two points P and -phi(P)                             -> items[2(3n+1)], digits[item][26]
//   k_lc_buckets  one 256-thread workgroup per batch, a wave = 13 (window, class) tasks x 16 buckets = 208 lists: LDS counting
//                 sort of the class's items by |digit|; the lists are ranked by length, dealt longest-first to the least loaded
//                 lane and walked back to back in one loop                          -> bucket sums B[class][window][b]
//   k_lc_horner   one lane per (batch, class, b): Horner over the 26 windows (5 doublings + 1 addition each), then the
//                 weights b over the 16 lanes of a class (suffix scan + butterfly), to affine
//   k_lc_wsum + k_lc_hchain_quad   the tail for many batches (from 1024 on: half the instructions of the 16 chains per class): window sums weighted first, one
//   Horner chain per class
// Window width: the bucket kernel's work is (items x windows) additions -- 4-bit digits 11.6 k per 64-blob batch, 5-bit 9.7 k,
// 6-bit 8.4 k -- while the Horner tail has one chain per bucket index: 16 lanes per class at 5 bits still leave it a
// latency-bound kernel of ~one wave per SIMD at 2048 batches; at 6 bits it would be as much work as the buckets.
// ~2.7x less issue work per batch than the windowed form above; its dependent chain is no shorter (the Horner tail),
// so it is used when many batches are in flight and the windowed form otherwise.
constexpr int LC_BITS = 5;
constexpr int LC_BUCKETS = 1 << (LC_BITS - 1);       // |digit| in 1..16
constexpr int LC_WINDOWS = 26;                       // 25 signed 5-bit digits + the top digit (3 value bits and the carry)
constexpr int LC_DIG_STRIDE = 28;
__host__ __device__ inline int lc_items(int n) { return 2 * (3 * n + 1); }

__global__ void __launch_bounds__(64) k_lc_prep(const G1Affine *pts, const uint32_t *scal_a, const uint32_t *scal_b, const uint32_t *scal_c, int n,
                                                 G1Affine *items, int8_t *digits) {
    const int nt = 3 * n + 1, bpg = (nt + 63) / 64;
    const int g = blockIdx.x / bpg, t = (blockIdx.x % bpg) * 64 + threadIdx.x;
    if (t >= nt) return;
    const G1Affine *gp = pts + (size_t)g * 2 * n;
    G1Affine p; uint32_t k[8];
    if (t < n) { p = gp[n + t]; for (int q = 0; q < 8; q++) k[q] = scal_a[8 * ((size_t)g * n + t) + q]; }
    else if (t < 2 * n) { p = gp[n + (t - n)]; for (int q = 0; q < 8; q++) k[q] = scal_b[8 * ((size_t)g * n + (t - n)) + q]; }
    else if (t < 3 * n) { p = gp[t - 2 * n]; for (int q = 0; q < 8; q++) k[q] = scal_a[8 * ((size_t)g * n + (t - 2 * n)) + q]; }
    else {
        const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
        for (int q = 0; q < NFP; q++) { p.x.l[q] = gx[q]; p.y.l[q] = gy[q]; }
        fp_neg(p.y, p.y);
        for (int q = 0; q < 8; q++) k[q] = scal_c[8 * (size_t)g + q];
    }
    uint32_t half[2][4];
    glv_split(half[0], half[1], k);
    G1Affine q2; g1a_neg_phi(q2, p);
    const size_t base = (size_t)g * lc_items(n) + 2 * (size_t)t;
    items[base] = p; items[base + 1] = q2;
    // k = sum_w d_w 32^w with d_w = chunk_w(k + BIAS) - 16 for w < 25 and d_25 = the remaining top bits: BIAS has bit 4 of every
    // 5-bit chunk below the top one set (sum_{w<25} 16 * 32^w)
    const uint32_t bias[4] = {0x21084210u, 0x08421084u, 0x42108421u, 0x10842108u};
    for (int h = 0; h < 2; h++) {
        uint32_t e[6]; uint64_t c = 0;
        for (int i = 0; i < 4; i++) { c += (uint64_t)half[h][i] + bias[i]; e[i] = (uint32_t)c; c >>= 32; }
        e[4] = (uint32_t)c; e[5] = 0;
        int8_t *d = digits + (base + h) * LC_DIG_STRIDE;
        for (int w = 0; w < LC_WINDOWS; w++) {
            const int bit = LC_BITS * w;
            const uint64_t two = (uint64_t)e[bit >> 5] | ((uint64_t)e[(bit >> 5) + 1] << 32);
            const int chunk = (int)((two >> (bit & 31)) & 31u);
            d[w] = (int8_t)(w < LC_WINDOWS - 1 ? chunk - 16 : chunk);
        }
    }
}

// Lists: per task (= window x class), item | sign << 15 grouped by bucket.  In LDS up to 129 blobs per batch; beyond that
// (multi-GPU batches of 64 x world blobs) in a global scratch slab of this workgroup -- 2 bytes per addition of ~5,000
// instructions either way.  Balance: a wave runs as long as its busiest lane, so the 208 lists of a wave are ranked by length
// and dealt longest-first to whichever lane has the least to do so far; a lane then walks its lists back to back in ONE loop
// (per 7-window wave of a 64-blob batch: one list per lane and pass, four passes: 57.6 steps; this: ~42; ideal 40.2).  The bucket
// sums B[window][b] are NOT weighted here: since
//   sum_w 32^w sum_b b B[w][b] = sum_b b ( sum_w 32^w B[w][b] ),
// the Horner kernel runs one chain per bucket index and applies the weights b once per class.
constexpr int LC_TASKS = 13;                     // tasks (window x class) per wave: 26 windows x 2 classes over 4 waves
constexpr int LC_LDS_LIST = 400;                 // LDS list entries per task: 13 x 400 >= 40 n + 14 entries of a wave for n <= 129
constexpr int LC_MAX_LISTS = 8;                  // lists one lane may be dealt
__host__ __device__ inline int lc_list_stride(int n) { return 2 * (2 * n + 1) + 2; }
// wave wid: class 1 (2 (2n+1) items per window) windows [c1w0, c1w0 + n1), class 0 (2n items) windows [c0w0, c0w0 + 13 - n1):
// 7 + 6, 7 + 6, 6 + 7, 6 + 7 -- within 5 % of each other in additions (whole windows per wave: 7 against 6, 17 %)
__host__ __device__ inline int lc_wave_n1(int wid) { return wid < 2 ? 7 : 6; }
__host__ __device__ inline int lc_wave_c1w0(int wid) { return wid < 2 ? 7 * wid : 14 + 6 * (wid - 2); }
__host__ __device__ inline int lc_wave_c0w0(int wid) { return wid < 2 ? 6 * wid : 12 + 7 * (wid - 2); }
// A bucket sum's slot in the scratch: the bucket kernel parks the raw (lazy) accumulator there the moment a list ends and turns it
// into a canonical Jacobian point -- in place, at the start of the slot -- after its loop; the Horner kernel reads the Jacobian point.
union LcSlot { G1X raw; G1Jac jac; };
// One 256-thread workgroup per batch -- one wave per SIMD of the CU (one-wave workgroups of long chains get placed unevenly, see
// k_pairing.hip), each with its own slice of the LDS arrays; the waves only meet at workgroup barriers that all four reach
// the same number of times.
__global__ void __launch_bounds__(256) k_lc_buckets(const G1Affine *items, const int8_t *digits, int n, LcSlot *S, uint16_t *glists, int keep_raw) {
    __shared__ uint16_t lists_all[4][LC_TASKS * LC_LDS_LIST];
    __shared__ int cnt_all[4][LC_TASKS][LC_BUCKETS + 1], cursor_all[4][LC_TASKS][LC_BUCKETS + 1];
    __shared__ uint8_t order_all[4][LC_BUCKETS * LC_TASKS];
    __shared__ uint8_t seq_all[4][64][LC_MAX_LISTS];
    const int g = blockIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n1 = lc_wave_n1(wid), c1w0 = lc_wave_c1w0(wid), c0w0 = lc_wave_c0w0(wid);
    constexpr int ntasks = LC_TASKS, nlists = LC_BUCKETS * LC_TASKS;
    int (*cnt)[LC_BUCKETS + 1] = cnt_all[wid], (*cursor)[LC_BUCKETS + 1] = cursor_all[wid];
    uint8_t *order = order_all[wid];
    uint8_t (*seq)[LC_MAX_LISTS] = seq_all[wid];
    const int ni = lc_items(n);
    const G1Affine *it = items + (size_t)g * ni;
    const int8_t *dg = digits + (size_t)g * ni * LC_DIG_STRIDE;
    const bool in_lds = 40 * n + 14 <= LC_TASKS * LC_LDS_LIST;
    // the wave's lists, back to back in the order its lanes walk them (at most 7 * 2 (2n+1) + 6 * 2n = 40 n + 14 entries)
    uint16_t *lists = in_lds ? lists_all[wid] : glists + ((size_t)blockIdx.x * 4 + wid) * LC_TASKS * lc_list_stride(n);
    LcSlot *out = S + (size_t)g * 2 * LC_WINDOWS * LC_BUCKETS;      // bucket sums [class][window][bucket - 1]; weighted by the Horner kernel
    auto task_window = [&](int tk) { return tk < n1 ? c1w0 + tk : c0w0 + (tk - n1); };
    auto slot_of = [&](int L) { const int tk = L / LC_BUCKETS; return ((tk < n1 ? 1 : 0) * LC_WINDOWS + task_window(tk)) * LC_BUCKETS + (L % LC_BUCKETS); };
    auto len_of = [&](int L) { return cnt[L / LC_BUCKETS][(L % LC_BUCKETS) + 1]; };
    for (int q = lane; q < (LC_BUCKETS + 1) * LC_TASKS; q += 64) cnt[q / (LC_BUCKETS + 1)][q % (LC_BUCKETS + 1)] = 0;
    __syncthreads();
#pragma unroll 1
    for (int tk = 0; tk < ntasks; tk++) {
        const int w = task_window(tk), lo = tk < n1 ? 2 * n : 0, hi = tk < n1 ? ni : 2 * n;      // item range of the task's class (terms t < n are class 0)
        for (int j = lo + lane; j < hi; j += 64) { const int d = dg[(size_t)j * LC_DIG_STRIDE + w]; if (d) atomicAdd(&cnt[tk][d < 0 ? -d : d], 1); }
    }
    __syncthreads();
    // rank the lists by length (ties by index): order[rank] = list; an empty list is the point at infinity
#pragma unroll 1
    for (int L = lane; L < nlists; L += 64) {
        const int len = len_of(L);
        int rank = 0;
        for (int M = 0; M < nlists; M++) { const int lm = len_of(M); rank += (lm > len) || (lm == len && M < L); }
        order[rank] = (uint8_t)L;
        if (len == 0) { if (keep_raw) out[slot_of(L)].raw = g1x_inf(); else out[slot_of(L)].jac = g1_inf(); }
    }
    __syncthreads();
    // deal: rank i to lane i, then every further list to the least loaded lane (ties: lowest lane)
    int total = 0, nmine = 0;
    {
        const int L = order[lane], len = len_of(L);
        if (len > 0) { seq[lane][0] = (uint8_t)L; total = len; nmine = 1; }
    }
#pragma unroll 1
    for (int r = 64; r < nlists; r++) {
        const int L = order[r], len = len_of(L);                 // the same for every lane
        if (len == 0) break;
        int key = nmine >= LC_MAX_LISTS ? 0x7fffffff : (total << 6) | lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int v = __shfl_xor(key, off); key = v < key ? v : key; }
        if (lane == (key & 63) && nmine < LC_MAX_LISTS) { seq[lane][nmine++] = (uint8_t)L; total += len; }
    }
    int lane_start = total;                           // exclusive prefix sum over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(lane_start, off); if (lane >= off) lane_start += v; }
    lane_start -= total;
    {
        int run = lane_start;
#pragma unroll 1
        for (int k = 0; k < nmine; k++) { const int L = seq[lane][k]; cursor[L / LC_BUCKETS][(L % LC_BUCKETS) + 1] = run; run += len_of(L); }
    }
    __syncthreads();
#pragma unroll 1
    for (int tk = 0; tk < ntasks; tk++) {
        const int w = task_window(tk), lo = tk < n1 ? 2 * n : 0, hi = tk < n1 ? ni : 2 * n;
        for (int j = lo + lane; j < hi; j += 64) {
            const int d = dg[(size_t)j * LC_DIG_STRIDE + w];
            if (d) { const int pos = atomicAdd(&cursor[tk][d < 0 ? -d : d], 1); lists[pos] = (uint16_t)(j | (d < 0 ? 0x8000 : 0)); }
        }
    }
    __threadfence_block();                       // the global-slab form of the lists is read back by other lanes of this wave
    __syncthreads();
    int tmax = total, kmax = nmine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int v = __shfl_xor(tmax, off), u = __shfl_xor(kmax, off); tmax = v > tmax ? v : tmax;
            kmax = u > kmax ? u : kmax; }
    {
        G1X accx = g1x_inf(); bool started = false;  // lazy extended-Jacobian accumulator: 8M + 2S per item, no reductions
        // the list entry two steps ahead and the point one step ahead are in flight during an addition (with the lists in the
        // global slab each step would otherwise wait for two dependent loads)
        const uint16_t *lst = lists + lane_start;
        auto entry = [&](int q) -> uint32_t { return q < total ? (in_lds ? (uint32_t)lst[q] : (uint32_t)__builtin_nontemporal_load(lst + q)) : 0u; };
        uint32_t v0 = entry(0), v1 = entry(1);
        G1Affine pn = it[v0 & 0x7fff];
        int k = 0, cur = nmine ? (int)seq[lane][0] : 0, end = nmine ? len_of(cur) : 0;      // current list and the position after its last entry
#pragma unroll 1
        for (int q = 0; q < tmax; q++) {
            if (q < total) {
                G1Affine p = pn;
                const uint32_t v = v0;
                v0 = v1; v1 = entry(q + 2);
                pn = it[v0 & 0x7fff];                // next point (index 0 when past the end: a harmless in-range load)
                if (v & 0x8000) fp_neg(p.y, p.y);
                g1x_add_mixed_lazy(accx, started, p);
                if (q + 1 == end) {                  // the list ends here: park the raw accumulator, start the next list
                    if (!started) accx = g1x_inf();  // (the items cancelled out: all-zero limbs, which g1x_is_inf sees after canonicalisation)
                    out[slot_of(cur)].raw = accx;
                    started = false;
                    k++;
                    if (k < nmine) { cur = seq[lane][k]; end += len_of(cur); }
                }
            }
        }
    }
    // every lane turns the accumulators it parked into canonical Jacobian points, in place -- unless the consumer takes them as they
    // are (keep_raw: k_lc_wsum adds in the same lazy extended-Jacobian coordinates)
    if (keep_raw) return;
#pragma unroll 1
    for (int k = 0; k < kmax; k++) {
        if (k >= nmine) continue;
        LcSlot *sl = out + slot_of(seq[lane][k]);
        G1X raw = sl->raw, cx;
        g1x_from_lazy(cx, raw, true);
        G1Jac acc; g1x_to_jac(acc, cx);
        sl->jac = acc;
    }
}

__device__ __forceinline__ G1Jac g1_shfl_down16(const G1Jac &v, int delta) {     // within segments of 16 lanes
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_down(v.x.l[i], delta, 16); r.y.l[i] = __shfl_down(v.y.l[i], delta, 16); r.z.l[i] = __shfl_down(v.z.l[i],
            delta, 16); }
    return r;
}
// 256-thread workgroups: four waves, one per SIMD of the CU the workgroup lands on (64-thread workgroups of this latency-bound chain
// were placed two to a SIMD while other SIMDs idled: 2.9 instead of 1.9 ms per 2048 batches); the waves share nothing.
__global__ void __launch_bounds__(256) k_lc_horner(const LcSlot *S, int groups, PairPt *pair_pts) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;         // (batch, class, bucket - 1): 16-lane segments never straddle a wave
    const int lane = threadIdx.x & 63;
    const bool live = id < 2 * LC_BUCKETS * groups;
    const int gc = live ? id / LC_BUCKETS : 0, bi = id % LC_BUCKETS;     // gc = 2 g + class
    const LcSlot *s = S + (size_t)gc * LC_WINDOWS * LC_BUCKETS;
    G1Jac acc = live ? s[(size_t)(LC_WINDOWS - 1) * LC_BUCKETS + bi].jac : g1_inf();
    // lazy chain (g1.h): no reductions until the end; one doubling body, one addition body
#pragma unroll 1
    for (int k = LC_BITS * (LC_WINDOWS - 1) - 1; k >= 0; k--) {
        g1_dbl_lazy(acc, acc);
        if (k % LC_BITS == 0) { G1Jac v = live ? s[(size_t)(k / LC_BITS) * LC_BUCKETS + bi].jac : g1_inf(); g1_add_lazy(acc, acc, v); }
    }
    g1_canon_lazy(acc, acc);
    // sum_b b * A_b = sum_k T_k with the suffix sums T_k = sum_{b >= k} A_b: 4-step suffix scan over the 16 lanes, 4-step butterfly
    G1Jac r = acc;
#pragma unroll 1
    for (int off = 1; off < LC_BUCKETS; off <<= 1) {
        G1Jac o = g1_shfl_down16(r, off), t;
        g1_add(t, r, o);
        if ((lane % LC_BUCKETS) + off < LC_BUCKETS) r = t;
    }
#pragma unroll 1
    for (int off = 1; off < LC_BUCKETS; off <<= 1) { G1Jac o = g1_shfl_xor(r, off); g1_add(r, r, o); }
    if (!live || bi != 0) return;
    PairPt a; pairpt_from_jac(a, r, (gc & 1) == 0);               // pairings_verify negates its first G1 argument (utils.rs:198-201)
    pair_pts[gc] = a;
}

// Many batches (issue-bound): the 16 chains per class above walk 125 doublings each -- 4000 doublings per batch.  Weighting the
// buckets first leaves ONE chain per class:
//   k_lc_wsum    one lane per (batch, class, window):  W = sum_b b B[b]  by running sums (acc += B[b]; W += acc, b = 16 .. 1)
//   k_lc_hchain_quad  one DPP quad per (batch, class): Horner over the 26 W's (5 doublings + 1 addition each), to affine
// ~200 k wave instructions per batch instead of ~390 k; the dependent chain is ~20 % longer, so the form above stays for fewer batches.
__global__ void __launch_bounds__(256, 2) k_lc_wsum(const LcSlot *S, int groups, G1Jac *W) {
    // the running total is parked in LDS between its additions (limb-major: conflict-free): with acc, the total, the bucket being
    // added and the temporaries of an addition all in registers the kernel spills 68 VGPRs to scratch
    __shared__ uint32_t tot[4 * NFP][256];
    const int id = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;         // (batch, class, window)
    if (id >= 2 * LC_WINDOWS * groups) return;                    // (no barrier below: every lane keeps to its own LDS column)
    const LcSlot *s = S + (size_t)id * LC_BUCKETS;                // the bucket kernel's raw (lazy extended-Jacobian) sums; all-zero = infinity
    // Leading empty buckets are COMMON, not rare: the top window's digit has 3 value bits and a carry, so its buckets 9 .. 16 are empty in every
    // batch, and every wave holds two or three top-window lanes.  g1x_add_lazy2 sends an operand at infinity through the canonical complete
    // addition -- a divergent branch, so the whole wave ran it: 18 of a wave's 30 additions.  A lane now starts at its highest non-empty bucket
    // and sits out the steps above it (the sums are the same: infinity + infinity); an empty bucket further down still takes the complete addition.
    int start = -1;
#pragma unroll 1
    for (int b = LC_BUCKETS - 1; b >= 0 && start < 0; b--) {
        const Fp zz = s[b].raw.zz;
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NFP; i++) o |= zz.l[i];
        if (o) start = b;
    }
    if (start < 0) { W[id] = g1_inf(); return; }                 // nothing in this window (no barrier below)
    G1X acc = s[start].raw;
#pragma unroll
    for (int i = 0; i < NFP; i++) { tot[i][tid] = acc.x.l[i]; tot[NFP + i][tid] = acc.y.l[i]; tot[2 * NFP + i][tid] = acc.zz.l[i];
            tot[3 * NFP + i][tid] = acc.zzz.l[i]; }
#pragma unroll 1
    for (int b = LC_BUCKETS - 2; b >= 0; b--) {
        if (b >= start) continue;
        g1x_add_lazy2(acc, acc, s[b].raw);
        asm volatile("" ::: "memory");                            // the total is fetched only now ...
        G1X sum;
#pragma unroll
        for (int i = 0; i < NFP; i++) { sum.x.l[i] = tot[i][tid]; sum.y.l[i] = tot[NFP + i][tid]; sum.zz.l[i] = tot[2 * NFP + i][tid];
                sum.zzz.l[i] = tot[3 * NFP + i][tid]; }
        g1x_add_lazy2(sum, sum, acc);
#pragma unroll
        for (int i = 0; i < NFP; i++) { tot[i][tid] = sum.x.l[i]; tot[NFP + i][tid] = sum.y.l[i]; tot[2 * NFP + i][tid] = sum.zz.l[i];
                tot[3 * NFP + i][tid] = sum.zzz.l[i]; }
        asm volatile("" ::: "memory");                            // ... and is out of the registers before the next bucket comes in
    }
    G1X sum, c;
#pragma unroll
    for (int i = 0; i < NFP; i++) { sum.x.l[i] = tot[i][tid]; sum.y.l[i] = tot[NFP + i][tid]; sum.zz.l[i] = tot[2 * NFP + i][tid];
            sum.zzz.l[i] = tot[3 * NFP + i][tid]; }
    g1x_from_lazy(c, sum, true);
    G1Jac j; g1x_to_jac(j, c);
    W[id] = j;
}
// The same chain walked by a DPP quad per (batch, class) (g1_quad.h): 256 waves of pure latency become 1024, each 3 + 3 + .. product stages
// deep per window instead of 5 x 7 + 16 products on one lane (1.4 -> 0.85 ms per 8192 batches).
__global__ void __launch_bounds__(256) k_lc_hchain_quad(const G1Jac *W, int groups, PairPt *pair_pts) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, role = tid & 3;
    const bool live = (tid >> 2) < 2 * groups;
    const int gc = live ? (tid >> 2) : 2 * groups - 1;            // idle quads redo the last chain (every lane takes part in the DPP moves)
    const G1Jac *w = W + (size_t)gc * LC_WINDOWS;
    G1Jac acc = w[LC_WINDOWS - 1];
    // one loop, one inlined instance of each quad routine
#pragma unroll 1
    for (int k = LC_BITS * (LC_WINDOWS - 1) - 1; k >= 0; k--) {
        g1_dbl_quad(acc, role);
        if (k % LC_BITS == 0) { const G1Jac v = w[k / LC_BITS]; g1_add_quad(acc, acc, v, role); }      // (uniform branch)
    }
    if (!live || role != 0) return;
    G1Jac r; g1_canon_lazy(r, acc);
    PairPt a; pairpt_from_jac(a, r, (gc & 1) == 0);               // pairings_verify negates its first G1 argument (utils.rs:198-201)
    pair_pts[gc] = a;
}

// ------------------------------------------------------------------------------------------------ lincomb, pre-shifted form
// Latency form for FEW batches.  The three sums need the batch challenge r, and r needs every y_i, i.e. the whole SHA-256 chain
// of the blobs (3.7 ms for one batch) -- but the POINTS are inputs.  So while the hash runs, a side-stream kernel walks the
// doubling chain of every input point once and keeps  Q[pt][w] = 32^w P  for the 26 windows (k_ps_shift: 125 doublings per
// lane, the whole dependent chain of a scalar multiplication).  Once r is known the sums are pure bucket sums with NO doubling
// left:  sum_items sum_w d_w Q[item][w] = sum_b b (sum of the +-Q with |digit| = b):
//   k_lc_prep      (as above)  GLV split + signed 5-bit digits of the 2 (3n + 1) half-scalars
//   k_ps_buckets   four 256-thread workgroups per (batch, class), four buckets each: LDS counting sort of the (item, window)
//                  pairs of its buckets by |digit|; one wave per bucket adds up its share (~6 Jacobian additions per lane at
//                  n = 64), butterfly over the 64 lanes                              -> the 16 bucket sums
//   k_ps_weights   lane per (batch, class, b): the weights b over 16 lanes (suffix scan + butterfly), to affine.
// ~20 dependent additions after r instead of 125 doublings + 33 additions: 2.0 -> 0.5 ms for one 64-blob batch, and the
// shifting hides under the hash next to the point validation.  More total work than either other form (every point is doubled
// 125 times whatever its scalars), so it is used only while the card has idle SIMDs to give (fewer than 64 batches of <= 128 blobs; from 64 batches on the
// bucket form).
constexpr int PS_MAX_N = 128;                              // blobs per batch the LDS list is sized for
constexpr int PS_THREADS = 256;                            // one workgroup per (batch, class, bucket): 64 quads
__host__ __device__ inline int ps_points(int n) { return 2 * n + 1; }               // commitments, proofs, -G
// thread (g, pt): Q[w] = 32^w P for w = 0..25.  Point 2n of every batch is -G (the term -[sum r^i y_i] G).
// The chain starts from x ALONE, so it does not wait for the square root of the decompression (a second 0.6 ms chain; with the
// challenges hashed on the host nothing else hides it): with s = x^3 + 4 = y^2, the map (X, Y) -> (X / s, y Y / s^2) takes the curve
// E'': Y^2 = X^3 + 4 s^3 to E, and (s x, s^2) on E'' is the image of P = (x, y).  The doubling formulas of a = 0 curves do not involve the
// constant, so the chain runs on E'' from (s x, s^2, 1) unchanged, and a Jacobian point (X, Y, Z) of E'' IS the Jacobian point
// (X, Y, y Z) of E:  x = X / (y Z)^2 = (X / Z^2) / s  and  y = Y / (y Z)^3 = y (Y / Z^3) / s^2.  The consumer (k_ps_buckets)
// multiplies Z by y -- the sign of y included -- when it reads a shifted point; by then the decompression kernel has long finished.
// Inputs: validated affine points (pts != null: entry points without a stage 1) or the compressed bytes themselves (a bad encoding
// is treated as the point at infinity here; the decompression kernel raises the error).
// (the DPP-quad doubling and addition the chain is walked with: g1_quad.h)
constexpr int PS_SHIFT_THREADS = 256;
__global__ void __launch_bounds__(PS_SHIFT_THREADS) k_ps_shift(const G1Affine *pts, const uint8_t *cbytes, const uint8_t *pbytes, int stride, int n,
        int groups, G1Jac *shifts) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, np = ps_points(n);
    const int role = tid & 3;
    const bool live = (tid >> 2) < np * groups;
    const int id = live ? (tid >> 2) : np * groups - 1;           // idle quads redo the last point (every lane takes part in the DPP moves)
    const int g = id / np, pt = id % np;
    Fp x, s;
    bool inf = false;
    if (pt == 2 * n) {
        const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT;
        Fp y;
        for (int q = 0; q < NFP; q++) { x.l[q] = gx[q]; y.l[q] = gy[q]; }
        fp_sqr(s, y);
    } else if (pts) {
        const G1Affine p = pts[(size_t)g * 2 * n + pt];
        inf = g1a_is_inf(p);
        x = p.x; fp_sqr(s, p.y);
    } else {
        const uint8_t *src = pt < n ? cbytes + (size_t)stride * ((size_t)g * n + pt) : pbytes + (size_t)stride * ((size_t)g * n + (pt - n));
        uint8_t b[48];
        for (int k = 0; k < 48; k++) b[k] = src[k];
        bool large;
        x = fp_zero();
        if (g1_parse_compressed(x, inf, large, b)) inf = true;
        g1_curve_rhs(s, x);
    }
    G1Jac acc;
    fp_mul(acc.x, s, x); fp_sqr(acc.y, s); acc.z = fp_one();
    if (inf) acc = g1_inf();
    G1Jac *out = shifts + (size_t)id * LC_WINDOWS;
    if (live && role == 0) out[0] = acc;
#pragma unroll 1
    for (int k = 1; k <= LC_BITS * (LC_WINDOWS - 1); k++) {
        g1_dbl_quad(acc, role);
        if (k % LC_BITS == 0) {                                   // window boundary: lane r of the quad makes coordinate r canonical and stores it
            Fp c; fp_select(c, role == 1, acc.x, acc.y); fp_select(c, role == 2, c, acc.z);
            fp_canon64(c, c);
            Fp *dst = &out[k / LC_BITS].x + (role < 3 ? role : 0);
            if (live && role < 3) *dst = c;
        }
    }
}
// item j of a batch -> index of its point in the shift table (items 2t, 2t+1 belong to term t; see k_lc_prep)
__device__ __forceinline__ int ps_point_of_item(int j, int n) {
    const int t = j >> 1;
    return t < n ? n + t : t < 2 * n ? n + (t - n) : t < 3 * n ? t - 2 * n : 2 * n;
}
// Sums with "no term yet" as a flag instead of the point at infinity (see k_ps_buckets): acc (+)= o where `have` / `ho` say which of the two
// hold a sum.  The quad addition always runs (every lane takes part in its DPP moves) -- on fixed stand-ins for an absent operand, G and
// phi(G) = (beta x_G, y_G): two points that differ and are not each other's negatives, so the stand-in addition never takes the rare path.
struct PsDummies { G1Jac a, b; };
__device__ __forceinline__ PsDummies ps_dummies() {
    const uint32_t gx[NFP] = G1_GEN_X_INIT, gy[NFP] = G1_GEN_Y_INIT, bc[NFP] = FP_BETA_INIT;
    PsDummies d;
    Fp beta;
    for (int i = 0; i < NFP; i++) { d.a.x.l[i] = gx[i]; d.a.y.l[i] = gy[i]; beta.l[i] = bc[i]; }
    d.a.z = fp_one();
    d.b = d.a;
    fp_mul(d.b.x, d.a.x, beta);
    return d;
}
__device__ __forceinline__ void g1_select(G1Jac &r, bool take_b, const G1Jac &a, const G1Jac &b) {
    fp_select(r.x, take_b, a.x, b.x); fp_select(r.y, take_b, a.y, b.y); fp_select(r.z, take_b, a.z, b.z);
}
__device__ __forceinline__ void ps_add_present(G1Jac &acc, bool &have, const G1Jac &o, bool ho, const PsDummies &dum, int role) {
    G1Jac a2, b2, t;
    g1_select(a2, have, dum.a, acc);
    g1_select(b2, ho, dum.b, o);
    g1_add_quad(t, a2, b2, role);
    g1_select(t, !ho, t, acc);                                    // both present: the sum; only acc: acc
    g1_select(acc, have, o, t);                                   // acc absent: o (itself absent -- the point at infinity -- when ho is false)
    have = have || ho;
}
// One 256-thread workgroup per (batch, class, bucket): 64 DPP quads, each an accumulator of its own (g1_add_quad: an addition five
// products deep instead of sixteen).  A quad takes entries q, q + 64, ... of the bucket's list (~6.5 at n = 64), the 16 quads of a wave
// are summed by a shuffle butterfly (4 levels), the four waves through LDS (2 levels): ~13 quad additions in a row where the
// lane-per-accumulator form had ~6.5 + 6 lane additions three times as deep.  Writes the bucket sum (canonical).
constexpr int PS_QUADS = PS_THREADS / 4;
__global__ void __launch_bounds__(PS_THREADS) k_ps_buckets(const G1Jac *shifts, const G1Affine *pts, const int8_t *digits, int n, LcSlot *S) {
    // item | window << 10 | sign << 15 of this bucket; sized for every pair of the class landing here (27 KB)
    __shared__ uint16_t list[2 * (2 * PS_MAX_N + 1) * LC_WINDOWS];
    __shared__ int cnt;
    __shared__ G1Jac wsum[4];
    __shared__ int whave[4];
    const int bk = blockIdx.x % LC_BUCKETS, gc = blockIdx.x / LC_BUCKETS, g = gc >> 1, cls = gc & 1, tid = threadIdx.x;
    const int role = tid & 3, quad = tid >> 2, lane = tid & 63, wid = tid >> 6;
    const int ni = lc_items(n), lo = cls == 0 ? 0 : 2 * n, hi = cls == 0 ? 2 * n : ni;      // the class's items (terms t < n are class 0)
    const int8_t *dg = digits + (size_t)g * ni * LC_DIG_STRIDE;
    const G1Jac *sh = shifts + (size_t)g * ps_points(n) * LC_WINDOWS;
    const int npairs = (hi - lo) * LC_WINDOWS;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int q = tid; q < npairs; q += PS_THREADS) {
        const int j = lo + q / LC_WINDOWS, w = q % LC_WINDOWS;
        const int d = dg[(size_t)j * LC_DIG_STRIDE + w];
        if ((d < 0 ? -d : d) == bk + 1) { const int pos = atomicAdd(&cnt, 1); list[pos] = (uint16_t)((j - lo) | (w << 10) | (d < 0 ? 0x8000 : 0)); }
    }
    __syncthreads();
    const int c = cnt;
    const uint32_t bc[NFP] = FP_BETA_INIT;
    Fp beta; for (int i = 0; i < NFP; i++) beta.l[i] = bc[i];
    auto fetch = [&](int j, int w, bool neg) -> G1Jac {
        const int pt = ps_point_of_item(j, n);
        G1Jac p = sh[(size_t)pt * LC_WINDOWS + w];
        {   // the shift table holds points of E'' (k_ps_shift): Z picks up the y of the input point (zero for the point at infinity)
            Fp y0;
            if (pt < 2 * n) y0 = pts[(size_t)g * 2 * n + pt].y;
            else { const uint32_t gy[NFP] = G1_GEN_Y_INIT; for (int q = 0; q < NFP; q++) y0.l[q] = gy[q]; fp_neg(y0, y0); }
            Fp zz; fp_mul(zz, p.z, y0); p.z = zz;
        }
        if (j & 1) { Fp bx; fp_mul(bx, p.x, beta); p.x = bx; neg = !neg; }       // the odd item of a term is -phi(P) = (beta x, -y)
        if (neg) fp_neg(p.y, p.y);
        return p;
    };
    // ONE loop with ONE inlined instance of the addition (hipcc 7.2 has miscompiled kernels with several inlined instances of the G1
    // routines in a row): the operand of a step is, in turn, the quad's next list entry (steps 0 .. rounds - 1;
    // none once its share is used up), the partner quad's sum in the wave (4 butterfly steps), and -- behind a workgroup barrier --
    // the wave sums: quads 0, 1 of every wave add (w0 + w1), (w2 + w3), then their sum.
    // A sum that has no term yet (every accumulator at step 0, a quad whose share is used up, ...) is NOT carried as the point at infinity:
    // g1_add_quad sends an operand at infinity through the complete addition on the whole wave (~3x an addition; at n = 64 that was step 0 and
    // the last list step of every workgroup: 40 of the kernel's 144 us).  `have` / `ho` say whether acc / o hold a sum; an absent operand is
    // replaced by a fixed point (G, phi(G)) and the result of that addition is dropped.
    const int rounds = (c + PS_QUADS - 1) / PS_QUADS;             // the same for every thread of the workgroup
    const PsDummies dum = ps_dummies();
    G1Jac acc = g1_inf();
    bool have = false;
#pragma unroll 1
    for (int step = 0; step < rounds + 6; step++) {
        G1Jac o = g1_inf();
        bool ho = false;
        if (step < rounds) {
            const int q = quad + step * PS_QUADS;
            if (q < c) { const uint32_t v = list[q]; o = fetch(lo + (int)(v & 0x3ff), (int)((v >> 10) & 31), (v & 0x8000) != 0); ho = true; }
        } else if (step < rounds + 4) {
            o = g1_shfl_xor(acc, 4 << (step - rounds)); ho = __shfl_xor((int)have, 4 << (step - rounds)) != 0;
        } else if (step == rounds + 4) {
            if (lane == 0) { wsum[wid] = acc; whave[wid] = have ? 1 : 0; }
            __syncthreads();                                      // (uniform: every thread reaches this step)
            acc = wsum[2 * (quad & 1)]; have = whave[2 * (quad & 1)] != 0; o = wsum[2 * (quad & 1) + 1]; ho = whave[2 * (quad & 1) + 1] != 0;
        } else {
            o = g1_shfl_xor(acc, 4); ho = __shfl_xor((int)have, 4) != 0;
        }
        ps_add_present(acc, have, o, ho, dum, role);
    }
    if (tid == 0) { g1_canon_lazy(acc, acc); S[(size_t)gc * LC_BUCKETS + bk].jac = acc; }      // (no term at all: acc is still the point at infinity)
}
// The weights of the bucket sums, one wave per (batch, class): quad b of the wave holds S_{b+1};  sum_b b * S_b = sum_k T_k with the suffix
// sums T_k = sum_{b >= k} S_b (as in k_lc_horner): 4-step suffix scan + 4-step butterfly of quad additions, to the pairing's form.
__global__ void __launch_bounds__(64) k_ps_weights(const LcSlot *S, int groups, PairPt *pair_pts) {
    const int gc = blockIdx.x, lane = threadIdx.x, role = lane & 3, bi = lane >> 2;
    G1Jac r = S[(size_t)gc * LC_BUCKETS + bi].jac;
    bool hr = !fp_is_zero(r.z);                                   // (canonical: an empty bucket is the point at infinity)
    const PsDummies dum = ps_dummies();
    // one loop, one inlined addition (see k_ps_buckets): steps 0..3 the suffix scan (a quad adds only while a partner exists: without one the
    // shuffle hands it its own sum, and P + P went through the complete addition on the whole wave -- 80 of the kernel's 161 us), steps 4..7
    // the butterfly
#pragma unroll 1
    for (int step = 0; step < 8; step++) {
        const int off = 1 << (step & 3);
        const G1Jac o = step < 4 ? g1_shfl_down_w(r, 4 * off) : g1_shfl_xor(r, 4 * off);
        bool ho = (step < 4 ? __shfl_down((int)hr, 4 * off, 64) : __shfl_xor((int)hr, 4 * off)) != 0;
        if (step < 4 && bi + off >= LC_BUCKETS) ho = false;
        ps_add_present(r, hr, o, ho, dum, role);
    }
    if (lane != 0) return;
    g1_canon_lazy(r, r);
    PairPt a; pairpt_from_jac(a, r, (gc & 1) == 0);               // pairings_verify negates its first G1 argument (utils.rs:198-201)
    pair_pts[gc] = a;
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_validate_points(const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, int n_per_group, G1Affine *d_pts, int *d_err,
                            hipStream_t st, int stride) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_validate_points, dim3((2 * n_total + 255) / 256), dim3(256), 0, st, d_commitments, d_proofs, n_total, n_per_group, d_pts, d_err,
            stride);
}
// Test / audit readback of stage 2 (tests/test_gpu_parity.py): per batch  r (32 bytes big-endian, utils.rs:472) | proof_lincomb (48) |
// rhs (48), the latter two ZCash-compressed like bytes_from_g1 (utils.rs:221-227).  pair_pts holds -proof_lincomb (utils.rs:198-201).
__global__ void __launch_bounds__(64) k_dump_intermediates(const uint32_t *scal_a, const PairPt *pair_pts, int n, int groups, uint8_t *out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = id >> 1, which = id & 1;
    if (g >= groups) return;
    uint8_t *o = out + 128 * (size_t)g;
    G1Affine p; pairpt_to_affine(p, pair_pts[2 * (size_t)g + which]);
    if (which == 0 && !g1a_is_inf(p)) fp_neg(p.y, p.y);
    uint8_t b[48]; g1_compress_affine(b, p);
    for (int k = 0; k < 48; k++) o[32 + 48 * which + k] = b[k];
    if (which == 0) {
        const uint32_t *r = scal_a + 8 * ((size_t)g * n + (n > 1 ? 1 : 0));      // a_1 = r (a_0 = 1)
        for (int k = 0; k < 8; k++) { const uint32_t v = r[7 - k]; o[4 * k] = (uint8_t)(v >> 24); o[4 * k + 1] = (uint8_t)(v >> 16);
                o[4 * k + 2] = (uint8_t)(v >> 8); o[4 * k + 3] = (uint8_t)v; }
    }
}
void launch_dump_intermediates(const uint32_t *d_scal_a, const PairPt *d_pair_pts, int n_per_group, int groups, uint8_t *d_out, hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_dump_intermediates, dim3((2 * groups + 63) / 64), dim3(64), 0, st, d_scal_a, d_pair_pts, n_per_group, groups, d_out);
}
void launch_decompress_points(const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, int n_per_group, G1Affine *d_pts, int *d_err, hipStream_t st,
        int stride) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_decompress_points, dim3((2 * n_total + 63) / 64), dim3(64), 0, st, d_commitments, d_proofs, n_total, n_per_group, d_pts, d_err,
            stride);
}
void launch_subgroup_points(const G1Affine *d_pts, int n_total, int n_per_group, int *d_err, hipStream_t st, int commitments_only) {
    if (n_total <= 0) return;
    if (2 * n_total <= 1024) {          // few points: four lanes per point (latency form)
        hipLaunchKernelGGL(k_subgroup_points_quad, dim3((8 * n_total + 255) / 256), dim3(256), 0, st, d_pts, 2 * n_total, n_per_group, d_err, commitments_only);
        return;
    }
    hipLaunchKernelGGL(k_subgroup_points, dim3((2 * n_total + 63) / 64), dim3(64), 0, st, d_pts, 2 * n_total, n_per_group, d_err, commitments_only);
}
void launch_subgroup_ladder_from_x(const uint8_t *d_commitments, int stride, int n, G1Jac *d_T, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_subgroup_ladder_from_x_quad, dim3((4 * n + 255) / 256), dim3(256), 0, st, d_commitments, stride, n, d_T);
}
void launch_subgroup_finish(const G1Affine *d_pts, const G1Jac *d_T, int n, int *d_err, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_subgroup_finish, dim3((n + 63) / 64), dim3(64), 0, st, d_pts, d_T, n, d_err);
}
void launch_points_from_records(const uint8_t *d_records, int n_total, int n_per_group, G1Affine *d_pts, int *d_err, hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_points_from_records, dim3((2 * n_total + 63) / 64), dim3(64), 0, st, d_records, n_total, n_per_group, d_pts, d_err);
}
void launch_lincomb(const G1Affine *d_pts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c, int n_per_group,
                    int groups, G1Jac *d_partials, PairPt *d_pair_pts, hipStream_t st) {
    if (groups <= 0) return;
    const int wpg = lincomb_waves_per_group(n_per_group);
    // the per-wave window tables follow the partial sums in the same scratch allocation (lincomb_partials_bytes)
    uint32_t *d_wtabs = reinterpret_cast<uint32_t *>(d_partials + 2 * (size_t)wpg * groups);
    hipLaunchKernelGGL(k_lincomb_terms, dim3(groups * wpg), dim3(64), 0, st, d_pts, d_scal_a, d_scal_b, d_scal_c, n_per_group, d_partials, d_wtabs);
    hipLaunchKernelGGL(k_lincomb_finish, dim3(groups), dim3(64), 0, st, d_partials, n_per_group, d_pair_pts);
}
// n_per_group == 1, many groups: 16 checks per wave; d_scratch: lincomb_single_bytes(groups) (the per-wave window tables)
size_t lincomb_single_bytes(int groups) { return sizeof(uint32_t) * W4_ENTRIES * 3 * NFP * 64 * (size_t)((groups + 15) / 16); }
void launch_lincomb_single(const G1Affine *d_pts, const uint32_t *d_scal_b, const uint32_t *d_scal_c, int groups, void *d_scratch, PairPt *d_pair_pts,
                           hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_lincomb_single, dim3((groups + 15) / 16), dim3(64), 0, st, d_pts, d_scal_b, d_scal_c, groups, d_pair_pts,
            reinterpret_cast<uint32_t *>(d_scratch));
}
// entries of the global list slab (16-byte aligned count; 0 when the lists fit the LDS)
static size_t lc_glists_entries(int n_per_group, int groups) {
    if (40 * (size_t)n_per_group + 14 <= (size_t)LC_TASKS * LC_LDS_LIST) return 0;
    return (((size_t)groups * 4 * LC_TASKS * lc_list_stride(n_per_group)) + 7) & ~(size_t)7;
}
void launch_lincomb_buckets(const G1Affine *d_pts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c, int n_per_group,
                            int groups, void *d_scratch, PairPt *d_pair_pts, hipStream_t st, int stage, int chain_from) {
    if (groups <= 0) return;
    const size_t ni = (size_t)lc_items(n_per_group) * groups;
    G1Affine *items = reinterpret_cast<G1Affine *>(d_scratch);
    LcSlot *S = reinterpret_cast<LcSlot *>(items + ni);
    int8_t *digits = reinterpret_cast<int8_t *>(S + (size_t)2 * LC_WINDOWS * LC_BUCKETS * groups);
    uint16_t *glists = reinterpret_cast<uint16_t *>(digits + ((ni * LC_DIG_STRIDE + 255) & ~(size_t)255));
    const int nt = 3 * n_per_group + 1;
    if (stage == 0 || stage == 1) hipLaunchKernelGGL(k_lc_prep, dim3(groups * ((nt + 63) / 64)), dim3(64), 0, st, d_pts, d_scal_a, d_scal_b, d_scal_c,
            n_per_group, items, digits);
    if (stage == 0 || stage == 2) hipLaunchKernelGGL(k_lc_buckets, dim3(groups), dim3(256), 0, st, items, digits, n_per_group, S, glists,
            groups >= chain_from ? 1 : 0);
    if (stage == 0 || stage == 3) {
        if (groups >= chain_from) {
            G1Jac *W = reinterpret_cast<G1Jac *>(glists + lc_glists_entries(n_per_group, groups));
            hipLaunchKernelGGL(k_lc_wsum, dim3((2 * LC_WINDOWS * groups + 255) / 256), dim3(256), 0, st, S, groups, W);
            hipLaunchKernelGGL(k_lc_hchain_quad, dim3((8 * groups + 255) / 256), dim3(256), 0, st, W, groups, d_pair_pts);
        } else hipLaunchKernelGGL(k_lc_horner, dim3((2 * LC_BUCKETS * groups + 255) / 256), dim3(256), 0, st, S, groups, d_pair_pts);
    }
}
size_t lincomb_buckets_scratch_bytes(int n_per_group, int groups) {
    const size_t ni = (size_t)lc_items(n_per_group) * groups;
    const size_t lists = lc_glists_entries(n_per_group, groups) * sizeof(uint16_t) + (size_t)2 * LC_WINDOWS * groups * sizeof(G1Jac);    // + the window sums W
    return ni * sizeof(G1Affine) + (size_t)2 * LC_WINDOWS * LC_BUCKETS * groups * sizeof(LcSlot) + ni * LC_DIG_STRIDE + 512 + lists;
}
bool lincomb_preshift_fits(int n_per_group, int groups) { return n_per_group >= 1 && n_per_group <= PS_MAX_N && groups >= 1 && groups < 64; }
size_t lincomb_preshift_bytes(int n_per_group, int groups) { return sizeof(G1Jac) * (size_t)ps_points(n_per_group) * LC_WINDOWS * groups; }
void launch_lincomb_preshift(const G1Affine *d_pts, int n_per_group, int groups, G1Jac *d_shifts, hipStream_t st) {
    if (groups <= 0) return;
    const int total = ps_points(n_per_group) * groups;
    hipLaunchKernelGGL(k_ps_shift, dim3((4 * total + PS_SHIFT_THREADS - 1) / PS_SHIFT_THREADS), dim3(PS_SHIFT_THREADS), 0, st, d_pts, (const uint8_t *)nullptr,
            (const uint8_t *)nullptr, 0, n_per_group, groups, d_shifts);
}
void launch_lincomb_preshift_bytes(const uint8_t *d_commitments, const uint8_t *d_proofs, int stride, int n_per_group, int groups, G1Jac *d_shifts,
        hipStream_t st) {
    if (groups <= 0) return;
    const int total = ps_points(n_per_group) * groups;
    hipLaunchKernelGGL(k_ps_shift, dim3((4 * total + PS_SHIFT_THREADS - 1) / PS_SHIFT_THREADS), dim3(PS_SHIFT_THREADS), 0, st, (const G1Affine *)nullptr,
            d_commitments, d_proofs, stride, n_per_group, groups, d_shifts);
}
// stage 0: everything; 1: the digits only (needs the decoded points and the r powers, not the shift table); 2: the sums
void launch_lincomb_preshifted(const G1Affine *d_pts, const G1Jac *d_shifts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c,
        int n_per_group,
                               int groups, void *d_scratch, PairPt *d_pair_pts, hipStream_t st, int stage) {
    if (groups <= 0) return;
    const size_t ni = (size_t)lc_items(n_per_group) * groups;
    G1Affine *items = reinterpret_cast<G1Affine *>(d_scratch);          // same scratch layout as the bucket form (lincomb_buckets_scratch_bytes)
    LcSlot *S = reinterpret_cast<LcSlot *>(items + ni);
    int8_t *digits = reinterpret_cast<int8_t *>(S + (size_t)2 * LC_WINDOWS * LC_BUCKETS * groups);
    const int nt = 3 * n_per_group + 1;
    if (stage == 0 || stage == 1) hipLaunchKernelGGL(k_lc_prep, dim3(groups * ((nt + 63) / 64)), dim3(64), 0, st, d_pts, d_scal_a, d_scal_b, d_scal_c,
            n_per_group, items, digits);
    if (stage == 1) return;
    hipLaunchKernelGGL(k_ps_buckets, dim3(2 * groups * LC_BUCKETS), dim3(PS_THREADS), 0, st, d_shifts, d_pts, digits, n_per_group, S);
    hipLaunchKernelGGL(k_ps_weights, dim3(2 * groups), dim3(64), 0, st, S, groups, d_pair_pts);
}
size_t lincomb_partials_bytes(int n_per_group, int groups) {
    const size_t waves = (size_t)lincomb_waves_per_group(n_per_group) * groups;
    return sizeof(G1Jac) * 2 * waves + sizeof(uint32_t) * W4_ENTRIES * 3 * NFP * 64 * waves;
}

}  // namespace kzg
