// kernels.h -- internal interface between the C-ABI host code (engine.h and the host translation units it lists) and the HIP kernel translation
// units (k_setup.hip, k_verify.hip, k_msm.hip, k_prove.hip).  Not installed; the public boundary is
// include/kzg355.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.h"
#include "tower.h"
#include "g1.h"
#include "pairing.h"
#include "sha256.h"
#include "pairing_coop.h"
#include "eval_core.h"

namespace kzg {

constexpr int N_FE = 4096;                 // FIELD_ELEMENTS_PER_BLOB (consts.rs:13)
constexpr int BLOB_BYTES = N_FE * 32;      // consts.rs:16
constexpr int RECORD_BYTES = 160;          // C | z | y | proof  (utils.rs:454-463)
constexpr int MSM_WINDOW_BITS = 8;
constexpr int MSM_WINDOWS = 32;            // 32 x 8 bits cover the 255-bit scalars
constexpr int MSM_BUCKETS = 128;           // signed digits in [-127, 128]
// wide-window form of the same MSM (k_msm_wide.hip): signed c-bit digits, every multiple 1..2^(c-1) tabulated; c is chosen when
// the handle is created (12: 22 windows, 23.6 GB; 13: 20 windows, 42.9 GB; 14: 19 windows, 81.6 GB)
// Shape of the fixed-base MSM table (k_msm_wide.hip): `windows` signed `bits`-bit windows of `rows` = 2^(bits-1) multiples each, except the
// top window, which takes an unsigned digit (no carry leaves it) and has rows_top rows.  glv = 0: the windows span the 256 bits of the scalar.
// glv = 1 (round 4): the scalar is split k = a + b x^2 (g1.h glv_split_fast; a, b < x^2 < 2^127.5) and the windows span 128 bits -- the
// SAME table serves both halves, because sum_i [b_i](-phi P_i) = -phi(sum_i [b_i] P_i): the rows of the b halves are added up as they
// are and the endomorphism is applied once, to their sum.  Half the table at the same number of rows per scalar, or (16-bit windows: 8
// windows per half) 16 rows per scalar in 143 GB where the 256-bit form took 17.45 rows in 155 GB.
struct WideShape { int bits, windows, rows, glv, rows_top; };
inline WideShape wide_shape(int bits, bool glv = false) {
    WideShape w; w.bits = bits; w.rows = 1 << (bits - 1); w.glv = glv ? 1 : 0;
    if (!glv) { w.windows = (256 + bits - 1) / bits; w.rows_top = w.rows; return w; }
    w.windows = (128 + bits - 1) / bits;
    // largest top digit: the top bits of x^2 - 1 = 0xac45a4010001a40200000000ffffffff (both halves are below x^2), plus the carry coming in
    const int shift = bits * (w.windows - 1);                     // >= 96 for every supported width
    const unsigned top = (unsigned)(0xac45a401u >> (shift - 96)) + 1u;
    w.rows_top = (int)((top + 256u) & ~255u);                     // (segments of 256 rows: k_wide_rows)
    return w;
}
inline size_t wide_rows_per_point(WideShape w) { return (size_t)(w.windows - 1) * w.rows + w.rows_top; }
constexpr int N_G2 = 65;

// error bits accumulated on the device; any bit => the call returns Err (reference: `?` on each step)
constexpr int ERR_BAD_POINT = 1;           // validate_kzg_g1 failed (utils.rs:282-310)
constexpr int ERR_NONCANONICAL_FR = 2;     // bytes_to_bls_field failed (utils.rs:262-275)
constexpr int ERR_SETUP_POINT = 4;         // load_trusted_setup: bad g1/g2 bytes (kzg.rs:863, 878)
constexpr int ERR_SETUP_MONOMIAL = 8;      // is_trusted_setup_in_lagrange_form (kzg.rs:823-826)

struct alignas(128) WideRow { Fp x, y; uint32_t pad[4]; };       // one affine point per 128-byte line

struct DeviceTables {
    int n_fe;                // FIELD_ELEMENTS_PER_BLOB of this handle: 4096 (mainnet) or a small power of two (minimal preset: 4)
    Fr *roots;               // [4096] bit-reversal order, Montgomery (kzg.rs:34)
    EvalPiece *eval_tab;     // [EVAL_TAB_PIECES] the node roots of k_eval's tree, three per group, levels 1..6, in 16-byte pieces (eval_core.h)
    WideShape wide;          // shape of wide_table
    WideRow *wide_table;        // [22][4096][2048] multiples m * 2^(12w) * g1_values[i], 23.6 GB; null: 8-bit bucket form only
    G1Affine *msm_table;     // [32][4096]: window w holds 2^(8w) * g1_values[i]; window 0 IS g1_values (kzg.rs:37)
    LineCoeff *lines;        // [3][68]: Miller-loop lines of G2_GENERATOR, setup g2[0], setup g2[1]
    int *lines_inf;          // [3] 1 if that G2 point is the point at infinity
    G1Affine *g1_first2;     // file-order g1[0], g1[1] (only for the Lagrange-form check)
    LineW *lines_w;          // [3][68]: the same lines in the w basis (pairing_coop.h)
    FrobTables *frob;        // w-basis Frobenius tables
    CoopInsn *pairing_prog;  // the pairing check as an instruction list (pairing_coop.h)
    int pairing_prog_len;
    int pairing_hard_start;  // index of the first instruction of the final exponentiation's hard part (k_pairing_hard12 runs the tail)
    CoopScheds *coop_scheds; // product, square, line-product and cyclotomic-square work schedules
};

// ---- k_setup.hip
int launch_setup(const uint8_t *d_g1_bytes, const uint8_t *d_g2_bytes, DeviceTables t, int *d_err, hipStream_t st);

// ---- k_verify.hip
// stage 1 (per blob): validate points, Fiat-Shamir challenge, barycentric evaluation -> 160-byte records
// d_proofs may be null (commitments only, e.g. compute_blob_kzg_proof's challenge step, kzg.rs:321)
void launch_validate_points(const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, int n_per_group,
                            G1Affine *d_pts /* [group][2*npg]: commitments then proofs; may be null */, int *d_err /* per group */,
                            hipStream_t st, int stride = 48 /* bytes between consecutive inputs: 48 packed, 160 inside records */);
// validate_kzg_g1 in two launches: decoding (-> points, error on a bad encoding / off-curve x) and the subgroup test (-> error only)
void launch_decompress_points(const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, int n_per_group, G1Affine *d_pts, int *d_err, hipStream_t st,
                              int stride = 48);
void launch_subgroup_points(const G1Affine *d_pts, int n_total, int n_per_group, int *d_err, hipStream_t st, int commitments_only = 0);
void launch_dump_intermediates(const uint32_t *d_scal_a, const PairPt *d_pair_pts, int n_per_group, int groups, uint8_t *d_out /* [groups][128] */,
        hipStream_t st);
// d_zpow (may be null): EVAL_ZPOWERS values per blob, z^2, z^4, .. z^4096 (Montgomery) -- what launch_eval needs beside z
void launch_challenges(const uint8_t *d_blobs, const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total,
                       Fr *d_z, Fr *d_zpow, uint8_t *d_records, hipStream_t st, int form = 0 /* 0 by size, 1 one wave, 2 two waves */);
// the same records / z from digests hashed on the host (32 bytes per blob, host_sha256.h)
void launch_challenges_from_digests(const uint8_t *d_digests, const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, Fr *d_z, Fr *d_zpow,
                                    uint8_t *d_records, hipStream_t st);
void launch_eval(const uint8_t *d_blobs, const Fr *d_z, const Fr *d_zpow, DeviceTables t, int n_total, int n_per_group, Fr *d_y /* may be null */,
                 uint8_t *d_records /* y written at +80; may be null */, int *d_err, hipStream_t st);
// stage 2 (per group of n records): points from records, r-powers, lincomb, pairing
// few commitments (compute_blob_kzg_proof): the subgroup ladder from x alone (beside the square root of the decoding), then the test once y is there
void launch_subgroup_ladder_from_x(const uint8_t *d_commitments, int stride, int n, G1Jac *d_T, hipStream_t st);
void launch_subgroup_finish(const G1Affine *d_pts, const G1Jac *d_T, int n, int *d_err, hipStream_t st);
void launch_points_from_records(const uint8_t *d_records, int n_total, int n_per_group, G1Affine *d_pts, int *d_err, hipStream_t st);
void launch_rpowers(const uint8_t *d_records, int n_per_group, int groups, int check_zy, uint32_t *d_scal_a, uint32_t *d_scal_b,
                    uint32_t *d_scal_c, int *d_err, hipStream_t st, int n_fe = N_FE /* the u64be(FIELD_ELEMENTS_PER_BLOB) field of the transcript */,
                    int lanes_from = 1024 /* batches from which the hash runs one lane per batch (k_rhash_lanes) */,
                    int have_digest = 0 /* 1: the transcript digests are already in d_scal_c (8 little-endian words per batch: hashed on the host) */);
void launch_lincomb(const G1Affine *d_pts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c,
                    int n_per_group, int groups, G1Jac *d_partials /* lincomb_partials_bytes() */, PairPt *d_pair_pts /* [group][2]: -proof_lincomb, rhs */,
                    hipStream_t st);
size_t lincomb_partials_bytes(int n_per_group, int groups);
// n_per_group == 1 with many groups (the *_many forms of the single-proof functions): rhs = C + [z] proof - [y] G, four lanes per check
void launch_lincomb_single(const G1Affine *d_pts, const uint32_t *d_scal_b, const uint32_t *d_scal_c, int groups, void *d_scratch /* lincomb_single_bytes() */,
                           PairPt *d_pair_pts, hipStream_t st);
size_t lincomb_single_bytes(int groups);
// bucket-method (Pippenger) form of the same sums, for many batches in flight; n_per_group <= 4096 (item indices are 15-bit)
void launch_lincomb_buckets(const G1Affine *d_pts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c,
                            int n_per_group, int groups, void *d_scratch /* lincomb_buckets_scratch_bytes() */, PairPt *d_pair_pts, hipStream_t st,
                            int stage = 0 /* 0: all three kernels; 1 prep, 2 buckets, 3 horner (per-kernel timing) */,
                            int chain_from = 6144 /* batches from which the tail is k_lc_wsum + k_lc_hchain instead of k_lc_horner */);
size_t lincomb_buckets_scratch_bytes(int n_per_group, int groups);
// pre-shifted form for few batches: launch_lincomb_preshift needs only the validated points (it can run beside the Fiat-Shamir
// hash), launch_lincomb_preshifted finishes once the r powers exist.  d_scratch: lincomb_buckets_scratch_bytes().
bool lincomb_preshift_fits(int n_per_group, int groups);
size_t lincomb_preshift_bytes(int n_per_group, int groups);
void launch_lincomb_preshift(const G1Affine *d_pts, int n_per_group, int groups, G1Jac *d_shifts, hipStream_t st);
// the same chains straight from the compressed inputs (x only: they do not wait for the square root of the decompression)
void launch_lincomb_preshift_bytes(const uint8_t *d_commitments, const uint8_t *d_proofs, int stride, int n_per_group, int groups, G1Jac *d_shifts,
        hipStream_t st);
void launch_lincomb_preshifted(const G1Affine *d_pts, const G1Jac *d_shifts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c,
                               int n_per_group, int groups, void *d_scratch, PairPt *d_pair_pts, hipStream_t st,
                                       int stage = 0 /* 1: digits only; 2: sums only */);
void launch_pairing(const PairPt *d_pair_pts, DeviceTables t, int groups, int *d_ok, hipStream_t st,
                    int two_wave_upto = 256 /* batches up to which the two Miller loops of a check run on two waves */,
                    Fp *d_f12 = nullptr /* groups * 12 Fp of scratch */,
                            int hard12_from = 0 /* batches from which the hard part runs twelve lanes per check; 0: never */,
                    int miller_segments = 0 /* few batches: segments per Miller loop, 2 waves each (1: the two-wave kernel; 0: the default, 2) */);
size_t pairing_f12_bytes(int groups);
void launch_pairing_lane(const PairPt *d_pair_pts, DeviceTables t, int groups, int *d_ok, hipStream_t st);   // one lane per batch (A/B, tests)

// ---- k_pairing.hip
void launch_lines_to_w(DeviceTables t, hipStream_t st);

// ---- k_msm_wide.hip
size_t wide_table_bytes(WideShape ws);
int build_wide_table(DeviceTables t, hipStream_t st);                 // needs t.msm_table; 0 on success
int msm_wide_partials_per_blob(int n);
// scalars from blobs (canonical check fused, err per blob) or, if d_scalars != null, from Montgomery field elements
void launch_msm_wide(const uint8_t *d_blobs, const Fr *d_scalars, DeviceTables t, int n, G1Jac *d_partials /* [n][msm_wide_partials_per_blob(n)] */,
                     int *d_err, hipStream_t st);
// ---- k_msm.hip
void launch_digits_from_blobs(const uint8_t *d_blobs, int n, uint8_t *d_digits /* [n][32][4096] */, int *d_err /* per blob */, hipStream_t st);
void launch_digits_from_fr(const Fr *d_scalars /* [n][4096] Montgomery */, int n, uint8_t *d_digits, hipStream_t st);
void launch_msm_bucket(const uint8_t *d_digits, DeviceTables t, int n, G1Jac *d_partials /* [n][32] */, hipStream_t st);
void launch_msm_finalize(const G1Jac *d_partials, int n, uint8_t *d_out48 /* [n][48] */, hipStream_t st, int ppb = 0 /* partials per blob; 0: bucket form */);

// ---- k_small.hip: handles with 4 <= FIELD_ELEMENTS_PER_BLOB <= 64 (minimal preset); t.msm_table holds the n bit-reversed points
constexpr int SMALL_N_MIN = 4, SMALL_N_MAX = 64;
void launch_lagrange_from_monomial(const uint8_t *d_mono, int n, uint8_t *d_out /* n*48 */, int *d_err, hipStream_t st);
void launch_setup_small(const uint8_t *d_g1_bytes, int n, DeviceTables t, int *d_err, hipStream_t st);
void launch_small_records(const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p /* may be null */, int n_total, int npg, DeviceTables t,
        Fr *d_z /* may be null */,
                          uint8_t *d_records /* may be null */, int *d_err /* per group */, hipStream_t st);
void launch_small_commit(const uint8_t *d_blobs, int n_blobs, DeviceTables t, uint8_t *d_out48, int *d_err /* per blob */, hipStream_t st);
// proofs at d_z (Montgomery, one per blob) or, if d_c != null, at each blob's own Fiat-Shamir challenge; d_y32 (32-byte y per blob) may be null
void launch_small_proof(const uint8_t *d_blobs, const uint8_t *d_c, const Fr *d_z, int n_blobs, DeviceTables t, uint8_t *d_out48, uint8_t *d_y32, int *d_err,
        hipStream_t st);

// ---- k_prove.hip
// quotient polynomial q(X) = (p(X) - y)/(X - z) in evaluation form (kzg.rs:461-523) for n blobs; also y.
// d_q: the quotient of every blob in the BLOB format (n x 131,072 bytes: 4096 canonical 32-byte big-endian integers; 16-byte aligned) -- the fixed-base MSM
// reads it like a blob; d_scratch: quotient_scratch_bytes(n) of device memory; form: 0 by size, 2 / 4 / 6 = 2^form leaves per lane.  Non-zero: a HIP call
// failed.
size_t quotient_scratch_bytes(int n);
int launch_quotient(const uint8_t *d_blobs, const Fr *d_z, DeviceTables t, int n, Fr *d_y, uint8_t *d_q /* [n][131072] */, void *d_scratch, int *d_err,
        hipStream_t st,
                    int form = 0);
void launch_fr_from_bytes(const uint8_t *d_in32, int n, Fr *d_out, int *d_err /* per element, ERR_NONCANONICAL_FR */, hipStream_t st);
void launch_fr_to_bytes(const Fr *d_in, int n, uint8_t *d_out32, hipStream_t st);
void launch_status_words(const int *d_err, const int *d_ok /* or null */, int32_t *d_words, int groups, hipStream_t st);

}  // namespace kzg
