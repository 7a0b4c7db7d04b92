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

namespace kzg {

// ------------------------------------------------------------------------------------------------ points
// thread j < n_total: commitment j ; j >= n_total: proof j - n_total.
__global__ void __launch_bounds__(64) k_validate_points(const uint8_t *commitments, const uint8_t *proofs, int n_total, int n_per_group,
                                                         G1Affine *pts, int *err) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * n_total) return;
    const bool is_proof = j >= n_total;
    if (is_proof && !proofs) return;
    const int i = is_proof ? j - n_total : j;
    const uint8_t *src = (is_proof ? proofs : commitments) + 48 * (size_t)i;
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = src[k];
    G1Affine p;
    int rc = g1_decompress(p, b);
    if (rc == 0 && !g1a_is_inf(p) && !g1_in_subgroup(p)) rc = 3;     // infinity is accepted (utils.rs:298-301)
    const int g = i / n_per_group, k = i % n_per_group;
    if (rc != 0) { atomicOr(&err[g], ERR_BAD_POINT); p = g1a_inf(); }
    if (pts) pts[(size_t)g * 2 * n_per_group + (is_proof ? n_per_group + k : k)] = p;
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
// One 256-thread workgroup per batch.  Terms (3n + 1 scalar multiplications, each 256-bit double-and-add):
//   class 0:  a_i * proof_i                                   -> proof_lincomb         (kzg.rs:601)
//   class 1:  b_i * proof_i,  a_i * C_i,  c * (-G)            -> rhs                   (kzg.rs:603-622)
// then an LDS tree reduction per class.  Output: (-proof_lincomb, rhs) as affine points for the pairing.
constexpr int LINCOMB_THREADS = 256;
__global__ void __launch_bounds__(LINCOMB_THREADS) k_lincomb(const G1Affine *pts, const uint32_t *scal_a, const uint32_t *scal_b,
                                                               const uint32_t *scal_c, int n, G1Affine *pair_pts) {
    __shared__ G1Jac red[LINCOMB_THREADS];
    const int g = blockIdx.x, tid = threadIdx.x;
    const G1Affine *gp = pts + (size_t)g * 2 * n;        // [0,n) commitments, [n,2n) proofs
    G1Jac acc0 = g1_inf(), acc1 = g1_inf();
    for (int t = tid; t < 3 * n + 1; t += LINCOMB_THREADS) {
        G1Affine p; uint32_t k[8]; int cls;
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
        G1Jac m; g1_mul_words(m, p, k, 8);
        if (cls == 0) g1_add(acc0, acc0, m); else g1_add(acc1, acc1, m);
    }
    G1Jac total[2];
    for (int cls = 0; cls < 2; cls++) {
        red[tid] = cls == 0 ? acc0 : acc1;
        __syncthreads();
        for (int s = LINCOMB_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) { G1Jac a = red[tid], b = red[tid + s]; g1_add(a, a, b); red[tid] = a; }
            __syncthreads();
        }
        if (tid == 0) total[cls] = red[0];
        __syncthreads();
    }
    if (tid == 0) {
        G1Affine a0, a1;
        g1_to_affine(a0, total[0]); g1_to_affine(a1, total[1]);
        if (!g1a_is_inf(a0)) fp_neg(a0.y, a0.y);             // pairings_verify negates its first G1 argument (utils.rs:198-201)
        pair_pts[2 * (size_t)g] = a0;
        pair_pts[2 * (size_t)g + 1] = a1;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_validate_points(const uint8_t *d_commitments, const uint8_t *d_proofs, int n_total, int n_per_group, G1Affine *d_pts, int *d_err,
                            hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_validate_points, dim3((2 * n_total + 63) / 64), dim3(64), 0, st, d_commitments, d_proofs, n_total, n_per_group, d_pts, d_err);
}
void launch_points_from_records(const uint8_t *d_records, int n_total, int n_per_group, G1Affine *d_pts, int *d_err, hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_points_from_records, dim3((2 * n_total + 63) / 64), dim3(64), 0, st, d_records, n_total, n_per_group, d_pts, d_err);
}
void launch_lincomb(const G1Affine *d_pts, const uint32_t *d_scal_a, const uint32_t *d_scal_b, const uint32_t *d_scal_c, int n_per_group,
                    int groups, G1Affine *d_pair_pts, hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_lincomb, dim3(groups), dim3(LINCOMB_THREADS), 0, st, d_pts, d_scal_a, d_scal_b, d_scal_c, n_per_group, d_pair_pts);
}

}  // namespace kzg
