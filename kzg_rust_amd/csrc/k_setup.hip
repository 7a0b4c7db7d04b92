// k_setup.hip -- load_trusted_setup on the device (reference src/kzg.rs:833-899): decompress the 4096 G1 and
// 65 G2 points, check Lagrange form with one pairing, expand the roots of unity, bit-reverse both tables, and
// build what the hot path sweeps: the per-window multiples of the G1 table (fixed-base MSM) and the Miller-loop
// line tables of the G2 points the verify path pairs against.
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"

namespace kzg {

__device__ __forceinline__ uint32_t brp12(uint32_t i) { return __brev(i) >> 20; }   // reverse_bits(i, 4096)  (kzg.rs:700-710)

// thread i: g1_bytes[i] -> window-0 table slot brp(i)   (kzg.rs:859-872 + 895-896).  On-curve only, NO subgroup check.
__global__ void __launch_bounds__(64) k_setup_g1(const uint8_t *g1_bytes, G1Affine *table0, G1Affine *first2, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_FE) return;
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = g1_bytes[48 * (size_t)i + k];
    G1Affine p;
    if (g1_decompress(p, b) != 0) { atomicOr(err, ERR_SETUP_POINT); p = g1a_inf(); }
    table0[brp12((uint32_t)i)] = p;
    if (i < 2) first2[i] = p;
}

// thread i < 65: g2_bytes[i] must decode (kzg.rs:874-887); only [0] and [1] are ever used afterwards (kzg.rs:418, 625, 818-820).
__global__ void __launch_bounds__(128) k_setup_g2(const uint8_t *g2_bytes, G2Affine *g2_first2, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_G2) return;
    uint8_t b[96];
    for (int k = 0; k < 96; k++) b[k] = g2_bytes[96 * (size_t)i + k];
    G2Affine q;
    if (g2_decompress(q, b) != 0) { atomicOr(err, ERR_SETUP_POINT); q.x = fp2_zero(); q.y = fp2_zero(); }
    if (i < 2) g2_first2[i] = q;
}

// thread q in {0: G2_GENERATOR (consts.rs:98-154), 1: setup g2[0], 2: setup g2[1]}
__global__ void __launch_bounds__(64) k_setup_lines(const G2Affine *g2_first2, LineCoeff *lines, int *lines_inf) {
    const int q = threadIdx.x;
    if (q >= 3) return;
    G2Affine Q;
    if (q == 0) {
        const uint32_t x0[NFP] = G2_GEN_X0_INIT, x1[NFP] = G2_GEN_X1_INIT, y0[NFP] = G2_GEN_Y0_INIT, y1[NFP] = G2_GEN_Y1_INIT;
        for (int k = 0; k < NFP; k++) { Q.x.c0.l[k] = x0[k]; Q.x.c1.l[k] = x1[k]; Q.y.c0.l[k] = y0[k]; Q.y.c1.l[k] = y1[k]; }
    } else Q = g2_first2[q - 1];
    const bool inf = g2a_is_inf(Q);
    lines_inf[q] = inf ? 1 : 0;
    if (!inf) precompute_lines(lines + (size_t)q * N_LINES, Q);
}

// is_trusted_setup_in_lagrange_form (kzg.rs:802-830): e(g1[1], g2[0]) == e(g1[0], g2[1]) means MONOMIAL form -> error.
__global__ void __launch_bounds__(64) k_setup_lagrange_check(const G1Affine *first2, const LineCoeff *lines, const int *lines_inf, int *err) {
    if (threadIdx.x != 0) return;
    G1Affine a = first2[1], b = first2[0];
    if (!g1a_is_inf(a)) fp_neg(a.y, a.y);
    if (lines_inf[1]) a = g1a_inf();
    if (lines_inf[2]) b = g1a_inf();
    Fp12 f;
    miller_loop_pair(f, lines + 1 * N_LINES, a, lines + 2 * N_LINES, b);
    if (final_exp_is_one(f)) atomicOr(err, ERR_SETUP_MONOMIAL);
}

// compute_roots_of_unity (kzg.rs:764-799): w = SCALE2_ROOT_OF_UNITY[12] = 7^((r-1)/4096); thread i stores w^i at brp(i).
__global__ void __launch_bounds__(256) k_setup_roots(Fr *roots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_FE) return;
    const uint32_t rootc[NFR] = FR_ROOT4096_INIT;
    Fr base; for (int k = 0; k < NFR; k++) base.l[k] = rootc[k];
    Fr acc = fr_one();
    for (int b = 11; b >= 0; b--) {
        fr_sqr(acc, acc);
        if ((i >> b) & 1) fr_mul(acc, acc, base);
    }
    roots[brp12((uint32_t)i)] = acc;
}

// k_eval's tree (eval_core.h): level l = 1..6, group m -> the roots of its three nodes, one flat table of EVAL_TAB_GROUPS.
__global__ void __launch_bounds__(256) k_setup_eval_tab(const Fr *roots, EvalPiece *tab) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= EVAL_TAB_GROUPS) return;
    const int level = e < EVAL_TAB_L2 ? 1 : e < EVAL_TAB_L3 ? 2 : e < EVAL_TAB_L4 ? 3 : e < EVAL_TAB_L5 ? 4 : e < EVAL_TAB_L6 ? 5 : 6;
    const int m = e - eval_tab_first(level);
    EvalPiece p[7];
    eval_group_pack(p, EvalGroup{roots[4 * m], roots[4 * m + 2], roots[2 * m]});
    for (int q = 0; q < 7; q++) tab[eval_tab_piece(e, q)] = p[q];
}

// Fixed-base precomputation: table[w][i] = 2^(8w) * g1_values[i] in affine form, w = 1..31 (window 0 is g1_values).
// One thread per point; 8 doublings + one inversion per window.
__global__ void __launch_bounds__(64) k_setup_msm_table(G1Affine *table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_FE) return;
    G1Affine cur = table[i];
    for (int w = 1; w < MSM_WINDOWS; w++) {
        G1Jac j; g1_from_affine(j, cur);
        for (int k = 0; k < MSM_WINDOW_BITS; k++) g1_dbl(j, j);
        g1_to_affine(cur, j);
        table[(size_t)w * N_FE + i] = cur;
    }
}

int launch_setup(const uint8_t *d_g1_bytes, const uint8_t *d_g2_bytes, DeviceTables t, int *d_err, hipStream_t st) {
    G2Affine *d_g2_first2 = nullptr;
    if (hipMalloc(&d_g2_first2, 2 * sizeof(G2Affine)) != hipSuccess) { (void)hipGetLastError(); return 1; }
    const bool mainnet = t.n_fe == N_FE;       // small handles: launch_setup_small (k_small.hip) has filled the G1 table and the roots
    if (mainnet) hipLaunchKernelGGL(k_setup_g1, dim3(N_FE / 64), dim3(64), 0, st, d_g1_bytes, t.msm_table, t.g1_first2, d_err);
    hipLaunchKernelGGL(k_setup_g2, dim3(1), dim3(128), 0, st, d_g2_bytes, d_g2_first2, d_err);
    hipLaunchKernelGGL(k_setup_lines, dim3(1), dim3(64), 0, st, d_g2_first2, t.lines, t.lines_inf);
    hipLaunchKernelGGL(k_setup_lagrange_check, dim3(1), dim3(64), 0, st, t.g1_first2, t.lines, t.lines_inf, d_err);
    if (mainnet) {
        hipLaunchKernelGGL(k_setup_roots, dim3(N_FE / 256), dim3(256), 0, st, t.roots);
        hipLaunchKernelGGL(k_setup_eval_tab, dim3((EVAL_TAB_GROUPS + 255) / 256), dim3(256), 0, st, t.roots, t.eval_tab);
        hipLaunchKernelGGL(k_setup_msm_table, dim3(N_FE / 64), dim3(64), 0, st, t.msm_table);
    }
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(d_g2_first2);
    return e == hipSuccess ? 0 : 1;
}

}  // namespace kzg
