// k_pairing.hip -- the pairing check of verify_kzg_proof_batch (reference src/utils.rs:189-214, called at kzg.rs:625)
// as a wave-cooperative kernel: one 64-lane workgroup per batch, Fp12 coefficients spread over the lanes
// (pairing_coop.h).  ~20x shorter dependent chain than the one-lane-per-batch kernel in k_verify.hip.
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"

namespace kzg {

__global__ void __launch_bounds__(64) k_pairing_coop(const G1Affine *pair_pts, const LineW *lines_w, const int *lines_inf, const FrobTables *frob,
                                                      const CoopInsn *prog, int n_insn, const CoopSched *scheds, int *ok) {
    __shared__ CoopMem mem;
    const int g = blockIdx.x;
    G1Affine p1 = pair_pts[2 * (size_t)g], p2 = pair_pts[2 * (size_t)g + 1];
    if (lines_inf[2]) p1 = g1a_inf();          // e(P, infinity) = 1
    if (lines_inf[0]) p2 = g1a_inf();
    // ML([tau]G2, -proof_lincomb) * ML(G2, rhs): lines_w[2] = setup g2[1] = [tau]G2, lines_w[0] = G2 generator
    const bool r = coop_pairing_check(mem, prog, n_insn, scheds, lines_w + 2 * N_LINES, p1, lines_w, p2, *frob);
    if (threadIdx.x == 0) ok[g] = r ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_lines_to_w(const LineCoeff *lines, LineW *lines_w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LineW o; line_to_w(o, lines[i]);
    lines_w[i] = o;
}

void launch_pairing(const G1Affine *d_pair_pts, DeviceTables t, int groups, int *d_ok, hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_pairing_coop, dim3(groups), dim3(64), 0, st, d_pair_pts, t.lines_w, t.lines_inf, t.frob, t.pairing_prog, t.pairing_prog_len, t.coop_scheds, d_ok);
}
void launch_lines_to_w(DeviceTables t, hipStream_t st) {
    hipLaunchKernelGGL(k_lines_to_w, dim3(1), dim3(256), 0, st, t.lines, t.lines_w, 3 * N_LINES);
}

}  // namespace kzg
