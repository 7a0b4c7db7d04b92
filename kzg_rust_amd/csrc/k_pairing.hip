// k_pairing.hip -- the pairing check of verify_kzg_proof_batch (reference src/utils.rs:189-214, called at kzg.rs:625)
// as a wave-cooperative kernel: one 64-lane wave per batch, Fp12 coefficients spread over the lanes
// (pairing_coop.h).  ~20x shorter dependent chain than the one-lane-per-batch kernel in k_verify.hip.
// tower routines out of line: the interpreter bodies stay small (fully inlined, the kernel was ~6x larger and 25 % slower: 6.4 vs 4.85 ms per 2048
// batches)
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"
#include "pairing_lanes.h"

namespace kzg {

// PAIRING_WAVES batches per workgroup, one wave each.  The waves of ONE workgroup are spread over the CU's four SIMDs, and
// the padded LDS request (launch_pairing) lets exactly ceil(groups / 1024) workgroups onto a CU: every wave of this
// latency-bound kernel gets a SIMD to itself as long as groups <= 1024.  (With one-wave workgroups the placement was left
// to the dispatcher: 4.1 ms on a good day, 6.1 ms when two waves shared a SIMD -- same wave-cycles, same clocks.)
constexpr int PAIRING_WAVES = 4;
// f_out (or null): stop after instruction n_insn - 1 and hand slot F over (12 coefficients per batch) instead of deciding the verdict: with many
// batches per launch set the hard part of the final exponentiation runs in k_pairing_hard12, twelve lanes per check.
// EU = waves per SIMD the register budget is cut for.  <1>: the kernel as the compiler likes it (196 VGPRs, two waves per SIMD) -- the form of the lone
// check and of every launch that fits two workgroups per CU.  <3>: 168 VGPRs (22 of them spilled) so that a THIRD workgroup fits beside the 51 KB of LDS
// each holds: with more than 2048 batches in a launch the extra wave per SIMD hides more of the dependent chains than the spills cost (round 6, one box,
// 8192 batches: pairing 6.67 -> 6.40 ms; the same cut on k_pairing_hard12: 6.67 -> 6.79, not taken).
template <int EU> __global__ void __launch_bounds__(64 * PAIRING_WAVES, EU) k_pairing_coop(const PairPt *pair_pts, int groups,
                                                                      const LineW *lines_w, const int *lines_inf, const FrobTables *frob,
                                                                      const CoopInsn *prog, int n_insn, const CoopScheds *scheds, int *ok, Fp *f_out) {
    __shared__ CoopMem mems[PAIRING_WAVES];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g_raw = blockIdx.x * PAIRING_WAVES + wid;
    if (g_raw >= groups) return;                                  // the waves of a workgroup never synchronise with each other
    const int g = g_raw;
    CoopMem &mem = mems[wid];
    const PairPt p1 = pair_pts[2 * (size_t)g], p2 = pair_pts[2 * (size_t)g + 1];
    const bool use1 = !fp_is_zero(p1.az) && !lines_inf[2], use2 = !fp_is_zero(p2.az) && !lines_inf[0];      // e(P, infinity) = e(infinity, Q) = 1
    // ML([tau]G2, -proof_lincomb) * ML(G2, rhs): lines_w[2] = setup g2[1] = [tau]G2, lines_w[0] = G2 generator
    coop_init(mem, scheds, p1, p2);
    coop_run(mem, prog, 0, n_insn, lines_w + 2 * N_LINES, lines_w, use1, use2, *frob);
    if (f_out) {
        // canonical: within the next kernel's invariant
        if (lane < 12) { Fp c; fp_norm_lz(c, mem.f.c[lane]); fp_canon64(c, c); f_out[12 * (size_t)g + lane] = c; }
        return;
    }
    const bool r = coop_is_one(mem, mem.t0);
    if (lane == 0) ok[g] = r ? 1 : 0;
}

// The hard part of the final exponentiation for many batches: twelve lanes per check, five checks per wave (pairing_lanes.h).  f_in: slot F of
// every batch as k_pairing_coop left it; prog[pc0 .. pc1): the tail of the pairing program; verdict = (slot T0 == 1).
constexpr int HARD12_WAVES = 4;
__global__ void __launch_bounds__(64 * HARD12_WAVES) k_pairing_hard12(const Fp *f_in, int groups, const FrobTables *frob, const CoopInsn *prog, int pc0,
        int pc1, int *ok) {
    __shared__ L12Mem mems[HARD12_WAVES * L12_BATCHES];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane / 12, k = lane % 12;
    const int g_raw = (blockIdx.x * HARD12_WAVES + wid) * L12_BATCHES + grp;
    const int g = g_raw < groups ? g_raw : groups - 1;               // a tail group redoes the last batch (no out-of-range loads) and reports nothing
    L12Mem &m = mems[wid * L12_BATCHES + (grp < L12_BATCHES ? grp : 0)];
    if (lane < 60) m.s[S_F].c[k] = f_in[12 * (size_t)g + k];
    L12_SYNC();
    l12_run(m, prog, pc0, pc1, *frob);
    const bool one = lane < 60 ? l12_coeff_is_one(m.s[S_T0], k) : true;
    const unsigned long long all = __ballot(one);
    if (lane < 60 && k == 0 && g_raw < groups) ok[g] = (((all >> (12 * grp)) & 0xfffull) == 0xfffull) ? 1 : 0;
}

// Few batches (latency): the two Miller loops of a check on TWO waves of one workgroup, each with its own f and only its own pair's
// line products (two operations per doubling step instead of three), then one product f = f0 f1 and the final exponentiation on
// wave 0.  68 of ~620 operations off the dependent chain (2.5 -> 2.3 ms for a lone batch) for twice the waves, which idle SIMDs
// absorb as long as there are few batches.
// f_in (or null): the Miller loops were made by k_pairing_coop_split, which left their value (12 coefficients per batch) there: wave 0 goes straight
// to the final exponentiation.
__global__ void __launch_bounds__(128) k_pairing_coop2(const PairPt *pair_pts, int groups, const LineW *lines_w, const int *lines_inf,
                                                       const FrobTables *frob, const CoopInsn *prog, int n_insn, const CoopScheds *scheds, int *ok,
                                                               const Fp *f_in) {
    __shared__ CoopMem mems[2];
    __shared__ Fp pre[2 * N_LINES * 6];                           // every line of both pairs evaluated at its point, ahead of the loops
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = blockIdx.x;
    CoopMem &m = mems[wid];
    const PairPt p1 = pair_pts[2 * (size_t)g], p2 = pair_pts[2 * (size_t)g + 1];
    const bool use1 = !fp_is_zero(p1.az) && !lines_inf[2], use2 = !fp_is_zero(p2.az) && !lines_inf[0];      // e(P, infinity) = e(infinity, Q) = 1
    const LineW *lines1 = lines_w + 2 * N_LINES, *lines2 = lines_w;      // lines_w[2] = setup g2[1] = [tau]G2 ; lines_w[0] = G2 generator
    coop_init(m, scheds, p1, p2);
    if (!f_in) {
#pragma unroll 1
        for (int item = threadIdx.x; item < 2 * N_LINES * 6; item += 128) coop_eval_lines_item(pre, item, lines1, lines2, pair_pts + 2 * (size_t)g);
        __syncthreads();
        coop_run(m, prog, 0, COOP_MILLER_INSNS, lines1, lines2, wid == 0 && use1, wid == 1 && use2, *frob, pre);
        __syncthreads();                                          // both waves reach this; wave 1 is done afterwards
        if (wid == 1) return;
        if (lane < 12) m.t0.c[lane] = mems[1].f.c[lane];
        COOP_SYNC();
        coop_product(m, m.sc.mul, m.f, m.f, m.t0, FULL_MASK);
    } else {
        if (wid == 1) return;
        if (lane < 12) m.f.c[lane] = f_in[12 * (size_t)g + lane];
        COOP_SYNC();
    }
    coop_run(m, prog, COOP_MILLER_INSNS, n_insn, lines1, lines2, false, false, *frob);
    const bool r = coop_is_one(m, m.t0);
    if (lane == 0) ok[g] = r ? 1 : 0;
}

// Few batches, round 4: K segments per Miller loop, 2 K waves per check (pairing_coop.h, miller_split): wave w owns segment w / 2 of pair w % 2.
// The dependent chain of the loops is then 63 squarings + the line products of ONE segment instead of all 68, followed by a tree of full
// products over the 2 K partial values.  Measured for a lone check (loops + hand-over + the launch of the second kernel): 0.397 ms on two waves,
// 0.330 at K = 2, 0.356 at K = 3 (the tree and the imbalance eat the shorter chain); whole check 1.276 -> 1.198 ms at K = 2.
template <int K> __global__ void __launch_bounds__(128 * K) k_pairing_coop_split(const PairPt *pair_pts, int groups, const LineW *lines_w, const int *lines_inf,
                                                                                const FrobTables *frob, const CoopInsn *prog, const CoopScheds *scheds,
                                                                                        Fp *f_out,
                                                                                MillerSplit sp) {
    __shared__ CoopMem mems[2 * K];
    __shared__ Fp pre[2 * N_LINES * 6];                           // every line of both pairs evaluated at its point, ahead of the loops
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, g = blockIdx.x;
    CoopMem &m = mems[wid];
    const PairPt p1 = pair_pts[2 * (size_t)g], p2 = pair_pts[2 * (size_t)g + 1];
    const bool use1 = !fp_is_zero(p1.az) && !lines_inf[2], use2 = !fp_is_zero(p2.az) && !lines_inf[0];      // e(P, infinity) = e(infinity, Q) = 1
    const LineW *lines1 = lines_w + 2 * N_LINES, *lines2 = lines_w;      // lines_w[2] = setup g2[1] = [tau]G2 ; lines_w[0] = G2 generator
    coop_init(m, scheds, p1, p2);
#pragma unroll 1
    for (int item = threadIdx.x; item < 2 * N_LINES * 6; item += 128 * K) coop_eval_lines_item(pre, item, lines1, lines2, pair_pts + 2 * (size_t)g);
    __syncthreads();
    const int pair = wid & 1, seg = wid >> 1;
    if (seg > 0) coop_set_one(m.f);                               // (segment 0 starts at the program's SET_ONE)
    coop_run(m, prog, sp.pc_start[seg], COOP_MILLER_INSNS, lines1, lines2, pair == 0 && use1, pair == 1 && use2, *frob, pre, sp.pc_lines_end[seg]);
    // the product of the 2 K partial values, pairwise: every wave passes every barrier
#pragma unroll 1
    for (int stride = 1; stride < 2 * K; stride <<= 1) {
        __syncthreads();
        if (wid % (2 * stride) == 0 && wid + stride < 2 * K) {
            if (lane < 12) m.t0.c[lane] = mems[wid + stride].f.c[lane];
            COOP_SYNC();
            coop_product(m, m.sc.mul, m.f, m.f, m.t0, FULL_MASK);
        }
    }
    // the final exponentiation follows in a kernel of its own (k_pairing_coop2 with f_in): built into this one, the same interpreter came out ~0.11 ms
    // slower over the final exponentiation than in the two-wave kernel (1.31 against 1.27 ms for the whole check although the loops were 0.075 ms
    // shorter) -- two launches back to back on one stream have no gap between them
    if (wid != 0) return;
    if (lane < 12) { Fp c; fp_norm_lz(c, m.f.c[lane]); fp_canon64(c, c); f_out[12 * (size_t)g + lane] = c; }
}

__global__ void __launch_bounds__(256) k_lines_to_w(const LineCoeff *lines, LineW *lines_w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LineW o; line_to_w(o, lines[i]);
    lines_w[i] = o;
}

// d_f12 (or null): groups * 12 Fp of scratch; with it and at least hard12_from batches the check is two kernels -- Miller loops and easy part wave-
// cooperatively, then the hard part twelve lanes per check (k_pairing_hard12)
void launch_pairing(const PairPt *d_pair_pts, DeviceTables t, int groups, int *d_ok, hipStream_t st, int two_wave_upto, Fp *d_f12, int hard12_from,
        int miller_segments) {
    if (groups <= 0) return;
    if (groups <= two_wave_upto) {          // several waves per batch while that still leaves the SIMDs a single wave each
        static CoopInsn host_prog[COOP_PROGRAM_MAX];
        static const int host_prog_len = build_pairing_program(host_prog);
        (void)host_prog_len;
        const int k = d_f12 ? (miller_segments > 0 ? miller_segments : 2) : 1;      // (the segment form hands f over through d_f12)
        if (k >= 2) {
            const MillerSplit sp = miller_split(host_prog, k);
            if (k == 2) hipLaunchKernelGGL(k_pairing_coop_split<2>, dim3(groups), dim3(256), 0, st, d_pair_pts, groups, t.lines_w, t.lines_inf, t.frob,
                    t.pairing_prog, t.coop_scheds, d_f12, sp);
            else if (k == 3) hipLaunchKernelGGL(k_pairing_coop_split<3>, dim3(groups), dim3(384), 0, st, d_pair_pts, groups, t.lines_w, t.lines_inf, t.frob,
                    t.pairing_prog, t.coop_scheds, d_f12, sp);
            else hipLaunchKernelGGL(k_pairing_coop_split<4>, dim3(groups), dim3(512), 0, st, d_pair_pts, groups, t.lines_w, t.lines_inf, t.frob,
                    t.pairing_prog, t.coop_scheds, d_f12, sp);
        }
        hipLaunchKernelGGL(k_pairing_coop2, dim3(groups), dim3(128), 0, st, d_pair_pts, groups, t.lines_w, t.lines_inf, t.frob, t.pairing_prog,
                           t.pairing_prog_len, t.coop_scheds, d_ok, k >= 2 ? (const Fp *)d_f12 : (const Fp *)nullptr);
        return;
    }
    const bool split = d_f12 && hard12_from > 0 && groups >= hard12_from && t.pairing_hard_start > 0;
    const int wgs = (groups + PAIRING_WAVES - 1) / PAIRING_WAVES;
    // pad the LDS request so that no more than ceil(wgs / 256) workgroups fit on a CU: an even spread by construction
    const int per_cu = (wgs + 255) / 256;
    const size_t fixed = sizeof(CoopMem) * PAIRING_WAVES + 1024;
    size_t pad = 0;
    const size_t want = ((size_t)160 * 1024 / per_cu) & ~(size_t)1023;
    if (want > fixed && want - fixed < 64 * 1024) pad = want - fixed;
    else if (want > fixed) pad = 64 * 1024 - 1024;
    if (per_cu >= 3) hipLaunchKernelGGL(k_pairing_coop<3>, dim3(wgs), dim3(64 * PAIRING_WAVES), pad, st, d_pair_pts, groups, t.lines_w, t.lines_inf,
                t.frob, t.pairing_prog, split ? t.pairing_hard_start : t.pairing_prog_len, t.coop_scheds, d_ok, split ? d_f12 : nullptr);
    else hipLaunchKernelGGL(k_pairing_coop<1>, dim3(wgs), dim3(64 * PAIRING_WAVES), pad, st, d_pair_pts, groups, t.lines_w, t.lines_inf,
                t.frob, t.pairing_prog, split ? t.pairing_hard_start : t.pairing_prog_len, t.coop_scheds, d_ok, split ? d_f12 : nullptr);
    if (split) {
        const int per_wg = HARD12_WAVES * L12_BATCHES;
        hipLaunchKernelGGL(k_pairing_hard12, dim3((groups + per_wg - 1) / per_wg), dim3(64 * HARD12_WAVES), 0, st, d_f12, groups, t.frob, t.pairing_prog,
                           t.pairing_hard_start, t.pairing_prog_len, d_ok);
    }
}
size_t pairing_f12_bytes(int groups) { return sizeof(Fp) * 12 * (size_t)(groups > 0 ? groups : 1); }
void launch_lines_to_w(DeviceTables t, hipStream_t st) {
    hipLaunchKernelGGL(k_lines_to_w, dim3(1), dim3(256), 0, st, t.lines, t.lines_w, 3 * N_LINES);
}

}  // namespace kzg
