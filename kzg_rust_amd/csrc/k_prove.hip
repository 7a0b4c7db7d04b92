// k_prove.hip -- quotient polynomial of compute_kzg_proof_impl (reference src/kzg.rs:461-528), gfx950.
//   y   = p(z)                                         (kzg.rs:467, evaluate_polynomial_in_evaluation_form)
//   q_i = (p_i - y) / (w_i - z)            i != m      (kzg.rs:470-490)
//   q_m = sum_{i != m} (p_i - y) w_i / (z (z - w_i))   when z == w_m is inside the domain (kzg.rs:494-523)
// Round 5: three kernels.  k_quotient_prep (a lane per blob: powers of z, the ONE inversion 1 / (z^N - 1), the list of blobs whose z is inside the
// domain), k_quotient_tree (quot_core.h: the inverses come down a binary tree over the domain, 4.4 field products per element, q leaves as canonical
// big-endian bytes that the fixed-base MSM reads like a blob) and, for the listed in-domain blobs only, k_quotient_scan -- the kernel of rounds 1-4:
// One 1024-thread workgroup per blob, 4 elements per thread.  Instead of the reference's three 4096-long batch
// inversions, T_i = prod_{j != i} (z - w_j) comes from the same "product of all the others" scan as k_eval, and
//     1/(z - w_i) = T_i * W,   W = 1/(z^N - 1)                       (one Fr inversion per blob)
// In-domain z = w_m: the factor (z - w_m) is replaced by 1 in the scan, so T_i = prod_{j != i,m}(w_m - w_j) and
//     1/(w_m - w_i) = T_i * w_m / N   (prod_{j != m}(w_m - w_j) = N w_m^(N-1) = N / w_m),  z^-1 = z^(N-1):  no inversion.
// y = (1/N) sum p_i w_i T_i  (or p_m), q_i = (y - p_i) / (z - w_i), q_m = -z^-1 sum_{i != m} q_i w_i.
// The 4096-point MSM over q is k_msm.hip.
#include "kernels.h"
#include "fr_block.h"
#include "quot_core.h"

namespace kzg {

__device__ __forceinline__ Fr block_sum_1024(Fr *wave_sum, Fr v, int lane, int wid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { Fr o = fr_shfl_down(v, off); fr_add(v, v, o); }
    __syncthreads();                       // wave_sum may still be read from a previous call
    if (lane == 0) wave_sum[wid] = v;
    __syncthreads();
    Fr sum = wave_sum[0];
    for (int i = 1; i < 16; i++) fr_add(sum, sum, wave_sum[i]);
    return sum;
}

__device__ __forceinline__ void ts_store(uint32_t *ts, int k, int tid, const Fr &v) {
#pragma unroll
    for (int i = 0; i < NFR; i++) ts[(k * NFR + i) * 1024 + tid] = v.l[i];
}
__device__ __forceinline__ void ts_load(Fr &v, const uint32_t *ts, int k, int tid) {
#pragma unroll
    for (int i = 0; i < NFR; i++) v.l[i] = ts[(k * NFR + i) * 1024 + tid];
}
__device__ __forceinline__ Fr load_p(const uint8_t *blob, int e) {
    uint32_t w[8]; load_blob_element_words(w, blob, e);
    Fr p; fr_from_words(p, w);
    return p;
}

// q_i as the canonical 32-byte big-endian integer (the blob format: the MSM reads the quotient like a blob)
__device__ __forceinline__ void store_q_bytes(uint8_t *q, int e, const uint32_t w[8]) {
    uint4 *dst = reinterpret_cast<uint4 *>(q + 32 * (size_t)e);
    dst[0] = make_uint4(bswap32(w[7]), bswap32(w[6]), bswap32(w[5]), bswap32(w[4]));
    dst[1] = make_uint4(bswap32(w[3]), bswap32(w[2]), bswap32(w[1]), bswap32(w[0]));
}

// The scan form (rounds 1-4), now for the blobs k_quotient_prep listed only: z inside the domain.  A fixed grid walks the list.
__global__ void __launch_bounds__(1024) k_quotient_scan(const uint8_t *blobs, const Fr *z_in, const Fr *roots, Fr *y_out, uint8_t *q_out, int *err,
                                                        const int *list, const int *count) {
    __shared__ Fr wave_tot[16], wave_ex[16], wave_sum[16], bcast;
    __shared__ uint32_t ts[4 * NFR * 1024];           // T_k per element, [k][limb][thread] (registers are capped at 128)
    __shared__ int hit;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n_listed = *count;
#pragma unroll 1
    for (int idx = blockIdx.x; idx < n_listed; idx += gridDim.x) {
    const int blob_i = list[idx];
    const uint8_t *blob = blobs + (size_t)BLOB_BYTES * blob_i;
    uint8_t *q = q_out + (size_t)BLOB_BYTES * blob_i;
    __syncthreads();                                   // (the previous blob of this workgroup is done with the shared buffers)
    if (tid == 0) hit = -1;
    __syncthreads();
    const Fr z = z_in[blob_i];
    const Fr one = fr_one();
    Fr L;
    {
        Fr d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = k * 1024 + tid;
            Fr t; fr_sub(t, z, roots[e]);
            const bool zero = fr_is_zero(t);
            if (zero) hit = e;                                    // at most one element can match
            fr_select(d[k], zero, t, one);
        }
        Fr a, b, c;
        fr_mul(a, d[0], d[1]); fr_mul(b, d[2], d[3]); fr_mul(L, a, b);
        fr_mul(c, d[1], b); ts_store(ts, 0, tid, c);
        fr_mul(c, d[0], b); ts_store(ts, 1, tid, c);
        fr_mul(c, a, d[3]); ts_store(ts, 2, tid, c);
        fr_mul(c, a, d[2]); ts_store(ts, 3, tid, c);
    }
    Fr ex;
    {
        Fr tot;
        wave_product_except_self(ex, tot, L, lane);
        if (lane == 0) wave_tot[wid] = tot;
    }
    __syncthreads();
    if (wid == 0) {
        Fr v = lane < 16 ? wave_tot[lane] : one, e2, t2;
        wave_product_except_self(e2, t2, v, lane);
        if (lane < 16) wave_ex[lane] = e2;
    }
    __syncthreads();
    const int m = hit;                                            // block-uniform from here on
    fr_mul(ex, ex, wave_ex[wid]);
    Fr S = fr_zero();
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int e = k * 1024 + tid;
        uint32_t w[8]; load_blob_element_words(w, blob, e);
        bad = bad || !fr_words_canonical(w);                      // blob_to_polynomial (kzg.rs:282-291)
        Fr p, T, t;
        fr_from_words(p, w);
        ts_load(T, ts, k, tid);
        fr_mul(T, T, ex);                                         // prod over all other (non-hit) elements of (z - w_j)
        ts_store(ts, k, tid, T);
        fr_mul(t, p, roots[e]); fr_mul(t, t, T);
        fr_add(S, S, t);
    }
    if (bad) atomicOr(&err[blob_i], ERR_NONCANONICAL_FR);
    S = block_sum_1024(wave_sum, S, lane, wid);
    const uint32_t inv4096[NFR] = FR_INV4096_INIT;
    Fr k4096; for (int i = 0; i < NFR; i++) k4096.l[i] = inv4096[i];
    Fr y;
    if (m >= 0) y = load_p(blob, m);                              // kzg.rs:360-362
    else fr_mul(y, S, k4096);
    // W with 1/(z - w_i) = T_i * W
    if (tid == 0) {
        Fr W;
        if (m >= 0) fr_mul(W, z, k4096);                          // w_m / N
        else {
            Fr zn = z;
            for (int i = 0; i < 12; i++) fr_sqr(zn, zn);
            fr_sub(zn, zn, one);                                  // z^N - 1 != 0 outside the domain
            fr_inv(W, zn);
        }
        bcast = W;
        y_out[blob_i] = y;
    }
    __syncthreads();
    const Fr W = bcast;
    Fr S2 = fr_zero();
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int e = k * 1024 + tid;
        if (e == m) continue;
        Fr inv, ymp, qe, T;
        ts_load(T, ts, k, tid);
        fr_mul(inv, T, W);                                        // 1 / (z - w_i)
        const Fr p = load_p(blob, e);
        fr_sub(ymp, y, p);
        fr_mul(qe, ymp, inv);                                     // (p_i - y) / (w_i - z)
        { uint32_t qw[8]; fr_to_words(qw, qe); store_q_bytes(q, e, qw); }
        if (m >= 0) { Fr t; fr_mul(t, qe, roots[e]); fr_sub(S2, S2, t); }   // + (p_i - y) w_i / (z - w_i)
    }
    if (m >= 0) {
        S2 = block_sum_1024(wave_sum, S2, lane, wid);
        if (tid == 0) {
            Fr zi = one, zz = z;                                  // z^-1 = z^(N-1) = z^4095 for z in the domain
            for (int i = 0; i < 12; i++) { fr_mul(zi, zi, zz); fr_sqr(zz, zz); }
            Fr qm; fr_mul(qm, S2, zi);
            uint32_t qw[8]; fr_to_words(qw, qm); store_q_bytes(q, m, qw);
        }
    }
    }
}

// a lane per blob: quot_prep; blobs whose z is inside the domain go to the list of k_quotient_scan
__global__ void __launch_bounds__(64) k_quotient_prep(const Fr *z_in, int n, QuotPrep *prep, int *list, int *count) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    QuotPrep pp;
    if (quot_prep(pp, z_in[i])) list[atomicAdd(count, 1)] = i;
    prep[i] = pp;
}

// The tree form: 4096 >> LG threads per blob, 2^LG leaves (consecutive positions of the bit-reversed domain: 32 x 2^LG bytes of the blob) per lane,
// walked in groups of four leaves = one 128-byte line.  Pass 1: inverses down the tree, u_i = p_i / (z - w_i), the inverse parked in q's slot;
// the blob's two sums; pass 2: q_i = (y - p_i) / (z - w_i).  (quot_core.h)
template <int LG> __global__ void __launch_bounds__(4096 >> LG, LG == 4 ? 2 : 1) k_quotient_tree(const uint8_t *blobs, const QuotPrep *prep, const Fr *roots,
        Fr *y_out, uint8_t *q_out, int *err) {
    constexpr int TPB = 4096 >> LG, NW = TPB / 64, L = LG - 2, D0 = 12 - LG, GROUPS = 1 << L;
    static_assert(LG >= 2 && LG <= 6 && (TPB % 64) == 0, "2^LG leaves per lane, whole waves per blob");
    __shared__ __attribute__((aligned(16))) uint32_t red[NW][2][NFR + 1];
    __shared__ uint32_t ybuf[NFR + 1];
    const int blob_i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const QuotPrep &pp = prep[blob_i];
    if (fr_is_zero(pp.W)) return;                                  // z inside the domain (block-uniform): k_quotient_scan's
    const uint4 *blob = reinterpret_cast<const uint4 *>(blobs + (size_t)BLOB_BYTES * blob_i);
    uint4 *q = reinterpret_cast<uint4 *>(q_out + (size_t)BLOB_BYTES * blob_i);
    Fr inv0;
    quot_path(inv0, pp, roots, D0, tid);
    Fr c[L > 0 ? L : 1][2];
    Fr Su = fr_zero(), Sp = fr_zero();
    bool bad = false;
#pragma unroll 1
    for (int g = 0; g < GROUPS; g++) {
#pragma unroll
        for (int l = 0; l < L; l++) {                              // level l: the children of this lane's node at depth D0 + l that g enters now
            if ((g & ((1 << (L - l)) - 1)) == 0) {
                Fr src;
                if (l == 0) src = inv0;
                else fr_select(src, ((g >> (L - l)) & 1) != 0, c[l > 0 ? l - 1 : 0][0], c[l > 0 ? l - 1 : 0][1]);
                const int a = (tid << l) + (g >> (L - l));
                quot_children(c[l][0], c[l][1], src, pp.zsq[11 - (D0 + l)], roots[2 * a]);
            }
        }
        Fr inv10;
        if (L == 0) inv10 = inv0;
        else fr_select(inv10, (g & 1) != 0, c[L > 0 ? L - 1 : 0][0], c[L > 0 ? L - 1 : 0][1]);
        const int a10 = (tid << L) + g;
        uint4 cur[8];                                             // the group's 128-byte line, requested as one burst at the top of the group's work
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = blob[8 * a10 + k];
        uint32_t pw[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 a = cur[2 * j], b = cur[2 * j + 1];
            pw[j][7] = bswap32(a.x); pw[j][6] = bswap32(a.y); pw[j][5] = bswap32(a.z); pw[j][4] = bswap32(a.w);
            pw[j][3] = bswap32(b.x); pw[j][2] = bswap32(b.y); pw[j][1] = bswap32(b.z); pw[j][0] = bswap32(b.w);
        }
        // bytes_to_bls_field (utils.rs:267-271): value < r.  The top word settles it unless it EQUALS r's top word.
        const uint32_t top = max(max(pw[0][7], pw[1][7]), max(pw[2][7], pw[3][7]));
        if (top >= FR_MOD_TOP_WORD) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) bad = bad || !fr_words_canonical(pw[j]);
        }
        Fr inv12[4];
        quot_group_pass1(inv12, Su, Sp, pw, inv10, a10, pp, roots);
#pragma unroll
        for (int j = 0; j < 4; j++) {                             // parked in q's own slot until pass 2 (this lane writes it, this lane reads it back)
            uint32_t w[8]; limbs_to_words<NFR, 8>(w, inv12[j].l);
            q[8 * a10 + 2 * j] = make_uint4(w[0], w[1], w[2], w[3]);
            q[8 * a10 + 2 * j + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    if (bad) atomicOr(&err[blob_i], ERR_NONCANONICAL_FR);
    // the blob's sums: fold, add over the wave, fold, add over the waves
    quot_fold(Su); quot_fold(Sp);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const Fr a = fr_shfl_down(Su, off), b = fr_shfl_down(Sp, off);
        fr_add_lazy(Su, Su, a); fr_add_lazy(Sp, Sp, b);
    }
    quot_fold(Su); quot_fold(Sp);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NFR; k++) { red[wid][0][k] = Su.l[k]; red[wid][1][k] = Sp.l[k]; }
    }
    __syncthreads();
    // y is the same for every lane of the blob: the compiler moves its three products and the folds to the SCALAR unit (~2700 scalar instructions).  ONE
    // wave does that and leaves y in LDS for the others -- with every wave at it the kernel spent 11 % of its instructions there (below 10 %).
    if (wid == 0) {
        Su = fr_zero(); Sp = fr_zero();
#pragma unroll 1
        for (int w = 0; w < NW; w++) {
            Fr a, b;
#pragma unroll
            for (int k = 0; k < NFR; k++) { a.l[k] = red[w][0][k]; b.l[k] = red[w][1][k]; }
            fr_add_lazy(Su, Su, a); fr_add_lazy(Sp, Sp, b);
        }
        Fr y0;
        quot_y(y0, Su, Sp, pp);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NFR; k++) ybuf[k] = y0.l[k];
            const uint32_t r2[NFR] = FR_R2_INIT;                  // Montgomery form for the callers that hand y out (compute_kzg_proof, kzg.rs:455)
            Fr R2; for (int i = 0; i < NFR; i++) R2.l[i] = r2[i];
            Fr ym; fr_mul(ym, y0, R2);
            y_out[blob_i] = ym;
        }
    }
    __syncthreads();
    Fr y;
#pragma unroll
    for (int k = 0; k < NFR; k++) y.l[k] = ybuf[k];
#pragma unroll 1
    for (int g = 0; g < GROUPS; g++) {
        const int a10 = (tid << L) + g;
        uint4 cb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cb[k] = blob[8 * a10 + k];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 a = cb[2 * j], b = cb[2 * j + 1];
            const uint4 ia = q[8 * a10 + 2 * j], ib = q[8 * a10 + 2 * j + 1];
            uint32_t pw[8], iw[8], qw[8];
            pw[7] = bswap32(a.x); pw[6] = bswap32(a.y); pw[5] = bswap32(a.z); pw[4] = bswap32(a.w);
            pw[3] = bswap32(b.x); pw[2] = bswap32(b.y); pw[1] = bswap32(b.z); pw[0] = bswap32(b.w);
            iw[0] = ia.x; iw[1] = ia.y; iw[2] = ia.z; iw[3] = ia.w; iw[4] = ib.x; iw[5] = ib.y; iw[6] = ib.z; iw[7] = ib.w;
            quot_leaf_pass2(qw, pw, iw, y);
            q[8 * a10 + 2 * j] = make_uint4(bswap32(qw[7]), bswap32(qw[6]), bswap32(qw[5]), bswap32(qw[4]));
            q[8 * a10 + 2 * j + 1] = make_uint4(bswap32(qw[3]), bswap32(qw[2]), bswap32(qw[1]), bswap32(qw[0]));
        }
    }
}

__global__ void __launch_bounds__(64) k_fr_from_bytes(const uint8_t *in32, int n, Fr *out, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[32];
    for (int k = 0; k < 32; k++) b[k] = in32[32 * (size_t)i + k];
    Fr v;
    if (!fr_from_be32_checked(v, b)) atomicOr(&err[i], ERR_NONCANONICAL_FR);
    out[i] = v;
}
__global__ void __launch_bounds__(64) k_fr_to_bytes(const Fr *in, int n, uint8_t *out32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[32]; fr_to_be32(b, in[i]);
    for (int k = 0; k < 32; k++) out32[32 * (size_t)i + k] = b[k];
}

// Per-batch result words left ON the device for the sharded path (kzg_rust_amd/sharded.py merges them with one all-reduce and reads them back
// once): ok == null: words[g] = status of batch g (KZG355_BADARGS if any error bit is set, else 0); else words[g] = 1 + ok + 256 * status.
__global__ void __launch_bounds__(256) k_status_words(const int *err, const int *ok, int32_t *words, int groups) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const int st = err[g] ? 1 /* KZG355_BADARGS */ : 0;
    words[g] = ok ? 1 + (ok[g] != 0 && st == 0 ? 1 : 0) + 256 * st : st;
}
void launch_status_words(const int *d_err, const int *d_ok, int32_t *d_words, int groups, hipStream_t st) {
    if (groups <= 0) return;
    hipLaunchKernelGGL(k_status_words, dim3((groups + 255) / 256), dim3(256), 0, st, d_err, d_ok, d_words, groups);
}

size_t quotient_scratch_bytes(int n) { return sizeof(QuotPrep) * (size_t)n + sizeof(int) * ((size_t)n + 4); }
// d_q: n x 131,072 bytes -- the quotient of every blob in the BLOB format (4096 canonical 32-byte big-endian integers), 16-byte aligned;
// d_scratch: quotient_scratch_bytes(n).  form: 0 by size, 2 / 4 / 6 = 2^form leaves per lane
int launch_quotient(const uint8_t *d_blobs, const Fr *d_z, DeviceTables t, int n, Fr *d_y, uint8_t *d_q, void *d_scratch, int *d_err, hipStream_t st,
        int form) {
    if (n <= 0) return 0;
    QuotPrep *prep = reinterpret_cast<QuotPrep *>(d_scratch);
    int *count = reinterpret_cast<int *>(prep + n), *list = count + 4;
    if (hipMemsetAsync(count, 0, sizeof(int), st) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_quotient_prep, dim3((n + 63) / 64), dim3(64), 0, st, d_z, n, prep, list, count);
    // few blobs: 1024 lanes per blob (a lone proof is a chain: 10 + 6 + 8 products deep instead of 8 + 30 + 32); many: 256 (fewer repeated path levels)
    if (form == 0) form = n < 512 ? 2 : 4;
    if (form == 2) hipLaunchKernelGGL(k_quotient_tree<2>, dim3(n), dim3(1024), 0, st, d_blobs, prep, t.roots, d_y, d_q, d_err);
    else if (form == 6) hipLaunchKernelGGL(k_quotient_tree<6>, dim3(n), dim3(64), 0, st, d_blobs, prep, t.roots, d_y, d_q, d_err);
    else hipLaunchKernelGGL(k_quotient_tree<4>, dim3(n), dim3(256), 0, st, d_blobs, prep, t.roots, d_y, d_q, d_err);
    // z inside the domain happens with probability 2^-243 per honest blob: the walker of that list is a SMALL fixed grid (one workgroup for few blobs: a
    // lone proof is latency-bound and every extra 1024-thread, 150 KB-of-LDS workgroup dispatched behind the tree kernel is on its critical path), and
    // its workgroups leave at once when the list is empty
    hipLaunchKernelGGL(k_quotient_scan, dim3(n < 512 ? 1 : n < 8192 ? 4 : 32), dim3(1024), 0, st, d_blobs, d_z, t.roots, d_y, d_q, d_err, list, count);
    return 0;
}
void launch_fr_from_bytes(const uint8_t *d_in32, int n, Fr *d_out, int *d_err, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fr_from_bytes, dim3((n + 63) / 64), dim3(64), 0, st, d_in32, n, d_out, d_err);
}
void launch_fr_to_bytes(const Fr *d_in, int n, uint8_t *d_out32, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fr_to_bytes, dim3((n + 63) / 64), dim3(64), 0, st, d_in, n, d_out32);
}

}  // namespace kzg
