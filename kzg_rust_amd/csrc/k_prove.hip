// k_prove.hip -- quotient polynomial of compute_kzg_proof_impl (reference src/kzg.rs:461-528), gfx950.
//   y   = p(z)                                         (kzg.rs:467, evaluate_polynomial_in_evaluation_form)
//   q_i = (p_i - y) / (w_i - z)            i != m      (kzg.rs:470-490)
//   q_m = sum_{i != m} (p_i - y) w_i / (z (z - w_i))   when z == w_m is inside the domain (kzg.rs:494-523)
// One 1024-thread workgroup per blob, 4 elements per thread.  Instead of the reference's three 4096-long batch
// inversions, T_i = prod_{j != i} (z - w_j) comes from the same "product of all the others" scan as k_eval, and
//     1/(z - w_i) = T_i * W,   W = 1/(z^N - 1)                       (one Fr inversion per blob)
// In-domain z = w_m: the factor (z - w_m) is replaced by 1 in the scan, so T_i = prod_{j != i,m}(w_m - w_j) and
//     1/(w_m - w_i) = T_i * w_m / N   (prod_{j != m}(w_m - w_j) = N w_m^(N-1) = N / w_m),  z^-1 = z^(N-1):  no inversion.
// y = (1/N) sum p_i w_i T_i  (or p_m), q_i = (y - p_i) / (z - w_i), q_m = -z^-1 sum_{i != m} q_i w_i.
// The 4096-point MSM over q is k_msm.hip.
#include "kernels.h"
#include "fr_block.h"

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

__global__ void __launch_bounds__(1024) k_quotient(const uint8_t *blobs, const Fr *z_in, const Fr *roots, Fr *y_out, Fr *q_out, int *err) {
    __shared__ Fr wave_tot[16], wave_ex[16], wave_sum[16], bcast;
    __shared__ uint32_t ts[4 * NFR * 1024];           // T_k per element, [k][limb][thread] (registers are capped at 128)
    __shared__ int hit;
    const int blob_i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t *blob = blobs + (size_t)BLOB_BYTES * blob_i;
    Fr *q = q_out + (size_t)N_FE * blob_i;
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
        q[e] = qe;
        if (m >= 0) { Fr t; fr_mul(t, qe, roots[e]); fr_sub(S2, S2, t); }   // + (p_i - y) w_i / (z - w_i)
    }
    if (m >= 0) {
        S2 = block_sum_1024(wave_sum, S2, lane, wid);
        if (tid == 0) {
            Fr zi = one, zz = z;                                  // z^-1 = z^(N-1) = z^4095 for z in the domain
            for (int i = 0; i < 12; i++) { fr_mul(zi, zi, zz); fr_sqr(zz, zz); }
            Fr qm; fr_mul(qm, S2, zi);
            q[m] = qm;
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

void launch_quotient(const uint8_t *d_blobs, const Fr *d_z, DeviceTables t, int n, Fr *d_y, Fr *d_q, int *d_err, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_quotient, dim3(n), dim3(1024), 0, st, d_blobs, d_z, t.roots, d_y, d_q, d_err);
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
