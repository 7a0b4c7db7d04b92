// k_prove.hip -- quotient polynomial of compute_kzg_proof_impl (reference src/kzg.rs:461-528), gfx950.
//   y   = p(z)                                         (kzg.rs:467, evaluate_polynomial_in_evaluation_form)
//   q_i = (p_i - y) / (w_i - z)            i != m      (kzg.rs:470-490)
//   q_m = sum_{i != m} (p_i - y) w_i / (z (z - w_i))   when z == w_m is inside the domain (kzg.rs:494-523)
// One 1024-thread workgroup per blob, 4 elements per thread; the per-element inverses 1/(z - w_i) computed for the
// evaluation are reused for the quotient, so the reference's second and third batch inversions disappear:
//   q_i = (y - p_i) * inv_i ,   q_m = z^-1 * sum_{i != m} (p_i - y) w_i inv_i .
// The 4096-point MSM over q is k_msm.hip.
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"

namespace kzg {

__device__ __forceinline__ void load_element_words(uint32_t w[8], const uint8_t *blob, int e) {
    const uint4 *p = reinterpret_cast<const uint4 *>(blob + 32 * (size_t)e);
    uint4 a = p[0], b = p[1];
    w[7] = __builtin_bswap32(a.x); w[6] = __builtin_bswap32(a.y); w[5] = __builtin_bswap32(a.z); w[4] = __builtin_bswap32(a.w);
    w[3] = __builtin_bswap32(b.x); w[2] = __builtin_bswap32(b.y); w[1] = __builtin_bswap32(b.z); w[0] = __builtin_bswap32(b.w);
}

__device__ __forceinline__ Fr block_sum_1024(Fr *red, const Fr &v, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) { Fr a = red[tid], b = red[tid + s]; fr_add(a, a, b); red[tid] = a; }
        __syncthreads();
    }
    Fr r = red[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(1024) k_quotient(const uint8_t *blobs, const Fr *z_in, const Fr *roots, Fr *y_out, Fr *q_out, int *err) {
    __shared__ Fr red[1024];
    __shared__ int hit;
    const int blob_i = blockIdx.x, tid = threadIdx.x;
    const uint8_t *blob = blobs + (size_t)BLOB_BYTES * blob_i;
    Fr *q = q_out + (size_t)N_FE * blob_i;
    if (tid == 0) hit = -1;
    __syncthreads();
    const Fr z = z_in[blob_i];
    const Fr one = fr_one();
    Fr inv[4];
    {
        Fr dd[4], pp[3];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = k * 1024 + tid;
            uint32_t w[8]; load_element_words(w, blob, e);
            bad = bad || !fr_words_canonical(w);
            Fr d; fr_sub(d, z, roots[e]);
            const bool zero = fr_is_zero(d);
            if (zero) hit = e;
            fr_select(dd[k], zero, d, one);
        }
        if (bad) atomicOr(&err[blob_i], ERR_NONCANONICAL_FR);
        fr_mul(pp[0], dd[0], dd[1]);
        fr_mul(pp[1], pp[0], dd[2]);
        fr_mul(pp[2], pp[1], dd[3]);
        Fr t; fr_inv(t, pp[2]);
        fr_mul(inv[3], t, pp[1]); fr_mul(t, t, dd[3]);
        fr_mul(inv[2], t, pp[0]); fr_mul(t, t, dd[2]);
        fr_mul(inv[1], t, dd[0]); fr_mul(t, t, dd[1]);
        inv[0] = t;
    }
    Fr sum = fr_zero();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = k * 1024 + tid;
        uint32_t w[8]; load_element_words(w, blob, e);
        Fr p; fr_from_words(p, w);
        Fr t; fr_mul(t, inv[k], roots[e]); fr_mul(t, t, p);
        fr_add(sum, sum, t);
    }
    sum = block_sum_1024(red, sum, tid);
    const int m = hit;                                   // uniform after the barriers above
    Fr y;
    if (m >= 0) {
        uint32_t w[8]; load_element_words(w, blob, m);
        fr_from_words(y, w);                             // kzg.rs:360-362
    } else {
        const uint32_t inv4096[NFR] = FR_INV4096_INIT;
        Fr k4096; for (int i = 0; i < NFR; i++) k4096.l[i] = inv4096[i];
        Fr zn = z;
        for (int i = 0; i < 12; i++) fr_sqr(zn, zn);
        fr_sub(zn, zn, one);
        fr_mul(y, sum, k4096);
        fr_mul(y, y, zn);
    }
    if (tid == 0) y_out[blob_i] = y;
    Fr sum2 = fr_zero();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = k * 1024 + tid;
        uint32_t w[8]; load_element_words(w, blob, e);
        Fr p; fr_from_words(p, w);
        Fr ymp; fr_sub(ymp, y, p);                       // y - p_i
        if (e != m) {
            Fr qe; fr_mul(qe, ymp, inv[k]);              // (p_i - y)/(w_i - z)
            q[e] = qe;
            if (m >= 0) {                                // (p_i - y) w_i / (z - w_i), accumulated for q_m
                Fr t; fr_mul(t, qe, roots[e]);           // = (y - p_i) w_i inv_i
                fr_sub(sum2, sum2, t);
            }
        }
    }
    if (m >= 0) {
        sum2 = block_sum_1024(red, sum2, tid);
        if (tid == 0) {
            Fr zi; fr_inv(zi, z);
            Fr qm; fr_mul(qm, sum2, zi);
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
