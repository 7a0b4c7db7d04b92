// k_msm.hip -- fixed-base Pippenger MSM over the 4096 trusted-setup G1 points: the kernel family behind
// blob_to_kzg_commitment / compute_*_proof (reference src/utils.rs:367-410 g1_lincomb_fast ->
// blst_p1s_mult_pippenger, called from src/kzg.rs:397 and :524).
//
// Because the base points are constants of the trusted setup, load_trusted_setup stores
// table[w][i] = 2^(8w) * g1_values[i] (affine), so  sum_i s_i P_i = sum_w sum_i d_{w,i} * table[w][i]
// with signed 8-bit digits d in [-127,128]: no doublings on the hot path, every window is independent.
//   k_digits_*     scalar -> 32 biased digit bytes, stored window-major so a window's 4096 digits are contiguous
//   k_msm_bucket   one 128-thread workgroup per (blob, window): LDS counting sort of the 4096 digits by |d|,
//                  thread b accumulates bucket b+1 (mixed Jacobian+affine adds, table rows gathered from L2/HBM),
//                  weights it by (b+1) and the workgroup tree-reduces the 128 weighted buckets through LDS
//   k_msm_finalize one workgroup per blob: sum the 32 window partials, to affine, ZCash-compress (48 bytes)
// The result is the same group element blst's Pippenger returns, hence the same 48 bytes (utils.rs:221-227).
#define KZG_MID_INLINE 1
#include "kernels.h"
#include "g1_quad.h"

namespace kzg {

__device__ __forceinline__ uint32_t bswap32m(uint32_t x) { return __builtin_bswap32(x); }

// s + K, K = sum_{w<31} 127 * 256^w  (digit e_w = byte w of the sum; d_w = e_w - 127 for w < 31, d_31 = e_31)
KZG_HD void recode_words(uint32_t out[8], const uint32_t s[8]) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t k = i < 7 ? 0x7f7f7f7fu : 0x007f7f7fu;
        c += (uint64_t)s[i] + k;
        out[i] = (uint32_t)c;
        c >>= 32;
    }
}
__device__ __forceinline__ void store_digits(uint8_t *digits, int blob, int i, const uint32_t e[8]) {
#pragma unroll
    for (int w = 0; w < MSM_WINDOWS; w++)
        digits[((size_t)blob * MSM_WINDOWS + w) * N_FE + i] = (uint8_t)(e[w >> 2] >> (8 * (w & 3)));
}

// thread per field element of every blob: canonical check (blob_to_polynomial, kzg.rs:282-291) + recode.
// The scalar IS the canonical integer of the field element (utils.rs:390-392 converts Montgomery -> scalar; here it
// never left integer form).
__global__ void __launch_bounds__(256) k_digits_from_blobs(const uint8_t *blobs, int n, uint8_t *digits, int *err) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * N_FE) return;
    const int blob = (int)(gid / N_FE), i = (int)(gid % N_FE);
    const uint4 *p = reinterpret_cast<const uint4 *>(blobs + (size_t)BLOB_BYTES * blob + 32 * (size_t)i);
    uint4 a = p[0], b = p[1];
    uint32_t w[8], e[8];
    w[7] = bswap32m(a.x); w[6] = bswap32m(a.y); w[5] = bswap32m(a.z); w[4] = bswap32m(a.w);
    w[3] = bswap32m(b.x); w[2] = bswap32m(b.y); w[1] = bswap32m(b.z); w[0] = bswap32m(b.w);
    if (!fr_words_canonical(w)) {
        // the blob is rejected (its output is never returned); its digits must still stay inside the bucket range the MSM kernel
        // indexes with (a top byte >= 0x81 would give bucket 129..255), so the offending scalar is recoded as 0
        atomicOr(&err[blob], ERR_NONCANONICAL_FR);
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = 0;
    }
    recode_words(e, w);
    store_digits(digits, blob, i, e);
}

// same from Montgomery-form field elements (the quotient polynomial of the proof path, kzg.rs:524)
__global__ void __launch_bounds__(256) k_digits_from_fr(const Fr *scalars, int n, uint8_t *digits) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * N_FE) return;
    const int blob = (int)(gid / N_FE), i = (int)(gid % N_FE);
    uint32_t w[8], e[8];
    fr_to_words(w, scalars[gid]);
    recode_words(e, w);
    store_digits(digits, blob, i, e);
}

constexpr int MSM_THREADS = MSM_BUCKETS;   // 128: one lane per bucket

// One workgroup handles `wpb` consecutive windows of one blob, MSM_MERGE (4) at a time.  Because the table is pre-shifted
// per window, a point with digit d contributes d * table[w][i] whatever its window, so
//   * the 16,384 digits of 4 windows go through ONE LDS counting sort (entry = index | window << 12 | sign << 15) and
//     lane b walks one list of ~128 points: the lane-to-lane spread that a wave pays for (max over its 64 lanes) drops
//     from +40 % (32 +- 6 points) to +20 % (128 +- 11);
//   * the 128 bucket accumulators are SHARED by all wpb windows and the per-bucket weighting + tree reduction (about as
//     expensive as 27 point additions per lane) is paid once per workgroup instead of once per window.
// wpb = 4 keeps the dependent chain short (one blob alone: 8 workgroups); wpb = 8 when there are blobs enough to fill
// the chip anyway.  The reduction scratch reuses the sort buffer, so LDS stays at 33 KB (4 workgroups per CU).
// MSM_MERGE = 1 / wpb = 1 (one window per workgroup, 32 workgroups per blob) is the latency form used for small calls.
template <int MSM_MERGE>
__global__ void __launch_bounds__(MSM_THREADS) k_msm_bucket(const uint8_t *digits, const G1Affine *table, G1Jac *partials, int wpb) {
    constexpr int SORT_BYTES = MSM_MERGE * N_FE * 2, RED_BYTES = MSM_THREADS * (int)sizeof(G1Jac);
    __shared__ __attribute__((aligned(16))) uint16_t sorted[(SORT_BYTES > RED_BYTES ? SORT_BYTES : RED_BYTES) / 2];   // sort buffer, reused for reductions
    __shared__ int cnt[MSM_BUCKETS + 1], start[MSM_BUCKETS + 1], cursor[MSM_BUCKETS + 1];
    G1Jac *red = reinterpret_cast<G1Jac *>(sorted);
    constexpr int LANE_CAP = MSM_MERGE == 1 ? 64 : 192;   // entries one lane accumulates alone per pass (uniform digits: 32 +- 6 / 128 +- 11)
    const int groups_per_blob = MSM_WINDOWS / wpb;
    const int blob = blockIdx.x / groups_per_blob, wg = blockIdx.x % groups_per_blob, tid = threadIdx.x;
    G1Jac acc = g1_inf();
    for (int w0 = wg * wpb; w0 < (wg + 1) * wpb; w0 += MSM_MERGE) {
        __syncthreads();                                  // previous lists / reduction scratch no longer in use
        for (int b = tid; b <= MSM_BUCKETS; b += MSM_THREADS) cnt[b] = 0;
        __syncthreads();
        // per window: 32 consecutive digits per thread, two 16-byte loads (each window is one 4 KiB coalesced read)
        uint32_t dw[MSM_MERGE][8];
#pragma unroll
        for (int wi = 0; wi < MSM_MERGE; wi++) {
            const uint4 *dp = reinterpret_cast<const uint4 *>(digits + ((size_t)blob * MSM_WINDOWS + w0 + wi) * N_FE + 32 * tid);
            const uint4 d0 = dp[0], d1 = dp[1];
            dw[wi][0] = d0.x; dw[wi][1] = d0.y; dw[wi][2] = d0.z; dw[wi][3] = d0.w;
            dw[wi][4] = d1.x; dw[wi][5] = d1.y; dw[wi][6] = d1.z; dw[wi][7] = d1.w;
        }
#pragma unroll
        for (int wi = 0; wi < MSM_MERGE; wi++) {
            const int bias = (w0 + wi) == MSM_WINDOWS - 1 ? 0 : 127;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int d = (int)((dw[wi][k >> 2] >> (8 * (k & 3))) & 0xff) - bias;
                const int bucket = d < 0 ? -d : d;
                if (bucket) atomicAdd(&cnt[bucket], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int b = 1; b <= MSM_BUCKETS; b++) { start[b] = run; cursor[b] = run; run += cnt[b]; }
        }
        __syncthreads();
#pragma unroll
        for (int wi = 0; wi < MSM_MERGE; wi++) {
            const int bias = (w0 + wi) == MSM_WINDOWS - 1 ? 0 : 127;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int d = (int)((dw[wi][k >> 2] >> (8 * (k & 3))) & 0xff) - bias;
                const int bucket = d < 0 ? -d : d;
                if (bucket) {
                    const int pos = atomicAdd(&cursor[bucket], 1);
                    sorted[pos] = (uint16_t)((32 * tid + k) | (wi << 12) | (d < 0 ? 0x8000 : 0));
                }
            }
        }
        __syncthreads();
        // bucket accumulation: lane b sums the points whose |digit| == b + 1 -- but at most MSM_LANE_CAP of them per pass.
        // Real blobs are far from uniform (31-byte packing leaves the top byte 0, so window 31 only sees digits 0/1, and a
        // constant blob puts all 4096 points of every window in ONE bucket): entries beyond the cap are summed by the whole
        // workgroup (strided partial sums + tree) and handed back to the bucket's lane, so no lane walks a long list.
        const G1Affine *tw = table + (size_t)w0 * N_FE;
        {
            const int b = tid + 1, s0 = start[b];
            const int c = cnt[b] < LANE_CAP ? cnt[b] : LANE_CAP;
            // software-pipelined gather: the table row of entry j+1 is requested before the addition of entry j, so the
            // L2 / Infinity-Cache latency of the 112-byte row hides under ~6.6k instructions of field arithmetic
            uint32_t v = c > 0 ? sorted[s0] : 0u;
            G1Affine p = tw[v & 0x3fff];                   // window-in-group * 4096 + index
            for (int j = 0; j < c; j++) {
                const uint32_t vn = j + 1 < c ? sorted[s0 + j + 1] : v;
                const G1Affine pn = tw[vn & 0x3fff];
                if (v & 0x8000) fp_neg(p.y, p.y);
                g1_add_mixed(acc, acc, p);
                v = vn; p = pn;
            }
        }
        for (int b = 1; b <= MSM_BUCKETS; b++) {            // workgroup-uniform loop; normally no bucket qualifies
            const int extra = cnt[b] - LANE_CAP;
            if (extra <= 0) continue;
            const int base = start[b] + LANE_CAP;
            G1Jac part = g1_inf();
            for (int j = tid; j < extra; j += MSM_THREADS) {
                const uint32_t v = sorted[base + j];
                G1Affine p = tw[v & 0x3fff];
                if (v & 0x8000) fp_neg(p.y, p.y);
                g1_add_mixed(part, part, p);
            }
            // tree over the lanes by shuffles + one LDS-free cross-wave step is not worth it here (rare path): go through
            // global-memory-free registers: wave butterfly, then wave 1 hands its sum to wave 0 through `cnt`-sized scratch
            for (int off = 1; off < 64; off <<= 1) {
                G1Jac o;
#pragma unroll
                for (int q = 0; q < NFP; q++) { o.x.l[q] = __shfl_xor(part.x.l[q], off, 64); o.y.l[q] = __shfl_xor(part.y.l[q], off, 64);
                        o.z.l[q] = __shfl_xor(part.z.l[q], off, 64); }
                g1_add(part, part, o);
            }
            // every lane of a wave now holds that wave's total; combine the two waves via the partial-sum slot of lane b-1
            __shared__ G1Jac wave_part[2];
            if ((tid & 63) == 0) wave_part[tid >> 6] = part;
            __syncthreads();
            if (tid == b - 1) { G1Jac x = wave_part[0], y = wave_part[1]; g1_add(x, x, y); g1_add(acc, acc, x); }
            __syncthreads();
        }
    }
    // weight by the bucket index (b+1 <= 128: 8-bit double-and-add), then tree-reduce across the workgroup
    {
        const uint32_t m = (uint32_t)(tid + 1);
        G1Jac r = g1_inf();
        for (int bit = 7; bit >= 0; bit--) {
            g1_dbl(r, r);
            if ((m >> bit) & 1) g1_add(r, r, acc);
        }
        __syncthreads();                                  // the sort buffer becomes the reduction scratch
        red[tid] = r;
    }
    __syncthreads();
    for (int s = MSM_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) { G1Jac a = red[tid], b = red[tid + s]; g1_add(a, a, b); red[tid] = a; }
        __syncthreads();
    }
    if (tid == 0) partials[(size_t)blob * groups_per_blob + wg] = red[0];
}

__device__ __forceinline__ G1Jac g1_shfl_xor_w(const G1Jac &v, int mask) {
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_xor(v.x.l[i], mask, 64); r.y.l[i] = __shfl_xor(v.y.l[i], mask, 64); r.z.l[i] = __shfl_xor(v.z.l[i], mask,
            64); }
    return r;
}
// One wave per blob, up to 32 partial sums (a lone blob's MSM is spread over 32 workgroups): quad q adds partials 2q and 2q + 1, then a butterfly over the
// sixteen quads -- five quad additions in a row (g1_quad.h: five products deep each) where the tree of complete single-lane additions took 110 of the
// kernel's 203 us; lane 0 converts and compresses.
__global__ void __launch_bounds__(64) k_msm_finalize(const G1Jac *partials, uint8_t *out48, int ppb) {
    const int blob = blockIdx.x, tid = threadIdx.x, role = tid & 3, quad = tid >> 2;
    const G1Jac *p = partials + (size_t)blob * ppb;
    G1Jac acc = 2 * quad < ppb ? p[2 * quad] : g1_inf();
    // one loop, one inlined instance of the quad addition (k_g1.hip: several instances in a row have been miscompiled)
#pragma unroll 1
    for (int step = 0; step < 5; step++) {
        G1Jac o;
        if (step == 0) o = 2 * quad + 1 < ppb ? p[2 * quad + 1] : g1_inf();
        else o = g1_shfl_xor_w(acc, 2 << step);               // lanes 4, 8, 16, 32 away: the partner quad
        g1_add_quad(acc, acc, o, role);
    }
    if (tid == 0) {
        g1_canon_lazy(acc, acc);
        G1Affine a; g1_to_affine(a, acc);
        uint8_t b[48]; g1_compress_affine(b, a);              // bytes_from_g1 (utils.rs:221-227)
        for (int k = 0; k < 48; k++) out48[48 * (size_t)blob + k] = b[k];
    }
}
// One partial per blob (1024 blobs or more per launch): nothing to add up -- a lane per blob converts and compresses.  (A wave per blob kept 63 lanes idle
// through a ~30 k-instruction inversion: 6.7 ms per 16,384 blobs, 4 % of a commitment step.)
__global__ void __launch_bounds__(64) k_msm_finalize_lanes(const G1Jac *partials, uint8_t *out48, int n) {
    const int blob = blockIdx.x * 64 + threadIdx.x;
    if (blob >= n) return;
    G1Affine a; g1_to_affine(a, partials[blob]);
    uint8_t b[48]; g1_compress_affine(b, a);
    for (int k = 0; k < 48; k++) out48[48 * (size_t)blob + k] = b[k];
}

void launch_digits_from_blobs(const uint8_t *d_blobs, int n, uint8_t *d_digits, int *d_err, hipStream_t st) {
    if (n <= 0) return;
    const size_t total = (size_t)n * N_FE;
    hipLaunchKernelGGL(k_digits_from_blobs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_blobs, n, d_digits, d_err);
}
void launch_digits_from_fr(const Fr *d_scalars, int n, uint8_t *d_digits, hipStream_t st) {
    if (n <= 0) return;
    const size_t total = (size_t)n * N_FE;
    hipLaunchKernelGGL(k_digits_from_fr, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_scalars, n, d_digits);
}
// windows per workgroup: 1 (latency form) for small calls, 8 (two merged passes of 4) once there are blobs enough to fill the chip
int msm_windows_per_block(int n) { return n >= 128 ? 8 : 1; }
void launch_msm_bucket(const uint8_t *d_digits, DeviceTables t, int n, G1Jac *d_partials, hipStream_t st) {
    if (n <= 0) return;
    const int wpb = msm_windows_per_block(n);
    if (wpb == 1) hipLaunchKernelGGL(k_msm_bucket<1>, dim3(n * MSM_WINDOWS), dim3(MSM_THREADS), 0, st, d_digits, t.msm_table, d_partials, 1);
    else hipLaunchKernelGGL(k_msm_bucket<4>, dim3(n * (MSM_WINDOWS / wpb)), dim3(MSM_THREADS), 0, st, d_digits, t.msm_table, d_partials, wpb);
}
void launch_msm_finalize(const G1Jac *d_partials, int n, uint8_t *d_out48, hipStream_t st, int ppb) {
    if (n <= 0) return;
    const int per = ppb > 0 ? ppb : MSM_WINDOWS / msm_windows_per_block(n);
    if (per == 1) hipLaunchKernelGGL(k_msm_finalize_lanes, dim3((n + 63) / 64), dim3(64), 0, st, d_partials, d_out48, n);
    else hipLaunchKernelGGL(k_msm_finalize, dim3(n), dim3(64), 0, st, d_partials, d_out48, per);
}

}  // namespace kzg
