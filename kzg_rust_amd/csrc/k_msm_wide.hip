// k_msm_wide.hip -- the fixed-base MSM of blob_to_kzg_commitment / compute_*_proof (reference src/utils.rs:367-410
// g1_lincomb_fast -> blst_p1s_mult_pippenger over the 4096 trusted-setup points; called from src/kzg.rs:397 and :524) with
// the table sized for an MI355X's 288 GB of HBM instead of a CPU cache:
//
//     wide[w][i][m-1] = m * 2^(c w) * g1_values[i]        w < ceil(256 / c),  i < 4096,  m = 1..2^(c-1)     (affine, 128-byte rows)
//
// c = 12: 22 x 4096 x 2048 x 128 B = 23.6 GB (c = 13: 42.9 GB, c = 14: 81.6 GB; KZG355_MSM_BITS), built once by
// load_trusted_setup.  With signed c-bit digits d_w in [-2^(c-1), 2^(c-1) - 1]
//     sum_i s_i P_i = sum_i sum_w sign(d_{w,i}) * wide[w][i][|d_{w,i}| - 1]
// is a plain sum of ceil(256 / c) x 4096 table rows per blob (90,112 for c = 12): no buckets, no doublings, no sorting, no digit buffer -- one
// 128-byte gather and one mixed addition per row (the 8-bit bucket form in k_msm.hip needs 131,072 additions plus 32 bucket
// reductions per blob and a 131 KB digit pass).  ~11.5 MB of random 128-byte HBM reads per blob.
//   k_wide_base / k_wide_rows   setup: row 1 from the 8-bit table (k_setup.hip), then the 2048 multiples in 8 segments of
//                               256 per (w, i): Jacobian run + in-lane batch inversion (one divstep inversion per 256 rows)
//   k_msm_wide                  one 256-thread workgroup per (blob, part): lane = scalars i = l, l+256, ...; recodes its scalar
//                               in registers (canonical check fused), walks the windows with the next row's gather in
//                               flight, then shuffle butterfly + LDS sum of the workgroup
// The result is the same group element blst's Pippenger returns, hence the same 48 bytes (utils.rs:221-227).
#define KZG_MID_INLINE 1
#define KZG_G1_ADD_MUL2 1       // the accumulation's Y3 as two products under one reduction (g1.h g1x_add_mixed_lazy)
#include "kernels.h"
#include "g1_quad.h"
#include "fr_block.h"

namespace kzg {

static_assert(sizeof(WideRow) == 128, "one table row per 128-byte line");

__device__ __forceinline__ G1Jac g1_shfl_xor_w(const G1Jac &v, int mask) {
    G1Jac r;
#pragma unroll
    for (int i = 0; i < NFP; i++) { r.x.l[i] = __shfl_xor(v.x.l[i], mask, 64); r.y.l[i] = __shfl_xor(v.y.l[i], mask, 64); r.z.l[i] = __shfl_xor(v.z.l[i], mask,
            64); }
    return r;
}

// ------------------------------------------------------------------------------------------------ setup
// row 1 of (w, i) for the points i_lo .. i_lo + i_n - 1 of ONE window: 2^(c w) P_i = 2^r * table8[(c w) >> 3][i], r = (c w) & 7.
// wbase: the window's block of the table, rows_w rows per point.
__global__ void __launch_bounds__(64) k_wide_base(const G1Affine *table8, WideRow *wbase, int bits_total, int rows_w, int i_lo, int i_n) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= i_n) return;
    const int i = i_lo + id;
    const int w8 = bits_total >> 3, r = bits_total & 7;
    G1Affine q = table8[(size_t)w8 * N_FE + i];
    if (r) {
        G1Jac j; g1_from_affine(j, q);
        for (int k = 0; k < r; k++) g1_dbl(j, j);
        g1_to_affine(q, j);
    }
    WideRow row; row.x = q.x; row.y = q.y;
    for (int k = 0; k < 4; k++) row.pad[k] = 0;
    wbase[(size_t)i * rows_w] = row;
}
// rows m = 256 seg + 1 .. 256 seg + 256 of (w, i), one lane per (i, seg):  Jacobian run acc += Q (parked in `jac`, with
// the running product of the z's in `pre`), ONE inversion, then backwards: z_k^-1 = inv * pre_{k-1}, inv *= z_k.
constexpr int WIDE_SEG = 256;
__global__ void __launch_bounds__(64) k_wide_rows(WideRow *wbase, G1Jac *jac, Fp *pre, int rows_w, int i_lo, int i_n) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int SEGS = rows_w / WIDE_SEG;
    if (id >= i_n * SEGS) return;
    const int seg = id % SEGS, i = i_lo + id / SEGS;
    WideRow *rows = wbase + (size_t)i * rows_w;
    G1Affine q; q.x = rows[0].x; q.y = rows[0].y;
    G1Jac *myj = jac + (size_t)id * WIDE_SEG;
    Fp *myp = pre + (size_t)id * WIDE_SEG;
    // acc = (256 seg) Q
    G1Jac acc = g1_inf();
    if (seg) {
        G1Jac b; g1_from_affine(b, q);
        for (int k = 0; k < 8; k++) g1_dbl(b, b);                 // 256 Q
        for (int bit = 8; bit >= 0; bit--) { g1_dbl(acc, acc); if ((seg >> bit) & 1) g1_add(acc, acc, b); }      // seg < 512
    }
    Fp run = fp_one();
    for (int k = 0; k < WIDE_SEG; k++) {
        g1_add_mixed(acc, acc, q);
        myj[k] = acc;
        if (!fp_is_zero(acc.z)) fp_mul(run, run, acc.z);          // infinity (only if P_i is) does not enter the product
        myp[k] = run;
    }
    Fp inv; fp_inv(inv, run);
    for (int k = WIDE_SEG - 1; k >= 0; k--) {
        const G1Jac p = myj[k];
        WideRow row;
        for (int t = 0; t < 4; t++) row.pad[t] = 0;
        if (fp_is_zero(p.z)) { row.x = fp_zero(); row.y = fp_zero(); }
        else {
            Fp zi, zi2, zi3;
            if (k) fp_mul(zi, inv, myp[k - 1]); else zi = inv;
            fp_mul(inv, inv, p.z);
            fp_sqr(zi2, zi); fp_mul(zi3, zi2, zi);
            fp_mul(row.x, p.x, zi2); fp_mul(row.y, p.y, zi3);
        }
        const int m = WIDE_SEG * seg + k;                         // row index m holds (m + 1) Q
        if (m) rows[m] = row;                                     // row 0 is Q itself (k_wide_base)
    }
}

// ------------------------------------------------------------------------------------------------ the MSM
// signed 12-bit digits of a 256-bit integer (8 little-endian words): d_w in [-2048, 2047], top digit small and >= 0
__device__ __forceinline__ int wide_raw_digit(const uint32_t s[8], int w, int bits) {
    const int bit = bits * w, wi = bit >> 5, sh = bit & 31;
    if (wi >= 8) return 0;
    uint32_t v = s[wi] >> sh;
    if (sh > 32 - bits && wi + 1 < 8) v |= s[wi + 1] << (32 - sh);
    return (int)(v & ((1u << bits) - 1));
}

template <bool FROM_FR>
__global__ void __launch_bounds__(256) k_msm_wide(const uint8_t *blobs, const Fr *scalars, const WideRow *wide, WideShape ws, G1Jac *partials, int *err,
                                                   int spl /* scalars per lane */, int parts /* window parts */) {
    __shared__ G1Jac red[4];
    const int chunks = N_FE / (256 * spl), wgpb = chunks * parts;
    const int blob = blockIdx.x / wgpb, wg = blockIdx.x % wgpb, chunk = wg / parts, part = wg % parts;
    const int tid = threadIdx.x;
    const int w_lo = (ws.windows * part) / parts, w_hi = (ws.windows * (part + 1)) / parts;
    const int half = ws.rows;                                    // 2^(c-1): digits run over [-half, half - 1]
    G1X accx = g1x_inf(); bool started = false;                   // lazy extended-Jacobian accumulator: 8M + 2S per row, no reductions
    WideRow cur; bool have = false; bool cur_neg = false;
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < spl; k++) {
        const int i = chunk * 256 * spl + k * 256 + tid;
        uint32_t s[8];
        if (FROM_FR) fr_to_words(s, scalars[(size_t)blob * N_FE + i]);
        else { load_blob_element_words(s, blobs + (size_t)BLOB_BYTES * blob, i); bad = bad || !fr_words_canonical(s); }
        int carry = 0;
        for (int w = 0; w < w_lo; w++) { const int raw = wide_raw_digit(s, w, ws.bits) + carry; carry = raw >= half; }
#pragma unroll 1
        for (int w = w_lo; w < w_hi; w++) {
            int d = wide_raw_digit(s, w, ws.bits) + carry;
            carry = (w < ws.windows - 1) && d >= half;
            if (carry) d -= 2 * half;
            if (d == 0) continue;
            const int m = d < 0 ? -d : d;
            const WideRow nxt = wide[((size_t)w * N_FE + i) * ws.rows + (m - 1)];      // in flight during the addition below
            if (have) {
                G1Affine p; p.x = cur.x; p.y = cur.y;
                if (cur_neg) fp_neg(p.y, p.y);
                g1x_add_mixed_lazy(accx, started, p);
            }
            cur = nxt; cur_neg = d < 0; have = true;
        }
    }
    if (have) {
        G1Affine p; p.x = cur.x; p.y = cur.y;
        if (cur_neg) fp_neg(p.y, p.y);
        g1x_add_mixed_lazy(accx, started, p);
    }
    G1Jac acc;
    { G1X cx; g1x_from_lazy(cx, accx, started); g1x_to_jac(acc, cx); }
    if (!FROM_FR && bad && part == 0) atomicOr(&err[blob], ERR_NONCANONICAL_FR);          // blob_to_polynomial (kzg.rs:282-291)
    // the workgroup's 256 sums -> one: two butterfly levels of complete additions, then quad additions (see k_msm_wide_glv)
#pragma unroll 1
    for (int off = 1; off < 4; off <<= 1) { G1Jac o = g1_shfl_xor_w(acc, off); g1_add(acc, acc, o); }
    acc.x = fp_quad_bcast<0>(acc.x); acc.y = fp_quad_bcast<0>(acc.y); acc.z = fp_quad_bcast<0>(acc.z);      // one Jacobian representative per quad
    const int role = tid & 3, quad = (tid >> 2) & 15;
#pragma unroll 1
    for (int step = 0; step < 6; step++) {
        G1Jac o;
        if (step < 4) o = g1_shfl_xor_w(acc, 4 << step);
        else if (step == 4) {
            g1_canon_lazy(acc, acc);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();                                      // (uniform: every thread reaches this step)
            acc = red[2 * (quad & 1)]; o = red[2 * (quad & 1) + 1];
        } else o = g1_shfl_xor_w(acc, 4);
        g1_add_quad(acc, acc, o, role);
    }
    if (tid == 0) { g1_canon_lazy(acc, acc); partials[(size_t)blob * wgpb + wg] = acc; }
}

// The GLV form (WideShape.glv): every scalar is split k = a + b x^2 and the table spans 128 bits.  Waves 0 and 1 of the workgroup walk the
// halves a, waves 2 and 3 the halves b, of the same 256 * spl scalars (2 * spl halves per lane: as many additions per lane as before per
// 17.45 / 16 rows), every lane into ONE accumulator; the sums of the two wave pairs meet at the end as  T_a + (-phi)(T_b),
// -phi(X, Y, Z) = (beta X, -Y, Z): the endomorphism once per workgroup instead of once per gathered row.
__device__ __forceinline__ int wide_raw_digit128(const uint32_t s[4], int w, int bits) {
    const int bit = bits * w, wi = bit >> 5, sh = bit & 31;
    if (wi >= 4) return 0;
    uint32_t v = s[wi] >> sh;
    if (sh > 32 - bits && wi + 1 < 4) v |= s[wi + 1] << (32 - sh);
    return (int)(v & ((1u << bits) - 1));
}
template <bool FROM_FR>
__global__ void __launch_bounds__(256) k_msm_wide_glv(const uint8_t *blobs, const Fr *scalars, const WideRow *wide, WideShape ws, G1Jac *partials, int *err,
                                                       int spl /* scalars per lane */, int parts /* window parts */) {
    __shared__ G1Jac red[4];
    const int chunks = N_FE / (256 * spl), wgpb = chunks * parts;
    const int blob = blockIdx.x / wgpb, wg = blockIdx.x % wgpb, chunk = wg / parts, part = wg % parts;
    const int tid = threadIdx.x, hsel = tid >> 7, j = tid & 127;
    const int W = ws.windows, w_lo = (W * part) / parts, w_hi = (W * (part + 1)) / parts;
    const int half = ws.rows;                                    // 2^(c-1): digits below the top window run over [-half, half - 1]
    const size_t top_base = (size_t)(W - 1) * N_FE * ws.rows;     // the top window's block: rows_top rows per point, unsigned digit
    G1X accx = g1x_inf(); bool started = false;
    WideRow cur; bool have = false; bool cur_neg = false;
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < 2 * spl; k++) {
        const int i = chunk * 256 * spl + k * 128 + j;
        uint32_t s[8], ha[4], hb[4], v[4];
        bool canon = true;
        if (FROM_FR) fr_to_words(s, scalars[(size_t)blob * N_FE + i]);
        else { load_blob_element_words(s, blobs + (size_t)BLOB_BYTES * blob, i); canon = fr_words_canonical(s); bad = bad || !canon; }
        glv_split_fast(ha, hb, s);
#pragma unroll
        // a non-canonical element (the blob is an Err anyway) must not index past the table
        for (int q = 0; q < 4; q++) v[q] = canon ? (hsel ? hb[q] : ha[q]) : 0u;
        int carry = 0;
        for (int w = 0; w < w_lo; w++) { const int raw = wide_raw_digit128(v, w, ws.bits) + carry; carry = raw >= half; }
#pragma unroll 1
        for (int w = w_lo; w < w_hi; w++) {
            int d = wide_raw_digit128(v, w, ws.bits) + carry;
            carry = (w < W - 1) && d >= half;
            if (carry) d -= 2 * half;
            if (d == 0) continue;
            const int m = d < 0 ? -d : d;
            const size_t off = w < W - 1 ? ((size_t)w * N_FE + i) * ws.rows : top_base + (size_t)i * ws.rows_top;
            const WideRow nxt = wide[off + (m - 1)];              // in flight during the addition below
            if (have) {
                G1Affine p; p.x = cur.x; p.y = cur.y;
                if (cur_neg) fp_neg(p.y, p.y);
                g1x_add_mixed_lazy(accx, started, p);
            }
            cur = nxt; cur_neg = d < 0; have = true;
        }
    }
    if (have) {
        G1Affine p; p.x = cur.x; p.y = cur.y;
        if (cur_neg) fp_neg(p.y, p.y);
        g1x_add_mixed_lazy(accx, started, p);
    }
    G1Jac acc;
    { G1X cx; g1x_from_lazy(cx, accx, started); g1x_to_jac(acc, cx); }
    if (!FROM_FR && bad && part == 0) atomicOr(&err[blob], ERR_NONCANONICAL_FR);          // blob_to_polynomial (kzg.rs:282-291)
    // The 256 sums of the workgroup -> one.  Two butterfly levels of complete additions leave every lane of a quad with the quad's sum -- the operand form
    // of the quad addition (g1_quad.h: five products deep instead of sixteen, no canonicalisation) -- which takes the other four levels, the two waves of
    // each half, and T_a + (-phi)(T_b).  (A lone blob's MSM is 8 additions per lane and was 9 complete additions of reduction: 345 us of which ~200.)
#pragma unroll 1
    for (int off = 1; off < 4; off <<= 1) { G1Jac o = g1_shfl_xor_w(acc, off); g1_add(acc, acc, o); }
    // (the four lanes hold the same POINT, but a + b and b + a come out as different Jacobian representatives -- (X, Y, Z) and (X, -Y, -Z) -- and the quad
    // addition needs identical coordinates in its four lanes: lane 0's go to all)
    acc.x = fp_quad_bcast<0>(acc.x); acc.y = fp_quad_bcast<0>(acc.y); acc.z = fp_quad_bcast<0>(acc.z);
    const int role = tid & 3, quad = (tid >> 2) & 15;
    // one loop, one inlined instance of the quad addition (k_g1.hip: several instances in a row have been miscompiled)
#pragma unroll 1
    for (int step = 0; step < 6; step++) {
        G1Jac o;
        if (step < 4) o = g1_shfl_xor_w(acc, 4 << step);
        else if (step == 4) {
            g1_canon_lazy(acc, acc);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();                                      // (uniform: every thread reaches this step)
            acc = red[2 * (quad & 1)]; o = red[2 * (quad & 1) + 1];      // even quads: the waves of the a halves, odd quads: of the b halves
        } else {
            g1_canon_lazy(acc, acc);
            if ((quad & 1) && !g1_is_inf(acc)) {                  // (-phi)(X, Y, Z) = (beta X, -Y, Z)
                const uint32_t bc[NFP] = FP_BETA_INIT;
                Fp beta; for (int q = 0; q < NFP; q++) beta.l[q] = bc[q];
                fp_mul(acc.x, acc.x, beta);
                fp_neg(acc.y, acc.y);
            }
            o = g1_shfl_xor_w(acc, 4);
        }
        g1_add_quad(acc, acc, o, role);
    }
    if (tid == 0) { g1_canon_lazy(acc, acc); partials[(size_t)blob * wgpb + wg] = acc; }
}

// ------------------------------------------------------------------------------------------------ launchers
size_t wide_table_bytes(WideShape ws) { return sizeof(WideRow) * (size_t)N_FE * wide_rows_per_point(ws); }
// Builds the table window by window, a slice of the points at a time (the Jacobian runs and their z products are parked in a scratch
// of <= 2^17 runs of 256 rows: 7.5 GB); the scratch is freed afterwards.
int build_wide_table(DeviceTables t, hipStream_t st) {
    const WideShape ws = t.wide;
    const int rows_max = ws.rows > ws.rows_top ? ws.rows : ws.rows_top;
    int slice = (1 << 17) / (rows_max / WIDE_SEG);                 // points per launch
    if (slice > N_FE) slice = N_FE;
    if (slice < 64) slice = 64;
    const size_t runs = (size_t)slice * (rows_max / WIDE_SEG);
    G1Jac *jac = nullptr; Fp *pre = nullptr;
    if (hipMalloc(&jac, sizeof(G1Jac) * runs * WIDE_SEG) != hipSuccess) { (void)hipGetLastError(); return 1; }
    if (hipMalloc(&pre, sizeof(Fp) * runs * WIDE_SEG) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(jac); return 1; }
    for (int w = 0; w < ws.windows; w++) {
        const int rows_w = w < ws.windows - 1 ? ws.rows : ws.rows_top;
        WideRow *wbase = t.wide_table + (size_t)w * N_FE * ws.rows;      // (the top window's block starts where a full one would)
        for (int i_lo = 0; i_lo < N_FE; i_lo += slice) {
            const int i_n = i_lo + slice <= N_FE ? slice : N_FE - i_lo;
            hipLaunchKernelGGL(k_wide_base, dim3((i_n + 63) / 64), dim3(64), 0, st, t.msm_table, wbase, ws.bits * w, rows_w, i_lo, i_n);
            hipLaunchKernelGGL(k_wide_rows, dim3((unsigned)(((size_t)i_n * (rows_w / WIDE_SEG) + 63) / 64)), dim3(64), 0, st, wbase, jac, pre, rows_w, i_lo,
                    i_n);
        }
    }
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(jac); (void)hipFree(pre);
    return e == hipSuccess && hipGetLastError() == hipSuccess ? 0 : 1;
}
// scalars per lane / window parts for n blobs: one workgroup per blob once the card is full, 32 per blob for a lone blob
void msm_wide_shape(int n, int *spl, int *parts) {
    if (n >= 1024) { *spl = 16; *parts = 1; }
    else if (n >= 128) { *spl = 4; *parts = 1; }
    else if (n >= 16) { *spl = 1; *parts = 1; }
    else { *spl = 1; *parts = 2; }
}
int msm_wide_partials_per_blob(int n) { int spl, parts; msm_wide_shape(n, &spl, &parts); return N_FE / (256 * spl) * parts; }
void launch_msm_wide(const uint8_t *d_blobs, const Fr *d_scalars, DeviceTables t, int n, G1Jac *d_partials, int *d_err, hipStream_t st) {
    if (n <= 0) return;
    int spl, parts; msm_wide_shape(n, &spl, &parts);
    const int wgpb = N_FE / (256 * spl) * parts;
    if (t.wide.glv) {
        if (d_scalars) hipLaunchKernelGGL(k_msm_wide_glv<true>, dim3(n * wgpb), dim3(256), 0, st, d_blobs, d_scalars, t.wide_table, t.wide, d_partials, d_err,
                spl, parts);
        else hipLaunchKernelGGL(k_msm_wide_glv<false>, dim3(n * wgpb), dim3(256), 0, st, d_blobs, d_scalars, t.wide_table, t.wide, d_partials, d_err, spl,
                parts);
        return;
    }
    if (d_scalars) hipLaunchKernelGGL(k_msm_wide<true>, dim3(n * wgpb), dim3(256), 0, st, d_blobs, d_scalars, t.wide_table, t.wide, d_partials, d_err, spl,
            parts);
    else hipLaunchKernelGGL(k_msm_wide<false>, dim3(n * wgpb), dim3(256), 0, st, d_blobs, d_scalars, t.wide_table, t.wide, d_partials, d_err, spl, parts);
}

}  // namespace kzg
