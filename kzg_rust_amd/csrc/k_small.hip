// k_small.hip -- the small-domain path: settings handles with 4 <= FIELD_ELEMENTS_PER_BLOB <= 64 (the `kzg_minimal` preset of
// the reference's README has 4; BASELINE.json configs[0]).  Per-blob stage of every entry point with ONE lane per blob and the
// plain formulas of the reference -- a domain of four points has nothing to tile, sort or tabulate:
//   blob -> field elements with the canonical check                                 (kzg.rs:282-291, utils.rs:262-275)
//   z = Fiat-Shamir challenge over  domain | u64be(0) | u64be(N) | blob | C          (kzg.rs:298-339)
//   y = p(z) = (1/N) sum_i p_i w_i prod_{j != i} (z - w_j)                           (kzg.rs:346-389; no special case at z = w_m)
//   q_i = (p_i - y)/(w_i - z), the in-domain entry by kzg.rs:494-523                (kzg.rs:461-528)
//   commitment / proof = sum_i [s_i] g1_values[i] by N independent scalar multiplications: g1_lincomb_fast hands fewer than 8
//   points to g1_lincomb_naive (utils.rs:369-371)
// Stage 2 (r powers, the linear combinations, the pairing) is the mainnet path's: it never depended on N beyond the u64be(N)
// field of the r-transcript (utils.rs:449).  The 4096-element kernels (k_verify.hip, k_prove.hip, k_msm*.hip) stay specialised.
#define KZG_FP_MUL_NOINLINE 1
#include "kernels.h"

namespace kzg {

constexpr int SMALL_MAX_N = 64;

__device__ __forceinline__ uint32_t brp_small(uint32_t i, int n) { return __brev(i) >> (32 - (31 - __clz(n))); }   // reverse_bits(i, n), kzg.rs:700-710

__global__ void __launch_bounds__(64) k_small_setup_g1(const uint8_t *g1_bytes, int n, G1Affine *table, G1Affine *first2, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[48];
    for (int k = 0; k < 48; k++) b[k] = g1_bytes[48 * (size_t)i + k];
    G1Affine p;
    if (g1_decompress(p, b) != 0) { atomicOr(err, ERR_SETUP_POINT); p = g1a_inf(); }      // on-curve only, no subgroup check (kzg.rs:859-872)
    table[brp_small((uint32_t)i, n)] = p;                                                   // kzg.rs:895-896
    if (i < 2) first2[i] = p;
}
// compute_roots_of_unity (kzg.rs:764-799): w_n = SCALE2_ROOT_OF_UNITY[log2 n] = w_4096^(4096 / n); thread i stores w_n^i at brp(i).
__global__ void __launch_bounds__(64) k_small_setup_roots(Fr *roots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rootc[NFR] = FR_ROOT4096_INIT;
    Fr base; for (int k = 0; k < NFR; k++) base.l[k] = rootc[k];
    for (int m = n; m < N_FE; m <<= 1) fr_sqr(base, base);
    Fr acc = fr_one();
    for (int b = 5; b >= 0; b--) { fr_sqr(acc, acc); if ((i >> b) & 1) fr_mul(acc, acc, base); }
    roots[brp_small((uint32_t)i, n)] = acc;
}

// incremental SHA-256 over bytes (sha256.h's block function)
struct ShaStream {
    Sha256 s; uint32_t w[16]; uint32_t fill; uint64_t total;
    __device__ void init() { sha256_init(s); fill = 0; total = 0; for (int i = 0; i < 16; i++) w[i] = 0; }
    __device__ void byte(uint8_t b) {
        w[fill >> 2] |= (uint32_t)b << (24 - 8 * (fill & 3));
        fill++; total++;
        if (fill == 64) { sha256_block(s, w); fill = 0; for (int i = 0; i < 16; i++) w[i] = 0; }
    }
    __device__ void bytes(const uint8_t *p, int n) { for (int i = 0; i < n; i++) byte(p[i]); }
    __device__ void u64be(uint64_t v) { for (int i = 7; i >= 0; i--) byte((uint8_t)(v >> (8 * i))); }
    __device__ void finish(uint32_t digest_words[8]) {      // little-endian words of the big-endian digest integer
        const uint64_t bits = total * 8;
        byte(0x80);
        while (fill != 56) byte(0);
        for (int i = 7; i >= 0; i--) byte((uint8_t)(bits >> (8 * i)));
        sha256_digest_to_words(digest_words, s);
    }
};

// blob -> polynomial (Montgomery residues); false if any element is >= r
__device__ bool small_load_poly(Fr *p, const uint8_t *blob, int n) {
    bool ok = true;
    for (int i = 0; i < n; i++) ok = fr_from_be32_checked(p[i], blob + 32 * i) && ok;
    return ok;
}
__device__ void small_challenge(Fr &z, const uint8_t *blob, const uint8_t *commitment, int n) {
    ShaStream h; h.init();
    const uint8_t dom[16] = {'F', 'S', 'B', 'L', 'O', 'B', 'V', 'E', 'R', 'I', 'F', 'Y', '_', 'V', '1', '_'};   // consts.rs:22
    h.bytes(dom, 16); h.u64be(0); h.u64be((uint64_t)n);
    h.bytes(blob, 32 * n); h.bytes(commitment, 48);
    uint32_t dw[8]; h.finish(dw);
    fr_from_words(z, dw);                                        // hash_to_bls_field (utils.rs:250-258)
}
__device__ void small_inv_n(Fr &r, int n) {
    uint32_t w[8] = {(uint32_t)n, 0, 0, 0, 0, 0, 0, 0};
    Fr nm; fr_from_words(nm, w);
    fr_inv(r, nm);
}
// y = (1/N) sum_i p_i w_i prod_{j != i} (z - w_j)
__device__ void small_eval(Fr &y, const Fr *p, const Fr &z, const Fr *roots, int n) {
    Fr sum = fr_zero();
    for (int i = 0; i < n; i++) {
        Fr t; fr_mul(t, p[i], roots[i]);
        for (int j = 0; j < n; j++) {
            if (j == i) continue;
            Fr d; fr_sub(d, z, roots[j]); fr_mul(t, t, d);
        }
        fr_add(sum, sum, t);
    }
    Fr ninv; small_inv_n(ninv, n);
    fr_mul(y, sum, ninv);
}
// sum_i [s_i] g1[i]  (N independent 255-bit scalar multiplications, utils.rs:329-342), compressed
__device__ void small_lincomb(uint8_t out[48], const Fr *scalars, const G1Affine *g1, int n) {
    G1Jac acc = g1_inf();
    for (int i = 0; i < n; i++) {
        uint32_t k[8]; fr_to_words(k, scalars[i]);
        G1Jac t; g1_mul_words(t, g1[i], k, 8);
        g1_add(acc, acc, t);
    }
    G1Affine a; g1_to_affine(a, acc);
    g1_compress_affine(out, a);
}

// records C | z | y | proof of stage 1 (point validation is launch_validate_points, as on the mainnet path)
__global__ void __launch_bounds__(64) k_small_records(const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, int n_total, int npg, int n,
                                                       const Fr *roots, Fr *z_out, uint8_t *records, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const uint8_t *blob = blobs + (size_t)32 * n * i, *cm = commitments + 48 * (size_t)i;
    Fr p[SMALL_MAX_N];
    if (!small_load_poly(p, blob, n)) atomicOr(&err[i / npg], ERR_NONCANONICAL_FR);
    Fr z, y;
    small_challenge(z, blob, cm, n);
    small_eval(y, p, z, roots, n);
    if (z_out) z_out[i] = z;
    if (records) {
        uint8_t *rec = records + (size_t)RECORD_BYTES * i;
        for (int k = 0; k < 48; k++) rec[k] = cm[k];
        fr_to_be32(rec + 48, z); fr_to_be32(rec + 80, y);
        if (proofs) for (int k = 0; k < 48; k++) rec[112 + k] = proofs[48 * (size_t)i + k];
    }
}
// blob_to_kzg_commitment (kzg.rs:401-406)
__global__ void __launch_bounds__(64) k_small_commit(const uint8_t *blobs, int n_blobs, int n, const G1Affine *g1, uint8_t *out48, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blobs) return;
    Fr p[SMALL_MAX_N];
    if (!small_load_poly(p, blobs + (size_t)32 * n * i, n)) atomicOr(&err[i], ERR_NONCANONICAL_FR);
    small_lincomb(out48 + 48 * (size_t)i, p, g1, n);
}
// compute_kzg_proof_impl (kzg.rs:461-528) at the points z_in (Montgomery), or at the blob's own challenge when commitments != null
__global__ void __launch_bounds__(64) k_small_proof(const uint8_t *blobs, const uint8_t *commitments, const Fr *z_in, int n_blobs, int n, const Fr *roots,
                                                     const G1Affine *g1, uint8_t *out48, uint8_t *y_out32, int *err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blobs) return;
    const uint8_t *blob = blobs + (size_t)32 * n * b;
    Fr p[SMALL_MAX_N], q[SMALL_MAX_N];
    if (!small_load_poly(p, blob, n)) atomicOr(&err[b], ERR_NONCANONICAL_FR);
    Fr z;
    if (commitments) small_challenge(z, blob, commitments + 48 * (size_t)b, n); else z = z_in[b];
    Fr y; small_eval(y, p, z, roots, n);
    int m = -1;
    for (int i = 0; i < n; i++) {                                 // kzg.rs:470-490
        Fr d; fr_sub(d, roots[i], z);
        if (fr_is_zero(d)) { m = i; q[i] = fr_zero(); continue; }
        Fr num, inv; fr_sub(num, p[i], y); fr_inv(inv, d); fr_mul(q[i], num, inv);
    }
    if (m >= 0) {                                                 // kzg.rs:494-523: q_m = sum_{i != m} (p_i - y) w_i / (z (z - w_i))
        Fr acc = fr_zero();
        for (int i = 0; i < n; i++) {
            if (i == m) continue;
            Fr num, den, inv, t;
            fr_sub(num, p[i], y); fr_mul(num, num, roots[i]);
            fr_sub(den, z, roots[i]); fr_mul(den, den, z);
            fr_inv(inv, den); fr_mul(t, num, inv);
            fr_add(acc, acc, t);
        }
        q[m] = acc;
    }
    small_lincomb(out48 + 48 * (size_t)b, q, g1, n);
    if (y_out32) fr_to_be32(y_out32 + 32 * (size_t)b, y);
}

// Size-n Lagrange setup from the first n monomial points [tau^k]G1:  L_j = (1/n) sum_k w^(-jk) [tau^k]G1, j in natural order
// (the loader bit-reverses afterwards, kzg.rs:895-896).  src/trusted_setup.rs:144-151 truncates the mainnet Lagrange points
// instead, which is not a basis of the smaller domain; this is the derivation SURVEY.md section 7 (hard part 7) asks for.
__global__ void __launch_bounds__(64) k_lagrange_from_monomial(const uint8_t *mono, int n, uint8_t *out, int *err) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t rootc[NFR] = FR_ROOT4096_INIT;
    Fr w; for (int k = 0; k < NFR; k++) w.l[k] = rootc[k];
    for (int m = n; m < N_FE; m <<= 1) fr_sqr(w, w);              // w_n
    Fr winv, ninv; fr_inv(winv, w); small_inv_n(ninv, n);
    Fr step = fr_one();                                           // w^-j
    for (int b = 5; b >= 0; b--) { fr_sqr(step, step); if ((j >> b) & 1) fr_mul(step, step, winv); }
    Fr s = ninv;                                                  // w^(-jk) / n
    G1Jac acc = g1_inf();
    for (int k = 0; k < n; k++) {
        uint8_t b[48];
        for (int q = 0; q < 48; q++) b[q] = mono[48 * (size_t)k + q];
        G1Affine p;
        if (g1_decompress(p, b) != 0) { atomicOr(err, ERR_SETUP_POINT); p = g1a_inf(); }
        uint32_t kw[8]; fr_to_words(kw, s);
        G1Jac t; g1_mul_words(t, p, kw, 8);
        g1_add(acc, acc, t);
        fr_mul(s, s, step);
    }
    G1Affine a; g1_to_affine(a, acc);
    g1_compress_affine(out + 48 * (size_t)j, a);
}
void launch_lagrange_from_monomial(const uint8_t *d_mono, int n, uint8_t *d_out, int *d_err, hipStream_t st) {
    hipLaunchKernelGGL(k_lagrange_from_monomial, dim3(1), dim3(64), 0, st, d_mono, n, d_out, d_err);
}

void launch_setup_small(const uint8_t *d_g1_bytes, int n, DeviceTables t, int *d_err, hipStream_t st) {
    hipLaunchKernelGGL(k_small_setup_g1, dim3(1), dim3(64), 0, st, d_g1_bytes, n, t.msm_table, t.g1_first2, d_err);
    hipLaunchKernelGGL(k_small_setup_roots, dim3(1), dim3(64), 0, st, t.roots, n);
}
void launch_small_records(const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int n_total, int npg, DeviceTables t, Fr *d_z, uint8_t *d_records,
        int *d_err, hipStream_t st) {
    if (n_total <= 0) return;
    hipLaunchKernelGGL(k_small_records, dim3((n_total + 63) / 64), dim3(64), 0, st, d_blobs, d_c, d_p, n_total, npg, t.n_fe, t.roots, d_z, d_records, d_err);
}
void launch_small_commit(const uint8_t *d_blobs, int n_blobs, DeviceTables t, uint8_t *d_out48, int *d_err, hipStream_t st) {
    if (n_blobs <= 0) return;
    hipLaunchKernelGGL(k_small_commit, dim3((n_blobs + 63) / 64), dim3(64), 0, st, d_blobs, n_blobs, t.n_fe, t.msm_table, d_out48, d_err);
}
void launch_small_proof(const uint8_t *d_blobs, const uint8_t *d_c, const Fr *d_z, int n_blobs, DeviceTables t, uint8_t *d_out48, uint8_t *d_y32, int *d_err,
        hipStream_t st) {
    if (n_blobs <= 0) return;
    hipLaunchKernelGGL(k_small_proof, dim3((n_blobs + 63) / 64), dim3(64), 0, st, d_blobs, d_c, d_z, n_blobs, t.n_fe, t.roots, t.msm_table, d_out48, d_y32,
            d_err);
}

}  // namespace kzg
