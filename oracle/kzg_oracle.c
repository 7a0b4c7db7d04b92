/*
 * oracle/kzg_oracle.c -- CPU restatement of the reference's KZG-4844 layer (src/kzg.rs, src/utils.rs).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg ("kind": "port").  It is never linked into, loaded by, or called from the
 * product library (libkzg355.so / kzg_rust_amd/).
 *
 * Control flow, transcript layouts and error rules follow the reference function by function
 * (file:line cited at each function, relative to /root/reference).  Parity is PINNED by the
 * reference's own 208 golden vectors (tests/golden/vectors.json <- /root/reference/tests/),
 * see tests/test_oracle_vectors.py.  The real reference cannot be built here (Rust + blst 0.3.11,
 * no cargo/rustc, blst source absent), so there is no oracle/_ref.
 *
 * Status codes mirror `enum Error` (kzg.rs:10-22): 0 OK, 1 BadArgs, 2 InternalError,
 * 3 InvalidBytesLength, 4 InvalidHexFormat, 5 InvalidTrustedSetup.
 */
#include "bls12_381.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <ctype.h>

/* FIELD_ELEMENTS_PER_BLOB is a compile-time constant in the reference (src/consts.rs:13); its README advertises a mainnet
 * (4096) and a minimal (4) preset.  The same source builds both oracles: -DN_FE=4 gives liboracle_minimal.so. */
#ifndef N_FE
#define N_FE 4096
#endif
#define BYTES_PER_BLOB (N_FE * 32)
#define N_G2 65
#define OK 0
#define BADARGS 1
#define INTERNAL 2
#define BADLEN 3
#define BADHEX 4
#define BADSETUP 5
#define EXPORT __attribute__((visibility("default")))

typedef struct okzg_settings {
    uint64_t max_width;          /* kzg.rs:30 */
    fr_t *roots_of_unity;        /* kzg.rs:34, bit-reversal order */
    g1_t *g1_values;             /* kzg.rs:37, Lagrange form, bit-reversal order */
    g1a_t *g1_affine;            /* same points in affine form (what utils.rs:381-387 recomputes per call) */
    g2_t *g2_values;             /* kzg.rs:39 */
} okzg_settings;

/* ---------------------------------------------------------------- utils.rs helpers */
/* utils.rs:94-123 fr_batch_inv */
static int fr_batch_inv(fr_t *out, const fr_t *a, size_t n) {
    if (n == 0) return BADARGS;
    fr_t acc = FR_ONE;
    for (size_t i = 0; i < n; i++) { out[i] = acc; fr_mul(&acc, &acc, &a[i]); }
    if (fr_is_zero(&acc)) return BADARGS;
    fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) { fr_mul(&out[i], &out[i], &acc); fr_mul(&acc, &acc, &a[i]); }
    return OK;
}
/* utils.rs:57-75 fr_pow */
static fr_t fr_pow(fr_t a, uint64_t n) {
    fr_t tmp = a, res = FR_ONE;
    for (;;) {
        if (n & 1) fr_mul(&res, &res, &tmp);
        n >>= 1;
        if (!n) break;
        fr_sqr(&tmp, &tmp);
    }
    return res;
}
/* utils.rs:42-50 fr_div */
static fr_t fr_div(fr_t a, fr_t b) { fr_t t; fr_inv(&t, &b); fr_mul(&t, &a, &t); return t; }
/* utils.rs:126-140 g1_mul (nbits = 256) */
static g1_t g1_mul_fr(const g1_t *a, const fr_t *b) {
    uint8_t s[32]; g1_t r; fr_to_le_scalar(s, b); g1_mul(&r, a, s, 256); return r;
}
/* utils.rs:143-157 g2_mul */
static g2_t g2_mul_fr(const g2_t *a, const fr_t *b) {
    uint8_t s[32]; g2_t r; fr_to_le_scalar(s, b); g2_mul(&r, a, s, 256); return r;
}
/* utils.rs:162-170 g1_sub ; 175-183 g2_sub */
static g1_t g1_sub(const g1_t *a, const g1_t *b) { g1_t n, r; g1_neg(&n, b); g1_add(&r, a, &n); return r; }
static g2_t g2_sub(const g2_t *a, const g2_t *b) { g2_t n, r; g2_neg(&n, b); g2_add(&r, a, &n); return r; }
/* utils.rs:189-214 pairings_verify: e(a1,a2) == e(b1,b2) */
static bool pairings_verify(const g1_t *a1, const g2_t *a2, const g1_t *b1, const g2_t *b2) {
    g1_t a1n; g1a_t aa1, bb1; g2a_t aa2, bb2; fp12_t l0, l1, gt;
    g1_neg(&a1n, a1);
    g1_to_affine(&aa1, &a1n); g1_to_affine(&bb1, b1);
    g2_to_affine(&aa2, a2); g2_to_affine(&bb2, b2);
    miller_loop(&l0, &aa2, &aa1); miller_loop(&l1, &bb2, &bb1);
    fp12_mul(&gt, &l0, &l1);
    return final_exp_is_one(&gt);
}
/* utils.rs:282-310 validate_kzg_g1 (= bytes_to_kzg_commitment :313, bytes_to_kzg_proof :318) */
static int validate_kzg_g1(g1_t *out, const uint8_t b[48]) {
    g1a_t aff;
    if (g1_uncompress(&aff, b) != 0) return BADARGS;
    g1_from_affine(out, &aff);
    if (g1_is_inf(out)) return OK;              /* the point at infinity is accepted */
    if (!g1_in_subgroup(out)) return BADARGS;
    return OK;
}
/* utils.rs:329-342 g1_lincomb_naive */
static int g1_lincomb_naive(g1_t *out, const g1_t *p, const fr_t *coeffs, size_t len) {
    g1_t res; g1_set_inf(&res);
    for (size_t i = 0; i < len; i++) { g1_t tmp = g1_mul_fr(&p[i], &coeffs[i]); g1_add(&res, &res, &tmp); }
    *out = res; return OK;
}
/* utils.rs:367-410 g1_lincomb_fast: Pippenger over affine points, 255-bit scalars (what
 * blst_p1s_mult_pippenger computes; bucket method, window 8, signed digits in [-127,128]).
 * len < 8 falls back to the naive path exactly like utils.rs:369-371. */
static int g1_lincomb_fast_affine(g1_t *out, const g1_t *pj, const g1a_t *pa, const fr_t *coeffs, size_t len) {
    if (len < 8) return g1_lincomb_naive(out, pj, coeffs, len);
    enum { C = 8, NW = 32, NB = 128 };
    int16_t *digits = malloc(len * NW * sizeof(int16_t));
    if (!digits) return INTERNAL;
    for (size_t i = 0; i < len; i++) {
        uint8_t s[32]; fr_to_le_scalar(s, &coeffs[i]);   /* utils.rs:390-392 blst_scalar_from_fr */
        int carry = 0;
        for (int w = 0; w < NW; w++) {
            int d = s[w] + carry;
            if (d > 128) { d -= 256; carry = 1; } else carry = 0;
            digits[i * NW + w] = (int16_t)d;
        }
    }
    g1_t acc; g1_set_inf(&acc);
    g1_t *buckets = malloc(NB * sizeof(g1_t));
    for (int w = NW - 1; w >= 0; w--) {
        for (int k = 0; k < C; k++) g1_dbl(&acc, &acc);
        for (int b = 0; b < NB; b++) g1_set_inf(&buckets[b]);
        for (size_t i = 0; i < len; i++) {
            int d = digits[i * NW + w];
            if (d > 0) g1_add_affine(&buckets[d - 1], &buckets[d - 1], &pa[i]);
            else if (d < 0) { g1a_t n = pa[i]; fp_neg(&n.y, &n.y); g1_add_affine(&buckets[-d - 1], &buckets[-d - 1], &n); }
        }
        g1_t run, sum; g1_set_inf(&run); g1_set_inf(&sum);
        for (int b = NB - 1; b >= 0; b--) { g1_add(&run, &run, &buckets[b]); g1_add(&sum, &sum, &run); }
        g1_add(&acc, &acc, &sum);
    }
    free(buckets); free(digits);
    *out = acc; return OK;
}
/* utils.rs:413-423 compute_powers */
static void compute_powers(fr_t *out, const fr_t *x, size_t n) {
    fr_t cur = FR_ONE;
    for (size_t i = 0; i < n; i++) { out[i] = cur; fr_mul(&cur, &cur, x); }
}
static void put_u64_be(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); }
/* utils.rs:426-474 compute_r_powers */
static int compute_r_powers(fr_t *out, fr_t *r_out, const g1_t *c, const fr_t *zs, const fr_t *ys, const g1_t *pr, size_t n) {
    size_t input_size = 16 + 8 + 8 + n * (48 + 2 * 32 + 48);
    uint8_t *bytes = malloc(input_size), *p = bytes;
    if (!bytes) return INTERNAL;
    memcpy(p, "RCKZGBATCH___V1_", 16); p += 16;           /* consts.rs:25 */
    put_u64_be(p, N_FE); p += 8;                          /* utils.rs:449 */
    put_u64_be(p, (uint64_t)n); p += 8;                   /* utils.rs:452 */
    for (size_t i = 0; i < n; i++) {
        g1_compress(p, &c[i]); p += 48;                   /* utils.rs:456 */
        fr_to_be(p, &zs[i]); p += 32;                     /* utils.rs:458 */
        fr_to_be(p, &ys[i]); p += 32;                     /* utils.rs:460 */
        g1_compress(p, &pr[i]); p += 48;                  /* utils.rs:462 */
    }
    uint8_t h[32]; sha256(h, bytes, input_size); free(bytes);
    fr_t r; fr_from_be_reduce(&r, h);                     /* utils.rs:472 hash_to_bls_field */
    if (r_out) *r_out = r;
    compute_powers(out, &r, n);
    return OK;
}

/* ---------------------------------------------------------------- kzg.rs */
/* kzg.rs:282-291 blob_to_polynomial */
static int blob_to_polynomial(fr_t *poly, const uint8_t *blob) {
    for (int i = 0; i < N_FE; i++) if (!fr_from_be_checked(&poly[i], blob + 32 * i)) return BADARGS;
    return OK;
}
/* kzg.rs:298-339 compute_challenge */
static int compute_challenge(fr_t *out, const uint8_t *blob, const uint8_t c[48]) {
    g1_t tmp;
    if (validate_kzg_g1(&tmp, c) != OK) return BADARGS;    /* kzg.rs:321-323 */
    size_t sz = 16 + 16 + BYTES_PER_BLOB + 48;             /* consts.rs:19 */
    uint8_t *bytes = malloc(sz);
    if (!bytes) return INTERNAL;
    memcpy(bytes, "FSBLOBVERIFY_V1_", 16);                 /* consts.rs:22 */
    put_u64_be(bytes + 16, 0); put_u64_be(bytes + 24, N_FE);
    memcpy(bytes + 32, blob, BYTES_PER_BLOB);
    memcpy(bytes + 32 + BYTES_PER_BLOB, c, 48);
    uint8_t h[32]; sha256(h, bytes, sz); free(bytes);
    fr_from_be_reduce(out, h);
    return OK;
}
/* kzg.rs:346-389 evaluate_polynomial_in_evaluation_form */
static int evaluate_polynomial_in_evaluation_form(fr_t *out, const fr_t *p, const fr_t *x, const okzg_settings *s) {
    fr_t *inverses_in = malloc(2 * N_FE * sizeof(fr_t)), *inverses = inverses_in + N_FE;
    if (!inverses_in) return INTERNAL;
    for (int i = 0; i < N_FE; i++) {
        if (fr_eq(x, &s->roots_of_unity[i])) { *out = p[i]; free(inverses_in); return OK; }
        fr_sub(&inverses_in[i], x, &s->roots_of_unity[i]);
    }
    int rc = fr_batch_inv(inverses, inverses_in, N_FE);
    if (rc != OK) { free(inverses_in); return rc; }
    fr_t res = FR_ZERO, tmp;
    for (int i = 0; i < N_FE; i++) {
        fr_mul(&tmp, &inverses[i], &s->roots_of_unity[i]);
        fr_mul(&tmp, &tmp, &p[i]);
        fr_add(&res, &res, &tmp);
    }
    fr_t nfe; fr_from_u64(&nfe, N_FE);
    res = fr_div(res, nfe);
    tmp = fr_pow(*x, N_FE); fr_sub(&tmp, &tmp, &FR_ONE);
    fr_mul(&res, &res, &tmp);
    *out = res; free(inverses_in);
    return OK;
}
/* kzg.rs:409-426 verify_kzg_proof_impl */
static bool verify_kzg_proof_impl(const g1_t *c, const fr_t *z, const fr_t *y, const g1_t *proof, const okzg_settings *s) {
    g2_t x_g2 = g2_mul_fr(&G2_GENERATOR_J, z);
    g2_t x_minus_z = g2_sub(&s->g2_values[1], &x_g2);
    g1_t y_g1 = g1_mul_fr(&G1_GENERATOR_J, y);
    g1_t p_minus_y = g1_sub(c, &y_g1);
    return pairings_verify(&p_minus_y, &G2_GENERATOR_J, proof, &x_minus_z);
}
/* kzg.rs:461-528 compute_kzg_proof_impl */
static int compute_kzg_proof_impl(uint8_t proof_out[48], fr_t *y_out, const fr_t *poly, const fr_t *z, const okzg_settings *s) {
    fr_t *q = calloc(3 * N_FE, sizeof(fr_t)), *inverses_in = q + N_FE, *inverses = q + 2 * N_FE;
    if (!q) return INTERNAL;
    int rc = evaluate_polynomial_in_evaluation_form(y_out, poly, z, s);
    if (rc != OK) { free(q); return rc; }
    size_t m = 0;
    for (int i = 0; i < N_FE; i++) {
        if (fr_eq(z, &s->roots_of_unity[i])) { m = i + 1; inverses_in[i] = FR_ONE; continue; }
        fr_sub(&q[i], &poly[i], y_out);
        fr_sub(&inverses_in[i], &s->roots_of_unity[i], z);
    }
    rc = fr_batch_inv(inverses, inverses_in, N_FE);
    if (rc != OK) { free(q); return rc; }
    for (int i = 0; i < N_FE; i++) fr_mul(&q[i], &q[i], &inverses[i]);
    if (m != 0) {                                           /* kzg.rs:494-523, omega_{m-1} == z */
        fr_t tmp;
        m -= 1; q[m] = FR_ZERO;
        for (size_t i = 0; i < N_FE; i++) {
            if (i == m) continue;
            fr_sub(&tmp, z, &s->roots_of_unity[i]); fr_mul(&inverses_in[i], &tmp, z);
        }
        rc = fr_batch_inv(inverses, inverses_in, N_FE);
        if (rc != OK) { free(q); return rc; }
        for (size_t i = 0; i < N_FE; i++) {
            if (i == m) continue;
            fr_sub(&tmp, &poly[i], y_out); fr_mul(&tmp, &tmp, &s->roots_of_unity[i]);
            fr_mul(&tmp, &tmp, &inverses[i]); fr_add(&q[m], &q[m], &tmp);
        }
    }
    g1_t out_g1;
    rc = g1_lincomb_fast_affine(&out_g1, s->g1_values, s->g1_affine, q, N_FE);
    free(q);
    if (rc != OK) return rc;
    g1_compress(proof_out, &out_g1);
    return OK;
}
/* kzg.rs:579-627 verify_kzg_proof_batch.  `dump` (optional, tests only): r(32) | proof_lincomb(48) | rhs(48). */
static int verify_kzg_proof_batch(bool *ok, const g1_t *c, const fr_t *zs, const fr_t *ys, const g1_t *pr, size_t n,
                                  const okzg_settings *s, uint8_t *dump) {
    if (n == 0) return BADARGS;
    fr_t *r_powers = malloc(2 * n * sizeof(fr_t)), *r_times_z = r_powers + n, r;
    g1_t *c_minus_y = malloc(n * sizeof(g1_t));
    compute_r_powers(r_powers, &r, c, zs, ys, pr, n);
    g1_t proof_lincomb, proof_z_lincomb, c_minus_y_lincomb, rhs;
    g1_lincomb_naive(&proof_lincomb, pr, r_powers, n);
    for (size_t i = 0; i < n; i++) {
        g1_t ye = g1_mul_fr(&G1_GENERATOR_J, &ys[i]);
        c_minus_y[i] = g1_sub(&c[i], &ye);
        fr_mul(&r_times_z[i], &r_powers[i], &zs[i]);
    }
    g1_lincomb_naive(&proof_z_lincomb, pr, r_times_z, n);
    g1_lincomb_naive(&c_minus_y_lincomb, c_minus_y, r_powers, n);
    g1_add(&rhs, &c_minus_y_lincomb, &proof_z_lincomb);
    if (dump) { fr_to_be(dump, &r); g1_compress(dump + 32, &proof_lincomb); g1_compress(dump + 80, &rhs); }
    *ok = pairings_verify(&proof_lincomb, &s->g2_values[1], &rhs, &G2_GENERATOR_J);
    free(r_powers); free(c_minus_y);
    return OK;
}

/* ---------------------------------------------------------------- trusted setup (kzg.rs:699-979) */
static uint32_t reverse_bits(uint32_t n, uint32_t order) { /* kzg.rs:700-710 */
    uint32_t r = 0;
    for (uint32_t o = order; o > 1; o >>= 1) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
EXPORT void okzg_free_trusted_setup(okzg_settings *s) {
    if (!s) return;
    free(s->roots_of_unity); free(s->g1_values); free(s->g1_affine); free(s->g2_values); free(s);
}
/* kzg.rs:833-899 load_trusted_setup */
EXPORT int okzg_load_trusted_setup(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, okzg_settings **out) {
    bls_init();
    if (n1 != N_FE || n2 != N_G2) return BADARGS;              /* kzg.rs:843 (and :49-62 InvalidTrustedSetup) */
    okzg_settings *s = calloc(1, sizeof *s);
    int max_scale = 0; while ((1u << max_scale) < n1) max_scale++;
    s->max_width = 1u << max_scale;
    g1_t *g1 = malloc(n1 * sizeof(g1_t));
    s->g2_values = malloc(n2 * sizeof(g2_t));
    s->g1_values = malloc(n1 * sizeof(g1_t));
    s->g1_affine = malloc(n1 * sizeof(g1a_t));
    s->roots_of_unity = malloc(n1 * sizeof(fr_t));
    int rc = OK;
    for (size_t i = 0; i < n1 && rc == OK; i++) {               /* kzg.rs:859-872: on-curve only, no subgroup check */
        g1a_t a; if (g1_uncompress(&a, g1_bytes + 48 * i) != 0) { rc = BADARGS; break; }
        g1_from_affine(&g1[i], &a);
    }
    for (size_t i = 0; i < n2 && rc == OK; i++) {               /* kzg.rs:874-887 */
        g2a_t a; if (g2_uncompress(&a, g2_bytes + 96 * i) != 0) { rc = BADARGS; break; }
        g2_from_affine(&s->g2_values[i], &a);
    }
    if (rc == OK) {                                             /* kzg.rs:802-830 is_trusted_setup_in_lagrange_form */
        if (pairings_verify(&g1[1], &s->g2_values[0], &g1[0], &s->g2_values[1])) rc = BADARGS;
    }
    if (rc == OK) {                                             /* kzg.rs:764-799: root = 7^((r-1)/2^12) */
        fr_t seven, root, cur = FR_ONE; fr_from_u64(&seven, 7);
        /* (r-1)/2^max_scale as 64-bit chunks (= SCALE2_ROOT_OF_UNITY[max_scale], consts.rs:163-168): exponentiate via square-and-multiply */
        uint64_t e[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
        e[0] -= 1;
        if (max_scale < 1 || max_scale > 32) rc = BADARGS;
        else for (int i = 0; i < 4; i++) e[i] = (e[i] >> max_scale) | (i < 3 ? e[i + 1] << (64 - max_scale) : 0);
        root = FR_ONE;
        for (int i = 255; i >= 0; i--) { fr_sqr(&root, &root); if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&root, &root, &seven); }
        fr_t *expanded = malloc(n1 * sizeof(fr_t));
        for (size_t i = 0; i < n1; i++) { expanded[i] = cur; fr_mul(&cur, &cur, &root); }
        if (!fr_eq(&cur, &FR_ONE)) rc = BADARGS;                /* kzg.rs:755-759 */
        for (size_t i = 0; i < n1; i++) s->roots_of_unity[i] = expanded[reverse_bits((uint32_t)i, (uint32_t)n1)];
        free(expanded);
        for (size_t i = 0; i < n1; i++) {                       /* kzg.rs:895-896 */
            s->g1_values[i] = g1[reverse_bits((uint32_t)i, (uint32_t)n1)];
            g1_to_affine(&s->g1_affine[i], &s->g1_values[i]);
        }
    }
    free(g1);
    if (rc != OK) { okzg_free_trusted_setup(s); return rc; }
    *out = s; return OK;
}
static int hexval(int ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; }
/* kzg.rs:906-979 load_trusted_setup_file */
EXPORT int okzg_load_trusted_setup_file(const char *path, okzg_settings **out) {
    FILE *f = fopen(path, "r");
    if (!f) return BADSETUP;
    char line[512];
    size_t n1 = 0, n2 = 0;
    if (!fgets(line, sizeof line, f)) { fclose(f); return BADSETUP; }
    { char *end; n1 = strtoul(line, &end, 10); if (end == line) { fclose(f); return BADSETUP; } }
    if (n1 != N_FE) { fclose(f); return BADSETUP; }
    if (!fgets(line, sizeof line, f)) { fclose(f); return BADSETUP; }
    { char *end; n2 = strtoul(line, &end, 10); if (end == line) { fclose(f); return BADSETUP; } }
    if (n2 != N_G2) { fclose(f); return BADSETUP; }
    uint8_t *g1 = malloc(n1 * 48), *g2 = malloc(n2 * 96);
    int rc = OK;
    for (size_t i = 0; i < n1 + n2 && rc == OK; i++) {
        size_t want = i < n1 ? 48 : 96;
        uint8_t *dst = i < n1 ? g1 + 48 * i : g2 + 96 * (i - n1);
        if (!fgets(line, sizeof line, f)) { rc = BADSETUP; break; }
        char *p = line; size_t len = strlen(p);
        while (len && isspace((unsigned char)p[len - 1])) p[--len] = 0;
        if (len >= 2 && p[0] == '0' && p[1] == 'x') { p += 2; len -= 2; }
        if (len != 2 * want) { rc = len % 2 ? BADHEX : BADARGS; break; }
        for (size_t k = 0; k < want; k++) {
            int hi = hexval(p[2 * k]), lo = hexval(p[2 * k + 1]);
            if (hi < 0 || lo < 0) { rc = BADHEX; break; }
            dst[k] = (uint8_t)(hi * 16 + lo);
        }
    }
    fclose(f);
    if (rc == OK) rc = okzg_load_trusted_setup(g1, n1, g2, n2, out);
    free(g1); free(g2);
    return rc;
}

/* ---------------------------------------------------------------- the seven public entry points (kzg.rs:983-1079) */
/* kzg.rs:401-406 blob_to_kzg_commitment */
EXPORT int okzg_blob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)); g1_t c;
    int rc = blob_to_polynomial(poly, blob);
    if (rc == OK) rc = g1_lincomb_fast_affine(&c, s->g1_values, s->g1_affine, poly, N_FE);  /* kzg.rs:396-398 */
    free(poly);
    if (rc == OK) g1_compress(out, &c);
    return rc;
}
/* kzg.rs:446-457 compute_kzg_proof */
EXPORT int okzg_compute_kzg_proof(uint8_t proof[48], uint8_t y_out[32], const uint8_t *blob, const uint8_t z_bytes[32], const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)), z, y; uint8_t pr[48];
    int rc = blob_to_polynomial(poly, blob);
    if (rc == OK && !fr_from_be_checked(&z, z_bytes)) rc = BADARGS;
    if (rc == OK) rc = compute_kzg_proof_impl(pr, &y, poly, &z, s);
    free(poly);
    if (rc == OK) { memcpy(proof, pr, 48); fr_to_be(y_out, &y); }
    return rc;
}
/* kzg.rs:533-544 compute_blob_kzg_proof */
EXPORT int okzg_compute_blob_kzg_proof(uint8_t proof[48], const uint8_t *blob, const uint8_t c[48], const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)), z, y; uint8_t pr[48];
    int rc = blob_to_polynomial(poly, blob);
    if (rc == OK) rc = compute_challenge(&z, blob, c);
    if (rc == OK) rc = compute_kzg_proof_impl(pr, &y, poly, &z, s);
    free(poly);
    if (rc == OK) memcpy(proof, pr, 48);
    return rc;
}
/* kzg.rs:429-443 verify_kzg_proof */
EXPORT int okzg_verify_kzg_proof(bool *ok, const uint8_t c[48], const uint8_t zb[32], const uint8_t yb[32], const uint8_t pb[48], const okzg_settings *s) {
    g1_t cm, pr; fr_t z, y;
    if (validate_kzg_g1(&cm, c) != OK) return BADARGS;
    if (!fr_from_be_checked(&z, zb)) return BADARGS;
    if (!fr_from_be_checked(&y, yb)) return BADARGS;
    if (validate_kzg_g1(&pr, pb) != OK) return BADARGS;
    *ok = verify_kzg_proof_impl(&cm, &z, &y, &pr, s);
    return OK;
}
/* kzg.rs:547-569 verify_blob_kzg_proof */
EXPORT int okzg_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t c[48], const uint8_t pb[48], const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)), z, y; g1_t cm, pr;
    int rc = blob_to_polynomial(poly, blob);
    if (rc == OK) rc = validate_kzg_g1(&cm, c);
    if (rc == OK) rc = validate_kzg_g1(&pr, pb);
    if (rc == OK) rc = compute_challenge(&z, blob, c);
    if (rc == OK) rc = evaluate_polynomial_in_evaluation_form(&y, poly, &z, s);
    free(poly);
    if (rc == OK) *ok = verify_kzg_proof_impl(&cm, &z, &y, &pr, s);
    return rc;
}
/* kzg.rs:637-693 verify_blob_kzg_proof_batch.  The reference's length check (644-651) is on three slices;
 * the C boundary carries the three lengths explicitly so that the check can be restated. */
static int verify_batch_impl(bool *ok, const uint8_t *blobs, size_t n_blobs, const uint8_t *cs, size_t n_c,
                             const uint8_t *ps, size_t n_p, const okzg_settings *s, uint8_t *dump, uint8_t *zy_dump) {
    if (n_blobs != n_c || n_c != n_p) return BADARGS;
    size_t n = n_blobs;
    if (n == 0) { *ok = true; return OK; }
    if (n == 1 && !dump) return okzg_verify_blob_kzg_proof(ok, blobs, cs, ps, s);
    g1_t *cg = malloc(2 * n * sizeof(g1_t)), *pg = cg + n;
    fr_t *zs = malloc(2 * n * sizeof(fr_t)), *ys = zs + n, *poly = malloc(N_FE * sizeof(fr_t));
    int rc = OK;
    for (size_t i = 0; i < n && rc == OK; i++) {               /* kzg.rs:671-683 */
        rc = validate_kzg_g1(&cg[i], cs + 48 * i);
        if (rc == OK) rc = blob_to_polynomial(poly, blobs + (size_t)BYTES_PER_BLOB * i);
        if (rc == OK) rc = compute_challenge(&zs[i], blobs + (size_t)BYTES_PER_BLOB * i, cs + 48 * i);
        if (rc == OK) rc = evaluate_polynomial_in_evaluation_form(&ys[i], poly, &zs[i], s);
        if (rc == OK) rc = validate_kzg_g1(&pg[i], ps + 48 * i);
        if (rc == OK && zy_dump) { fr_to_be(zy_dump + 64 * i, &zs[i]); fr_to_be(zy_dump + 64 * i + 32, &ys[i]); }
    }
    if (rc == OK) rc = verify_kzg_proof_batch(ok, cg, zs, ys, pg, n, s, dump);
    free(cg); free(zs); free(poly);
    return rc;
}
EXPORT int okzg_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, size_t n_blobs, const uint8_t *cs, size_t n_c,
                                            const uint8_t *ps, size_t n_p, const okzg_settings *s) {
    return verify_batch_impl(ok, blobs, n_blobs, cs, n_c, ps, n_p, s, NULL, NULL);
}

/* ---------------------------------------------------------------- intermediates, for stage-by-stage GPU diffs (tests only) */
EXPORT int okzg_compute_challenge(uint8_t z[32], const uint8_t *blob, const uint8_t c[48]) {
    bls_init(); fr_t zf; int rc = compute_challenge(&zf, blob, c); if (rc == OK) fr_to_be(z, &zf); return rc;
}
EXPORT int okzg_evaluate_polynomial(uint8_t y[32], const uint8_t *blob, const uint8_t zb[32], const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)), z, yf;
    int rc = blob_to_polynomial(poly, blob);
    if (rc == OK && !fr_from_be_checked(&z, zb)) rc = BADARGS;
    if (rc == OK) rc = evaluate_polynomial_in_evaluation_form(&yf, poly, &z, s);
    free(poly); if (rc == OK) fr_to_be(y, &yf); return rc;
}
/* dump = r(32) | proof_lincomb(48) | rhs(48); zy = n x (z(32) | y(32)) */
EXPORT int okzg_verify_batch_intermediates(bool *ok, uint8_t dump[128], uint8_t *zy, const uint8_t *blobs, const uint8_t *cs,
                                           const uint8_t *ps, size_t n, const okzg_settings *s) {
    return verify_batch_impl(ok, blobs, n, cs, n, ps, n, s, dump, zy);
}
EXPORT void okzg_get_roots_of_unity(uint8_t *out, const okzg_settings *s) { for (int i = 0; i < N_FE; i++) fr_to_be(out + 32 * i, &s->roots_of_unity[i]); }
EXPORT void okzg_get_g1_values(uint8_t *out, const okzg_settings *s) { for (int i = 0; i < N_FE; i++) g1_compress(out + 48 * i, &s->g1_values[i]); }

/* ---------------------------------------------------------------- primitive probes (canonical big-endian bytes in/out) */
EXPORT void okzg_init(void) { bls_init(); }
EXPORT int okzg_field_elements_per_blob(void) { return N_FE; }
EXPORT void okzg_sha256(uint8_t out[32], const uint8_t *msg, size_t len) { sha256(out, msg, len); }
/* The -march=native build on a CPU with BMI2 + ADX (+ SHA) carries a second form of the hot primitives: mulx / adcx / adox Montgomery products and
   SHA-256 with the SHA extensions (bls12_381.c).  have: is it in this build; set: switch it at run time (bench.py's cpu_baseline times both). */
int okzg_have_adx(void);
void okzg_set_adx(int on);
EXPORT int okzg_have_fast_primitives(void) { return okzg_have_adx(); }
EXPORT void okzg_set_fast_primitives(int on) { okzg_set_adx(on); }
EXPORT int okzg_fp_op(int op, uint8_t out[48], const uint8_t a[48], const uint8_t b[48]) {
    bls_init(); fp_t x, y, r;
    if (!fp_from_be(&x, a) || !fp_from_be(&y, b)) return BADARGS;
    switch (op) {
        case 0: fp_add(&r, &x, &y); break; case 1: fp_sub(&r, &x, &y); break; case 2: fp_mul(&r, &x, &y); break;
        case 3: fp_inv(&r, &x); break; case 4: if (!fp_sqrt(&r, &x)) return 2; break; default: return BADARGS;
    }
    fp_to_be(out, &r); return OK;
}
EXPORT int okzg_fr_op(int op, uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
    bls_init(); fr_t x, y, r;
    fr_from_be_reduce(&x, a); fr_from_be_reduce(&y, b);
    switch (op) {
        case 0: fr_add(&r, &x, &y); break; case 1: fr_sub(&r, &x, &y); break; case 2: fr_mul(&r, &x, &y); break;
        case 3: fr_inv(&r, &x); break; default: return BADARGS;
    }
    fr_to_be(out, &r); return OK;
}
EXPORT int okzg_g1_validate(const uint8_t in[48]) { bls_init(); g1_t p; return validate_kzg_g1(&p, in); }
EXPORT int okzg_g1_uncompress_only(const uint8_t in[48]) { bls_init(); g1a_t p; return g1_uncompress(&p, in); }
/* out = [k]P (+ Q if q != NULL), all compressed; points only need to be on the curve */
EXPORT int okzg_g1_mul_add(uint8_t out[48], const uint8_t p[48], const uint8_t k_be[32], const uint8_t *q) {
    bls_init(); g1a_t pa, qa; g1_t pj, qj, r; fr_t k;
    if (g1_uncompress(&pa, p) != 0) return BADARGS;
    g1_from_affine(&pj, &pa); fr_from_be_reduce(&k, k_be);
    r = g1_mul_fr(&pj, &k);
    if (q) { if (g1_uncompress(&qa, q) != 0) return BADARGS; g1_from_affine(&qj, &qa); g1_add(&r, &r, &qj); }
    g1_compress(out, &r); return OK;
}
/* generic lincomb over arbitrary compressed points: fast (Pippenger) or naive */
EXPORT int okzg_g1_lincomb(uint8_t out[48], const uint8_t *points, const uint8_t *scalars_be, size_t n, int fast) {
    bls_init();
    g1_t *pj = malloc(n * sizeof(g1_t)); g1a_t *pa = malloc(n * sizeof(g1a_t)); fr_t *k = malloc(n * sizeof(fr_t));
    int rc = OK;
    for (size_t i = 0; i < n; i++) {
        if (g1_uncompress(&pa[i], points + 48 * i) != 0) { rc = BADARGS; break; }
        g1_from_affine(&pj[i], &pa[i]); fr_from_be_reduce(&k[i], scalars_be + 32 * i);
    }
    g1_t r;
    if (rc == OK) rc = fast ? g1_lincomb_fast_affine(&r, pj, pa, k, n) : g1_lincomb_naive(&r, pj, k, n);
    if (rc == OK) g1_compress(out, &r);
    free(pj); free(pa); free(k); return rc;
}
EXPORT int okzg_g2_uncompress_check(const uint8_t in[96]) { bls_init(); g2a_t q; return g2_uncompress(&q, in); }
/* e(p1,q1) == e(p2,q2) on compressed inputs */
EXPORT int okzg_pairings_verify(bool *ok, const uint8_t p1[48], const uint8_t q1[96], const uint8_t p2[48], const uint8_t q2[96]) {
    bls_init(); g1a_t a, b; g2a_t qa, qb; g1_t aj, bj; g2_t qaj, qbj;
    if (g1_uncompress(&a, p1) || g1_uncompress(&b, p2) || g2_uncompress(&qa, q1) || g2_uncompress(&qb, q2)) return BADARGS;
    g1_from_affine(&aj, &a); g1_from_affine(&bj, &b); g2_from_affine(&qaj, &qa); g2_from_affine(&qbj, &qb);
    *ok = pairings_verify(&aj, &qaj, &bj, &qbj); return OK;
}
/* [k]G2 generator compressed-free probe: returns affine x.c0|x.c1|y.c0|y.c1 canonical BE (4*48) */
EXPORT void okzg_g2_gen_mul(uint8_t out[192], const uint8_t k_be[32]) {
    bls_init(); fr_t k; fr_from_be_reduce(&k, k_be); g2_t r = g2_mul_fr(&G2_GENERATOR_J, &k); g2a_t a; g2_to_affine(&a, &r);
    fp_to_be(out, &a.x.c0); fp_to_be(out + 48, &a.x.c1); fp_to_be(out + 96, &a.y.c0); fp_to_be(out + 144, &a.y.c1);
}

/* ---------------------------------------------------------------- record-level probes for the sharded (multi-GPU) path tests.
 * A record is C(48) | z(32) | y(32) | proof(48): the per-blob body of the r-transcript (utils.rs:454-463). */
/* Stage 1 for n blobs (kzg.rs:671-683): returns the first error or OK; records are written only for valid blobs. */
EXPORT int okzg_shard_records(uint8_t *records, const uint8_t *blobs, const uint8_t *cs, const uint8_t *ps, size_t n, const okzg_settings *s) {
    fr_t *poly = malloc(N_FE * sizeof(fr_t)); int rc = OK;
    for (size_t i = 0; i < n && rc == OK; i++) {
        g1_t tmp; fr_t z, y;
        rc = validate_kzg_g1(&tmp, cs + 48 * i);
        if (rc == OK) rc = blob_to_polynomial(poly, blobs + (size_t)BYTES_PER_BLOB * i);
        if (rc == OK) rc = compute_challenge(&z, blobs + (size_t)BYTES_PER_BLOB * i, cs + 48 * i);
        if (rc == OK) rc = evaluate_polynomial_in_evaluation_form(&y, poly, &z, s);
        if (rc == OK) rc = validate_kzg_g1(&tmp, ps + 48 * i);
        if (rc == OK) {
            uint8_t *r = records + 160 * i;
            memcpy(r, cs + 48 * i, 48); fr_to_be(r + 48, &z); fr_to_be(r + 80, &y); memcpy(r + 112, ps + 48 * i, 48);
        }
    }
    free(poly); return rc;
}
/* Stage 2 on n gathered records (kzg.rs:579-627). */
EXPORT int okzg_verify_records(bool *ok, const uint8_t *records, size_t n, const okzg_settings *s) {
    if (n == 0) return BADARGS;
    g1_t *cg = malloc(2 * n * sizeof(g1_t)), *pg = cg + n; fr_t *zs = malloc(2 * n * sizeof(fr_t)), *ys = zs + n;
    int rc = OK;
    for (size_t i = 0; i < n && rc == OK; i++) {
        const uint8_t *r = records + 160 * i;
        rc = validate_kzg_g1(&cg[i], r);
        if (rc == OK && !fr_from_be_checked(&zs[i], r + 48)) rc = BADARGS;
        if (rc == OK && !fr_from_be_checked(&ys[i], r + 80)) rc = BADARGS;
        if (rc == OK) rc = validate_kzg_g1(&pg[i], r + 112);
    }
    if (rc == OK) rc = verify_kzg_proof_batch(ok, cg, zs, ys, pg, n, s, NULL);
    free(cg); free(zs); return rc;
}
