/*
 * oracle/bls12_381.c -- see bls12_381.h.  TEST INFRASTRUCTURE ONLY (CPU oracle / cpu_baseline).
 * Restates the arithmetic behind the blst symbols listed in SURVEY.md section 2.2; each block names
 * the blst entry point(s) it stands in for and the reference call sites that use them.
 */
#include "bls12_381.h"
#include <string.h>
#include <pthread.h>

typedef unsigned __int128 u128;
#define AINLINE static inline __attribute__((always_inline))

/* ------------------------------------------------------------------ moduli */
static const uint64_t P_MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                  0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t R_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                  0x73eda753299d7d48ULL};
#define X_ABS 0xd201000000010000ULL

static uint64_t P_INV, R_INV;            /* -m^-1 mod 2^64 */
fp_t FP_ONE, FP_ZERO;
fr_t FR_ONE, FR_ZERO;
static fp_t FP_R2;
static fr_t FR_R2;
static uint64_t EXP_P_MINUS_2[6], EXP_P_PLUS_1_DIV_4[6], EXP_P_MINUS_3_DIV_4[6], EXP_P_MINUS_1_DIV_2[6];
static uint64_t EXP_R_MINUS_2[4];
static uint64_t P_MINUS_1_DIV_2[6];
static fp2_t FROB_V1, FROB_V2, FROB_W;
g1_t G1_GENERATOR_J;
g2_t G2_GENERATOR_J;

/* ------------------------------------------------------------------ generic limb helpers */
AINLINE uint64_t limbs_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
AINLINE uint64_t limbs_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
AINLINE int limbs_cmp(const uint64_t *a, const uint64_t *b, const int n) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
    return 0;
}
AINLINE bool limbs_is_zero(const uint64_t *a, const int n) {
    uint64_t t = 0; for (int i = 0; i < n; i++) t |= a[i]; return t == 0;
}
AINLINE void mod_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, const int n) {
    uint64_t t[6], c = limbs_add(r, a, b, n);
    uint64_t br = limbs_sub(t, r, m, n);
    if (c || !br) memcpy(r, t, 8 * n);
}
AINLINE void mod_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, const int n) {
    uint64_t br = limbs_sub(r, a, b, n);
    if (br) limbs_add(r, r, m, n);
}
/* The same CIOS product with the multiplier kept in rdx (mulx) and two carry chains (adcx / adox): what hand-written field arithmetic of the blst class
   does on CPUs with BMI2 + ADX.  Compiled in when the build machine has both (-march=native builds of liboracle_native.so: bench.py's cpu_baseline
   leg) unless OKZG_NO_ADX is defined; bench.py reports both figures, so that "a port, not blst" becomes a measured bracket.
   a: any n-limb value, b < m, as mont_mul. */
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__) && !defined(OKZG_NO_ADX)
#define OKZG_HAVE_ADX 1
#define OKZG_MAC_ROW4(ptr_, mult_, t0, t1, t2, t3, t4, t5) do { uint64_t lo_, hi_, z_; \
    __asm__("xorq %[z], %[z]\n\t" \
            "mulxq 0(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a0]\n\t adoxq %[hi], %[a1]\n\t" \
            "mulxq 8(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a1]\n\t adoxq %[hi], %[a2]\n\t" \
            "mulxq 16(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a2]\n\t adoxq %[hi], %[a3]\n\t" \
            "mulxq 24(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a3]\n\t adoxq %[hi], %[a4]\n\t" \
            "adcxq %[z], %[a4]\n\t adoxq %[z], %[a5]\n\t adcxq %[z], %[a5]" \
            : [a0] "+r"(t0), [a1] "+r"(t1), [a2] "+r"(t2), [a3] "+r"(t3), [a4] "+r"(t4), [a5] "+r"(t5), [lo] "=&r"(lo_), [hi] "=&r"(hi_), [z] "=&r"(z_) \
            : [p] "r"(ptr_), "d"(mult_) : "cc", "memory"); } while (0)
static void mont_mul_adx4(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, const uint64_t inv) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, q;
    OKZG_MAC_ROW4(a, b[0], t0, t1, t2, t3, t4, t5);
    q = t0 * inv;
    OKZG_MAC_ROW4(m, q, t0, t1, t2, t3, t4, t5);
    OKZG_MAC_ROW4(a, b[1], t1, t2, t3, t4, t5, t0);
    q = t1 * inv;
    OKZG_MAC_ROW4(m, q, t1, t2, t3, t4, t5, t0);
    OKZG_MAC_ROW4(a, b[2], t2, t3, t4, t5, t0, t1);
    q = t2 * inv;
    OKZG_MAC_ROW4(m, q, t2, t3, t4, t5, t0, t1);
    OKZG_MAC_ROW4(a, b[3], t3, t4, t5, t0, t1, t2);
    q = t3 * inv;
    OKZG_MAC_ROW4(m, q, t3, t4, t5, t0, t1, t2);
    uint64_t t[5] = {t4, t5, t0, t1, t2}, s[4];
    uint64_t br = limbs_sub(s, t, m, 4);
    if (t[4] || !br) memcpy(r, s, 8 * 4); else memcpy(r, t, 8 * 4);
}
#define OKZG_MAC_ROW6(ptr_, mult_, t0, t1, t2, t3, t4, t5, t6, t7) do { uint64_t lo_, hi_, z_; \
    __asm__("xorq %[z], %[z]\n\t" \
            "mulxq 0(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a0]\n\t adoxq %[hi], %[a1]\n\t" \
            "mulxq 8(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a1]\n\t adoxq %[hi], %[a2]\n\t" \
            "mulxq 16(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a2]\n\t adoxq %[hi], %[a3]\n\t" \
            "mulxq 24(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a3]\n\t adoxq %[hi], %[a4]\n\t" \
            "mulxq 32(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a4]\n\t adoxq %[hi], %[a5]\n\t" \
            "mulxq 40(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[a5]\n\t adoxq %[hi], %[a6]\n\t" \
            "adcxq %[z], %[a6]\n\t adoxq %[z], %[a7]\n\t adcxq %[z], %[a7]" \
            : [a0] "+r"(t0), [a1] "+r"(t1), [a2] "+r"(t2), [a3] "+r"(t3), [a4] "+r"(t4), [a5] "+r"(t5), [a6] "+r"(t6), [a7] "+r"(t7), [lo] "=&r"(lo_), [hi] "=&r"(hi_), [z] "=&r"(z_) \
            : [p] "r"(ptr_), "d"(mult_) : "cc", "memory"); } while (0)
static void mont_mul_adx6(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, const uint64_t inv) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, q;
    OKZG_MAC_ROW6(a, b[0], t0, t1, t2, t3, t4, t5, t6, t7);
    q = t0 * inv;
    OKZG_MAC_ROW6(m, q, t0, t1, t2, t3, t4, t5, t6, t7);
    OKZG_MAC_ROW6(a, b[1], t1, t2, t3, t4, t5, t6, t7, t0);
    q = t1 * inv;
    OKZG_MAC_ROW6(m, q, t1, t2, t3, t4, t5, t6, t7, t0);
    OKZG_MAC_ROW6(a, b[2], t2, t3, t4, t5, t6, t7, t0, t1);
    q = t2 * inv;
    OKZG_MAC_ROW6(m, q, t2, t3, t4, t5, t6, t7, t0, t1);
    OKZG_MAC_ROW6(a, b[3], t3, t4, t5, t6, t7, t0, t1, t2);
    q = t3 * inv;
    OKZG_MAC_ROW6(m, q, t3, t4, t5, t6, t7, t0, t1, t2);
    OKZG_MAC_ROW6(a, b[4], t4, t5, t6, t7, t0, t1, t2, t3);
    q = t4 * inv;
    OKZG_MAC_ROW6(m, q, t4, t5, t6, t7, t0, t1, t2, t3);
    OKZG_MAC_ROW6(a, b[5], t5, t6, t7, t0, t1, t2, t3, t4);
    q = t5 * inv;
    OKZG_MAC_ROW6(m, q, t5, t6, t7, t0, t1, t2, t3, t4);
    uint64_t t[7] = {t6, t7, t0, t1, t2, t3, t4}, s[6];
    uint64_t br = limbs_sub(s, t, m, 6);
    if (t[6] || !br) memcpy(r, s, 8 * 6); else memcpy(r, t, 8 * 6);
}
#endif
int okzg_use_adx = 1;     /* run-time switch between the two forms (okzg_set_adx): bench.py times both */
/* CIOS Montgomery multiplication: r = a*b/2^(64n) mod m.  a may be any n-limb value, b < m. */
AINLINE void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m,
                      const uint64_t inv, const int n) {
#ifdef OKZG_HAVE_ADX
    if (okzg_use_adx) {
        if (n == 4) { mont_mul_adx4(r, a, b, m, inv); return; }
        if (n == 6) { mont_mul_adx6(r, a, b, m, inv); return; }
    }
#endif
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * inv;
        c = (u128)q * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) { c += (u128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[n]; t[n - 1] = (uint64_t)c; t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    uint64_t s[6];
    uint64_t br = limbs_sub(s, t, m, n);
    if (t[n] || !br) memcpy(r, s, 8 * n); else memcpy(r, t, 8 * n);
}
static void limbs_shr(uint64_t *a, int n, int k) {
    for (int i = 0; i < n; i++) a[i] = (a[i] >> k) | (i + 1 < n ? a[i + 1] << (64 - k) : 0);
}
static void limbs_sub_small(uint64_t *a, int n, uint64_t v) {
    for (int i = 0; i < n && v; i++) { uint64_t o = a[i]; a[i] -= v; v = o < v ? 1 : 0; }
}
static void limbs_add_small(uint64_t *a, int n, uint64_t v) {
    for (int i = 0; i < n && v; i++) { a[i] += v; v = a[i] < v ? 1 : 0; }
}
static uint64_t limbs_div_small(uint64_t *a, int n, uint64_t d) {
    u128 rem = 0;
    for (int i = n - 1; i >= 0; i--) { u128 cur = (rem << 64) | a[i]; a[i] = (uint64_t)(cur / d); rem = cur % d; }
    return (uint64_t)rem;
}
static void limbs_mul_small(uint64_t *a, int n, uint64_t v) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] * v; a[i] = (uint64_t)c; c >>= 64; }
}

/* ------------------------------------------------------------------ Fp  (blst_fp_*; used via G1/G2/pairing) */
void fp_add(fp_t *r, const fp_t *a, const fp_t *b) { mod_add(r->l, a->l, b->l, P_MOD, 6); }
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b) { mod_sub(r->l, a->l, b->l, P_MOD, 6); }
void fp_neg(fp_t *r, const fp_t *a) { fp_t z = {{0}}; mod_sub(r->l, z.l, a->l, P_MOD, 6); }
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b) { mont_mul(r->l, a->l, b->l, P_MOD, P_INV, 6); }
void fp_sqr(fp_t *r, const fp_t *a) { mont_mul(r->l, a->l, a->l, P_MOD, P_INV, 6); }
bool fp_is_zero(const fp_t *a) { return limbs_is_zero(a->l, 6); }
bool fp_eq(const fp_t *a, const fp_t *b) { return limbs_cmp(a->l, b->l, 6) == 0; }
static void fp_pow(fp_t *r, const fp_t *a, const uint64_t *e, int n) {
    fp_t acc = FP_ONE, base = *a;
    int top = n * 64 - 1;
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
    }
    *r = acc;
}
void fp_inv(fp_t *r, const fp_t *a) { fp_pow(r, a, EXP_P_MINUS_2, 6); }
bool fp_sqrt(fp_t *r, const fp_t *a) {
    fp_t s, chk; fp_pow(&s, a, EXP_P_PLUS_1_DIV_4, 6);
    fp_sqr(&chk, &s);
    if (!fp_eq(&chk, a)) return false;
    *r = s; return true;
}
static void fp_from_mont(uint64_t out[6], const fp_t *a) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(out, a->l, one, P_MOD, P_INV, 6);
}
bool fp_from_be(fp_t *r, const uint8_t in[48]) {
    uint64_t v[6];
    for (int i = 0; i < 6; i++) {
        uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | in[(5 - i) * 8 + j];
        v[i] = w;
    }
    if (limbs_cmp(v, P_MOD, 6) >= 0) return false;
    mont_mul(r->l, v, FP_R2.l, P_MOD, P_INV, 6);
    return true;
}
void fp_to_be(uint8_t out[48], const fp_t *a) {
    uint64_t v[6]; fp_from_mont(v, a);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) out[(5 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
}
bool fp_is_lex_largest(const fp_t *a) {
    uint64_t v[6]; fp_from_mont(v, a);
    return limbs_cmp(v, P_MINUS_1_DIV_2, 6) > 0;
}

/* ------------------------------------------------------------------ Fr  (blst_fr_add/sub/mul/sqr/eucl_inverse,
 * blst_fr_from_uint64, blst_uint64_from_fr, blst_scalar_from_bendian, blst_scalar_fr_check, blst_fr_from_scalar,
 * blst_scalar_from_fr, blst_bendian_from_scalar -- call sites: SURVEY.md section 2.2) */
void fr_add(fr_t *r, const fr_t *a, const fr_t *b) { mod_add(r->l, a->l, b->l, R_MOD, 4); }
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) { mod_sub(r->l, a->l, b->l, R_MOD, 4); }
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b) { mont_mul(r->l, a->l, b->l, R_MOD, R_INV, 4); }
void fr_sqr(fr_t *r, const fr_t *a) { mont_mul(r->l, a->l, a->l, R_MOD, R_INV, 4); }
bool fr_is_zero(const fr_t *a) { return limbs_is_zero(a->l, 4); }
bool fr_eq(const fr_t *a, const fr_t *b) { return limbs_cmp(a->l, b->l, 4) == 0; }
void fr_inv(fr_t *r, const fr_t *a) {
    fr_t acc = FR_ONE, base = *a;
    for (int i = 254; i >= 0; i--) {
        fr_sqr(&acc, &acc);
        if ((EXP_R_MINUS_2[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &base);
    }
    *r = acc;
}
static void fr_load_be(uint64_t v[4], const uint8_t in[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | in[(3 - i) * 8 + j];
        v[i] = w;
    }
}
bool fr_from_be_checked(fr_t *r, const uint8_t in[32]) {
    uint64_t v[4]; fr_load_be(v, in);
    if (limbs_cmp(v, R_MOD, 4) >= 0) return false;
    mont_mul(r->l, v, FR_R2.l, R_MOD, R_INV, 4);
    return true;
}
void fr_from_be_reduce(fr_t *r, const uint8_t in[32]) {
    /* blst_fr_from_scalar is a Montgomery multiplication by R^2, which reduces any 256-bit input */
    uint64_t v[4]; fr_load_be(v, in);
    mont_mul(r->l, v, FR_R2.l, R_MOD, R_INV, 4);
}
static void fr_from_mont(uint64_t out[4], const fr_t *a) {
    uint64_t one[4] = {1, 0, 0, 0};
    mont_mul(out, a->l, one, R_MOD, R_INV, 4);
}
void fr_to_be(uint8_t out[32], const fr_t *a) {
    uint64_t v[4]; fr_from_mont(v, a);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[(3 - i) * 8 + j] = (uint8_t)(v[i] >> (56 - 8 * j));
}
void fr_to_le_scalar(uint8_t out[32], const fr_t *a) {
    uint64_t v[4]; fr_from_mont(v, a);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[i * 8 + j] = (uint8_t)(v[i] >> (8 * j));
}
void fr_from_u64(fr_t *r, uint64_t x) {
    uint64_t v[4] = {x, 0, 0, 0};
    mont_mul(r->l, v, FR_R2.l, R_MOD, R_INV, 4);
}

/* ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2+1) */
void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
void fp2_neg(fp2_t *r, const fp2_t *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void fp2_conj(fp2_t *r, const fp2_t *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, s0, s1, m;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1); fp_mul(&m, &s0, &s1);
    fp_sub(&r->c0, &t0, &t1);
    fp_sub(&m, &m, &t0); fp_sub(&r->c1, &m, &t1);
}
void fp2_sqr(fp2_t *r, const fp2_t *a) {
    fp_t s, d, m;
    fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&r->c0, &s, &d); fp_add(&r->c1, &m, &m);
}
static void fp2_mul_fp(fp2_t *r, const fp2_t *a, const fp_t *s) { fp_mul(&r->c0, &a->c0, s); fp_mul(&r->c1, &a->c1, s); }
static void fp2_mul_xi(fp2_t *r, const fp2_t *a) { /* (1+u) */
    fp_t t0, t1; fp_sub(&t0, &a->c0, &a->c1); fp_add(&t1, &a->c0, &a->c1); r->c0 = t0; r->c1 = t1;
}
void fp2_inv(fp2_t *r, const fp2_t *a) {
    fp_t t0, t1; fp_sqr(&t0, &a->c0); fp_sqr(&t1, &a->c1); fp_add(&t0, &t0, &t1); fp_inv(&t0, &t0);
    fp_mul(&r->c0, &a->c0, &t0); fp_mul(&t1, &a->c1, &t0); fp_neg(&r->c1, &t1);
}
bool fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
bool fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_pow(fp2_t *r, const fp2_t *a, const uint64_t *e, int n) {
    fp2_t acc = {FP_ONE, FP_ZERO}, base = *a;
    for (int i = n * 64 - 1; i >= 0; i--) {
        fp2_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, &base);
    }
    *r = acc;
}
bool fp2_sqrt(fp2_t *r, const fp2_t *a) {
    /* p = 3 mod 4: Algorithm 9 of eprint 2012/685 */
    if (fp2_is_zero(a)) { *r = *a; return true; }
    fp2_t a1, alpha, x0, res, chk, minus_one = {FP_ZERO, FP_ZERO};
    fp_neg(&minus_one.c0, &FP_ONE);
    fp2_pow(&a1, a, EXP_P_MINUS_3_DIV_4, 6);
    fp2_sqr(&alpha, &a1); fp2_mul(&alpha, &alpha, a);
    fp2_mul(&x0, &a1, a);
    if (fp2_eq(&alpha, &minus_one)) {
        fp_neg(&res.c0, &x0.c1); res.c1 = x0.c0;   /* u * x0 */
    } else {
        fp2_t b; fp_add(&alpha.c0, &alpha.c0, &FP_ONE);
        fp2_pow(&b, &alpha, EXP_P_MINUS_1_DIV_2, 6);
        fp2_mul(&res, &b, &x0);
    }
    fp2_sqr(&chk, &res);
    if (!fp2_eq(&chk, a)) return false;
    *r = res; return true;
}

/* ------------------------------------------------------------------ Fp6 = Fp2[v]/(v^3 - xi) */
static void fp6_add(fp6_t *r, const fp6_t *a, const fp6_t *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6_t *r, const fp6_t *a, const fp6_t *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6_t *r, const fp6_t *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static void fp6_mul(fp6_t *r, const fp6_t *a, const fp6_t *b) {
    fp2_t t0, t1, t2, s0, s1, m, c0, c1, c2;
    fp2_mul(&t0, &a->c0, &b->c0); fp2_mul(&t1, &a->c1, &b->c1); fp2_mul(&t2, &a->c2, &b->c2);
    fp2_add(&s0, &a->c1, &a->c2); fp2_add(&s1, &b->c1, &b->c2); fp2_mul(&m, &s0, &s1);
    fp2_sub(&m, &m, &t1); fp2_sub(&m, &m, &t2); fp2_mul_xi(&m, &m); fp2_add(&c0, &t0, &m);
    fp2_add(&s0, &a->c0, &a->c1); fp2_add(&s1, &b->c0, &b->c1); fp2_mul(&m, &s0, &s1);
    fp2_sub(&m, &m, &t0); fp2_sub(&m, &m, &t1); fp2_mul_xi(&s0, &t2); fp2_add(&c1, &m, &s0);
    fp2_add(&s0, &a->c0, &a->c2); fp2_add(&s1, &b->c0, &b->c2); fp2_mul(&m, &s0, &s1);
    fp2_sub(&m, &m, &t0); fp2_sub(&m, &m, &t2); fp2_add(&c2, &m, &t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fp6_mul_v(fp6_t *r, const fp6_t *a) {
    fp2_t t; fp2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t;
}
/* a * (c0 + c1 v) */
static void fp6_mul_by_01(fp6_t *r, const fp6_t *a, const fp2_t *c0, const fp2_t *c1) {
    fp2_t a0c0, a1c1, a2c1, t, r0, r1, r2;
    fp2_mul(&a0c0, &a->c0, c0); fp2_mul(&a1c1, &a->c1, c1); fp2_mul(&a2c1, &a->c2, c1);
    fp2_mul_xi(&t, &a2c1); fp2_add(&r0, &a0c0, &t);
    fp2_mul(&t, &a->c0, c1); fp2_mul(&r1, &a->c1, c0); fp2_add(&r1, &r1, &t);
    fp2_mul(&r2, &a->c2, c0); fp2_add(&r2, &r2, &a1c1);
    r->c0 = r0; r->c1 = r1; r->c2 = r2;
}
/* a * (c1 v) */
static void fp6_mul_by_1(fp6_t *r, const fp6_t *a, const fp2_t *c1) {
    fp2_t r0, r1, r2;
    fp2_mul(&r0, &a->c2, c1); fp2_mul_xi(&r0, &r0);
    fp2_mul(&r1, &a->c0, c1); fp2_mul(&r2, &a->c1, c1);
    r->c0 = r0; r->c1 = r1; r->c2 = r2;
}
static void fp6_inv(fp6_t *r, const fp6_t *a) {
    fp2_t t0, t1, t2, m, d;
    fp2_sqr(&t0, &a->c0); fp2_mul(&m, &a->c1, &a->c2); fp2_mul_xi(&m, &m); fp2_sub(&t0, &t0, &m);
    fp2_sqr(&t1, &a->c2); fp2_mul_xi(&t1, &t1); fp2_mul(&m, &a->c0, &a->c1); fp2_sub(&t1, &t1, &m);
    fp2_sqr(&t2, &a->c1); fp2_mul(&m, &a->c0, &a->c2); fp2_sub(&t2, &t2, &m);
    fp2_mul(&d, &a->c2, &t1); fp2_mul(&m, &a->c1, &t2); fp2_add(&d, &d, &m); fp2_mul_xi(&d, &d);
    fp2_mul(&m, &a->c0, &t0); fp2_add(&d, &d, &m); fp2_inv(&d, &d);
    fp2_mul(&r->c0, &t0, &d); fp2_mul(&r->c1, &t1, &d); fp2_mul(&r->c2, &t2, &d);
}
static void fp6_frob(fp6_t *r, const fp6_t *a) {
    fp2_t t;
    fp2_conj(&r->c0, &a->c0);
    fp2_conj(&t, &a->c1); fp2_mul(&r->c1, &t, &FROB_V1);
    fp2_conj(&t, &a->c2); fp2_mul(&r->c2, &t, &FROB_V2);
}

/* ------------------------------------------------------------------ Fp12 = Fp6[w]/(w^2 - v)
 * (blst_fp12_mul, blst_fp12_is_one, and the internals of blst_final_exp: utils.rs:206-212) */
void fp12_set_one(fp12_t *r) { memset(r, 0, sizeof *r); r->c0.c0.c0 = FP_ONE; }
bool fp12_is_one(const fp12_t *a) { fp12_t o; fp12_set_one(&o); return memcmp(a, &o, sizeof o) == 0; }
void fp12_mul(fp12_t *r, const fp12_t *a, const fp12_t *b) {
    fp6_t t0, t1, s0, s1, m;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s0, &a->c0, &a->c1); fp6_add(&s1, &b->c0, &b->c1); fp6_mul(&m, &s0, &s1);
    fp6_sub(&m, &m, &t0); fp6_sub(&r->c1, &m, &t1);
    fp6_mul_v(&t1, &t1); fp6_add(&r->c0, &t0, &t1);
}
void fp12_sqr(fp12_t *r, const fp12_t *a) {
    /* (a0 + a1 w)^2 = (a0^2 + v a1^2) + 2 a0 a1 w, via (a0+a1)(a0+v a1) - a0a1 - v a0a1 */
    fp6_t ab, s0, s1, m, vab;
    fp6_mul(&ab, &a->c0, &a->c1);
    fp6_add(&s0, &a->c0, &a->c1); fp6_mul_v(&s1, &a->c1); fp6_add(&s1, &s1, &a->c0);
    fp6_mul(&m, &s0, &s1); fp6_mul_v(&vab, &ab);
    fp6_sub(&m, &m, &ab); fp6_sub(&r->c0, &m, &vab);
    fp6_add(&r->c1, &ab, &ab);
}
void fp12_conj(fp12_t *r, const fp12_t *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
void fp12_inv(fp12_t *r, const fp12_t *a) {
    fp6_t t0, t1;
    fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1);
    fp6_sub(&t0, &t0, &t1); fp6_inv(&t0, &t0);
    fp6_mul(&r->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&r->c1, &t1);
}
void fp12_frob(fp12_t *r, const fp12_t *a) {
    fp6_t t;
    fp6_frob(&r->c0, &a->c0); fp6_frob(&t, &a->c1);
    fp2_mul(&r->c1.c0, &t.c0, &FROB_W); fp2_mul(&r->c1.c1, &t.c1, &FROB_W); fp2_mul(&r->c1.c2, &t.c2, &FROB_W);
}
/* f * (c0 + c1 v + c4 v w) */
static void fp12_mul_by_014(fp12_t *f, const fp2_t *c0, const fp2_t *c1, const fp2_t *c4) {
    fp6_t t0, t1, s, m; fp2_t c14;
    fp6_mul_by_01(&t0, &f->c0, c0, c1);
    fp6_mul_by_1(&t1, &f->c1, c4);
    fp6_add(&s, &f->c0, &f->c1); fp2_add(&c14, c1, c4);
    fp6_mul_by_01(&m, &s, c0, &c14);
    fp6_sub(&m, &m, &t0); fp6_sub(&f->c1, &m, &t1);
    fp6_mul_v(&t1, &t1); fp6_add(&f->c0, &t0, &t1);
}

/* ------------------------------------------------------------------ G1  (blst_p1_add_or_double, blst_p1_cneg,
 * blst_p1_mult, blst_p1_to_affine, blst_p1_from_affine, blst_p1_compress, blst_p1_uncompress, blst_p1_is_inf,
 * blst_p1_in_g1 -- call sites utils.rs:126-140, 162-170, 221-227, 282-310) */
void g1_set_inf(g1_t *r) { memset(r, 0, sizeof *r); }
bool g1_is_inf(const g1_t *a) { return fp_is_zero(&a->z); }
void g1_neg(g1_t *r, const g1_t *a) { r->x = a->x; r->z = a->z; fp_neg(&r->y, &a->y); }
void g1_dbl(g1_t *r, const g1_t *p) {
    fp_t A, B, C, D, E, F, t, X3, Y3, Z3;
    fp_sqr(&A, &p->x); fp_sqr(&B, &p->y); fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B); fp_sqr(&t, &t); fp_sub(&t, &t, &A); fp_sub(&t, &t, &C); fp_add(&D, &t, &t);
    fp_add(&E, &A, &A); fp_add(&E, &E, &A); fp_sqr(&F, &E);
    fp_sub(&X3, &F, &D); fp_sub(&X3, &X3, &D);
    fp_mul(&Z3, &p->y, &p->z); fp_add(&Z3, &Z3, &Z3);
    fp_sub(&t, &D, &X3); fp_mul(&Y3, &E, &t);
    fp_add(&C, &C, &C); fp_add(&C, &C, &C); fp_add(&C, &C, &C); fp_sub(&Y3, &Y3, &C);
    r->x = X3; r->y = Y3; r->z = Z3;
}
void g1_add(g1_t *r, const g1_t *a, const g1_t *b) {
    if (g1_is_inf(a)) { *r = *b; return; }
    if (g1_is_inf(b)) { *r = *a; return; }
    fp_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
    fp_sqr(&Z1Z1, &a->z); fp_sqr(&Z2Z2, &b->z);
    fp_mul(&U1, &a->x, &Z2Z2); fp_mul(&U2, &b->x, &Z1Z1);
    fp_mul(&S1, &a->y, &b->z); fp_mul(&S1, &S1, &Z2Z2);
    fp_mul(&S2, &b->y, &a->z); fp_mul(&S2, &S2, &Z1Z1);
    fp_sub(&H, &U2, &U1); fp_sub(&Rr, &S2, &S1);
    if (fp_is_zero(&H)) {
        if (fp_is_zero(&Rr)) { g1_dbl(r, a); } else { g1_set_inf(r); }
        return;
    }
    fp_sqr(&HH, &H); fp_mul(&HHH, &H, &HH); fp_mul(&V, &U1, &HH);
    fp_sqr(&X3, &Rr); fp_sub(&X3, &X3, &HHH); fp_sub(&X3, &X3, &V); fp_sub(&X3, &X3, &V);
    fp_sub(&t, &V, &X3); fp_mul(&Y3, &Rr, &t); fp_mul(&t, &S1, &HHH); fp_sub(&Y3, &Y3, &t);
    fp_mul(&Z3, &a->z, &b->z); fp_mul(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
void g1_from_affine(g1_t *r, const g1a_t *a) {
    if (a->inf) { g1_set_inf(r); return; }
    r->x = a->x; r->y = a->y; r->z = FP_ONE;
}
void g1_add_affine(g1_t *r, const g1_t *a, const g1a_t *b) {
    if (b->inf) { *r = *a; return; }
    if (g1_is_inf(a)) { g1_from_affine(r, b); return; }
    fp_t Z1Z1, U2, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
    fp_sqr(&Z1Z1, &a->z); fp_mul(&U2, &b->x, &Z1Z1);
    fp_mul(&S2, &b->y, &a->z); fp_mul(&S2, &S2, &Z1Z1);
    fp_sub(&H, &U2, &a->x); fp_sub(&Rr, &S2, &a->y);
    if (fp_is_zero(&H)) {
        if (fp_is_zero(&Rr)) { g1_dbl(r, a); } else { g1_set_inf(r); }
        return;
    }
    fp_sqr(&HH, &H); fp_mul(&HHH, &H, &HH); fp_mul(&V, &a->x, &HH);
    fp_sqr(&X3, &Rr); fp_sub(&X3, &X3, &HHH); fp_sub(&X3, &X3, &V); fp_sub(&X3, &X3, &V);
    fp_sub(&t, &V, &X3); fp_mul(&Y3, &Rr, &t); fp_mul(&t, &a->y, &HHH); fp_sub(&Y3, &Y3, &t);
    fp_mul(&Z3, &a->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
void g1_mul(g1_t *r, const g1_t *a, const uint8_t s[32], int nbits) {
    /* 4-bit fixed window, MSB first (any correct algorithm yields the same group element as blst_p1_mult) */
    g1_t tab[16]; g1_set_inf(&tab[0]); tab[1] = *a;
    for (int i = 2; i < 16; i++) { if (i & 1) g1_add(&tab[i], &tab[i - 1], a); else g1_dbl(&tab[i], &tab[i / 2]); }
    g1_t acc; g1_set_inf(&acc);
    int nn = (nbits + 3) / 4;
    for (int i = nn - 1; i >= 0; i--) {
        g1_dbl(&acc, &acc); g1_dbl(&acc, &acc); g1_dbl(&acc, &acc); g1_dbl(&acc, &acc);
        int nib = (i / 2 < 32) ? (s[i / 2] >> ((i & 1) * 4)) & 15 : 0;
        if (nib) g1_add(&acc, &acc, &tab[nib]);
    }
    *r = acc;
}
void g1_to_affine(g1a_t *r, const g1_t *a) {
    if (g1_is_inf(a)) { memset(r, 0, sizeof *r); r->inf = true; return; }
    fp_t zi, zi2, zi3; fp_inv(&zi, &a->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&r->x, &a->x, &zi2); fp_mul(&r->y, &a->y, &zi3); r->inf = false;
}
bool g1_eq(const g1_t *a, const g1_t *b) {
    g1a_t x, y; g1_to_affine(&x, a); g1_to_affine(&y, b);
    if (x.inf || y.inf) return x.inf && y.inf;
    return fp_eq(&x.x, &y.x) && fp_eq(&x.y, &y.y);
}
void g1_compress(uint8_t out[48], const g1_t *a) {
    g1a_t p; g1_to_affine(&p, a);
    if (p.inf) { memset(out, 0, 48); out[0] = 0xc0; return; }
    fp_to_be(out, &p.x);
    out[0] |= 0x80 | (fp_is_lex_largest(&p.y) ? 0x20 : 0);
}
int g1_uncompress(g1a_t *r, const uint8_t in[48]) {
    if (!(in[0] & 0x80)) return 1;                       /* uncompressed encoding: bad encoding */
    if (in[0] & 0x40) {
        if (in[0] & 0x3f) return 1;
        for (int i = 1; i < 48; i++) if (in[i]) return 1;
        memset(r, 0, sizeof *r); r->inf = true; return 0;
    }
    uint8_t tmp[48]; memcpy(tmp, in, 48); tmp[0] &= 0x1f;
    fp_t x, y2, y, four;
    if (!fp_from_be(&x, tmp)) return 1;
    fp_sqr(&y2, &x); fp_mul(&y2, &y2, &x);
    fp_add(&four, &FP_ONE, &FP_ONE); fp_add(&four, &four, &four); fp_add(&y2, &y2, &four);
    if (!fp_sqrt(&y, &y2)) return 2;                     /* not on curve */
    if (fp_is_lex_largest(&y) != !!(in[0] & 0x20)) fp_neg(&y, &y);
    r->x = x; r->y = y; r->inf = false;
    return 0;
}
bool g1_in_subgroup(const g1_t *a) {
    /* [r]P == infinity (same predicate as blst_p1_in_g1) */
    uint8_t s[32];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) s[i * 8 + j] = (uint8_t)(R_MOD[i] >> (8 * j));
    g1_t t; g1_mul(&t, a, s, 256);
    return g1_is_inf(&t);
}

/* ------------------------------------------------------------------ G2  (blst_p2_*; utils.rs:143-157, 175-183, 203-204;
 * kzg.rs:877-884) */
bool g2_is_inf(const g2_t *a) { return fp2_is_zero(&a->z); }
void g2_neg(g2_t *r, const g2_t *a) { r->x = a->x; r->z = a->z; fp2_neg(&r->y, &a->y); }
void g2_dbl(g2_t *r, const g2_t *p) {
    fp2_t A, B, C, D, E, F, t, X3, Y3, Z3;
    fp2_sqr(&A, &p->x); fp2_sqr(&B, &p->y); fp2_sqr(&C, &B);
    fp2_add(&t, &p->x, &B); fp2_sqr(&t, &t); fp2_sub(&t, &t, &A); fp2_sub(&t, &t, &C); fp2_add(&D, &t, &t);
    fp2_add(&E, &A, &A); fp2_add(&E, &E, &A); fp2_sqr(&F, &E);
    fp2_sub(&X3, &F, &D); fp2_sub(&X3, &X3, &D);
    fp2_mul(&Z3, &p->y, &p->z); fp2_add(&Z3, &Z3, &Z3);
    fp2_sub(&t, &D, &X3); fp2_mul(&Y3, &E, &t);
    fp2_add(&C, &C, &C); fp2_add(&C, &C, &C); fp2_add(&C, &C, &C); fp2_sub(&Y3, &Y3, &C);
    r->x = X3; r->y = Y3; r->z = Z3;
}
void g2_add(g2_t *r, const g2_t *a, const g2_t *b) {
    if (g2_is_inf(a)) { *r = *b; return; }
    if (g2_is_inf(b)) { *r = *a; return; }
    fp2_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
    fp2_sqr(&Z1Z1, &a->z); fp2_sqr(&Z2Z2, &b->z);
    fp2_mul(&U1, &a->x, &Z2Z2); fp2_mul(&U2, &b->x, &Z1Z1);
    fp2_mul(&S1, &a->y, &b->z); fp2_mul(&S1, &S1, &Z2Z2);
    fp2_mul(&S2, &b->y, &a->z); fp2_mul(&S2, &S2, &Z1Z1);
    fp2_sub(&H, &U2, &U1); fp2_sub(&Rr, &S2, &S1);
    if (fp2_is_zero(&H)) {
        if (fp2_is_zero(&Rr)) { g2_dbl(r, a); } else { memset(r, 0, sizeof *r); }
        return;
    }
    fp2_sqr(&HH, &H); fp2_mul(&HHH, &H, &HH); fp2_mul(&V, &U1, &HH);
    fp2_sqr(&X3, &Rr); fp2_sub(&X3, &X3, &HHH); fp2_sub(&X3, &X3, &V); fp2_sub(&X3, &X3, &V);
    fp2_sub(&t, &V, &X3); fp2_mul(&Y3, &Rr, &t); fp2_mul(&t, &S1, &HHH); fp2_sub(&Y3, &Y3, &t);
    fp2_mul(&Z3, &a->z, &b->z); fp2_mul(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
void g2_mul(g2_t *r, const g2_t *a, const uint8_t s[32], int nbits) {
    g2_t acc; memset(&acc, 0, sizeof acc);
    for (int i = nbits - 1; i >= 0; i--) {
        g2_dbl(&acc, &acc);
        if (i < 256 && ((s[i / 8] >> (i % 8)) & 1)) g2_add(&acc, &acc, a);
    }
    *r = acc;
}
void g2_from_affine(g2_t *r, const g2a_t *a) {
    if (a->inf) { memset(r, 0, sizeof *r); return; }
    r->x = a->x; r->y = a->y; r->z.c0 = FP_ONE; r->z.c1 = FP_ZERO;
}
void g2_to_affine(g2a_t *r, const g2_t *a) {
    if (g2_is_inf(a)) { memset(r, 0, sizeof *r); r->inf = true; return; }
    fp2_t zi, zi2, zi3; fp2_inv(&zi, &a->z); fp2_sqr(&zi2, &zi); fp2_mul(&zi3, &zi2, &zi);
    fp2_mul(&r->x, &a->x, &zi2); fp2_mul(&r->y, &a->y, &zi3); r->inf = false;
}
int g2_uncompress(g2a_t *r, const uint8_t in[96]) {
    if (!(in[0] & 0x80)) return 1;
    if (in[0] & 0x40) {
        if (in[0] & 0x3f) return 1;
        for (int i = 1; i < 96; i++) if (in[i]) return 1;
        memset(r, 0, sizeof *r); r->inf = true; return 0;
    }
    uint8_t tmp[48]; memcpy(tmp, in, 48); tmp[0] &= 0x1f;
    fp2_t x, y2, y, b2;
    if (!fp_from_be(&x.c1, tmp)) return 1;
    if (!fp_from_be(&x.c0, in + 48)) return 1;
    fp2_sqr(&y2, &x); fp2_mul(&y2, &y2, &x);
    fp_add(&b2.c0, &FP_ONE, &FP_ONE); fp_add(&b2.c0, &b2.c0, &b2.c0); b2.c1 = b2.c0;
    fp2_add(&y2, &y2, &b2);
    if (!fp2_sqrt(&y, &y2)) return 2;
    bool big = fp_is_zero(&y.c1) ? fp_is_lex_largest(&y.c0) : fp_is_lex_largest(&y.c1);
    if (big != !!(in[0] & 0x20)) fp2_neg(&y, &y);
    r->x = x; r->y = y; r->inf = false;
    return 0;
}

/* ------------------------------------------------------------------ pairing  (blst_miller_loop, blst_final_exp;
 * utils.rs:206-212).  Lines are scaled by w^3 and an Fp2 factor -- both vanish in the final exponentiation:
 *   l = c0 + c1 v + c4 v w,  tangent at T=(X,Y,Z) Jacobian:  c0 = 3X^3 - 2Y^2, c1 = -3X^2 Z^2 xP, c4 = Z3 Z^2 yP
 *   chord T,Q:  c0 = R xQ - yQ Z3, c1 = -R xP, c4 = Z3 yP   with H = xQ Z^2 - X, R = yQ Z^3 - Y, Z3 = Z H. */
void miller_loop(fp12_t *f, const g2a_t *q, const g1a_t *p) {
    fp12_set_one(f);
    if (q->inf || p->inf) return;
    fp2_t X = q->x, Y = q->y, Z = {FP_ONE, FP_ZERO};
    for (int i = 62; i >= 0; i--) {
        fp2_t A, B, C, D, E, F, t, X3, Y3, Z3, Zsq, c0, c1, c4;
        fp12_sqr(f, f);
        fp2_sqr(&A, &X); fp2_sqr(&B, &Y); fp2_sqr(&C, &B); fp2_sqr(&Zsq, &Z);
        fp2_add(&t, &X, &B); fp2_sqr(&t, &t); fp2_sub(&t, &t, &A); fp2_sub(&t, &t, &C); fp2_add(&D, &t, &t);
        fp2_add(&E, &A, &A); fp2_add(&E, &E, &A); fp2_sqr(&F, &E);
        fp2_sub(&X3, &F, &D); fp2_sub(&X3, &X3, &D);
        fp2_mul(&Z3, &Y, &Z); fp2_add(&Z3, &Z3, &Z3);
        fp2_sub(&t, &D, &X3); fp2_mul(&Y3, &E, &t);
        fp2_add(&C, &C, &C); fp2_add(&C, &C, &C); fp2_add(&C, &C, &C); fp2_sub(&Y3, &Y3, &C);
        fp2_mul(&c0, &E, &X); fp2_sub(&c0, &c0, &B); fp2_sub(&c0, &c0, &B);
        fp2_mul(&c1, &E, &Zsq); fp2_neg(&c1, &c1); fp2_mul_fp(&c1, &c1, &p->x);
        fp2_mul(&c4, &Z3, &Zsq); fp2_mul_fp(&c4, &c4, &p->y);
        fp12_mul_by_014(f, &c0, &c1, &c4);
        X = X3; Y = Y3; Z = Z3;
        if ((X_ABS >> i) & 1) {
            fp2_t U2, S2, H, Rr, HH, HHH, V;
            fp2_sqr(&Zsq, &Z); fp2_mul(&U2, &q->x, &Zsq);
            fp2_mul(&S2, &q->y, &Z); fp2_mul(&S2, &S2, &Zsq);
            fp2_sub(&H, &U2, &X); fp2_sub(&Rr, &S2, &Y);
            fp2_sqr(&HH, &H); fp2_mul(&HHH, &H, &HH); fp2_mul(&V, &X, &HH);
            fp2_sqr(&X3, &Rr); fp2_sub(&X3, &X3, &HHH); fp2_sub(&X3, &X3, &V); fp2_sub(&X3, &X3, &V);
            fp2_sub(&t, &V, &X3); fp2_mul(&Y3, &Rr, &t); fp2_mul(&t, &Y, &HHH); fp2_sub(&Y3, &Y3, &t);
            fp2_mul(&Z3, &Z, &H);
            fp2_mul(&c0, &Rr, &q->x); fp2_mul(&t, &q->y, &Z3); fp2_sub(&c0, &c0, &t);
            fp2_neg(&c1, &Rr); fp2_mul_fp(&c1, &c1, &p->x);
            fp2_mul_fp(&c4, &Z3, &p->y);
            fp12_mul_by_014(f, &c0, &c1, &c4);
            X = X3; Y = Y3; Z = Z3;
        }
    }
    fp12_conj(f, f);  /* x < 0 */
}
static void cyc_exp_x(fp12_t *r, const fp12_t *a) {
    /* a^x for a in the cyclotomic subgroup; x<0 so conjugate at the end */
    fp12_t acc = *a;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(&acc, &acc);
        if ((X_ABS >> i) & 1) fp12_mul(&acc, &acc, a);
    }
    fp12_conj(r, &acc);
}
bool final_exp_is_one(const fp12_t *fin) {
    /* easy part (p^6-1)(p^2+1), then f^(3(p^4-p^2+1)/r) = f^((x-1)^2 (x+p)(x^2+p^2-1)) * f^3 */
    fp12_t f, t, a, b, c, d;
    fp12_conj(&t, fin); fp12_inv(&f, fin); fp12_mul(&f, &t, &f);
    fp12_frob(&t, &f); fp12_frob(&t, &t); fp12_mul(&f, &t, &f);
    cyc_exp_x(&a, &f); fp12_conj(&t, &f); fp12_mul(&a, &a, &t);
    cyc_exp_x(&b, &a); fp12_conj(&t, &a); fp12_mul(&a, &b, &t);
    cyc_exp_x(&b, &a); fp12_frob(&t, &a); fp12_mul(&b, &b, &t);
    cyc_exp_x(&c, &b); cyc_exp_x(&c, &c);
    fp12_frob(&t, &b); fp12_frob(&t, &t); fp12_mul(&c, &c, &t);
    fp12_conj(&t, &b); fp12_mul(&c, &c, &t);
    fp12_sqr(&d, &f); fp12_mul(&d, &d, &f);
    fp12_mul(&c, &c, &d);
    return fp12_is_one(&c);
}

/* ------------------------------------------------------------------ SHA-256 (blst_sha256) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t st[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
#if defined(__x86_64__) && defined(__SHA__) && defined(__SSE4_1__) && defined(__SSSE3__) && !defined(OKZG_NO_ADX)
#include <immintrin.h>
#define OKZG_HAVE_SHANI 1
/* The same compression with the SHA extensions (one message; part of the "blst-class" form of the baseline: blst hashes with them too).
   sha256rnds2 works on the state as (A,B,E,F) / (C,D,G,H); W_G = msg2(msg1(W_{G-4}, W_{G-3}) + alignr(W_{G-1}, W_{G-2}, 4), W_{G-1}). */
static void sha256_blocks_shani(uint32_t st[8], const uint8_t *p, size_t nblocks) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i abcd = _mm_loadu_si128((const __m128i *)&st[0]), efgh = _mm_loadu_si128((const __m128i *)&st[4]);
    abcd = _mm_shuffle_epi32(abcd, 0xB1);                       /* b a d c */
    efgh = _mm_shuffle_epi32(efgh, 0x1B);                       /* h g f e */
    __m128i s0 = _mm_alignr_epi8(abcd, efgh, 8);                /* (A,B,E,F) */
    __m128i s1 = _mm_blend_epi16(efgh, abcd, 0xF0);             /* (C,D,G,H) */
    for (size_t b = 0; b < nblocks; b++, p += 64) {
        const __m128i k0 = s0, k1 = s1;
        __m128i w[4];
        for (int q = 0; q < 4; q++) w[q] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * q)), bswap);
        for (int g = 0; g < 16; g++) {
            if (g >= 4) {
                __m128i x = _mm_sha256msg1_epu32(w[g & 3], w[(g + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(w[(g + 3) & 3], w[(g + 2) & 3], 4));
                w[g & 3] = _mm_sha256msg2_epu32(x, w[(g + 3) & 3]);
            }
            __m128i m = _mm_add_epi32(w[g & 3], _mm_loadu_si128((const __m128i *)&K256[4 * g]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, m);
            m = _mm_shuffle_epi32(m, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, m);
        }
        s0 = _mm_add_epi32(s0, k0); s1 = _mm_add_epi32(s1, k1);
    }
    const __m128i t = _mm_shuffle_epi32(s0, 0x1B), u = _mm_shuffle_epi32(s1, 0xB1);
    _mm_storeu_si128((__m128i *)&st[0], _mm_blend_epi16(t, u, 0xF0));
    _mm_storeu_si128((__m128i *)&st[4], _mm_alignr_epi8(u, t, 8));
}
#endif
/* 1 if this build has the mulx / adcx / adox field products (and the SHA extensions): okzg_set_adx switches them on and off at run time */
int okzg_have_adx(void) {
#ifdef OKZG_HAVE_ADX
    return 1;
#else
    return 0;
#endif
}
void okzg_set_adx(int on) { okzg_use_adx = on != 0; }
static void sha256_blocks(uint32_t st[8], const uint8_t *p, size_t nblocks) {
#ifdef OKZG_HAVE_SHANI
    if (okzg_use_adx) { sha256_blocks_shani(st, p, nblocks); return; }
#endif
    for (size_t i = 0; i < nblocks; i++) sha256_block(st, p + 64 * i);
}
void sha256(uint8_t out[32], const uint8_t *msg, size_t len) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    sha256_blocks(st, msg, full);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_blocks(st, tail, tl / 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = st[i] >> 24; out[4 * i + 1] = st[i] >> 16; out[4 * i + 2] = st[i] >> 8; out[4 * i + 3] = st[i]; }
}

/* ------------------------------------------------------------------ start-up constants */
static uint64_t neg_inv64(uint64_t m0) {
    uint64_t x = 1;
    for (int i = 0; i < 7; i++) x *= 2 - m0 * x;
    return (uint64_t)0 - x;
}
static void compute_r_and_r2(uint64_t *one, uint64_t *r2, const uint64_t *m, int n) {
    uint64_t v[6] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 2 * 64 * n; i++) {
        if (i == 64 * n) memcpy(one, v, 8 * n);
        mod_add(v, v, v, m, n);
    }
    memcpy(r2, v, 8 * n);
}
static fp_t fp_from_hex_be(const char *hex) {
    uint8_t b[48];
    for (int i = 0; i < 48; i++) {
        unsigned v = 0;
        for (int k = 0; k < 2; k++) { char ch = hex[2 * i + k]; v = v * 16 + (ch <= '9' ? ch - '0' : ch - 'a' + 10); }
        b[i] = (uint8_t)v;
    }
    fp_t r; fp_from_be(&r, b); return r;
}
static void bls_init_once(void) {
    P_INV = neg_inv64(P_MOD[0]); R_INV = neg_inv64(R_MOD[0]);
    memset(&FP_ZERO, 0, sizeof FP_ZERO); memset(&FR_ZERO, 0, sizeof FR_ZERO);
    compute_r_and_r2(FP_ONE.l, FP_R2.l, P_MOD, 6);
    compute_r_and_r2(FR_ONE.l, FR_R2.l, R_MOD, 4);
    memcpy(EXP_P_MINUS_2, P_MOD, 48); limbs_sub_small(EXP_P_MINUS_2, 6, 2);
    memcpy(EXP_P_PLUS_1_DIV_4, P_MOD, 48); limbs_add_small(EXP_P_PLUS_1_DIV_4, 6, 1); limbs_shr(EXP_P_PLUS_1_DIV_4, 6, 2);
    memcpy(EXP_P_MINUS_3_DIV_4, P_MOD, 48); limbs_sub_small(EXP_P_MINUS_3_DIV_4, 6, 3); limbs_shr(EXP_P_MINUS_3_DIV_4, 6, 2);
    memcpy(EXP_P_MINUS_1_DIV_2, P_MOD, 48); limbs_sub_small(EXP_P_MINUS_1_DIV_2, 6, 1); limbs_shr(EXP_P_MINUS_1_DIV_2, 6, 1);
    memcpy(P_MINUS_1_DIV_2, EXP_P_MINUS_1_DIV_2, 48);
    memcpy(EXP_R_MINUS_2, R_MOD, 32); limbs_sub_small(EXP_R_MINUS_2, 4, 2);
    /* Frobenius: v^p = xi^((p-1)/3) v, w^p = xi^((p-1)/6) w */
    uint64_t e[6];
    fp2_t xi = {FP_ONE, FP_ONE};
    memcpy(e, P_MOD, 48); limbs_sub_small(e, 6, 1); limbs_div_small(e, 6, 3); fp2_pow(&FROB_V1, &xi, e, 6);
    limbs_mul_small(e, 6, 2); fp2_pow(&FROB_V2, &xi, e, 6);
    memcpy(e, P_MOD, 48); limbs_sub_small(e, 6, 1); limbs_div_small(e, 6, 6); fp2_pow(&FROB_W, &xi, e, 6);
    G1_GENERATOR_J.x = fp_from_hex_be("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb");
    G1_GENERATOR_J.y = fp_from_hex_be("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1");
    G1_GENERATOR_J.z = FP_ONE;
    G2_GENERATOR_J.x.c0 = fp_from_hex_be("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8");
    G2_GENERATOR_J.x.c1 = fp_from_hex_be("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e");
    G2_GENERATOR_J.y.c0 = fp_from_hex_be("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801");
    G2_GENERATOR_J.y.c1 = fp_from_hex_be("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be");
    G2_GENERATOR_J.z.c0 = FP_ONE; G2_GENERATOR_J.z.c1 = FP_ZERO;
}
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
void bls_init(void) { pthread_once(&g_once, bls_init_once); }
