/*
 * oracle/bls12_381.h -- CPU restatement of the BLS12-381 arithmetic the reference gets from blst.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call anything under oracle/.  The product (kzg_rust_amd/, libkzg355.so) never does.
 *
 * The reference (pawanjay176/kzg_rust) delegates all field/curve/pairing/SHA-256 work to the
 * third-party crate blst 0.3.11 (Cargo.toml:9, Cargo.lock:44-47), whose source is NOT under
 * /root/reference.  What follows restates the published algorithms (Montgomery arithmetic, the
 * Fp2/Fp6/Fp12 tower, Jacobian group law, ZCash point serialisation, optimal-ate pairing with
 * x = -0xd201000000010000, FIPS 180-4 SHA-256) and is anchored on the reference's call sites
 * (src/utils.rs, src/kzg.rs -- cited per function in kzg_oracle.c) and on its 208 golden vectors
 * (tests/golden/vectors.json).  Parity is PINNED: tests/test_oracle_vectors.py runs every vector.
 *
 * Representation: 64-bit limbs, little-endian limb order, Montgomery form with R = 2^384 (Fp)
 * and R = 2^256 (Fr).  All constants other than p, r, xi and the generators are computed at
 * start-up (bls_init) so that this file does not share generated tables with the product.
 */
#ifndef ORACLE_BLS12_381_H
#define ORACLE_BLS12_381_H
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

typedef struct { uint64_t l[6]; } fp_t;
typedef struct { uint64_t l[4]; } fr_t;
typedef struct { fp_t c0, c1; } fp2_t;
typedef struct { fp2_t c0, c1, c2; } fp6_t;
typedef struct { fp6_t c0, c1; } fp12_t;
typedef struct { fp_t x, y, z; } g1_t;          /* Jacobian; z == 0 <=> infinity */
typedef struct { fp_t x, y; bool inf; } g1a_t;  /* affine */
typedef struct { fp2_t x, y, z; } g2_t;
typedef struct { fp2_t x, y; bool inf; } g2a_t;

void bls_init(void);

/* Fp */
extern fp_t FP_ONE, FP_ZERO;
void fp_add(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *r, const fp_t *a);
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sqr(fp_t *r, const fp_t *a);
void fp_inv(fp_t *r, const fp_t *a);
bool fp_sqrt(fp_t *r, const fp_t *a);
bool fp_is_zero(const fp_t *a);
bool fp_eq(const fp_t *a, const fp_t *b);
bool fp_from_be(fp_t *r, const uint8_t in[48]);  /* false if >= p */
void fp_to_be(uint8_t out[48], const fp_t *a);
bool fp_is_lex_largest(const fp_t *a);           /* canonical value > (p-1)/2 */

/* Fr */
extern fr_t FR_ONE, FR_ZERO;
void fr_add(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b);
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sqr(fr_t *r, const fr_t *a);
void fr_inv(fr_t *r, const fr_t *a);
bool fr_is_zero(const fr_t *a);
bool fr_eq(const fr_t *a, const fr_t *b);
bool fr_from_be_checked(fr_t *r, const uint8_t in[32]); /* false if >= r (utils.rs:262-275) */
void fr_from_be_reduce(fr_t *r, const uint8_t in[32]);  /* any 256-bit value, mod r (utils.rs:250-258) */
void fr_to_be(uint8_t out[32], const fr_t *a);
void fr_to_le_scalar(uint8_t out[32], const fr_t *a);   /* canonical, little-endian bytes */
void fr_from_u64(fr_t *r, uint64_t v);

/* Fp2 / Fp6 / Fp12 */
void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_neg(fp2_t *r, const fp2_t *a);
void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b);
void fp2_sqr(fp2_t *r, const fp2_t *a);
void fp2_inv(fp2_t *r, const fp2_t *a);
bool fp2_sqrt(fp2_t *r, const fp2_t *a);
bool fp2_is_zero(const fp2_t *a);
bool fp2_eq(const fp2_t *a, const fp2_t *b);
void fp12_mul(fp12_t *r, const fp12_t *a, const fp12_t *b);
void fp12_sqr(fp12_t *r, const fp12_t *a);
void fp12_inv(fp12_t *r, const fp12_t *a);
void fp12_conj(fp12_t *r, const fp12_t *a);
void fp12_frob(fp12_t *r, const fp12_t *a);
bool fp12_is_one(const fp12_t *a);
void fp12_set_one(fp12_t *r);

/* G1 */
extern g1_t G1_GENERATOR_J;
void g1_set_inf(g1_t *r);
bool g1_is_inf(const g1_t *a);
void g1_dbl(g1_t *r, const g1_t *a);
void g1_add(g1_t *r, const g1_t *a, const g1_t *b);          /* complete: add_or_double */
void g1_add_affine(g1_t *r, const g1_t *a, const g1a_t *b);  /* complete mixed */
void g1_neg(g1_t *r, const g1_t *a);
void g1_mul(g1_t *r, const g1_t *a, const uint8_t scalar_le[32], int nbits);
void g1_to_affine(g1a_t *r, const g1_t *a);
void g1_from_affine(g1_t *r, const g1a_t *a);
void g1_compress(uint8_t out[48], const g1_t *a);
int g1_uncompress(g1a_t *r, const uint8_t in[48]);           /* 0 ok, else error */
bool g1_in_subgroup(const g1_t *a);
bool g1_eq(const g1_t *a, const g1_t *b);

/* G2 */
extern g2_t G2_GENERATOR_J;
bool g2_is_inf(const g2_t *a);
void g2_dbl(g2_t *r, const g2_t *a);
void g2_add(g2_t *r, const g2_t *a, const g2_t *b);
void g2_neg(g2_t *r, const g2_t *a);
void g2_mul(g2_t *r, const g2_t *a, const uint8_t scalar_le[32], int nbits);
void g2_to_affine(g2a_t *r, const g2_t *a);
void g2_from_affine(g2_t *r, const g2a_t *a);
int g2_uncompress(g2a_t *r, const uint8_t in[96]);

/* pairing */
void miller_loop(fp12_t *f, const g2a_t *q, const g1a_t *p);
bool final_exp_is_one(const fp12_t *f);

/* SHA-256 (FIPS 180-4), the function blst_sha256 provides (kzg.rs:332, utils.rs:470) */
void sha256(uint8_t out[32], const uint8_t *msg, size_t len);

#endif
