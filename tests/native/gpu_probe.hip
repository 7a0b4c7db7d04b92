// GPU unit test of the field arithmetic under kzg_rust_amd/csrc (field.h, modinv.h): the same operation switch as the host build
// tests/native/hd_probe.cpp, compiled for gfx950 and run one element per lane, so that tests/test_gpu_field_ops.py can compare the
// DEVICE results with Python big integers directly (fr_batch_inv / fr_div / fr_pow / the Fp layer under them;
// reference src/utils.rs:35-140 delegates these to blst).  Test infrastructure only: never part of libkzg355.so.
#define KZG_MID_INLINE 1
#include <hip/hip_runtime.h>
#include "../../kzg_rust_amd/csrc/field.h"
#include "../../kzg_rust_amd/csrc/modinv.h"
#include "../../kzg_rust_amd/csrc/eval_core.h"
using namespace kzg;

// Fp: 48-byte big-endian operands (canonical, < p).  rc[i]: 0 ok, 1 operand rejected, 2 no square root
__global__ void __launch_bounds__(64) k_fp_op(int op, int n, const uint8_t *a, const uint8_t *b, uint8_t *out, int *rc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Fp x, y, r = fp_zero();
    if (!fp_from_be48(x, a + 48 * (size_t)i, false) || !fp_from_be48(y, b + 48 * (size_t)i, false)) { rc[i] = 1; return; }
    int code = 0;
    switch (op) {
        case 0: fp_add(r, x, y); break;
        case 1: fp_sub(r, x, y); break;
        case 2: fp_mul(r, x, y); break;
        case 3: fp_inv_fermat(r, x); break;
        case 4: if (!fp_sqrt(r, x)) code = 2; break;
        case 5: fp_neg(r, x); break;
        case 6: fp_dbl(r, x); break;
        case 7: code = fp_is_lex_largest(x) ? 101 : 100; break;            // the sign rule of the compressed encoding
        case 8: fp_inv(r, x); break;
        case 9: fp_sqr(r, x); break;
        case 10: {                                    // a chain of lazy products and sums ended by the canonicalisation the kernels use: (x y + x) y
            Fp t; fp_mul_lz(t, x, y); fp_add_lz(t, t, x); fp_mul_lz(t, t, y); fp_canon64(r, t); break;
        }
        default: code = 1;
    }
    rc[i] = code;
    fp_to_be48(out + 48 * (size_t)i, r);
}
// Fr: 32-byte big-endian operands, ANY 256-bit value (reduced as hash_to_bls_field does).
__global__ void __launch_bounds__(64) k_fr_op(int op, int n, const uint8_t *a, const uint8_t *b, uint8_t *out, int *rc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Fr x, y, r = fr_zero(); uint32_t w[8];
    be32_to_words(w, a + 32 * (size_t)i); fr_from_words(x, w);
    const bool canon = fr_words_canonical(w);
    be32_to_words(w, b + 32 * (size_t)i); fr_from_words(y, w);
    int code = 0;
    switch (op) {
        case 0: fr_add(r, x, y); break;
        case 1: fr_sub(r, x, y); break;
        case 2: fr_mul(r, x, y); break;
        case 3: fr_inv_fermat(r, x); break;
        case 4: code = canon ? 100 : 101; break;      // bytes_to_bls_field's range check (utils.rs:267-271)
        case 5: fr_inv(r, x); break;
        case 6: { Fr t; fr_inv(t, y); fr_mul(r, x, t); break; }            // fr_div (utils.rs:96-101): a / b, 0 for b = 0
        case 7: {                                     // fr_pow (utils.rs:113-131): x ^ (low 32 bits of b), square-and-multiply
            const uint32_t e = w[0];
            Fr acc = fr_one();
            for (int k = 31; k >= 0; k--) { fr_sqr(acc, acc); if ((e >> k) & 1) fr_mul(acc, acc, x); }
            r = acc; break;
        }
        case 8: {                                     // lazy chain as k_eval runs it: (x y + x y) y through mul2_lazy, then canonical
            Fr t, u; fr_mul2_lazy(t, x, y, y, x); fr_mul_lazy(u, t, y); fr_mul(r, u, fr_one()); break;      // 2 x y^2
        }
        default: code = 1;
    }
    rc[i] = code;
    fr_to_be32(out + 32 * (size_t)i, r);
}
template <typename K, typename... A> static int run(K kernel, dim3 grid, size_t in_bytes, size_t out_bytes, int n_rc, const uint8_t *a, const uint8_t *b, uint8_t *out, int *rc, A... head) {
    uint8_t *da = nullptr, *db = nullptr, *dout = nullptr; int *drc = nullptr;
    if (hipMalloc(&da, in_bytes) != hipSuccess || hipMalloc(&db, in_bytes) != hipSuccess || hipMalloc(&dout, out_bytes) != hipSuccess || hipMalloc(&drc, sizeof(int) * n_rc) != hipSuccess) return -1;
    (void)hipMemcpy(da, a, in_bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b ? b : a, in_bytes, hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0, out_bytes); (void)hipMemset(drc, 0xff, sizeof(int) * n_rc);
    hipLaunchKernelGGL(kernel, grid, dim3(64), 0, 0, head..., da, db, dout, drc);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost); (void)hipMemcpy(rc, drc, sizeof(int) * n_rc, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout); (void)hipFree(drc);
    return e == hipSuccess ? 0 : -2;
}
// fr_batch_inv (utils.rs:60-94) as one lane runs it: prefix products, one inversion, back substitution; zero inputs are an error there
// (utils.rs:70-72) and are reported as rc = 3 here.
__global__ void __launch_bounds__(64) k_fr_batch_inv_entry(int n, const uint8_t *a, const uint8_t *, uint8_t *out, int *rc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr acc = fr_one();
    for (int i = 0; i < n; i++) {
        uint32_t w[8]; be32_to_words(w, a + 32 * (size_t)i); Fr x; fr_from_words(x, w);
        if (fr_is_zero(x)) { rc[0] = 3; return; }
        fr_to_be32(out + 32 * (size_t)i, acc);
        fr_mul(acc, acc, x);
    }
    Fr inv; fr_inv(inv, acc);
    for (int i = n - 1; i >= 0; i--) {
        uint32_t w[8]; Fr pre, x;
        be32_to_words(w, out + 32 * (size_t)i); fr_from_words(pre, w);
        be32_to_words(w, a + 32 * (size_t)i); fr_from_words(x, w);
        Fr r; fr_mul(r, inv, pre); fr_mul(inv, inv, x);
        fr_to_be32(out + 32 * (size_t)i, r);
    }
    rc[0] = 0;
}

extern "C" {
int gpu_fp_ops(int op, int n, const uint8_t *a, const uint8_t *b, uint8_t *out, int *rc) {
    return run(k_fp_op, dim3((n + 63) / 64), 48 * (size_t)n, 48 * (size_t)n, n, a, b, out, rc, op, n);
}
int gpu_fr_ops(int op, int n, const uint8_t *a, const uint8_t *b, uint8_t *out, int *rc) {
    return run(k_fr_op, dim3((n + 63) / 64), 32 * (size_t)n, 32 * (size_t)n, n, a, b, out, rc, op, n);
}
int gpu_fr_batch_inv(int n, const uint8_t *a, uint8_t *out, int *rc) {
    return run(k_fr_batch_inv_entry, dim3(1), 32 * (size_t)n, 32 * (size_t)n, 1, a, nullptr, out, rc, n);
}
}
