// fr_block.h -- device-only helpers shared by the evaluation (k_verify.hip) and quotient (k_prove.hip) kernels:
// blob element loads, wave shuffles of Fr values and the "product of all the others" scan.
#pragma once
#include "kernels.h"

namespace kzg {

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
constexpr uint32_t FR_MOD_TOP_WORD = 0x73eda753u;    // r = 0x73eda753 299d7d48 ...: a blob element whose top word is below it is canonical

KZG_HD void load_blob_element_words(uint32_t w[8], const uint8_t *blob, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 *p = reinterpret_cast<const uint4 *>(blob + 32 * (size_t)e);
    uint4 a = p[0], b = p[1];
    w[7] = bswap32(a.x); w[6] = bswap32(a.y); w[5] = bswap32(a.z); w[4] = bswap32(a.w);
    w[3] = bswap32(b.x); w[2] = bswap32(b.y); w[1] = bswap32(b.z); w[0] = bswap32(b.w);
#else
    be32_to_words(w, blob + 32 * (size_t)e);
#endif
}
__device__ __forceinline__ Fr fr_shfl_up(const Fr &v, int delta) {
    Fr r;
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = __shfl_up(v.l[i], delta, 64);
    return r;
}
__device__ __forceinline__ Fr fr_shfl_down(const Fr &v, int delta) {
    Fr r;
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = __shfl_down(v.l[i], delta, 64);
    return r;
}
__device__ __forceinline__ Fr fr_shfl(const Fr &v, int src) {
    Fr r;
#pragma unroll
    for (int i = 0; i < NFR; i++) r.l[i] = __shfl(v.l[i], src, 64);
    return r;
}
// product over the other lanes of the wave: exclusive prefix * exclusive suffix; also returns the wave total.
// LAZY = true uses non-canonical products (values < 1.1 r; see mont_mul_lazy) -- for chains that end in a canonical product.
template <bool LAZY = false>
__device__ __forceinline__ void wave_product_except_self(Fr &excl, Fr &total, const Fr &v, int lane) {
    const Fr one = fr_one();
    Fr pre = v, suf = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Fr a = fr_shfl_up(pre, off), b = fr_shfl_down(suf, off), t;
        if (LAZY) fr_mul_lazy(t, pre, a); else fr_mul(t, pre, a);
        fr_select(pre, lane >= off, pre, t);
        if (LAZY) fr_mul_lazy(t, suf, b); else fr_mul(t, suf, b);
        fr_select(suf, lane + off < 64, suf, t);
    }
    total = fr_shfl(pre, 63);
    Fr pe = fr_shfl_up(pre, 1), se = fr_shfl_down(suf, 1);
    fr_select(pe, lane == 0, pe, one);
    fr_select(se, lane == 63, se, one);
    if (LAZY) fr_mul_lazy(excl, pe, se); else fr_mul(excl, pe, se);
}


}  // namespace kzg
