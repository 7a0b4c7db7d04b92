// fr_rates.hip -- how much the Fr arithmetic of k_eval gains from occupancy (round 5).  Body: one level-1 node of the evaluation tree (eval_core.h: 5 lazy
// 9-limb Montgomery products + the carry-free sums) on register-resident operands, chained through its output so that nothing is hoisted; and, for comparison,
// one Fp product (14 limbs) per iteration.  Reported: wall ns per iteration per SIMD at 1 / 2 / 3 / 4 / 6 / 8 waves per SIMD (256-thread workgroups, the count per CU
// capped by an LDS request) -- the RATIOS between the columns are the point: k_eval runs at 2 waves per SIMD (182 VGPRs, 68 KB of LDS per workgroup).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I kzg_rust_amd/csrc -o tools/ubench/fr_rates tools/ubench/fr_rates.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "eval_core.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP> __global__ void __launch_bounds__(256) bench(unsigned *sink, int iters) {
    extern __shared__ unsigned pad[];
    const unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u;
    uint32_t pw[4][8];
    for (int e = 0; e < 4; e++) for (int k = 0; k < 8; k++) pw[e][k] = (a0 * (8 * e + k + 3) + a1) & (k == 7 ? 0x3fffffffu : 0xffffffffu);
    kzg::Fr T, imag, h;
    for (int i = 0; i < kzg::NFR; i++) { T.l[i] = (a0 * (i + 1)) & 0x1fffffffu; imag.l[i] = (a1 * (i + 7)) & 0x1fffffffu; }
    T.l[kzg::NFR - 1] &= 0xfffff; imag.l[kzg::NFR - 1] &= 0xfffff;
    kzg::Fp x, y;
    for (int i = 0; i < kzg::NFP; i++) { x.l[i] = (a0 * (i + 1)) & 0x1fffffffu; y.l[i] = (a1 * (i + 7)) & 0x1fffffffu; }
    for (int k = 0; k < iters; k++) {
        if (OP == 0) {
            kzg::eval_level1(h, pw, T, imag);
            for (int i = 0; i < kzg::NFR - 1; i++) T.l[i] = h.l[i] & 0x1fffffffu;      // the next node's T depends on this one's h: a chain, as in the kernel's Horner steps
            pw[0][0] ^= h.l[3];
        } else {
            kzg::fp_mul(x, x, y);
        }
    }
    unsigned r = 0;
    for (int i = 0; i < kzg::NFR; i++) r ^= h.l[i] ^ T.l[i];
    for (int i = 0; i < kzg::NFP; i++) r ^= x.l[i];
    if (r == 0x12345678u) sink[0] = r + pad[0];
}

template <int OP> int run(const char *name, int iters) {
    unsigned *d_sink;
    CHECK(hipMalloc(&d_sink, 4));
    printf("%-34s", name);
    double base = 0;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int lds = (160 * 1024) / wps - 2048;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<OP>, dim3(256 * wps), dim3(256), lds, 0, d_sink, 4);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<OP>, dim3(256 * wps), dim3(256), lds, 0, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms * 1e6 / ((double)iters * wps);           // wall ns per iteration per SIMD
        if (wps == 2) base = ns;
        printf(" | w%d %8.1f ns", wps, ns);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("\n");
    hipFree(d_sink);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d\ncolumns: wall ns per iteration per SIMD at N waves per SIMD\n", p.name, p.multiProcessorCount);
    run<0>("eval_level1 node (5 Fr products)", 4000);
    run<1>("fp_mul (one 14-limb product)", 20000);
    return 0;
}
