// fr_rates.hip -- how much the Fr arithmetic of k_eval gains from occupancy (round 5).  Body: one level-1 group of the evaluation tree (eval_core.h: three nodes
// = three two-product Montgomery reductions of 9 limbs + the carry-free sums; until late round 5 one radix-4 node of 5 lazy products) on register-resident
// operands, chained through its output so that nothing is hoisted; one lone two-product reduction; and, for comparison, one Fp product (14 limbs) per iteration.  Reported: wall ns per iteration per SIMD at 1 / 2 / 3 / 4 / 6 / 8 waves per SIMD (256-thread workgroups, the count per CU
// capped by an LDS request) -- the RATIOS between the columns are the point: k_eval runs at 2 waves per SIMD (182 VGPRs, 68 KB of LDS per workgroup).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I kzg_rust_amd/csrc -o tools/ubench/fr_rates tools/ubench/fr_rates.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "eval_core.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef FR_RATES_WAVES
#define FR_RATES_WAVES 1      // -DFR_RATES_WAVES=3 / 4: the register budget of three / four waves per SIMD (168 / 128), so that those columns are real
#endif
template <int OP> __global__ void __launch_bounds__(256, FR_RATES_WAVES) bench(unsigned *sink, int iters) {
    extern __shared__ unsigned pad[];
    const unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u;
    uint32_t pw[4][8];
    for (int e = 0; e < 4; e++) for (int k = 0; k < 8; k++) pw[e][k] = (a0 * (8 * e + k + 3) + a1) & (k == 7 ? 0x3fffffffu : 0xffffffffu);
    kzg::Fr T, imag, h, Sp = kzg::fr_zero();
    for (int i = 0; i < kzg::NFR; i++) { T.l[i] = (a0 * (i + 1)) & 0x1fffffffu; imag.l[i] = (a1 * (i + 7)) & 0x1fffffffu; }
    T.l[kzg::NFR - 1] &= 0xfffff; imag.l[kzg::NFR - 1] &= 0xfffff;
    kzg::EvalGroup grp; grp.sa = T; grp.sb = imag; grp.st = T; grp.st.l[0] ^= 5u;
    h = T;
    kzg::EvalGroup grpB = grp; grpB.sb.l[1] ^= 9u;
    kzg::Fr SpB = kzg::fr_zero();
    uint32_t pwB[4][8];
    for (int e = 0; e < 4; e++) for (int k = 0; k < 8; k++) pwB[e][k] = pw[e][k] ^ (0x55u << k);
    kzg::Fp x, y;
    for (int i = 0; i < kzg::NFP; i++) { x.l[i] = (a0 * (i + 1)) & 0x1fffffffu; y.l[i] = (a1 * (i + 7)) & 0x1fffffffu; }
    for (int k = 0; k < iters; k++) {
        if (OP == 0) {
            kzg::eval_group_leaves(h, Sp, pw, T, imag, grp);
            for (int i = 0; i < kzg::NFR - 1; i++) grp.sa.l[i] = h.l[i] & 0x1fffffffu;      // the next group's root depends on this one's result: nothing is hoisted
            pw[0][0] ^= h.l[3]; Sp.l[kzg::NFR - 1] &= 0xffff;
        } else if (OP == 3) {                                    // two independent groups per iteration: does the compiler's interleaving buy what a third wave does?
            kzg::Fr hB;
            kzg::eval_group_leaves(h, Sp, pw, T, imag, grp);
            kzg::eval_group_leaves(hB, SpB, pwB, T, imag, grpB);
            for (int i = 0; i < kzg::NFR - 1; i++) { grp.sa.l[i] = h.l[i] & 0x1fffffffu; grpB.sa.l[i] = hB.l[i] & 0x1fffffffu; }
            pw[0][0] ^= h.l[3]; pwB[0][0] ^= hB.l[3]; Sp.l[kzg::NFR - 1] &= 0xffff; SpB.l[kzg::NFR - 1] &= 0xffff;
        } else if (OP == 2) {
            kzg::Fr t; kzg::fr_mul2_lazy(t, h, T, grp.sa, imag);
            for (int i = 0; i < kzg::NFR - 1; i++) h.l[i] = t.l[i] & 0x1fffffffu;
            h.l[kzg::NFR - 1] = t.l[kzg::NFR - 1] & 0xfffff;
        } else {
            kzg::fp_mul(x, x, y);
        }
    }
    unsigned r = 0;
    for (int i = 0; i < kzg::NFR; i++) r ^= h.l[i] ^ T.l[i] ^ Sp.l[i] ^ grp.sa.l[i] ^ SpB.l[i] ^ grpB.sa.l[i];
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
    run<0>("level-1 group (3 two-product nodes)", 4000);
    run<3>("two level-1 groups, interleaved", 2000);
    run<2>("one two-product reduction (Fr)", 20000);
    run<1>("fp_mul (one 14-limb product)", 20000);
    return 0;
}
