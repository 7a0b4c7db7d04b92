// Measures the issue cost (shader cycles per wave-instruction, per SIMD) of the VALU instructions this engine is built
// from, at 1 / 2 / 4 / 6 / 8 waves per SIMD on gfx950 -- the practical ceiling of an integer-ALU-bound kernel, next to the
// nominal 2 cycles per wave64 instruction of a SIMD-32 (MI355X_MICROARCH.md constants table).  Besides single
// instructions it times the two real bodies of the hot path with no memory traffic: one SHA-256 compression (the
// k_challenge_1w loop body) and one 29-bit-limb Montgomery Fp product (field.h), so their instruction counts convert to a
// "best possible" kernel time at every occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../kzg_rust_amd/csrc -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "field.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x

enum { OP_MAD64, OP_MAD64_DEP, OP_MUL_LO, OP_MUL_HI, OP_ADDC, OP_LSHL_ADD64, OP_FMA64, OP_MOV, OP_ADD3, OP_ALIGN_XOR, OP_FMA32, OP_ADD_U32, OP_XOR,
       OP_BITOP3, OP_ALIGNBIT, OP_PERM, OP_AND_OR, OP_LSHR64, OP_SHA_BLOCK, OP_FP_MUL, OP_COUNT };

__device__ __forceinline__ unsigned ror(unsigned x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ unsigned xor3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ unsigned ch3(unsigned e, unsigned f, unsigned g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }
__device__ __forceinline__ unsigned maj3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }
__constant__ unsigned SHA_K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
    0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
    0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

template <int OP> __global__ void __launch_bounds__(256) bench(unsigned long long *cycles, unsigned *sink, int iters) {
    extern __shared__ unsigned pad[];            // only to cap the workgroups per CU (even placement)
    unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3 + 7, a3 = a1 * 5 + 11;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a0 + 1, d5 = a1 + 1, d6 = a2 + 1, d7 = a3 + 1;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = 1.5, f5 = 2.5, f6 = 3.5, f7 = 4.5, fa = 1.0000001, fb = 0.999999;
    float g0 = a0, g1 = a1, g2 = a2, g3 = a3, ga = 1.0000001f, gb = 0.999999f;
    unsigned hh[8] = {a0, a1, a2, a3, a0 + 5, a1 + 5, a2 + 5, a3 + 5}, w[16];
    for (int i = 0; i < 16; i++) w[i] = a0 * (i + 3) + a1;
    kzg::Fp x, y;
    for (int i = 0; i < kzg::NFP; i++) { x.l[i] = (a0 * (i + 1)) & 0x1fffffffu; y.l[i] = (a1 * (i + 7)) & 0x1fffffffu; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        if (OP == OP_MAD64) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(a1) : "vcc");)
        } else if (OP == OP_MAD64_DEP) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              : "+v"(d0) : "v"(a0), "v"(a1) : "vcc");)
        } else if (OP == OP_MUL_LO) {
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a0 | 1));)
        } else if (OP == OP_MUL_HI) {
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xfffffff1u));)
        } else if (OP == OP_ADDC) {
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xfffffff1u) : "vcc");)
        } else if (OP == OP_LSHL_ADD64) {
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                              "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (OP == OP_LSHR64) {
            REP8(asm volatile("v_lshrrev_b64 %0, 29, %1\n v_lshrrev_b64 %1, 29, %2\n v_lshrrev_b64 %2, 29, %3\n v_lshrrev_b64 %3, 29, %0\n"
                              "v_lshrrev_b64 %0, 29, %1\n v_lshrrev_b64 %1, 29, %2\n v_lshrrev_b64 %2, 29, %3\n v_lshrrev_b64 %3, 29, %0\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (OP == OP_FMA64) {
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));)
        } else if (OP == OP_FMA32) {
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                              "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(ga), "v"(gb));)
        } else if (OP == OP_MOV) {
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_ADD_U32) {
            REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_XOR) {
            REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_ADD3) {
            REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n"
                              "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_BITOP3) {
            REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0xca\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0xe8\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96\n"
                              "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0xca\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0xe8\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_ALIGNBIT) {
            REP8(asm volatile("v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_alignbit_b32 %3, %0, %0, 7\n"
                              "v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_alignbit_b32 %3, %0, %0, 7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_ALIGN_XOR) {
            REP8(asm volatile("v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_xor_b32 %3, %0, %1\n"
                              "v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_xor_b32 %3, %0, %1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_PERM) {
            REP8(asm volatile("v_perm_b32 %0, %1, %2, %4\n v_perm_b32 %1, %2, %3, %4\n v_perm_b32 %2, %3, %0, %4\n v_perm_b32 %3, %0, %1, %4\n"
                              "v_perm_b32 %0, %1, %2, %4\n v_perm_b32 %1, %2, %3, %4\n v_perm_b32 %2, %3, %0, %4\n v_perm_b32 %3, %0, %1, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x00010203u));)
        } else if (OP == OP_AND_OR) {
            REP8(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1\n"
                              "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == OP_SHA_BLOCK) {          // one compression, schedule and rounds in one wave (k_challenge_1w's body)
            unsigned a = hh[0], bb = hh[1], c = hh[2], d = hh[3], e = hh[4], f = hh[5], g = hh[6], h = hh[7];
#pragma unroll
            for (int t = 0; t < 64; t++) {
                if (t >= 16) {
                    const unsigned w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                    w[t & 15] = w[t & 15] + xor3(ror(w15, 7), ror(w15, 18), w15 >> 3) + w[(t + 9) & 15] + xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
                }
                const unsigned t1 = h + xor3(ror(e, 6), ror(e, 11), ror(e, 25)) + ch3(e, f, g) + (w[t & 15] + SHA_K[t]);
                const unsigned t2 = xor3(ror(a, 2), ror(a, 13), ror(a, 22)) + maj3(a, bb, c);
                h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
            }
            hh[0] += a; hh[1] += bb; hh[2] += c; hh[3] += d; hh[4] += e; hh[5] += f; hh[6] += g; hh[7] += h;
        } else if (OP == OP_FP_MUL) {             // 8 dependent Montgomery products (29-bit limbs, field.h)
#pragma unroll 1
            for (int q = 0; q < 8; q++) { kzg::fp_mul(x, x, y); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (unsigned)(g0 + g1 + g2 + g3);
    for (int i = 0; i < 8; i++) r ^= hh[i];
    for (int i = 0; i < kzg::NFP; i++) r ^= x.l[i];
    if (r == 0x12345678u) sink[0] = r + pad[0];
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// per_iter: wave-instructions one loop iteration issues (for the two real bodies: counted from the disassembly, see the table
// printed by tools/ubench/count_insns.sh; 0 = report cycles per iteration instead)
template <int OP> int run(const char *name, double per_iter, int iters) {
    unsigned long long *d_cyc; unsigned *d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 8 * 4));
    CHECK(hipMalloc(&d_sink, 4));
    printf("%-26s", name);
    for (int wps : {1, 2, 4, 6, 8}) {      // waves per SIMD = 256-thread workgroups per CU; an LDS request caps the count per CU
        const int lds = (160 * 1024) / wps - 2048;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<OP>, dim3(256 * wps), dim3(256), lds, 0, d_cyc, d_sink, 4);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<OP>, dim3(256 * wps), dim3(256), lds, 0, d_cyc, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(256 * 4 * wps);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], worst = (double)h[h.size() - 1];
        const double per_wave = med / ((double)iters * (per_iter > 0 ? per_iter : 1));
        // wall: ns per wave-instruction per SIMD from the event time (includes the launch and the slowest CU)
        const double wall_ns = ms * 1e6 / ((double)iters * (per_iter > 0 ? per_iter : 1) * wps);
        printf(" | w%d %6.2f/%5.2f (%.2f ns, max/med %.2f)", wps, per_wave, per_wave / wps, wall_ns, worst / med);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("\n");
    hipFree(d_cyc); hipFree(d_sink);
    return 0;
}

int main(int argc, char **argv) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    printf("columns: wN = N waves per SIMD: s_memtime cycles per instruction per WAVE / per SIMD (wall ns per instruction per SIMD, slowest/median wave)\n");
    const double sha_insns = argc > 1 ? atof(argv[1]) : 0, fpmul_insns = argc > 2 ? atof(argv[2]) : 0;
    run<OP_MOV>("v_mov_b32", 64, 2000);
    run<OP_ADD_U32>("v_add_u32", 64, 2000);
    run<OP_XOR>("v_xor_b32", 64, 2000);
    run<OP_FMA32>("v_fma_f32", 64, 2000);
    run<OP_ADD3>("v_add3_u32", 64, 2000);
    run<OP_BITOP3>("v_bitop3_b32", 64, 2000);
    run<OP_ALIGNBIT>("v_alignbit_b32", 64, 2000);
    run<OP_ALIGN_XOR>("v_alignbit+xor", 64, 2000);
    run<OP_PERM>("v_perm_b32", 64, 2000);
    run<OP_AND_OR>("v_and_or_b32", 64, 2000);
    run<OP_ADDC>("v_add_co/addc chain", 64, 2000);
    run<OP_LSHL_ADD64>("v_lshl_add_u64", 64, 2000);
    run<OP_LSHR64>("v_lshrrev_b64", 64, 2000);
    run<OP_MAD64>("v_mad_u64_u32 (indep)", 64, 2000);
    run<OP_MAD64_DEP>("v_mad_u64_u32 (dep)", 64, 2000);
    run<OP_MUL_LO>("v_mul_lo_u32", 64, 2000);
    run<OP_MUL_HI>("v_mul_hi_u32", 64, 2000);
    run<OP_FMA64>("v_fma_f64", 64, 2000);
    run<OP_SHA_BLOCK>(sha_insns > 0 ? "SHA-256 block (per instr)" : "SHA-256 block (cycles)", sha_insns, 400);
    run<OP_FP_MUL>(fpmul_insns > 0 ? "8 x fp_mul (per instr)" : "8 x fp_mul (cycles)", fpmul_insns, 200);
    return 0;
}
