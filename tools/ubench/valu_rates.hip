// Measures issue cost (shader cycles per wave-instruction) of the integer/FP64 instructions a
// big-integer Montgomery multiplier can be built from, at 1/2/4 waves per SIMD on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP> __global__ void __launch_bounds__(1024) bench(unsigned long long *cycles, unsigned *sink, int iters) {
    unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3 + 7, a3 = a1 * 5 + 11;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a0 + 1, d5 = a1 + 1, d6 = a2 + 1, d7 = a3 + 1;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = 1.5, f5 = 2.5, f6 = 3.5, f7 = 4.5, fa = 1.0000001, fb = 0.999999;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent accumulators
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(a1) : "vcc");)
        } else if (OP == 1) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a0 | 1));)
        } else if (OP == 2) {  // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xfffffff1u));)
        } else if (OP == 3) {  // v_mad_u32_u24
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                              "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x00fffff1u));)
        } else if (OP == 4) {  // v_add_co_u32 / v_addc_co_u32 chain
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xfffffff1u) : "vcc");)
        } else if (OP == 5) {  // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                              "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (OP == 6) {  // v_fma_f64, 8 independent
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));)
        } else if (OP == 7) {  // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == 8) {  // v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n"
                              "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == 9) {  // v_mad_u64_u32 dependent chain (latency)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                              : "+v"(d0) : "v"(a0), "v"(a1) : "vcc");)
        } else if (OP == 10) {  // v_alignbit_b32 (rotate) + v_xor3 mix: SHA-256 style
            REP8(asm volatile("v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_xor_b32 %3, %0, %1\n"
                              "v_alignbit_b32 %0, %1, %1, 6\n v_alignbit_b32 %1, %2, %2, 11\n v_alignbit_b32 %2, %3, %3, 25\n v_xor_b32 %3, %0, %1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == 11) {  // v_add_f64
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (r == 0x12345678u) sink[0] = r;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP> int run(const char *name, int per_iter) {
    const int iters = 2000;
    unsigned long long *d_cyc; unsigned *d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 16));
    CHECK(hipMalloc(&d_sink, 4));
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {      // waves per SIMD: block = 256*wps threads, one block per CU
        int threads = 256 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        bench<OP><<<256, threads>>>(d_cyc, d_sink, 10);
        CHECK(hipEventRecord(e0));
        bench<OP><<<256, threads>>>(d_cyc, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(256 * 4 * wps);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double med = (double)h[h.size() / 2];
        double per_wave = med / ((double)iters * per_iter);          // cycles a wave spends per instruction
        printf("  wps=%d: %6.2f cyc/instr/wave -> %5.2f cyc/instr/SIMD (%.3f ms)", wps, per_wave, per_wave / wps, ms);
    }
    printf("\n");
    hipFree(d_cyc); hipFree(d_sink);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    run<0>("v_mad_u64_u32 (indep)", 64);
    run<9>("v_mad_u64_u32 (dep chain)", 64);
    run<1>("v_mul_lo_u32", 64);
    run<2>("v_mul_hi_u32", 64);
    run<3>("v_mad_u32_u24", 64);
    run<4>("v_add_co/addc chain", 64);
    run<5>("v_lshl_add_u64", 64);
    run<6>("v_fma_f64 (indep)", 64);
    run<11>("v_add_f64 (indep)", 64);
    run<7>("v_mov_b32", 64);
    run<8>("v_add3_u32", 64);
    run<10>("v_alignbit+xor", 64);
    return 0;
}
