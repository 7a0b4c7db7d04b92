// Experiment (rejected): can the single-blob SHA-256 chain -- 3.7 ms of single-call latency -- run faster on the SCALAR unit when
// everything is wave-uniform (one wave per message)?  Measured on MI355X, 64 messages of 131 KB: 15.1 ms against 6.5 ms for the plain
// one-lane-per-message vector form (the compiler mixes ~1650 s_ and ~1350 v_ instructions per block -- there is no scalar rotate --
// and a dependent scalar instruction issues no faster than a vector one).  Build: hipcc --offload-arch=gfx950 -O3 sha_scalar.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__constant__ uint32_t K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
__device__ __forceinline__ uint32_t ror(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
// one workgroup (one wave) per message: everything is wave-uniform -> the compiler may keep it on the scalar unit
__global__ void __launch_bounds__(64) k_sha_uniform(const uint32_t *__restrict__ msg, int nblocks, uint32_t *out) {
    const uint32_t *m = msg + (size_t)blockIdx.x * nblocks * 16;
    uint32_t h0 = 0x6a09e667u, h1 = 0xbb67ae85u, h2 = 0x3c6ef372u, h3 = 0xa54ff53au, h4 = 0x510e527fu, h5 = 0x9b05688cu, h6 = 0x1f83d9abu, h7 = 0x5be0cd19u;
#pragma unroll 1
    for (int b = 0; b < nblocks; b++) {
        uint32_t w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) { const uint32_t v = m[16 * b + t]; w[t] = __builtin_bswap32(v); }
        uint32_t a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, g = h6, h = h7;
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16) {
                const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                const uint32_t s0 = ror(w15, 7) ^ ror(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = ror(w2, 17) ^ ror(w2, 19) ^ (w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
            }
            const uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + w[t & 15] + K[t];
            const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += h;
    }
    if (threadIdx.x == 0) { uint32_t *o = out + 8 * blockIdx.x; o[0] = h0; o[1] = h1; o[2] = h2; o[3] = h3; o[4] = h4; o[5] = h5; o[6] = h6; o[7] = h7; }
}
// reference: one lane per message (the vector form), 64 messages per wave
__global__ void __launch_bounds__(64) k_sha_lanes(const uint32_t *__restrict__ msg, int nblocks, uint32_t *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const uint32_t *m = msg + (size_t)i * nblocks * 16;
    uint32_t h0 = 0x6a09e667u, h1 = 0xbb67ae85u, h2 = 0x3c6ef372u, h3 = 0xa54ff53au, h4 = 0x510e527fu, h5 = 0x9b05688cu, h6 = 0x1f83d9abu, h7 = 0x5be0cd19u;
#pragma unroll 1
    for (int b = 0; b < nblocks; b++) {
        uint32_t w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) { const uint32_t v = m[16 * b + t]; w[t] = __builtin_bswap32(v); }
        uint32_t a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, g = h6, h = h7;
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16) {
                const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                const uint32_t s0 = ror(w15, 7) ^ ror(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = ror(w2, 17) ^ ror(w2, 19) ^ (w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
            }
            const uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + w[t & 15] + K[t];
            const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += h;
    }
    uint32_t *o = out + 8 * i; o[0] = h0; o[1] = h1; o[2] = h2; o[3] = h3; o[4] = h4; o[5] = h5; o[6] = h6; o[7] = h7;
}
int main() {
    const int nmsg = 64, nblocks = 2050;
    std::vector<uint32_t> h((size_t)nmsg * nblocks * 16);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    uint32_t *d, *o1, *o2;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o1, nmsg * 32); hipMalloc(&o2, nmsg * 32);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_sha_uniform, dim3(nmsg), dim3(64), 0, 0, d, nblocks, o1);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_sha_lanes, dim3(1), dim3(64), 0, 0, d, nblocks, o2);
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("uniform (one wave per message): %.3f ms   lanes (one lane per message): %.3f ms\n",
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    std::vector<uint32_t> a(nmsg * 8), b(nmsg * 8);
    hipMemcpy(a.data(), o1, nmsg * 32, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, nmsg * 32, hipMemcpyDeviceToHost);
    printf("digests %s\n", a == b ? "agree" : "DIFFER");
    return 0;
}
