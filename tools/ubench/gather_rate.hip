// Random 128-byte row gathers from a table far larger than the caches: the access pattern of the fixed-base MSM (k_msm_wide.hip: one
// table row per (window, scalar), each lane fetching a whole row) and of any scheme that would re-read or spill such rows (batch-affine
// accumulation keeps its pending operands in memory: 2.5 - 3.3x the row traffic).  Prints achieved GB/s for lane-per-row gathers at
// several grid sizes, next to a streaming read of the same buffer.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run: ./gather_rate [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_gather(const uint4 *table, size_t rows, int per_lane, uint4 *sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = tid * 0x9e3779b97f4a7c15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t r = (size_t)(h >> 11) % rows;
    uint4 nxt[7];
    for (int q = 0; q < 7; q++) nxt[q] = table[r * 8 + q];
    for (int k = 0; k < per_lane; k++) {
        uint4 cur[7];
        for (int q = 0; q < 7; q++) cur[q] = nxt[q];
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        r = (size_t)(h >> 11) % rows;
        for (int q = 0; q < 7; q++) nxt[q] = table[r * 8 + q];            // next row in flight while this one is "used"
        for (int q = 0; q < 7; q++) { acc.x ^= cur[q].x; acc.y += cur[q].y; acc.z ^= cur[q].z; acc.w += cur[q].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 7) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const uint4 *table, size_t n16, uint4 *sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = table[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    if (acc.x == 0x12345678u && acc.y == 7) sink[0] = acc;
}
int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 24;
    const size_t bytes = gib << 30, rows = bytes / 128;
    uint4 *table, *sink;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(table, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("table %zu GiB (%zu rows of 128 B)\n", gib, rows);
    for (int wgs : {512, 2048, 8192, 32768}) {
        const int per_lane = 256;
        hipLaunchKernelGGL(k_gather, dim3(wgs), dim3(256), 0, 0, table, rows, 8, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather, dim3(wgs), dim3(256), 0, 0, table, rows, per_lane, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double rows_read = (double)wgs * 256 * per_lane;
        printf("gather  %6d workgroups x 256 lanes x %d rows: %8.2f ms  %7.1f G rows/s  %7.1f GB/s (128 B per row: the lines moved)\n", wgs, per_lane, ms, rows_read / ms / 1e6,
               rows_read * 128 / ms / 1e6);
    }
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, table, bytes / 16, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, table, bytes / 16, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream  read of the same buffer: %8.2f ms  %7.1f GB/s\n", ms, (double)bytes / ms / 1e6);
    return 0;
}
