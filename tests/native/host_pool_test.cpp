// ThreadSanitizer / stress test of kzg_rust_amd/csrc/host_pool.h (the handle's host worker threads: one job object per call, several jobs at once).
// Built and run by tests/test_host_pool.py with -fsanitize=thread.  Exit code 0 = every index of every job ran exactly once and nothing raced.
#include "../../kzg_rust_amd/csrc/host_pool.h"
#include <cstdio>

int main() {
    HostPool pool(7);
    const int CALLERS = 6, ROUNDS = 200;
    std::atomic<long> total{0};
    std::atomic<int> bad{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < CALLERS; c++) {
        callers.emplace_back([&, c] {
            for (int r = 0; r < ROUNDS; r++) {
                const size_t n = (size_t)((c * 7 + r * 3) % 40);          // incl. empty jobs
                std::vector<std::atomic<int>> hit(n ? n : 1);
                for (auto &h : hit) h = 0;
                auto job = pool.begin(n, [&](size_t i) { hit[i]++; total++; });
                // the caller does something else meanwhile (the library queues copies and kernels here), then joins
                volatile int spin = 0; for (int k = 0; k < 200; k++) spin += k;
                pool.finish(job);
                for (size_t i = 0; i < n; i++) if (hit[i] != 1) bad++;
            }
        });
    }
    for (auto &t : callers) t.join();
    // parallel_for and the slice copy on top of it
    std::vector<uint8_t> src((size_t)3 << 20), dst(src.size());
    for (size_t i = 0; i < src.size(); i++) src[i] = (uint8_t)(i * 131 + 7);
    pool.copy(dst.data(), src.data(), src.size());
    if (memcmp(src.data(), dst.data(), src.size()) != 0) bad++;
    long want = 0;
    for (int c = 0; c < CALLERS; c++) for (int r = 0; r < ROUNDS; r++) want += (c * 7 + r * 3) % 40;
    if (total != want) bad++;
    HostPool none(0);                                                      // no workers: the caller does all of it
    std::atomic<int> solo{0};
    none.parallel_for(17, [&](size_t) { solo++; });
    if (solo != 17) bad++;
    printf("host_pool_test: %ld indices over %d jobs, %d problems\n", (long)total, CALLERS * ROUNDS, (int)bad);
    return bad ? 1 : 0;
}
