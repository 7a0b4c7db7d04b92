// host_pool.h -- the host worker threads of a settings handle (engine.h).  Header-only and free of HIP so that tests/native/host_pool_test.cpp can
// run it under ThreadSanitizer on the build box.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// Host worker threads of a handle.  Two users: the Fiat-Shamir hashing of small host-buffer calls (host_sha256.h; the blobs of such
// a call are hashed here while its H2D copy and point kernels run) and the parallel copy into pinned staging buffers
// (KZG355_STAGING=ring: a single memcpy stream moves ~10 GB/s, a PCIe 5 x16 link ~55 GB/s).  A job is a range of indices dealt out
// through an atomic counter; the thread that began it takes indices too.  Several jobs run at once (round 4: one job object per call on
// the shared workers -- round 3 had one slot per handle, and the second of two simultaneous small calls fell back to the 3.7 ms
// device hash): the workers drain the oldest job that still has indices to hand out, every caller works on its own.
class HostPool {
public:
    struct Job {
        std::function<void(size_t)> fn;
        size_t count = 0;
        std::atomic<size_t> next{0}, left{0};
    };
    explicit HostPool(int workers) {
        for (int i = 0; i < workers; i++) th_.emplace_back([this] { run(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    std::shared_ptr<Job> begin(size_t count, std::function<void(size_t)> fn) {
        auto j = std::make_shared<Job>();
        j->fn = std::move(fn); j->count = count; j->left = count;
        if (count && !th_.empty()) {
            { std::lock_guard<std::mutex> lk(mu_); open_.push_back(j); }
            cv_.notify_all();
        }
        return j;
    }
    // the caller works through what is left of the job it began, then waits for the indices still in other hands
    void finish(const std::shared_ptr<Job> &j) {
        work(*j);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return j->left.load() == 0; });
    }
    void parallel_for(size_t count, std::function<void(size_t)> fn) { finish(begin(count, std::move(fn))); }
    void copy(void *dst, const void *src, size_t bytes) {
        if (th_.empty() || bytes < ((size_t)1 << 20)) { memcpy(dst, src, bytes); return; }
        const size_t parts = th_.size() + 1, per = ((bytes / parts) + 65535) & ~(size_t)65535;     // 64 KiB-granular slices
        parallel_for((bytes + per - 1) / per, [=](size_t k) {
            const size_t lo = k * per, n = bytes - lo < per ? bytes - lo : per;
            memcpy((uint8_t *)dst + lo, (const uint8_t *)src + lo, n);
        });
    }
private:
    void work(Job &j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1);
            if (i >= j.count) return;
            j.fn(i);
            if (j.left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(mu_); done_.notify_all(); }
        }
    }
    void run() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    while (!open_.empty() && open_.front()->next.load() >= open_.front()->count) open_.pop_front();   // every index handed out
                    if (stop_ || !open_.empty()) break;
                    cv_.wait(lk);
                }
                if (stop_) return;
                j = open_.front();
            }
            work(*j);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<std::shared_ptr<Job>> open_;      // jobs that may still have indices to hand out, oldest first
    bool stop_ = false;
};

