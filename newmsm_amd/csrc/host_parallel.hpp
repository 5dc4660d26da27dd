// host_parallel.hpp -- worker threads for the host-side set-up (octree builds per label, ray table, weight lists).
// std::thread only: the library is loaded next to OpenMP runtimes of other libraries and must not bring its own.
// The threads are kept in a pool (starting sixteen threads costs about as much as building half an ico6 octree); a
// parallel_for that finds the pool taken -- the background build of a ray table and the caller's set-up can overlap --
// starts threads of its own instead of waiting.
#pragma once

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace msm {

// set inside a worker of parallel_for: nested set-up code then runs serially instead of multiplying the threads
inline thread_local bool t_inside_worker = false;

// host threads of one process: the cores it may run on (affinity aware), but no more than 16 by default -- one process
// drives one GPU, and an 8-GPU node runs eight of them side by side; MSMHIP_HOST_THREADS overrides (at most 64).
// 1 inside a worker.
inline int host_workers() {
    if (t_inside_worker) return 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    n = std::min(n, 16);
    if (const char *e = std::getenv("MSMHIP_HOST_THREADS")) n = std::atoi(e);
    return std::max(1, std::min(n, 64));
}

class WorkerPool {
public:
    static WorkerPool &instance() {
        // never destroyed: a background set-up job (api.cpp: RayJob) may still be running when the process exits, and its
        // parallel sections must not find a pool that static destruction has taken apart; the idle threads end with the process
        static WorkerPool *pool = new WorkerPool;
        return *pool;
    }
    // runs fn(0..n-1) on the caller and up to workers-1 pool threads; false when the pool is in use (caller falls back)
    bool run(int n, int workers, const std::function<void(int)> &fn) {
        std::unique_lock<std::mutex> owner(owner_, std::try_to_lock);
        if (!owner.owns_lock()) return false;
        grow(workers - 1);
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            helpers_ = std::min(workers - 1, (int)threads_.size());
            pending_ = helpers_;
            ++epoch_;
        }
        cv_work_.notify_all();
        const bool was_inside = t_inside_worker;
        t_inside_worker = true;
        for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
        t_inside_worker = was_inside;
        std::unique_lock<std::mutex> g(m_);
        cv_done_.wait(g, [&] { return pending_ == 0; });
        fn_ = nullptr;
        return true;
    }

private:
    WorkerPool() = default;
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_work_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void grow(int want) {
        want = std::min(want, 63);
        while ((int)threads_.size() < want) {
            const int id = (int)threads_.size();
            uint64_t seen;
            {
                std::lock_guard<std::mutex> g(m_);
                seen = epoch_;
            }
            threads_.emplace_back([this, id, seen]() mutable {
                t_inside_worker = true;
                for (;;) {
                    std::unique_lock<std::mutex> g(m_);
                    cv_work_.wait(g, [&] { return epoch_ != seen; });
                    seen = epoch_;
                    if (stop_) return;
                    if (id >= helpers_) continue;  // this job wants fewer threads
                    const std::function<void(int)> *fn = fn_;
                    const int n = n_;
                    g.unlock();
                    for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*fn)(i);
                    g.lock();
                    if (--pending_ == 0) cv_done_.notify_all();
                }
            });
        }
    }
    std::mutex owner_;  // one job at a time
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0, helpers_ = 0, pending_ = 0;
    std::atomic<int> next_{0};
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// runs fn(0..n-1) on up to `workers` threads; fn must not touch HIP or any msm handle
template <class F>
void parallel_for(int n, int workers, F fn) {
    workers = std::max(1, std::min(workers, n));
    if (workers == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    {
        const std::function<void(int)> call = [&fn](int i) { fn(i); };
        if (WorkerPool::instance().run(n, workers, call)) return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w)
        pool.emplace_back([&]() {
            t_inside_worker = true;
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        });
    for (auto &t : pool) t.join();
}

// the same over contiguous chunks [begin, end) of 0..n (one chunk per worker and a few more for balance)
template <class F>
void parallel_chunks(int n, int workers, F fn) {
    workers = std::max(1, std::min(workers, n));
    const int chunks = workers == 1 ? 1 : 4 * workers;
    parallel_for(chunks, workers, [&](int c) {
        const int b = (int)((long long)n * c / chunks), e = (int)((long long)n * (c + 1) / chunks);
        if (b < e) fn(c, b, e);
    });
}

}  // namespace msm
