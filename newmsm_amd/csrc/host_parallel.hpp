// host_parallel.hpp -- worker threads for the host-side set-up (octree builds per label, ray table, weight lists).
// std::thread only: the library is loaded next to OpenMP runtimes of other libraries and must not bring its own.
#pragma once

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace msm {

// set inside a worker of parallel_for: nested set-up code then runs serially instead of multiplying the threads
inline thread_local bool t_inside_worker = false;

// host cores this process may use (affinity aware; MSMHIP_HOST_THREADS overrides), at most 64; 1 inside a worker
inline int host_workers() {
    if (t_inside_worker) return 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (const char *e = std::getenv("MSMHIP_HOST_THREADS")) n = std::atoi(e);
    return std::max(1, std::min(n, 64));
}

// runs fn(0..n-1) on up to `workers` threads; fn must not touch HIP or any msm handle
template <class F>
void parallel_for(int n, int workers, F fn) {
    workers = std::max(1, std::min(workers, n));
    if (workers == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
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
