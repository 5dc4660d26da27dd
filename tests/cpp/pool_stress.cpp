// pool_stress.cpp -- six threads build the same octree eight times each through the library (the worker pool serves one of them
// at a time, the others start their own threads): the signatures must agree.  Built with -fsanitize=thread by
// tools/tsan_pool_stress.sh (no GPU needed).
#include <cstdio>
#include <thread>
#include <vector>
#include "msmhip.h"
int main() {
    int32_t V, T;
    msm_icosphere_counts(5, &V, &T);
    std::vector<double> xyz(3 * (size_t)V);
    std::vector<int32_t> tri(3 * (size_t)T);
    msm_icosphere(5, 100.0, xyz.data(), tri.data());
    std::vector<uint64_t> sig(6, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < 6; ++k)
        th.emplace_back([&, k]() {
            for (int r = 0; r < 8; ++r) {
                int64_t stats[5];
                uint64_t s = 0;
                msm_octree_signature(xyz.data(), tri.data(), V, T, stats, &s);
                if (r && s != sig[k]) std::printf("signature changed!\n");
                sig[k] = s;
            }
        });
    for (auto &t : th) t.join();
    for (int k = 1; k < 6; ++k)
        if (sig[k] != sig[0]) std::printf("threads disagree!\n");
    std::printf("done %llx\n", (unsigned long long)sig[0]);
    return 0;
}
