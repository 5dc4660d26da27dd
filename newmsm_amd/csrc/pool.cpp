// pool.cpp -- device memory of the library's handles comes from here instead of straight from hipMalloc / hipFree.
//
// A registration creates and drops a handful of meshes, a cost function and their search structures per resolution level, gMSM a
// tree per subject and label; every one of those was a dozen hipMalloc calls (100-300 us each) on first use and as many hipFree
// calls (each a device-wide wait plus an unmap) when the handle went -- 3 ms per ico6 mesh that lives for one level, about as
// much as all the kernels that ever touch it.  Buffers given back are kept, by size class, and handed out again: the next
// level's meshes have the same sizes as this level's.
//
//   size classes   4 KB steps up to 1 MB, then eighths of the power of two below the size (at most 12.5 % over-allocation)
//   pool_free      waits for the device (as hipFree does) before the buffer can be handed to another stream's owner
//   limit          MSMHIP_POOL_MB of idle memory per process (default 4096); beyond it buffers really go back to the driver;
//                  a failed hipMalloc trims the pool and tries again.  MSMHIP_POOL=off: plain hipMalloc / hipFree.
//   pool_trim      everything idle goes back to the driver (called when the last context is destroyed)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "internal.hpp"

namespace msm {

namespace {

struct Pool {
    std::mutex mu;
    struct Live {
        size_t bytes;
        int device;
    };
    std::unordered_map<void *, Live> live;  // handed out
    // idle buffers by (device, size class)
    std::unordered_map<unsigned long long, std::vector<void *>> idle;
    size_t idle_bytes = 0;
    size_t limit = 0;
    bool enabled = true;
    Pool() {
        const char *off = std::getenv("MSMHIP_POOL");
        enabled = !(off && std::strcmp(off, "off") == 0);
        const char *mb = std::getenv("MSMHIP_POOL_MB");
        limit = (size_t)(mb ? std::max(0L, std::atol(mb)) : 4096L) << 20;
    }
};
Pool &pool() {
    static Pool *p = new Pool;  // never destroyed: handles may be released during static destruction
    return *p;
}

size_t size_class(size_t bytes) {
    if (bytes <= (1u << 20)) return (bytes + 4095) & ~(size_t)4095;
    size_t p2 = (size_t)1 << 20;
    while ((p2 << 1) <= bytes) p2 <<= 1;
    const size_t step = p2 >> 3;
    return (bytes + step - 1) / step * step;
}
unsigned long long key_of(int device, size_t cls) { return ((unsigned long long)device << 56) ^ (unsigned long long)cls; }

void trim_locked(Pool &p) {
    for (auto &kv : p.idle)
        for (void *q : kv.second) (void)hipFree(q);
    p.idle.clear();
    p.idle_bytes = 0;
}

}  // namespace

hipError_t pool_malloc(void **out, size_t bytes) {
    Pool &p = pool();
    if (!p.enabled) return hipMalloc(out, bytes);
    if (bytes == 0) bytes = 1;
    int device = 0;
    (void)hipGetDevice(&device);
    const size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.idle.find(key_of(device, cls));
        if (it != p.idle.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            p.idle_bytes -= cls;
            p.live[*out] = Pool::Live{cls, device};
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {  // out of memory with buffers lying idle: give them back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(p.mu);
            trim_locked(p);
        }
        e = hipMalloc(out, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(p.mu);
    p.live[*out] = Pool::Live{cls, device};
    return hipSuccess;
}

hipError_t pool_free(void *ptr) {
    if (!ptr) return hipSuccess;
    Pool &p = pool();
    if (!p.enabled) return hipFree(ptr);
    Pool::Live info{0, 0};
    {
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.live.find(ptr);
        if (it == p.live.end()) return hipFree(ptr);  // not ours (allocated before the pool was switched on, or by the caller)
        info = it->second;
        p.live.erase(it);
    }
    // what hipFree guarantees too: nothing queued anywhere still uses the buffer when its next owner gets it
    int device = 0;
    (void)hipGetDevice(&device);
    if (device != info.device) (void)hipSetDevice(info.device);
    (void)hipDeviceSynchronize();
    if (device != info.device) (void)hipSetDevice(device);
    std::lock_guard<std::mutex> lock(p.mu);
    if (p.idle_bytes + info.bytes > p.limit) return hipFree(ptr);
    p.idle[key_of(info.device, info.bytes)].push_back(ptr);
    p.idle_bytes += info.bytes;
    return hipSuccess;
}

void pool_trim() {
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    trim_locked(p);
}

size_t pool_idle_bytes() {
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    return p.idle_bytes;
}

}  // namespace msm
