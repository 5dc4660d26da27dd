// devbuf.hpp -- minimal growable device buffer (HBM allocations are kept and reused across calls).
#pragma once

#include <vector>

#include <hip/hip_runtime.h>

#include <cstddef>

namespace msm {

// pool.cpp
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool owned = true;  // false: a window into another allocation (view())

    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }

    void release() {
        if (p && owned) (void)msm::pool_free(p);  // waits, like hipFree, for work that may still use the buffer
        p = nullptr;
        cap = 0;
        owned = true;
    }
    // n elements at ptr, which belongs to another buffer that outlives this one
    void view(T *ptr, size_t n) {
        release();
        p = ptr;
        cap = n;
        owned = false;
    }
    hipError_t ensure(size_t n) {
        if (n <= cap && p) return hipSuccess;
        release();
        cap = n + n / 4 + 16;
        hipError_t e = msm::pool_malloc((void **)&p, cap * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            cap = 0;
        }
        return e;
    }
    hipError_t upload(const T *host, size_t n, hipStream_t s) {
        hipError_t e = ensure(n);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s);
    }
    // a whole vector; an empty one still leaves a valid (1-element) allocation behind so that kernels get a non-null pointer
    hipError_t upload_vec(const std::vector<T> &v, hipStream_t s) {
        hipError_t e = ensure(v.empty() ? 1 : v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    hipError_t download(T *host, size_t n, hipStream_t s) const { return hipMemcpyAsync(host, p, n * sizeof(T), hipMemcpyDeviceToHost, s); }
    hipError_t zero(size_t n, hipStream_t s) {
        hipError_t e = ensure(n);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(p, 0, n * sizeof(T), s);
    }
};

}  // namespace msm
