// devbuf.hpp -- minimal growable device buffer (HBM allocations are kept and reused across calls).
#pragma once

#include <vector>

#include <hip/hip_runtime.h>

#include <cstddef>

struct msm_ctx;

namespace msm {

// pool.cpp
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);
// stager.cpp: host <-> device copies through pinned blocks of the context (the GPU never touches the caller's or a local's pages); MSM_* status
int stage_h2d(msm_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t stream = nullptr);
int stage_d2h(msm_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int stage_alloc_failed(size_t bytes);  // sets the error text, returns MSM_ERR_HIP

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
    // host -> device on the context's stream, through its pinned staging blocks: `host` is consumed when the call returns (stager.cpp).
    // Returns an MSM_* status (0 = MSM_OK), like everything that can fail for more than one reason.
    int upload(const T *host, size_t n, msm_ctx *ctx) {
        if (ensure(n ? n : 1) != hipSuccess) return stage_alloc_failed(n * sizeof(T));
        return stage_h2d(ctx, p, host, n * sizeof(T));
    }
    // a whole vector; an empty one still leaves a valid (1-element) allocation behind so that kernels get a non-null pointer
    int upload_vec(const std::vector<T> &v, msm_ctx *ctx) { return upload(v.data(), v.size(), ctx); }
    // device -> host: the bytes are in `host` after the next ctx_sync / check_status of the context (stager.cpp)
    int download(T *host, size_t n, msm_ctx *ctx) const { return stage_d2h(ctx, host, p, n * sizeof(T)); }
    hipError_t zero(size_t n, hipStream_t s) {
        hipError_t e = ensure(n);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(p, 0, n * sizeof(T), s);
    }
};

}  // namespace msm
