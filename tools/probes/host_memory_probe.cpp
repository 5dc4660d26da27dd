// host_memory_probe.cpp -- what the HIP runtime of the GPU box does with host memory (round 5, the round-4 memory-access fault):
//   1. is the device address of hipHostRegister'ed / hipHostMalloc'ed memory the host address itself?
//   2. does hipMemcpyAsync from / to PAGEABLE memory return before the copy has read / written the host buffer?
//      (the source is overwritten right after the call returns; what arrives on the device tells)
// Build: hipcc -O2 --offload-arch=gfx950 tools/probes/host_memory_probe.cpp -o gpurun_out/host_memory_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                       \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    {
        const size_t n = 8u << 20;
        char *h = (char *)malloc(n + 4096);
        char *reg = (char *)(((uintptr_t)h + 4095) & ~(uintptr_t)4095);
        CK(hipHostRegister(reg, n, hipHostRegisterMapped | hipHostRegisterPortable));
        void *d = nullptr;
        CK(hipHostGetDevicePointer(&d, reg, 0));
        printf("registered malloc memory: host %p device %p same=%d\n", (void *)reg, d, (int)(d == (void *)reg));
        CK(hipHostUnregister(reg));
        free(h);
        void *hm = nullptr;
        CK(hipHostMalloc(&hm, n, hipHostMallocMapped));
        CK(hipHostGetDevicePointer(&d, hm, 0));
        printf("hipHostMalloc mapped:     host %p device %p same=%d\n", hm, d, (int)(d == hm));
        CK(hipHostFree(hm));
        void *dm = nullptr;
        CK(hipMalloc(&dm, n));
        printf("hipMalloc:                device %p\n", dm);
        CK(hipFree(dm));
    }
    const size_t sizes[] = {256, 4096, 16384, 65536, 262144, 1u << 20, 4u << 20, 16u << 20, 64u << 20};
    for (size_t n : sizes) {
        std::vector<unsigned char> src(n, 1), back(n, 0);
        unsigned char *d = nullptr;
        CK(hipMalloc((void **)&d, n));
        CK(hipMemsetAsync(d, 0, n, s));
        CK(hipStreamSynchronize(s));
        // H2D from pageable memory; overwrite the source as soon as the call returns
        double t0 = now_us();
        CK(hipMemcpyAsync(d, src.data(), n, hipMemcpyHostToDevice, s));
        double t1 = now_us();
        std::memset(src.data(), 2, n);
        CK(hipStreamSynchronize(s));
        double t2 = now_us();
        CK(hipMemcpy(back.data(), d, n, hipMemcpyDeviceToHost));
        size_t twos = 0;
        for (size_t i = 0; i < n; ++i) twos += back[i] == 2;
        // D2H into pageable memory: is the data there when the call returns?
        CK(hipMemsetAsync(d, 7, n, s));
        CK(hipStreamSynchronize(s));
        std::memset(back.data(), 0, n);
        double t3 = now_us();
        CK(hipMemcpyAsync(back.data(), d, n, hipMemcpyDeviceToHost, s));
        double t4 = now_us();
        size_t sevens_at_return = 0;
        for (size_t i = 0; i < n; ++i) sevens_at_return += back[i] == 7;
        CK(hipStreamSynchronize(s));
        double t5 = now_us();
        printf("%9zu B  H2D pageable: call %8.1f us, sync after %8.1f us, bytes that saw the LATER host write: %zu   |  D2H pageable: call %8.1f us, sync %8.1f us, arrived at return: %zu of %zu\n",
               n, t1 - t0, t2 - t1, twos, t4 - t3, t5 - t4, sevens_at_return, n);
        CK(hipFree(d));
    }
    // the same H2D behind a kernel-free but busy stream: a long memset queued first, so that an asynchronous copy would have to wait
    {
        const size_t big = 1u << 30, n = 4u << 20;
        unsigned char *dbig = nullptr, *d = nullptr;
        CK(hipMalloc((void **)&dbig, big));
        CK(hipMalloc((void **)&d, n));
        std::vector<unsigned char> src(n, 1), back(n, 0);
        for (int k = 0; k < 8; ++k) CK(hipMemsetAsync(dbig, k, big, s));
        double t0 = now_us();
        CK(hipMemcpyAsync(d, src.data(), n, hipMemcpyHostToDevice, s));
        double t1 = now_us();
        std::memset(src.data(), 2, n);
        CK(hipStreamSynchronize(s));
        double t2 = now_us();
        CK(hipMemcpy(back.data(), d, n, hipMemcpyDeviceToHost));
        size_t twos = 0;
        for (size_t i = 0; i < n; ++i) twos += back[i] == 2;
        printf("behind 8 GB of queued memsets: H2D pageable 4 MB call %.1f us, sync after %.1f us, bytes that saw the later host write: %zu\n", t1 - t0, t2 - t1, twos);
        CK(hipFree(d));
        CK(hipFree(dbig));
    }
    printf("done\n");
    return 0;
}
