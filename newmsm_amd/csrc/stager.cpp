// stager.cpp -- every byte the GPU reads from or writes to HOST memory goes through pinned blocks this library allocated itself.
//
// Why (round 5; DESIGN.md section 3, "Host memory the GPU touches"): on the MI355X boxes the device address of page-locked host memory IS
// the host address (tools/probes/host_memory_probe.cpp: hipHostRegister'ed malloc memory, hipHostMalloc memory), page-locking works on whole
// pages, and the HIP runtime page-locks a PAGEABLE buffer behind the caller's back for the length of an asynchronous copy (device -> host
// into pageable memory returns before the data has arrived).  Two heap buffers that share a page -- a registered numpy array and its
// neighbour in the heap, two std::vectors filled by two set-up threads -- then share a GPU mapping, and the first to be released takes the
// page away from under the other's copy: a GPU memory-access fault at a page-aligned HEAP address (gpurun_out/r4_bench_final.err).
// The rule that removes the class: the copy engines and kernels only ever see hipHostMalloc blocks (whole pages, owned by a context,
// released when the device is idle) -- the staging blocks below, ctx_io_pinned, msm_host_alloc -- or page-aligned whole-page ranges the
// caller registered (msm_host_register refuses anything else).
//
//   stage_h2d   memcpy into a staging block, then an asynchronous copy from there: the caller's buffer is consumed when the call returns
//               (it may be a local that dies, or be written again), nothing waits for the stream
//   stage_d2h   an asynchronous copy into a staging block; the bytes reach the caller's buffer at the next ctx_sync / check_status of
//               the context (the buffer must live until then: every entry point synchronises before it returns)
//   blocks      never move and never shrink; a block is used again once the event recorded behind its last copy has completed and its
//               deliveries have been made; all go with the context (after the device is idle)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <vector>

#include "internal.hpp"

namespace msm {

struct Stager {
    struct Block {
        char *p = nullptr;
        size_t cap = 0, used = 0;
        hipEvent_t ev = nullptr;
        hipStream_t stream = nullptr;  // of the copies queued from / into the block since it was opened
        int state = 0;                 // 0 free, 1 open, 2 retired (ev recorded behind its last copy)
        int undelivered = 0;           // device -> host copies whose bytes have not been handed to the caller yet
    };
    struct Delivery {
        int block;
        const char *pinned;
        void *dst;
        size_t bytes;
    };
    std::mutex mu;
    std::vector<Block> blocks;
    std::vector<Delivery> deliveries;
    int open = -1;
    size_t total = 0;
    uint64_t allocated_blocks = 0, waits = 0;
    size_t min_block = 0;
};

namespace {

// contexts with a stager, and per thread the context it last queued a device -> host copy on (stage_on_failure)
std::mutex g_live_mu;
std::set<msm_ctx *> g_live;
thread_local msm_ctx *t_fetch_ctx = nullptr;

// smallest block: MSMHIP_STAGE_MIN_KB (default 4096), read when a context is created -- the tests make it small so that the blocks have to grow and rotate
// under the copies of a set-up
constexpr size_t kMinBlockDefault = (size_t)4 << 20;
constexpr size_t kSoftLimit = (size_t)256 << 20;  // beyond this much staging memory a busy block is waited for rather than another one allocated

Stager &stager_of(msm_ctx *ctx) { return *ctx->stager; }  // created with the context (stager_create)

// a region of `bytes` (256-byte aligned) in a block whose copies all run on `stream`; the lock is held by the caller
int acquire(msm_ctx *ctx, Stager &s, size_t bytes, hipStream_t stream, int *block, char **ptr) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (s.open >= 0) {
        Stager::Block &b = s.blocks[(size_t)s.open];
        if (b.stream == stream && b.used + need <= b.cap) {
            *block = s.open;
            *ptr = b.p + b.used;
            b.used += need;
            return MSM_OK;
        }
        // full, or another stream's turn: an event behind its last copy tells when it may be used again
        if (b.used > 0) {
            MSM_HIP(hipEventRecord(b.ev, b.stream));
            b.state = 2;
        } else {
            b.state = 0;
        }
        s.open = -1;
    }
    auto reusable = [&](Stager::Block &b) {
        if (b.state == 0) return true;
        if (b.state == 2 && b.undelivered == 0 && hipEventQuery(b.ev) == hipSuccess) {
            b.state = 0;
            b.used = 0;
            return true;
        }
        return false;
    };
    int pick = -1;
    for (size_t i = 0; i < s.blocks.size(); ++i)
        if (s.blocks[i].cap >= need && reusable(s.blocks[i]) && (pick < 0 || s.blocks[i].cap < s.blocks[(size_t)pick].cap)) pick = (int)i;
    if (pick < 0 && s.total >= kSoftLimit) {  // plenty of staging memory already: wait for a block that is large enough instead of growing further
        for (size_t i = 0; i < s.blocks.size() && pick < 0; ++i) {
            Stager::Block &b = s.blocks[i];
            if (b.cap >= need && b.state == 2 && b.undelivered == 0) {
                MSM_HIP(hipEventSynchronize(b.ev));
                b.state = 0;
                b.used = 0;
                pick = (int)i;
                ++s.waits;
            }
        }
    }
    if (pick < 0) {
        Stager::Block b;
        b.cap = std::max(s.min_block, (need + need / 4 + 4095) & ~(size_t)4095);
        if (hipHostMalloc((void **)&b.p, b.cap) != hipSuccess) {
            (void)hipGetLastError();
            return fail(MSM_ERR_HIP, "pinned staging block of %zu bytes could not be allocated", b.cap);
        }
        if (hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(b.p);
            return fail(MSM_ERR_HIP, "hipEventCreate failed");
        }
        s.blocks.push_back(b);
        s.total += b.cap;
        ++s.allocated_blocks;
        pick = (int)s.blocks.size() - 1;
    }
    Stager::Block &b = s.blocks[(size_t)pick];
    b.state = 1;
    b.used = need;
    b.stream = stream;
    s.open = pick;
    *block = pick;
    *ptr = b.p;
    return MSM_OK;
}

}  // namespace

void stager_create(msm_ctx *ctx) {
    if (!ctx->stager) {
        ctx->stager = new Stager();
        const char *e = std::getenv("MSMHIP_STAGE_MIN_KB");
        const long kb = e ? std::atol(e) : 0;
        ctx->stager->min_block = kb > 0 ? (size_t)kb << 10 : kMinBlockDefault;
    }
    std::lock_guard<std::mutex> live(g_live_mu);
    g_live.insert(ctx);
}

int stage_alloc_failed(size_t bytes) { return fail(MSM_ERR_HIP, "device allocation of %zu bytes failed", bytes); }

int stage_h2d(msm_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return MSM_OK;
    if (!stream) stream = ctx->stream;
    Stager &s = stager_of(ctx);
    std::lock_guard<std::mutex> lock(s.mu);
    int block = -1;
    char *pin = nullptr;
    const int st = acquire(ctx, s, bytes, stream, &block, &pin);
    if (st) return st;
    std::memcpy(pin, src, bytes);
    MSM_HIP(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, stream));
    return MSM_OK;
}

int stage_d2h(msm_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return MSM_OK;
    Stager &s = stager_of(ctx);
    std::lock_guard<std::mutex> lock(s.mu);
    int block = -1;
    char *pin = nullptr;
    const int st = acquire(ctx, s, bytes, ctx->stream, &block, &pin);
    if (st) return st;
    MSM_HIP(hipMemcpyAsync(pin, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    s.blocks[(size_t)block].undelivered++;
    s.deliveries.push_back(Stager::Delivery{block, pin, host_dst, bytes});
    t_fetch_ctx = ctx;
    return MSM_OK;
}

// after the context's stream has been synchronised: what stage_d2h fetched goes to where the callers asked for it
void stage_deliver(msm_ctx *ctx) {
    Stager *sp = ctx->stager;
    if (!sp) return;
    std::lock_guard<std::mutex> lock(sp->mu);
    for (const Stager::Delivery &d : sp->deliveries) {
        std::memcpy(d.dst, d.pinned, d.bytes);
        sp->blocks[(size_t)d.block].undelivered--;
    }
    sp->deliveries.clear();
}

// deliveries nobody will take
void stage_forget(msm_ctx *ctx) {
    Stager *sp = ctx->stager;
    if (!sp) return;
    std::lock_guard<std::mutex> lock(sp->mu);
    for (const Stager::Delivery &d : sp->deliveries) sp->blocks[(size_t)d.block].undelivered--;
    sp->deliveries.clear();
}

// fail() on this thread (internal.hpp): the context this thread last fetched through, if it is still alive, forgets what it has not delivered
void stage_on_failure() {
    msm_ctx *ctx = t_fetch_ctx;
    if (!ctx) return;
    t_fetch_ctx = nullptr;
    std::lock_guard<std::mutex> live(g_live_mu);
    if (g_live.count(ctx)) stage_forget(ctx);
}

int ctx_sync(msm_ctx *ctx) {
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        stage_forget(ctx);
        return fail(MSM_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    stage_deliver(ctx);
    return MSM_OK;
}

void stager_stats(msm_ctx *ctx, int64_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    Stager *sp = ctx->stager;
    if (!sp) return;
    std::lock_guard<std::mutex> lock(sp->mu);
    out[0] = (int64_t)sp->blocks.size();
    out[1] = (int64_t)sp->total;
    out[2] = (int64_t)sp->allocated_blocks;
    out[3] = (int64_t)sp->waits;
}

// with the context (the caller has made the device idle: no copy engine reads or writes a block any more)
void stager_destroy(msm_ctx *ctx) {
    {
        std::lock_guard<std::mutex> live(g_live_mu);
        g_live.erase(ctx);
    }
    Stager *sp = ctx->stager;
    if (!sp) return;
    for (Stager::Block &b : sp->blocks) {
        if (b.ev) (void)hipEventDestroy(b.ev);
        if (b.p) (void)hipHostFree(b.p);
    }
    delete sp;
    ctx->stager = nullptr;
}

}  // namespace msm
