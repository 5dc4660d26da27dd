// resample_kernels.hip -- Resampler::get_adaptive_barycentric_weights (R/resampler.cpp:72-140) and barycentric_data_interpolation
// (:30-70) with everything after the nearest-triangle queries on the GPU as well.
//
// The reference builds, per new vertex, a std::map of forward weights (its triangle in the old mesh) and the transposed reverse
// weights (old vertices whose triangle in the new mesh has it as a corner), keeps the longer list, multiplies by the new vertex's
// area, scatter-adds into correction[old vertex] IN THE SERIAL ORDER OF THE NEW VERTICES, rescales by oldArea / correction and
// normalises each row.  Round 1 did that list surgery on the host in the same serial order (2 ms at ico6, nineteen times per gMSM
// subject).  Here the order-defining sums are reproduced by construction instead of by serial execution:
//   transposed reverse lists   counts by atomics, offsets by a prefix sum, entries placed by atomics and then each (short) list
//                              SORTED by old vertex id -- the order std::map iteration gives;
//   correction[j]              the contributions (k, w) of a column are collected the same way, sorted by new vertex id k and summed
//                              serially in that order -- the order of the reference's loop over k (:99-118);
//   row sums                   a thread per row, in entry (= key) order.
// So every floating-point sum has the reference's operand order and the weights are bit-identical to the host surgery's
// (tests/test_gpu_search.py, tests/fuzz_resample.py compare them with the oracle bit for bit).  The exclusion-mask variant keeps
// the host path (api.cpp: adaptive_surgery).
#include <climits>
#include <algorithm>

#include "devbuf.hpp"
#include "kernels.hpp"

namespace msm {

namespace {

struct Entry {
    int key;
    double w;
};

// a std::map<int,double> holding the three weights of one query: ascending key, later writes win
__device__ __forceinline__ int small_map(const int *__restrict__ vid, const double *__restrict__ w, int stride, int k, Entry out[3]) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int key = vid[(size_t)j * stride + k];
        const double wt = w[(size_t)j * stride + k];
        int pos = 0;
        while (pos < n && out[pos].key < key) ++pos;
        if (pos < n && out[pos].key == key) {
            out[pos].w = wt;
            continue;
        }
        for (int q = n; q > pos; --q) out[q] = out[q - 1];
        out[pos] = Entry{key, wt};
        ++n;
    }
    return n;
}

// component c of vertex i of coordinate set blockIdx.y at xyz[c * comp + blockIdx.y * set + i] (one mesh: comp = V, set = 0)
__global__ __launch_bounds__(256) void k_tri_areas(const double *__restrict__ xyz, size_t comp, size_t set, const int32_t *__restrict__ tri, int T, double *__restrict__ ta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    xyz += blockIdx.y * set;
    ta += (size_t)blockIdx.y * T;
    const int a = tri[t], b = tri[T + t], c = tri[2 * (size_t)T + t];
    ta[t] = tri_area(mk(xyz[a], xyz[comp + a], xyz[2 * comp + a]), mk(xyz[b], xyz[comp + b], xyz[2 * comp + b]), mk(xyz[c], xyz[comp + c], xyz[2 * comp + c]));
}
// compute_vertex_area, R/mesh.cpp:1275-1283: mean area of the adjacent faces, in trID order
__global__ __launch_bounds__(256) void k_vertex_areas(const double *__restrict__ ta, int T, const int32_t *__restrict__ tid_ptr, const int32_t *__restrict__ tid, int V,
                                                       double *__restrict__ area) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    ta += (size_t)blockIdx.y * T;
    area += (size_t)blockIdx.y * V;
    double sum = 0;
    for (int j = tid_ptr[v]; j < tid_ptr[v + 1]; ++j) sum += ta[tid[j]];
    area[v] = sum / (tid_ptr[v + 1] - tid_ptr[v]);
}

// in-place exclusive prefix sum of n ints, data[n] receives the total.  Two launches: sums of 4096-item blocks, then every block
// adds up the sums before it and scans its items (coalesced rows of 256; a first version with one workgroup and a contiguous
// piece per thread read with a 160-byte stride between lanes and took 60 us per call, three calls per resampling).
// blockIdx.y: problem (data and sums s_data / s_sums further on).
constexpr int kScanBlock = 4096;
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
__global__ __launch_bounds__(256) void k_scan_block_sums(const int *__restrict__ data, int n, int *__restrict__ sums, size_t s_data, size_t s_sums) {
    __shared__ int s_w[4];
    data += blockIdx.y * s_data;
    sums += blockIdx.y * s_sums;
    const int base = blockIdx.x * kScanBlock, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    for (int j = threadIdx.x; j < kScanBlock && base + j < n; j += 256) acc += data[base + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) s_w[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_scan_apply(int *__restrict__ data, int n, const int *__restrict__ sums, size_t s_data, size_t s_sums) {
    __shared__ int s_w[4], s_before;
    data += blockIdx.y * s_data;
    sums += blockIdx.y * s_sums;
    const int base = blockIdx.x * kScanBlock, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 256) acc += sums[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) s_w[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) s_before = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    int run = s_before;
    for (int row = 0; row < kScanBlock && base + row < n; row += 256) {  // uniform
        const int i = base + row + threadIdx.x;
        const int v = i < n ? data[i] : 0;
        const int incl = wave_incl_scan_i(v, lane);
        __syncthreads();  // s_w of the previous row has been read
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int before = run;
        for (int k = 0; k < wv; ++k) before += s_w[k];
        if (i < n) data[i] = before + incl - v;
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) data[n] = run;
}

// the arrays of problem blockIdx.y
__device__ __forceinline__ AdaptiveDevArgs problem_view(AdaptiveDevArgs a) {
    const size_t b = blockIdx.y;
    if (b == 0) return a;
    a.fvid += b * a.s_f, a.fw += b * a.s_f;
    a.rvid += b * a.s_r, a.rw += b * a.s_r;
    a.oldA += b * a.s_oldA, a.newA += b * a.s_newA;
    a.roff += b * a.s_roff, a.rfill += b * a.s_rfill, a.rkey += b * a.s_r3, a.rwt += b * a.s_r3;
    a.coff += b * a.s_coff, a.cfill += b * a.s_cfill, a.ckey += b * a.s_cap, a.cval += b * a.s_cap, a.correction += b * a.s_corr;
    a.tkey += b * a.s_cap, a.tval += b * a.s_cap;
    a.row_ptr += b * a.s_rowptr, a.col += b * a.s_cap, a.val += b * a.s_cap;
    return a;
}

__global__ __launch_bounds__(256) void k_rev_count(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.nOld) return;
    Entry e[3];
    const int n = small_map(a.rvid, a.rw, (int)a.rstride, o, e);
    for (int j = 0; j < n; ++j)
        if (e[j].key >= 0) atomicAdd(&a.roff[e[j].key], 1);
}
__global__ __launch_bounds__(256) void k_rev_fill(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.nOld) return;
    Entry e[3];
    const int n = small_map(a.rvid, a.rw, (int)a.rstride, o, e);
    for (int j = 0; j < n; ++j) {
        if (e[j].key < 0) continue;
        const int pos = a.roff[e[j].key] + atomicAdd(&a.rfill[e[j].key], 1);
        a.rkey[pos] = o;
        a.rwt[pos] = e[j].w;
    }
}
// each list sorted by its int key.  Lists of up to kShortList entries (all of them when the two meshes have similar
// resolutions): a thread per list, insertion sort in place.  which: 0 the transposed reverse lists, 1 the columns.
constexpr int kShortList = 16;
__global__ __launch_bounds__(256) void k_sort_lists(AdaptiveDevArgs a, int which) {
    a = problem_view(a);
    const int *off = which ? a.coff : a.roff;
    const int n = which ? a.nOld : a.nNew;
    int *key = which ? a.ckey : a.rkey;
    double *val = which ? a.cval : a.rwt;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = off[k], e = off[k + 1];
    if (e - b > kShortList) {  // k_sort_long_lists, which only runs when somebody says so
        a.long_flag[2 * blockIdx.y + which] = 1;
        return;
    }
    // the list in registers, every entry to the place its rank gives it (equal keys keep their order, as the insertion sort this replaces did: that one
    // walked the list in memory, a chain of dependent loads and stores per entry -- 32 us for the 780 k lists of a gMSM subject)
    const int len = e - b;
    if (len < 2) return;
    int kk[kShortList];
    double vv[kShortList];
#pragma unroll
    for (int i = 0; i < kShortList; ++i) {
        kk[i] = i < len ? key[b + i] : INT_MAX;
        vv[i] = i < len ? val[b + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < kShortList; ++i) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < kShortList; ++j) rank += (j < len && (kk[j] < kk[i] || (kk[j] == kk[i] && j < i))) ? 1 : 0;
        if (i < len && rank != i) key[b + rank] = kk[i], val[b + rank] = vv[i];
    }
}
// Longer lists (a coarse mesh against a fine one: hundreds of entries per list): a workgroup per list, rank sort -- the keys of
// a list are distinct, so an entry's place is the number of smaller keys -- through a scratch copy.
__global__ __launch_bounds__(256) void k_sort_long_lists(AdaptiveDevArgs a, int which) {
    a = problem_view(a);
    const int *off = which ? a.coff : a.roff;
    int *key = which ? a.ckey : a.rkey;
    double *val = which ? a.cval : a.rwt;
    int *tkey = a.tkey;
    double *tval = a.tval;
    if (!a.long_flag[2 * blockIdx.y + which]) return;  // no long list in this problem (the usual case: meshes of similar resolution)
    const int n = which ? a.nOld : a.nNew;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {  // uniform
        const int b = off[k], len = off[k + 1] - b;
        if (len <= kShortList) continue;
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const int mine = key[b + i];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += key[b + j] < mine;
            tkey[b + rank] = mine;
            tval[b + rank] = val[b + i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            key[b + i] = tkey[b + i];
            val[b + i] = tval[b + i];
        }
        __syncthreads();
    }
}
// :105-109: the forward list unless the transposed reverse list is longer
__global__ __launch_bounds__(256) void k_row_len(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nNew) return;
    Entry f[3];
    const int nf = small_map(a.fvid, a.fw, (int)a.fstride, k, f), nr = a.roff[k + 1] - a.roff[k];
    a.row_ptr[k] = nr <= nf ? nf : nr;
}
// The four kernels below walk a row (a column) of the result.  With meshes of similar resolution a row has 3 - 6 entries and a thread per
// row is right; a coarse new mesh under a fine old one (resample_weights of every iteration: ico6 source -> ico4 control grid) has rows of
// 3 nOld / nNew = 48 entries (768 at ico6 -> ico2) and a thread per row turns the kernel into 40 wavefronts walking dependent loads
// (60 - 90 us each at ico6 -> ico4).  LPR = 1, 8 or 64 lanes share a row; where the reference's sum runs over the row in entry order the
// lanes compute their terms side by side and every lane of the row adds them up in that order (broadcast term by term: same operands,
// same order, same bits).
template <int LPR>
__global__ __launch_bounds__(256) void k_row_write(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, k = gid / LPR, sub = gid % LPR;
    if (k >= a.nNew) return;
    Entry f[3];
    const int nf = small_map(a.fvid, a.fw, (int)a.fstride, k, f), nr = a.roff[k + 1] - a.roff[k];
    const int n = nr <= nf ? nf : nr, at = a.row_ptr[k];
    const double area = a.newA[k];
    for (int j = sub; j < n; j += LPR) {
        int key;
        double w;
        if (nr <= nf) {
            key = j == 0 ? f[0].key : (j == 1 ? f[1].key : f[2].key);
            w = j == 0 ? f[0].w : (j == 1 ? f[1].w : f[2].w);
        } else {
            key = a.rkey[a.roff[k] + j];
            w = a.rwt[a.roff[k] + j];
        }
        a.col[at + j] = key;
        a.val[at + j] = w * area;
        if (key >= 0) atomicAdd(&a.coff[key], 1);
    }
}
template <int LPR>
__global__ __launch_bounds__(256) void k_col_fill(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, k = gid / LPR, sub = gid % LPR;
    if (k >= a.nNew) return;
    for (int e = a.row_ptr[k] + sub; e < a.row_ptr[k + 1]; e += LPR) {
        const int c = a.col[e];
        if (c < 0) continue;
        const int pos = a.coff[c] + atomicAdd(&a.cfill[c], 1);
        a.ckey[pos] = k;
        a.cval[pos] = a.val[e];
    }
}
// the sum of the terms of a row's (column's) entries b <= e < end in entry order; term(e, ok) is evaluated by the lane that owns
// entry e.  All LPR lanes of the row must call this together (LPR lanes of one wavefront, LPR | 64); every lane returns the sum.
template <int LPR, class Term>
__device__ __forceinline__ double ordered_sum(int b, int end, int sub, Term term) {
    double acc = 0.0;
    if (LPR == 1) {
        for (int e = b; e < end; ++e) {
            bool ok;
            const double v = term(e, ok);
            if (ok) acc += v;
        }
        return acc;
    }
    const int first = (threadIdx.x & 63) & ~(LPR - 1);
    for (int c = b; __any(c < end); c += LPR) {
        bool ok = false;
        double v = 0.0;
        if (c + sub < end) v = term(c + sub, ok);
        const unsigned long long oks = __ballot(ok) >> first;
#pragma unroll 8
        for (int j = 0; j < LPR; ++j) {
            const double vj = __shfl(v, first + j, 64);
            if ((oks >> j) & 1ull) acc += vj;
        }
    }
    return acc;
}
// correction[j] = the column's contributions summed in ascending new-vertex order (:111-116 visits k = 0, 1, ...)
template <int LPR>
__global__ __launch_bounds__(256) void k_col_sum(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, j = gid / LPR, sub = gid % LPR;
    const bool live = j < a.nOld;
    if (LPR == 1 && !live) return;
    const int b = live ? a.coff[j] : 0, end = live ? a.coff[j + 1] : 0;
    const double s = ordered_sum<LPR>(b, end, sub, [&](int e, bool &ok) {
        ok = true;
        return a.cval[e];
    });
    if (live && sub == 0) a.correction[j] = s;
}
// :120-137: rescale by oldArea / correction, normalise the row
template <int LPR>
__global__ __launch_bounds__(256) void k_row_finish(AdaptiveDevArgs a) {
    a = problem_view(a);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, k = gid / LPR, sub = gid % LPR;
    const bool live = k < a.nNew;
    if (LPR == 1 && !live) return;
    const int b = live ? a.row_ptr[k] : 0, end = live ? a.row_ptr[k + 1] : 0;
    // a failed search (col < 0, reported through the status word) takes no part in the sum
    const double wsum = ordered_sum<LPR>(b, end, sub, [&](int e, bool &ok) {
        const int c = a.col[e];
        ok = c >= 0;
        return ok ? a.val[e] * (a.oldA[c] / a.correction[c]) : 0.0;
    });
    for (int e = b + sub; e < end; e += LPR) {  // the same product again (same operands: same bits), then the division
        const int c = a.col[e];
        double v = a.val[e];
        if (c >= 0) v *= a.oldA[c] / a.correction[c];
        if (wsum != 0.0) v /= wsum;
        a.val[e] = v;
    }
}
// barycentric_data_interpolation, R/resampler.cpp:40-52: out[d][k] = sum over the row, in entry order
template <int LPR>
__global__ __launch_bounds__(256) void k_apply_rows(AdaptiveDevArgs a, int D, const double *__restrict__ data, double *__restrict__ out, size_t out_stride) {
    out += blockIdx.y * out_stride;
    a = problem_view(a);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = gid / LPR;
    const int sub = (int)(gid % LPR);
    const bool live = i < (size_t)a.nNew * D;
    if (LPR == 1 && !live) return;
    const int d = live ? (int)(i / a.nNew) : 0, k = live ? (int)(i - (size_t)d * a.nNew) : 0;
    const int b = live ? a.row_ptr[k] : 0, end = live ? a.row_ptr[k + 1] : 0;
    const double acc = ordered_sum<LPR>(b, end, sub, [&](int e, bool &ok) {
        const int c = a.col[e];
        ok = c >= 0;
        return ok ? data[(size_t)d * a.nOld + c] * a.val[e] : 0.0;
    });
    if (live && sub == 0) out[i] = acc;
}

}  // namespace

#define MSM_LAUNCH2D(kernel, n, B, ...)                                                                                          \
    do {                                                                                                                         \
        if ((n) > 0) hipLaunchKernelGGL((kernel), dim3((unsigned)(((n) + 255) / 256), (unsigned)(B)), dim3(256), 0, ctx->stream, __VA_ARGS__); \
    } while (0)

static void scan_excl(msm_ctx *ctx, int *data, int n, int *tmp, int B, size_t s_data, size_t s_tmp) {
    const int nb = std::max(1, (n + kScanBlock - 1) / kScanBlock);
    if (nb > 1)  // a single block adds up no sums before it
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nb, (unsigned)B), dim3(256), 0, ctx->stream, data, n, tmp, s_data, s_tmp);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb, (unsigned)B), dim3(256), 0, ctx->stream, data, n, tmp, s_data, s_tmp);
}
int launch_scan_exclusive(msm_ctx *ctx, int *d_data, int n, int *d_tmp) {
    if (n <= 0) return MSM_OK;
    scan_excl(ctx, d_data, n, d_tmp, 1, 0, 0);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_vertex_areas_batch(msm_ctx *ctx, const double *d_xyz, size_t comp, size_t set, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid,
                              int B, double *d_ta, double *d_area) {
    MSM_LAUNCH2D(k_tri_areas, T, B, d_xyz, comp, set, d_tri, T, d_ta);
    MSM_LAUNCH2D(k_vertex_areas, V, B, d_ta, T, d_tid_ptr, d_tid, V, d_area);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_vertex_areas(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid, double *d_ta,
                        double *d_area) {
    return launch_vertex_areas_batch(ctx, d_xyz, (size_t)V, 0, V, d_tri, T, d_tid_ptr, d_tid, 1, d_ta, d_area);
}

// the surgery: every argument is device memory; the caller sizes col / val for 3 * nNew + 3 * nOld entries per problem (a row is
// the forward list, at most 3, or the transposed reverse list, whose lengths add up to at most 3 * nOld)
// lanes per row for rows of about `len` entries
static int lanes_per_row(double len) { return len <= 6.0 ? 1 : (len <= 96.0 ? 8 : 64); }
#define MSM_LAUNCH_LPR(kernel, lpr, n, B, ...)                               \
    do {                                                                      \
        if ((lpr) == 64) MSM_LAUNCH2D(kernel<64>, (size_t)(n) * 64, B, __VA_ARGS__); \
        else if ((lpr) == 8) MSM_LAUNCH2D(kernel<8>, (size_t)(n) * 8, B, __VA_ARGS__); \
        else MSM_LAUNCH2D(kernel<1>, (size_t)(n), B, __VA_ARGS__);            \
    } while (0)

int launch_adaptive_surgery(msm_ctx *ctx, const AdaptiveDevArgs &arg) {
    AdaptiveDevArgs a = arg;
    const int nOld = a.nOld, nNew = a.nNew, B = std::max(a.B, 1);
    if (a.fstride == 0) a.fstride = (size_t)nNew;
    if (a.rstride == 0) a.rstride = (size_t)nOld;
    // rows: the forward list (3) or the transposed reverse list (3 nOld / nNew on average); columns: 3 nNew / nOld or the reverse list's 3
    const int row_lpr = lanes_per_row(3.0 * nOld / std::max(nNew, 1)), col_lpr = lanes_per_row(3.0 * nNew / std::max(nOld, 1));
    if (B == 1) {
        if (a.rfill == a.roff + nNew + 1 && a.coff == a.rfill + nNew && a.cfill == a.coff + nOld + 1 && a.long_flag == a.cfill + nOld) {
            MSM_HIP(hipMemsetAsync(a.roff, 0, sizeof(int) * (2 * (size_t)nNew + 2 * (size_t)nOld + 4), ctx->stream));  // one block (adaptive_weights_dev)
        } else {
            MSM_HIP(hipMemsetAsync(a.roff, 0, sizeof(int) * ((size_t)nNew + 1), ctx->stream));
            MSM_HIP(hipMemsetAsync(a.rfill, 0, sizeof(int) * (size_t)nNew, ctx->stream));
            MSM_HIP(hipMemsetAsync(a.coff, 0, sizeof(int) * ((size_t)nOld + 1), ctx->stream));
            MSM_HIP(hipMemsetAsync(a.cfill, 0, sizeof(int) * (size_t)nOld, ctx->stream));
            MSM_HIP(hipMemsetAsync(a.long_flag, 0, sizeof(int) * 2, ctx->stream));
        }
    } else {  // the problems' arrays lie one after the other
        MSM_HIP(hipMemsetAsync(a.roff, 0, sizeof(int) * a.s_roff * B, ctx->stream));
        MSM_HIP(hipMemsetAsync(a.rfill, 0, sizeof(int) * a.s_rfill * B, ctx->stream));
        MSM_HIP(hipMemsetAsync(a.coff, 0, sizeof(int) * a.s_coff * B, ctx->stream));
        MSM_HIP(hipMemsetAsync(a.cfill, 0, sizeof(int) * a.s_cfill * B, ctx->stream));
        MSM_HIP(hipMemsetAsync(a.long_flag, 0, sizeof(int) * 2 * (size_t)B, ctx->stream));
    }
    MSM_LAUNCH2D(k_rev_count, nOld, B, a);
    scan_excl(ctx, a.roff, nNew, a.scan_tmp, B, a.s_roff, a.s_scan);
    MSM_LAUNCH2D(k_rev_fill, nOld, B, a);
    MSM_LAUNCH2D(k_sort_lists, nNew, B, a, 0);
    // (256 workgroups per problem walk the lists: the usual launch finds no long list, and 2048 x B workgroups that only look at the flag took 26 us to dispatch)
    hipLaunchKernelGGL(k_sort_long_lists, dim3((unsigned)std::min(nNew, B > 1 ? 256 : 2048), (unsigned)B), dim3(256), 0, ctx->stream, a, 0);
    MSM_LAUNCH2D(k_row_len, nNew, B, a);
    scan_excl(ctx, a.row_ptr, nNew, a.scan_tmp, B, a.s_rowptr, a.s_scan);
    MSM_LAUNCH_LPR(k_row_write, row_lpr, nNew, B, a);
    scan_excl(ctx, a.coff, nOld, a.scan_tmp, B, a.s_coff, a.s_scan);
    MSM_LAUNCH_LPR(k_col_fill, row_lpr, nNew, B, a);
    MSM_LAUNCH2D(k_sort_lists, nOld, B, a, 1);
    hipLaunchKernelGGL(k_sort_long_lists, dim3((unsigned)std::min(nOld, B > 1 ? 256 : 2048), (unsigned)B), dim3(256), 0, ctx->stream, a, 1);
    MSM_LAUNCH_LPR(k_col_sum, col_lpr, nOld, B, a);
    MSM_LAUNCH_LPR(k_row_finish, row_lpr, nNew, B, a);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_apply_rows_batch(msm_ctx *ctx, const AdaptiveDevArgs &a, int D, const double *d_data, double *d_out, size_t out_stride) {
    const size_t total = (size_t)a.nNew * D;
    const int lpr = lanes_per_row(3.0 * a.nOld / std::max(a.nNew, 1));
    MSM_LAUNCH_LPR(k_apply_rows, lpr, total, std::max(a.B, 1), a, D, d_data, d_out, out_stride);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_apply_rows(msm_ctx *ctx, int nNew, int nOld, int D, const int *row_ptr, const int *col, const double *val, const double *d_data, double *d_out) {
    AdaptiveDevArgs a{};
    a.nNew = nNew, a.nOld = nOld, a.B = 1;
    a.row_ptr = const_cast<int *>(row_ptr), a.col = const_cast<int *>(col), a.val = const_cast<double *>(val);
    return launch_apply_rows_batch(ctx, a, D, d_data, d_out, 0);
}

}  // namespace msm
