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

__global__ __launch_bounds__(256) void k_tri_areas(const double *__restrict__ xyz, int V, const int32_t *__restrict__ tri, int T, double *__restrict__ ta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int a = tri[t], b = tri[T + t], c = tri[2 * (size_t)T + t];
    ta[t] = tri_area(mk(xyz[a], xyz[V + a], xyz[2 * (size_t)V + a]), mk(xyz[b], xyz[V + b], xyz[2 * (size_t)V + b]), mk(xyz[c], xyz[V + c], xyz[2 * (size_t)V + c]));
}
// compute_vertex_area, R/mesh.cpp:1275-1283: mean area of the adjacent faces, in trID order
__global__ __launch_bounds__(256) void k_vertex_areas(const double *__restrict__ ta, const int32_t *__restrict__ tid_ptr, const int32_t *__restrict__ tid, int V,
                                                       double *__restrict__ area) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double sum = 0;
    for (int j = tid_ptr[v]; j < tid_ptr[v + 1]; ++j) sum += ta[tid[j]];
    area[v] = sum / (tid_ptr[v + 1] - tid_ptr[v]);
}

// in-place exclusive prefix sum of n ints, data[n] receives the total.  Two launches: sums of 4096-item blocks, then every block
// adds up the sums before it and scans its items (coalesced rows of 256; a first version with one workgroup and a contiguous
// piece per thread read with a 160-byte stride between lanes and took 60 us per call, three calls per resampling).
constexpr int kScanBlock = 4096;
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
__global__ __launch_bounds__(256) void k_scan_block_sums(const int *__restrict__ data, int n, int *__restrict__ sums) {
    __shared__ int s_w[4];
    const int base = blockIdx.x * kScanBlock, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    for (int j = threadIdx.x; j < kScanBlock && base + j < n; j += 256) acc += data[base + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) s_w[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_scan_apply(int *__restrict__ data, int n, const int *__restrict__ sums) {
    __shared__ int s_w[4], s_before;
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

__global__ __launch_bounds__(256) void k_rev_count(const int *__restrict__ rvid, const double *__restrict__ rw, int nOld, int *__restrict__ rcount) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nOld) return;
    Entry e[3];
    const int n = small_map(rvid, rw, nOld, o, e);
    for (int j = 0; j < n; ++j)
        if (e[j].key >= 0) atomicAdd(&rcount[e[j].key], 1);
}
__global__ __launch_bounds__(256) void k_rev_fill(const int *__restrict__ rvid, const double *__restrict__ rw, int nOld, const int *__restrict__ roff,
                                                   int *__restrict__ rfill, int *__restrict__ rkey, double *__restrict__ rwt) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nOld) return;
    Entry e[3];
    const int n = small_map(rvid, rw, nOld, o, e);
    for (int j = 0; j < n; ++j) {
        if (e[j].key < 0) continue;
        const int pos = roff[e[j].key] + atomicAdd(&rfill[e[j].key], 1);
        rkey[pos] = o;
        rwt[pos] = e[j].w;
    }
}
// each list sorted by its int key.  Lists of up to kShortList entries (all of them when the two meshes have similar
// resolutions): a thread per list, insertion sort in place.
constexpr int kShortList = 16;
__global__ __launch_bounds__(256) void k_sort_lists(const int *__restrict__ off, int n, int *__restrict__ key, double *__restrict__ val) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = off[k], e = off[k + 1];
    if (e - b > kShortList) return;  // k_sort_long_lists
    for (int i = b + 1; i < e; ++i) {
        const int kk = key[i];
        const double vv = val[i];
        int j = i - 1;
        while (j >= b && key[j] > kk) {
            key[j + 1] = key[j];
            val[j + 1] = val[j];
            --j;
        }
        key[j + 1] = kk;
        val[j + 1] = vv;
    }
}
// Longer lists (a coarse mesh against a fine one: hundreds of entries per list): a workgroup per list, rank sort -- the keys of
// a list are distinct, so an entry's place is the number of smaller keys -- through a scratch copy.
__global__ __launch_bounds__(256) void k_sort_long_lists(const int *__restrict__ off, int n, int *__restrict__ key, double *__restrict__ val, int *__restrict__ tkey,
                                                          double *__restrict__ tval) {
    const int k = blockIdx.x;
    const int b = off[k], len = off[k + 1] - b;
    if (len <= kShortList) return;
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
}
// :105-109: the forward list unless the transposed reverse list is longer
__global__ __launch_bounds__(256) void k_row_len(const int *__restrict__ fvid, const double *__restrict__ fw, int nNew, const int *__restrict__ roff,
                                                  int *__restrict__ len) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nNew) return;
    Entry f[3];
    const int nf = small_map(fvid, fw, nNew, k, f), nr = roff[k + 1] - roff[k];
    len[k] = nr <= nf ? nf : nr;
}
__global__ __launch_bounds__(256) void k_row_write(const int *__restrict__ fvid, const double *__restrict__ fw, int nNew, const int *__restrict__ roff,
                                                    const int *__restrict__ rkey, const double *__restrict__ rwt, const double *__restrict__ newA,
                                                    const int *__restrict__ row_ptr, int *__restrict__ col, double *__restrict__ val, int *__restrict__ ccount) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nNew) return;
    Entry f[3];
    const int nf = small_map(fvid, fw, nNew, k, f), nr = roff[k + 1] - roff[k];
    const int n = nr <= nf ? nf : nr, at = row_ptr[k];
    for (int j = 0; j < n; ++j) {
        const int key = nr <= nf ? f[j].key : rkey[roff[k] + j];
        const double w = nr <= nf ? f[j].w : rwt[roff[k] + j];
        col[at + j] = key;
        val[at + j] = w * newA[k];
        if (key >= 0) atomicAdd(&ccount[key], 1);
    }
}
__global__ __launch_bounds__(256) void k_col_fill(int nNew, const int *__restrict__ row_ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                   const int *__restrict__ coff, int *__restrict__ cfill, int *__restrict__ ckey, double *__restrict__ cval) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nNew) return;
    for (int e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
        if (col[e] < 0) continue;
        const int pos = coff[col[e]] + atomicAdd(&cfill[col[e]], 1);
        ckey[pos] = k;
        cval[pos] = val[e];
    }
}
// correction[j] = the column's contributions summed in ascending new-vertex order (:111-116 visits k = 0, 1, ...)
__global__ __launch_bounds__(256) void k_col_sum(const int *__restrict__ coff, int nOld, const double *__restrict__ cval, double *__restrict__ correction) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nOld) return;
    double s = 0.0;
    for (int e = coff[j]; e < coff[j + 1]; ++e) s += cval[e];
    correction[j] = s;
}
// :120-137: rescale by oldArea / correction, normalise the row
__global__ __launch_bounds__(256) void k_row_finish(int nNew, const int *__restrict__ row_ptr, const int *__restrict__ col, double *__restrict__ val,
                                                     const double *__restrict__ oldA, const double *__restrict__ correction) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nNew) return;
    double wsum = 0.0;
    for (int e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
        if (col[e] < 0) continue;  // a failed search (reported through the status word)
        val[e] *= oldA[col[e]] / correction[col[e]];
        wsum += val[e];
    }
    if (wsum != 0.0)
        for (int e = row_ptr[k]; e < row_ptr[k + 1]; ++e) val[e] /= wsum;
}
// barycentric_data_interpolation, R/resampler.cpp:40-52: out[d][k] = sum over the row, in entry order
__global__ __launch_bounds__(256) void k_apply_rows(int nNew, int nOld, int D, const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, const double *__restrict__ data, double *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nNew * D) return;
    const int d = (int)(i / nNew), k = (int)(i - (size_t)d * nNew);
    double acc = 0.0;
    for (int e = row_ptr[k]; e < row_ptr[k + 1]; ++e)
        if (col[e] >= 0) acc += data[(size_t)d * nOld + col[e]] * val[e];
    out[i] = acc;
}

}  // namespace

#define MSM_LAUNCH1D(kernel, n, ...)                                                                      \
    do {                                                                                                  \
        if ((n) > 0) hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, ctx->stream, __VA_ARGS__); \
    } while (0)

static void scan_excl(msm_ctx *ctx, int *data, int n, int *tmp) {
    const int nb = std::max(1, (n + kScanBlock - 1) / kScanBlock);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, ctx->stream, data, n, tmp);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, ctx->stream, data, n, tmp);
}

int launch_vertex_areas(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid, double *d_ta,
                        double *d_area) {
    MSM_LAUNCH1D(k_tri_areas, T, d_xyz, V, d_tri, T, d_ta);
    MSM_LAUNCH1D(k_vertex_areas, V, d_ta, d_tid_ptr, d_tid, V, d_area);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// the surgery: every argument is device memory; the caller sizes col / val for 3 * nNew + 3 * nOld entries (a row is the
// forward list, at most 3, or the transposed reverse list, whose lengths add up to at most 3 * nOld).  *d_nnz_out (device int)
// = row_ptr[nNew].
int launch_adaptive_surgery(msm_ctx *ctx, const AdaptiveDevArgs &a) {
    const int nOld = a.nOld, nNew = a.nNew;
    MSM_HIP(hipMemsetAsync(a.roff, 0, sizeof(int) * ((size_t)nNew + 1), ctx->stream));
    MSM_HIP(hipMemsetAsync(a.rfill, 0, sizeof(int) * (size_t)nNew, ctx->stream));
    MSM_HIP(hipMemsetAsync(a.coff, 0, sizeof(int) * ((size_t)nOld + 1), ctx->stream));
    MSM_HIP(hipMemsetAsync(a.cfill, 0, sizeof(int) * (size_t)nOld, ctx->stream));
    MSM_LAUNCH1D(k_rev_count, nOld, a.rvid, a.rw, nOld, a.roff);
    scan_excl(ctx, a.roff, nNew, a.scan_tmp);
    MSM_LAUNCH1D(k_rev_fill, nOld, a.rvid, a.rw, nOld, a.roff, a.rfill, a.rkey, a.rwt);
    MSM_LAUNCH1D(k_sort_lists, nNew, a.roff, nNew, a.rkey, a.rwt);
    hipLaunchKernelGGL(k_sort_long_lists, dim3(nNew), dim3(256), 0, ctx->stream, a.roff, nNew, a.rkey, a.rwt, a.tkey, a.tval);
    MSM_LAUNCH1D(k_row_len, nNew, a.fvid, a.fw, nNew, a.roff, a.row_ptr);
    scan_excl(ctx, a.row_ptr, nNew, a.scan_tmp);
    MSM_LAUNCH1D(k_row_write, nNew, a.fvid, a.fw, nNew, a.roff, a.rkey, a.rwt, a.newA, a.row_ptr, a.col, a.val, a.coff);
    scan_excl(ctx, a.coff, nOld, a.scan_tmp);
    MSM_LAUNCH1D(k_col_fill, nNew, nNew, a.row_ptr, a.col, a.val, a.coff, a.cfill, a.ckey, a.cval);
    MSM_LAUNCH1D(k_sort_lists, nOld, a.coff, nOld, a.ckey, a.cval);
    hipLaunchKernelGGL(k_sort_long_lists, dim3(nOld), dim3(256), 0, ctx->stream, a.coff, nOld, a.ckey, a.cval, a.tkey, a.tval);
    MSM_LAUNCH1D(k_col_sum, nOld, a.coff, nOld, a.cval, a.correction);
    MSM_LAUNCH1D(k_row_finish, nNew, nNew, a.row_ptr, a.col, a.val, a.oldA, a.correction);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_apply_rows(msm_ctx *ctx, int nNew, int nOld, int D, const int *row_ptr, const int *col, const double *val, const double *d_data, double *d_out) {
    const size_t total = (size_t)nNew * D;
    MSM_LAUNCH1D(k_apply_rows, total, nNew, nOld, D, row_ptr, col, val, d_data, d_out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
