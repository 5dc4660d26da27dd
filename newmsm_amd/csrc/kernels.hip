// kernels.hip -- hand-written gfx950 kernels of libmsmhip and their launch wrappers.
//
// All arithmetic that decides an index (which triangle, which patch) is FP64 in the reference's own
// operation order (this file is compiled with -ffp-contract=off).  No MFMA: the path is gather /
// compare / short reductions.  Wavefront = 64 lanes throughout.
#include "kernels.hpp"
#include "search_device.hpp"

namespace msm {

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// msm_query_triangles: one lane per query point (Resampler::get_barycentric_weights, R/resampler.cpp:142-167)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_query(DevTree T, const double *__restrict__ q, int N, int *__restrict__ tri_id,
                                                int *__restrict__ vid, double *__restrict__ w, int mode, int *status) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const V3 p = mk(q[i], q[N + i], q[2 * N + i]);
        const int t = find_closest_triangle(T, p);
        if (t < 0) {
            raise_status(status, t);
            if (tri_id) tri_id[i] = t;
            if (vid) vid[i] = vid[N + i] = vid[2 * N + i] = -1;
            if (w) w[i] = w[N + i] = w[2 * N + i] = 0.0;
            continue;
        }
        const TriRec &r = T.rec[t];
        if (tri_id) tri_id[i] = t;
        if (vid) {
            vid[i] = r.id[0];
            vid[N + i] = r.id[1];
            vid[2 * N + i] = r.id[2];
        }
        if (w) {
            // calc_barycentric_weights projects the query first (R/triangle.cpp:130); barycentric_interpolation does not
            const V3 v0 = rec_v0(r), v1 = rec_v1(r), v2 = rec_v2(r);
            const V3 pp = (mode == MSM_WEIGHTS_PROJECTED) ? project_point(p, v0, v1, v2) : p;
            double wa, wb, wc;
            area_weights(v0, v1, v2, pp, wa, wb, wc);
            w[i] = wa;
            w[N + i] = wb;
            w[2 * N + i] = wc;
        }
    }
}

// Octree::get_closest_vertex_ID, R/octree.cpp:216-233
__global__ __launch_bounds__(256) void k_closest_vertex(DevTree T, const double *__restrict__ q, int N, int *__restrict__ out, int *status) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const V3 p = mk(q[i], q[N + i], q[2 * N + i]);
        const int t = find_closest_triangle(T, p);
        if (t < 0) {
            raise_status(status, t);
            out[i] = t;
            continue;
        }
        const TriRec &r = T.rec[t];
        double dist = DBL_MAX;
        int best = 0;
        const V3 vv[3] = {rec_v0(r), rec_v1(r), rec_v2(r)};
        const int ids[3] = {r.id[0], r.id[1], r.id[2]};
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double d = norm(sub(p, vv[v]));
            if (d < dist) {
                best = ids[v];
                dist = d;
            }
        }
        out[i] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// get_source_data range test (within_controlpt_range, M/DiscreteCostFunction.cpp:102-107): one
// workgroup per control point sweeps all source vertices and writes the in-range ones, in ascending
// order, to the control point's slot.  The geodesic distance needs asin(); device and host libm may
// differ in the last bit, and this workload has exact ties (a control point's farthest neighbour
// sits exactly at range*MAXSEP), so entries within 1e-11 of the threshold are only flagged (bit 31)
// and the host decides them with its own libm, like the reference would.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_range(const double *__restrict__ cp, int Ncp, const double *__restrict__ src, int Nsrc,
                                                const double *__restrict__ maxsep, double range, int cap,
                                                uint32_t *__restrict__ slots, int *__restrict__ counts) {
    const int k = blockIdx.x;
    if (k >= Ncp) return;
    const V3 c = mk(cp[k], cp[Ncp + k], cp[2 * Ncp + k]);
    const double thr = range * maxsep[k];
    const double slack = fabs(thr) * 1e-11;
    const double lo = thr - slack, hi = thr + slack;
    __shared__ int wave_count[4];
    __shared__ int running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < Nsrc; base += 256) {
        const int i = base + threadIdx.x;
        int state = 0;  // 0 out, 1 in, 2 undecided
        if (i < Nsrc) {
            const double d = norm(sub(c, mk(src[i], src[Nsrc + i], src[2 * Nsrc + i])));
            if (!(d > hi)) {  // arc >= chord, so chord > hi is certainly out of range
                const double arc = chord_to_arc(d);
                state = (arc < lo) ? 1 : ((arc > hi) ? 0 : 2);
            }
        }
        const unsigned long long ball = __ballot(state != 0);
        if (lane == 0) wave_count[wave] = __popcll(ball);
        __syncthreads();
        if (state != 0) {
            int pos = running + __popcll(ball & ((1ull << lane) - 1));
            for (int wv = 0; wv < wave; ++wv) pos += wave_count[wv];
            if (pos < cap) slots[(size_t)k * cap + pos] = (uint32_t)i | (state == 2 ? 0x80000000u : 0u);
        }
        __syncthreads();
        if (threadIdx.x == 0) running += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[k] = running;
}

// ------------------------------------------------------------------------------------------------
// Univariate unary table (computeUnaryCosts -> UnivariateNonLinearSRegDiscreteCostFunction::
// computeUnaryCost, M/DiscreteCostFunction.cpp:236-243, :353-383).
//
// One 256-thread workgroup per control point, all labels:
//   phase 0  the control point's patch (source coords, moving feature, weights: "neighbour ring") is
//            gathered once into LDS and reused by every label; lanes 0..L-1 build the L rotation
//            matrices estimate_rotation_matrix(CP, ROT[node]*label) into LDS;
//   phase 1  the L*P point samples are spread over the 256 lanes: rotate, nearest triangle,
//            barycentric interpolation of the reference feature -> LDS;
//   phase 2  one wavefront per label: weighted two-pass Pearson correlation by shuffle reduction,
//            cost = AbsoluteWeights[node] * (1 - (1 + r)/2)  (or weighted SSD).
// blockIdx -> node mapping keeps the nodes of one XCD (blockIdx % 8) contiguous in id, i.e. spatially
// close on the icosphere, so each XCD's L2 holds its own part of the target structures.
// ------------------------------------------------------------------------------------------------
struct UnaryArgs {
    DevTree tree;
    const double *tfeat;  // target feature, V x D (D == 1 here)
    int N;                // control points
    int L;                // labels
    const double *cp;     // 3 x N SoA
    const double *rnl;    // N x L x 9: estimate_rotation_matrix(CP[node], ROT[node]*label[l]) from k_label_rotations
    const double *labels; // 3 x L SoA
    const double *src;    // 3 x Nsrc SoA
    int Nsrc;
    const double *sfeat;  // moving feature D x Nsrc
    const double *cfw;    // weights rows x Nsrc or nullptr (all ones)
    int cfw_rows;
    const int *pptr;      // patches CSR
    const int *pidx;
    const double *absw;   // N
    int pmax;             // largest patch
    int lchunk;           // labels per LDS pass
    int simmeasure;
    double *U;            // L x N
    int *status;
    unsigned long long *nsamples;
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int per = (n + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

__global__ __launch_bounds__(256) void k_unary_univariate(UnaryArgs a) {
    extern __shared__ double lds[];
    const int node = xcd_remap(blockIdx.x, a.N);
    if (node >= a.N) return;
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    double *sx = lds, *sy = sx + a.pmax, *sz = sy + a.pmax, *sA = sz + a.pmax, *sW = sA + a.pmax;
    double *sR = sW + a.pmax;             // lchunk x 9 (reused per chunk)
    double *sT = sR + 9 * a.lchunk;       // lchunk x pmax
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < P; i += 256) {
        const int s = a.pidx[beg + i];
        sx[i] = a.src[s];
        sy[i] = a.src[a.Nsrc + s];
        sz[i] = a.src[2 * a.Nsrc + s];
        sA[i] = a.sfeat[s];
        sW[i] = (a.cfw && a.cfw_rows >= 1) ? a.cfw[s] : 1.0;
    }
    const double absw = a.absw[node];

    for (int l0 = 0; l0 < a.L; l0 += a.lchunk) {
        const int nl = min(a.lchunk, a.L - l0);
        __syncthreads();  // patch ready / previous chunk consumed
        for (int k = tid; k < 9 * nl; k += 256) sR[k] = a.rnl[((size_t)node * a.L + l0) * 9 + k];
        __syncthreads();
        const int total = nl * P;
        for (int s = tid; s < total; s += 256) {
            const int ll = s / P, i = s - ll * P;
            const V3 p = rotate(sR + 9 * ll, mk(sx[i], sy[i], sz[i]));
            const int t = find_closest_triangle(a.tree, p);
            double val;
            if (t < 0) {
                raise_status(a.status, t);
                val = __longlong_as_double(0x7ff8000000000000ll);
            } else {
                const TriRec &r = a.tree.rec[t];
                double wa, wb, wc;
                area_weights(rec_v0(r), rec_v1(r), rec_v2(r), p, wa, wb, wc);  // barycentric_interpolation: raw point
                val = wa * a.tfeat[r.id[0]] + wb * a.tfeat[r.id[1]] + wc * a.tfeat[r.id[2]];
            }
            sT[ll * a.pmax + i] = val;
        }
        __syncthreads();
        for (int ll = wave; ll < nl; ll += 4) {
            const double *B = sT + ll * a.pmax;
            double cost;
            if (a.simmeasure == 2) {
                // sparsesimkernel::corr, M/similarities.cpp:129-158
                double sw = 0, ma = 0, mb = 0;
                for (int i = lane; i < P; i += 64) {
                    sw += sW[i];
                    ma += sW[i] * sA[i];
                    mb += sW[i] * B[i];
                }
                sw = wave_sum(sw);
                ma = wave_sum(ma);
                mb = wave_sum(mb);
                if (sw > 0.0) {
                    ma /= sw;
                    mb /= sw;
                }
                double pr = 0, va = 0, vb = 0;
                for (int i = lane; i < P; i += 64) {
                    const double da = sA[i] - ma, db = B[i] - mb;
                    pr += sW[i] * da * db;
                    va += sW[i] * da * da;
                    vb += sW[i] * db * db;
                }
                pr = wave_sum(pr);
                va = wave_sum(va);
                vb = wave_sum(vb);
                if (sw > 0.0) {
                    pr /= sw;
                    va /= sw;
                    vb /= sw;
                }
                const double r = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
                cost = 1 - (1 + r) * 0.5;  // get_sim_for_min, M/similarities.h:51-52
            } else {
                // sparsesimkernel::SSD, M/similarities.cpp:179-188
                double pr = 0;
                for (int i = lane; i < P; i += 64) {
                    const double df = sA[i] - B[i];
                    pr += sW[i] * df * df;
                }
                pr = wave_sum(pr);
                cost = sqrt(pr) / P;
            }
            if (lane == 0) a.U[(size_t)(l0 + ll) * a.N + node] = absw * cost;
        }
    }
    if (tid == 0 && a.nsamples) atomicAdd(a.nsamples, (unsigned long long)a.L * P);
}

// ------------------------------------------------------------------------------------------------
// The rotation every (control point, label) evaluation starts with:
// estimate_rotation_matrix(_CPgrid[node], ROTATIONS[node] * labels[label]), M/DiscreteCostFunction.cpp:380.
// Kept out of the table kernels: acos/sincos would otherwise set their register budget.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_label_rotations(const double *__restrict__ cp, int N, const double *__restrict__ rot,
                                                          const double *__restrict__ labels, int L, double *__restrict__ rnl,
                                                          double *__restrict__ moved_out, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int node = i / L, l = i - node * L;
    const V3 lab = mk(labels[l], labels[L + l], labels[2 * L + l]);
    const V3 moved = rotate(rot + 9 * (size_t)node, lab);
    if (moved_out) {
        moved_out[3 * (size_t)i] = moved.x;
        moved_out[3 * (size_t)i + 1] = moved.y;
        moved_out[3 * (size_t)i + 2] = moved.z;
    }
    double R[9];
    if (!rotation_matrix(mk(cp[node], cp[N + node], cp[2 * N + node]), moved, R)) raise_status(status, MSM_ERR_ROTATION);
#pragma unroll
    for (int k = 0; k < 9; ++k) rnl[9 * (size_t)i + k] = R[k];
}

int launch_label_rotations(msm_ctx *ctx, const double *d_cp, int N, const double *d_rot, const double *d_labels, int L, double *d_rnl,
                           double *d_moved) {
    hipLaunchKernelGGL(k_label_rotations, dim3((N * L + 255) / 256), dim3(256), 0, ctx->stream, d_cp, N, d_rot, d_labels, L, d_rnl, d_moved,
                       ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline int grid_for(int n, int block, int cap) {
    int g = (n + block - 1) / block;
    return g < 1 ? 1 : (g > cap ? cap : g);
}

int launch_query(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode) {
    if (N <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_query, dim3(grid_for(N, 256, 4096)), dim3(256), 0, ctx->stream, T, d_q, N, d_tri, d_vid, d_w, mode, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_closest_vertex(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_out) {
    if (N <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_closest_vertex, dim3(grid_for(N, 256, 4096)), dim3(256), 0, ctx->stream, T, d_q, N, d_out, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_range(msm_ctx *ctx, const double *d_cp, int Ncp, const double *d_src, int Nsrc, const double *d_maxsep, double range,
                 int cap, uint32_t *d_slots, int *d_counts) {
    hipLaunchKernelGGL(k_range, dim3(Ncp), dim3(256), 0, ctx->stream, d_cp, Ncp, d_src, Nsrc, d_maxsep, range, cap, d_slots, d_counts);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

size_t unary_univariate_lds(int pmax, int lchunk) { return sizeof(double) * ((size_t)5 * pmax + 9 * (size_t)lchunk + (size_t)lchunk * pmax); }

int launch_unary_univariate(msm_ctx *ctx, const UnaryLaunch &u) {
    UnaryArgs a;
    a.tree = u.tree;
    a.tfeat = u.tfeat;
    a.N = u.N;
    a.L = u.L;
    a.cp = u.cp;
    a.rnl = u.rnl;
    a.labels = u.labels;
    a.src = u.src;
    a.Nsrc = u.Nsrc;
    a.sfeat = u.sfeat;
    a.cfw = u.cfw;
    a.cfw_rows = u.cfw_rows;
    a.pptr = u.pptr;
    a.pidx = u.pidx;
    a.absw = u.absw;
    a.pmax = u.pmax;
    a.simmeasure = u.simmeasure;
    a.U = u.U;
    a.status = ctx->d_status;
    a.nsamples = u.nsamples;
    // as many labels per LDS pass as fit in 64 KiB (keeps >= 2 workgroups per CU)
    const size_t budget = 64 * 1024;
    int lchunk = u.L;
    while (lchunk > 1 && unary_univariate_lds(u.pmax, lchunk) > budget) --lchunk;
    if (unary_univariate_lds(u.pmax, lchunk) > 160 * 1024) return fail(MSM_ERR_CAPACITY, "patch of %d points does not fit in LDS", u.pmax);
    if (lchunk > 256) lchunk = 256;
    a.lchunk = lchunk;
    const int per = (u.N + 7) / 8;
    hipLaunchKernelGGL(k_unary_univariate, dim3(8 * per), dim3(256), unary_univariate_lds(u.pmax, lchunk), ctx->stream, a);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
