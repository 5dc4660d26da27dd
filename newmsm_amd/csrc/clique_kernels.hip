// clique_kernels.hip -- pairwise and triplet label costs (computePairwiseCost M/DiscreteCostFunction.cpp:190-226,
// computeTripletCost :135-188 with triangle_strain M/reg_tools.cpp:551-646, and the HO classes' triplet_likelihood
// :487-531 / :565-618) as batch kernels: one lane per (clique, label tuple) query.
//
// These evaluations are small closed-form FP64 computations on six or so gathered points; the batch shapes the
// optimisers ask for (8 x T per fusion move, 4 x P, the P x L x L table) expose enough lanes to fill the chip.
#include "clique_device.hpp"

namespace msm {


template <bool kAnat>
__global__ __launch_bounds__(128) void k_triplet_batch(CliqueArgs a, const int *__restrict__ qt, const int *__restrict__ qa,
                                                        const int *__restrict__ qb, const int *__restrict__ qc, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = triplet_cost<kAnat>(a, qt[i], qa[i], qb[i], qc[i], nullptr);
}

// ------------------------------------------------------------------------------------------------
// Triclique (HO) evaluations, 16 lanes each.  A bin holds ~8 source vertices, each needing a nearest-triangle search;
// with a lane per evaluation a fusion move (8 x T = 40 960 evaluations at ico4) is 640 wavefronts, each a serial
// chain of 8 searches: 126 us, latency bound at less than one wavefront per SIMD.  Here the lanes of a 16-lane group
// sample the bin's points side by side into the group's LDS slice; the group's first lane then runs the
// reference's serial similarity over those values (same operand order as before) and adds the strain term.
// Measured at ico6 / ico4 (bins of ~8): 1 lane 126 us, 4 lanes 111 us, 8 lanes 152 us, 16 lanes 215 us (univariate);
// 395 / 294 / 298 / 323 us (32 features) -- the complete search behind the ray table costs 244 registers, two
// wavefronts per SIMD, so wider groups only add idle lanes to the serial part.  Four lanes for bins up to 256 points,
// sixteen for bins up to 1 024, 64 up to 4 096, the whole workgroup up to 16 384 (the LDS slices must fit).
// ------------------------------------------------------------------------------------------------

template <int kHoLanes>
__device__ void ho_group_eval(const CliqueArgs &a, bool valid, int t, int la, int lb, int lc, double *out) {
    extern __shared__ __align__(16) double s_vals[];  // (256 / kHoLanes) slices of a.bin_cap values
    const int grp = threadIdx.x / kHoLanes, sub = threadIdx.x % kHoLanes;
    // bins beyond kHoBinMax points (kHoLanes = 256: the workgroup works on one evaluation) keep their values in the workgroup's slice in HBM
    double *vals = (kHoLanes == 256 && a.ho_big) ? a.ho_big + (size_t)blockIdx.x * a.bin_cap : s_vals + (size_t)grp * a.bin_cap;
    if (valid) {
        const int id[3] = {a.triplets[3 * t], a.triplets[3 * t + 1], a.triplets[3 * t + 2]};
        const V3 r0 = aos(a.moved, (size_t)id[0] * a.L + la), r1 = aos(a.moved, (size_t)id[1] * a.L + lb), r2 = aos(a.moved, (size_t)id[2] * a.L + lc);
        const V3 cp0 = soa(a.cp, a.N, id[0]), cp1 = soa(a.cp, a.N, id[1]), cp2 = soa(a.cp, a.N, id[2]);
        // a folded proposal never looks at the data (computeTripletCost, :151-152)
        if (!(dot(tri_normal(r0, r1, r2), tri_normal(cp0, cp1, cp2)) < 0.0)) {
            const int beg = a.bin_ptr[t], n = a.bin_ptr[t + 1] - beg;  // n <= a.bin_cap
            V3 s3;
            double pd;
            plane_of(cp0, cp1, cp2, s3, pd);
            for (int i = sub; i < n; i += kHoLanes) vals[i] = ho_point_value(a, a.bin_idx[beg + i], cp0, cp1, cp2, s3, pd, r0, r1, r2);
        }
    }
    __syncthreads();  // (also makes the slice in HBM visible to the lane that reduces it)
    if (valid && sub == 0) *out = triplet_cost<true>(a, t, la, lb, lc, vals);
    __syncthreads();  // the slice is reused by the workgroup's next evaluation
}

// the three kernels walk their evaluations with a workgroup-uniform stride (gridDim.x workgroups of 256 / kHoLanes evaluations each; one round
// unless the launch was capped for the HBM slices of very large bins)
template <int kHoLanes>
__global__ __launch_bounds__(256) void k_triplet_batch_ho(CliqueArgs a, const int *__restrict__ qt, const int *__restrict__ qa,
                                                           const int *__restrict__ qb, const int *__restrict__ qc, int n, double *__restrict__ out) {
    constexpr int per = 256 / kHoLanes;
    for (size_t base = (size_t)blockIdx.x * per; base < (size_t)n; base += (size_t)gridDim.x * per) {
        const size_t i = base + threadIdx.x / kHoLanes;
        const bool valid = i < (size_t)n;
        ho_group_eval<kHoLanes>(a, valid, valid ? qt[i] : 0, valid ? qa[i] : 0, valid ? qb[i] : 0, valid ? qc[i] : 0, out + (valid ? i : 0));
    }
}

template <int kHoLanes>
__global__ __launch_bounds__(256) void k_triplet_octets_ho(CliqueArgs a, const int *__restrict__ labeling, int label, double *__restrict__ out) {
    constexpr int per = 256 / kHoLanes;
    const size_t total = (size_t)8 * a.T;
    for (size_t base = (size_t)blockIdx.x * per; base < total; base += (size_t)gridDim.x * per) {
        const size_t i = base + threadIdx.x / kHoLanes;
        const bool valid = i < total;
        const int t = valid ? (int)(i >> 3) : 0, k = (int)(i & 7);
        const int la = (k & 4) ? label : labeling[a.triplets[3 * t]];
        const int lb = (k & 2) ? label : labeling[a.triplets[3 * t + 1]];
        const int lc = (k & 1) ? label : labeling[a.triplets[3 * t + 2]];
        ho_group_eval<kHoLanes>(a, valid, t, la, lb, lc, out + (valid ? i : 0));
    }
}

template <int kHoLanes>
__global__ __launch_bounds__(256) void k_triplet_table_ho(CliqueArgs a, int t0, int t1, double *__restrict__ out) {
    constexpr int per = 256 / kHoLanes;
    const size_t L = (size_t)a.L, cube = L * L * L, total = (size_t)(t1 - t0) * cube;
    for (size_t base = (size_t)blockIdx.x * per; base < total; base += (size_t)gridDim.x * per) {
        const size_t i = base + threadIdx.x / kHoLanes;
        const bool valid = i < total;
        const size_t r = valid ? i % cube : 0;
        ho_group_eval<kHoLanes>(a, valid, valid ? t0 + (int)(i / cube) : 0, (int)(r / (L * L)), (int)((r / L) % L), (int)(r % L), out + (valid ? i : 0));
    }
}

// ------------------------------------------------------------------------------------------------
// Fusion move of the HO classes on a ray-table target, as three lean kernels (the unary table's split): sample every
// (evaluation, bin point) with the ray table only (no complete search in the kernel: 8 waves/SIMD instead of 2),
// complete search for the few the table could not settle, then the serial similarity + strain per evaluation.
// The values live in a.ho_vals at 8 * bin_ptr[t] + k * n_t + i.
// ------------------------------------------------------------------------------------------------
struct OctetEval {
    int t, la, lb, lc;
    size_t offset;
};
__device__ __forceinline__ OctetEval octet_eval(const CliqueArgs &a, const int *labeling, int label, int e) {
    OctetEval q;
    q.t = e >> 3;
    const int k = e & 7;
    q.la = (k & 4) ? label : labeling[a.triplets[3 * q.t]];
    q.lb = (k & 2) ? label : labeling[a.triplets[3 * q.t + 1]];
    q.lc = (k & 1) ? label : labeling[a.triplets[3 * q.t + 2]];
    const int beg = a.bin_ptr[q.t], n = a.bin_ptr[q.t + 1] - beg;
    q.offset = (size_t)8 * beg + (size_t)k * n;
    return q;
}
struct OctetGeometry {
    V3 r0, r1, r2, cp0, cp1, cp2;
    V3 s3;      // plane of the current control triangle (the per-triangle half of project_point), shared by all its bin points
    double pd;
    bool folded;
};
__device__ __forceinline__ OctetGeometry octet_geometry(const CliqueArgs &a, const OctetEval &q) {
    const int id[3] = {a.triplets[3 * q.t], a.triplets[3 * q.t + 1], a.triplets[3 * q.t + 2]};
    OctetGeometry g;
    g.r0 = aos(a.moved, (size_t)id[0] * a.L + q.la), g.r1 = aos(a.moved, (size_t)id[1] * a.L + q.lb), g.r2 = aos(a.moved, (size_t)id[2] * a.L + q.lc);
    g.cp0 = soa(a.cp, a.N, id[0]), g.cp1 = soa(a.cp, a.N, id[1]), g.cp2 = soa(a.cp, a.N, id[2]);
    g.folded = dot(tri_normal(g.r0, g.r1, g.r2), tri_normal(g.cp0, g.cp1, g.cp2)) < 0.0;  // computeTripletCost, :151-152
    plane_of(g.cp0, g.cp1, g.cp2, g.s3, g.pd);
    return g;
}

constexpr int kOctLanes = 4;

// workgroup -> first evaluation: the workgroups of one XCD (blockIdx % 8, round-robin dispatch) take a contiguous range
// of evaluations, i.e. of control-grid triangles, so that an XCD's L2 holds one region of the target and of the data
__device__ __forceinline__ int xcd_block(int per_block_evals, int total_evals) {
    const int nblocks = (total_evals + per_block_evals - 1) / per_block_evals;
    const int per = (nblocks + 7) >> 3;
    const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    return b < nblocks ? b : -1;
}

__global__ __launch_bounds__(256) void k_ho_octets_sample(CliqueArgs a, const int *__restrict__ labeling, int label) {
    const int blk = xcd_block(256 / kOctLanes, 8 * a.T);
    if (blk < 0) return;
    const int e = blk * (256 / kOctLanes) + threadIdx.x / kOctLanes, sub = threadIdx.x % kOctLanes;
    if (e >= 8 * a.T) return;
    const OctetEval q = octet_eval(a, labeling, label, e);
    const OctetGeometry g = octet_geometry(a, q);
    if (g.folded) return;  // a folded proposal never looks at the data
    const int beg = a.bin_ptr[q.t], n = a.bin_ptr[q.t + 1] - beg;
    for (int i = sub; i < n; i += kOctLanes) {
        const int sv = a.bin_idx[beg + i];
        const V3 tmp = ho_point_position_on(a, sv, g.cp0, g.cp1, g.cp2, g.s3, g.pd, g.r0, g.r1, g.r2);
        const int tt = ray_find(a.tree, tmp);
        if (tt >= 0) a.ho_vals[q.offset + i] = ho_value_on(a, sv, tmp, tt);
        else a.ho_pending[atomicAdd(a.ho_count, 1u)] = ((unsigned)e << 10) | (unsigned)i;
    }
}

// HO multivariate, D <= 64, SSD / correlation.  A wavefront takes eight evaluations; per round of eight bin points each:
//   geometry   a lane per (evaluation, point): position on the moved triangle, ray-table lookup, barycentric weights
//              (about 1 300 instructions of FP64 geometry -- with eight lanes per point from the start, as in the first
//              version of this kernel, every one of them was issued for 8 points instead of 64: 204 us per move);
//   similarity eight passes; in pass p the eight lanes of group g take the p-th point of evaluation g (its triangle,
//              weights and vertex id come over from the lane that did its geometry) and split its D dimensions
//              (similarity_device.hpp: feature_vector_similarity8).
__global__ __launch_bounds__(256) void k_ho_octets_sample_mv8(CliqueArgs a, const int *__restrict__ labeling, int label) {
    const int blk = xcd_block(32, 8 * a.T);
    if (blk < 0) return;
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const int e = blk * 32 + (threadIdx.x >> 6) * 8 + grp;  // this group's evaluation
    const bool ev = e < 8 * a.T;
    OctetEval q{0, 0, 0, 0, 0};
    OctetGeometry g{};
    g.folded = true;
    int beg = 0, n = 0;
    if (ev) {
        q = octet_eval(a, labeling, label, e);
        g = octet_geometry(a, q);
        beg = a.bin_ptr[q.t];
        n = g.folded ? 0 : a.bin_ptr[q.t + 1] - beg;  // a folded proposal never looks at the data
    }
    int nmax = n;  // wavefront-uniform round count
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
    const int D = a.D;
    for (int base = 0; base < nmax; base += 8) {
        // ---- geometry: lane (grp, j) owns point base + j of evaluation grp
        const int i = base + j;
        const bool have = i < n;
        int sv = 0, tt = -1;
        double wa = 0, wb = 0, wc = 0;
        if (have) {
            sv = a.bin_idx[beg + i];
            const V3 tmp = ho_point_position_on(a, sv, g.cp0, g.cp1, g.cp2, g.s3, g.pd, g.r0, g.r1, g.r2);
            tt = ray_find(a.tree, tmp);
            if (tt >= 0) {
                const TriRec &r = a.tree.rec[tt];
                area_weights(rec_v0(r), rec_v1(r), rec_v2(r), tmp, wa, wb, wc);
            } else {
                a.ho_pending[atomicAdd(a.ho_count, 1u)] = ((unsigned)e << 10) | (unsigned)i;
            }
        }
        // ---- similarity: pass p, group grp -> point base + p of evaluation grp
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            const int src = (lane & ~7) | p;
            const int ptt = __shfl(tt, src, 64), psv = __shfl(sv, src, 64);
            const double pwa = __shfl(wa, src, 64), pwb = __shfl(wb, src, 64), pwc = __shfl(wc, src, 64);
            const bool go = ptt >= 0;
            if (!__any(go)) continue;
            const double *f0 = a.tfeat, *f1 = a.tfeat, *f2 = a.tfeat, *sa = a.sfeat_vm, *cw = nullptr;
            if (go) {
                const TriRec &r = a.tree.rec[ptt];
                f0 = a.tfeat + (size_t)r.id[0] * D, f1 = a.tfeat + (size_t)r.id[1] * D, f2 = a.tfeat + (size_t)r.id[2] * D;
                sa = a.sfeat_vm + (size_t)psv * D;
                cw = a.cfw_vm ? a.cfw_vm + (size_t)psv * a.cfw_rows : nullptr;
            }
            const double c = feature_vector_similarity8(a.simmeasure, go, j, D, sa, cw, a.cfw_rows, f0, f1, f2, pwa, pwb, pwc);
            if (go && j == 0) a.ho_vals[q.offset + base + p] = c;
        }
    }
}

__global__ __launch_bounds__(256) void k_ho_octets_fix(CliqueArgs a, const int *__restrict__ labeling, int label) {
    const unsigned n = *a.ho_count;
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const unsigned per_block = 256 / 8, stride = gridDim.x * per_block;
    // eight lanes per pending point (search_device.hpp: group8_find); wavefront-uniform loop for its ballots
    for (unsigned j0 = blockIdx.x * per_block + (threadIdx.x >> 6) * 8; j0 < n; j0 += stride) {
        const unsigned j = j0 + grp;
        const bool valid = j < n;
        V3 tmp = mk(0.0, 0.0, 0.0);
        size_t slot = 0;
        int sv = 0;
        if (valid) {
            const unsigned p = a.ho_pending[j];
            const int e = (int)(p >> 10), i = (int)(p & 1023u);
            const OctetEval q = octet_eval(a, labeling, label, e);
            const OctetGeometry g = octet_geometry(a, q);
            sv = a.bin_idx[a.bin_ptr[q.t] + i];
            tmp = ho_point_position_on(a, sv, g.cp0, g.cp1, g.cp2, g.s3, g.pd, g.r0, g.r1, g.r2);
            slot = q.offset + i;
        }
        const int found = group8_find(a.tree, valid, tmp, lane);
        if (valid && sub == 0) {
            const int tt = found == kGroupUndecided ? find_closest_triangle(a.tree, tmp) : found;
            if (tt < 0) {
                raise_status(a.status, tt);
                a.ho_vals[slot] = __longlong_as_double(0x7ff8000000000000ll);
            } else {
                a.ho_vals[slot] = ho_value_on(a, sv, tmp, tt);
            }
        }
    }
}

__global__ __launch_bounds__(128) void k_ho_octets_reduce(CliqueArgs a, const int *__restrict__ labeling, int label, double *__restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.ho_count = 0u;  // the pending list has been consumed: ready for the next move
    const int blk = xcd_block(128, 8 * a.T);
    if (blk < 0) return;
    const int e = blk * 128 + threadIdx.x;
    if (e >= 8 * a.T) return;
    const OctetEval q = octet_eval(a, labeling, label, e);
    out[e] = triplet_cost<false>(a, q.t, q.la, q.lb, q.lc, a.ho_vals + q.offset);
}

// the 8 costs per triplet of one fusion move, I/Fusion/Fusion.h:181-196: bit order (A,B,C), 0 = current label
template <bool kAnat>
__global__ __launch_bounds__(128) void k_triplet_octets(CliqueArgs a, const int *__restrict__ labeling, int label, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * a.T) return;
    const int t = i >> 3, k = i & 7;
    const int la = (k & 4) ? label : labeling[a.triplets[3 * t]];
    const int lb = (k & 2) ? label : labeling[a.triplets[3 * t + 1]];
    const int lc = (k & 1) ? label : labeling[a.triplets[3 * t + 2]];
    out[i] = triplet_cost<kAnat>(a, t, la, lb, lc, nullptr);
}

// The same with the labeling in the kernel arguments (a byte per control point, as the fused move of the triclique classes has it) and
// the costs written where the optimiser reads them (mapped pinned memory, contiguous stores): the strain-only label step of a
// --regoption=3 registration is one launch and one synchronisation, no copy command.  A status raised by an earlier kernel of the
// context goes to the mapped flag the host looks at.
template <bool kAnat>
__global__ __launch_bounds__(128) void k_triplet_octets_packed(CliqueArgs a, MoveLabels lab, int label, double *__restrict__ out, int *host_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && host_flag) {
        const int raised = *a.status;
        if (raised) __hip_atomic_store(host_flag, raised, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i >= 8 * a.T) return;
    const int t = i >> 3, k = i & 7;
    auto cur = [&](int node) { return (int)((lab.w[node >> 2] >> ((node & 3) * 8)) & 255u); };
    const int la = (k & 4) ? label : cur(a.triplets[3 * t]);
    const int lb = (k & 2) ? label : cur(a.triplets[3 * t + 1]);
    const int lc = (k & 1) ? label : cur(a.triplets[3 * t + 2]);
    out[i] = triplet_cost<kAnat>(a, t, la, lb, lc, nullptr);
}

__global__ __launch_bounds__(256) void k_pairwise_batch(CliqueArgs a, const int *__restrict__ qp, const int *__restrict__ qa,
                                                         const int *__restrict__ qb, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pairwise_cost(a, qp[i], qa[i], qb[i]);
}

// computePairwiseCosts, :228-234: paircosts[(pair*L + labelB)*L + labelA] = computePairwiseCost(pair, labelA, labelB)
__global__ __launch_bounds__(256) void k_pairwise_table(CliqueArgs a, double *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.P * a.L * a.L;
    if (i >= total) return;
    const int la = (int)(i % a.L), lb = (int)((i / a.L) % a.L), pair = (int)(i / ((size_t)a.L * a.L));
    out[i] = pairwise_cost(a, pair, la, lb);
}

static bool is_ho(const CliqueArgs &a) { return a.kind == MSM_COST_HO_UNIVARIATE || a.kind == MSM_COST_HO_MULTIVARIATE; }
// lanes per evaluation by the largest bin: the 256 / lanes LDS slices of bin_cap values must fit 128 KB; beyond kHoBinMax points the whole
// workgroup works on one evaluation with its values in HBM (CliqueArgs::ho_big)
static int ho_lanes(const CliqueArgs &a) { return a.bin_cap <= 256 ? 4 : (a.bin_cap <= 1024 ? 16 : (a.bin_cap <= 4096 ? 64 : 256)); }
// grid and dynamic LDS of an HO launch of `evals` evaluations with `lanes` lanes each
template <class K>
static int ho_config(const CliqueArgs &a, K kernel, int lanes, size_t evals, dim3 &grid, size_t &lds) {
    const int per = 256 / lanes;
    const bool big = lanes == 256 && a.ho_big;
    lds = big ? 0 : sizeof(double) * per * (size_t)a.bin_cap;
    grid = dim3((unsigned)std::min<size_t>((evals + per - 1) / per, big ? (size_t)kHoBigBlocks : ((size_t)1 << 30)));
    if (lds > 64 * 1024) MSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MSM_OK;
}
#define MSM_HO_LAUNCH(kernel, evals, ...)                                                                   \
    do {                                                                                                    \
        dim3 grid_;                                                                                         \
        size_t lds_;                                                                                        \
        const int lanes_ = ho_lanes(a);                                                                     \
        if (lanes_ == 4) {                                                                                  \
            int st_ = ho_config(a, kernel<4>, 4, evals, grid_, lds_);                                       \
            if (st_) return st_;                                                                            \
            hipLaunchKernelGGL(kernel<4>, grid_, dim3(256), lds_, ctx->stream, __VA_ARGS__);                \
        } else if (lanes_ == 16) {                                                                          \
            int st_ = ho_config(a, kernel<16>, 16, evals, grid_, lds_);                                     \
            if (st_) return st_;                                                                            \
            hipLaunchKernelGGL(kernel<16>, grid_, dim3(256), lds_, ctx->stream, __VA_ARGS__);               \
        } else if (lanes_ == 64) { /* bins beyond 1 024 points: control grids far coarser than the data */  \
            int st_ = ho_config(a, kernel<64>, 64, evals, grid_, lds_);                                     \
            if (st_) return st_;                                                                            \
            hipLaunchKernelGGL(kernel<64>, grid_, dim3(256), lds_, ctx->stream, __VA_ARGS__);               \
        } else {                                                                                            \
            int st_ = ho_config(a, kernel<256>, 256, evals, grid_, lds_);                                   \
            if (st_) return st_;                                                                            \
            hipLaunchKernelGGL(kernel<256>, grid_, dim3(256), lds_, ctx->stream, __VA_ARGS__);              \
        }                                                                                                   \
    } while (0)

int launch_triplet_batch(msm_ctx *ctx, const CliqueArgs &a, const int *qt, const int *qa, const int *qb, const int *qc, int n, double *out) {
    if (n <= 0) return MSM_OK;
    if (is_ho(a))
        MSM_HO_LAUNCH(k_triplet_batch_ho, (size_t)n, a, qt, qa, qb, qc, n, out);
    else {
        if (a.rmode == 4 || a.rmode == 5) hipLaunchKernelGGL(k_triplet_batch<true>, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, a, qt, qa, qb, qc, n, out);
        else hipLaunchKernelGGL(k_triplet_batch<false>, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, a, qt, qa, qb, qc, n, out);
    }
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_triplet_octets(msm_ctx *ctx, const CliqueArgs &a, const int *labeling, int label, double *out) {
    if (a.T <= 0) return MSM_OK;
    if (is_ho(a) && a.ho_vals && a.tree.simple && a.tree.ray_G > 0 && a.rmode != 4 && a.rmode != 5) {
        const int per = 256 / kOctLanes;
        const bool mv8 = a.kind == MSM_COST_HO_MULTIVARIATE && a.sfeat_vm && a.D >= 12 && a.D <= kMvLanes * kMvKeep && (a.simmeasure == 1 || a.simmeasure == 2);
        auto grid8 = [](int evals, int per_block) { return dim3((unsigned)(8 * (((evals + per_block - 1) / per_block + 7) / 8))); };  // 8 x blocks per XCD
        if (mv8) hipLaunchKernelGGL(k_ho_octets_sample_mv8, grid8(8 * a.T, 32), dim3(256), 0, ctx->stream, a, labeling, label);
        else hipLaunchKernelGGL(k_ho_octets_sample, grid8(8 * a.T, per), dim3(256), 0, ctx->stream, a, labeling, label);
        MSM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_ho_octets_fix, dim3(64), dim3(256), 0, ctx->stream, a, labeling, label);
        MSM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_ho_octets_reduce, grid8(8 * a.T, 128), dim3(128), 0, ctx->stream, a, labeling, label, out);
    } else if (is_ho(a))
        MSM_HO_LAUNCH(k_triplet_octets_ho, (size_t)8 * a.T, a, labeling, label, out);
    else {
        if (a.rmode == 4 || a.rmode == 5) hipLaunchKernelGGL(k_triplet_octets<true>, dim3((8 * a.T + 127) / 128), dim3(128), 0, ctx->stream, a, labeling, label, out);
        else hipLaunchKernelGGL(k_triplet_octets<false>, dim3((8 * a.T + 127) / 128), dim3(128), 0, ctx->stream, a, labeling, label, out);
    }
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_triplet_octets_packed(msm_ctx *ctx, const CliqueArgs &a, const MoveLabels &lab, int label, double *out, int *host_flag) {
    if (a.T <= 0) return MSM_OK;
    if (is_ho(a)) return fail(MSM_ERR_INVALID, "launch_triplet_octets_packed: the triclique classes have their own move kernel");
    if (a.rmode == 4 || a.rmode == 5) hipLaunchKernelGGL(k_triplet_octets_packed<true>, dim3((8 * a.T + 127) / 128), dim3(128), 0, ctx->stream, a, lab, label, out, host_flag);
    else hipLaunchKernelGGL(k_triplet_octets_packed<false>, dim3((8 * a.T + 127) / 128), dim3(128), 0, ctx->stream, a, lab, label, out, host_flag);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_pairwise_batch(msm_ctx *ctx, const CliqueArgs &a, const int *qp, const int *qa, const int *qb, int n, double *out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_pairwise_batch, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, qp, qa, qb, n, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
// computeTripletCosts, M/DiscreteCostFunction.cpp:245-253: tcosts[t][a][b][c] for the triplets t0 <= t < t1
template <bool kAnat>
__global__ __launch_bounds__(128) void k_triplet_table(CliqueArgs a, int t0, int t1, double *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t L = (size_t)a.L, per = L * L * L;
    if (i >= (size_t)(t1 - t0) * per) return;
    const int t = t0 + (int)(i / per);
    const size_t r = i % per;
    out[i] = triplet_cost<kAnat>(a, t, (int)(r / (L * L)), (int)((r / L) % L), (int)(r % L), nullptr);
}

int launch_triplet_table(msm_ctx *ctx, const CliqueArgs &a, int t0, int t1, double *out) {
    const size_t total = (size_t)(t1 - t0) * a.L * a.L * a.L;
    if (total == 0) return MSM_OK;
    if (is_ho(a)) {
        MSM_HO_LAUNCH(k_triplet_table_ho, total, a, t0, t1, out);
    } else {
        if (a.rmode == 4 || a.rmode == 5) hipLaunchKernelGGL(k_triplet_table<true>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, ctx->stream, a, t0, t1, out);
        else hipLaunchKernelGGL(k_triplet_table<false>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, ctx->stream, a, t0, t1, out);
    }
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_pairwise_table(msm_ctx *ctx, const CliqueArgs &a, double *out) {
    const size_t total = (size_t)a.P * a.L * a.L;
    if (total == 0) return MSM_OK;
    hipLaunchKernelGGL(k_pairwise_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
