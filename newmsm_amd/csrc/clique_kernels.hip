// clique_kernels.hip -- pairwise and triplet label costs (computePairwiseCost M/DiscreteCostFunction.cpp:190-226,
// computeTripletCost :135-188 with triangle_strain M/reg_tools.cpp:551-646, and the HO classes' triplet_likelihood
// :487-531 / :565-618) as batch kernels: one lane per (clique, label tuple) query.
//
// These evaluations are small closed-form FP64 computations on six or so gathered points; the batch shapes the
// optimisers ask for (8 x T per fusion move, 4 x P, the P x L x L table) expose enough lanes to fill the chip.
#include "kernels.hpp"
#include "search_device.hpp"
#include "similarity_device.hpp"
#include "strain_device.hpp"

namespace msm {

namespace {

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ V3 soa(const double *p, int n, int i) { return mk(p[i], p[n + i], p[2 * n + i]); }
__device__ __forceinline__ V3 aos(const double *p, size_t i) { return mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

constexpr int kBinCache = 64;  // sampled target values kept per lane; longer bins re-sample in the second pass

// one point of an HO bin: the source vertex is projected on the current control triangle, carried to the moved
// triangle by its barycentric coordinates, pushed back to the sphere and sampled on the target
// (HO*::get_target_data, M/DiscreteCostFunction.cpp:498-517 / :574-598).  Returns the hit triangle or an error code.
__device__ __forceinline__ int ho_sample(const CliqueArgs &a, int sv, const V3 &cp0, const V3 &cp1, const V3 &cp2, const V3 &n0, const V3 &n1,
                                         const V3 &n2, double &wa, double &wb, double &wc) {
    const V3 sp = project_point(soa(a.src, a.Nsrc, sv), cp0, cp1, cp2);
    area_weights(cp0, cp1, cp2, sp, wa, wb, wc);  // barycentric(), R/triangle.cpp:159-172
    V3 tmp = mk(n0.x * wa + n1.x * wb + n2.x * wc, n0.y * wa + n1.y * wb + n2.y * wc, n0.z * wa + n1.z * wb + n2.z * wc);
    tmp = scale(normalized(tmp), kRad);
    const int tt = find_closest_triangle(a.tree, tmp);
    if (tt < 0) return tt;
    const TriRec &r = a.tree.rec[tt];
    area_weights(rec_v0(r), rec_v1(r), rec_v2(r), tmp, wa, wb, wc);
    return tt;
}

// HO*::triplet_likelihood, M/DiscreteCostFunction.cpp:520-531 (univariate) / :601-618 (multivariate)
__device__ double triplet_likelihood(const CliqueArgs &a, int t, const int *id, const V3 &n0, const V3 &n1, const V3 &n2) {
    const V3 cp0 = soa(a.cp, a.N, id[0]), cp1 = soa(a.cp, a.N, id[1]), cp2 = soa(a.cp, a.N, id[2]);
    const int beg = a.bin_ptr[t], n = a.bin_ptr[t + 1] - beg;
    const double wmean = (a.absw[id[0]] + a.absw[id[1]] + a.absw[id[2]]) / 3.0;
    const int D = a.D;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double cost = 0.0;
    if (a.kind == MSM_COST_HO_UNIVARIATE) {
        // weighted similarity of the bin's moving values A with the sampled target values B, in the reference's
        // serial order (sparsesimkernel::corr M/similarities.cpp:129-158: means first, then moments)
        double cache[kBinCache];
        auto A = [&](int i) { return a.sfeat[a.bin_idx[beg + i]]; };
        auto W = [&](int i) { return a.cfw ? a.cfw[a.bin_idx[beg + i]] : 1.0; };
        bool failed = false;
        auto B = [&](int i) {
            double wa, wb, wc;
            const int tt = ho_sample(a, a.bin_idx[beg + i], cp0, cp1, cp2, n0, n1, n2, wa, wb, wc);
            if (tt < 0) {
                raise_status(a.status, tt);
                failed = true;
                return nan;
            }
            const TriRec &r = a.tree.rec[tt];
            return wa * a.tfeat[(size_t)r.id[0] * D] + wb * a.tfeat[(size_t)r.id[1] * D] + wc * a.tfeat[(size_t)r.id[2] * D];
        };
        if (a.simmeasure == 4 || a.simmeasure == 5) {
            // DICE ranks every value against every other one: sample each point once (bins longer than the cache re-sample)
            for (int i = 0; i < n && i < kBinCache; ++i) cache[i] = B(i);
            cost = dice_serial(a.simmeasure, n, a.percentile, A, [&](int i) { return i < kBinCache ? cache[i] : B(i); });
        } else if (a.simmeasure == 2) {
            double prod = 0.0, varA = 0.0, varB = 0.0, meanA = 0.0, meanB = 0.0, sum = 0.0;
            for (int i = 0; i < n; ++i) sum += W(i);
            for (int i = 0; i < n; ++i) {
                const double b = B(i);
                if (i < kBinCache) cache[i] = b;
                meanA += W(i) * A(i);
                meanB += W(i) * b;
            }
            if (sum > 0.0) {
                meanA /= sum;
                meanB /= sum;
            }
            for (int i = 0; i < n; ++i) {
                const double b = (i < kBinCache) ? cache[i] : B(i);
                prod += W(i) * (A(i) - meanA) * (b - meanB);
                varA += W(i) * (A(i) - meanA) * (A(i) - meanA);
                varB += W(i) * (b - meanB) * (b - meanB);
            }
            if (sum > 0.0) {
                prod /= sum;
                varA /= sum;
                varB /= sum;
            }
            const double r = (varA == 0.0 || varB == 0.0) ? 0.0 : prod / (sqrt(varA) * sqrt(varB));
            cost = 1 - (1 + r) * 0.5;
        } else {
            double prod = 0.0;
            for (int i = 0; i < n; ++i) {
                const double b = B(i);
                prod += W(i) * (A(i) - b) * (A(i) - b);
            }
            cost = sqrt(prod) / n;
        }
        if (failed) return nan;
    } else {
        for (int i = 0; i < n; ++i) {
            const int sv = a.bin_idx[beg + i];
            double wa, wb, wc;
            const int tt = ho_sample(a, sv, cp0, cp1, cp2, n0, n1, n2, wa, wb, wc);
            if (tt < 0) {
                raise_status(a.status, tt);
                return nan;
            }
            const TriRec &r = a.tree.rec[tt];
            const double *f0 = a.tfeat + (size_t)r.id[0] * D, *f1 = a.tfeat + (size_t)r.id[1] * D, *f2 = a.tfeat + (size_t)r.id[2] * D;
            cost += feature_vector_similarity(a.simmeasure, a.percentile, a.sfeat, a.cfw, a.cfw_rows, a.Nsrc, sv, D, f0, f1, f2, wa, wb, wc);
        }
        if (n > 0) cost /= n;
    }
    return wmean * cost;
}

// deform_anatomy, M/DiscreteCostFunction.cpp:255-301, for one vertex of an anatomical face: the vertex follows the
// proposed control triangle through its barycentric weights (a control point outside the triplet enters as the
// default Point (0,0,0) that std::map::operator[] inserts, :269), is located on the anatomical-resolution sphere
// and carried to the target anatomy with calc_barycentric_weights, summed in ascending vertex id (std::map, :290).
// The reference's moved/transformed maps only cache this per evaluation.
__device__ V3 deform_anatomy_vertex(const CliqueArgs &a, int tindex, const int *id, const V3 *moved, bool &failed) {
    V3 np = mk(0.0, 0.0, 0.0);
    for (int j = a.aw_ptr[tindex]; j < a.aw_ptr[tindex + 1]; ++j) {
        const int cp = a.aw_cp[j];
        const double w = a.aw_val[j];
        V3 v = mk(0.0, 0.0, 0.0);
        if (cp == id[0]) v = moved[0];
        else if (cp == id[1]) v = moved[1];
        else if (cp == id[2]) v = moved[2];
        np = mk(np.x + v.x * w, np.y + v.y * w, np.z + v.z * w);
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int tt = find_closest_triangle(a.atree, np);
    if (tt < 0) {  // the reference warns and continues with a zero triangle (:272-278): NaN weights
        raise_status(a.status, tt);
        failed = true;
        return mk(nan, nan, nan);
    }
    const TriRec &r = a.atree.rec[tt];
    const V3 v0 = rec_v0(r), v1 = rec_v1(r), v2 = rec_v2(r);
    double w[3];
    area_weights(v0, v1, v2, project_point(np, v0, v1, v2), w[0], w[1], w[2]);  // calc_barycentric_weights, R/triangle.cpp:124-145
    int o0 = 0, o1 = 1, o2 = 2;  // ascending vertex id
    if (r.id[o1] < r.id[o0]) { const int s = o0; o0 = o1; o1 = s; }
    if (r.id[o2] < r.id[o0]) { const int s = o0; o0 = o2; o2 = s; }
    if (r.id[o2] < r.id[o1]) { const int s = o1; o1 = o2; o2 = s; }
    V3 out = mk(0.0, 0.0, 0.0);
    const int ord[3] = {o0, o1, o2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int vid = r.id[ord[q]];
        const double wq = w[ord[q]];
        out = mk(out.x + a.atarget[vid] * wq, out.y + a.atarget[a.Va + vid] * wq, out.z + a.atarget[2 * a.Va + vid] * wq);
    }
    return out;
}

// computeTripletCost, M/DiscreteCostFunction.cpp:135-188 (regoption 2/3: spherical strain; 4/5: anatomical strain)
__device__ double triplet_cost(const CliqueArgs &a, int t, int la, int lb, int lc) {
    const int id[3] = {a.triplets[3 * t], a.triplets[3 * t + 1], a.triplets[3 * t + 2]};
    const V3 r[3] = {aos(a.moved, (size_t)id[0] * a.L + la), aos(a.moved, (size_t)id[1] * a.L + lb), aos(a.moved, (size_t)id[2] * a.L + lc)};
    const V3 cur[3] = {soa(a.cp, a.N, id[0]), soa(a.cp, a.N, id[1]), soa(a.cp, a.N, id[2])};
    // only estimate the cost if the move does not fold the triangle
    if (dot(tri_normal(r[0], r[1], r[2]), tri_normal(cur[0], cur[1], cur[2])) < 0.0) return MSM_FOLDING * a.lambda;
    double likelihood = 0.0;
    if (a.kind == MSM_COST_HO_UNIVARIATE || a.kind == MSM_COST_HO_MULTIVARIATE) likelihood = triplet_likelihood(a, t, id, r[0], r[1], r[2]);
    double w;
    if (a.rmode == 4 || a.rmode == 5) {  // :169-182: mean strain of the anatomical faces under this control triangle
        const int beg = a.af_ptr[t], nf = a.af_ptr[t + 1] - beg;
        bool failed = false;
        w = 0.0;
        for (int n = 0; n < nf; ++n) {
            const int f = a.af_idx[beg + n];
            V3 o[3], d[3];
            for (int k = 0; k < 3; ++k) {
                const int v = a.asrc_tri[k * a.Ts + f];
                o[k] = soa(a.asrc, a.Vs, v);
                d[k] = deform_anatomy_vertex(a, v, id, r, failed);
            }
            w += triangular_strain(o, d, a.mu, a.kappa, a.k_exp);
        }
        w = w / (double)nf;
    } else {
        const V3 org[3] = {soa(a.orig, a.Norig, id[0]), soa(a.orig, a.Norig, id[1]), soa(a.orig, a.Norig, id[2])};
        w = triangular_strain(org, r, a.mu, a.kappa, a.k_exp);
    }
    return likelihood + a.lambda * pow(w, a.rexp);
}

// computePairwiseCost, M/DiscreteCostFunction.cpp:190-226, without mutating the control grid
__device__ double pairwise_cost(const CliqueArgs &a, int pair, int la, int lb) {
    const int na = a.pairs[2 * pair], nb = a.pairs[2 * pair + 1];
    const double *R1 = a.rnl + ((size_t)na * a.L + la) * 9, *R2 = a.rnl + ((size_t)nb * a.L + lb) * 9;
    double trace = 0.0;
    for (int r = 0; r < 3; ++r) {  // trace of R1^T R2
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += R1[3 * k + r] * R2[3 * k + r];
        trace = (r == 0) ? s : trace + s;
    }
    const double theta_MVD = 2 * asin(a.mvdmax / (2 * kRad));
    const double theta = acos((trace - 1) / 2);
    double cost = 0.0;
    if (fabs(1 - (trace - 1) / 2) > kEps) {
        const V3 pa = aos(a.moved, (size_t)na * a.L + la), pb = aos(a.moved, (size_t)nb * a.L + lb);
        // folding test over the triangles adjacent to the FIRST node only (:205-211)
        for (int j = a.cp_tid_ptr[na]; j < a.cp_tid_ptr[na + 1]; ++j) {
            const int tt = a.cp_tid[j];
            V3 o[3], p[3];
            for (int k = 0; k < 3; ++k) {
                const int v = a.cp_tri[k * a.Tc + tt];
                o[k] = soa(a.ocp, a.N, v);
                p[k] = (v == na) ? pa : ((v == nb) ? pb : soa(a.cp, a.N, v));
            }
            if (dot(tri_normal(o[0], o[1], o[2]), tri_normal(p[0], p[1], p[2])) < 0.0) return MSM_FOLDING;
        }
        if (a.rexp == 1)
            cost = a.lambda * ((sqrt(2.0) * theta) / theta_MVD);
        else
            cost = a.lambda * pow(((sqrt(2.0) * theta) / theta_MVD), a.rexp);
    }
    return cost;
}

}  // namespace

__global__ __launch_bounds__(128) void k_triplet_batch(CliqueArgs a, const int *__restrict__ qt, const int *__restrict__ qa,
                                                        const int *__restrict__ qb, const int *__restrict__ qc, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = triplet_cost(a, qt[i], qa[i], qb[i], qc[i]);
}

// the 8 costs per triplet of one fusion move, I/Fusion/Fusion.h:181-196: bit order (A,B,C), 0 = current label
__global__ __launch_bounds__(128) void k_triplet_octets(CliqueArgs a, const int *__restrict__ labeling, int label, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * a.T) return;
    const int t = i >> 3, k = i & 7;
    const int la = (k & 4) ? label : labeling[a.triplets[3 * t]];
    const int lb = (k & 2) ? label : labeling[a.triplets[3 * t + 1]];
    const int lc = (k & 1) ? label : labeling[a.triplets[3 * t + 2]];
    out[i] = triplet_cost(a, t, la, lb, lc);
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

int launch_triplet_batch(msm_ctx *ctx, const CliqueArgs &a, const int *qt, const int *qa, const int *qb, const int *qc, int n, double *out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_triplet_batch, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, a, qt, qa, qb, qc, n, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}
int launch_triplet_octets(msm_ctx *ctx, const CliqueArgs &a, const int *labeling, int label, double *out) {
    if (a.T <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_triplet_octets, dim3((8 * a.T + 127) / 128), dim3(128), 0, ctx->stream, a, labeling, label, out);
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
__global__ __launch_bounds__(128) void k_triplet_table(CliqueArgs a, int t0, int t1, double *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t L = (size_t)a.L, per = L * L * L;
    if (i >= (size_t)(t1 - t0) * per) return;
    const int t = t0 + (int)(i / per);
    const size_t r = i % per;
    out[i] = triplet_cost(a, t, (int)(r / (L * L)), (int)((r / L) % L), (int)(r % L));
}

int launch_triplet_table(msm_ctx *ctx, const CliqueArgs &a, int t0, int t1, double *out) {
    const size_t total = (size_t)(t1 - t0) * a.L * a.L * a.L;
    if (total == 0) return MSM_OK;
    hipLaunchKernelGGL(k_triplet_table, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, ctx->stream, a, t0, t1, out);
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
