// clique_device.hpp -- device functions shared by the clique-cost kernels (clique_kernels.hip) and the fused fusion-move
// kernels (move_kernels.hip): computeTripletCost M/DiscreteCostFunction.cpp:135-188, the HO classes' triplet_likelihood
// :487-531 / :565-618, computePairwiseCost :190-226.
#pragma once

#include "kernels.hpp"
#include "search_device.hpp"
#include "similarity_device.hpp"
#include "strain_device.hpp"

namespace msm {

namespace {

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ V3 soa(const double *p, int n, int i) { return mk(p[i], p[n + i], p[2 * n + i]); }
__device__ __forceinline__ V3 aos(const double *p, size_t i) { return mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

// The complete search is the rare path behind the ray table; kept out of line so that the kernels around it do not pay
// its ~170 registers (a call through the stack instead of inlining).
__device__ __noinline__ int search_slow(const DevTree *T, double x, double y, double z) { return find_closest_triangle(*T, mk(x, y, z)); }

// one point of an HO bin: the source vertex is projected on the current control triangle, carried to the moved
// triangle by its barycentric coordinates and pushed back to the sphere (HO*::get_target_data,
// M/DiscreteCostFunction.cpp:498-510 / :574-590)
__device__ __forceinline__ V3 ho_point_position_on(const CliqueArgs &a, int sv, const V3 &cp0, const V3 &cp1, const V3 &cp2, const V3 &s3, double pd,
                                                   const V3 &n0, const V3 &n1, const V3 &n2) {
    const V3 sp = project_with_plane(soa(a.src, a.Nsrc, sv), s3, pd);  // project_point(src, cp0, cp1, cp2) with the plane hoisted
    double wa, wb, wc;
    area_weights(cp0, cp1, cp2, sp, wa, wb, wc);  // barycentric(), R/triangle.cpp:159-172
    const V3 tmp = mk(n0.x * wa + n1.x * wb + n2.x * wc, n0.y * wa + n1.y * wb + n2.y * wc, n0.z * wa + n1.z * wb + n2.z * wc);
    return scale(normalized(tmp), kRad);
}
__device__ __forceinline__ V3 ho_point_position(const CliqueArgs &a, int sv, const V3 &cp0, const V3 &cp1, const V3 &cp2, const V3 &n0, const V3 &n1,
                                                const V3 &n2) {
    V3 s3;
    double pd;
    plane_of(cp0, cp1, cp2, s3, pd);  // the triangle-only half of project_point, R/point.cpp:46-60
    return ho_point_position_on(a, sv, cp0, cp1, cp2, s3, pd, n0, n1, n2);
}

// ... and sampled on the target triangle tt (:511-517 / :591-598): HO univariate -> the interpolated target value; HO
// multivariate -> the point's feature-vector similarity (triplet_likelihood, :601-618)
__device__ __forceinline__ double ho_value_on(const CliqueArgs &a, int sv, const V3 &tmp, int tt) {
    const TriRec &r = a.tree.rec[tt];
    double wa, wb, wc;
    area_weights(rec_v0(r), rec_v1(r), rec_v2(r), tmp, wa, wb, wc);
    const int D = a.D;
    if (a.kind == MSM_COST_HO_UNIVARIATE) return wa * a.tfeat[(size_t)r.id[0] * D] + wb * a.tfeat[(size_t)r.id[1] * D] + wc * a.tfeat[(size_t)r.id[2] * D];
    const double *f0 = a.tfeat + (size_t)r.id[0] * D, *f1 = a.tfeat + (size_t)r.id[1] * D, *f2 = a.tfeat + (size_t)r.id[2] * D;
    if (a.sfeat_vm) return feature_vector_similarity(a.simmeasure, a.percentile, a.sfeat_vm, a.cfw_vm, a.cfw_rows, 0, sv, D, f0, f1, f2, wa, wb, wc);
    return feature_vector_similarity(a.simmeasure, a.percentile, a.sfeat, a.cfw, a.cfw_rows, a.Nsrc, sv, D, f0, f1, f2, wa, wb, wc);
}

// One point of a bin, complete: NaN (and the status word) on a failed search.
__device__ double ho_point_value(const CliqueArgs &a, int sv, const V3 &cp0, const V3 &cp1, const V3 &cp2, const V3 &s3, double pd, const V3 &n0,
                                 const V3 &n1, const V3 &n2) {
    const V3 tmp = ho_point_position_on(a, sv, cp0, cp1, cp2, s3, pd, n0, n1, n2);
    int tt = ray_find(a.tree, tmp);  // simple-surface targets: settled by the ray table nearly always
    if (tt < 0) {
        const DevTree T = a.tree;
        tt = search_slow(&T, tmp.x, tmp.y, tmp.z);
    }
    if (tt < 0) {
        raise_status(a.status, tt);
        return __longlong_as_double(0x7ff8000000000000ll);
    }
    return ho_value_on(a, sv, tmp, tt);
}

// HO*::triplet_likelihood (:520-531 univariate, :601-618 multivariate) from the bin's point values B(0..n) in the
// reference's serial operand order; A(i) / W(i) = moving feature and weight of bin point i (univariate), wmean = the
// mean AbsoluteWeight of the three control points.  A NaN value (failed search) makes the likelihood NaN.
template <class FA, class FW, class FB>
__device__ __forceinline__ double ho_likelihood_core(bool univariate, int simmeasure, double percentile, int n, double wmean, FA A, FW W, FB B) {
    for (int i = 0; i < n; ++i) {
        const double b = B(i);
        if (b != b) return __longlong_as_double(0x7ff8000000000000ll);
    }
    double cost = 0.0;
    if (univariate) {
        if (simmeasure == 4 || simmeasure == 5) {
            cost = dice_serial(simmeasure, n, percentile, A, B);
        } else if (simmeasure == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158
            double prod = 0.0, varA = 0.0, varB = 0.0, meanA = 0.0, meanB = 0.0, sum = 0.0;
            for (int i = 0; i < n; ++i) sum += W(i);
            for (int i = 0; i < n; ++i) {
                meanA += W(i) * A(i);
                meanB += W(i) * B(i);
            }
            if (sum > 0.0) {
                meanA /= sum;
                meanB /= sum;
            }
            for (int i = 0; i < n; ++i) {
                prod += W(i) * (A(i) - meanA) * (B(i) - meanB);
                varA += W(i) * (A(i) - meanA) * (A(i) - meanA);
                varB += W(i) * (B(i) - meanB) * (B(i) - meanB);
            }
            if (sum > 0.0) {
                prod /= sum;
                varA /= sum;
                varB /= sum;
            }
            const double r = (varA == 0.0 || varB == 0.0) ? 0.0 : prod / (sqrt(varA) * sqrt(varB));
            cost = 1 - (1 + r) * 0.5;
        } else {  // sparsesimkernel::SSD, :179-188
            double prod = 0.0;
            for (int i = 0; i < n; ++i) prod += W(i) * (A(i) - B(i)) * (A(i) - B(i));
            cost = sqrt(prod) / n;
        }
    } else {
        for (int i = 0; i < n; ++i) cost += B(i);
        if (n > 0) cost /= n;
    }
    return wmean * cost;
}

__device__ double ho_likelihood(const CliqueArgs &a, int t, const int *id, const double *vals) {
    const int beg = a.bin_ptr[t], n = a.bin_ptr[t + 1] - beg;
    const double wmean = (a.absw[id[0]] + a.absw[id[1]] + a.absw[id[2]]) / 3.0;
    return ho_likelihood_core(
        a.kind == MSM_COST_HO_UNIVARIATE, a.simmeasure, a.percentile, n, wmean, [&](int i) { return a.sfeat[a.bin_idx[beg + i]]; },
        [&](int i) { return a.cfw ? a.cfw[a.bin_idx[beg + i]] : 1.0; }, [&](int i) { return vals[i]; });
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
    const DevTree AT = a.atree;
    const int tt = search_slow(&AT, np.x, np.y, np.z);
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

// computeTripletCost, M/DiscreteCostFunction.cpp:135-188 (regoption 2/3: spherical strain; 4/5: anatomical strain).
// `vals`: the bin's point values for the HO classes (nullptr for the others, whose triplet_likelihood is 0).
// kAnat = false leaves the anatomical branch (and the complete search it drags in) out of the instantiation.
template <bool kAnat>
__device__ double triplet_cost(const CliqueArgs &a, int t, int la, int lb, int lc, const double *vals) {
    const int id[3] = {a.triplets[3 * t], a.triplets[3 * t + 1], a.triplets[3 * t + 2]};
    const V3 r[3] = {aos(a.moved, (size_t)id[0] * a.L + la), aos(a.moved, (size_t)id[1] * a.L + lb), aos(a.moved, (size_t)id[2] * a.L + lc)};
    const V3 cur[3] = {soa(a.cp, a.N, id[0]), soa(a.cp, a.N, id[1]), soa(a.cp, a.N, id[2])};
    // only estimate the cost if the move does not fold the triangle
    if (dot(tri_normal(r[0], r[1], r[2]), tri_normal(cur[0], cur[1], cur[2])) < 0.0) return MSM_FOLDING * a.lambda;
    const double likelihood = vals ? ho_likelihood(a, t, id, vals) : 0.0;
    double w;
    if (kAnat && (a.rmode == 4 || a.rmode == 5)) {  // :169-182: mean strain of the anatomical faces under this control triangle
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
    return likelihood + a.lambda * pow_exp(w, a.rexp);
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

}  // namespace msm
