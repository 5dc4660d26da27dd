// regtools.cpp -- the callers' side of one registration iteration that touches the meshes of the path:
// unfold (M/reg_tools.cpp:59-178; fold test on the GPU, the serial repair of the few folded vertices on the host in the
// reference's order), variance_normalise (:804-843, host) and the Monte Carlo optimiser that consumes the unary and
// triplet tables (M/mcmc_opt.h:31-134, host).
#include <algorithm>
#include <cmath>
#include <random>

#include "devbuf.hpp"
#include "host_parallel.hpp"
#include "kernels.hpp"

using namespace msm;

namespace msm {
const Adjacency &mesh_adjacency(msm_mesh *m);
}

namespace {

struct FoldMesh {
    double *x;  // 3 x V SoA
    const int32_t *tri;
    const Adjacency *adj;
    int V, T;
    V3 at(int v) const { return mk(x[v], x[V + v], x[2 * (size_t)V + v]); }
    void set(int v, const V3 &p) {
        x[v] = p.x;
        x[V + v] = p.y;
        x[2 * (size_t)V + v] = p.z;
    }
    V3 normal_of(int t) const { return tri_normal(at(tri[t]), at(tri[T + t]), at(tri[2 * (size_t)T + t])); }
    // check_for_intersections, M/reg_tools.cpp:118-129
    bool folded(int v) const {
        const int b = adj->tid_ptr[v], e = adj->tid_ptr[v + 1];
        const V3 n0 = normal_of(adj->tid[b]);
        for (int k = b; k < e; ++k)
            if (dot(n0, normal_of(adj->tid[k])) <= 0.5) return true;
        return false;
    }
};

V3 unit_or_zero(const V3 &a) { return norm(a) > 1e-10 ? normalized(a) : mk(0, 0, 0); }

// computeGradientOfBarycentricTriangle over computeNormal2EdgeOfTriangle, M/reg_tools.cpp:59-93
V3 area_gradient(const V3 &v0, const V3 &v1, const V3 &v2) {
    const V3 s1 = unit_or_zero(sub(v2, v0)), s2 = unit_or_zero(sub(v1, v0));
    const V3 n = unit_or_zero(cross(s1, s2));
    V3 n2e = cross(s2, n);
    if (dot(s1, n2e) < 0) n2e = scale(n2e, -1);
    const double base = norm(sub(v1, v0));
    return scale(scale(n2e, 0.5), base);
}

// spatialgradient, :95-116
V3 spatial_gradient(const FoldMesh &m, int v) {
    const V3 ci = m.at(v);
    V3 grad = mk(0, 0, 0);
    for (int k = m.adj->tid_ptr[v]; k < m.adj->tid_ptr[v + 1]; ++k) {
        const int t = m.adj->tid[k];
        const V3 v0 = m.at(m.tri[t]), v1 = m.at(m.tri[m.T + t]), v2 = m.at(m.tri[2 * (size_t)m.T + t]);
        V3 dA;
        if (norm(sub(ci, v0)) == 0) dA = area_gradient(v1, v2, v0);
        else if (norm(sub(ci, v1)) == 0) dA = area_gradient(v2, v0, v1);
        else dA = area_gradient(v0, v1, v2);
        grad = mk(grad.x + dA.x, grad.y + dA.y, grad.z + dA.z);
    }
    return grad;
}

}  // namespace

extern "C" {

int msm_mesh_unfold(msm_mesh *m, double radius, int32_t *passes, int32_t *first_folded) {
    if (!m) return fail(MSM_ERR_INVALID, "msm_mesh_unfold: null mesh");
    ++m->ctx->epoch;  // (the coordinates may change: a label step queued ahead is not taken, msm_ctx::epoch)
    msm_ctx *ctx = m->ctx;
    MSM_HIP(hipSetDevice(ctx->device));
    const int V = m->V, T = m->T;
    const Adjacency &adj = mesh_adjacency(m);
    if (!m->d_tid_ptr) {
        MSM_HIP(msm::pool_malloc((void **)&m->d_tid_ptr, sizeof(int32_t) * adj.tid_ptr.size()));
        MSM_HIP(msm::pool_malloc((void **)&m->d_tid, sizeof(int32_t) * std::max<size_t>(adj.tid.size(), 1)));
        MSM_HIP(msm::pool_malloc((void **)&m->d_fold, sizeof(int32_t) * (2 + (size_t)V)));
        // (through the context's pinned staging block, like every upload of the library: the GPU never reads the caller's pageable pages, whose pinning by the
        // runtime for an asynchronous copy outlives nothing the library controls)
        int st = upload_staged(ctx, m->d_tid_ptr, adj.tid_ptr.data(), sizeof(int32_t) * adj.tid_ptr.size());
        if (!st && !adj.tid.empty()) st = upload_staged(ctx, m->d_tid, adj.tid.data(), sizeof(int32_t) * adj.tid.size());
        if (st) return st;
    }
    if (passes) *passes = 0;
    if (first_folded) *first_folded = 0;
    std::vector<int32_t> flags;
    std::vector<int> folded;
    std::vector<V3> grads;
    FoldMesh fm{m->xyz.data(), m->tri.data(), &adj, V, T};
    for (int it = 0;; ++it) {
        int st = launch_fold_detect(ctx, m->d_xyz, V, m->d_tri, T, m->d_tid_ptr, m->d_tid, m->d_fold);
        if (st) return st;
        int32_t head[2];
        MSM_TRY(stage_d2h(ctx, head, m->d_fold, sizeof(head)));
        MSM_TRY(ctx_sync(ctx));
        if (head[1] > 0) return fail(MSM_ERR_INVALID, "get_triangle: index exceeds face dimensions");  // a vertex without triangles, R/mesh.h:82-86
        if (it == 0 && first_folded) *first_folded = head[0];
        if (head[0] == 0) break;  // the usual case: nothing is folded and nothing leaves the GPU but two counters
        flags.resize(V);
        MSM_TRY(stage_d2h(ctx, flags.data(), m->d_fold + 2, sizeof(int32_t) * (size_t)V));
        MSM_TRY(ctx_sync(ctx));
        folded.clear();
        for (int v = 0; v < V; ++v)
            if (flags[v]) folded.push_back(v);
        // gradients of all folded vertices first (:152-154), then the moves one by one, each test seeing the earlier moves
        grads.resize(folded.size());
        for (size_t k = 0; k < folded.size(); ++k) grads[k] = spatial_gradient(fm, folded[k]);
        for (size_t k = 0; k < folded.size(); ++k) {
            double step = 1.0;
            const V3 ci = fm.at(folded[k]), g = grads[k];
            V3 pp;
            do {
                pp = normalized(sub(ci, scale(g, step)));
                fm.set(folded[k], scale(pp, radius));
                step *= 0.5;
            } while (fm.folded(folded[k]) && step > 1e-3);
            fm.set(folded[k], scale(pp, radius));
        }
        m->tree_valid = false;
        st = upload_staged(ctx, m->d_xyz, m->xyz.data(), sizeof(double) * 3 * (size_t)V);
        if (st) return st;
        if (passes) *passes = it + 1;
        if (it + 1 == 1000) break;
    }
    MSM_TRY(ctx_sync(ctx));
    return MSM_OK;
}

int msm_variance_normalise(double *data, int32_t D, int32_t V, const double *excl) {
    if (!data || D < 0 || V < 0) return fail(MSM_ERR_INVALID, "msm_variance_normalise: bad arguments");
    // the recurrence of a row is serial (a division per value: 0.4 ms per ico6 row); the rows are independent and run on the host workers
    parallel_chunks(D, D >= 4 ? host_workers() : 1, [&](int, int d_begin, int d_end) {
    for (int d = d_begin; d < d_end; ++d) {
        double *row = data + (size_t)d * V;
        double mean = 0.0, var = 0.0;
        size_t n = 0;
        for (int i = 0; i < V; ++i) {
            if (excl && !(excl[i] > 0.0)) continue;
            const double delta = row[i] - mean;
            mean += delta / (double)(n + 1);
            var += delta * (row[i] - mean);
            ++n;
        }
        var /= (double)(n - 1);  // unsigned, as _data[i].size() - 1 at :829
        const double sd = std::sqrt(var);
        for (int i = 0; i < V; ++i) {
            if (excl && !(excl[i] > 0.0)) continue;
            row[i] -= mean;
            if (var > 0.0) row[i] /= sd;
        }
    }
    });
    return MSM_OK;
}

int msm_mcmc_optimise(const double *unary, const double *tcosts, const int32_t *triplets, int32_t N, int32_t L, int32_t T, double mcparam,
                      int32_t iters, uint64_t seed, int32_t *labeling) {
    if (!unary || !tcosts || !triplets || !labeling || N <= 0 || L <= 0 || T < 0 || iters < 0) return fail(MSM_ERR_INVALID, "msm_mcmc_optimise: bad arguments");
    if (!(mcparam > 0.0 && mcparam <= 1.0)) return fail(MSM_ERR_INVALID, "msm_mcmc_optimise: the geometric distribution parameter must be in (0, 1]");
    for (int i = 0; i < N; ++i)
        if (labeling[i] < 0 || labeling[i] >= L) return fail(MSM_ERR_INVALID, "msm_mcmc_optimise: label of node %d out of range", i);
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i)
        if (triplets[i] < 0 || triplets[i] >= N) return fail(MSM_ERR_INVALID, "msm_mcmc_optimise: triplet node out of range");
    std::mt19937 gen((std::mt19937::result_type)seed);
    std::geometric_distribution<> distribution(mcparam);
    const size_t L2 = (size_t)L * L, L3 = L2 * L;
    double costs[8];
    for (int i = 0; i < iters; ++i)
        for (int t = 0; t < T; ++t) {
            int label;
            do { label = distribution(gen); } while (label >= L);
            const int node[3] = {triplets[3 * t], triplets[3 * t + 1], triplets[3 * t + 2]};
            const int cur[3] = {labeling[node[0]], labeling[node[1]], labeling[node[2]]};
            const double *tc = tcosts + (size_t)t * L3;
            // costs[4a + 2b + c]: a/b/c = 1 takes the proposed label for node A/B/C (mcmc_opt.h:58-90)
            for (int k = 0; k < 8; ++k) {
                const int la = (k & 4) ? label : cur[0], lb = (k & 2) ? label : cur[1], lc = (k & 1) ? label : cur[2];
                costs[k] = tc[(size_t)la * L2 + (size_t)lb * L + lc] +
                           (unary[(size_t)la * N + node[0]] + unary[(size_t)lb * N + node[1]] + unary[(size_t)lc * N + node[2]]) / 3.0;
            }
            const int best = (int)(std::min_element(costs, costs + 8) - costs);
            if (best & 4) labeling[node[0]] = label;
            if (best & 2) labeling[node[1]] = label;
            if (best & 1) labeling[node[2]] = label;
        }
    return MSM_OK;
}

int msm_fusion_icm_step(const double *unary2, const double *quads, const int32_t *pairs, int32_t P, const double *octets, const int32_t *triplets, int32_t T,
                        int32_t N, int32_t max_passes, int32_t *x) {
    if (!x || N <= 0 || T < 0 || P < 0 || max_passes < 0 || (T > 0 && (!octets || !triplets)) || (P > 0 && (!quads || !pairs)))
        return fail(MSM_ERR_INVALID, "msm_fusion_icm_step: bad arguments");
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i)
        if (triplets[i] < 0 || triplets[i] >= N) return fail(MSM_ERR_INVALID, "msm_fusion_icm_step: triplet node out of range");
    for (int64_t i = 0; i < 2 * (int64_t)P; ++i)
        if (pairs[i] < 0 || pairs[i] >= N) return fail(MSM_ERR_INVALID, "msm_fusion_icm_step: pair node out of range");
    // the cliques of every node (counting sort by node; pairs first, then triplets, each ascending): entry = clique * arity + position
    std::vector<int64_t> pptr((size_t)N + 1, 0), tptr((size_t)N + 1, 0);
    for (int64_t i = 0; i < 2 * (int64_t)P; ++i) ++pptr[(size_t)pairs[i] + 1];
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i) ++tptr[(size_t)triplets[i] + 1];
    for (int v = 0; v < N; ++v) pptr[(size_t)v + 1] += pptr[(size_t)v], tptr[(size_t)v + 1] += tptr[(size_t)v];
    std::vector<int64_t> pinc(2 * (size_t)P), tinc(3 * (size_t)T);
    {
        std::vector<int64_t> fill(pptr.begin(), pptr.end() - 1);
        for (int64_t p = 0; p < P; ++p)
            for (int j = 0; j < 2; ++j) pinc[(size_t)fill[(size_t)pairs[2 * p + j]]++] = 2 * p + j;
        fill.assign(tptr.begin(), tptr.end() - 1);
        for (int64_t t = 0; t < T; ++t)
            for (int j = 0; j < 3; ++j) tinc[(size_t)fill[(size_t)triplets[3 * t + j]]++] = 3 * t + j;
    }
    std::fill(x, x + N, 0);
    for (int pass = 0; pass < max_passes; ++pass) {
        bool changed = false;
        for (int v = 0; v < N; ++v) {
            double e[2] = {unary2 ? unary2[2 * (size_t)v] : 0.0, unary2 ? unary2[2 * (size_t)v + 1] : 0.0};
            for (int64_t k = pptr[(size_t)v]; k < pptr[(size_t)v + 1]; ++k) {
                const int64_t p = pinc[(size_t)k] >> 1;
                const int j = (int)(pinc[(size_t)k] & 1);
                const int other = x[pairs[2 * p + (1 - j)]];
                // quads[4 p + 2 x_a + x_b]
                e[0] += quads[4 * (size_t)p + (j == 0 ? other : 2 * other)];
                e[1] += quads[4 * (size_t)p + (j == 0 ? 2 + other : 2 * other + 1)];
            }
            for (int64_t k = tptr[(size_t)v]; k < tptr[(size_t)v + 1]; ++k) {
                const int64_t t = tinc[(size_t)k] / 3;
                const int j = (int)(tinc[(size_t)k] - 3 * t);
                int bits = 0;  // the combination with x_v = 0: 4 x_a + 2 x_b + x_c
                for (int q = 0; q < 3; ++q)
                    if (q != j) bits |= x[triplets[3 * t + q]] << (2 - q);
                e[0] += octets[8 * (size_t)t + bits];
                e[1] += octets[8 * (size_t)t + (bits | (1 << (2 - j)))];
            }
            const int cur = x[v];
            if (e[1 - cur] < e[cur]) {  // strictly lower (never for NaN)
                x[v] = 1 - cur;
                changed = true;
            }
        }
        if (!changed) break;
    }
    return MSM_OK;
}

// The multi-label pairwise MRF of --regoption=1 (unary table + pair tables: what FPD::FastPD reads, I/FastPD/FastPD.h:126,213,224) solved by
// iterated conditional modes -- a STAND-IN for FastPD (licence-restricted, FSL-bound), the counterpart of msm_fusion_icm_step for the
// regoption-1 caller loop (M/mesh_registration.cpp:182-188): nodes in ascending order, each takes the label of lowest cost given its
// neighbours (the lowest label on ties), until a pass changes nothing or max_passes.  unary[label * N + node]; paircosts[(pair * L +
// labelB) * L + labelA] with labelA the label of pairs[2 pair] (M/DiscreteCostFunction.cpp:228-234); labeling: start in, result out.
int msm_pairwise_icm(const double *unary, const double *paircosts, const int32_t *pairs, int32_t N, int32_t L, int32_t P, int32_t max_passes, int32_t *labeling) {
    if (!unary || !labeling || N <= 0 || L <= 0 || P < 0 || max_passes < 0 || (P > 0 && (!paircosts || !pairs))) return fail(MSM_ERR_INVALID, "msm_pairwise_icm: bad arguments");
    for (int64_t i = 0; i < 2 * (int64_t)P; ++i)
        if (pairs[i] < 0 || pairs[i] >= N) return fail(MSM_ERR_INVALID, "msm_pairwise_icm: pair node out of range");
    for (int v = 0; v < N; ++v)
        if (labeling[v] < 0 || labeling[v] >= L) return fail(MSM_ERR_INVALID, "msm_pairwise_icm: labeling[%d] out of range", v);
    std::vector<int64_t> ptr((size_t)N + 1, 0);
    for (int64_t i = 0; i < 2 * (int64_t)P; ++i) ++ptr[(size_t)pairs[i] + 1];
    for (int v = 0; v < N; ++v) ptr[(size_t)v + 1] += ptr[(size_t)v];
    std::vector<int64_t> inc(2 * (size_t)P), fill(ptr.begin(), ptr.end() - 1);
    for (int64_t p = 0; p < P; ++p)
        for (int j = 0; j < 2; ++j) inc[(size_t)fill[(size_t)pairs[2 * p + j]]++] = 2 * p + j;
    std::vector<double> e((size_t)L);
    for (int pass = 0; pass < max_passes; ++pass) {
        bool changed = false;
        for (int v = 0; v < N; ++v) {
            for (int l = 0; l < L; ++l) e[(size_t)l] = unary[(size_t)l * N + v];
            for (int64_t k = ptr[(size_t)v]; k < ptr[(size_t)v + 1]; ++k) {
                const int64_t p = inc[(size_t)k] >> 1;
                const int j = (int)(inc[(size_t)k] & 1);
                const int other = labeling[pairs[2 * p + (1 - j)]];
                const double *tab = paircosts + (size_t)p * L * L;
                if (j == 0) for (int l = 0; l < L; ++l) e[(size_t)l] += tab[(size_t)other * L + l];  // v is node A: labelA = l, labelB = other
                else for (int l = 0; l < L; ++l) e[(size_t)l] += tab[(size_t)l * L + other];
            }
            int best = labeling[v];
            for (int l = 0; l < L; ++l)
                if (e[(size_t)l] < e[(size_t)best] || (e[(size_t)l] == e[(size_t)best] && l < best)) best = l;
            if (best != labeling[v]) {
                labeling[v] = best;
                changed = true;
            }
        }
        if (!changed) break;
    }
    return MSM_OK;
}

}  // extern "C"
