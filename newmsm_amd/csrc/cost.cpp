// cost.cpp -- C ABI of the discrete cost function (include/msmhip.h, "discrete cost function").
//
// msm_cost mirrors NonLinearSRegDiscreteCostFunction (M/DiscreteCostFunction.h:83-224): the setters
// take what DiscreteModel hands the cost function each iteration, get_source_data() builds the
// per-control-point patches, and the table / batch evaluators replace the OpenMP loops of
// computeUnaryCosts, computeTripletCosts and Fusion's per-label sweeps with kernel launches.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "cost_internal.hpp"
#include "host_parallel.hpp"

using namespace msm;

namespace msm {
// vertex-major copies of the moving features and their weights: the D values of a vertex in one or two cache lines
int ensure_vertex_major(msm_cost *c) {
    if (c->vm_valid) return MSM_OK;
    msm_ctx *ctx = c->ctx;
    const int Ns = c->source->V, D = c->D, R = c->cfw_rows;
    std::vector<double> vm((size_t)Ns * D);
    for (int d = 0; d < D; ++d)
        for (int v = 0; v < Ns; ++v) vm[(size_t)v * D + d] = c->sfeat[(size_t)d * Ns + v];
    // (through the pinned staging block: an asynchronous copy from a pageable vector of this size is staged by the runtime at well under 1 GB/s --
    // 25 ms for the 10.5 MB of 32 features at ico6, paid by whichever call synchronises next)
    MSM_HIP(c->d_sfeat_vm.ensure(vm.size()));
    {
        const int st = upload_staged(ctx, c->d_sfeat_vm.p, vm.data(), sizeof(double) * vm.size());
        if (st) return st;
    }
    if (!c->cfw.empty()) {
        std::vector<double> wm((size_t)Ns * R);
        for (int r = 0; r < R; ++r)
            for (int v = 0; v < Ns; ++v) wm[(size_t)v * R + r] = c->cfw[(size_t)r * Ns + v];
        MSM_HIP(c->d_cfw_vm.ensure(wm.size()));
        const int st = upload_staged(ctx, c->d_cfw_vm.p, wm.data(), sizeof(double) * wm.size());
        if (st) return st;
    }
    MSM_TRY(ctx_sync(ctx));
    c->vm_valid = true;
    return MSM_OK;
}

int ensure_label_rotations(msm_cost *c) {
    if (c->rotations_valid) return MSM_OK;  // a pure function of the control grid, ROT and the label set
    if (c->L <= 0 || !c->cpgrid) return fail(MSM_ERR_STATE, "msm_cost: labels must be set first");
    const int N = c->cpgrid->V;
    MSM_HIP(c->d_rnl.ensure((size_t)N * c->L * 9));
    MSM_HIP(c->d_moved.ensure((size_t)N * c->L * 3));
    int st = launch_label_rotations(c->ctx, c->cpgrid->d_xyz, N, c->d_rot.p, c->d_labels.p, c->L, c->d_rnl.p, c->d_moved.p);
    if (st) return st;
    c->rotations_valid = true;
    return MSM_OK;
}
}  // namespace msm

namespace {

void invalidate_table(msm_cost *c) {
    (void)drop_pending_move(c);  // a label step queued ahead reads what the caller is about to change
    c->move_valid = false;
    c->table_valid = false;
    c->rotations_valid = false;
    c->h_U.clear();
}

bool is_ho(const msm_cost *c) { return cost_is_ho(c); }

int need(const msm_cost *c, bool cond, const char *what) {
    if (!cond) return fail(MSM_ERR_STATE, "msm_cost: %s must be set first", what);
    return MSM_OK;
}

// cfweight(row, vertex); all ones when none was given (M/mesh_registration.cpp:234-238)
inline double cfw_at(const msm_cost *c, int row, int i) { return c->cfw.empty() ? 1.0 : c->cfw[(size_t)row * c->source->V + i]; }
inline int cfw_nrows(const msm_cost *c) { return c->cfw.empty() ? 1 : c->cfw_rows; }

// Patches by range test (Univariate :338-349, Multivariate :399-404, Patchwise :636-648).
int patches_by_range(msm_cost *c) {
    msm_ctx *ctx = c->ctx;
    const int N = c->cpgrid->V, Ns = c->source->V;
    MSM_TRY(c->d_maxsep.upload(c->maxsep.data(), N, ctx));
    MSM_HIP(c->d_counts.ensure((size_t)N + 1));  // + the number of undecided entries
    MSM_HIP(c->d_chunkb.ensure((size_t)(Ns + 63) / 64 + 1));
    int cap = std::max(128, c->patch_cap_hint);  // the previous call's largest patch: one k_range pass instead of two
    std::vector<int> counts((size_t)N + 1);
    std::vector<uint32_t> slots;
    c->pidx_asc_on_device = false;
    for (int attempt = 0; attempt < 3; ++attempt) {
        MSM_HIP(c->d_slots.ensure((size_t)N * cap));
        int st = launch_range(ctx, c->cpgrid->d_xyz, N, c->source->d_xyz, Ns, c->d_maxsep.p, c->p.range, cap, c->d_slots.p, c->d_counts.p, c->d_chunkb.p,
                              c->d_counts.p + N);
        if (st) return st;
        MSM_TRY(c->d_counts.download(counts.data(), (size_t)N + 1, ctx));
        MSM_TRY(ctx_sync(ctx));
        const int mx = *std::max_element(counts.begin(), counts.begin() + N);
        c->patch_cap_hint = std::max(c->patch_cap_hint, mx + 16);
        if (mx <= cap) break;
        if (attempt == 2) return fail(MSM_ERR_CAPACITY, "patch capacity");
        cap = mx + 16;
    }
    if (counts[N] == 0) {
        // nothing sits on the threshold: the rows are final, and the list is put together where it is used (the host receives
        // the compact list, a third of a megabyte at ico6 / ico4, instead of the slot array)
        c->pptr.assign((size_t)N + 1, 0);
        for (int k = 0; k < N; ++k) c->pptr[k + 1] = c->pptr[k] + counts[k];
        const size_t total = (size_t)c->pptr[N];
        MSM_TRY(c->d_pptr.upload(c->pptr.data(), c->pptr.size(), ctx));
        MSM_HIP(c->d_pidx_asc.ensure(std::max<size_t>(total, 1)));
        int st = launch_patch_compact(ctx, c->d_slots.p, cap, c->d_pptr.p, N, c->d_pidx_asc.p);
        if (st) return st;
        c->pidx.resize(total);
        if (total) MSM_TRY(c->d_pidx_asc.download(c->pidx.data(), total, ctx));
        MSM_TRY(ctx_sync(ctx));
        c->pidx_asc_on_device = true;
        c->ngroups = N;
        return MSM_OK;
    }
    slots.resize((size_t)N * cap);
    MSM_TRY(c->d_slots.download(slots.data(), slots.size(), ctx));
    MSM_TRY(ctx_sync(ctx));
    const double *cp = c->cpgrid->xyz.data(), *src = c->source->xyz.data();
    c->pptr.assign(N + 1, 0);
    c->pidx.clear();
    for (int k = 0; k < N; ++k) {
        c->pptr[k] = (int32_t)c->pidx.size();
        const V3 ck = mk(cp[k], cp[N + k], cp[2 * N + k]);
        for (int j = 0; j < counts[k]; ++j) {
            const uint32_t e = slots[(size_t)k * cap + j];
            const int i = (int)(e & 0x7fffffffu);
            if (e & 0x80000000u) {  // within 1e-11 of the threshold: decide with the host libm, as the reference does
                const double arc = chord_to_arc(norm(sub(ck, mk(src[i], src[Ns + i], src[2 * Ns + i]))));
                if (!(arc < c->p.range * c->maxsep[k])) continue;
            }
            c->pidx.push_back(i);
        }
    }
    c->pptr[N] = (int32_t)c->pidx.size();
    c->ngroups = N;
    return MSM_OK;
}

// Bins by closest control-grid triangle (HOUnivariate :472-483, HOMultivariate :545-561).
int patches_by_triangle(msm_cost *c) {
    const int Ns = c->source->V, Tc = c->cpgrid->T;
    std::vector<int> tri(Ns);
    // resample_weights (next) needs the moved source's tree: its GPU build is queued now and runs while the host builds the control grid's
    // tree for the queries below (0.4 ms at ico4) -- the overlap ensure_tree_pair gives the classes that bin by range
    int st = mesh_tree_on_gpu(c->cpgrid) ? MSM_OK : ensure_tree_begin(c->source);
    if (st) return st;
    st = query_host(c->cpgrid, c->source->xyz.data(), Ns, tri.data(), nullptr, nullptr, MSM_WEIGHTS_RAW, "get_source_data (HO)",
                    c->source->ctx == c->cpgrid->ctx ? c->source->d_xyz : nullptr);  // the source's vertices are in HBM already
    if (st) return st;
    // both trees on the GPU (a control grid of 2 048 triangles or more): the source's build is queued now and runs while the host bins
    if (mesh_tree_on_gpu(c->cpgrid)) {
        st = ensure_tree_begin(c->source);
        if (st) return st;
    }
    c->pptr.assign(Tc + 1, 0);
    for (int i = 0; i < Ns; ++i) c->pptr[tri[i] + 1]++;
    for (int t = 0; t < Tc; ++t) c->pptr[t + 1] += c->pptr[t];
    c->pidx.resize(Ns);
    std::vector<int32_t> fill(c->pptr.begin(), c->pptr.end() - 1);
    for (int i = 0; i < Ns; ++i) c->pidx[fill[tri[i]]++] = i;
    c->ngroups = Tc;
    return MSM_OK;
}

// resample_weights, M/DiscreteCostFunction.cpp:303-323
int resample_weights(msm_cost *c) {
    const int Ns = c->source->V, N = c->cpgrid->V;
    std::vector<double> mw(Ns);
    for (int k = 0; k < Ns; ++k) {
        double best = -DBL_MAX;
        for (int j = 0; j < cfw_nrows(c); ++j)
            if (cfw_at(c, j, k) > best) best = cfw_at(c, j, k);
        mw[k] = best;
    }
    static const bool host_surgery = [] { const char *e = std::getenv("MSMHIP_SURGERY"); return e && std::strcmp(e, "host") == 0; }();
    if (!host_surgery) {
        // queries, weight lists and the weighted sum in HBM (resample_kernels.hip); the N sums come back for msm_cost_absolute_weights
        msm_ctx *ctx = c->ctx;
        AdaptiveDev w;
        int st = adaptive_weights_dev(c->source, c->cpgrid, w);
        if (st) return st;
        MSM_HIP(c->d_maxw.ensure(Ns));
        MSM_HIP(c->d_absw.ensure(N));
        st = upload_staged(ctx, c->d_maxw.p, mw.data(), sizeof(double) * (size_t)Ns);
        if (st) return st;
        st = apply_weights_dev(ctx, w, c->d_maxw.p, 1, c->d_absw.p);
        if (st) return st;
        c->absw.resize(N);
        MSM_TRY(stage_d2h(ctx, c->absw.data(), c->d_absw.p, sizeof(double) * (size_t)N));
        MSM_TRY(ctx_sync(ctx));
        return MSM_OK;
    }
    std::vector<int32_t> rp, col;
    std::vector<double> val;
    int st = adaptive_weights(c->source, c->cpgrid, nullptr, rp, col, val);
    if (st) return st;
    c->absw.resize(N);
    for (int k = 0; k < N; ++k) {
        double acc = 0.0;
        for (int e = rp[k]; e < rp[k + 1]; ++e) acc += mw[col[e]] * val[e];
        c->absw[k] = acc;
    }
    return MSM_OK;
}

int ensure_unary_table(msm_cost *c) {
    if (c->table_valid) return MSM_OK;
    int st = need(c, c->have_source, "get_source_data()");
    if (st) return st;
    st = need(c, c->L > 0, "labels");
    if (st) return st;
    st = need(c, c->target->d_feat != nullptr && c->target->D == c->D, "target features matching the source features");
    if (st) return st;
    msm_ctx *ctx = c->ctx;
    const int N = c->cpgrid->V;
    MSM_HIP(c->d_U.ensure((size_t)c->L * N));
    if (is_ho(c)) {  // the HO classes' computeUnaryCost returns 0, M/DiscreteCostFunction.h:249,258
        MSM_HIP(hipMemsetAsync(c->d_U.p, 0, sizeof(double) * (size_t)c->L * N, ctx->stream));
        c->table_valid = true;
        return MSM_OK;
    }
    st = ensure_rays(c->target);
    if (st) return st;
    UnaryLaunch u;
    u.tree = dev_tree(c->target);
    u.tfeat = c->target->d_feat;
    u.D = c->D;
    u.N = N;
    u.L = c->L;
    u.cp = c->cpgrid->d_xyz;
    st = ensure_label_rotations(c);
    if (st) return st;
    u.rnl = c->d_rnl.p;
    u.labels = c->d_labels.p;
    u.src = c->source->d_xyz;
    u.Nsrc = c->source->V;
    u.sfeat = c->d_sfeat.p;
    u.cfw = c->cfw.empty() ? nullptr : c->d_cfw.p;
    u.cfw_rows = c->cfw_rows;
    u.pptr = c->d_pptr.p;
    u.pidx = c->d_pidx.p;
    u.order = c->d_order.p;
    u.absw = c->d_absw.p;
    u.pmax = c->pmax;
    u.simmeasure = c->p.simmeasure;
    u.percentile = c->p.percentile;
    u.U = c->d_U.p;
    const size_t nsamp = (size_t)c->L * c->pidx.size();
    MSM_HIP(c->d_tval.ensure(nsamp));
    MSM_HIP(c->d_fix_list.ensure(nsamp));
    MSM_HIP(c->d_fix_pt.ensure(3 * nsamp));
    if (!c->d_fix_count.p) MSM_HIP(c->d_fix_count.zero(unary_fix_counter_words(), ctx->stream));  // every launch leaves them zero again
    if (!c->fix_off_valid) {
        std::vector<uint32_t> off;
        unary_fix_offsets(N, c->L, c->pmax, c->pptr.data(), c->order.data(), off);
        MSM_TRY(c->d_fix_off.upload(off.data(), off.size(), ctx));
        MSM_TRY(ctx_sync(ctx));
        c->fix_off_valid = true;
    }
    MSM_HIP(c->d_queues.ensure(N));
    u.ntri = c->target->T;
    u.tval = c->d_tval.p;
    u.fix_list = c->d_fix_list.p;
    u.fix_pt = c->d_fix_pt.p;
    u.fix_cnt = c->d_fix_count.p;
    u.fix_off = c->d_fix_off.p;
    u.redo_list = c->d_queues.p;
    if (c->timing) {
        u.ev_start = c->ev0[c->ev_next];
        u.ev_stop = c->ev1[c->ev_next];
        c->ev_next = (c->ev_next + 1) % (int)c->ev0.size();
        c->ev_count = std::min(c->ev_count + 1, (int)c->ev0.size());
    }
    switch (c->p.kind) {
        case MSM_COST_UNIVARIATE: st = launch_unary_univariate(ctx, u); break;
        case MSM_COST_MULTIVARIATE:
        case MSM_COST_PATCHWISE: {
            {  // both classes: vertex-major copies for the eight-lanes-per-point reductions
                st = ensure_vertex_major(c);
                if (st) return st;
                u.sfeat_vm = c->d_sfeat_vm.p;
                u.cfw_vm = c->cfw.empty() ? nullptr : c->d_cfw_vm.p;
            }
            MSM_HIP(c->d_stri.ensure(nsamp));
            MSM_HIP(c->d_sw3.ensure(3 * nsamp));
            UnaryWeightsScratch w{c->d_stri.p, c->d_sw3.p};
            st = launch_unary_multivariate(ctx, u, w, c->p.kind == MSM_COST_PATCHWISE);
            break;
        }
        default: return fail(MSM_ERR_INVALID, "unknown cost kind %d", c->p.kind);
    }
    if (st) return st;
    c->counters[0] += (int64_t)nsamp;
    c->counters[1] += (int64_t)c->L * N;
    c->table_valid = true;
    return MSM_OK;
}

}  // namespace

extern "C" {

msm_cost *msm_cost_create(msm_ctx *ctx, const msm_cost_params *params) {
    if (!ctx || !params) {
        fail(MSM_ERR_INVALID, "msm_cost_create: null argument");
        return nullptr;
    }
    if (params->kind < MSM_COST_UNIVARIATE || params->kind > MSM_COST_HO_MULTIVARIATE) {
        fail(MSM_ERR_INVALID, "msm_cost_create: unknown cost kind %d", params->kind);
        return nullptr;
    }
    if (params->simmeasure != 1 && params->simmeasure != 2 && params->simmeasure != 4 && params->simmeasure != 5) {
        fail(MSM_ERR_INVALID, "Unknown similarity metric");  // get_sim_for_min, M/similarities.h:57
        return nullptr;
    }
    if ((params->simmeasure == 4 || params->simmeasure == 5) && !(params->percentile > 0.0 + kEps && params->percentile < 1.0 - kEps)) {
        fail(MSM_ERR_INVALID, "Percentile must be between 0 and 1.");  // M/mesh_registration.cpp:782-783
        return nullptr;
    }
    msm_cost *c = new msm_cost();
    c->ctx = ctx;
    c->p = *params;
    (void)hipSetDevice(ctx->device);
    if (c->d_counters.zero(4, ctx->stream) != hipSuccess) {
        fail(MSM_ERR_HIP, "msm_cost_create: allocation failed");
        delete c;
        return nullptr;
    }
    return c;
}

void msm_cost_destroy(msm_cost *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)drop_pending_move(c);
    (void)hipStreamSynchronize(c->ctx->stream);
    for (hipEvent_t e : c->ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev1) (void)hipEventDestroy(e);
    delete c;
}

int msm_cost_set_meshes(msm_cost *c, msm_mesh *target, msm_mesh *source, msm_mesh *cpgrid) {
    if (!c || !target || !source || !cpgrid) return fail(MSM_ERR_INVALID, "msm_cost_set_meshes: null argument");
    if (target->ctx != c->ctx || source->ctx != c->ctx || cpgrid->ctx != c->ctx) return fail(MSM_ERR_INVALID, "meshes belong to another context");
    if (source->V < cpgrid->V) return fail(MSM_ERR_INVALID, "source mesh has fewer vertices than the control grid");
    c->target = target;
    c->source = source;
    c->cpgrid = cpgrid;
    c->cp_conn_valid = false;
    c->orig_xyz = source->xyz;
    c->ocp_xyz = cpgrid->xyz;
    MSM_TRY(c->d_orig.upload(c->orig_xyz.data(), c->orig_xyz.size(), c->ctx));
    MSM_TRY(c->d_ocp.upload(c->ocp_xyz.data(), c->ocp_xyz.size(), c->ctx));
    MSM_TRY(ctx_sync(c->ctx));
    c->have_source = false;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_reset_source(msm_cost *c, msm_mesh *source) {
    if (!c || !source) return fail(MSM_ERR_INVALID, "msm_cost_reset_source: null argument");
    if (source->ctx != c->ctx) return fail(MSM_ERR_INVALID, "msm_cost_reset_source: the mesh belongs to another context");
    if (c->source && source->V != c->source->V) return fail(MSM_ERR_INVALID, "source mesh size changed");
    c->source = source;
    c->have_source = false;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_reset_cpgrid(msm_cost *c, msm_mesh *cpgrid) {
    if (!c || !cpgrid) return fail(MSM_ERR_INVALID, "msm_cost_reset_cpgrid: null argument");
    if (cpgrid->ctx != c->ctx) return fail(MSM_ERR_INVALID, "msm_cost_reset_cpgrid: the mesh belongs to another context");
    if (c->cpgrid && cpgrid->V != c->cpgrid->V) return fail(MSM_ERR_INVALID, "control grid size changed");
    c->cpgrid = cpgrid;
    c->cp_conn_valid = false;
    c->have_source = false;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_set_source_features(msm_cost *c, const double *feat, int32_t D) {
    if (!c || !feat || D <= 0) return fail(MSM_ERR_INVALID, "msm_cost_set_source_features: bad arguments");
    int st = need(c, c->source != nullptr, "meshes");
    if (st) return st;
    // the univariate classes read feature row 1 only (M/DiscreteCostFunction.cpp:343-347, :371, :477-481): with D > 1 their
    // kernels index row 0 of this D x V array and column 0 of the target's vertex-major features
    c->D = D;
    MSM_TRY(ctx_sync(c->ctx));
    c->sfeat.assign(feat, feat + (size_t)D * c->source->V);
    c->vm_valid = false;
    MSM_HIP(c->d_sfeat.ensure(c->sfeat.size()));
    {
        const int st = upload_staged(c->ctx, c->d_sfeat.p, c->sfeat.data(), sizeof(double) * c->sfeat.size());  // D x V: megabytes
        if (st) return st;
        MSM_TRY(ctx_sync(c->ctx));
    }
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_set_cfweight(msm_cost *c, const double *w, int32_t rows) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    int st = need(c, c->source != nullptr, "meshes");
    if (st) return st;
    MSM_TRY(ctx_sync(c->ctx));
    if (!w) {
        c->cfw.clear();
        c->cfw_rows = 0;
        c->vm_valid = false;
    } else {
        if (rows <= 0) return fail(MSM_ERR_INVALID, "msm_cost_set_cfweight: rows must be positive");
        // initialize(), M/DiscreteCostFunction.cpp:114-115
        if (rows != 1 && c->D > 0 && rows != c->D)
            return fail(MSM_ERR_INVALID, "DiscreteModel ERROR:: costfunction weighting has dimensions incompatible with data");
        c->cfw.assign(w, w + (size_t)rows * c->source->V);
        c->cfw_rows = rows;
        c->vm_valid = false;
        MSM_HIP(c->d_cfw.ensure(c->cfw.size()));
        st = upload_staged(c->ctx, c->d_cfw.p, c->cfw.data(), sizeof(double) * c->cfw.size());
        if (st) return st;
        MSM_TRY(ctx_sync(c->ctx));
    }
    c->have_source = false;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_set_spacings(msm_cost *c, const double *maxsep, double mvdmax) {
    if (!c || !maxsep) return fail(MSM_ERR_INVALID, "msm_cost_set_spacings: null argument");
    int st = need(c, c->cpgrid != nullptr, "meshes");
    if (st) return st;
    c->maxsep.assign(maxsep, maxsep + c->cpgrid->V);
    c->mvdmax = mvdmax;
    c->have_source = false;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_set_labels(msm_cost *c, const double *labels, int32_t L, const double *rot) {
    if (!c || !labels || !rot || L <= 0) return fail(MSM_ERR_INVALID, "msm_cost_set_labels: bad arguments");
    int st = need(c, c->cpgrid != nullptr, "meshes");
    if (st) return st;
    MSM_TRY(ctx_sync(c->ctx));
    c->L = L;
    c->fix_off_valid = false;
    c->labels.assign(labels, labels + 3 * (size_t)L);
    c->rot.assign(rot, rot + 9 * (size_t)c->cpgrid->V);
    MSM_TRY(c->d_labels.upload(c->labels.data(), c->labels.size(), c->ctx));
    MSM_TRY(c->d_rot.upload(c->rot.data(), c->rot.size(), c->ctx));
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_set_triplets(msm_cost *c, const int32_t *triplets, int32_t T) {
    if (!c || (!triplets && T > 0) || T < 0) return fail(MSM_ERR_INVALID, "msm_cost_set_triplets: bad arguments");
    int st = need(c, c->cpgrid != nullptr, "meshes");
    if (st) return st;
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i)
        if (triplets[i] < 0 || triplets[i] >= c->cpgrid->V) return fail(MSM_ERR_INVALID, "triplet node id out of range");
    MSM_TRY(ctx_sync(c->ctx));
    c->triplets.assign(triplets, triplets + 3 * (size_t)T);
    c->move_valid = false;
    if (T > 0) MSM_TRY(c->d_triplets.upload(c->triplets.data(), c->triplets.size(), c->ctx));
    return MSM_OK;
}

int msm_cost_set_pairs(msm_cost *c, const int32_t *pairs, int32_t P) {
    if (!c || (!pairs && P > 0) || P < 0) return fail(MSM_ERR_INVALID, "msm_cost_set_pairs: bad arguments");
    int st = need(c, c->cpgrid != nullptr, "meshes");
    if (st) return st;
    for (int64_t i = 0; i < 2 * (int64_t)P; ++i)
        if (pairs[i] < 0 || pairs[i] >= c->cpgrid->V) return fail(MSM_ERR_INVALID, "pair node id out of range");
    MSM_TRY(ctx_sync(c->ctx));
    c->pairs.assign(pairs, pairs + 2 * (size_t)P);
    if (P > 0) MSM_TRY(c->d_pairs.upload(c->pairs.data(), c->pairs.size(), c->ctx));
    return MSM_OK;
}

int msm_cost_get_source_data(msm_cost *c) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    (void)drop_pending_move(c);
    int st = need(c, c->target && c->source && c->cpgrid, "meshes");
    if (st) return st;
    st = need(c, c->D > 0, "source features");
    if (st) return st;
    st = need(c, (int)c->maxsep.size() == c->cpgrid->V, "spacings");
    if (st) return st;
    // NonLinearSRegDiscreteCostFunction::initialize, M/DiscreteCostFunction.cpp:109-117
    if (c->target->V == 0 || c->source->V == 0) return fail(MSM_ERR_STATE, "CostFunction::You must supply source and target meshes.");
    const bool timing = std::getenv("MSMHIP_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  get_source_data: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    st = is_ho(c) ? patches_by_triangle(c) : patches_by_range(c);
    if (st) return st;
    lap("patches");
    c->pmax = 1;
    for (int g = 0; g < c->ngroups; ++g) c->pmax = std::max(c->pmax, c->pptr[g + 1] - c->pptr[g]);
    st = resample_weights(c);
    if (st) return st;
    lap("resample_weights");
    msm_ctx *ctx = c->ctx;
    MSM_TRY(c->d_pptr.upload(c->pptr.data(), c->pptr.size(), ctx));
    if (is_ho(c) || std::getenv("MSMHIP_NO_PATCH_SORT")) {
        MSM_TRY(c->d_pidx.upload_vec(c->pidx, ctx));
    } else {
        // Device-side order of the points of each patch: Morton order of their positions (launch_sort_patches), so that the
        // lanes of a wavefront sample neighbouring places of the target (neighbouring direction cells and triangle records
        // share cache lines).  The similarity does not depend on the order of the patch points; the API keeps reporting
        // ascending ids.
        if (!c->pidx_asc_on_device) MSM_TRY(c->d_pidx_asc.upload_vec(c->pidx, ctx));
        MSM_HIP(c->d_pidx.ensure(std::max<size_t>(c->pidx.size(), 1)));
        MSM_HIP(c->d_code.ensure(c->source->V));
        st = launch_sort_patches(ctx, c->source->d_xyz, c->source->V, c->d_pptr.p, c->ngroups, c->d_pidx_asc.p, c->d_code.p, c->d_pidx.p);
        if (st) return st;
    }
    MSM_TRY(c->d_absw.upload(c->absw.data(), c->absw.size(), ctx));
    {
        // Launch order of the control points in the unary kernels: Morton order of their positions, so that the
        // contiguous share of each XCD (kernels map blockIdx % 8 to a range of this list) is one region of the sphere
        // and that XCD's L2 only has to hold the matching part of the target.
        const int N = c->cpgrid->V;
        const double *cp = c->cpgrid->xyz.data();
        auto spread = [](uint32_t v) {  // 10 bits -> every third bit
            v &= 0x3ff;
            v = (v | (v << 16)) & 0x030000ff;
            v = (v | (v << 8)) & 0x0300f00f;
            v = (v | (v << 4)) & 0x030c30c3;
            v = (v | (v << 2)) & 0x09249249;
            return v;
        };
        std::vector<std::pair<uint32_t, int32_t>> key(N);
        for (int i = 0; i < N; ++i) {
            uint32_t q[3];
            for (int a = 0; a < 3; ++a) {
                const double u = (cp[(size_t)a * N + i] + kBounds) / (2 * kBounds);
                q[a] = (uint32_t)std::max(0.0, std::min(1023.0, u == u ? u * 1024.0 : 0.0));
            }
            key[i] = {spread(q[0]) << 2 | spread(q[1]) << 1 | spread(q[2]), i};
        }
        std::sort(key.begin(), key.end());
        c->order.resize(N);
        for (int i = 0; i < N; ++i) c->order[i] = key[i].second;
        MSM_TRY(c->d_order.upload(c->order.data(), c->order.size(), ctx));
        c->fix_off_valid = false;
    }
    MSM_TRY(ctx_sync(ctx));
    lap("orders + uploads");
    c->have_source = true;
    invalidate_table(c);
    return MSM_OK;
}

int msm_cost_patches(msm_cost *c, int32_t *ngroups, int32_t *ptr, int32_t *idx, int64_t cap) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    int st = need(c, c->have_source, "get_source_data()");
    if (st) return st;
    if (ngroups) *ngroups = c->ngroups;
    if (ptr) std::copy(c->pptr.begin(), c->pptr.end(), ptr);
    if (idx) {
        if ((int64_t)c->pidx.size() > cap) return fail(MSM_ERR_CAPACITY, "patch index buffer too small");
        std::copy(c->pidx.begin(), c->pidx.end(), idx);
    }
    return MSM_OK;
}

int msm_cost_absolute_weights(msm_cost *c, double *absw) {
    if (!c || !absw) return fail(MSM_ERR_INVALID, "msm_cost_absolute_weights: null argument");
    int st = need(c, c->have_source, "get_source_data()");
    if (st) return st;
    std::copy(c->absw.begin(), c->absw.end(), absw);
    return MSM_OK;
}

int msm_cost_unary_table_async(msm_cost *c) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    MSM_TRY(drop_ctx_pending(c->ctx));  // a label step queued ahead on this context shares the stream, the status word and the flags with this call
    // every call is a fresh computeUnaryCosts(): the per (control point, label) rotations are recomputed too, as they are in
    // every iteration of a registration (new labels / a moved control grid each time)
    c->table_valid = false;
    c->rotations_valid = false;
    c->h_U.clear();
    return ensure_unary_table(c);
}

int msm_cost_unary_table_fetch(msm_cost *c, double *U) {
    if (!c || !U) return fail(MSM_ERR_INVALID, "msm_cost_unary_table_fetch: null argument");
    MSM_TRY(drop_ctx_pending(c->ctx));
    int st = need(c, c->table_valid, "msm_cost_unary_table_async()");
    if (st) return st;
    msm_ctx *ctx = c->ctx;
    const size_t n = (size_t)c->L * c->cpgrid->V;
    void *mapped = ctx_mapped(ctx, U, sizeof(double) * n);
    if (mapped && reinterpret_cast<uintptr_t>(mapped) % 16 == 0) {
        // U lies in a msm_host_alloc block: a copy KERNEL writes it there (and the status word into the mapped flag) -- a copy-engine
        // command of this size costs 15 us of latency plus a second one for the status
        st = ctx_flag(ctx);
        if (st) return st;
        st = launch_copy_to_mapped(ctx, c->d_U.p, static_cast<double *>(mapped), n, ctx->d_flag_map);
        if (st) return st;
        MSM_TRY(ctx_sync(ctx));
        volatile int *flags = ctx->h_flag;
        if (flags[0] != 0) {
            flags[0] = 0;
            return check_status(ctx, "computeUnaryCosts");
        }
        return MSM_OK;
    }
    MSM_TRY(c->d_U.download(U, n, ctx));
    return check_status(ctx, "computeUnaryCosts");
}

int msm_cost_unary_table(msm_cost *c, double *U) {
    int st = msm_cost_unary_table_async(c);
    if (st) return st;
    return msm_cost_unary_table_fetch(c, U);
}

int msm_cost_unary_batch(msm_cost *c, const int32_t *nodes, const int32_t *labels, int32_t n, double *out) {
    if (!c || !nodes || !labels || !out || n < 0) return fail(MSM_ERR_INVALID, "msm_cost_unary_batch: bad arguments");
    MSM_TRY(drop_ctx_pending(c->ctx));
    // computeUnaryCost(node,label) is a pure function between two get_source_data()/set_labels() calls,
    // so the batch is served from the table (computed once)
    if (!c->table_valid || c->h_U.size() != (size_t)c->L * c->cpgrid->V) {
        int st = ensure_unary_table(c);
        if (st) return st;
        c->h_U.resize((size_t)c->L * c->cpgrid->V);
        st = msm_cost_unary_table_fetch(c, c->h_U.data());
        if (st) return st;
    }
    const int N = c->cpgrid->V;
    for (int i = 0; i < n; ++i) {
        if (nodes[i] < 0 || nodes[i] >= N || labels[i] < 0 || labels[i] >= c->L) return fail(MSM_ERR_INVALID, "unary query %d out of range", i);
        out[i] = c->h_U[(size_t)labels[i] * N + nodes[i]];
    }
    return MSM_OK;
}

int msm_cost_enable_timing(msm_cost *c, int enable) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    if (enable && c->ev0.empty()) {
        c->ev0.resize(64);
        c->ev1.resize(64);
        for (size_t i = 0; i < c->ev0.size(); ++i) {
            MSM_HIP(hipEventCreate(&c->ev0[i]));
            MSM_HIP(hipEventCreate(&c->ev1[i]));
        }
    }
    c->timing = enable != 0;
    c->ev_next = 0;
    c->ev_count = 0;
    return MSM_OK;
}

int msm_cost_kernel_times(msm_cost *c, double *ms, int32_t cap, int32_t *n) {
    if (!c || !ms || !n) return fail(MSM_ERR_INVALID, "msm_cost_kernel_times: null argument");
    MSM_TRY(drop_ctx_pending(c->ctx));
    MSM_TRY(ctx_sync(c->ctx));
    const int total = (int)c->ev0.size(), cnt = std::min(c->ev_count, (int)cap);
    for (int k = 0; k < cnt; ++k) {
        const int slot = ((c->ev_next - cnt + k) % total + total) % total;
        float f = 0.f;
        MSM_HIP(hipEventElapsedTime(&f, c->ev0[slot], c->ev1[slot]));
        ms[k] = f;
    }
    *n = cnt;
    return MSM_OK;
}

int msm_cost_counters(msm_cost *c, int64_t counters[4]) {
    if (!c || !counters) return fail(MSM_ERR_INVALID, "msm_cost_counters: null argument");
    counters[0] = c->counters[0];  // point samples (counted on the host: L x patch points per table)
    counters[1] = c->counters[1];
    counters[2] = c->counters[2];
    counters[3] = c->counters[3];
    return MSM_OK;
}

// ---- clique costs: implemented in cost_cliques.cpp ----

}  // extern "C"
