// cost_cliques.cpp -- C ABI of the pairwise / triplet clique costs and of evaluateTotalCostSum
// (include/msmhip.h).  Kernels: clique_kernels.hip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "cost_internal.hpp"

using namespace msm;

namespace msm {
// A label step queued by msm_cost_triplet_octets_prefetch that nobody took: wait for its kernel and discard what it may have raised (its evaluations
// were never asked for).  Called by every entry point of a cost function except the msm_cost_triplet_octets call that matches the prefetch.
int drop_pending_move(msm_cost *c) {
    if (!c->pending.valid) return MSM_OK;
    c->pending.valid = false;
    ++c->prefetch_drops;
    msm_ctx *ctx = c->ctx;
    if (ctx->pending_cost == c) ctx->pending_cost = nullptr;
    MSM_TRY(ctx_sync(ctx));
    volatile int *flags = ctx->h_flag;
    if (flags) {
        flags[1] = 0;
        if (flags[0] != 0) {
            flags[0] = 0;
            (void)check_status(ctx, "a dropped prefetch");  // clears the device status word
        }
    }
    return MSM_OK;
}

// The same from the context's side: whichever cost function holds the context's queued step.  Every entry point that synchronises the stream and reads the
// status word or the mapped flags calls this first (or drop_pending_move on its own object) -- a status raised by a speculative kernel must not be reported
// against another call, a tail request in flags[1] not be consumed by another cost function's move.
int drop_ctx_pending(msm_ctx *ctx) {
    msm_cost *c = ctx->pending_cost;
    if (!c) return MSM_OK;
    ctx->pending_cost = nullptr;
    return drop_pending_move(c);
}

}  // namespace msm

namespace {

// everything the clique kernels read; uploads the control grid's connectivity on first use
int clique_args(msm_cost *c, bool need_triplets, bool need_pairs, CliqueArgs &a) {
    if (int st = drop_pending_move(c)) return st;
    if (int st = drop_ctx_pending(c->ctx)) return st;  // (another cost function's, on the same context)
    if (!c->cpgrid || !c->source || !c->target) return fail(MSM_ERR_STATE, "msm_cost: meshes must be set first");
    if (need_triplets && c->triplets.empty()) return fail(MSM_ERR_STATE, "msm_cost: triplets must be set first");
    if (need_pairs && c->pairs.empty()) return fail(MSM_ERR_STATE, "msm_cost: pairs must be set first");
    int st = ensure_label_rotations(c);
    if (st) return st;
    msm_ctx *ctx = c->ctx;
    msm_mesh *g = c->cpgrid;
    if (!c->cp_conn_valid) {
        const Adjacency &adj = mesh_adjacency(g);
        MSM_TRY(c->d_cp_tri.upload(g->tri.data(), g->tri.size(), ctx));
        MSM_TRY(c->d_cp_tid_ptr.upload(adj.tid_ptr.data(), adj.tid_ptr.size(), ctx));
        MSM_TRY(c->d_cp_tid.upload(adj.tid.data(), adj.tid.size(), ctx));
        MSM_TRY(ctx_sync(ctx));
        c->cp_conn_valid = true;
    }
    a.kind = c->p.kind;
    a.simmeasure = c->p.simmeasure;
    a.N = g->V;
    a.L = c->L;
    a.T = (int)(c->triplets.size() / 3);
    a.P = (int)(c->pairs.size() / 2);
    a.triplets = c->d_triplets.p;
    a.pairs = c->d_pairs.p;
    a.cp = g->d_xyz;
    a.ocp = c->d_ocp.p;
    a.orig = c->d_orig.p;
    a.Norig = (int)(c->orig_xyz.size() / 3);
    a.moved = c->d_moved.p;
    a.rnl = c->d_rnl.p;
    a.cp_tri = c->d_cp_tri.p;
    a.Tc = g->T;
    a.cp_tid_ptr = c->d_cp_tid_ptr.p;
    a.cp_tid = c->d_cp_tid.p;
    a.lambda = c->p.lambda;
    a.mu = c->p.mu;
    a.kappa = c->p.kappa;
    a.k_exp = c->p.k_exp;
    a.rexp = c->p.rexp;
    a.mvdmax = c->mvdmax;
    a.percentile = c->p.percentile;
    a.tfeat = c->target->d_feat;
    a.D = c->D;
    a.src = c->source->d_xyz;
    a.Nsrc = c->source->V;
    a.sfeat = c->d_sfeat.p;
    a.cfw = c->cfw.empty() ? nullptr : c->d_cfw.p;
    a.cfw_rows = c->cfw_rows;
    a.sfeat_vm = a.cfw_vm = nullptr;
    a.bin_ptr = c->d_pptr.p;
    a.bin_idx = c->d_pidx.p;
    a.bin_cap = std::max(c->pmax, 1);
    a.ho_vals = nullptr;
    a.ho_big = nullptr;
    a.ho_pending = nullptr;
    a.ho_count = nullptr;
    a.absw = c->d_absw.p;
    a.status = ctx->d_status;
    if (need_triplets && cost_is_ho(c)) {
        if (!c->have_source) return fail(MSM_ERR_STATE, "msm_cost: get_source_data() must be called first");
        if (!c->target->d_feat || c->target->D != c->D) return fail(MSM_ERR_STATE, "msm_cost: target features must match the source features");
        // the triclique likelihood is evaluated by hundreds of fusion moves per level and its two kernel families sum in
        // different orders, so this path waits for the direction table instead of switching to it mid-run
        st = ensure_rays(c->target, true);  // includes the sub-cell masks the group search uses
        if (st) return st;
        if (c->p.kind == MSM_COST_HO_MULTIVARIATE) {
            st = ensure_vertex_major(c);
            if (st) return st;
            a.sfeat_vm = c->d_sfeat_vm.p;
            a.cfw_vm = c->cfw.empty() ? nullptr : c->d_cfw_vm.p;
        }
        if (c->target->tree.ray_G > 0 && c->pmax <= 1024 && a.T < (1 << 18)) {  // fusion moves: sample -> fix up -> reduce
            const size_t nv = 8 * c->pidx.size();
            MSM_HIP(c->d_ho_vals.ensure(std::max<size_t>(nv, 1)));
            MSM_HIP(c->d_ho_pending.ensure(std::max<size_t>(nv, 1)));
            if (!c->d_ho_count.p) MSM_HIP(c->d_ho_count.zero(1, ctx->stream));
            a.ho_vals = c->d_ho_vals.p;
            a.ho_pending = c->d_ho_pending.p;
            a.ho_count = c->d_ho_count.p;
        }
        if (c->pmax > kHoBinMax) {
            // a bin that outgrows a workgroup's LDS (an ico7 data mesh under an ico0 control grid; the reference pushes into a std::vector,
            // M/DiscreteCostFunction.cpp:468-485, and has no limit): the sampled values of an evaluation live in HBM, a slice per workgroup
            MSM_HIP(c->d_ho_big.ensure((size_t)kHoBigBlocks * (size_t)c->pmax));
            a.ho_big = c->d_ho_big.p;
        }
    }
    a.tree = dev_tree(c->target);
    a.rmode = c->p.rmode;
    a.atree = DevTree{};
    a.atarget = a.asrc = a.aw_val = nullptr;
    a.asrc_tri = a.aw_ptr = a.aw_cp = a.af_ptr = a.af_idx = nullptr;
    a.Va = a.Vs = a.Ts = 0;
    if (need_triplets && (c->p.rmode == 4 || c->p.rmode == 5)) {
        if (!c->have_anat) return fail(MSM_ERR_STATE, "MeshREG ERROR:: regoption 5 requires anatomical meshes (msm_cost_set_anatomical)");  // M/mesh_registration.cpp:103
        st = ensure_tree(c->asphere);
        if (st) return st;
        a.atree = dev_tree(c->asphere);
        a.atarget = c->d_atarget.p;
        a.Va = c->asphere->V;
        a.asrc = c->d_asrc.p;
        a.Vs = c->aVs;
        a.asrc_tri = c->d_asrc_tri.p;
        a.Ts = c->aTs;
        a.aw_ptr = c->d_aw_ptr.p;
        a.aw_cp = c->d_aw_cp.p;
        a.aw_val = c->d_aw_val.p;
        a.af_ptr = c->d_af_ptr.p;
        a.af_idx = c->d_af_idx.p;
    } else if (need_triplets && c->p.rmode != 2 && c->p.rmode != 3) {
        return fail(MSM_ERR_INVALID, "DiscreteModel computeTripletCost regoption does not exist");  // M/DiscreteCostFunction.cpp:184
    }
    return MSM_OK;
}

// ---- fused fusion move of the HO classes (move_kernels.hip) ----
// bin slots and control triangles per workgroup (at most 64 and 8: the kernel's LDS holds 64 proposed triangles)
static int env_int(const char *name, int dflt, int lo, int hi) {
    const char *e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return v < lo ? lo : (v > hi ? hi : v);
}
static const int kMoveSlots = env_int("MSMHIP_MOVE_SLOTS", 64, 8, 64), kMoveTriangles = env_int("MSMHIP_MOVE_TRIANGLES", 8, 1, 8);

bool fused_move_applies(const msm_cost *c, const CliqueArgs &a) {
    static const bool split = [] {
        const char *e = std::getenv("MSMHIP_MOVE");  // "split": the three-kernel path of round 1 (kept for comparison)
        return e && std::strcmp(e, "split") == 0;
    }();
    return !split && cost_is_ho(c) && a.ho_vals && a.tree.simple && a.tree.ray_G > 0 && a.rmode != 4 && a.rmode != 5 && c->pmax <= 128 && a.T > 0;
}

// per get_source_data(): the workgroups' runs of control triangles and the label-independent part of every bin point
int ensure_move(msm_cost *c, const CliqueArgs &a) {
    if (c->move_valid) return MSM_OK;
    msm_ctx *ctx = c->ctx;
    const int T = a.T;
    std::vector<int4> blk;
    // with many features per sample (the eight-lanes-per-sample passes of k_ho_move<., 2>) smaller workgroups do better: 88 against
    // 95 us at D = 32 with 48 slots / 6 triangles; with one feature the two shapes are level in the kernel and the larger is cheaper to launch
    const bool wide = a.kind == MSM_COST_HO_MULTIVARIATE && a.D >= 12;
    const bool env_s = std::getenv("MSMHIP_MOVE_SLOTS") != nullptr, env_t = std::getenv("MSMHIP_MOVE_TRIANGLES") != nullptr;
    int max_slots = (wide && !env_s) ? 48 : kMoveSlots, max_tris = (wide && !env_t) ? 6 : kMoveTriangles;
    if (!env_s && !env_t) {
        // A workgroup's time is a chain (proposed triangles, sampling rounds, similarity passes, last phase) whose length grows with the
        // triangles it holds, and the whole grid is resident at once up to ~1000 workgroups: on the coarser control grids of a registration's
        // first levels fewer triangles per workgroup shorten every chain at no cost (D = 32: 55 -> 41 us at ico3, 55 -> 34 us at ico2; the
        // ico4 grid keeps 6 / 8).  The smallest run of triangles that still fits the grid into one residency round:
        const int most = max_tris;
        for (int t : {1, 2, 4, 6, 8})
            if ((T + most - 1) / most <= 512 && t <= most && (T + t - 1) / t <= 1024) {  // only when the usual run leaves half the GPU empty
                max_tris = t;
                max_slots = 8 * t;
                break;
            }
    }
    int slots = 0, nt = 0, cap = max_slots, first = 0;
    auto close = [&](int t_end) {
        if (nt > 0) blk.push_back(make_int4(first, nt, c->pptr[first], c->pptr[t_end] - c->pptr[first]));
        first = t_end;
        slots = nt = 0;
    };
    for (int t = 0; t < T; ++t) {
        const int n = c->pptr[t + 1] - c->pptr[t];
        if (nt > 0 && (slots + n > max_slots || nt == max_tris)) close(t);
        slots += n;
        ++nt;
        cap = std::max(cap, slots);
    }
    close(T);
    const size_t ns = std::max<size_t>(c->pidx.size(), 1);
    MSM_TRY(c->d_blk.upload(blk.data(), blk.size(), ctx));
    MSM_HIP(c->d_tri_frame.ensure(5 * (size_t)T));
    MSM_HIP(c->d_tri_stat.ensure(3 * (size_t)T));
    MSM_HIP(c->d_slot_wda.ensure(ns));
    MSM_HIP(c->d_slot_tri.ensure(ns));
    MSM_HIP(c->d_slot_w.ensure(3 * ns));
    MSM_HIP(c->d_slot_sf.ensure(ns));
    if (a.cfw) MSM_HIP(c->d_slot_cw.ensure(ns));
    MSM_HIP(c->d_defer_list.ensure((size_t)8 * T));
    if (!c->d_defer_cnt.p) MSM_HIP(c->d_defer_cnt.zero(2, ctx->stream));
    int st = launch_move_prepare(ctx, a, (int)c->pidx.size(), c->d_slot_tri.p, c->d_slot_w.p, c->d_slot_sf.p, a.cfw ? c->d_slot_cw.p : nullptr, c->d_slot_wda.p,
                                 c->d_tri_frame.p, c->d_tri_stat.p);
    if (st) return st;
    MSM_TRY(ctx_sync(ctx));  // blk is a local
    c->move_nblk = (int)blk.size();
    c->move_cap = cap;
    c->move_valid = true;
    return MSM_OK;
}

int upload_ints(msm_ctx *ctx, DevBuf<int> &buf, const int32_t *host, size_t n) {
    MSM_TRY(buf.upload(host, n, ctx));
    return MSM_OK;
}

}  // namespace

extern "C" {

int msm_cost_set_anatomical(msm_cost *c, msm_mesh *sphere, const double *atarget_xyz, const double *asource_xyz, int32_t Vs,
                            const int32_t *asource_tri, int32_t Ts, const int32_t *w_ptr, const int32_t *w_cp, const double *w_val,
                            const int32_t *face_ptr, const int32_t *face_idx) {
    if (!c || !sphere || !atarget_xyz || !asource_xyz || !asource_tri || !w_ptr || !w_cp || !w_val || !face_ptr || !face_idx || Vs <= 0 || Ts <= 0)
        return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: bad arguments");
    const int T = (int)(c->triplets.size() / 3);
    if (T == 0 || !c->cpgrid) return fail(MSM_ERR_STATE, "msm_cost_set_anatomical: set_meshes and set_triplets first");
    const int N = c->cpgrid->V;
    if (w_ptr[0] != 0 || face_ptr[0] != 0) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: CSR rows must start at 0");
    for (int v = 0; v < Vs; ++v)
        if (w_ptr[v + 1] < w_ptr[v]) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: weight rows are not monotone");
    for (int j = 0; j < w_ptr[Vs]; ++j)
        if (w_cp[j] < 0 || w_cp[j] >= N) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: control point id %d out of range", w_cp[j]);
    for (int v = 0; v < Vs; ++v)
        for (int j = w_ptr[v] + 1; j < w_ptr[v + 1]; ++j)
            if (w_cp[j] <= w_cp[j - 1]) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: control point ids must ascend within a row (std::map order)");
    for (int t = 0; t < T; ++t)
        if (face_ptr[t + 1] < face_ptr[t]) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: face rows are not monotone");
    for (int j = 0; j < face_ptr[T]; ++j)
        if (face_idx[j] < 0 || face_idx[j] >= Ts) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: face id %d out of range", face_idx[j]);
    for (int j = 0; j < 3 * Ts; ++j)
        if (asource_tri[j] < 0 || asource_tri[j] >= Vs) return fail(MSM_ERR_INVALID, "msm_cost_set_anatomical: vertex id %d out of range", asource_tri[j]);
    msm_ctx *ctx = c->ctx;
    MSM_HIP(hipSetDevice(ctx->device));
    MSM_TRY(c->d_atarget.upload(atarget_xyz, 3 * (size_t)sphere->V, ctx));
    MSM_TRY(c->d_asrc.upload(asource_xyz, 3 * (size_t)Vs, ctx));
    MSM_TRY(c->d_asrc_tri.upload(asource_tri, 3 * (size_t)Ts, ctx));
    MSM_TRY(c->d_aw_ptr.upload(w_ptr, (size_t)Vs + 1, ctx));
    MSM_TRY(c->d_aw_cp.upload(w_cp, std::max<size_t>(w_ptr[Vs], 1), ctx));
    MSM_TRY(c->d_aw_val.upload(w_val, std::max<size_t>(w_ptr[Vs], 1), ctx));
    MSM_TRY(c->d_af_ptr.upload(face_ptr, (size_t)T + 1, ctx));
    MSM_TRY(c->d_af_idx.upload(face_idx, std::max<size_t>(face_ptr[T], 1), ctx));
    MSM_TRY(ctx_sync(ctx));  // the caller's arrays may go away
    c->asphere = sphere;
    c->aVs = Vs;
    c->aTs = Ts;
    c->have_anat = true;
    return MSM_OK;
}

int msm_cost_triplet_batch(msm_cost *c, const int32_t *triplet, const int32_t *la, const int32_t *lb, const int32_t *lc, int32_t n, double *out) {
    if (!c || !triplet || !la || !lb || !lc || !out || n < 0) return fail(MSM_ERR_INVALID, "msm_cost_triplet_batch: bad arguments");
    if (n == 0) return MSM_OK;
    CliqueArgs a;
    int st = clique_args(c, true, false, a);
    if (st) return st;
    for (int i = 0; i < n; ++i)
        if (triplet[i] < 0 || triplet[i] >= a.T || la[i] < 0 || la[i] >= a.L || lb[i] < 0 || lb[i] >= a.L || lc[i] < 0 || lc[i] >= a.L)
            return fail(MSM_ERR_INVALID, "triplet query %d out of range", i);
    msm_ctx *ctx = c->ctx;
    DevBuf<int> qt, qa, qb, qc;
    DevBuf<double> dout;
    if ((st = upload_ints(ctx, qt, triplet, n)) || (st = upload_ints(ctx, qa, la, n)) || (st = upload_ints(ctx, qb, lb, n)) ||
        (st = upload_ints(ctx, qc, lc, n)))
        return st;
    MSM_HIP(dout.ensure(n));
    st = launch_triplet_batch(ctx, a, qt.p, qa.p, qb.p, qc.p, n, dout.p);
    if (st) return st;
    MSM_TRY(dout.download(out, n, ctx));
    c->counters[2] += n;
    return check_status(ctx, "computeTripletCost");
}

// The fused fusion move of the triclique classes (move_kernels.hip) for one labeling: all eight combinations of every control triangle
// (E: 8 x T, a label step of Fusion) or, single, combination 000 only (E: T values -- the triplet part of evaluateTotalCostSum,
// M/DiscreteCostFunction.cpp:55-77, which used to go through the general on-demand kernel at 143 us per call at ico4 / 32 features).
static int fused_move_finish(msm_cost *c, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *lab, bool staged_copy, bool direct, void *pin, size_t in_pad,
                             size_t out_bytes, double *out_dev, double *E);

// prefetch != nullptr: queue the move and return without waiting (requires the fast form of the call: labeling in the kernel arguments, costs written into
// the caller's mapped array); *prefetch tells whether it was queued
static int fused_move(msm_cost *c, const CliqueArgs &a, const int32_t *labeling, int32_t label, double *E, bool single, bool *prefetch = nullptr) {
    msm_ctx *ctx = c->ctx;
    int st;
    if (prefetch) *prefetch = false;
    const size_t in_bytes = sizeof(int32_t) * (size_t)a.N, out_bytes = sizeof(double) * (single ? 1 : 8) * (size_t)a.T, in_pad = (in_bytes + 255) & ~(size_t)255;
    // The call of the optimisers' inner loop (once per label step).  Two launches and one synchronisation: the labeling
    // rides in the kernel arguments, the costs are written into mapped pinned memory (the caller's own array when it
    // came from msm_host_alloc), a raised status shows up in a mapped flag.
    st = ensure_move(c, a);
    if (st) return st;
    st = ctx_flag(ctx);
    if (st) return st;
    // diagnostics: MSMHIP_MOVE_LABELS=device sends the labeling with a copy command, MSMHIP_MOVE_OUT=device brings the costs back with one
    static const bool labels_by_copy = [] { const char *e = std::getenv("MSMHIP_MOVE_LABELS"); return e && std::strcmp(e, "device") == 0; }();
    static const bool out_by_copy = [] { const char *e = std::getenv("MSMHIP_MOVE_OUT"); return e && std::strcmp(e, "device") == 0; }();
    const bool packed = a.N <= 4 * kMoveLabelWords && a.L <= 256 && !labels_by_copy;
    MoveLabels lab;
    if (packed) std::memset(lab.w, 0, sizeof(uint32_t) * (size_t)((a.N + 3) / 4));
    for (int i = 0; i < a.N; ++i) {
        if (labeling[i] < 0 || labeling[i] >= a.L) return fail(MSM_ERR_INVALID, "labeling[%d] out of range", i);
        if (packed) lab.w[i >> 2] |= (uint32_t)labeling[i] << ((i & 3) * 8);
    }
    void *pin = nullptr;
    double *out_dev = out_by_copy ? nullptr : (double *)ctx_mapped(ctx, E, out_bytes);
    const bool direct = out_dev != nullptr;
    if (prefetch && (!direct || !packed || single)) return MSM_OK;  // not the fast form: the hint is ignored, the call itself will do everything
    if (!direct || !packed) {
        st = ctx_io_pinned(ctx, in_pad + out_bytes, &pin);
        if (st) return st;
    }
    bool staged_copy = false;  // the costs come back with a copy command (the pinned block could not be mapped)
    if (!direct) {
        if (ctx->io_dev && !out_by_copy) {
            out_dev = (double *)((char *)ctx->io_dev + in_pad);
        } else {
            MSM_HIP(c->d_clique_out.ensure((size_t)8 * a.T));  // (also large enough for the T values of a single-combination call)
            out_dev = c->d_clique_out.p;
            staged_copy = true;
        }
    }
    if (!packed) {
        std::memcpy(pin, labeling, in_bytes);
        MSM_HIP(c->d_labeling.ensure(a.N));
        MSM_HIP(hipMemcpyAsync(c->d_labeling.p, pin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    MoveArgs m;
    m.slot_tri = c->d_slot_tri.p;
    m.slot_w = c->d_slot_w.p;
    m.slot_sf = c->d_slot_sf.p;
    m.slot_cw = a.cfw ? c->d_slot_cw.p : nullptr;
    m.slot_wda = c->d_slot_wda.p;
    m.tri_frame = c->d_tri_frame.p;
    m.tri_stat = c->d_tri_stat.p;
    m.blk = c->d_blk.p;
    m.nblk = c->move_nblk;
    m.cap = c->move_cap;
    m.labeling = packed ? nullptr : c->d_labeling.p;
    m.label = label;
    m.vals = c->d_ho_vals.p;
    m.defer_list = c->d_defer_list.p;
    m.defer_cnt = c->d_defer_cnt.p;
    m.parity = c->move_parity;
    c->move_parity ^= 1;
    m.out = out_dev;
    m.host_flags = ctx->d_flag_map;
    m.trace = nullptr;
    m.single = single ? 1 : 0;
#ifdef MSM_MOVE_TRACE
    static DevBuf<unsigned long long> trace_buf;
    const size_t trace_words = 8 * (size_t)(8 * ((c->move_nblk + 7) / 8)) + 8;  // 8 stamps per workgroup of the grid, then 4 counters (ray_open_reason)
    MSM_HIP(trace_buf.zero(trace_words, ctx->stream));
    m.trace = trace_buf.p;
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        e0 = c->ev0[c->ev_next];
        e1 = c->ev1[c->ev_next];
        c->ev_next = (c->ev_next + 1) % (int)c->ev0.size();
        c->ev_count = std::min(c->ev_count + 1, (int)c->ev0.size());
    }
    st = launch_move(ctx, a, m, packed ? &lab : nullptr, e0, e1);
    if (st) return st;
    if (staged_copy) MSM_HIP(hipMemcpyAsync((char *)pin + in_pad, out_dev, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    c->counters[2] += (int64_t)(single ? 1 : 8) * a.T;
    if (prefetch) {  // queued: msm_cost_triplet_octets(labeling, label, E) will wait for it
        c->pending.kind = 1;
        c->pending.label = label;
        c->pending.E = E;
        c->pending.epoch = ctx->epoch;
        ctx->pending_cost = c;
        c->pending.labeling.assign(labeling, labeling + a.N);
        c->pending.a = a;
        c->pending.m = m;
        c->pending.lab = lab;
        c->pending.valid = true;
        *prefetch = true;
        return MSM_OK;
    }
    MSM_TRY(ctx_sync(ctx));
#ifdef MSM_MOVE_TRACE
    if (const char *path = std::getenv("MSMHIP_MOVE_TRACE")) {
        std::vector<unsigned long long> h(trace_words);
        MSM_HIP(hipMemcpy(h.data(), trace_buf.p, sizeof(unsigned long long) * trace_words, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(path, "wb")) {
            std::fwrite(h.data(), sizeof(unsigned long long), trace_words, f);
            std::fclose(f);
        }
    }
#endif
    return fused_move_finish(c, a, m, packed ? &lab : nullptr, staged_copy, direct, pin, in_pad, out_bytes, out_dev, E);
}

// after the move's kernel has completed (the stream is synchronised): the tail kernel when the main one asked for it, a raised status, the staged copy
static int fused_move_finish(msm_cost *c, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *lab, bool staged_copy, bool direct, void *pin, size_t in_pad,
                             size_t out_bytes, double *out_dev, double *E) {
    msm_ctx *ctx = c->ctx;
    int st;
    volatile int *flags = ctx->h_flag;
    if (flags[1] != 0) {  // rare: some evaluations need the complete search (sibling leaves, nearest vertex)
        flags[1] = 0;
        st = launch_move_tail(ctx, a, m, lab);
        if (st) return st;
        if (staged_copy) MSM_HIP(hipMemcpyAsync((char *)pin + in_pad, out_dev, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        MSM_TRY(ctx_sync(ctx));
        c->move_tails++;
    }
    st = MSM_OK;
    if (flags[0] != 0) {
        flags[0] = 0;
        st = check_status(ctx, "computeTripletCost");
    }
    if (!direct) std::memcpy(E, (char *)pin + in_pad, out_bytes);
    return st;
}

// the strain-only label step in its fast form (labeling in the kernel arguments, costs into mapped memory): launch, and unless queued ahead, wait
static int packed_strain_move(msm_cost *c, const CliqueArgs &a, const int32_t *labeling, int32_t label, double *E, double *out_dev, bool direct, void *pin, size_t in_pad,
                              size_t out_bytes, bool queue_only) {
    msm_ctx *ctx = c->ctx;
    MoveLabels lab;
    std::memset(lab.w, 0, sizeof(uint32_t) * (size_t)((a.N + 3) / 4));
    for (int i = 0; i < a.N; ++i) lab.w[i >> 2] |= (uint32_t)labeling[i] << ((i & 3) * 8);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {  // msm_cost_enable_timing: events around this move's kernel, like the triclique move's
        e0 = c->ev0[c->ev_next];
        e1 = c->ev1[c->ev_next];
        c->ev_next = (c->ev_next + 1) % (int)c->ev0.size();
        c->ev_count = std::min(c->ev_count + 1, (int)c->ev0.size());
        MSM_HIP(hipEventRecord(e0, ctx->stream));
    }
    int st = launch_triplet_octets_packed(ctx, a, lab, label, out_dev, ctx->d_flag_map);
    if (st) return st;
    if (e1) MSM_HIP(hipEventRecord(e1, ctx->stream));
    c->counters[2] += (int64_t)8 * a.T;
    if (queue_only) {
        c->pending.kind = 2;
        c->pending.label = label;
        c->pending.E = E;
        c->pending.epoch = ctx->epoch;
        ctx->pending_cost = c;
        c->pending.labeling.assign(labeling, labeling + a.N);
        c->pending.valid = true;
        return MSM_OK;
    }
    MSM_TRY(ctx_sync(ctx));
    volatile int *flags = ctx->h_flag;
    st = MSM_OK;
    if (flags[0] != 0) {
        flags[0] = 0;
        st = check_status(ctx, "computeTripletCost");
    }
    if (!direct) std::memcpy(E, static_cast<char *>(pin) + in_pad, out_bytes);
    return st;
}

// the msm_cost_triplet_octets call a prefetch was queued for: wait for the kernel, then what follows a move's kernel in the synchronous call
static int take_pending_move(msm_cost *c, double *E) {
    msm_ctx *ctx = c->ctx;
    msm_cost::PendingMove &p = c->pending;
    p.valid = false;
    if (ctx->pending_cost == c) ctx->pending_cost = nullptr;
    ++c->prefetch_hits;
    MSM_TRY(ctx_sync(ctx));
    if (p.kind == 1) return fused_move_finish(c, p.a, p.m, &p.lab, false, true, nullptr, 0, 0, nullptr, E);
    volatile int *flags = ctx->h_flag;
    if (flags[0] != 0) {
        flags[0] = 0;
        return check_status(ctx, "computeTripletCost");
    }
    return MSM_OK;
}

static int triplet_octets_impl(msm_cost *c, const int32_t *labeling, int32_t label, double *E, bool *prefetch);

int msm_cost_triplet_octets(msm_cost *c, const int32_t *labeling, int32_t label, double *E) {
    if (!c || !labeling || !E) return fail(MSM_ERR_INVALID, "msm_cost_triplet_octets: null argument");
    if (c->pending.valid && c->pending.label == label && c->pending.E == E && c->pending.epoch == c->ctx->epoch && c->ctx->pending_cost == c && c->cpgrid && (int)c->pending.labeling.size() == c->cpgrid->V &&
        std::memcmp(c->pending.labeling.data(), labeling, sizeof(int32_t) * c->pending.labeling.size()) == 0)
        return take_pending_move(c, E);  // queued ahead by msm_cost_triplet_octets_prefetch: the kernel ran while the host solved
    return triplet_octets_impl(c, labeling, label, E, nullptr);
}

// A hint: queue the label step (labeling, label) into E without waiting -- the optimiser's host-side solve of the step before runs meanwhile, and most
// steps of a converging level leave the labeling as it was (three of four in the MSMAll schedule).  Honoured when E lies in msm_host_alloc memory and the
// step takes the one-kernel form (triclique fused move or strain-only packed move); silently ignored otherwise.  The results do not depend on it.
int msm_cost_triplet_octets_prefetch(msm_cost *c, const int32_t *labeling, int32_t label, double *E) {
    if (!c || !labeling || !E) return fail(MSM_ERR_INVALID, "msm_cost_triplet_octets_prefetch: null argument");
    bool queued = false;
    return triplet_octets_impl(c, labeling, label, E, &queued);
}

int msm_cost_prefetch_stats(msm_cost *c, int64_t *taken, int64_t *dropped) {
    if (!c) return fail(MSM_ERR_INVALID, "null cost");
    if (taken) *taken = c->prefetch_hits;
    if (dropped) *dropped = c->prefetch_drops;
    return MSM_OK;
}

static int triplet_octets_impl(msm_cost *c, const int32_t *labeling, int32_t label, double *E, bool *prefetch) {
    CliqueArgs a;
    int st = clique_args(c, true, false, a);  // (drops a queued step nobody took)
    if (st) return st;
    if (label < 0 || label >= a.L) return fail(MSM_ERR_INVALID, "label %d out of range", label);
    msm_ctx *ctx = c->ctx;
    const size_t in_bytes = sizeof(int32_t) * (size_t)a.N, out_bytes = sizeof(double) * 8 * (size_t)a.T, in_pad = (in_bytes + 255) & ~(size_t)255;
    if (fused_move_applies(c, a)) return fused_move(c, a, labeling, label, E, false, prefetch);  // the call of the optimisers' inner loop (once per label step)
    for (int i = 0; i < a.N; ++i)
        if (labeling[i] < 0 || labeling[i] >= a.L) return fail(MSM_ERR_INVALID, "labeling[%d] out of range", i);
    if (prefetch && !ctx_mapped(ctx, E, out_bytes)) return MSM_OK;  // the hint needs the caller's mapped array
    // labeling and energies travel through pinned memory (pageable copies of these sizes cost more than the kernels)
    void *pin = nullptr;
    st = ctx_io_pinned(ctx, in_pad + out_bytes, &pin);
    if (st) return st;
    const char *octets_env = std::getenv("MSMHIP_OCTETS");  // read per call: the tests run both ways in one process
    const bool octets_by_copy = octets_env && std::strcmp(octets_env, "copy") == 0;
    if (!cost_is_ho(c) && a.N <= 4 * kMoveLabelWords && a.L <= 256 && !octets_by_copy && ctx_flag(ctx) == MSM_OK) {
        // The strain-only label step (--regoption=3 without --triclique: BASELINE config 2) like the triclique move: the labeling rides in
        // the kernel arguments, the costs land in mapped pinned memory -- the caller's array when it came from msm_host_alloc --, a raised
        // status in the mapped flag: one launch + one synchronisation instead of two copy commands around the kernel and a status read-back
        // (57 -> 25 us per call at ico4; MSMHIP_OCTETS=copy: the old way)
        double *out_dev = static_cast<double *>(ctx_mapped(ctx, E, out_bytes));
        const bool direct = out_dev != nullptr;
        if (!direct && ctx->io_dev) out_dev = reinterpret_cast<double *>(static_cast<char *>(ctx->io_dev) + in_pad);
        if (out_dev) {
            st = packed_strain_move(c, a, labeling, label, E, out_dev, direct, pin, in_pad, out_bytes, prefetch != nullptr && direct);
            if (!st && prefetch && direct) *prefetch = true;
            if (prefetch && !direct) return MSM_OK;
            return st;
        }
    }
    if (prefetch) return MSM_OK;  // the general path is not queued ahead
    std::memcpy(pin, labeling, in_bytes);
    MSM_HIP(c->d_labeling.ensure(a.N));
    MSM_HIP(hipMemcpyAsync(c->d_labeling.p, pin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    MSM_HIP(c->d_clique_out.ensure((size_t)8 * a.T));
    hipEvent_t g0 = nullptr, g1 = nullptr;
    if (c->timing) {  // msm_cost_enable_timing: events around this move's kernels (the on-demand kernels: anatomical strain, large bins, general targets)
        g0 = c->ev0[c->ev_next];
        g1 = c->ev1[c->ev_next];
        c->ev_next = (c->ev_next + 1) % (int)c->ev0.size();
        c->ev_count = std::min(c->ev_count + 1, (int)c->ev0.size());
        MSM_HIP(hipEventRecord(g0, ctx->stream));
    }
    st = launch_triplet_octets(ctx, a, c->d_labeling.p, label, c->d_clique_out.p);
    if (st) return st;
    if (g1) MSM_HIP(hipEventRecord(g1, ctx->stream));
    MSM_HIP(hipMemcpyAsync((char *)pin + in_pad, c->d_clique_out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    c->counters[2] += (int64_t)8 * a.T;
    st = check_status(ctx, "computeTripletCost");  // synchronises
    std::memcpy(E, (char *)pin + in_pad, out_bytes);
    return st;
}

int msm_cost_pairwise_batch(msm_cost *c, const int32_t *pair, const int32_t *la, const int32_t *lb, int32_t n, double *out) {
    if (!c || !pair || !la || !lb || !out || n < 0) return fail(MSM_ERR_INVALID, "msm_cost_pairwise_batch: bad arguments");
    if (n == 0) return MSM_OK;
    CliqueArgs a;
    int st = clique_args(c, false, true, a);
    if (st) return st;
    for (int i = 0; i < n; ++i)
        if (pair[i] < 0 || pair[i] >= a.P || la[i] < 0 || la[i] >= a.L || lb[i] < 0 || lb[i] >= a.L) return fail(MSM_ERR_INVALID, "pairwise query %d out of range", i);
    msm_ctx *ctx = c->ctx;
    DevBuf<int> qp, qa, qb;
    DevBuf<double> dout;
    if ((st = upload_ints(ctx, qp, pair, n)) || (st = upload_ints(ctx, qa, la, n)) || (st = upload_ints(ctx, qb, lb, n))) return st;
    MSM_HIP(dout.ensure(n));
    st = launch_pairwise_batch(ctx, a, qp.p, qa.p, qb.p, n, dout.p);
    if (st) return st;
    MSM_TRY(dout.download(out, n, ctx));
    c->counters[3] += n;
    return check_status(ctx, "computePairwiseCost");
}

int msm_cost_pairwise_table(msm_cost *c, double *paircosts) {
    if (!c || !paircosts) return fail(MSM_ERR_INVALID, "msm_cost_pairwise_table: null argument");
    CliqueArgs a;
    int st = clique_args(c, false, true, a);
    if (st) return st;
    msm_ctx *ctx = c->ctx;
    const size_t total = (size_t)a.P * a.L * a.L;
    MSM_HIP(c->d_clique_out.ensure(total));
    st = launch_pairwise_table(ctx, a, c->d_clique_out.p);
    if (st) return st;
    MSM_TRY(c->d_clique_out.download(paircosts, total, ctx));
    c->counters[3] += (int64_t)total;
    return check_status(ctx, "computePairwiseCosts");
}

int msm_cost_triplet_table(msm_cost *c, int32_t t0, int32_t t1, double *tcosts) {
    if (!c || !tcosts) return fail(MSM_ERR_INVALID, "msm_cost_triplet_table: null argument");
    CliqueArgs a;
    int st = clique_args(c, true, false, a);
    if (st) return st;
    if (t0 < 0 || t1 < t0 || t1 > a.T) return fail(MSM_ERR_INVALID, "msm_cost_triplet_table: triplet range [%d, %d) out of [0, %d)", t0, t1, a.T);
    msm_ctx *ctx = c->ctx;
    const size_t total = (size_t)(t1 - t0) * a.L * a.L * a.L;
    if (total == 0) return MSM_OK;
    if (total > ((size_t)1 << 31)) return fail(MSM_ERR_CAPACITY, "msm_cost_triplet_table: %zu values in one call; ask for a smaller triplet range", total);
    MSM_HIP(c->d_clique_out.ensure(total));
    st = launch_triplet_table(ctx, a, t0, t1, c->d_clique_out.p);
    if (st) return st;
    MSM_TRY(c->d_clique_out.download(tcosts, total, ctx));
    c->counters[2] += (int64_t)total;
    return check_status(ctx, "computeTripletCosts");
}

// evaluateTotalCostSum, M/DiscreteCostFunction.cpp:55-77: the three sums run in the reference's serial order on
// the host over device-evaluated terms (N + P + T values), because the order defines the reported energy
int msm_cost_total(msm_cost *c, const int32_t *labeling, double *total, double parts[3]) {
    if (!c || !labeling || !total) return fail(MSM_ERR_INVALID, "msm_cost_total: null argument");
    if (!c->cpgrid) return fail(MSM_ERR_STATE, "msm_cost: meshes must be set first");
    const int N = c->cpgrid->V;
    double u = 0.0, pw = 0.0, tc = 0.0;
    if (!cost_is_ho(c)) {  // the HO classes' computeUnaryCost returns 0 (M/DiscreteCostFunction.h:249,258): a sum of N zeros
        std::vector<int32_t> nodes(N);
        std::vector<double> vals(N);
        for (int i = 0; i < N; ++i) nodes[i] = i;
        int st = msm_cost_unary_batch(c, nodes.data(), labeling, N, vals.data());
        if (st) return st;
        for (int i = 0; i < N; ++i) u += vals[i];
    }
    const int P = (int)(c->pairs.size() / 2), T = (int)(c->triplets.size() / 3);
    if (P > 0) {
        std::vector<int32_t> id(P), la(P), lb(P);
        std::vector<double> vals(P);
        for (int p = 0; p < P; ++p) {
            id[p] = p;
            la[p] = labeling[c->pairs[2 * p]];
            lb[p] = labeling[c->pairs[2 * p + 1]];
        }
        int st = msm_cost_pairwise_batch(c, id.data(), la.data(), lb.data(), P, vals.data());
        if (st) return st;
        for (int p = 0; p < P; ++p) pw += vals[p];
    }
    bool triplets_done = false;
    if (T > 0 && cost_is_ho(c)) {
        // combination 000 of a fusion move IS computeTripletCost(t, labeling[a], labeling[b], labeling[c]): the fused move kernel with one
        // combination per control triangle instead of the general on-demand kernel (143 -> about 20 us at ico4 / 32 features)
        CliqueArgs a;
        int st = clique_args(c, true, false, a);
        if (st) return st;
        if (fused_move_applies(c, a)) {
            std::vector<double> vals(T);
            st = fused_move(c, a, labeling, 0, vals.data(), true);
            if (st) return st;
            for (int t = 0; t < T; ++t) tc += vals[t];  // serial, in triplet order (:70-75)
            triplets_done = true;
        }
    }
    if (T > 0 && !triplets_done) {
        std::vector<int32_t> id(T), la(T), lb(T), lc(T);
        std::vector<double> vals(T);
        for (int t = 0; t < T; ++t) {
            id[t] = t;
            la[t] = labeling[c->triplets[3 * t]];
            lb[t] = labeling[c->triplets[3 * t + 1]];
            lc[t] = labeling[c->triplets[3 * t + 2]];
        }
        int st = msm_cost_triplet_batch(c, id.data(), la.data(), lb.data(), lc.data(), T, vals.data());
        if (st) return st;
        for (int t = 0; t < T; ++t) tc += vals[t];
    }
    if (parts) {
        parts[0] = u;
        parts[1] = pw;
        parts[2] = tc;
    }
    *total = u + pw + tc;
    return MSM_OK;
}

}  // extern "C"
