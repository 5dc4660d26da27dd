// group.cpp -- C ABI of the groupwise (gMSM) path (include/msmhip.h, "groupwise registration").
//
// DiscreteGroupModel::setupCostFunction (M/DiscreteGroupModel.cpp:163-196) is per-iteration set-up: closest
// control points between subjects, and for every (subject, label) a rigid rotation of the data mesh followed by
// an adaptive-barycentric resample of its features to the template.  All of it runs on the GPU (trees as a forest,
// queries, weight-list surgery, weighted sums, patch lists: DESIGN.md section 5.6); the resampled feature maps
// F[subject][label] (D x V_template) and the patch lists stay in HBM, where the pairwise kernel reads them
// (group_kernels.hip).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include "devbuf.hpp"
#include "host_parallel.hpp"
#include "kernels.hpp"

using namespace msm;

namespace msm {
int adaptive_weights(msm_mesh *in_mesh, msm_mesh *new_mesh, const double *excl, std::vector<int32_t> &row_ptr, std::vector<int32_t> &col,
                     std::vector<double> &val);
const Adjacency &mesh_adjacency(msm_mesh *m);
int query_host(msm_mesh *target, const double *q, int N, int *tri_id, int *vid, double *w, int mode, const char *what, const double *q_on_device);
bool mesh_tree_on_gpu(const msm_mesh *m);
}  // namespace msm

struct msm_group {
    msm_ctx *ctx = nullptr;
    std::atomic<int> patch_cap_hint{0};
    std::atomic<size_t> pidx_hint{0};
    // a uniform grid over the template's vertices for the range test of a set-up's patch lists (built by group_setup_pipeline, valid while it runs)
    DevBuf<int> d_rg_start, d_rg_cursor, d_rg_ids, d_rg_bad, d_rg_tmp;
    RangeGrid rgrid;
    bool rgrid_valid = false;  // the longest patch index list of a subject so far: the next one's is compacted before the look at its counts
    int patch_max = 0;  // largest patch of any subject (msm_group_finalize)
    int pair_lanes = 32;  // lanes per query of k_group_pairwise: 16 when nearly all patches fit a quarter wavefront's registers (msm_group_finalize)
    msm_group_params p{};
    int S = 0;
    msm_mesh *tmpl = nullptr;
    std::vector<double> mask;
    DevBuf<double> d_mask;
    int N = 0, Tc = 0;
    std::vector<int32_t> cp_tri;                 // 3 x Tc SoA
    std::vector<msm_mesh *> cpmesh;              // per subject (owned)
    std::vector<msm_mesh *> data;                // per subject (borrowed)
    std::vector<msm_mesh *> scratch;             // per subject: the rotated data mesh (owned)
    std::vector<std::vector<double>> feat;       // per subject D x V
    std::vector<std::vector<double>> orig;       // per subject 3 x N: _ORIG_MESHES coords of the control-point ids
    std::vector<char> have_orig;
    int D = 0, L = 0;
    std::vector<double> labels;                  // 3 x L SoA
    // setup products
    bool ready = false, common_ready = false;
    std::vector<char> have_subject;
    std::vector<int32_t> pairs, triplets;
    DevBuf<int32_t> d_pairs, d_triplets;
    std::vector<double> rot, moved;              // (S*N) x 9, (S*N) x L x 3
    DevBuf<double> d_moved, d_cp, d_orig;
    std::vector<std::vector<double>> spacing;
    std::vector<std::unique_ptr<DevBuf<double>>> F;  // S * L: windows into Fslab[subject]
    std::vector<std::unique_ptr<DevBuf<double>>> Fslab;  // per subject: its L resampled feature maps in one allocation (one hipMalloc instead of nineteen per subject
                                                         // and set-up, and large pages under the label steps' gathers)
    std::vector<std::unique_ptr<DevBuf<int32_t>>> pptr, pidx;  // per subject
    // per subject: the D values of every patch entry beside its id (GroupArgs::pval; built by msm_group_finalize, 50 MB per subject at ico6 / ico4, D = 2)
    std::vector<std::unique_ptr<DevBuf<double>>> pval;
    DevBuf<double *> d_pvalp;
    DevBuf<GroupPatchRef> d_patch_dir;  // GroupArgs::dir
    bool pval_ready = false;
    // A group with imported subjects (a rank of a sharded run) evaluates a SLICE of the pair list, which touches a fraction of the patches (an eighth of the
    // control points with eight ranks): the value copies are then written for the nodes of the slice only, when the slice is first asked for
    // (ensure_patch_values); pval_p0 / pval_p1: the slice they cover (the whole list after a set-up without imports)
    bool pval_deferred = false;
    int64_t pval_p0 = -1, pval_p1 = -1;
    DevBuf<int> d_node_flags;
    std::vector<std::vector<int32_t>> h_pptr, h_pidx;
    // subjects imported in a batch from device memory keep their row offsets on the device only (fetched when msm_group_patch asks): what
    // msm_group_finalize needs of them -- index count, largest patch, patches of at most kPairSmallPatch entries -- comes from the check kernel
    struct ImportStat {
        int64_t npidx = 0;
        int largest = 0, small = 0;
    };
    std::vector<ImportStat> imp_stat;
    DevBuf<const double *> d_Fp;
    DevBuf<const int *> d_pptrp, d_pidxp;
    DevBuf<int> d_query[4];   // index columns of a batch of evaluations (kept between calls)
    DevBuf<double> d_answer;
    // the lanes of the set-up: contexts with streams of their own and a scratch copy of the data mesh each, so that the per-label
    // pipelines of a subject (some eighty small kernels in a dependent chain each) run side by side
    struct Lane {
        msm_ctx *ctx = nullptr;
        msm_mesh *mesh = nullptr;
    };
    std::vector<Lane> lanes;
    // what the set-up of ONE subject needs before its per-label work can start: the L rotated copies of its data mesh, their trees
    // (a forest), its features.  Two of them: while the lanes work through subject i the main stream prepares subject i + 1.
    static constexpr int kStages = 3;  // of a pipeline: the preparing loop runs this many subjects ahead of the consuming one (run_setup_pipe)
    struct Stage {
        DevBuf<double> d_rot, d_feat, d_rot9;  // d_rot9: the vertices' rotation matrices when they come from the host (rotation_mode 1)
        std::vector<double> rot9;
        Forest forest;
        bool forest_ok = false;
    };
    // the per-label remainder of a subject's set-up with the label as the second grid dimension of every launch (stage_batch)
    struct Batch {
        msm_ctx *ctx = nullptr;  // a stream of its own: runs beside the main stream's preparation of the next subject
        DevBuf<int> fvid, rvid, roff, rfill, rkey, coff, cfill, ckey, row_ptr, col, tkey, scan_tmp, long_flag, open;
        DevBuf<double> fw, rw, oldA, newA, ta, rwt, cval, correction, val, tval;
        DevBuf<int2> info[2];
    };
    // One set-up pipeline: a main stream (rotations, forest of the next subject), a batch stream (the per-label work and the patch lists of the
    // current one) and everything they write before a subject's products.  Two pipelines work through alternate subjects side by side (round 4):
    // each is a chain of small dependent launches that leaves most of the GPU idle.
    struct Pipe {
        msm_ctx *main = nullptr;  // pipe 0: the group's context; pipe 1: a context (stream) of its own
        bool own_main = false;
        Stage stage[kStages];
        Batch batch;
        // scratch of subject_patches
        DevBuf<double> d_centres, d_sep;
        DevBuf<double4> d_chunkb;       // k_range: bounding balls of the template's vertices, 64 ids at a time
        DevBuf<int> d_scan_tmp;         // subject_patches: block sums of the row-offset scan
        DevBuf<uint32_t> d_slots;
        DevBuf<int> d_counts;
    };
    static constexpr int kMaxPipes = 4;
    Pipe pipe[kMaxPipes];
    std::mutex lanes_mu;  // the per-label lanes (the path of a subject whose forest could not be built) are shared by the pipelines
    DevBuf<double> d_move_out;           // msm_group_fusion_move: the step's 4 P + 8 T results before they go to the host
    std::vector<int32_t> pair_order;     // the pair list in processing order (control points along a space-filling curve)
    DevBuf<int> d_pair_order;            // ... restricted to the slice [order_p0, order_p1) last asked for
    DevBuf<int4> d_pair_order4;          // the slice's positions with their pairs' nodes (GroupArgs::move_order4): rebuilt when the slice or the pair list changed
    uint64_t pairs_gen = 0, order4_gen = ~0ull;  // pairs_gen: counts estimate_pairs runs (the list's second nodes move with the control grids)
    int64_t order_p0 = -1, order_p1 = -1;
    int order_S = 0, order_N = 0;        // the sizes pair_order was built for
    // msm_group_set_pair_layout: 0 = the reference's list order (subject A, control point, subject B); 1 = control point by control point along the
    // curve -- the list itself in what is otherwise only the processing order, so that a contiguous slice of it is a REGION of the sphere
    int pair_layout = 0;
    // msm_group_set_rotation_mode: who computes estimate_rotation_matrix(centre, vertex) for the data meshes' vertices in get_patch_data.  1 (default): the
    // host's libm, as the reference does -- the rotated meshes are then the reference's to the bit; 0: the device (acos / sincos of the GPU's math library)
    int rotation_mode = 1;
    DevBuf<int> d_pair_perm, d_pairs_tmp;  // layout 1: reference position of every list position (kept with pair_order), and the list as the search wrote it
    Forest cp_forest;                    // the search trees of the S control grids, built together (estimate_pairs)
    DevBuf<double> d_cp_soa, d_rot, d_spacing, d_labels3;  // control points by component; ROT per node; spacing per node; labels 3 x L
    DevBuf<int2> d_forest_info;
    int64_t npairs = 0;                  // N * S * (S - 1) / 2; the host copy of the list (pairs) is fetched when asked for
    bool moved_on_host = false;          // g->moved mirrors d_moved (fetched for the rare host decisions of subject_patches)
    // the (current, current) pair costs of the last label step and the labeling they were computed for (GroupArgs::move_e00)
    DevBuf<double> d_e00;
    DevBuf<int> d_prev_labeling;
    bool e00_valid = false;
    int64_t e00_p0 = -1, e00_p1 = -1;
    // the (proposed, proposed) pair costs per label of the slice [e11_p0, e11_p1): valid from the label's step in the first sweep of
    // Fusion until the next set-up (GroupArgs::move_e11)
    DevBuf<double> d_e11;
    std::vector<unsigned char> e11_have;
    int64_t e11_p0 = -1, e11_p1 = -1;
    void drop_kept() {
        e00_valid = false;
        std::fill(e11_have.begin(), e11_have.end(), (unsigned char)0);
    }
    std::vector<int64_t> order_chunk;    // the slice's order is cut into pieces by OUTPUT range: piece k holds the pairs [order_chunk[k], order_chunk[k+1]) of the slice
    // msm_group_time_moves: HIP events around the kernels of a label step on the context's stream
    hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
    bool timing = false, timed = false;
    hipStream_t copy_stream = nullptr;   // the finished pieces of a label step leave for the host while the next ones are computed
    std::vector<hipEvent_t> copy_events;
    DevBuf<double> d_rotated;  // the L rotated data meshes of the subject being set up (group_subject_setup, the comparison path)
};

namespace {

inline V3 pt(const double *xyz, int V, int i) { return mk(xyz[i], xyz[V + i], xyz[2 * V + i]); }

// patch lists of one subject: template vertices within range*spacing of each rotated control point
// (get_patch_data, M/DiscreteGroupModel.cpp:109-117), via the range kernel + host tie resolution (see k_range)
// F[s][0..L) as windows of `per` doubles each into the subject's slab
int subject_feature_slab(msm_group *g, int s, size_t per) {
    auto &slab = g->Fslab[s];
    if (!slab) slab.reset(new DevBuf<double>());
    const double *before = slab->p;
    MSM_HIP(slab->ensure(per * g->L));
    for (int l = 0; l < g->L; ++l) {
        auto &buf = g->F[(size_t)s * g->L + l];
        if (!buf) buf.reset(new DevBuf<double>());
        if (buf->p != slab->p + per * l || slab->p != before || buf->cap != per) buf->view(slab->p + per * l, per);
    }
    return MSM_OK;
}

// deferred: work queued on the stream before this call whose status has not been looked at yet (stage_batch) -- named in the error should it have failed;
// the first synchronisation here covers it
int subject_patches(msm_group *g, int s, msm_ctx *ctx = nullptr, msm_group::Pipe *pipe = nullptr, const char *deferred = nullptr) {  // ctx: the context (stream) to work on, pipe: whose scratch
    if (!pipe) pipe = &g->pipe[0];
    if (!ctx) ctx = g->ctx;
    const int N = g->N, L = g->L, M = N * L, Vt = g->tmpl->V;
    const bool timing = std::getenv("MSMHIP_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "      patches: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    DevBuf<double> &d_c = pipe->d_centres, &d_sep = pipe->d_sep;
    DevBuf<uint32_t> &d_slots = pipe->d_slots;
    DevBuf<int> &d_counts = pipe->d_counts;
    MSM_HIP(d_c.ensure(3 * (size_t)M));
    MSM_HIP(d_sep.ensure(M));
    {
        int st = launch_group_centres(ctx, g->d_moved.p + 3 * (size_t)s * M, g->d_spacing.p + (size_t)s * N, N, L, d_c.p, d_sep.p);
        if (st) return st;
    }
    MSM_HIP(d_counts.ensure((size_t)M + 1));  // + the number of undecided entries
    MSM_HIP(pipe->d_chunkb.ensure((size_t)(Vt + 63) / 64 + 1));
    int cap = std::max(256, g->patch_cap_hint.load());  // the previous call's largest patch: one k_range pass instead of two
    std::vector<int> counts((size_t)M + 1);
    auto first_sync = [&]() -> int {
        if (!deferred) return ctx_sync(ctx);
        const char *what = deferred;
        deferred = nullptr;
        return check_status(ctx, what);
    };
    // Round 5: from a set-up's second subject on the row offsets are summed and the list compacted on the device BEFORE the host has seen the counts (the list
    // sized from the longest one so far, the kernel guarded), so that a subject's batch stream synchronises once -- it was three round trips (after the
    // resampling, after the counts, after the compaction), each followed by launches onto an idle stream: a quarter of the stream's time was gaps.
    const size_t list_hint = g->pidx_hint.load();
    bool early = false;
    for (int attempt = 0; attempt < 3; ++attempt) {
        MSM_HIP(d_slots.ensure((size_t)M * cap));
        int st = launch_range(ctx, d_c.p, M, g->tmpl->d_xyz, Vt, d_sep.p, g->p.range, cap, d_slots.p, d_counts.p, pipe->d_chunkb.p, d_counts.p + M, L,
                              g->rgrid_valid ? &g->rgrid : nullptr);
        if (st) return st;
        early = attempt == 0 && list_hint > 0;
        if (early) {
            MSM_HIP(g->pptr[s]->ensure((size_t)M + 1));
            MSM_HIP(pipe->d_scan_tmp.ensure((size_t)M / 4096 + 2));
            MSM_HIP(g->pidx[s]->ensure(list_hint + list_hint / 8 + 1024));
            MSM_HIP(hipMemcpyAsync(g->pptr[s]->p, d_counts.p, sizeof(int) * (size_t)M, hipMemcpyDeviceToDevice, ctx->stream));
            st = launch_scan_exclusive(ctx, g->pptr[s]->p, M, pipe->d_scan_tmp.p);
            if (st) return st;
            st = launch_patch_compact(ctx, d_slots.p, cap, g->pptr[s]->p, M, g->pidx[s]->p, g->pidx[s]->cap);
            if (st) return st;
        }
        MSM_TRY(d_counts.download(counts.data(), (size_t)M + 1, ctx));
        MSM_TRY(first_sync());
        const int mx = *std::max_element(counts.begin(), counts.begin() + M);
        if (mx + 16 > g->patch_cap_hint.load()) g->patch_cap_hint.store(mx + 16);
        if (mx <= cap) break;
        if (attempt == 2) return fail(MSM_ERR_CAPACITY, "group patch capacity");
        cap = mx + 16;
        early = false;
    }
    lap("range kernel");
    if (counts[M] == 0) {
        // nothing sits on the threshold: the rows are final.  The list is compacted where it is used; the host keeps the row
        // offsets and fetches the 12.7 MB of indices only if msm_group_patch / msm_group_export_subject ask for them.
        auto &pp = g->h_pptr[s];
        pp.assign((size_t)M + 1, 0);
        for (int k = 0; k < M; ++k) pp[k + 1] = pp[k] + counts[k];
        g->h_pidx[s].clear();
        if ((size_t)pp[M] > g->pidx_hint.load()) g->pidx_hint.store((size_t)pp[M]);
        if (early && (size_t)pp[M] <= g->pidx[s]->cap) {
            lap("lists (device, no second look)");
            return MSM_OK;
        }
        MSM_TRY(g->pptr[s]->upload(pp.data(), pp.size(), ctx));
        MSM_HIP(g->pidx[s]->ensure(std::max<size_t>((size_t)pp[M], 1)));
        int st = launch_patch_compact(ctx, d_slots.p, cap, g->pptr[s]->p, M, g->pidx[s]->p);
        if (st) return st;
        MSM_TRY(ctx_sync(ctx));
        lap("lists (device)");
        return MSM_OK;
    }
    // 50 MB at ico6 / 19 labels: through pinned memory (a pageable copy of this size took most of this function's time)
    void *pin = nullptr;
    {
        int st = ctx_io_pinned(ctx, sizeof(uint32_t) * (size_t)M * cap, &pin);
        if (st) return st;
    }
    const uint32_t *slots = static_cast<const uint32_t *>(pin);
    MSM_HIP(hipMemcpyAsync(pin, d_slots.p, sizeof(uint32_t) * (size_t)M * cap, hipMemcpyDeviceToHost, ctx->stream));
    MSM_TRY(ctx_sync(ctx));
    lap("slots to the host");
    auto &pp = g->h_pptr[s];
    auto &pi = g->h_pidx[s];
    pp.assign(M + 1, 0);
    const double *tx = g->tmpl->xyz.data();
    // the centres and spacings the kernel used, for the host's decisions (this branch is rare: exact ties)
    std::vector<double> centres(3 * (size_t)M), sep(M);
    MSM_TRY(d_c.download(centres.data(), centres.size(), ctx));
    MSM_TRY(d_sep.download(sep.data(), sep.size(), ctx));
    MSM_TRY(ctx_sync(ctx));
    // an entry flagged by the kernel sits within 1e-11 of the threshold: decided with the host libm, as the reference does
    auto keeps = [&](int k, uint32_t e) {
        if (!(e & 0x80000000u)) return true;
        const V3 c = mk(centres[k], centres[M + k], centres[2 * (size_t)M + k]);
        const double arc = chord_to_arc(norm(sub(c, pt(tx, Vt, (int)(e & 0x7fffffffu)))));
        return arc < g->p.range * sep[k];
    };
    const int workers = host_workers();
    parallel_chunks(M, workers, [&](int, int k0, int k1) {  // rows are independent: count, then (below) fill at the prefix sums
        for (int k = k0; k < k1; ++k) {
            int n = 0;
            for (int j = 0; j < counts[k]; ++j) n += keeps(k, slots[(size_t)k * cap + j]) ? 1 : 0;
            pp[k + 1] = n;
        }
    });
    for (int k = 0; k < M; ++k) pp[k + 1] += pp[k];
    pi.resize((size_t)pp[M]);
    parallel_chunks(M, workers, [&](int, int k0, int k1) {
        for (int k = k0; k < k1; ++k) {
            int32_t *dst = pi.data() + pp[k];
            for (int j = 0; j < counts[k]; ++j) {
                const uint32_t e = slots[(size_t)k * cap + j];
                if (keeps(k, e)) *dst++ = (int32_t)(e & 0x7fffffffu);
            }
        }
    });
    lap("lists");
    MSM_TRY(g->pptr[s]->upload(pp.data(), pp.size(), ctx));
    MSM_HIP(g->pidx[s]->ensure(std::max<size_t>(pi.size(), 1)));
    {
        int st = upload_staged(ctx, g->pidx[s]->p, pi.data(), pi.size() * sizeof(int32_t));  // 12.7 MB at ico6 / 19 labels
        if (st) return st;
    }
    MSM_TRY(ctx_sync(ctx));
    lap("uploads");
    return MSM_OK;
}

int group_args(msm_group *g, GroupArgs &a) {
    if (!g->ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup() must be called first");
    a.S = g->S;
    a.N = g->N;
    a.L = g->L;
    a.D = g->D;
    a.Tc = g->Tc;
    a.Vt = g->tmpl->V;
    a.simmeasure = g->p.simmeasure;
    a.fixnan = g->p.fixnan;
    a.pairs = g->d_pairs.p;
    a.triplets = g->d_triplets.p;
    a.pptr = g->d_pptrp.p;
    a.pidx = g->d_pidxp.p;
    a.F = g->d_Fp.p;
    const bool pval_all = g->pval_ready && g->pval_p0 == 0 && g->pval_p1 == g->npairs;  // (a slice's copies serve that slice's label steps only: group_move_compute)
    a.pval = pval_all ? const_cast<const double *const *>(g->d_pvalp.p) : nullptr;
    a.dir = pval_all ? g->d_patch_dir.p : nullptr;
    a.mask = g->mask.empty() ? nullptr : g->d_mask.p;
    a.moved = g->d_moved.p;
    a.cp = g->d_cp.p;
    a.orig = g->d_orig.p;
    a.lambda = g->p.lambda;
    a.mu = g->p.mu;
    a.kappa = g->p.kappa;
    a.k_exp = g->p.k_exp;
    a.rexp = g->p.rexp;
    a.subcorr = 0.1 * g->S;  // set_meshes, M/DiscreteGroupCostFunction.h:45
    a.percentile = g->p.percentile;
    a.move_labeling = nullptr;
    a.move_label = a.move_offset = 0;
    a.move_order = nullptr;
    a.move_order4 = nullptr;
    a.move_first = a.move_count = 0;
    a.move_base = 0;
    a.move_combos = 0;
    a.move_prev = nullptr;
    a.move_e00 = nullptr;
    a.move_e11 = nullptr;
    a.patch_cap = g->patch_max;
    a.pair_lanes = g->pair_lanes;
    a.status = g->ctx->d_status;
    return MSM_OK;
}

constexpr int kBatchChunk = 1 << 22;  // evaluations per launch of a batch call (84 MB of pinned staging at most)

// the index columns of a batch go through the context's pinned block into buffers kept on the handle (no allocation and no
// pageable copy per call); *pinned_out is where the caller's kernel output is to be copied for the way back
int stage_batch(msm_group *g, const int32_t *const *cols, int ncols, int n, double **pinned_out) {
    msm_ctx *ctx = g->ctx;
    const size_t bi = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255, bo = sizeof(double) * (size_t)n;
    void *pin = nullptr;
    int st = ctx_io_pinned(ctx, bi * ncols + bo, &pin);
    if (st) return st;
    for (int k = 0; k < ncols; ++k) {
        MSM_HIP(g->d_query[k].ensure(n));
        std::memcpy((char *)pin + bi * k, cols[k], sizeof(int32_t) * (size_t)n);
        MSM_HIP(hipMemcpyAsync(g->d_query[k].p, (char *)pin + bi * k, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    MSM_HIP(g->d_answer.ensure(n));
    *pinned_out = reinterpret_cast<double *>((char *)pin + bi * ncols);
    return MSM_OK;
}

}  // namespace

extern "C" {

msm_group *msm_group_create(msm_ctx *ctx, const msm_group_params *params, int32_t S) {
    if (!ctx || !params || S < 1) {
        fail(MSM_ERR_INVALID, "msm_group_create: bad arguments");
        return nullptr;
    }
    if (params->simmeasure != 1 && params->simmeasure != 2 && params->simmeasure != 4 && params->simmeasure != 5) {
        fail(MSM_ERR_INVALID, "Unknown similarity metric");  // get_sim_for_min, M/similarities.h:57
        return nullptr;
    }
    const double percentile = params->percentile == 0.0 ? 0.75 : params->percentile;  // sparsesimkernel's default, M/similarities.h:68
    if ((params->simmeasure == 4 || params->simmeasure == 5) && !(percentile > 0.0 + 1e-8 && percentile < 1.0 - 1e-8)) {
        fail(MSM_ERR_INVALID, "Percentile must be between 0 and 1.");  // M/mesh_registration.cpp:782-783
        return nullptr;
    }
    msm_group *g = new msm_group();
    g->ctx = ctx;
    g->p = *params;
    g->p.percentile = percentile;
    g->S = S;
    g->cpmesh.assign(S, nullptr);
    g->data.assign(S, nullptr);
    g->scratch.assign(S, nullptr);
    g->feat.resize(S);
    g->orig.resize(S);
    g->have_orig.assign(S, 0);
    g->spacing.resize(S);
    g->h_pptr.resize(S);
    g->h_pidx.resize(S);
    g->imp_stat.resize(S);
    for (int s = 0; s < S; ++s) {
        g->pptr.emplace_back(new DevBuf<int32_t>());
        g->pidx.emplace_back(new DevBuf<int32_t>());
    }
    return g;
}

void msm_group_destroy(msm_group *g) {
    if (!g) return;
    (void)hipStreamSynchronize(g->ctx->stream);
    for (msm_mesh *m : g->cpmesh) msm_mesh_destroy(m);
    for (msm_mesh *m : g->scratch) msm_mesh_destroy(m);
    for (auto &lane : g->lanes) {
        if (lane.ctx) (void)hipStreamSynchronize(lane.ctx->stream);
        msm_mesh_destroy(lane.mesh);
        msm_ctx_destroy(lane.ctx);
    }
    for (auto &pp : g->pipe) {
        if (pp.batch.ctx) {
            (void)hipStreamSynchronize(pp.batch.ctx->stream);
            msm_ctx_destroy(pp.batch.ctx);
            pp.batch.ctx = nullptr;
        }
        if (pp.own_main && pp.main) {
            (void)hipStreamSynchronize(pp.main->stream);
            msm_ctx_destroy(pp.main);
        }
        pp.main = nullptr;
    }
    if (g->copy_stream) {
        (void)hipStreamSynchronize(g->copy_stream);
        (void)hipStreamDestroy(g->copy_stream);
    }
    for (hipEvent_t e : g->copy_events) (void)hipEventDestroy(e);
    if (g->t_ev0) (void)hipEventDestroy(g->t_ev0);
    if (g->t_ev1) (void)hipEventDestroy(g->t_ev1);
    delete g;
}

int msm_group_set_template(msm_group *g, msm_mesh *t, const double *mask) {
    if (!g || !t) return fail(MSM_ERR_INVALID, "msm_group_set_template: null argument");
    g->tmpl = t;
    g->mask.clear();
    if (mask) {
        g->mask.assign(mask, mask + t->V);
        MSM_TRY(g->d_mask.upload(g->mask.data(), g->mask.size(), g->ctx));
        MSM_TRY(ctx_sync(g->ctx));
    }
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return MSM_OK;
}

int msm_group_set_controlgrid(msm_group *g, const double *xyz, const int32_t *tri, int32_t N, int32_t Tc) {
    if (!g || !xyz || !tri || N <= 0 || Tc <= 0) return fail(MSM_ERR_INVALID, "msm_group_set_controlgrid: bad arguments");
    g->N = N;
    g->Tc = Tc;
    g->cp_tri.assign(tri, tri + 3 * (size_t)Tc);
    for (int s = 0; s < g->S; ++s) {
        msm_mesh_destroy(g->cpmesh[s]);
        g->cpmesh[s] = msm_mesh_create(g->ctx, xyz, N, tri, Tc);
        if (!g->cpmesh[s]) return MSM_ERR_HIP;
    }
    // estimate_triplets, M/DiscreteGroupModel.cpp:57-75
    g->triplets.resize(3 * (size_t)g->S * Tc);
    for (int s = 0; s < g->S; ++s)
        for (int t = 0; t < Tc; ++t) {
            int32_t v[3] = {tri[t] + s * N, tri[Tc + t] + s * N, tri[2 * Tc + t] + s * N};
            std::sort(v, v + 3);
            std::copy(v, v + 3, &g->triplets[3 * ((size_t)s * Tc + t)]);
        }
    MSM_TRY(g->d_triplets.upload(g->triplets.data(), g->triplets.size(), g->ctx));
    MSM_TRY(ctx_sync(g->ctx));
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return MSM_OK;
}

int msm_group_set_subject(msm_group *g, int32_t s, msm_mesh *data, const double *feat, int32_t D) {
    if (!g || !data || !feat || s < 0 || s >= g->S || D <= 0) return fail(MSM_ERR_INVALID, "msm_group_set_subject: bad arguments");
    if (g->N <= 0) return fail(MSM_ERR_STATE, "msm_group: the control grid must be set first");
    if (data->V < g->N) return fail(MSM_ERR_INVALID, "data mesh has fewer vertices than the control grid");
    for (int o = 0; o < g->S; ++o)
        if (o != s && g->data[o] && g->D != D) return fail(MSM_ERR_INVALID, "all subjects must have the same number of feature dimensions");
    g->D = D;
    g->data[s] = data;
    g->feat[s].assign(feat, feat + (size_t)D * data->V);
    if (!g->have_orig[s]) {
        g->orig[s].resize(3 * (size_t)g->N);
        for (int v = 0; v < g->N; ++v) {
            g->orig[s][v] = data->xyz[v];
            g->orig[s][g->N + v] = data->xyz[data->V + v];
            g->orig[s][2 * (size_t)g->N + v] = data->xyz[2 * (size_t)data->V + v];
        }
        g->have_orig[s] = 1;
    }
    if (!g->scratch[s] || g->scratch[s]->V != data->V || g->scratch[s]->T != data->T) {
        msm_mesh_destroy(g->scratch[s]);
        g->scratch[s] = msm_mesh_create(g->ctx, data->xyz.data(), data->V, data->tri.data(), data->T);
        if (!g->scratch[s]) return MSM_ERR_HIP;
    }
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return MSM_OK;
}

int msm_group_reset_cpgrid(msm_group *g, int32_t s, const double *xyz) {
    if (!g || !xyz || s < 0 || s >= g->S || !g->cpmesh[s]) return fail(MSM_ERR_INVALID, "msm_group_reset_cpgrid: bad arguments");
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return msm_mesh_update_coords(g->cpmesh[s], xyz);
}

int msm_group_set_labels(msm_group *g, const double *labels, int32_t L) {
    if (!g || !labels || L <= 0) return fail(MSM_ERR_INVALID, "msm_group_set_labels: bad arguments");
    g->L = L;
    g->labels.assign(labels, labels + 3 * (size_t)L);
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return MSM_OK;
}

// The order of the pair list (what msm_group_get_pairs returns and every pair index of this interface refers to).  0 (default): the reference's,
// estimate_pairs' loops -- subject A, control point, subject B (M/DiscreteGroupModel.cpp:37-55).  1: control point by control point along a space-
// filling curve, all subject pairs of one control point together: a contiguous slice of THAT list is a region of the sphere, so a rank evaluating
// an eighth of it touches an eighth of every resampled map instead of all of eight subjects' and most of everyone else's (an eighth of a label
// step at S = 64, ico6 / ico4: 1.59 -> 1.42 ms of kernels).  The optimiser takes the list as it comes (I/Fusion/Fusion.h:157-196 reads pairs[i]
// beside the i-th costs); the same set of pairs either way.  Takes effect at the next msm_group_setup / msm_group_setup_subjects.
msm_ctx *msm_group_context(msm_group *g) { return g ? g->ctx : nullptr; }

int msm_group_set_pair_layout(msm_group *g, int32_t layout) {
    if (!g || (layout != 0 && layout != 1)) return fail(MSM_ERR_INVALID, "msm_group_set_pair_layout: layout must be 0 (reference order) or 1 (control-point major)");
    if (g->pair_layout == layout) return MSM_OK;
    g->pair_layout = layout;
    g->pair_order.clear();  // rebuilt (and with it the slices and pieces) by the next set-up
    g->order_S = g->order_N = 0;
    g->order_p0 = g->order_p1 = -1;
    g->pairs.clear();
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    return MSM_OK;
}

// Who computes estimate_rotation_matrix(centre, vertex) (R/point.cpp:97-152) for the vertices of the data meshes in get_patch_data (M/DiscreteGroupModel.cpp:
// 97-105).  1 (default): the host's libm -- acos / sincos as the reference calls them -- with the matrices applied on the device in the reference's operation
// order: the rotated meshes are then the reference's to the bit.  That matters when a label carries data vertices EXACTLY onto template vertices (a regular
// icosphere as the template under regular data grids, as gMSM's own scripts set a run up): which triangle around such a vertex "contains" it is decided by the
// last bits of the rotation.  V x 0.15 us of host time per subject and set-up, on the host workers beside the GPU's work on other subjects
// (S = 64 at ico6: 145 against 144 ms per set-up).  0: the device computes them (its own acos / sincos).  Takes effect at the next set-up.
int msm_group_set_rotation_mode(msm_group *g, int32_t mode) {
    if (!g || (mode != 0 && mode != 1)) return fail(MSM_ERR_INVALID, "msm_group_set_rotation_mode: mode must be 0 (device) or 1 (the host's libm)");
    if (g->rotation_mode != mode) {
        g->rotation_mode = mode;
        g->ready = false;
        g->drop_kept();
    }
    return MSM_OK;
}

}  // extern "C" (re-opened below)

namespace {

// estimate_pairs, spacings, rotations: cheap, needs every subject's control grid, identical on every rank
int group_common_setup(msm_group *g) {
    if (!g->tmpl || g->N <= 0 || g->L <= 0) return fail(MSM_ERR_STATE, "msm_group: template, control grid and labels must be set first");
    msm_ctx *ctx = g->ctx;
    const int S = g->S, N = g->N, L = g->L;
    const double centre[3] = {g->labels[0], g->labels[L], g->labels[2 * (size_t)L]};
    const bool timing = std::getenv("MSMHIP_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  group set-up, common: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    // estimate_pairs, M/DiscreteGroupModel.cpp:37-55: closest control point of subject B for every control point of A.  The S
    // control grids share their triangle list, so their search trees are built together as a forest and one kernel writes the
    // whole list (N S (S - 1) / 2 pairs: 5.2 M at S = 64) where the cost kernels read it.  Round 1 / the first half of round 2:
    // S - 1 search calls with host trees and host loops, 67 ms at S = 64 on every rank.
    // get_spacings :123-139 and get_rotations :77-86 on the host (their asin / acos decide patch membership to the last bit, so they use the host's libm as
    // the reference does), subjects spread over the host threads -- started here, beside the GPU's estimate_pairs below (round 5: they used to follow it,
    // 2 ms that every rank of a sharded run repeats)
    g->rot.resize(9 * (size_t)S * N);
    g->moved.clear();
    g->moved_on_host = false;
    std::vector<double> cp_all(3 * (size_t)S * N), orig_all(3 * (size_t)S * N, 0.0), spacing_all((size_t)S * N);
    std::vector<int> sub_status(S, MSM_OK);
    std::thread host_side([&] {
        parallel_for(S, host_workers(), [&](int s) {
            msm_mesh *cm = g->cpmesh[s];
            g->spacing[s].resize(N);
            double mvd;
            int st = msm_cp_spacings(cm->xyz.data(), cm->tri.data(), N, g->Tc, g->spacing[s].data(), &mvd);
            if (!st) st = msm_cp_rotations(centre, cm->xyz.data(), N, &g->rot[9 * (size_t)s * N]);
            sub_status[s] = st;
            std::copy(g->spacing[s].begin(), g->spacing[s].end(), spacing_all.begin() + (size_t)s * N);
            std::copy(cm->xyz.begin(), cm->xyz.end(), cp_all.begin() + 3 * (size_t)s * N);
            if (g->have_orig[s]) std::copy(g->orig[s].begin(), g->orig[s].end(), orig_all.begin() + 3 * (size_t)s * N);
        });
    });
    struct Joiner {  // every return below leaves with the thread joined
        std::thread &t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{host_side};
    g->npairs = (int64_t)N * S * (S - 1) / 2;
    g->pairs.clear();
    ++g->pairs_gen;
    std::vector<double> cp_soa(3 * (size_t)S * N);
    for (int s = 0; s < S; ++s)
        for (int ax = 0; ax < 3; ++ax)
            std::copy(g->cpmesh[s]->xyz.begin() + (size_t)ax * N, g->cpmesh[s]->xyz.begin() + (size_t)(ax + 1) * N, cp_soa.begin() + (size_t)ax * S * N + (size_t)s * N);
    MSM_HIP(g->d_cp_soa.ensure(cp_soa.size()));
    {
        int st = upload_staged(ctx, g->d_cp_soa.p, cp_soa.data(), sizeof(double) * cp_soa.size());
        if (st) return st;
    }
    MSM_HIP(g->d_pairs.ensure(std::max<size_t>(2 * (size_t)g->npairs, 1)));
    bool pairs_done = false;
    if (S >= 2) {
        int st = gpu_build_forest(ctx, g->cp_forest, g->d_cp_soa.p, (size_t)S * N, (size_t)N, N, g->cpmesh[0]->d_tri, g->Tc, S);
        if (st == MSM_OK) {
            std::vector<int2> info(S);
            for (int b = 0; b < S; ++b) info[b] = make_int2(g->cp_forest.info[b].nnodes, g->cp_forest.info[b].grid_depth);
            MSM_TRY(g->d_forest_info.upload(info.data(), info.size(), ctx));
            ForestDev fd;
            fd.node = g->cp_forest.node.p, fd.parent = g->cp_forest.parent.p, fd.leaf_tri = g->cp_forest.leaf_tri.p, fd.grid = g->cp_forest.grid.p;
            fd.cone = g->cp_forest.cone.p, fd.rec = g->cp_forest.rec.p;
            fd.s_node = g->cp_forest.s_node, fd.s_leaf = g->cp_forest.s_leaf, fd.s_rec = g->cp_forest.s_rec, fd.s_grid = g->cp_forest.s_grid;
            fd.info = g->d_forest_info.p;
            st = launch_group_pairs(ctx, fd, g->d_cp_soa.p, S, N, g->d_pairs.p);
            if (st) return st;
            st = check_status(ctx, "estimate_pairs");  // synchronises (info is a local)
            if (st) return st;
            pairs_done = true;
        } else if (st != MSM_ERR_CAPACITY) {
            return st;
        }
    }
    if (!pairs_done && S >= 2) {  // a control grid whose tree outgrew the forest's arrays: one search per target subject, as before
        g->pairs.resize(2 * (size_t)g->npairs);
        std::vector<std::vector<int32_t>> closest((size_t)S * S);
        std::vector<double> q;
        std::vector<int32_t> found;
        for (int b = 1; b < S; ++b) {
            const size_t Mq = (size_t)b * N;
            q.resize(3 * Mq);
            for (int a = 0; a < b; ++a)
                for (int ax = 0; ax < 3; ++ax)
                    std::copy(g->cpmesh[a]->xyz.begin() + (size_t)ax * N, g->cpmesh[a]->xyz.begin() + (size_t)(ax + 1) * N, q.begin() + ax * Mq + (size_t)a * N);
            found.resize(Mq);
            int st = msm_closest_vertex(g->cpmesh[b], q.data(), (int32_t)Mq, found.data());
            if (st) return st;
            for (int a = 0; a < b; ++a) closest[(size_t)a * S + b].assign(found.begin() + (size_t)a * N, found.begin() + (size_t)(a + 1) * N);
        }
        size_t pair = 0;
        for (int a = 0; a < S; ++a)
            for (int v = 0; v < N; ++v)
                for (int b = a + 1; b < S; ++b) {
                    g->pairs[2 * pair] = a * N + v;
                    g->pairs[2 * pair + 1] = b * N + closest[(size_t)a * S + b][v];
                    ++pair;
                }
        MSM_TRY(g->d_pairs.upload_vec(g->pairs, ctx));
    }
    lap("estimate_pairs");
    {
        // Processing order of a label step's pairs.  The list above runs subject A, control point, subject B: neighbours in it
        // share A's patch, but consecutive control-point ids are not neighbours on the sphere and every pair pulls another
        // subject's patch in, so at S = 16 a step fetched 2.9 GB through an L2 hit rate of 69 %.  Processed control point by
        // control point along a space-filling curve -- all S (S - 1) / 2 subject pairs of one, then the next -- the patches of a
        // neighbourhood (S subjects x 2 labels x a few KB) stay in the L2 while every pair that needs them runs.  Results keep
        // the list's positions (GroupArgs::move_order).
        // The order only serves locality: any permutation is correct, and the control grids move little from one iteration to the
        // next, so the order of the first set-up with these sizes is kept (with it the slice on the device and its four pieces).
        if (g->pair_order.size() != (size_t)g->npairs || g->order_S != S || g->order_N != N) {
            const double *c0 = g->cpmesh[0]->xyz.data();
            auto spread = [](uint32_t v) {
                v &= 0x3ff;
                v = (v | (v << 16)) & 0x030000ff;
                v = (v | (v << 8)) & 0x0300f00f;
                v = (v | (v << 4)) & 0x030c30c3;
                v = (v | (v << 2)) & 0x09249249;
                return v;
            };
            std::vector<std::pair<uint32_t, int>> key(N);
            for (int v = 0; v < N; ++v) {
                uint32_t q[3];
                for (int ax = 0; ax < 3; ++ax) {
                    const double u = (c0[(size_t)ax * N + v] + kBounds) / (2 * kBounds);
                    q[ax] = (uint32_t)std::max(0.0, std::min(1023.0, u == u ? u * 1024.0 : 0.0));
                }
                key[v] = {spread(q[0]) << 2 | spread(q[1]) << 1 | spread(q[2]), v};
            }
            std::sort(key.begin(), key.end());
            std::vector<int64_t> base(S, 0);
            for (int a = 1; a < S; ++a) base[a] = base[a - 1] + (int64_t)N * (S - a);
            g->pair_order.resize((size_t)g->npairs);
            size_t at = 0;
            for (int i = 0; i < N; ++i) {
                const int v = key[i].second;
                for (int a = 0; a + 1 < S; ++a)
                    for (int b = a + 1; b < S; ++b) g->pair_order[at++] = (int32_t)(base[a] + (int64_t)v * (S - 1 - a) + (b - a - 1));
            }
            g->order_S = S, g->order_N = N;
            g->order_p0 = g->order_p1 = -1;
            if (g->pair_layout == 1) {  // the curve order becomes the list's own: kept on the device for every set-up's permutation, the processing order is the identity
                MSM_HIP(g->d_pair_perm.ensure(std::max<size_t>(g->pair_order.size(), 1)));
                int st = upload_staged(ctx, g->d_pair_perm.p, g->pair_order.data(), sizeof(int32_t) * g->pair_order.size());
                if (st) return st;
                MSM_TRY(ctx_sync(ctx));
                std::iota(g->pair_order.begin(), g->pair_order.end(), 0);
            }
        }
        if (g->pair_layout == 1 && g->npairs > 0) {
            MSM_HIP(g->d_pairs_tmp.ensure(2 * (size_t)g->npairs));
            std::swap(g->d_pairs.p, g->d_pairs_tmp.p);  // d_pairs_tmp: the list in the reference's order, as written above
            std::swap(g->d_pairs.cap, g->d_pairs_tmp.cap);
            std::swap(g->d_pairs.owned, g->d_pairs_tmp.owned);
            int st = launch_group_permute_pairs(ctx, g->d_pairs_tmp.p, g->d_pair_perm.p, (int)g->npairs, g->d_pairs.p);
            if (st) return st;
            if (!g->pairs.empty()) {  // the fallback above filled the host copy and queued its upload: in the other order now -- fetched again when asked for
                MSM_TRY(ctx_sync(ctx));  // (the upload reads the vector)
                g->pairs.clear();
            }
        }
    }
    lap("pair order");
    // ROT * label for every (node, label) -- 3.1 M products at S = 64, 75 MB -- on the device from the uploaded rotations (multiplications and additions
    // only, in the host's order: the same bits)
    host_side.join();
    for (int s = 0; s < S; ++s)
        if (sub_status[s]) return fail(sub_status[s], "msm_group: spacings / rotations of subject %d's control grid failed", s);
    MSM_HIP(g->d_rot.ensure(g->rot.size()));
    MSM_HIP(g->d_moved.ensure(3 * (size_t)S * N * L));
    {
        int st = upload_staged(ctx, g->d_rot.p, g->rot.data(), sizeof(double) * g->rot.size());
        if (st) return st;
    }
    // through the pinned staging block: asynchronous copies out of freshly allocated pageable vectors (these are locals) left a stall of 10-30 ms
    // behind them that the NEXT work on the stream paid for -- the first subject's rotations (tools/time_group_rank.py; DESIGN 5.4b has the same
    // finding for the mesh uploads)
    MSM_HIP(g->d_labels3.ensure(3 * (size_t)L));
    MSM_HIP(g->d_spacing.ensure(spacing_all.size()));
    MSM_HIP(g->d_cp.ensure(cp_all.size()));
    MSM_HIP(g->d_orig.ensure(orig_all.size()));
    {
        int st = upload_staged(ctx, g->d_labels3.p, g->labels.data(), sizeof(double) * 3 * (size_t)L);
        if (!st) st = upload_staged(ctx, g->d_spacing.p, spacing_all.data(), sizeof(double) * spacing_all.size());
        if (!st) st = upload_staged(ctx, g->d_cp.p, cp_all.data(), sizeof(double) * cp_all.size());
        if (!st) st = upload_staged(ctx, g->d_orig.p, orig_all.data(), sizeof(double) * orig_all.size());
        if (st) return st;
    }
    {
        int st = launch_group_moved(ctx, g->d_rot.p, S * N, g->d_labels3.p, L, g->d_moved.p);
        if (st) return st;
    }
    MSM_TRY(ctx_sync(ctx));
    g->F.resize((size_t)S * L);
    g->Fslab.resize(S);
    g->have_subject.assign(S, 0);
    g->common_ready = true;
    lap("spacings, rotations, uploads");
    return MSM_OK;
}

// the lanes, each with a scratch mesh of dm's topology (kept from subject to subject while the topology stays the same)
int ensure_lanes(msm_group *g, const msm_mesh *dm) {
    static const int want = [] {
        const char *e = std::getenv("MSMHIP_GROUP_LANES");
        const int v = e ? std::atoi(e) : 6;  // 14.1 ms per ico6 subject with one lane, 8.2 with two, 7.8 with four, 6.0-6.4 with six to eight
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    if ((int)g->lanes.size() != want) {
        for (auto &lane : g->lanes) {
            msm_mesh_destroy(lane.mesh);
            msm_ctx_destroy(lane.ctx);
        }
        g->lanes.assign(want, msm_group::Lane{});
    }
    for (auto &lane : g->lanes) {
        if (!lane.ctx) {
            lane.ctx = msm_ctx_create(g->ctx->device);
            if (!lane.ctx) return MSM_ERR_HIP;
        }
        if (lane.mesh && (lane.mesh->V != dm->V || lane.mesh->T != dm->T || lane.mesh->tri != dm->tri)) {
            MSM_TRY(ctx_sync(lane.ctx));
            msm_mesh_destroy(lane.mesh);
            lane.mesh = nullptr;
        }
        if (!lane.mesh) {
            lane.mesh = msm_mesh_create(lane.ctx, dm->xyz.data(), dm->V, dm->tri.data(), dm->T);
            if (!lane.mesh) return MSM_ERR_HIP;
            lane.mesh->gpu_tree_always = true;
            int st = ensure_adjacency_dev(lane.mesh);
            if (st) return st;
        }
    }
    return MSM_OK;
}

static bool group_device_path() {
    static const bool host_surgery = [] { const char *e = std::getenv("MSMHIP_SURGERY"); return e && std::strcmp(e, "host") == 0; }();
    static const bool host_trees = [] { const char *e = std::getenv("MSMHIP_OCTREE"); return e && std::strcmp(e, "host") == 0; }();
    return !host_trees && !host_surgery;
}

// get_patch_data of one subject with everything in HBM, in three stages (msm_group_setup_subjects runs them as a pipeline):
//   prepare   main stream: the L rotated copies of the data mesh side by side in one 3 x (L * V) array (x of label 0, x of label
//             1, ..., y of label 0, ...), the subject's features, the L trees built together as a forest; ends synchronised
//   lanes     per label: the rotated coordinates become a lane mesh's, then queries, weight-list surgery (resample_kernels.hip) and
//             the weighted sums write F[s][l] directly.  One such pipeline is a dependent chain of some thirty kernels of a few
//             microseconds each (eighty with the tree build, when the forest could not be used); the lanes run K of them side by
//             side, driven by (two) host threads, each with its share of the lanes and every other label -- submitting the launches
//             is itself a third of a millisecond of host time per label.  A thread queues the first half of its lanes' labels
//             (up to where a tree build's outcome is looked at), then the second half of each, so that it never waits for work it
//             has only just submitted.
//   patches   main stream: subject_patches
// the L rotated copies of subject s's data mesh into d_out (3 x (L * V)), on ctx's stream: one launch; with rotation_mode 1 the V rotation matrices are
// computed here on the host workers first (libm's acos / sincos: 0.15 us each) and uploaded (72 V bytes)
static int rotate_subject(msm_group *g, msm_mesh *dm, msm_ctx *ctx, std::vector<double> &rot9, DevBuf<double> &d_rot9, double *d_out) {
    const int L = g->L, V = dm->V;
    const double centre[3] = {g->labels[0], g->labels[L], g->labels[2 * (size_t)L]};
    static const int env_mode = [] {
        const char *e = std::getenv("MSMHIP_GROUP_ROTATIONS");
        return !e ? -1 : (std::strcmp(e, "host") == 0 ? 1 : (std::strcmp(e, "device") == 0 ? 0 : -1));
    }();
    const int mode = env_mode >= 0 ? env_mode : g->rotation_mode;
    const double *d_mats = nullptr;
    if (mode == 1) {
        if (dm->host_xyz_stale) {
            MSM_TRY(stage_d2h(ctx, dm->xyz.data(), dm->d_xyz, sizeof(double) * 3 * (size_t)V));
            MSM_TRY(ctx_sync(ctx));
            dm->host_xyz_stale = false;
        }
        rot9.resize(9 * (size_t)V);
        std::atomic<int> bad{0};
        const double *xyz = dm->xyz.data();
        parallel_chunks(V, host_workers(), [&](int, int v0, int v1) {
            for (int v = v0; v < v1; ++v)
                if (!rotation_matrix(mk(centre[0], centre[1], centre[2]), pt(xyz, V, v), rot9.data() + 9 * (size_t)v)) bad.store(1);
        });
        if (bad.load()) return fail(MSM_ERR_ROTATION, "rotation angle is greater than 90 degrees");
        MSM_HIP(d_rot9.ensure(rot9.size()));
        int st = upload_staged(ctx, d_rot9.p, rot9.data(), sizeof(double) * rot9.size());
        if (st) return st;
        d_mats = d_rot9.p;
    }
    return launch_rotate_to_labels(ctx, dm->d_xyz, V, centre, g->d_labels3.p, L, d_mats, d_out, (size_t)L * V);
}

static int stage_prepare(msm_group *g, int s, msm_group::Stage &b, msm_ctx *ctx) {
    if (!g->data[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d has no data", s);
    const int L = g->L, D = g->D, Vt = g->tmpl->V;
    msm_mesh *dm = g->data[s];
    const int V = dm->V, T = dm->T;
    const size_t LV = (size_t)L * V;
    static const bool timing = std::getenv("MSMHIP_TIMING") != nullptr && std::getenv("MSMHIP_TIMING")[0] == '2';
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "      prepare subject %d: %s %.2f ms\n", s, what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    MSM_HIP(b.d_rot.ensure(3 * LV));
    MSM_HIP(b.d_feat.ensure((size_t)D * V));
    int st = upload_staged(ctx, b.d_feat.p, g->feat[s].data(), sizeof(double) * (size_t)D * V);  // (first: the host's copy into the staging block is not between two launches then)
    if (st) return st;
    lap("feature upload");
    st = rotate_subject(g, dm, ctx, b.rot9, b.d_rot9, b.d_rot.p);  // one launch for the L labels (round 4: L - 1 launches that each computed the matrices again, and 3 copies)
    if (st) return st;
    lap("rotations");
    st = subject_feature_slab(g, s, (size_t)D * Vt);
    if (st) return st;
    lap("slab");
    // the L trees, built together (one chain of launches per subject instead of one per label); a tree that outgrows its arrays
    // (a degenerate mesh) sends the subject down the per-label builds of the lanes
    static const bool no_forest = [] { const char *e = std::getenv("MSMHIP_GROUP_FOREST"); return e && std::strcmp(e, "off") == 0; }();
    b.forest_ok = !no_forest;
    if (b.forest_ok) {
        st = gpu_build_forest(ctx, b.forest, b.d_rot.p, LV, (size_t)V, V, dm->d_tri, T, L);
        if (st == MSM_ERR_CAPACITY) b.forest_ok = false;
        else if (st) return st;
    }
    lap("forest");
    st = check_status(ctx, "get_patch_data (rotation)");  // synchronises: rotations, features and trees are where the lanes will read them
    return st;
}

// The per-label remainder with all L labels in every launch (the trees came from the forest): forward queries of the template's
// vertices in every tree, reverse queries of all rotated vertices in the template's tree (one launch over the L * V points of
// d_rot), vertex areas, weight-list surgery and the weighted sums into the subject's slab -- some forty launches per SUBJECT
// where the lanes made thirty per label.  On a stream of its own (batch.ctx), beside the main stream's work on the next subject.
static int stage_batch(msm_group *g, int s, msm_group::Stage &b, int which, msm_group::Batch &w, bool defer_status = false) {
    if (!w.ctx) {
        w.ctx = msm_ctx_create(g->ctx->device);
        if (!w.ctx) return MSM_ERR_HIP;
    }
    msm_ctx *ctx = w.ctx;  // (created by group_setup_pipeline)
    const int L = g->L, D = g->D;
    msm_mesh *dm = g->data[s], *tm = g->tmpl;
    const int V = dm->V, T = dm->T, Vt = tm->V, Tt = tm->T;
    const size_t LV = (size_t)L * V, LVt = (size_t)L * Vt, cap = 3 * (size_t)Vt + 3 * (size_t)V;
    int st = MSM_OK;  // (the data mesh's adjacency lists are on the device: group_setup_pipeline saw to it before the pipelines started)
    const size_t nscan = (size_t)std::max(V, Vt) / 4096 + 2;
    MSM_HIP(w.fvid.ensure(3 * LVt));
    MSM_HIP(w.fw.ensure(3 * LVt));
    MSM_HIP(w.rvid.ensure(3 * LV));
    MSM_HIP(w.rw.ensure(3 * LV));
    MSM_HIP(w.oldA.ensure(LV));
    MSM_HIP(w.newA.ensure(Vt));
    MSM_HIP(w.ta.ensure((size_t)L * std::max(T, Tt)));
    MSM_HIP(w.roff.ensure((size_t)L * (Vt + 1)));
    MSM_HIP(w.rfill.ensure(LVt));
    MSM_HIP(w.rkey.ensure(3 * LV));
    MSM_HIP(w.rwt.ensure(3 * LV));
    MSM_HIP(w.coff.ensure((size_t)L * (V + 1)));
    MSM_HIP(w.cfill.ensure(LV));
    MSM_HIP(w.ckey.ensure(L * cap));
    MSM_HIP(w.cval.ensure(L * cap));
    MSM_HIP(w.correction.ensure(LV));
    MSM_HIP(w.row_ptr.ensure((size_t)L * (Vt + 1)));
    MSM_HIP(w.col.ensure(L * cap));
    MSM_HIP(w.val.ensure(L * cap));
    MSM_HIP(w.tkey.ensure(L * cap));
    MSM_HIP(w.tval.ensure(L * cap));
    MSM_HIP(w.scan_tmp.ensure((size_t)L * nscan));
    MSM_HIP(w.long_flag.ensure(2 * (size_t)L));
    std::vector<int2> info(L);
    for (int l = 0; l < L; ++l) info[l] = make_int2(b.forest.info[l].nnodes, b.forest.info[l].grid_depth);
    MSM_HIP(w.info[which].ensure(L));
    MSM_TRY(stage_h2d(ctx, w.info[which].p, info.data(), sizeof(int2) * (size_t)L));
    MSM_TRY(ctx_sync(ctx));  // the previous subject's batch has finished with the scratch
    ForestDev fd;
    fd.node = b.forest.node.p, fd.parent = b.forest.parent.p, fd.leaf_tri = b.forest.leaf_tri.p, fd.grid = b.forest.grid.p;
    fd.cone = b.forest.cone.p, fd.rec = b.forest.rec.p;
    fd.s_node = b.forest.s_node, fd.s_leaf = b.forest.s_leaf, fd.s_rec = b.forest.s_rec, fd.s_grid = b.forest.s_grid;
    fd.info = w.info[which].p;
    // forward: the template's vertices in every label's tree; reverse: every label's vertices in the template's tree (:74-78)
    st = launch_query_forest(ctx, fd, L, tm->d_xyz, Vt, w.fvid.p, w.fw.p, LVt);
    if (st) return st;
    MSM_HIP(w.open.ensure(LV + 1));
    st = launch_query_rays(ctx, dev_tree(tm), b.d_rot.p, (int)LV, nullptr, w.rvid.p, w.rw.p, MSM_WEIGHTS_PROJECTED, w.open.p);  // (the template's direction table: group_setup_pipeline)
    if (st) return st;
    st = launch_vertex_areas_batch(ctx, b.d_rot.p, LV, (size_t)V, V, dm->d_tri, T, dm->d_tid_ptr, dm->d_tid, L, w.ta.p, w.oldA.p);
    if (st) return st;
    st = launch_vertex_areas_batch(ctx, tm->d_xyz, (size_t)Vt, 0, Vt, tm->d_tri, Tt, tm->d_tid_ptr, tm->d_tid, 1, w.ta.p, w.newA.p);
    if (st) return st;
    AdaptiveDevArgs a;
    a.nOld = V, a.nNew = Vt;
    a.fvid = w.fvid.p, a.fw = w.fw.p, a.rvid = w.rvid.p, a.rw = w.rw.p, a.oldA = w.oldA.p, a.newA = w.newA.p;
    a.roff = w.roff.p, a.rfill = w.rfill.p, a.rkey = w.rkey.p, a.rwt = w.rwt.p;
    a.coff = w.coff.p, a.cfill = w.cfill.p, a.ckey = w.ckey.p, a.cval = w.cval.p, a.correction = w.correction.p;
    a.scan_tmp = w.scan_tmp.p, a.long_flag = w.long_flag.p, a.tkey = w.tkey.p, a.tval = w.tval.p;
    a.row_ptr = w.row_ptr.p, a.col = w.col.p, a.val = w.val.p;
    a.B = L;
    a.fstride = LVt, a.rstride = LV;
    a.s_f = (size_t)Vt, a.s_r = (size_t)V, a.s_oldA = (size_t)V, a.s_newA = 0;
    a.s_roff = (size_t)Vt + 1, a.s_rfill = (size_t)Vt, a.s_r3 = 3 * (size_t)V;
    a.s_coff = (size_t)V + 1, a.s_cfill = (size_t)V, a.s_cap = cap, a.s_corr = (size_t)V, a.s_rowptr = (size_t)Vt + 1, a.s_scan = nscan;
    st = launch_adaptive_surgery(ctx, a);
    if (st) return st;
    st = launch_apply_rows_batch(ctx, a, D, b.d_feat.p, g->Fslab[s]->p, (size_t)D * Vt);
    if (st) return st;
    if (defer_status) return MSM_OK;  // the caller queues the subject's patch lists behind this and looks once (subject_patches)
    return check_status(ctx, "get_patch_data (resampling)");  // synchronises the batch stream
}

static int stage_lanes(msm_group *g, int s, msm_group::Stage &b) {
    std::lock_guard<std::mutex> only_one(g->lanes_mu);  // the lanes and their scratch meshes are shared by the set-up pipelines
    msm_ctx *ctx = g->ctx;
    const int L = g->L, D = g->D;
    msm_mesh *dm = g->data[s];
    const int V = dm->V;
    const size_t LV = (size_t)L * V;
    int st = ensure_lanes(g, dm);
    if (st) return st;
    const bool forest = b.forest_ok;
    const int K = (int)g->lanes.size();
    static const int want_threads = [] {
        const char *e = std::getenv("MSMHIP_GROUP_THREADS");
        const int v = e ? std::atoi(e) : 2;
        return v < 1 ? 1 : (v > 4 ? 4 : v);
    }();
    const int nthreads = std::max(1, std::min(want_threads, K));
    std::vector<int> status(nthreads, MSM_OK);
    std::vector<std::string> message(nthreads);
    auto drive = [&](int th) {
        int st = MSM_OK;
        (void)hipSetDevice(ctx->device);
        const int k0 = th * K / nthreads, k1 = (th + 1) * K / nthreads, Kt = k1 - k0;
        std::vector<int> mine;
        for (int l = th; l < L; l += nthreads) mine.push_back(l);
        for (size_t i0 = 0; i0 < mine.size() && !st; i0 += Kt) {
            for (int k = 0; k < Kt && i0 + k < mine.size() && !st; ++k) {
                const int l = mine[i0 + k];
                msm_group::Lane &lane = g->lanes[k0 + k];
                for (int a = 0; a < 3 && !st; ++a)
                    if (hipMemcpyAsync(lane.mesh->d_xyz + (size_t)a * V, b.d_rot.p + a * LV + (size_t)l * V, sizeof(double) * (size_t)V, hipMemcpyDeviceToDevice,
                                       lane.ctx->stream) != hipSuccess)
                        st = fail(MSM_ERR_HIP, "get_patch_data: device copy of the rotated coordinates failed");
                lane.mesh->tree_valid = false;
                lane.mesh->host_xyz_stale = true;
                if (!st && !forest) st = ensure_tree_begin(lane.mesh);
            }
            for (int k = 0; k < Kt && i0 + k < mine.size() && !st; ++k) {
                const int l = mine[i0 + k];
                msm_group::Lane &lane = g->lanes[k0 + k];
                AdaptiveDev w;
                DevTree tree;
                if (forest) tree = forest_tree(b.forest, l);
                st = adaptive_weights_dev(lane.mesh, g->tmpl, w, false, forest ? &tree : nullptr);
                if (!st) st = apply_weights_dev(lane.ctx, w, b.d_feat.p, D, g->F[(size_t)s * L + l]->p);
            }
        }
        for (int k = k0; k < k1; ++k) {
            const int st2 = check_status(g->lanes[k].ctx, "get_patch_data (resampling)");  // synchronises the lane
            if (!st) st = st2;
        }
        status[th] = st;
        if (st) message[th] = msm_last_error();  // the error text is per thread
    };
    if (nthreads == 1) {
        drive(0);
    } else {
        std::vector<std::thread> pool;
        for (int th = 1; th < nthreads; ++th) pool.emplace_back(drive, th);
        drive(0);
        for (auto &t : pool) t.join();
    }
    for (int th = 0; th < nthreads; ++th)
        if (status[th]) return fail(status[th], "%s", message[th].c_str());
    return MSM_OK;
}

// The subjects of this rank, pipelined: while the lanes work through subject i, the main stream prepares subject i + 1 (rotations,
// forest) and builds subject i's patch lists (which only need the template and the control grid).
// one pipeline over its share of the subjects: while the batch stream works through subject i, the main stream prepares subject i + 1
static int run_setup_pipe(msm_group *g, msm_group::Pipe &P, int pipe_no, const std::vector<int> &subjects) {
    const int n = (int)subjects.size();
    if (n == 0) return MSM_OK;
    msm_ctx *ctx = P.main;
    (void)hipSetDevice(ctx->device);
    const bool timing = std::getenv("MSMHIP_TIMING") != nullptr;
    static const bool no_batch = [] { const char *e = std::getenv("MSMHIP_GROUP_BATCH"); return e && std::strcmp(e, "off") == 0; }();
    // Two loops over a ring of kStages stages (round 5; until then one iteration = a thread for subject i's per-label work beside the preparation of subject
    // i + 1, joined: either stream idled 0.4 - 0.55 ms per subject for the other one, thread start and join included).  The preparing loop (this thread,
    // the main stream) runs up to kStages subjects ahead of the consuming one (a thread of its own for the whole pipeline, the batch stream).
    constexpr int K = msm_group::kStages;
    std::mutex mu;
    std::condition_variable cv;
    int prepared = 0, consumed = 0;  // subjects whose stage is ready / whose stage is free again
    bool stop = false;
    int st_batch = MSM_OK, st_main = MSM_OK;
    std::string msg_batch;
    std::thread consumer([&] {
        (void)hipSetDevice(ctx->device);
        for (int i = 0; i < n; ++i) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return prepared > i || stop; });
                if (prepared <= i) return;  // the preparing side failed
            }
            const int s = subjects[i];
            msm_group::Stage &cur = P.stage[i % K];
            const auto l0 = std::chrono::steady_clock::now();
            // the subject's patch lists (the range kernel) follow its per-label work on the batch stream, one look at the outcome of both
            const bool batch = cur.forest_ok && !no_batch;
            int st = batch ? stage_batch(g, s, cur, i & 1, P.batch, true) : stage_lanes(g, s, cur);
            if (!st) st = subject_patches(g, s, P.batch.ctx, &P, batch ? "get_patch_data (resampling)" : nullptr);
            if (timing && i < 4)
                fprintf(stderr, "  group set-up, pipeline %d, subject %d: per-label work + patches %.2f ms\n", pipe_no, s,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - l0).count());
            std::lock_guard<std::mutex> lk(mu);
            if (st) {
                st_batch = st, msg_batch = msm_last_error(), stop = true;
                cv.notify_all();
                return;
            }
            g->have_subject[s] = 1;
            consumed = i + 1;
            cv.notify_all();
        }
    });
    for (int i = 0; i < n && !st_main; ++i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return i - consumed < K || stop; });
            if (stop) break;
        }
        const auto m0 = std::chrono::steady_clock::now();
        st_main = stage_prepare(g, subjects[i], P.stage[i % K], ctx);
        if (timing && i < 4)
            fprintf(stderr, "  group set-up, pipeline %d, subject %d: rotations + forest %.2f ms\n", pipe_no, subjects[i],
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m0).count());
        std::lock_guard<std::mutex> lk(mu);
        if (st_main) stop = true;
        else prepared = i + 1;
        cv.notify_all();
    }
    consumer.join();
    if (st_batch) return fail(st_batch, "%s", msg_batch.c_str());
    return st_main;
}

// The subjects of this rank, pipelined.  Within a pipeline the batch stream works through subject i while the main stream prepares subject i + 1
// (rotations, forest); round 4: TWO pipelines take alternate subjects (streams, forests and scratch of their own, a host thread each): either is a
// chain of small dependent launches that leaves the GPU mostly idle (2.35 -> 1.5 ms per ico6 subject).  MSMHIP_GROUP_PIPES=1: one.
static int group_setup_pipeline(msm_group *g, const int32_t *subjects, int n) {
    if (n <= 0) return MSM_OK;
    msm_ctx *ctx = g->ctx;
    // the template's search structure and adjacency, and the data meshes' adjacency lists, are read by every pipeline: complete before they start
    int st = ensure_tree(g->tmpl);
    if (st) return st;
    // the template's direction table (a simple surface has one; built once per template content and kept, api.cpp: ensure_rays): the reverse queries of every
    // subject and label run through it (stage_batch)
    st = ensure_rays(g->tmpl, true);
    if (st) return st;
    st = ensure_adjacency_dev(g->tmpl);
    if (st) return st;
    {   // the grid the subjects' range tests take their candidates from (subject_patches): cells of about a sixteenth of the sphere's diameter
        const int G = g->tmpl->V > 60000 ? 64 : 32;
        const size_t cells = (size_t)G * G * G;
        MSM_HIP(g->d_rg_start.ensure(cells + 1));
        MSM_HIP(g->d_rg_cursor.ensure(cells));
        MSM_HIP(g->d_rg_ids.ensure((size_t)g->tmpl->V));
        MSM_HIP(g->d_rg_bad.ensure(1));
        MSM_HIP(g->d_rg_tmp.ensure(cells / 4096 + 2));
        const double origin = -1.01 * kRad, inv_h = (double)G / (2.02 * kRad);
        st = launch_range_grid_build(ctx, g->tmpl->d_xyz, g->tmpl->V, G, origin, inv_h, g->d_rg_start.p, g->d_rg_cursor.p, g->d_rg_ids.p, g->d_rg_bad.p, g->d_rg_tmp.p);
        if (st) return st;
        g->rgrid.start = g->d_rg_start.p, g->rgrid.ids = g->d_rg_ids.p, g->rgrid.bad = g->d_rg_bad.p, g->rgrid.G = G, g->rgrid.origin = origin, g->rgrid.inv_h = inv_h;
        g->rgrid_valid = true;
    }
    struct GridScope {  // the template may move before anybody else asks for patch lists
        msm_group *g;
        ~GridScope() { g->rgrid_valid = false; }
    } grid_scope{g};
    for (int i = 0; i < n; ++i) {
        if (!g->data[subjects[i]]) return fail(MSM_ERR_STATE, "msm_group: subject %d has no data", subjects[i]);
        st = ensure_adjacency_dev(g->data[subjects[i]]);
        if (st) return st;
    }
    MSM_TRY(ctx_sync(ctx));
    static const int pipes_env = [] { const char *e = std::getenv("MSMHIP_GROUP_PIPES"); return e ? std::max(1, std::min(msm_group::kMaxPipes, std::atoi(e))) : 2; }();
    const int npipes = std::max(1, std::min(pipes_env, n / 2));
    g->pipe[0].main = ctx;
    for (int k = 1; k < npipes; ++k)
        if (!g->pipe[k].main) {
            g->pipe[k].main = msm_ctx_create(ctx->device);
            if (!g->pipe[k].main) return MSM_ERR_HIP;
            g->pipe[k].own_main = true;
        }
    for (int k = 0; k < npipes; ++k)
        if (!g->pipe[k].batch.ctx) {
            g->pipe[k].batch.ctx = msm_ctx_create(ctx->device);
            if (!g->pipe[k].batch.ctx) return MSM_ERR_HIP;
        }
    std::vector<int> share[msm_group::kMaxPipes];
    for (int i = 0; i < n; ++i) share[i % npipes].push_back(subjects[i]);
    if (npipes == 1) return run_setup_pipe(g, g->pipe[0], 0, share[0]);
    int stk[msm_group::kMaxPipes] = {MSM_OK, MSM_OK, MSM_OK, MSM_OK};
    std::string msgk[msm_group::kMaxPipes];
    std::vector<std::thread> others;
    for (int k = 1; k < npipes; ++k)
        others.emplace_back([&, k] {
            stk[k] = run_setup_pipe(g, g->pipe[k], k, share[k]);
            if (stk[k]) msgk[k] = msm_last_error();
        });
    stk[0] = run_setup_pipe(g, g->pipe[0], 0, share[0]);
    if (stk[0]) msgk[0] = msm_last_error();
    for (auto &t : others) t.join();
    for (int k = 0; k < npipes; ++k)
        if (stk[k]) return fail(stk[k], "%s", msgk[k].c_str());
    return MSM_OK;
}

// DiscreteGroupModel::get_patch_data for one subject (M/DiscreteGroupModel.cpp:88-121), the comparison path of
// MSMHIP_OCTREE=host / MSMHIP_SURGERY=host (round 1's division of labour; the default is group_setup_pipeline above).  Per label:
// rotate the data mesh, build its octree, resample the features to the template with adaptive barycentric weights.  The
// rotations and the 2 x N nearest-triangle queries run on the GPU; the octree builds and the weight-list surgery are host
// work here, independent per label, spread over the host cores in two parallel phases around the GPU phase.
int group_subject_setup(msm_group *g, int s) {
    if (!g->data[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d has no data", s);
    msm_ctx *ctx = g->ctx;
    const int L = g->L, D = g->D, Vt = g->tmpl->V;
    msm_mesh *dm = g->data[s], *sm = g->scratch[s];
    const int V = dm->V, T = dm->T;
    const int workers = host_workers();
    const bool timing = std::getenv("MSMHIP_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  group set-up, subject %d: %s %.1f ms (%d workers)\n", s, what, std::chrono::duration<double, std::milli>(now - tick).count(), workers);
        tick = now;
    };
    // phase 1 (GPU): the L rotated meshes, side by side in one 3 x (L * V) array (x of label 0, x of label 1, ..., y of label 0, ...)
    // that stays in HBM: it is also the query set of the reverse searches of phase 3
    std::vector<std::vector<double>> rotated(L);
    const size_t LV = (size_t)L * V;
    DevBuf<double> &d_rot = g->d_rotated;
    {
        MSM_HIP(d_rot.ensure(3 * LV));
        int st = rotate_subject(g, dm, ctx, g->pipe[0].stage[0].rot9, g->pipe[0].stage[0].d_rot9, d_rot.p);
        if (st) return st;
        void *pin = nullptr;
        st = ctx_io_pinned(ctx, sizeof(double) * 3 * LV, &pin);
        if (st) return st;
        MSM_HIP(hipMemcpyAsync(pin, d_rot.p, sizeof(double) * 3 * LV, hipMemcpyDeviceToHost, ctx->stream));
        st = check_status(ctx, "get_patch_data (rotation)");
        if (st) return st;
        const double *big = static_cast<const double *>(pin);
        parallel_for(L, workers, [&](int l) {
            rotated[l].resize(3 * (size_t)V);
            for (int a = 0; a < 3; ++a) std::memcpy(rotated[l].data() + (size_t)a * V, big + a * LV + (size_t)l * V, sizeof(double) * (size_t)V);
        });
        rotated[0] = dm->xyz;  // the same numbers; kept as the mesh's own copy
    }
    lap("rotations");
    // phase 2: vertex areas of the rotated meshes (host threads), and -- for data meshes below the size at which the tree is
    // built on the GPU -- their octrees
    const bool gpu_trees = mesh_tree_on_gpu(sm);
    std::vector<FlatOctree> trees(gpu_trees ? 0 : L);
    std::vector<std::vector<double>> oldA(L);
    const Adjacency &adj = mesh_adjacency(sm);
    std::vector<double> newA;
    vertex_areas_of(g->tmpl->xyz.data(), g->tmpl->tri.data(), g->tmpl->V, g->tmpl->T, mesh_adjacency(g->tmpl), newA);
    parallel_for(L, workers, [&](int l) {
        if (!gpu_trees) build_octree(rotated[l].data(), sm->tri.data(), V, T, trees[l]);
        vertex_areas_of(rotated[l].data(), sm->tri.data(), V, T, adj, oldA[l]);
    });
    lap(gpu_trees ? "vertex areas" : "octrees");
    // phase 3 (GPU): forward and reverse queries of metric_resample(rotated_mesh, target_space); the tree of each rotated mesh
    // is built in HBM from the coordinates of phase 1 (octree_kernels.hip), or installed from the host build
    std::vector<AdaptiveQueries> queries(L);
    double t_install = 0, t_query = 0;
    for (int l = 0; l < L; ++l) {
        const auto t0 = std::chrono::steady_clock::now();
        int st;
        if (gpu_trees) {
            for (int a = 0; a < 3; ++a)
                MSM_HIP(hipMemcpyAsync(sm->d_xyz + (size_t)a * V, d_rot.p + a * LV + (size_t)l * V, sizeof(double) * (size_t)V, hipMemcpyDeviceToDevice, ctx->stream));
            sm->tree_valid = false;  // rebuilt by the first query below
            st = MSM_OK;
        } else {
            st = install_coords_and_tree(sm, rotated[l].data(), std::move(trees[l]));
        }
        if (st) return st;
        const auto t1 = std::chrono::steady_clock::now();
        st = adaptive_queries(sm, g->tmpl, false, queries[l], 1);  // forward: template vertices in this label's tree
        if (st) return st;
        const auto t2 = std::chrono::steady_clock::now();
        t_install += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t_query += std::chrono::duration<double, std::milli>(t2 - t1).count();
    }
    {
        // reverse: the vertices of all L rotated meshes in the template's tree, one launch over the array of phase 1
        std::vector<int> rvid(3 * LV);
        std::vector<double> rw(3 * LV);
        const auto t0 = std::chrono::steady_clock::now();
        int st = query_host(g->tmpl, nullptr, (int)LV, nullptr, rvid.data(), rw.data(), MSM_WEIGHTS_PROJECTED, "adaptive weights (reverse)", d_rot.p);
        if (st) return st;
        parallel_for(L, workers, [&](int l) {
            for (int a = 0; a < 3; ++a) {
                std::memcpy(queries[l].rvid.data() + (size_t)a * V, rvid.data() + a * LV + (size_t)l * V, sizeof(int) * (size_t)V);
                std::memcpy(queries[l].rw.data() + (size_t)a * V, rw.data() + a * LV + (size_t)l * V, sizeof(double) * (size_t)V);
            }
        });
        t_query += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (timing) fprintf(stderr, "    install %.1f ms, queries %.1f ms\n", t_install, t_query);
    lap("uploads + queries");
    // phase 4 (host threads): weight lists and the resampled features
    std::vector<std::vector<double>> resampled(L);
    parallel_for(L, workers, [&](int l) {
        std::vector<int32_t> rp, col;
        std::vector<double> val;
        adaptive_surgery(queries[l], V, Vt, oldA[l], newA, nullptr, rp, col, val);
        std::vector<double> &out = resampled[l];
        out.resize((size_t)D * Vt);
        for (int d = 0; d < D; ++d)
            for (int k = 0; k < Vt; ++k) {
                double acc = 0.0;
                for (int e = rp[k]; e < rp[k + 1]; ++e) acc += g->feat[s][(size_t)d * V + col[e]] * val[e];
                out[(size_t)d * Vt + k] = acc;
            }
    });
    {
        int st = subject_feature_slab(g, s, (size_t)D * Vt);
        if (st) return st;
    }
    for (int l = 0; l < L; ++l) {
        int st = upload_staged(ctx, g->F[(size_t)s * L + l]->p, resampled[l].data(), resampled[l].size() * sizeof(double));
        if (st) return st;
    }
    MSM_TRY(ctx_sync(ctx));
    lap("weights + resample");
    int st = subject_patches(g, s);
    if (st) return st;
    lap("patches");
    g->have_subject[s] = 1;
    return MSM_OK;
}

}  // namespace

extern "C" {

int msm_group_setup_subjects(msm_group *g, const int32_t *subjects, int32_t n) {
    if (!g || (n > 0 && !subjects) || n < 0) return fail(MSM_ERR_INVALID, "msm_group_setup_subjects: bad arguments");
    g->ready = false;
    g->common_ready = false;
    g->drop_kept();
    int st = group_common_setup(g);
    if (st) return st;
    for (int i = 0; i < n; ++i)
        if (subjects[i] < 0 || subjects[i] >= g->S) return fail(MSM_ERR_INVALID, "subject %d out of range", subjects[i]);
    if (group_device_path()) return group_setup_pipeline(g, subjects, n);
    for (int i = 0; i < n; ++i) {
        st = group_subject_setup(g, subjects[i]);
        if (st) return st;
    }
    return MSM_OK;
}

// further subjects of this rank after msm_group_setup_subjects (same control grids, labels and data): the set-up in chunks, so that the exchange of
// one chunk (an all-gather on another stream) runs while the next is set up
int msm_group_setup_more_subjects(msm_group *g, const int32_t *subjects, int32_t n) {
    if (!g || (n > 0 && !subjects) || n < 0) return fail(MSM_ERR_INVALID, "msm_group_setup_more_subjects: bad arguments");
    if (!g->common_ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup_subjects() must be called first (an empty list is fine)");
    g->ready = false;
    for (int i = 0; i < n; ++i)
        if (subjects[i] < 0 || subjects[i] >= g->S) return fail(MSM_ERR_INVALID, "subject %d out of range", subjects[i]);
    if (group_device_path()) return group_setup_pipeline(g, subjects, n);
    for (int i = 0; i < n; ++i) {
        int st = group_subject_setup(g, subjects[i]);
        if (st) return st;
    }
    return MSM_OK;
}

// the host copy of a subject's row offsets: there after a set-up on this rank or a single import, fetched when first asked for after a batched import
static int fetch_host_pptr(msm_group *g, int s) {
    if (!g->h_pptr[s].empty()) return MSM_OK;
    if (!g->have_subject[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d is neither set up nor imported", s);
    g->h_pptr[s].resize((size_t)g->N * g->L + 1);
    MSM_TRY(g->pptr[s]->download(g->h_pptr[s].data(), g->h_pptr[s].size(), g->ctx));
    MSM_TRY(ctx_sync(g->ctx));
    return MSM_OK;
}

// the host copy of a subject's index list is fetched when first asked for (set up or imported on the device)
static int fetch_host_pidx(msm_group *g, int s) {
    if (!g->h_pidx[s].empty() || g->h_pptr[s].empty() || g->h_pptr[s].back() <= 0) return MSM_OK;
    g->h_pidx[s].resize((size_t)g->h_pptr[s].back());
    MSM_TRY(g->pidx[s]->download(g->h_pidx[s].data(), g->h_pidx[s].size(), g->ctx));
    MSM_TRY(ctx_sync(g->ctx));
    return MSM_OK;
}

int msm_group_export_subject(msm_group *g, int32_t s, double *F, int32_t *pptr, int32_t *pidx, int64_t cap, int64_t *npidx) {
    if (!g || s < 0 || s >= g->S) return fail(MSM_ERR_INVALID, "msm_group_export_subject: bad arguments");
    if (!g->common_ready || !g->have_subject[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d has not been set up on this rank", s);
    const size_t per = (size_t)g->D * g->tmpl->V;
    {
        int st = fetch_host_pptr(g, s);
        if (st) return st;
    }
    if (npidx) *npidx = (int64_t)g->h_pptr[s].back();
    if (pidx) {
        int st = fetch_host_pidx(g, s);
        if (st) return st;
    }
    if (F) {
        for (int l = 0; l < g->L; ++l) MSM_TRY(g->F[(size_t)s * g->L + l]->download(F + per * l, per, g->ctx));
        MSM_TRY(ctx_sync(g->ctx));
    }
    if (pptr) std::copy(g->h_pptr[s].begin(), g->h_pptr[s].end(), pptr);
    if (pidx) {
        if ((int64_t)g->h_pidx[s].size() > cap) return fail(MSM_ERR_CAPACITY, "patch index buffer too small");
        std::copy(g->h_pidx[s].begin(), g->h_pidx[s].end(), pidx);
    }
    return MSM_OK;
}

int msm_group_import_subject(msm_group *g, int32_t s, const double *F, const int32_t *pptr, const int32_t *pidx, int64_t npidx) {
    if (!g || !F || !pptr || (!pidx && npidx > 0) || s < 0 || s >= g->S || npidx < 0) return fail(MSM_ERR_INVALID, "msm_group_import_subject: bad arguments");
    if (!g->common_ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup_subjects() must be called first (an empty list is fine)");
    msm_ctx *ctx = g->ctx;
    const size_t per = (size_t)g->D * g->tmpl->V;
    const size_t M = (size_t)g->N * g->L;
    // the arrays come from another rank: nothing of them is trusted before it indexes device or host memory
    if (pptr[0] != 0 || pptr[M] != npidx) return fail(MSM_ERR_INVALID, "patch CSR of subject %d is inconsistent", s);
    for (size_t r = 0; r < M; ++r)
        if (pptr[r + 1] < pptr[r]) return fail(MSM_ERR_INVALID, "patch CSR of subject %d: row %zu has a negative length", s, r);
    const int32_t Vt = g->tmpl->V;
    for (int64_t j = 0; j < npidx; ++j)
        if (pidx[j] < 0 || pidx[j] >= Vt) return fail(MSM_ERR_INVALID, "patch CSR of subject %d: template vertex id %d out of range [0,%d)", s, pidx[j], Vt);
    {
        int st = subject_feature_slab(g, s, per);
        if (st) return st;
    }
    for (int l = 0; l < g->L; ++l) MSM_TRY(upload_staged(ctx, g->F[(size_t)s * g->L + l]->p, F + per * l, sizeof(double) * per));
    g->h_pptr[s].assign(pptr, pptr + M + 1);
    g->h_pidx[s].assign(pidx, pidx + npidx);
    MSM_TRY(g->pptr[s]->upload(g->h_pptr[s].data(), M + 1, ctx));
    MSM_TRY(g->pidx[s]->upload_vec(g->h_pidx[s], ctx));
    MSM_TRY(ctx_sync(ctx));
    g->have_subject[s] = 2;  // (2: imported)
    return MSM_OK;
}

// The same exchange without the host in between: the caller's buffers are DEVICE memory on this context's GPU (e.g. torch
// tensors about to go through an RCCL all-gather, or just out of one).
int msm_group_export_subject_dev(msm_group *g, int32_t s, double *F_dev, int32_t *pptr_dev, int32_t *pidx_dev, int64_t cap, int64_t *npidx) {
    if (!g || s < 0 || s >= g->S) return fail(MSM_ERR_INVALID, "msm_group_export_subject_dev: bad arguments");
    if (!g->common_ready || !g->have_subject[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d has not been set up on this rank", s);
    msm_ctx *ctx = g->ctx;
    const size_t per = (size_t)g->D * g->tmpl->V, M = (size_t)g->N * g->L;
    {
        int st = fetch_host_pptr(g, s);
        if (st) return st;
    }
    const int64_t n = (int64_t)g->h_pptr[s][M];
    if (npidx) *npidx = n;
    if (F_dev)
        for (int l = 0; l < g->L; ++l)
            MSM_HIP(hipMemcpyAsync(F_dev + per * l, g->F[(size_t)s * g->L + l]->p, sizeof(double) * per, hipMemcpyDeviceToDevice, ctx->stream));
    if (pptr_dev) MSM_HIP(hipMemcpyAsync(pptr_dev, g->pptr[s]->p, sizeof(int32_t) * (M + 1), hipMemcpyDeviceToDevice, ctx->stream));
    if (pidx_dev) {
        if (n > cap) return fail(MSM_ERR_CAPACITY, "patch index buffer too small");
        if (n > 0) MSM_HIP(hipMemcpyAsync(pidx_dev, g->pidx[s]->p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    MSM_TRY(ctx_sync(ctx));  // the buffers may go straight into a collective on another stream
    return MSM_OK;
}

namespace {
__global__ void k_check_patch_csr(const int32_t *__restrict__ pptr, size_t M, const int32_t *__restrict__ pidx, int64_t npidx, int32_t Vt, int *bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && (pptr[0] != 0 || pptr[M] != npidx)) atomicOr(bad, 1);
    if (i < M && pptr[i + 1] < pptr[i]) atomicOr(bad, 2);
    if ((int64_t)i < npidx && (pidx[i] < 0 || pidx[i] >= Vt)) atomicOr(bad, 4);
}
}  // namespace

int msm_group_import_subject_dev(msm_group *g, int32_t s, const double *F_dev, const int32_t *pptr_dev, const int32_t *pidx_dev, int64_t npidx) {
    if (!g || !F_dev || !pptr_dev || (!pidx_dev && npidx > 0) || s < 0 || s >= g->S || npidx < 0) return fail(MSM_ERR_INVALID, "msm_group_import_subject_dev: bad arguments");
    if (!g->common_ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup_subjects() must be called first (an empty list is fine)");
    msm_ctx *ctx = g->ctx;
    const size_t per = (size_t)g->D * g->tmpl->V, M = (size_t)g->N * g->L;
    // the arrays come from another rank: checked (on the device, where they are) before any kernel indexes with them
    DevBuf<int> bad;
    MSM_HIP(bad.zero(1, ctx->stream));
    const size_t span = std::max<size_t>(M + 1, (size_t)npidx);
    hipLaunchKernelGGL(k_check_patch_csr, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, ctx->stream, pptr_dev, M, pidx_dev, npidx, g->tmpl->V, bad.p);
    MSM_HIP(hipGetLastError());
    int hbad = 0;
    MSM_TRY(stage_d2h(ctx, &hbad, bad.p, sizeof(int)));
    MSM_TRY(ctx_sync(ctx));
    if (hbad) return fail(MSM_ERR_INVALID, "patch CSR of subject %d is inconsistent (code %d: 1 ends, 2 row lengths, 4 vertex ids)", s, hbad);
    {
        int st = subject_feature_slab(g, s, per);
        if (st) return st;
    }
    for (int l = 0; l < g->L; ++l)
        MSM_HIP(hipMemcpyAsync(g->F[(size_t)s * g->L + l]->p, F_dev + per * l, sizeof(double) * per, hipMemcpyDeviceToDevice, ctx->stream));
    MSM_HIP(g->pptr[s]->ensure(M + 1));
    MSM_HIP(hipMemcpyAsync(g->pptr[s]->p, pptr_dev, sizeof(int32_t) * (M + 1), hipMemcpyDeviceToDevice, ctx->stream));
    MSM_HIP(g->pidx[s]->ensure(std::max<size_t>((size_t)npidx, 1)));
    if (npidx > 0) MSM_HIP(hipMemcpyAsync(g->pidx[s]->p, pidx_dev, sizeof(int32_t) * (size_t)npidx, hipMemcpyDeviceToDevice, ctx->stream));
    g->h_pptr[s].resize(M + 1);  // the row lengths are host knowledge too (largest patch, msm_group_patch)
    MSM_TRY(stage_d2h(ctx, g->h_pptr[s].data(), pptr_dev, sizeof(int32_t) * (M + 1)));
    g->h_pidx[s].clear();        // fetched on demand by msm_group_patch
    MSM_TRY(ctx_sync(ctx));
    g->have_subject[s] = 2;
    return MSM_OK;
}

namespace {
// the range checks of msm_group_import_subject_dev for n subjects at once (blockIdx.y), and what msm_group_finalize wants to know of their patches
__global__ void k_check_patch_csr_batch(const int32_t *__restrict__ pptr, int64_t pptr_stride, size_t M, const int32_t *__restrict__ pidx, int64_t pidx_stride,
                                        const int64_t *__restrict__ npidx, int32_t Vt, int small_len, int *__restrict__ out /* n x 4: bad, largest, small, - */) {
    const int k = blockIdx.y;
    const int32_t *pp = pptr + (size_t)k * pptr_stride, *pi = pidx + (size_t)k * pidx_stride;
    const int64_t np = npidx[k];
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M || (int64_t)i < np; i += (size_t)gridDim.x * blockDim.x) {
        if (i == 0 && (pp[0] != 0 || pp[M] != np)) bad |= 1;
        if (i < M) {
            const int len = pp[i + 1] - pp[i];
            if (len < 0) bad |= 2;
            if (len > out[4 * k + 1]) atomicMax(&out[4 * k + 1], len);
            const unsigned long long sm = __ballot(len >= 0 && len <= small_len);
            if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&out[4 * k + 2], (int)__popcll(sm));
        }
        if ((int64_t)i < np && (pi[i] < 0 || pi[i] >= Vt)) bad |= 4;
    }
    if (bad) atomicOr(&out[4 * k], bad);
}
}  // namespace

// n subjects at once into / out of strided device buffers -- the send and receive buffers of ONE all-gather: subject subjects[k]'s arrays start at
// F_dev + k * F_stride (doubles), pptr_dev + k * pptr_stride and pidx_dev + k * pidx_stride (int32).  One synchronisation per call instead of two per
// subject (56 imports of an 8-rank, 64-subject group: 9.2 -> under 1 ms, tools/time_group_rank.py).
int msm_group_export_subjects_dev(msm_group *g, const int32_t *subjects, int32_t n, double *F_dev, int64_t F_stride, int32_t *pptr_dev, int64_t pptr_stride,
                                  int32_t *pidx_dev, int64_t pidx_stride, int64_t *npidx) {
    if (!g || (n > 0 && (!subjects || !F_dev || !pptr_dev || !pidx_dev)) || n < 0) return fail(MSM_ERR_INVALID, "msm_group_export_subjects_dev: bad arguments");
    msm_ctx *ctx = g->ctx;
    const size_t per = (size_t)g->D * g->tmpl->V, M = (size_t)g->N * g->L;
    if ((int64_t)(per * g->L) > F_stride || (int64_t)(M + 1) > pptr_stride) return fail(MSM_ERR_CAPACITY, "msm_group_export_subjects_dev: strides too small");
    for (int k = 0; k < n; ++k) {
        const int s = subjects[k];
        if (s < 0 || s >= g->S) return fail(MSM_ERR_INVALID, "subject %d out of range", s);
        if (!g->common_ready || !g->have_subject[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d has not been set up on this rank", s);
        int st = fetch_host_pptr(g, s);
        if (st) return st;
        const int64_t cnt = (int64_t)g->h_pptr[s][M];
        if (npidx) npidx[k] = cnt;
        if (cnt > pidx_stride) return fail(MSM_ERR_CAPACITY, "patch index buffer too small");
        MSM_HIP(hipMemcpyAsync(F_dev + (size_t)k * F_stride, g->Fslab[s]->p, sizeof(double) * per * g->L, hipMemcpyDeviceToDevice, ctx->stream));
        MSM_HIP(hipMemcpyAsync(pptr_dev + (size_t)k * pptr_stride, g->pptr[s]->p, sizeof(int32_t) * (M + 1), hipMemcpyDeviceToDevice, ctx->stream));
        if (cnt > 0) MSM_HIP(hipMemcpyAsync(pidx_dev + (size_t)k * pidx_stride, g->pidx[s]->p, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToDevice, ctx->stream));
    }
    MSM_TRY(ctx_sync(ctx));  // the buffers may go straight into a collective on another stream
    return MSM_OK;
}

int msm_group_import_subjects_dev(msm_group *g, const int32_t *subjects, int32_t n, const double *F_dev, int64_t F_stride, const int32_t *pptr_dev,
                                  int64_t pptr_stride, const int32_t *pidx_dev, int64_t pidx_stride, const int64_t *npidx) {
    if (!g || n < 0 || (n > 0 && (!subjects || !F_dev || !pptr_dev || !pidx_dev || !npidx))) return fail(MSM_ERR_INVALID, "msm_group_import_subjects_dev: bad arguments");
    if (!g->common_ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup_subjects() must be called first (an empty list is fine)");
    if (n == 0) return MSM_OK;
    msm_ctx *ctx = g->ctx;
    const size_t per = (size_t)g->D * g->tmpl->V, M = (size_t)g->N * g->L;
    if ((int64_t)(per * g->L) > F_stride || (int64_t)(M + 1) > pptr_stride) return fail(MSM_ERR_INVALID, "msm_group_import_subjects_dev: strides too small");
    int64_t most = (int64_t)M + 1;
    for (int k = 0; k < n; ++k) {
        if (subjects[k] < 0 || subjects[k] >= g->S) return fail(MSM_ERR_INVALID, "subject %d out of range", subjects[k]);
        if (npidx[k] < 0 || npidx[k] > pidx_stride) return fail(MSM_ERR_INVALID, "patch CSR of subject %d: %lld entries do not fit the buffer", subjects[k], (long long)npidx[k]);
        most = std::max(most, npidx[k]);
    }
    // the arrays come from other ranks: checked (on the device, where they are) before any kernel indexes with them
    DevBuf<int64_t> d_np;
    DevBuf<int> d_out;
    MSM_TRY(d_np.upload(npidx, (size_t)n, ctx));
    MSM_HIP(d_out.zero(4 * (size_t)n, ctx->stream));
    const unsigned gx = (unsigned)std::min<int64_t>((most + 255) / 256, 4096);
    hipLaunchKernelGGL(k_check_patch_csr_batch, dim3(gx, (unsigned)n), dim3(256), 0, ctx->stream, pptr_dev, pptr_stride, M, pidx_dev, pidx_stride, d_np.p, g->tmpl->V,
                       kPairSmallPatch, d_out.p);
    MSM_HIP(hipGetLastError());
    std::vector<int> out(4 * (size_t)n);
    MSM_TRY(d_out.download(out.data(), out.size(), ctx));
    MSM_TRY(ctx_sync(ctx));
    for (int k = 0; k < n; ++k)
        if (out[4 * (size_t)k])
            return fail(MSM_ERR_INVALID, "patch CSR of subject %d is inconsistent (code %d: 1 ends, 2 row lengths, 4 vertex ids)", subjects[k], out[4 * (size_t)k]);
    for (int k = 0; k < n; ++k) {
        const int s = subjects[k];
        int st = subject_feature_slab(g, s, per);
        if (st) return st;
        MSM_HIP(hipMemcpyAsync(g->Fslab[s]->p, F_dev + (size_t)k * F_stride, sizeof(double) * per * g->L, hipMemcpyDeviceToDevice, ctx->stream));
        MSM_HIP(g->pptr[s]->ensure(M + 1));
        MSM_HIP(hipMemcpyAsync(g->pptr[s]->p, pptr_dev + (size_t)k * pptr_stride, sizeof(int32_t) * (M + 1), hipMemcpyDeviceToDevice, ctx->stream));
        MSM_HIP(g->pidx[s]->ensure(std::max<size_t>((size_t)npidx[k], 1)));
        if (npidx[k] > 0)
            MSM_HIP(hipMemcpyAsync(g->pidx[s]->p, pidx_dev + (size_t)k * pidx_stride, sizeof(int32_t) * (size_t)npidx[k], hipMemcpyDeviceToDevice, ctx->stream));
        g->h_pptr[s].clear();  // fetched on demand (msm_group_patch, a re-export)
        g->h_pidx[s].clear();
        g->imp_stat[s].npidx = npidx[k];
        g->imp_stat[s].largest = out[4 * (size_t)k + 1];
        g->imp_stat[s].small = out[4 * (size_t)k + 2];
    }
    MSM_TRY(ctx_sync(ctx));  // the caller may reuse its buffers
    for (int k = 0; k < n; ++k) g->have_subject[subjects[k]] = 2;
    return MSM_OK;
}

}  // extern "C"

namespace {
// The patch entries' values beside their ids and the patch directory (GroupArgs::pval / dir) for the label steps of the pairs [pair0, pair1): the whole list
// writes every patch (3.2 GB at S = 64, ico6 / ico4: 2.3 ms), a rank's slice only the patches of the nodes its pairs touch.
int ensure_patch_values(msm_group *g, int64_t pair0, int64_t pair1) {
    if (g->pval_ready && g->pval_p0 <= pair0 && g->pval_p1 >= pair1) return MSM_OK;
    msm_ctx *ctx = g->ctx;
    const bool ready_before = g->pval_ready;
    g->pval_ready = false;
    GroupArgs a;
    int st = group_args(g, a);
    if (st) return st;
    const bool all = pair0 == 0 && pair1 == g->npairs;
    if (!all) {
        MSM_HIP(g->d_node_flags.zero((size_t)g->S * g->N, ctx->stream));
        st = launch_group_mark_nodes(ctx, g->d_pairs.p, pair0, pair1, g->d_node_flags.p);
        if (st) return st;
    }
    st = launch_group_patch_values(ctx, a, g->d_pvalp.p, all ? nullptr : g->d_node_flags.p);
    if (!st && !ready_before) st = launch_group_patch_dir(ctx, a, g->d_pvalp.p, g->d_patch_dir.p);  // (every patch's record: it does not depend on the slice)
    if (!st) st = ctx_sync(ctx);
    if (st) return st;
    g->pval_ready = true;
    g->pval_p0 = pair0, g->pval_p1 = pair1;
    return MSM_OK;
}
}  // namespace

extern "C" {

int msm_group_finalize(msm_group *g) {
    if (!g) return fail(MSM_ERR_INVALID, "null group");
    if (!g->common_ready) return fail(MSM_ERR_STATE, "msm_group: nothing has been set up");
    for (int s = 0; s < g->S; ++s)
        if (!g->have_subject[s]) return fail(MSM_ERR_STATE, "msm_group: subject %d is neither set up nor imported", s);
    msm_ctx *ctx = g->ctx;
    const int S = g->S, L = g->L;
    std::vector<const double *> Fp((size_t)S * L);
    for (size_t k = 0; k < Fp.size(); ++k) Fp[k] = g->F[k]->p;
    std::vector<const int *> pp(S), pi(S);
    for (int s = 0; s < S; ++s) {
        pp[s] = g->pptr[s]->p;
        pi[s] = g->pidx[s]->p;
    }
    MSM_TRY(g->d_Fp.upload(Fp.data(), Fp.size(), ctx));
    MSM_TRY(g->d_pptrp.upload(pp.data(), pp.size(), ctx));
    MSM_TRY(g->d_pidxp.upload(pi.data(), pi.size(), ctx));
    MSM_TRY(ctx_sync(ctx));
    g->drop_kept();  // new patches: nothing kept from earlier label steps applies
    g->patch_max = 0;
    int64_t npatch = 0, nsmall = 0;
    for (int s = 0; s < S; ++s) {
        if (g->h_pptr[s].empty()) {  // imported in a batch: the check kernel counted
            g->patch_max = std::max(g->patch_max, g->imp_stat[s].largest);
            npatch += (int64_t)g->N * L;
            nsmall += g->imp_stat[s].small;
            continue;
        }
        for (size_t k = 0; k + 1 < g->h_pptr[s].size(); ++k) {
            const int n = g->h_pptr[s][k + 1] - g->h_pptr[s][k];
            g->patch_max = std::max(g->patch_max, n);
            ++npatch;
            nsmall += n <= kPairSmallPatch;
        }
    }
    // a quarter wavefront per pair cost when (nearly) all patches fit its registers: four queries share a wavefront's instruction stream and latency
    // instead of two (label step 12.5 -> 9.5 ms at ico6 / ico4, patches of ~65 entries); the others take the general path, at half the lanes
    g->pair_lanes = (npatch > 0 && 10 * nsmall >= 9 * npatch) ? 16 : 32;
    if (const char *e = std::getenv("MSMHIP_GROUP_PAIR_LANES")) g->pair_lanes = std::atoi(e) == 16 ? 16 : 32;
    // The patch entries' values beside their ids (GroupArgs::pval), for the register path of k_group_pairwise (D <= 2): one launch over all subjects, whether
    // they were set up here or imported -- 3.2 GB written at S = 64, ico6 / ico4 (0.5 ms), and the label steps' value gathers by vertex id become a coalesced
    // read of patch A and reads of one 1 KB window of patch B.  MSMHIP_GROUP_PVAL=off: the maps are gathered from, as until round 5.
    static const bool pval_off = [] { const char *e = std::getenv("MSMHIP_GROUP_PVAL"); return e && std::strcmp(e, "off") == 0; }();
    g->pval_ready = g->pval_deferred = false;
    g->pval_p0 = g->pval_p1 = -1;
    if (!pval_off && g->D <= 2) {
        g->pval.resize(S);
        std::vector<double *> pv(S);
        bool imported = false;
        for (int s = 0; s < S; ++s) {
            const size_t n = g->h_pptr[s].empty() ? (size_t)g->imp_stat[s].npidx : (size_t)g->h_pptr[s].back();
            if (!g->pval[s]) g->pval[s].reset(new DevBuf<double>());
            MSM_HIP(g->pval[s]->ensure(std::max<size_t>((size_t)g->D * n, 2)));
            pv[s] = g->pval[s]->p;
            imported = imported || g->have_subject[s] == 2;
        }
        MSM_TRY(g->d_pvalp.upload(pv.data(), pv.size(), ctx));
        if (g->d_patch_dir.ensure((size_t)S * g->N * L) != hipSuccess) return fail(MSM_ERR_HIP, "device allocation of the patch directory failed");
        g->ready = true;  // (group_args checks it)
        if (imported) {
            g->pval_deferred = true;  // a rank of a sharded run: written for the nodes of its slice when the slice is first evaluated
            MSM_TRY(ctx_sync(ctx));
        } else {
            const int st = ensure_patch_values(g, 0, g->npairs);
            if (st) {
                g->ready = false;
                return st;
            }
        }
    }
    if (std::getenv("MSMHIP_TIMING"))
        fprintf(stderr, "  group patches: %lld, %.1f %% of them with at most %d entries, largest %d -> %d lanes per pair cost\n", (long long)npatch,
                npatch ? 100.0 * (double)nsmall / (double)npatch : 0.0, kPairSmallPatch, g->patch_max, g->pair_lanes);
    g->ready = true;
    return MSM_OK;
}

int msm_group_setup(msm_group *g) {
    if (!g) return fail(MSM_ERR_INVALID, "null group");
    std::vector<int32_t> all(g->S);
    for (int s = 0; s < g->S; ++s) all[s] = s;
    int st = msm_group_setup_subjects(g, all.data(), g->S);
    if (st) return st;
    return msm_group_finalize(g);
}

int msm_group_sizes(msm_group *g, int32_t *nodes, int32_t *pairs, int32_t *triplets) {
    if (!g) return fail(MSM_ERR_INVALID, "null group");
    if (nodes) *nodes = g->S * g->N;
    if (pairs) *pairs = g->N * g->S * (g->S - 1) / 2;
    if (triplets) *triplets = g->S * g->Tc;
    return MSM_OK;
}

int msm_group_dims(msm_group *g, int32_t *S, int32_t *N, int32_t *L, int32_t *D, int32_t *Vt) {
    if (!g) return fail(MSM_ERR_INVALID, "null group");
    if (S) *S = g->S;
    if (N) *N = g->N;
    if (L) *L = g->L;
    if (D) *D = g->D;
    if (Vt) *Vt = g->tmpl ? g->tmpl->V : 0;
    return MSM_OK;
}

int msm_group_get_pairs(msm_group *g, int32_t *pairs) {
    if (!g || !pairs) return fail(MSM_ERR_INVALID, "msm_group_get_pairs: null argument");
    if (!g->ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup() must be called first");
    if (g->pairs.size() != 2 * (size_t)g->npairs) {  // the list was written on the device: fetched when first asked for
        g->pairs.resize(2 * (size_t)g->npairs);
        if (g->npairs > 0) MSM_TRY(g->d_pairs.download(g->pairs.data(), g->pairs.size(), g->ctx));
        MSM_TRY(ctx_sync(g->ctx));
    }
    std::copy(g->pairs.begin(), g->pairs.end(), pairs);
    return MSM_OK;
}

int msm_group_get_triplets(msm_group *g, int32_t *triplets) {
    if (!g || !triplets) return fail(MSM_ERR_INVALID, "msm_group_get_triplets: null argument");
    std::copy(g->triplets.begin(), g->triplets.end(), triplets);
    return MSM_OK;
}

int msm_group_patch(msm_group *g, int32_t s, int32_t v, int32_t l, int32_t *ids, double *data, int32_t cap, int32_t *n) {
    if (!g || !n) return fail(MSM_ERR_INVALID, "msm_group_patch: null argument");
    if (!g->ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup() must be called first");
    if (s < 0 || s >= g->S || v < 0 || v >= g->N || l < 0 || l >= g->L) return fail(MSM_ERR_INVALID, "msm_group_patch: index out of range");
    {
        int st = fetch_host_pptr(g, s);
        if (st) return st;
    }
    const int beg = g->h_pptr[s][v * g->L + l], cnt = g->h_pptr[s][v * g->L + l + 1] - beg;
    *n = cnt;
    if (g->h_pidx[s].empty() && g->h_pptr[s].back() > 0) {  // a subject imported from device memory: its index list is fetched when first asked for
        g->h_pidx[s].resize((size_t)g->h_pptr[s].back());
        MSM_TRY(g->pidx[s]->download(g->h_pidx[s].data(), g->h_pidx[s].size(), g->ctx));
        MSM_TRY(ctx_sync(g->ctx));
    }
    if (!ids && !data) return MSM_OK;
    const int Vt = g->tmpl->V;
    std::vector<double> F;
    if (data) {
        F.resize((size_t)g->D * Vt);
        MSM_TRY(g->F[(size_t)s * g->L + l]->download(F.data(), F.size(), g->ctx));
        MSM_TRY(ctx_sync(g->ctx));
    }
    for (int i = 0; i < cnt && i < cap; ++i) {
        const int id = g->h_pidx[s][beg + i];
        if (ids) ids[i] = id;
        if (data)
            for (int d = 0; d < g->D; ++d) data[(size_t)i * g->D + d] = F[(size_t)d * Vt + id];
    }
    return MSM_OK;
}

int msm_group_pairwise_batch(msm_group *g, const int32_t *pair, const int32_t *la, const int32_t *lb, int32_t n, double *out) {
    if (!g || !pair || !la || !lb || !out || n < 0) return fail(MSM_ERR_INVALID, "msm_group_pairwise_batch: bad arguments");
    if (n == 0) return MSM_OK;
    GroupArgs a;
    int st = group_args(g, a);
    if (st) return st;
    const int P = (int)g->npairs;
    for (int i = 0; i < n; ++i)
        if (pair[i] < 0 || pair[i] >= P || la[i] < 0 || la[i] >= g->L || lb[i] < 0 || lb[i] >= g->L) return fail(MSM_ERR_INVALID, "group pairwise query %d out of range", i);
    msm_ctx *ctx = g->ctx;
    for (int off = 0; off < n; off += kBatchChunk) {  // bounded pinned staging, whatever the batch
        const int m = std::min(kBatchChunk, n - off);
        const int32_t *cols[3] = {pair + off, la + off, lb + off};
        double *pinned_out = nullptr;
        st = stage_batch(g, cols, 3, m, &pinned_out);
        if (st) return st;
        st = launch_group_pairwise(ctx, a, g->d_query[0].p, g->d_query[1].p, g->d_query[2].p, m, g->d_answer.p);
        if (st) return st;
        MSM_HIP(hipMemcpyAsync(pinned_out, g->d_answer.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        st = check_status(ctx, "DiscreteGroupCostFunction::computePairwiseCost");
        std::memcpy(out + off, pinned_out, sizeof(double) * (size_t)m);
        if (st) return st;
    }
    return MSM_OK;
}

// the slice [pair0, pair1) of the pair list in processing order, on the device.  The order is cut into pieces by output range
// (a quarter of the slice each, for slices worth overlapping): within a piece the pairs run control point by control point, and a
// piece's results are one contiguous range of the output, which can leave for the host while the next piece is computed.
static int slice_pair_order(msm_group *g, int64_t pair0, int64_t pair1, const int **d_order) {
    if (g->order_p0 != pair0 || g->order_p1 != pair1) {
        msm_ctx *ctx = g->ctx;
        MSM_TRY(ctx_sync(ctx));
        const int64_t n = pair1 - pair0;
        static const int pieces_env = [] { const char *e = std::getenv("MSMHIP_GROUP_PIECES"); return e ? std::max(1, std::min(16, std::atoi(e))) : 0; }();
        const int pieces = pieces_env ? pieces_env : (n >= (1 << 18) ? 4 : 1);
        // pieces shrink geometrically (ratio MSMHIP_GROUP_PIECE_RATIO, default 0.5): a piece's results leave for the host while the next piece is
        // computed, so only the LAST piece's copy is not hidden behind kernels -- a fifteenth of the step's results instead of a quarter (S = 64,
        // ico6 / ico4: 9.0 -> 8.4 ms per delivered step; an eighth of a step, as a rank of eight evaluates it: 1.62 -> 1.52 ms)
        static const double ratio = [] { const char *e = std::getenv("MSMHIP_GROUP_PIECE_RATIO"); const double v = e ? std::atof(e) : 0.5; return v > 0.05 && v <= 1.0 ? v : 0.5; }();
        g->order_chunk.assign(pieces + 1, 0);
        {
            double total = 0.0, w = 1.0, acc = 0.0;
            for (int k = 0; k < pieces; ++k, w *= ratio) total += w;
            w = 1.0;
            for (int k = 1; k < pieces; ++k, w *= ratio) {
                acc += w;
                g->order_chunk[k] = std::min<int64_t>(n, std::max<int64_t>(g->order_chunk[k - 1], (int64_t)((double)n * acc / total)));
            }
            g->order_chunk[pieces] = n;
        }
        std::vector<std::vector<int32_t>> part(pieces);
        for (auto &v : part) v.reserve((size_t)(n / pieces + 1));
        for (int32_t p : g->pair_order) {
            if (p < pair0 || p >= pair1) continue;
            int k = (int)((int64_t)(p - pair0) * pieces / std::max<int64_t>(n, 1));
            while (k + 1 < pieces && p - pair0 >= g->order_chunk[k + 1]) ++k;
            while (k > 0 && p - pair0 < g->order_chunk[k]) --k;
            part[k].push_back(p);
        }
        MSM_HIP(g->d_pair_order.ensure(std::max<size_t>((size_t)n, 1)));
        size_t at = 0;
        for (auto &v : part) {
            if (!v.empty()) MSM_TRY(stage_h2d(ctx, g->d_pair_order.p + at, v.data(), sizeof(int32_t) * v.size()));
            at += v.size();
        }
        MSM_TRY(ctx_sync(ctx));
        g->order_p0 = pair0, g->order_p1 = pair1;
        g->order4_gen = ~0ull;
    }
    if (g->order4_gen != g->pairs_gen) {  // the positions with their pairs' nodes, for this slice and this set-up's pair list
        const int64_t n = pair1 - pair0;
        MSM_HIP(g->d_pair_order4.ensure(std::max<size_t>((size_t)n, 1)));
        MSM_TRY(launch_group_expand_order(g->ctx, g->d_pair_order.p, g->d_pairs.p, (int)n, g->d_pair_order4.p));
        g->order4_gen = g->pairs_gen;
    }
    *d_order = g->d_pair_order.p;
    return MSM_OK;
}

// the evaluations of a (slice of a) label step, queued on the stream; results in device memory
// after_piece(k, lo, hi): the kernels of piece k (pairs [pair0 + lo, pair0 + hi) of the output) have been queued; k = -1: the triplets
static int group_move_compute_untimed(msm_group *g, const int32_t *labeling, int32_t label, int64_t pair0, int64_t pair1, int64_t trip0, int64_t trip1,
                              double *quads_dev, double *octets_dev, const char *who, const std::function<int(int, int64_t, int64_t)> *after_piece = nullptr) {
    GroupArgs a;
    int st = group_args(g, a);
    if (st) return st;
    const int nodes = g->S * g->N;
    if (label < 0 || label >= g->L) return fail(MSM_ERR_INVALID, "%s: label %d out of range", who, label);
    for (int i = 0; i < nodes; ++i)
        if (labeling[i] < 0 || labeling[i] >= g->L) return fail(MSM_ERR_INVALID, "%s: label of node %d out of range", who, i);
    msm_ctx *ctx = g->ctx;
    void *pin = nullptr;
    st = ctx_io_pinned(ctx, sizeof(int32_t) * (size_t)nodes, &pin);
    if (st) return st;
    MSM_TRY(ctx_sync(ctx));  // the staging buffer may still be the source of an earlier copy
    std::memcpy(pin, labeling, sizeof(int32_t) * (size_t)nodes);
    MSM_HIP(g->d_query[0].ensure(nodes));
    MSM_HIP(hipMemcpyAsync(g->d_query[0].p, pin, sizeof(int32_t) * (size_t)nodes, hipMemcpyHostToDevice, ctx->stream));
    a.move_labeling = g->d_query[0].p;
    a.move_label = label;
    // Large steps evaluate the triplets (the cheap part: strain only) FIRST, so that their results leave for the host behind the pair kernels
    // instead of after them (S = 64 at ico4: 13.1 -> 12.8 ms per step).  Small steps keep them last: at ico2 (0.3 M pairs, 0.9 ms per step) the
    // early copy command cost 0.85 ms per step, measured both ways in the same run.  MSMHIP_GROUP_TRIPLETS=first|last forces either.
    static const int triplets_env = [] {
        const char *e = std::getenv("MSMHIP_GROUP_TRIPLETS");
        return !e ? 0 : (std::strcmp(e, "last") == 0 ? 1 : (std::strcmp(e, "first") == 0 ? 2 : 0));
    }();
    const bool triplets_last = triplets_env == 1 || (triplets_env == 0 && pair1 - pair0 < (1 << 20));
    auto triplets = [&]() -> int {
        a.move_order = nullptr;
        const int64_t first = 8 * trip0, total = 8 * (trip1 - trip0);
        for (int64_t off = 0; off < total; off += kBatchChunk) {
            const int m = (int)std::min<int64_t>(kBatchChunk, total - off);
            if (first + off > 0x7fffffffll - kBatchChunk) return fail(MSM_ERR_CAPACITY, "%s: evaluation index beyond 2^31", who);
            a.move_offset = (int)(first + off);
            st = launch_group_triplet(ctx, a, nullptr, nullptr, nullptr, nullptr, m, octets_dev + off);
            if (st) return st;
        }
        if (after_piece && total > 0) {
            st = (*after_piece)(-1, 0, 0);
            if (st) return st;
        }
        return MSM_OK;
    };
    if (!triplets_last) {
        st = triplets();
        if (st) return st;
    }
    if (pair1 > pair0) {
        st = slice_pair_order(g, pair0, pair1, &a.move_order);
        if (st) return st;
        a.move_order4 = g->d_pair_order4.p;
        if (g->pval_deferred || g->pval_ready) {  // the value copies of this slice's patches (a rank of a sharded run builds them here, once per set-up and slice)
            st = ensure_patch_values(g, pair0, pair1);
            if (st) return st;
            a.pval = const_cast<const double *const *>(g->d_pvalp.p);
            a.dir = g->d_patch_dir.p;
        }
        a.move_base = (int)pair0;
        // The (current, current) combination of a pair does not depend on the proposed label: it is evaluated in a pass of its own
        // (whole wavefronts of pairs whose two nodes kept their labels since the last step leave at once with the kept cost), the
        // three combinations with the proposed label in a second pass.  MSMHIP_GROUP_E00=off: all four together, nothing kept.
        static const bool keep_e00 = [] { const char *e = std::getenv("MSMHIP_GROUP_E00"); return !(e && std::strcmp(e, "off") == 0); }();
        const bool dice = g->p.simmeasure == 4 || g->p.simmeasure == 5;
        const bool two_pass = keep_e00 && !dice;
        if (two_pass) {
            const int64_t P = g->npairs;
            MSM_HIP(g->d_e00.ensure((size_t)std::max<int64_t>(P, 1)));
            MSM_HIP(g->d_prev_labeling.ensure(nodes));
            a.move_e00 = g->d_e00.p;
            a.move_prev = (g->e00_valid && g->e00_p0 == pair0 && g->e00_p1 == pair1) ? g->d_prev_labeling.p : nullptr;
            g->e00_valid = false;  // until this step has gone through: a failure half way leaves nothing to rely on
        }
        // The (proposed, proposed) combination depends on the label alone: the second sweep of Fusion over the labels (I/Fusion/Fusion.h:136-138)
        // takes it from the first (kept per label for this slice until the next set-up; 8 bytes x pairs x labels: 0.8 GB for 64 subjects at
        // ico4).  MSMHIP_GROUP_E11=off, or a slice whose table would pass MSMHIP_GROUP_E11_MB (default 16384): evaluated every time.
        static const bool keep_e11 = [] { const char *e = std::getenv("MSMHIP_GROUP_E11"); return !(e && std::strcmp(e, "off") == 0); }();
        static const double e11_cap_mb = [] { const char *e = std::getenv("MSMHIP_GROUP_E11_MB"); return e ? std::atof(e) : 16384.0; }();
        const int64_t nslice = pair1 - pair0;
        bool have11 = false;
        if (two_pass && keep_e11 && 8.0 * (double)nslice * g->L <= e11_cap_mb * 1048576.0) {
            if (g->e11_p0 != pair0 || g->e11_p1 != pair1 || (int)g->e11_have.size() != g->L) {
                g->e11_have.assign((size_t)g->L, 0);
                g->e11_p0 = pair0, g->e11_p1 = pair1;
            }
            MSM_HIP(g->d_e11.ensure((size_t)nslice * g->L));
            a.move_e11 = g->d_e11.p + (size_t)label * nslice;
            have11 = g->e11_have[(size_t)label] != 0;
            g->e11_have[(size_t)label] = 0;  // until this step has gone through
        }
        for (size_t k = 0; k + 1 < g->order_chunk.size(); ++k) {
            const int64_t n0 = g->order_chunk[k], n1 = g->order_chunk[k + 1];  // pairs of the piece, in processing order
            if (have11) {
                st = launch_group_kept(ctx, a.move_order + n0, (int)pair0, a.move_e11, (int)(n1 - n0), quads_dev);
                if (st) return st;
            }
            static const bool by_four = [] { const char *e = std::getenv("MSMHIP_GROUP_BY_FOUR"); return !(e && std::strcmp(e, "off") == 0); }();
            a.move_first = (int)n0, a.move_count = by_four ? (int)(n1 - n0) : 0;
            for (int pass = two_pass ? 1 : 0; pass <= (two_pass ? 2 : 0); ++pass) {
                const int64_t mult = pass == 0 ? 4 : (pass == 1 ? 1 : (have11 ? 2 : 3)), q0 = mult * n0, q1 = mult * n1;
                a.move_combos = pass == 2 && have11 ? 3 : pass;
                for (int64_t off = q0; off < q1; off += kBatchChunk) {
                    const int m = (int)std::min<int64_t>(kBatchChunk, q1 - off);
                    if (off > 0x7fffffffll - kBatchChunk) return fail(MSM_ERR_CAPACITY, "%s: evaluation index beyond 2^31", who);
                    a.move_offset = (int)off;
                    st = launch_group_pairwise(ctx, a, nullptr, nullptr, nullptr, m, quads_dev);
                    if (st) return st;
                }
            }
            if (after_piece) {
                st = (*after_piece)((int)k, n0, n1);
                if (st) return st;
            }
        }
        if (two_pass) {  // what the next step compares with
            MSM_HIP(hipMemcpyAsync(g->d_prev_labeling.p, g->d_query[0].p, sizeof(int32_t) * (size_t)nodes, hipMemcpyDeviceToDevice, ctx->stream));
            g->e00_valid = true;
            g->e00_p0 = pair0, g->e00_p1 = pair1;
        }
        if (a.move_e11) g->e11_have[(size_t)label] = 1;
        a.move_combos = 0;
        a.move_first = a.move_count = 0;
        a.move_prev = nullptr;
        a.move_e00 = nullptr;
        a.move_e11 = nullptr;
    }
    return triplets_last ? triplets() : MSM_OK;
}

// the same between two events when the caller asked for the kernels' time (msm_group_time_moves; bench.py's gmsm roofline)
static int group_move_compute(msm_group *g, const int32_t *labeling, int32_t label, int64_t pair0, int64_t pair1, int64_t trip0, int64_t trip1,
                              double *quads_dev, double *octets_dev, const char *who, const std::function<int(int, int64_t, int64_t)> *after_piece = nullptr) {
    if (!g->timing) return group_move_compute_untimed(g, labeling, label, pair0, pair1, trip0, trip1, quads_dev, octets_dev, who, after_piece);
    MSM_HIP(hipEventRecord(g->t_ev0, g->ctx->stream));
    const int st = group_move_compute_untimed(g, labeling, label, pair0, pair1, trip0, trip1, quads_dev, octets_dev, who, after_piece);
    if (st) return st;
    MSM_HIP(hipEventRecord(g->t_ev1, g->ctx->stream));
    g->timed = true;
    return MSM_OK;
}

// the copy stream and its events (one per piece + one for the triplets)
static int ensure_copy_stream(msm_group *g) {
    if (!g->copy_stream) MSM_HIP(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
    while (g->copy_events.size() < 8) {
        hipEvent_t e;
        MSM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g->copy_events.push_back(e);
    }
    return MSM_OK;
}
// finished results leave for pinned host memory on the copy stream while the compute stream goes on
static int copy_behind(msm_group *g, int slot, double *host_dst, const double *dev_src, size_t n) {
    msm_ctx *ctx = g->ctx;
    hipEvent_t e = g->copy_events[(size_t)slot % g->copy_events.size()];
    MSM_HIP(hipEventRecord(e, ctx->stream));
    MSM_HIP(hipStreamWaitEvent(g->copy_stream, e, 0));
    MSM_HIP(hipMemcpyAsync(host_dst, dev_src, sizeof(double) * n, hipMemcpyDeviceToHost, g->copy_stream));
    return MSM_OK;
}

static int group_fusion_move_impl(msm_group *g, const int32_t *labeling, int32_t label, double *pair_quads, double *triplet_octets) {
    if (!g || !labeling) return fail(MSM_ERR_INVALID, "msm_group_fusion_move: null argument");
    if (!g->ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup() must be called first");
    msm_ctx *ctx = g->ctx;
    const int64_t P = g->npairs, T = (int64_t)g->S * g->Tc;
    MSM_HIP(g->d_move_out.ensure((size_t)(4 * P + 8 * T) + 2));
    double *dq = g->d_move_out.p, *dt = dq + ((4 * P + 1) & ~1ll);  // both 16-byte aligned
    int st;
    const bool pinned_q = !pair_quads || P == 0 || ctx_mapped(ctx, pair_quads, sizeof(double) * 4 * (size_t)P);
    const bool pinned_t = !triplet_octets || T == 0 || ctx_mapped(ctx, triplet_octets, sizeof(double) * 8 * (size_t)T);
    if (pinned_q && pinned_t) {
        // Both arrays lie in msm_host_alloc / msm_host_register blocks: the results of each quarter of the pair list (and of the
        // triplets) are copied out by the DMA engine while the kernels of the next quarter run -- 186 MB at S = 64, 3.7 ms that
        // used to follow the kernels.
        st = ensure_copy_stream(g);
        if (st) return st;
        const std::function<int(int, int64_t, int64_t)> deliver = [&](int k, int64_t lo, int64_t hi) -> int {
            if (k < 0) return triplet_octets ? copy_behind(g, 7, triplet_octets, dt, 8 * (size_t)T) : MSM_OK;
            return copy_behind(g, k, pair_quads + 4 * lo, dq + 4 * lo, 4 * (size_t)(hi - lo));
        };
        st = group_move_compute(g, labeling, label, 0, pair_quads ? P : 0, 0, triplet_octets ? T : 0, dq, dt, "msm_group_fusion_move", &deliver);
        if (st) return st;
        st = check_status(ctx, "DiscreteGroupCostFunction (fusion move)");
        MSM_HIP(hipStreamSynchronize(g->copy_stream));
        return st;
    }
    st = group_move_compute(g, labeling, label, 0, pair_quads ? P : 0, 0, triplet_octets ? T : 0, dq, dt, "msm_group_fusion_move");
    if (st) return st;
    // delivery: arrays inside a msm_host_alloc / msm_host_register block are written by a copy kernel (full-width stores over
    // PCIe, no staging); others go through the pinned staging buffer in chunks
    bool flagged = false;
    for (int pass = 0; pass < 2; ++pass) {
        double *out = pass == 0 ? pair_quads : triplet_octets;
        const double *src = pass == 0 ? dq : dt;
        const size_t total = (size_t)(pass == 0 ? 4 * P : 8 * T);
        if (!out || total == 0) continue;
        double *mapped = static_cast<double *>(ctx_mapped(ctx, out, sizeof(double) * total));
        if (mapped && reinterpret_cast<uintptr_t>(mapped) % 16 == 0) {
            st = ctx_flag(ctx);
            if (st) return st;
            st = launch_copy_to_mapped(ctx, src, mapped, total, ctx->d_flag_map);
            if (st) return st;
            flagged = true;
            continue;
        }
        for (size_t off = 0; off < total; off += kBatchChunk) {
            const size_t m = std::min<size_t>(kBatchChunk, total - off);
            void *pin = nullptr;
            st = ctx_io_pinned(ctx, sizeof(double) * m, &pin);
            if (st) return st;
            MSM_HIP(hipMemcpyAsync(pin, src + off, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
            MSM_TRY(ctx_sync(ctx));
            std::memcpy(out + off, pin, sizeof(double) * m);
        }
    }
    st = check_status(ctx, "DiscreteGroupCostFunction (fusion move)");
    if (flagged) ctx->h_flag[0] = 0;
    return st;
}

// A slice of a label step with the results left in DEVICE memory (the caller's buffers, e.g. torch tensors that go into an
// RCCL gather, or the device address of mapped host memory): pairs [pair0, pair1) -> quads_dev[4 * (pair1 - pair0)],
// triplets [trip0, trip1) -> octets_dev[8 * (trip1 - trip0)].
// The kept (current, current) and (label, label) pair costs are marked valid when their kernels are queued; a step that then fails -- a status
// raised by a kernel (e.g. MSM_ERR_CAPACITY), a failed copy -- must not leave them behind for later steps to deliver: an entry point that
// returns an error drops everything kept.
int msm_group_fusion_move(msm_group *g, const int32_t *labeling, int32_t label, double *pair_quads, double *triplet_octets) {
    const int st = group_fusion_move_impl(g, labeling, label, pair_quads, triplet_octets);
    if (st && g) g->drop_kept();
    return st;
}

static int group_fusion_move_dev_impl(msm_group *g, const int32_t *labeling, int32_t label, int64_t pair0, int64_t pair1, int64_t trip0, int64_t trip1,
                                      double *quads_dev, double *octets_dev) {
    if (!g || !labeling) return fail(MSM_ERR_INVALID, "msm_group_fusion_move_dev: null argument");
    if (!g->ready) return fail(MSM_ERR_STATE, "msm_group: msm_group_setup() must be called first");
    const int64_t P = g->npairs, T = (int64_t)g->S * g->Tc;
    if (pair0 < 0 || pair1 < pair0 || pair1 > P || trip0 < 0 || trip1 < trip0 || trip1 > T) return fail(MSM_ERR_INVALID, "msm_group_fusion_move_dev: range out of bounds");
    if ((pair1 > pair0 && !quads_dev) || (trip1 > trip0 && !octets_dev)) return fail(MSM_ERR_INVALID, "msm_group_fusion_move_dev: missing output buffer");
    // an output inside a msm_host_alloc / msm_host_register block (pinned host memory, possibly shared with other processes): the
    // kernels' scattered 8-byte results are gathered in HBM first and cross PCIe as one copy
    msm_ctx *ctx = g->ctx;
    const size_t nq = (size_t)(4 * (pair1 - pair0)), nt = (size_t)(8 * (trip1 - trip0));
    const bool host_q = nq && ctx_mapped(ctx, quads_dev, sizeof(double) * nq), host_t = nt && ctx_mapped(ctx, octets_dev, sizeof(double) * nt);
    double *cq = quads_dev, *ct = octets_dev;
    if (host_q || host_t) {
        MSM_HIP(g->d_move_out.ensure(nq + nt + 2));
        if (host_q) cq = g->d_move_out.p;
        if (host_t) ct = g->d_move_out.p + ((nq + 1) & ~(size_t)1);
    }
    int st;
    if (host_q || host_t) {  // pieces leave for the host behind the kernels (see msm_group_fusion_move)
        st = ensure_copy_stream(g);
        if (st) return st;
        const std::function<int(int, int64_t, int64_t)> deliver = [&](int k, int64_t lo, int64_t hi) -> int {
            if (k < 0) return host_t ? copy_behind(g, 7, octets_dev, ct, nt) : MSM_OK;
            return host_q ? copy_behind(g, k, quads_dev + 4 * lo, cq + 4 * lo, 4 * (size_t)(hi - lo)) : MSM_OK;
        };
        st = group_move_compute(g, labeling, label, pair0, pair1, trip0, trip1, cq, ct, "msm_group_fusion_move_dev", &deliver);
        if (st) return st;
        st = check_status(ctx, "DiscreteGroupCostFunction (fusion move)");
        MSM_HIP(hipStreamSynchronize(g->copy_stream));
        return st;
    }
    st = group_move_compute(g, labeling, label, pair0, pair1, trip0, trip1, cq, ct, "msm_group_fusion_move_dev");
    if (st) return st;
    return check_status(ctx, "DiscreteGroupCostFunction (fusion move)");  // synchronises: the buffers may go into a collective on another stream
}

int msm_group_fusion_move_dev(msm_group *g, const int32_t *labeling, int32_t label, int64_t pair0, int64_t pair1, int64_t trip0, int64_t trip1,
                              double *quads_dev, double *octets_dev) {
    const int st = group_fusion_move_dev_impl(g, labeling, label, pair0, pair1, trip0, trip1, quads_dev, octets_dev);
    if (st && g) g->drop_kept();
    return st;
}

int msm_group_time_moves(msm_group *g, int enable) {
    if (!g) return fail(MSM_ERR_INVALID, "null group");
    (void)hipSetDevice(g->ctx->device);
    if (enable && !g->t_ev0) {
        MSM_HIP(hipEventCreate(&g->t_ev0));
        MSM_HIP(hipEventCreate(&g->t_ev1));
    }
    g->timing = enable != 0;
    g->timed = false;
    return MSM_OK;
}

int msm_group_move_kernels_ms(msm_group *g, double *ms) {
    if (!g || !ms) return fail(MSM_ERR_INVALID, "msm_group_move_kernels_ms: null argument");
    *ms = -1.0;
    if (!g->timed) return MSM_OK;
    MSM_HIP(hipEventSynchronize(g->t_ev1));
    float f = 0.f;
    MSM_HIP(hipEventElapsedTime(&f, g->t_ev0, g->t_ev1));
    *ms = f;
    return MSM_OK;
}

int msm_group_triplet_batch(msm_group *g, const int32_t *t, const int32_t *la, const int32_t *lb, const int32_t *lc, int32_t n, double *out) {
    if (!g || !t || !la || !lb || !lc || !out || n < 0) return fail(MSM_ERR_INVALID, "msm_group_triplet_batch: bad arguments");
    if (n == 0) return MSM_OK;
    GroupArgs a;
    int st = group_args(g, a);
    if (st) return st;
    const int T = g->S * g->Tc;
    for (int i = 0; i < n; ++i)
        if (t[i] < 0 || t[i] >= T || la[i] < 0 || la[i] >= g->L || lb[i] < 0 || lb[i] >= g->L || lc[i] < 0 || lc[i] >= g->L)
            return fail(MSM_ERR_INVALID, "group triplet query %d out of range", i);
    msm_ctx *ctx = g->ctx;
    for (int off = 0; off < n; off += kBatchChunk) {
        const int m = std::min(kBatchChunk, n - off);
        const int32_t *cols[4] = {t + off, la + off, lb + off, lc + off};
        double *pinned_out = nullptr;
        st = stage_batch(g, cols, 4, m, &pinned_out);
        if (st) return st;
        st = launch_group_triplet(ctx, a, g->d_query[0].p, g->d_query[1].p, g->d_query[2].p, g->d_query[3].p, m, g->d_answer.p);
        if (st) return st;
        MSM_HIP(hipMemcpyAsync(pinned_out, g->d_answer.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        st = check_status(ctx, "DiscreteGroupCostFunction::computeTripletCost");
        std::memcpy(out + off, pinned_out, sizeof(double) * (size_t)m);
        if (st) return st;
    }
    return MSM_OK;
}

}  // extern "C"
