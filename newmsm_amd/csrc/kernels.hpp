// kernels.hpp -- launch wrappers implemented in kernels.hip (device pointers only).
#pragma once

#include "internal.hpp"

namespace msm {

// per-leaf sub-cell masks (see FlatOctree::node); writes 64 words per mask block
int launch_build_masks(msm_ctx *ctx, const DevTree &T, const double4 *d_nodebox, unsigned long long *d_mask);
// triangle records and bounding cones (per triangle, then per padded leaf entry) from the mesh in HBM
int launch_build_recs(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, TriRec *d_rec, float4 *d_tcone, const int32_t *d_leaf_tri,
                      int nentries, float4 *d_cone);
// ray-table records (internal.hpp: kRayPieces) from the triangle records, the edge planes and, when given, a single feature row (V doubles)
// records and cones of every tree of a forest (the entry counts are read from the device: entries[b * entries_stride])
int launch_build_recs_forest(msm_ctx *ctx, const double *d_xyz, size_t comp_stride, size_t tree_stride, const int32_t *d_tri, int T, int B, TriRec *d_rec, float4 *d_tcone,
                             size_t s_rec, const int32_t *d_leaf_tri, float4 *d_cone, size_t s_leaf, const int *d_entries, size_t entries_stride);
int launch_build_raytri(msm_ctx *ctx, const TriRec *d_rec, const float4 *d_edge, int T, const double *d_feat1, int D, float4 *d_out);
// ---- adaptive barycentric weights on the device (resample_kernels.hip): Resampler::get_adaptive_barycentric_weights R/resampler.cpp:72-140
struct AdaptiveDevArgs {
    int nOld, nNew;
    const int *fvid, *rvid;        // forward (new -> old) / reverse (old -> new) hit-triangle vertex ids, 3 x N SoA
    const double *fw, *rw;         // their projected barycentric weights
    const double *oldA, *newA;     // vertex areas
    int *roff, *rfill, *rkey;      // transposed reverse lists: offsets (nNew + 1), fill counters (nNew), old vertex ids (3 nOld)
    double *rwt;
    int *coff, *cfill, *ckey;      // columns of the result: offsets (nOld + 1), fill counters (nOld), new vertex ids (3 nNew + 3 nOld)
    double *cval, *correction;     // (nOld)
    int *scan_tmp;                 // scratch of the prefix sums, max(nNew, nOld) / 4096 + 2
    int *long_flag;                // 2 per problem: does any transposed reverse list / any column exceed the short sort's limit?
    int *tkey;                     // scratch of the long-list sort, 3 nNew + 3 nOld
    double *tval;
    int *row_ptr, *col;            // the result as CSR: nNew + 1, 3 nNew + 3 nOld
    double *val;
    // B problems at once (gMSM: a subject's data mesh rotated to every label against the one template): problem b = blockIdx.y of
    // every launch uses the arrays b * stride elements further on.  fstride / rstride: distance between the three components of
    // fvid / fw and rvid / rw (nNew and nOld for one problem).  All strides 0 and B = 1 for one problem.
    int B = 1;
    size_t fstride = 0, rstride = 0;
    size_t s_f = 0, s_r = 0, s_oldA = 0, s_newA = 0, s_roff = 0, s_rfill = 0, s_r3 = 0, s_coff = 0, s_cfill = 0, s_cap = 0, s_corr = 0, s_rowptr = 0, s_scan = 0;
};
// vertex areas of B coordinate sets over one triangle list: component c of vertex i of set b at xyz[c * comp + b * set + i];
// ta: scratch, B x T; area: B x V
int launch_vertex_areas_batch(msm_ctx *ctx, const double *d_xyz, size_t comp, size_t set, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid,
                              int B, double *d_ta, double *d_area);
// out[b][d][k] = problem b's weights applied to data (D x nOld, shared by the problems); out: B blocks of out_stride doubles
int launch_apply_rows_batch(msm_ctx *ctx, const AdaptiveDevArgs &a, int D, const double *d_data, double *d_out, size_t out_stride);
int launch_vertex_areas(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid, double *d_ta,
                        double *d_area);
int launch_adaptive_surgery(msm_ctx *ctx, const AdaptiveDevArgs &a);
int launch_apply_rows(msm_ctx *ctx, int nNew, int nOld, int D, const int *row_ptr, const int *col, const double *val, const double *d_data, double *d_out);
// n doubles from device memory into mapped pinned host memory, and the context's status word into flags_mapped[0] when it is set
int launch_copy_to_mapped(msm_ctx *ctx, const double *d_src, double *mapped_dst, size_t n, int *flags_mapped);
int launch_query(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode);
// the same through the target's direction table where it has one (kernels.hip: k_query_rays + k_query_open); d_open: N + 1 ints of scratch
int launch_query_rays(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode, int *d_open);
int launch_closest_vertex(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_out);
// sphere_project_warp / surface_resample on the device: out[i] = the weights of query i in its triangle applied to d_to (3 x V); d_out may be d_q
int launch_warp(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, const double *d_to, int V, bool to_sphere, double *d_out);
int query_lanes(long long N);  // lanes per query of the search kernels for a launch of N queries (kernels.hip)
// smooth_data: unit vectors of the N vertices (d_unit: 3 x N scratch), then one wavefront per output vertex
// d_unit of launch_smooth holds smooth_scratch_doubles(N) doubles: the unit vectors and, from smooth_bounds_offset(N), a bounding ball per 64 of them
inline size_t smooth_bounds_offset(int N) { return ((size_t)3 * N + 3) / 4 * 4; }
inline size_t smooth_scratch_doubles(int N) { return smooth_bounds_offset(N) + 4 * (((size_t)N + 63) / 64) + 4; }
int launch_smooth(msm_ctx *ctx, const double *d_xyz, int N, double *d_unit, const int *d_cv, const double *d_data, int Vorig, int D, double sigma,
                  double cosang, const double *d_excl, double *d_out, double *d_excl_out);
// check_for_intersections (M/reg_tools.cpp:118-129) of every vertex: fold[0] += folded, fold[1] += vertices without a triangle, fold[2 + v] = folded
int launch_fold_detect(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid, int32_t *d_fold);
// d_sorted = each patch's members (d_pidx, CSR d_pptr) in Morton order of their positions d_xyz (3 x n SoA); d_code: n scratch words
int launch_sort_patches(msm_ctx *ctx, const double *d_xyz, int n, const int32_t *d_pptr, int ngroups, const int32_t *d_pidx, uint32_t *d_code,
                        int32_t *d_sorted);
// A uniform grid over the source vertices for launch_range's clustered form: cell (ix, iy, iz) = G * (G * ix + iy) + iz holds ids[start[cell] .. start[cell + 1]) in no
// particular order; *bad != 0: a vertex is not finite and the grid is not used.  Cell of a coordinate x: floor((x - origin) * inv_h) clamped to [0, G).
struct RangeGrid {
    const int *start = nullptr, *ids = nullptr, *bad = nullptr;
    int G = 0;
    double origin = 0.0, inv_h = 0.0;
};
// start: G^3 + 1 ints, cursor: G^3 ints, ids: Nsrc ints, bad: 1 int, tmp: G^3 / 4096 + 2 ints
int launch_range_grid_build(msm_ctx *ctx, const double *d_src, int Nsrc, int G, double origin, double inv_h, int *d_start, int *d_cursor, int *d_ids, int *d_bad, int *d_tmp);
// d_chunk_bounds: scratch, one double4 per 64 source vertices; *d_nflag (device int) = entries flagged as undecided (bit 31);
// cluster > 1: every `cluster` consecutive centres lie close together (gMSM: a control point's L candidate positions) and share the pruning
int launch_range(msm_ctx *ctx, const double *d_cp, int Ncp, const double *d_src, int Nsrc, const double *d_maxsep, double range, int cap,
                 uint32_t *d_slots, int *d_counts, double4 *d_chunk_bounds, int *d_nflag, int cluster = 1, const RangeGrid *grid = nullptr);
// rows of slots -> the contiguous list d_pidx at the offsets d_pptr (M + 1, already summed); valid when nothing was flagged
int launch_patch_compact(msm_ctx *ctx, const uint32_t *d_slots, int cap, const int32_t *d_pptr, int M, int32_t *d_pidx, size_t pidx_cap = (size_t)-1);
// data[0 .. n) -> its exclusive prefix sums in place, data[n] = the total; tmp: n / 4096 + 2 ints
int launch_scan_exclusive(msm_ctx *ctx, int *d_data, int n, int *d_tmp);

// rnl[(node*L + l)*9..] = estimate_rotation_matrix(cp[node], rot[node]*labels[l]); moved (optional, N x L x 3 AoS) = rot[node]*labels[l]
int launch_label_rotations(msm_ctx *ctx, const double *d_cp, int N, const double *d_rot, const double *d_labels, int L, double *d_rnl,
                           double *d_moved);

struct UnaryLaunch {
    DevTree tree;
    const double *tfeat;   // target features, V x D vertex-major
    int D;
    int N, L;
    const double *cp;      // 3 x N
    const double *rnl;     // N x L x 9 from launch_label_rotations
    const double *labels;  // 3 x L
    const double *src;     // 3 x Nsrc
    int Nsrc;
    const double *sfeat;   // D x Nsrc
    const double *cfw;     // rows x Nsrc or nullptr
    int cfw_rows;
    const double *sfeat_vm = nullptr;  // Nsrc x D and Nsrc x rows vertex-major copies (multivariate reduction), optional
    const double *cfw_vm = nullptr;
    const int *pptr, *pidx;
    const int *order;      // launch order of the control points (a permutation of 0..N-1)
    const double *absw;
    int pmax;
    int simmeasure;
    double percentile;     // DICE measures (sparsesimkernel::percentile)
    double *U;             // L x N
    // scratch owned by the cost object
    int ntri;                      // triangles in the target mesh
    double *tval;                  // one double per point sample (L * total patch points)
    unsigned long long *fix_list;  // one slot per point sample
    double *fix_pt;                // three doubles per slot
    unsigned int *fix_cnt;         // unary_fix_counter_words() words (zeroed by the launch)
    const unsigned int *fix_off;   // unary_fix_segments() + 1 offsets from unary_fix_offsets()
    int *redo_list;                // N ints
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // optional: recorded around the samples kernel
};
int unary_nsplit(int L, int pmax);
int unary_fix_segments();
size_t unary_fix_counter_words();
void unary_fix_offsets(int N, int L, int pmax, const int32_t *pptr, const int32_t *order, std::vector<uint32_t> &off);
int launch_unary_univariate(msm_ctx *ctx, const UnaryLaunch &u);
// multivariate / patchwise: the samples kernel stores (triangle, raw weights) per sample, a second kernel reduces
struct UnaryWeightsScratch {
    int *stri;      // one int per sample
    double *sw3;    // three doubles per sample
};
int launch_unary_multivariate(msm_ctx *ctx, const UnaryLaunch &u, const UnaryWeightsScratch &w, bool patchwise);

// pairwise / triplet clique costs
struct CliqueArgs {
    int kind, simmeasure;
    int N, L, T, P;
    const int *triplets;      // T x 3
    const int *pairs;         // P x 2
    const double *cp;         // 3 x N current control grid
    const double *ocp;        // 3 x N control grid at set_meshes (_oCPgrid)
    const double *orig;       // 3 x Norig source coordinates at set_meshes (_ORIG), indexed by control-point id
    int Norig;
    const double *moved;      // N x L x 3: ROT[node] * label
    const double *rnl;        // N x L x 9
    const int *cp_tri;        // 3 x Tc control grid triangles
    int Tc;
    const int *cp_tid_ptr, *cp_tid;  // triangles adjacent to each control point
    double lambda, mu, kappa, k_exp, rexp, mvdmax, percentile;
    // HO likelihood
    DevTree tree;
    const double *tfeat;      // V x D
    int D;
    const double *src;        // 3 x Nsrc
    int Nsrc;
    const double *sfeat;      // D x Nsrc
    const double *cfw;
    int cfw_rows;
    const double *sfeat_vm, *cfw_vm;  // vertex-major copies (HO multivariate), or nullptr
    const int *bin_ptr, *bin_idx;  // source vertices per control-grid triangle
    int bin_cap;                   // largest bin (LDS slice per evaluation in the HO kernels)
    // scratch of the three-kernel fusion-move path (ray-table targets): one value per (octet evaluation, bin point)
    double *ho_big;                // bins beyond kHoBinMax: a slice of bin_cap doubles per WORKGROUP in HBM instead of LDS (or nullptr)
    double *ho_vals;               // 8 x (all bin points) doubles, or nullptr
    unsigned *ho_pending;          // same count: evaluation << 10 | point, for the complete search
    unsigned *ho_count;            // 1 word, zero between calls
    const double *absw;
    // anatomical strain (rmode 4/5)
    int rmode;
    DevTree atree;            // octree of the anatomical-resolution sphere (_TARGEThi)
    const double *atarget;    // 3 x Va: _aTARGET
    int Va;
    const double *asrc;       // 3 x Vs: _aSOURCE
    int Vs;
    const int *asrc_tri;      // 3 x Ts
    int Ts;
    const int *aw_ptr, *aw_cp;  // _ANATbaryweights CSR
    const double *aw_val;
    const int *af_ptr, *af_idx;  // NEARESTFACES CSR
    int *status;
};
// ---- fused fusion move of the HO (triclique) classes on a direction-table target (move_kernels.hip) ----
// The labeling travels in the kernel arguments, one byte per control point (no copy engine on the call's critical path).
constexpr int kPairSmallPatch = 80;  // entries a 16-lane query of k_group_pairwise keeps in registers (5 per lane)
constexpr int kHoBinMax = 16384;  // source vertices under one control triangle whose sampled values fit one workgroup's 128 KB LDS slice; larger bins keep them in HBM
constexpr int kHoBigBlocks = 512;  // ... a slice per workgroup of at most this many workgroups, which loop over the evaluations
constexpr int kMoveLabelWords = 704;  // N <= 2816 control points (ico4: 2562), L <= 256
struct MoveLabels {
    uint32_t w[kMoveLabelWords];
};
struct MoveArgs {
    // prepared once per get_source_data() by launch_move_prepare: per bin slot (a source vertex in its control triangle's bin)
    const int *slot_tri;     // the control triangle of the slot
    const double *slot_w;    // 3 per slot: barycentric coordinates of the projected source vertex in the CURRENT control triangle
    const double *slot_sf;   // univariate: moving feature (row 1) of the slot's vertex
    const double *slot_cw;   // univariate: its cost-function weight (row 1), or nullptr
    const double *slot_wda;  // univariate: weight x (moving feature - weighted mean of its bin), M/similarities.cpp:146
    const double *tri_frame; // 5 per control triangle: the original triangle's half of the strain (strain_device.hpp: StrainFrame)
    const double *tri_stat;  // 3 per control triangle: sum of weights, weighted mean and variance of the moving patch (:135-150)
    const int4 *blk;         // per workgroup: first control triangle, their number, first bin slot, number of slots
    int nblk;
    int cap;                 // bin slots one workgroup holds at most
    const int *labeling;     // device copy of the labeling, or nullptr when it is packed into MoveLabels
    int label;
    double *vals;            // 8 x (all bin slots): values of the evaluations left to the tail kernel
    unsigned *defer_list;    // 8 x T evaluation ids
    unsigned *defer_cnt;     // two counters; a move uses [parity] and clears [parity ^ 1]
    int parity;
    double *out;             // 8 x T costs, or T with `single` (device memory or mapped pinned host memory)
    int *host_flags;         // mapped pinned host words: [0] a raised status, [1] set when evaluations were left to the tail kernel
    unsigned long long *trace;  // diagnostics (MSM_MOVE_TRACE builds): 8 timestamps per workgroup, or nullptr
    int single;              // 1: combination 000 only (the labeling's own cost of every control triangle); out holds T values
};
// launch_move runs the main kernel; the host launches the tail (launch_move_tail) only when host_flags[1] was set
int launch_move_prepare(msm_ctx *ctx, const CliqueArgs &a, int nslots, int *slot_tri, double *slot_w, double *slot_sf, double *slot_cw, double *slot_wda,
                        double *tri_frame, double *tri_stat);
int launch_move(msm_ctx *ctx, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *labels, hipEvent_t ev_start, hipEvent_t ev_stop);
int launch_move_tail(msm_ctx *ctx, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *labels);

int launch_triplet_batch(msm_ctx *ctx, const CliqueArgs &a, const int *qt, const int *qa, const int *qb, const int *qc, int n, double *out);
int launch_triplet_octets(msm_ctx *ctx, const CliqueArgs &a, const int *labeling, int label, double *out);
struct MoveLabels;
// the strain-only classes: labeling in the kernel arguments, costs to `out` (device or mapped host memory), raised statuses to host_flag
int launch_triplet_octets_packed(msm_ctx *ctx, const CliqueArgs &a, const MoveLabels &lab, int label, double *out, int *host_flag);
int launch_pairwise_batch(msm_ctx *ctx, const CliqueArgs &a, const int *qp, const int *qa, const int *qb, int n, double *out);
int launch_pairwise_table(msm_ctx *ctx, const CliqueArgs &a, double *out);
int launch_triplet_table(msm_ctx *ctx, const CliqueArgs &a, int t0, int t1, double *out);


// groupwise (gMSM)
// one patch (subject, control point, label) as the pair kernel wants it: where its ids and values lie, its position in the subject's lists, its length --
// one 32-byte record instead of four dependent look-ups (pointer tables, row offsets) per side of a query
struct alignas(32) GroupPatchRef {
    const int *ids;
    const double *vals;  // entry-major values (GroupArgs::pval), or nullptr
    int begin, count;
    int pad[2];
};

struct GroupArgs {
    int S, N, L, D, Tc, Vt;
    int simmeasure, fixnan;
    const int *pairs;            // P x 2 global node ids
    const int *triplets;         // T x 3 global node ids
    const int *const *pptr;      // per subject: CSR over (control point * L + label)
    const int *const *pidx;      // per subject: template vertex ids, ascending per patch
    const double *const *F;      // per (subject * L + label): D x Vt resampled features
    // per subject (or nullptr): the D values of every patch entry, entry-major beside pidx -- pval[s][D * e + d] = F[s][label of e's patch][d][pidx[s][e]] --
    // so that a pair cost reads patch A's values as it reads its ids (coalesced) and patch B's at the matched POSITIONS (one 1 KB window) instead of gathering
    // both from the Vt-sized maps by vertex id (msm_group_finalize builds them: launch_group_patch_values)
    const double *const *pval;
    const GroupPatchRef *dir;    // per (global node * L + label), or nullptr (built with pval)
    const double *mask;          // Vt or nullptr
    const double *moved;         // (S * N) x L x 3: ROT * label
    const double *cp;            // S x (3 x N) current control grids
    const double *orig;          // S x (3 x N) _ORIG_MESHES coordinates of the control-point ids
    double lambda, mu, kappa, k_exp, rexp, subcorr;
    double percentile;           // DICE threshold rank
    // fusion move (msm_group_fusion_move): evaluation e = move_offset + query index is pair e / 4 (triplet e / 8) with the
    // proposed label where bit k of e % 4 (e % 8) is set and the current labeling elsewhere; nullptr: explicit index columns
    const int *move_labeling;
    int move_label, move_offset;
    // processing order of a move's pairs (nullptr: list order): query i of a launch evaluates combination (move_offset + i) % 4 of
    // pair move_order[(move_offset + i) / 4] and writes out[4 * (pair - move_base) + combination] -- the results keep the list's
    // order, the work runs control-point tile by tile (group.cpp: pair_order)
    const int *move_order;
    const int4 *move_order4;     // beside move_order (or nullptr): {pair, its two global nodes, 0} per position -- one load instead of order -> pair -> nodes
    int move_base;
    // which combinations of a pair a launch evaluates: 0 all four (query i -> pair i / 4, combination i % 4), 1 only (current,
    // current) (query i -> pair i), 2 the three with the proposed label (query i -> pair i / 3, combination 1 + i % 3).  The
    // (current, current) cost of a pair does not depend on the proposed label: move_e00[pair] keeps it from one label step to the
    // next, and a pair neither of whose nodes changed its label since (move_prev: the previous step's labeling, nullptr: nothing
    // kept yet) takes it from there.
    // The (proposed, proposed) cost of a pair depends on the proposed label only: move_e11[pair - move_base] keeps it per label
    // from the first sweep of Fusion to the second (set with combination 3 evaluated: written; with move_combos 3: the launch evaluates
    // only combinations 1 and 2, query i -> pair i / 2, combination 1 + i % 2, and launch_group_kept copies the kept costs).
    int move_combos;
    // Round 5: with move_order4 and more than one combination per pair the evaluations of a launch's piece (positions [move_first, move_first + move_count) of the
    // order) are enumerated FOUR POSITIONS AT A TIME, combination by combination: evaluation r of the piece = chunk r / (4 m), within it combination (r % (4 m)) / 4 of
    // position 4 chunk + r % 4 (m combinations per pair; the last move_count % 4 positions pair by pair as before).  The four lane groups of a wavefront then hold
    // the same combination of four consecutive pairs -- in the control-point-major order these share their first node, so patch A is the same memory for all
    // four and the wavefront's loads of it touch a quarter of the cache lines.  move_count 0: pair by pair (evaluation i -> pair i / m, combination i % m).
    int move_first, move_count;
    const int *move_prev;
    double *move_e00;
    double *move_e11;
    int patch_cap;               // largest patch of any subject (DICE: LDS staging of the common entries)
    int pair_lanes;              // 32 or 16 lanes per pair cost (group.cpp: msm_group_finalize)
    int *status;
};
// out[i], out[stride + i], out[2 * stride + i] (stride 0: V)
int launch_rotate_to_label(msm_ctx *ctx, const double *d_xyz, int V, const double centre[3], const double label[3], double *d_out, size_t stride = 0);
// all L labels in one launch (out 3 x (L * V), stride = L * V); d_rot9 (optional, V x 9 on the device): the vertices' rotation matrices from the host
int launch_rotate_to_labels(msm_ctx *ctx, const double *d_xyz, int V, const double centre[3], const double *d_labels3, int L, const double *d_rot9, double *d_out,
                            size_t stride);
int launch_group_pairwise(msm_ctx *ctx, const GroupArgs &a, const int *qp, const int *qa, const int *qb, int n, double *out);
// pval[s][D * e + d] for every entry e of every patch of the S subjects (one launch; pval: device table of S device pointers, written through)
// node_flags (optional, S * N ints): only the patches of flagged nodes are written (a rank's slice of the pair list touches an eighth of them)
int launch_group_patch_values(msm_ctx *ctx, const GroupArgs &a, double *const *pval, const int *node_flags = nullptr);
// node_flags[node] = 1 for both nodes of the pairs [pair0, pair1)
int launch_group_mark_nodes(msm_ctx *ctx, const int *pairs, long long pair0, long long pair1, int *node_flags);
// out[i] = {order[i], pairs[2 * order[i]], pairs[2 * order[i] + 1], 0}
int launch_group_expand_order(msm_ctx *ctx, const int *order, const int *pairs, int n, int4 *out);
int launch_group_patch_dir(msm_ctx *ctx, const GroupArgs &a, double *const *pval, GroupPatchRef *dir);
// out[4 * (pair - base) + 3] = kept[pair - base] for the n pairs order[0 .. n)
int launch_group_kept(msm_ctx *ctx, const int *order, int base, const double *kept, int n, double *out);
int launch_group_triplet(msm_ctx *ctx, const GroupArgs &a, const int *qt, const int *qa, const int *qb, const int *qc, int n, double *out);
// the search trees of the S control grids as one forest (arrays of tree b start b * s_* elements in), as the kernels below see it
struct ForestDev {
    const int4 *node;
    const int32_t *parent, *leaf_tri, *grid;
    const float4 *cone;
    const TriRec *rec;
    size_t s_node, s_leaf, s_rec, s_grid;
    const int2 *info;  // per tree: number of nodes, grid depth
};
// N query points (3 x N SoA) in every tree of a forest: hit-triangle vertex ids and projected barycentric weights of point i in tree
// b at vid / w [c * comp + b * N + i] (Resampler::get_barycentric_weights, R/resampler.cpp:142-167)
int launch_query_forest(msm_ctx *ctx, const ForestDev &f, int B, const double *d_q, int N, int *d_vid, double *d_w, size_t comp);
// estimate_pairs (M/DiscreteGroupModel.cpp:37-55) for the whole group: pair (a, v, b > a) = (a * N + v, b * N + the control point of
// subject b closest to control point v of subject a, Octree::get_closest_vertex_ID), in the reference's list order.  cp: component
// c of control point v of subject s at cp[c * S * N + s * N + v].
int launch_group_pairs(msm_ctx *ctx, const ForestDev &f, const double *d_cp, int S, int N, int *d_pairs);
int launch_group_permute_pairs(msm_ctx *ctx, const int *d_in, const int *d_order, int n, int *d_out);
// moved[((s * N + v) * L + l) * 3 ..] = rot[s * N + v] (row-major 3 x 3) * label l: the control points' candidate positions
int launch_group_moved(msm_ctx *ctx, const double *d_rot, int nodes, const double *d_labels /* 3 x L SoA */, int L, double *d_moved);
// the patch centres of one subject as k_range takes them: centres 3 x (N * L) SoA and the per-centre spacing
int launch_group_centres(msm_ctx *ctx, const double *d_moved_subject, const double *d_spacing_subject, int N, int L, double *d_centres, double *d_sep);

}  // namespace msm
