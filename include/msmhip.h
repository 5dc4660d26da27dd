/*
 * msmhip.h -- C ABI of libmsmhip: the MI355X (gfx950) implementation of newMSM's data-parallel hot
 * path (octree nearest-triangle search + barycentric / adaptive-barycentric resampling, and the
 * discrete unary / pairwise / triplet label-cost evaluation).
 *
 * Every entry point names the reference interface it replaces (paths under
 * /root/reference/libraries/: R/ = msm-newresampler/src, M/ = msm-newmeshreg/src,
 * I/ = msm-newmeshreg/include).  INTEGRATION.md shows the binding a newMSM maintainer would add.
 *
 * Conventions
 *  - plain C types only; all arrays are caller-owned HOST memory unless a name ends in _dev;
 *  - point sets are SoA: xyz = x[0..N) y[0..N) z[0..N) (3 x N doubles); triangle lists are
 *    3 x T int32 (first, second, third vertex rows); feature matrices are D x V row-major doubles
 *    (the reference's pvalues[dim][vertex], R/mesh.h:45);
 *  - every function returning int returns MSM_OK (0) or a negative MSM_ERR_* code; the message is
 *    available from msm_last_error() (thread-local).  No exception crosses this boundary;
 *  - handles are opaque; a handle belongs to one context (one GPU, one HIP stream); calls on one
 *    context must be serialised by the caller (the reference's OpenMP loops become one launch);
 *  - lifetimes: a cost function / group keeps plain pointers to the meshes handed to it (target, source,
 *    control grid, anatomical sphere, template, subjects' data meshes): they must outlive it, and all of
 *    them must belong to its context (checked: MSM_ERR_INVALID).  Destroy cost functions and groups
 *    first, then meshes, then the context;
 *  - functions marked [host] need no GPU and may be called on a machine without one.  Everything
 *    else fails with MSM_ERR_NOGPU when no device is present: there is no CPU fallback.
 */
#ifndef MSMHIP_H
#define MSMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSM_ABI_VERSION 11  /* 11 (round 5): msm_ctx_wait_stream, msm_ctx_staging_stats, msm_group_context; msm_host_register takes whole pages only.  10 (round 4): msm_group_set_rotation_mode.  9 (round 4): msm_group_set_pair_layout.  8 (round 4): msm_query_lanes, msm_resample_anatomy_grid, msm_cost_triplet_octets_prefetch / msm_cost_prefetch_stats, msm_group_export_subjects_dev / msm_group_import_subjects_dev / msm_group_setup_more_subjects added.  6 (round 3): msm_pairwise_icm, msm_ctx_time_queries / msm_ctx_query_kernel_ms, msm_group_time_moves / msm_group_move_kernels_ms,
                             * msm_store_release_i64 / msm_load_acquire_i64 / msm_min_acquire_i64, msm_mesh_sphere_project_warp added; nothing removed or changed */

#define MSM_OK 0
#define MSM_ERR_INVALID (-1)  /* bad argument / inconsistent sizes */
#define MSM_ERR_HIP (-2)      /* HIP runtime failure */
#define MSM_ERR_OUTSIDE (-3)  /* "Point is not in the bounding box of the mesh", R/octree.cpp:158 */
#define MSM_ERR_NOTFOUND (-4) /* "Error in octree. ... too distorted mesh face", R/octree.cpp:210 */
#define MSM_ERR_NOGPU (-5)    /* no gfx950 device / extension unusable */
#define MSM_ERR_STATE (-6)    /* call order violated (e.g. cost evaluated before get_source_data) */
#define MSM_ERR_CAPACITY (-7) /* caller buffer too small */
#define MSM_ERR_ROTATION (-8) /* "rotation angle is greater than 90 degrees", R/point.cpp:112 */

#define MSM_RAD 100.0    /* R/point.h:32 */
#define MSM_FOLDING 1e7  /* M/reg_tools.h:30 */

typedef struct msm_ctx msm_ctx;
typedef struct msm_mesh msm_mesh;
typedef struct msm_cost msm_cost;

/* ------------------------------------------------------------------------------------------------
 * library
 * ---------------------------------------------------------------------------------------------- */
int         msm_abi_version(void);                 /* [host] */
const char *msm_last_error(void);                  /* [host] message of the last failing call on this thread */
int         msm_device_count(void);                /* [host] number of visible HIP devices (0 when none) */

/* ------------------------------------------------------------------------------------------------
 * [host] mesh utilities the reference keeps in R/mesh.cpp and M/DiscreteModel.cpp.  They produce
 * the inputs of the device path (numbering is observable in outputs, so they reproduce the
 * reference's vertex / triangle / neighbour order exactly).
 * ---------------------------------------------------------------------------------------------- */
/* vertex and triangle count of make_mesh_from_icosa(order); Mesh::get_resolution R/mesh.cpp:810-830 */
int msm_icosphere_counts(int order, int32_t *V, int32_t *T);
/* make_mesh_from_icosa(order) R/mesh.cpp:1111-1196 followed by true_rescale(radius) :1210-1219
 * (radius <= 0: keep the unit sphere).  Same vertex and triangle numbering as the reference. */
int msm_icosphere(int order, double radius, double *xyz, int32_t *tri);
/* Mpoint::nID / trID lists in Mesh::push_triangle order, R/mesh.cpp:115-134, as CSR.
 * nbr / tid may be NULL to size: nbr_ptr[V] / tid_ptr[V] give the lengths. */
int msm_mesh_adjacency(const int32_t *tri, int32_t V, int32_t T, int32_t *nbr_ptr, int32_t *nbr, int32_t *tid_ptr, int32_t *tid);
/* per-vertex mean adjacent triangle area: compute_vertex_area R/mesh.cpp:1275-1283 with the
 * triangle areas taken from the given coordinates (Triangle::calc_area R/triangle.cpp:52-55) */
int msm_vertex_areas(const double *xyz, const int32_t *tri, int32_t V, int32_t T, double *area);
/* Mesh_registration::resample_anatomy, M/mesh_registration.cpp:250-332, without its surface_resample call (msm_barycentric_coords_resample does that):
 * what a --regoption=5 (aMSM) level prepares for msm_cost_set_anatomical.  The control grid (3 x N SoA, 3 x Tc SoA) is retessellated `levels` (=
 * --anatgrid minus --CPgrid, >= 0) times as retessellate(mesh, old_tr_nbours) does it (R/mesh.cpp:1007-1109: children of triangle t numbered 4t .. 4t+3,
 * every vertex re-normalised) and rescaled to `rad` (true_rescale) -> ANAT_ico (axyz 3 x Va SoA, atri 3 x Ta SoA); face_ptr (Tc + 1) / face_idx (Ta):
 * NEARESTFACES, the faces of ANAT_ico under every control triangle IN THE REFERENCE'S ORDER (each pass after the first puts an entry's children in
 * front of the list, :268-283; the order is the summation order of the mean anatomical strain); w_ptr (Va + 1) / w_cp / w_val (3 Va each):
 * _ANATbaryweights, calc_barycentric_weights (R/triangle.cpp:124-143) of every ANAT_ico vertex in the control triangle that names it LAST in the loop
 * :303-321, control point ids ascending within a row (std::map order).  With every output pointer NULL the call only returns the sizes *Va, *Ta. */
int msm_resample_anatomy_grid(const double *cp_xyz, int32_t N, const int32_t *cp_tri, int32_t Tc, int32_t levels, double rad, int32_t *Va, int32_t *Ta,
                              double *axyz, int32_t *atri, int32_t *w_ptr, int32_t *w_cp, double *w_val, int32_t *face_ptr, int32_t *face_idx);
/* NonLinearSRegDiscreteModel::Initialize M/DiscreteModel.cpp:72-89: per control point the largest
 * geodesic distance to a neighbour, and Mesh::calculate_MaxVD R/mesh.cpp:263-277 */
int msm_cp_spacings(const double *xyz, const int32_t *tri, int32_t V, int32_t T, double *maxsep, double *mvdmax);
/* Initialize_sampling_grid + label_sampling_grid M/DiscreteModel.cpp:110-190 on the icosphere of
 * order sg_order (radius 100).  samples / barycentres: 3 x cap SoA, entry 0 is the centre.
 * abs_is_int != 0 selects the int abs() overload the reference may bind at :170 (SURVEY hard parts). */
int msm_label_sampling_grid(int sg_order, double max_dist, int abs_is_int, int32_t cap,
                            double *samples, int32_t *nsamples, double *barycentres, int32_t *nbarycentres);
/* rescale_sampling_grid M/DiscreteModel.cpp:192-214; *scale is read and updated (x0.8) */
int msm_rescale_sampling_grid(const double *samples, int32_t n, double *scale, double *labels);
/* estimate_rotation_matrix R/point.cpp:97-152, row-major 3x3 */
int msm_rotation_matrix(const double ci[3], const double index[3], double R[9]);
/* get_rotations M/DiscreteModel.cpp:310-319: rot[9*k..] = R(centre -> cp[k]) row-major */
int msm_cp_rotations(const double centre[3], const double *cp_xyz, int32_t N, double *rot);
/* estimate_triplets M/DiscreteModel.cpp:291-308 (triplets: 3 per triangle, ascending, AoS T x 3 as
 * the optimisers expect) and estimate_pairs :271-289 (pairs may be NULL to count; returns count) */
int msm_estimate_triplets(const int32_t *tri, int32_t T, int32_t *triplets);
int msm_estimate_pairs(const int32_t *tri, int32_t V, int32_t T, int32_t *pairs);
/* testing hook: the search tree (newresampler::Octree, R/octree.cpp:31-141) of a mesh, built on the host without a GPU:
 * stats as msm_mesh_octree_stats, and a signature of the leaves (box + triangle list in stored order of every leaf) */
int msm_octree_signature(const double *xyz, const int32_t *tri, int32_t V, int32_t T, int64_t stats[5], uint64_t *signature);
/* testing hook, host only: the guarantee of the direction table that the cost kernels search simple-surface targets with (csrc/octree.cpp:
 * build_ray_table), checked at nsamples points (random directions; points next to random edges and vertices) against the first pass of
 * Octree::get_closest_triangle (R/octree.cpp:156-178) over the host-built tree: whatever a kernel may accept must be the one listed triangle
 * that passes the inside test.  report: [0] points [1] accepted by the float test [2] only by the FP64 re-test [3] left to the complete search
 * [4] violations (must be 0) [5] triangles the table cannot use [6] triangles with exclusion boxes [7] simple surface [8] exclusion boxes
 * checked one by one (the centre of the leaf a box stands for must be refused) [9] sampled points refused by the boxes alone */
int msm_ray_table_check(const double *xyz, const int32_t *tri, int32_t V, int32_t T, int32_t nsamples, uint64_t seed, int64_t report[10]);

/* ------------------------------------------------------------------------------------------------
 * context: one per GPU
 * ---------------------------------------------------------------------------------------------- */
msm_ctx *msm_ctx_create(int device);                         /* owns a new non-blocking HIP stream */
msm_ctx *msm_ctx_create_on_stream(int device, void *hip_stream); /* launches on the caller's stream */
void     msm_ctx_destroy(msm_ctx *ctx);
int      msm_ctx_synchronize(msm_ctx *ctx);
void    *msm_ctx_stream(msm_ctx *ctx);                        /* hipStream_t, for event timing */
/* Optional HIP-event timing of the search kernel of msm_query_triangles / msm_closest_vertex (bench.py's `resample` object): with
 * enable != 0 every such call records two events around its kernel on the context's stream; msm_ctx_query_kernel_ms returns the
 * duration of the most recent one in milliseconds (-1 when none was timed). */
/* [host] Release / acquire accesses to 64-bit counters in memory shared between the ranks of a node (newmsm_amd/dist.py:
 * SharedStepBuffer: a producer's slice of a label step must be visible before its progress counter, the consumer's reads must not
 * move before its look at the counters).  __atomic_store_n(.., __ATOMIC_RELEASE) / __atomic_load_n(.., __ATOMIC_ACQUIRE). */
void     msm_store_release_i64(int64_t *addr, int64_t value);
int64_t  msm_load_acquire_i64(const int64_t *addr);
int64_t  msm_min_acquire_i64(const int64_t *addr, int32_t n);   /* the smallest of n counters, each read with acquire semantics */
/* Stream contract.  A context from msm_ctx_create owns a NON-BLOCKING stream: nothing it runs orders against the caller's streams (not even the
 * default stream), and every entry point that takes HOST arrays is complete when it returns.  The entry points that take DEVICE pointers of the
 * caller (msm_group_export_subject(s)_dev, msm_group_import_subject(s)_dev, msm_group_fusion_move_dev, ...) read and write them on the context's
 * stream, therefore:
 *   before the call   the caller's pending work on those buffers (a memset, a collective that fills them, a kernel) must be complete -- or ordered
 *                     by msm_ctx_wait_stream(ctx, your_stream): everything the library queues after it waits for what your_stream held when it
 *                     was called (one event, no host wait; your_stream = NULL names the default stream).  Without either, a late fill of yours
 *                     lands on top of what the library wrote (the race DESIGN.md section 6 describes).
 *   after the call    the library's work on the buffers is complete: every such entry point synchronises the context's stream before it returns,
 *                     so any stream may use them.
 * A context from msm_ctx_create_on_stream runs on the caller's stream and orders with the rest of that stream as usual. */
int      msm_ctx_wait_stream(msm_ctx *ctx, void *hip_stream);
/* [diagnostics] the pinned staging blocks of the context's host <-> device copies (csrc/stager.cpp): out[0] blocks, out[1] their bytes, out[2] blocks ever
 * allocated, out[3] times a copy waited for a busy block.  Blocks are never moved or freed before the context goes. */
int      msm_ctx_staging_stats(msm_ctx *ctx, int64_t out[4]);
int      msm_ctx_time_queries(msm_ctx *ctx, int enable);
int      msm_ctx_query_kernel_ms(msm_ctx *ctx, double *ms);
int      msm_query_lanes(int64_t n_queries);                 /* [host] lanes per query (4 or 8) the search kernels use for a launch of n_queries: names the instantiation a profile shows */
/* Pinned host memory mapped into the GPU's address space.  An output array that lies inside such a block is written by the
 * kernels directly (no staging copy, no copy-engine command): use it for the arrays of the optimisers' inner loop --
 * msm_cost_triplet_octets' E, msm_group_fusion_move's pair_quads / triplet_octets -- the counterpart of the buffers
 * Fusion::optimize keeps per label step (I/Fusion/Fusion.h:142-146).  Plain malloc'ed arrays keep working everywhere.
 * The block belongs to the context and is released with it at the latest. */
void    *msm_host_alloc(msm_ctx *ctx, size_t bytes);
void     msm_host_free(msm_ctx *ctx, void *p);
/* The same for memory the caller owns -- e.g. a POSIX shared-memory segment that the processes of one node (one per GPU) map, so
 * that every rank's kernels deliver their slice of a label step into the optimiser rank's address space without a collective.
 * msm_host_free(ctx, p) undoes the registration (the memory stays the caller's).  p must be page aligned and bytes a multiple of the page
 * size (4096; MSM_ERR_INVALID otherwise): page-locking works on whole pages and the device address of the block is its host address, so a
 * range that shares a page with other data would share that page's GPU mapping with whatever else gets locked there -- the HIP runtime
 * page-locks pageable buffers of asynchronous copies on its own -- and the first of the two to be released unmaps it under the other. */
int      msm_host_register(msm_ctx *ctx, void *p, size_t bytes);

/* ------------------------------------------------------------------------------------------------
 * mesh + search structure.  Replaces newresampler::Mesh (coords/triangles/pvalues) as seen by the
 * hot path and newresampler::Octree(const Mesh&) R/octree.h:48-52, R/octree.cpp:31-141.  The
 * octree is built on first use and rebuilt after msm_mesh_update_coords (the reference constructs a
 * new Octree per call site).
 * ---------------------------------------------------------------------------------------------- */
msm_mesh *msm_mesh_create(msm_ctx *ctx, const double *xyz, int32_t V, const int32_t *tri, int32_t T);
void      msm_mesh_destroy(msm_mesh *m);
int       msm_mesh_update_coords(msm_mesh *m, const double *xyz);   /* Mesh::set_coord for all vertices */
int       msm_mesh_get_coords(msm_mesh *m, double *xyz);
int       msm_mesh_set_features(msm_mesh *m, const double *feat, int32_t D); /* Mesh::set_pvalues */
int       msm_mesh_sizes(const msm_mesh *m, int32_t *V, int32_t *T, int32_t *D);
/* [host part] stats[0]=nodes [1]=leaves [2]=max depth (root 0) [3]=triangle references [4]=largest leaf */
int       msm_mesh_octree_stats(msm_mesh *m, int64_t stats[5]);
/* testing hook: stats + the leaf signature of msm_octree_signature, computed from the tree as it sits in HBM.  Trees of
 * meshes with >= 8192 triangles are built on the GPU (level by level, the same leaves in the same order); MSMHIP_OCTREE=host|gpu
 * forces one of the two builds. */
int       msm_mesh_octree_signature(msm_mesh *m, int64_t stats[5], uint64_t *signature);
/* The same signature for B coordinate sets (xyz: B consecutive 3 x V SoA blocks) over one triangle list, the B trees built together
 * as one forest -- how the gMSM set-up builds the trees of a subject's data mesh rotated to every label (testing aid). */
int       msm_octree_forest_signatures(msm_ctx *ctx, const double *xyz, int32_t V, const int32_t *tri, int32_t T, int32_t B, uint64_t *signatures);
/* Builds the search structures a cost function uses on this mesh as its target (octree, and for a closed star-shaped
 * surface the direction table that settles most searches with one lookup).  The direction table takes tens of ms of
 * host time, so by default it is built on a background thread when a cost function first evaluates against the mesh,
 * and the complete octree search serves until it is ready -- with bit-identical results.  wait != 0 blocks until
 * everything is in place (benchmarks, many evaluations against one target); wait == 0 only starts the work.
 * *ready (optional) = 1 when nothing is pending.  MSMHIP_RAYTABLE=sync|off changes the default for all meshes. */
int       msm_mesh_prepare_search(msm_mesh *m, int wait, int32_t *ready);

/* ------------------------------------------------------------------------------------------------
 * resampler (R/resampler.h:38-53)
 * ---------------------------------------------------------------------------------------------- */
#define MSM_WEIGHTS_PROJECTED 0 /* calc_barycentric_weights R/triangle.cpp:124-143 (query ray-projected) */
#define MSM_WEIGHTS_RAW 1       /* the area ratios inside barycentric_interpolation :145-157 (query as given) */
/* Octree::get_closest_triangle R/octree.cpp:156-214 for N points + Resampler::get_barycentric_weights
 * R/resampler.cpp:142-167.  tri_id[N]; v_id 3 x N (triangle vertex order); w 3 x N.  Any output may be NULL.
 * A failing query yields tri_id < 0 (MSM_ERR_OUTSIDE / MSM_ERR_NOTFOUND) and the call returns the first such code. */
int msm_query_triangles(msm_mesh *target, const double *q_xyz, int32_t N, int32_t *tri_id, int32_t *v_id, double *w, int weight_mode);
/* Octree::get_closest_vertex_ID R/octree.cpp:216-233 */
int msm_closest_vertex(msm_mesh *target, const double *q_xyz, int32_t N, int32_t *v_id);
/* Resampler::get_adaptive_barycentric_weights R/resampler.cpp:72-140 as CSR (rows = vertices of new_mesh,
 * columns ascending = std::map order).  excl (length V of in_mesh) may be NULL.  Vertex areas are taken
 * from the meshes' current coordinates.  Call with col == NULL to obtain *nnz only. */
int msm_adaptive_barycentric_weights(msm_mesh *in_mesh, msm_mesh *new_mesh, const double *excl,
                                     int32_t *row_ptr, int32_t *col, double *val, int64_t cap, int64_t *nnz);
/* metric_resample R/resampler.cpp:304-309 (= barycentric_data_interpolation :30-70): data D x V(in_mesh) -> out D x V(new_mesh).
 * excl (optional, V(in_mesh) values: the EXCL mesh's data, 0 = excluded) masks the weights and the sums as in :38-52;
 * excl_out (optional, V(new_mesh)) is the resampled mask the reference writes back into EXCL (:54-67). */
int msm_metric_resample(msm_mesh *in_mesh, const double *data, int32_t D, msm_mesh *new_mesh, const double *excl, double *out, double *excl_out);
/* sphere_project_warp R/resampler.cpp:311-328: sphere (3 x N, in/out) is carried through from -> to_xyz (3 x V(from)) */
int msm_sphere_project_warp(msm_mesh *from, const double *to_xyz, double *sphere_xyz, int32_t N);
/* The same for the coordinates a mesh handle already holds: `sphere`'s vertices are located on `from` and moved through (from -> to_xyz) in place on
 * the device, its host copy follows -- SPH_reg of Mesh_registration::run_discrete_opt (M/mesh_registration.cpp:224) without three host round trips.
 * to_xyz: 3 x V(from) SoA.  Same context for both meshes.  Bit-identical to msm_sphere_project_warp on the same inputs. */
int msm_mesh_sphere_project_warp(msm_mesh *sphere, msm_mesh *from, const double *to_xyz);
/* surface_resample :284-302 / project_anatomical_mesh :260-282 core: out = sum_j w_j * coords[v_j] for the
 * barycentric weights of q against `from` (no renormalisation); coords 3 x V(from), out 3 x N */
int msm_barycentric_coords_resample(msm_mesh *from, const double *coords_xyz, const double *q_xyz, int32_t N, double *out_xyz);
/* smooth_data R/resampler.cpp:168-230: Gaussian smoothing (std sigma, geodesic) of D x V(orig) data rows onto the vertices
 * of sphlow, with the reference's indexing (orig's octree finds the centre, whose id then indexes sphlow; neighbours
 * run over sphlow's vertices and read orig's data by the same id -- i.e. meant for orig and sphlow being the same
 * sphere).  excl (optional, V(orig) values) is the EXCL mesh's data and excl_out (optional, V(sphlow)) the smoothed
 * mask the reference writes back.  check_scale (R/mesh.cpp:1198-1208) is left to the caller.  out: D x V(sphlow). */
int msm_smooth_data(msm_mesh *orig, const double *data, int32_t D, msm_mesh *sphlow, double sigma, const double *excl, double *out, double *excl_out);
/* nearest_neighbour_interpolation R/resampler.cpp:232-258: data D x V(orig) -> out D x N.  With excl (optional, V(orig)) a
 * query whose closest vertex is excluded (0) gets zeros, and excl_out (optional, N) receives the mask at the queries. */
int msm_nearest_neighbour(msm_mesh *orig, const double *data, int32_t D, const double *q_xyz, int32_t N, const double *excl, double *out, double *excl_out);
/* [host] create_exclusion R/mesh.cpp:1257-1273: excl[i] = 1 when some feature of vertex i lies outside [thrl - EPSILON,
 * thru + EPSILON], else 0 (as written in the reference: the mask marks the vertices to cut with 1).  data D x V. */
int msm_create_exclusion(const double *data, int32_t D, int32_t V, double thrl, double thru, double *excl);

/* ------------------------------------------------------------------------------------------------
 * the callers' side of one iteration (run_discrete_opt, M/mesh_registration.cpp:164-232): what sits between two
 * evaluations of the cost tables besides the optimiser's own data structures
 * ---------------------------------------------------------------------------------------------- */
/* unfold M/reg_tools.cpp:131-178 on the mesh's current coordinates (sphere of the given radius; the reference uses RAD = 100):
 * the fold test check_for_intersections :118-129 runs on the GPU for all vertices; when some are folded (rare) their
 * gradients and the step-halving moves are applied on the host in the reference's serial order, and the test repeats
 * (at most 1000 passes).  *passes = passes that moved vertices (0: nothing folded, coordinates untouched),
 * *first_folded = folded vertices found by the first pass; both optional.  Read the result with msm_mesh_get_coords. */
int msm_mesh_unfold(msm_mesh *m, double radius, int32_t *passes, int32_t *first_folded);
/* [host] variance_normalise M/reg_tools.cpp:804-843: data D x V in place; excl (V values, > 0 keeps the vertex) or NULL.
 * The running mean / variance recurrence of the reference is serial per feature row, so it stays on the host. */
int msm_variance_normalise(double *data, int32_t D, int32_t V, const double *excl);
/* [host] MCMC::optimise M/mcmc_opt.h:31-134 over the tables of msm_cost_unary_table (L x N) and msm_cost_triplet_table
 * (T x L x L x L): `iters` sweeps over the triplets, each proposing one label drawn from std::geometric_distribution(mcparam)
 * (std::mt19937 seeded with `seed`; the reference seeds from std::random_device) and keeping the cheapest of the eight
 * combinations.  labeling (N) is read and updated.  The energy the reference returns is msm_cost_total(labeling). */
int msm_mcmc_optimise(const double *unary, const double *tcosts, const int32_t *triplets, int32_t N, int32_t L, int32_t T, double mcparam,
                      int32_t iters, uint64_t seed, int32_t *labeling);
/* [host] A STAND-IN for the binary solve of one label step of Fusion::optimize (I/Fusion/Fusion.h:198-229 hands the step to ELC's
 * reduction + FastPD, which are licence-restricted and FSL-bound: not reproduced).  Iterated conditional modes over x in {0,1}^N
 * (0: the node keeps its label, 1: it takes the proposed one) for
 *     E(x) = sum_i unary2[2 i + x_i] + sum_p quads[4 p + 2 x_a + x_b] + sum_t octets[8 t + 4 x_a + 2 x_b + x_c]
 * (quads / octets as msm_group_fusion_move and msm_cost_triplet_octets write them; unary2 NULL: no unary costs; P or T may be 0) from
 * x = 0, nodes in ascending order, a node flips only when that lowers E strictly, at most max_passes passes.  Deterministic; it exists
 * so that the fusion-move path can be driven end to end (tools, tests, bench) the way the HCP configurations drive it -- a registration
 * run with it is NOT the reference's optimisation result. */
int msm_fusion_icm_step(const double *unary2 /* N x 2 or NULL */, const double *quads /* P x 4 */, const int32_t *pairs /* P x 2 */, int32_t P,
                        const double *octets /* T x 8 */, const int32_t *triplets /* T x 3 */, int32_t T, int32_t N, int32_t max_passes,
                        int32_t *x /* N, out */);
/* [host] A STAND-IN for FPD::FastPD on the multi-label pairwise MRF of --regoption=1 (M/mesh_registration.cpp:182-188: computeUnaryCosts,
 * computePairwiseCosts, FastPD): iterated conditional modes over unary[label * N + node] and paircosts[(pair * L + labelB) * L + labelA]
 * (the layouts FastPD reads, I/FastPD/FastPD.h:126,213,224), nodes in ascending order, lowest label on ties, at most max_passes passes.
 * labeling: the start (resetLabeling gives zeros) in, the result out.  Not FastPD's optimum: it lets the regoption-1 caller loop run. */
int msm_pairwise_icm(const double *unary, const double *paircosts, const int32_t *pairs, int32_t N, int32_t L, int32_t P, int32_t max_passes, int32_t *labeling);

/* ------------------------------------------------------------------------------------------------
 * discrete cost function.  Replaces NonLinearSRegDiscreteCostFunction and its five subclasses
 * (M/DiscreteCostFunction.h:83-283) behind the DiscreteCostFunction evaluator interface
 * (M/DiscreteCostFunction.h:41-59) the optimisers call.
 * ---------------------------------------------------------------------------------------------- */
#define MSM_COST_UNIVARIATE 0      /* UnivariateNonLinearSRegDiscreteCostFunction */
#define MSM_COST_MULTIVARIATE 1    /* MultivariateNonLinearSRegDiscreteCostFunction */
#define MSM_COST_PATCHWISE 2       /* PatchwiseMultivariateNonLinearSRegDiscreteCostFunction */
#define MSM_COST_HO_UNIVARIATE 3   /* HOUnivariateNonLinearSRegDiscreteCostFunction (--triclique) */
#define MSM_COST_HO_MULTIVARIATE 4 /* HOMultivariateNonLinearSRegDiscreteCostFunction */

typedef struct msm_cost_params { /* set_parameters M/DiscreteCostFunction.cpp:119-133 */
    int32_t kind;        /* MSM_COST_* (chosen in initialize_cost_function M/DiscreteModel.cpp:43-61) */
    int32_t simmeasure;  /* "simmeasure": 1 SSD, 2 correlation, 4 DICE, 5 genDICE (get_sim_for_min M/similarities.h:48-58) */
    int32_t rmode;       /* "regularisermode": 1 pairwise angle, 2/3 triangle strain, 4/5 anatomical strain (msm_cost_set_anatomical) */
    int32_t reserved;
    double  lambda;      /* "lambda" */
    double  mu;          /* "shearmodulus" */
    double  kappa;       /* "bulkmodulus" */
    double  k_exp;       /* "kexponent" */
    double  rexp;        /* "exponent" */
    double  range;       /* "range" (_controlptrange) */
    double  percentile;  /* "percentile" (DICE measures; M/similarities.h:68 default 0.75, must lie in (0,1)) */
} msm_cost_params;

msm_cost *msm_cost_create(msm_ctx *ctx, const msm_cost_params *params);
void      msm_cost_destroy(msm_cost *c);
/* set_anatomical + set_anatomical_neighbourhood (M/DiscreteCostFunction.h:160-170), for regularisermode 4/5 (aMSM:
 * computeTripletCost :169-182 with deform_anatomy :255-301).  sphere = _TARGEThi, the anatomical-resolution sphere
 * (its octree is the reference's `anattree`); atarget_xyz = _aTARGET coordinates, 3 x (vertices of sphere) SoA;
 * asource_* = _aSOURCE (3 x Vs SoA, 3 x Ts SoA); w_* = _ANATbaryweights as CSR over the Vs vertices, control point
 * ids ascending within a row (std::map order); face_* = NEARESTFACES as CSR over the triplets (set_triplets first). */
int msm_cost_set_anatomical(msm_cost *c, msm_mesh *sphere, const double *atarget_xyz, const double *asource_xyz, int32_t Vs,
                            const int32_t *asource_tri, int32_t Ts, const int32_t *w_ptr, const int32_t *w_cp, const double *w_val,
                            const int32_t *face_ptr, const int32_t *face_idx);
/* set_meshes M/DiscreteCostFunction.h:196-198: captures _ORIG (source coords) and _oCPgrid */
int msm_cost_set_meshes(msm_cost *c, msm_mesh *target, msm_mesh *source, msm_mesh *cpgrid);
/* reset_source :208 / reset_CPgrid :209: pick up the meshes' current coordinates */
int msm_cost_reset_source(msm_cost *c, msm_mesh *source);
int msm_cost_reset_cpgrid(msm_cost *c, msm_mesh *cpgrid);
/* set_featurespace :203-205: input (moving) features D x V(source); reference features are the
 * target mesh's features (msm_mesh_set_features) */
int msm_cost_set_source_features(msm_cost *c, const double *feat, int32_t D);
/* set_dataaffintyweighting :165: rows x V(source), rows == 1 or D; NULL = all ones (M/mesh_registration.cpp:234-238) */
int msm_cost_set_cfweight(msm_cost *c, const double *w, int32_t rows);
/* set_spacings :207 */
int msm_cost_set_spacings(msm_cost *c, const double *maxsep, double mvdmax);
/* set_labels :199-202: labels 3 x L SoA, rot 9 per control point (row-major) */
int msm_cost_set_labels(msm_cost *c, const double *labels, int32_t L, const double *rot);
/* setTriplets / setPairs M/DiscreteCostFunction.h:44-45 (AoS, node ids ascending within a clique) */
int msm_cost_set_triplets(msm_cost *c, const int32_t *triplets, int32_t T);
int msm_cost_set_pairs(msm_cost *c, const int32_t *pairs, int32_t P);
/* initialize() + get_source_data(): Univariate M/DiscreteCostFunction.cpp:334-351, Multivariate :393-416,
 * Patchwise :629-650 (range test per control point), HO :468-485 / :541-563 (bin by closest CP triangle),
 * then resample_weights :303-323 */
int msm_cost_get_source_data(msm_cost *c);
/* _sourceinrange as CSR: ptr[groups+1], idx[ptr[groups]]; idx may be NULL to size */
int msm_cost_patches(msm_cost *c, int32_t *ngroups, int32_t *ptr, int32_t *idx, int64_t cap);
int msm_cost_absolute_weights(msm_cost *c, double *absw /* N */);
/* computeUnaryCosts M/DiscreteCostFunction.cpp:236-243: U[label * N + node] (the layout FastPD and MCMC read,
 * I/FastPD/FastPD.h:126, M/mcmc_opt.h:59).  _async only enqueues on the context's stream; _fetch copies back. */
int msm_cost_unary_table(msm_cost *c, double *U);
int msm_cost_unary_table_async(msm_cost *c);
int msm_cost_unary_table_fetch(msm_cost *c, double *U);
/* computeUnaryCost(node,label) :378 for a list of (node,label) queries (Fusion's per-label sweep, I/Fusion/Fusion.h:148-155) */
int msm_cost_unary_batch(msm_cost *c, const int32_t *nodes, const int32_t *labels, int32_t n, double *out);
/* computeTripletCost M/DiscreteCostFunction.cpp:135-188 for n (triplet, labelA, labelB, labelC) queries */
int msm_cost_triplet_batch(msm_cost *c, const int32_t *triplet, const int32_t *la, const int32_t *lb, const int32_t *lc, int32_t n, double *out);
/* the 8 costs per triplet of one fusion move, I/Fusion/Fusion.h:181-196: E[8*t + k], k = 000..111 with bit
 * order (A,B,C) and 0 = current labeling, 1 = `label` */
int msm_cost_triplet_octets(msm_cost *c, const int32_t *labeling, int32_t label, double *E);
/* A hint that costs nothing to ignore: queue the label step (labeling, label) into E and return without waiting.  Between two label steps of
 * Fusion::optimize the host solves a binary problem (ELC + FastPD, I/Fusion/Fusion.h:204-221) while the GPU idles, and most steps of a converging level
 * change no label (three of four in the HCP MSMAll schedule): the next step's evaluations can run meanwhile.  The next msm_cost_triplet_octets on this
 * cost function with the same labeling, label and E only waits for the queued kernel; any other call drops the queued step (its evaluations were never
 * asked for: a status they raised is discarded).  Honoured when E lies in msm_host_alloc memory and the step has its one-kernel form (the fused triclique
 * move, or the strain-only move with the labeling in the kernel arguments); silently ignored otherwise.  Results do not depend on it.
 * msm_cost_prefetch_stats: steps taken from a prefetch / prefetches dropped, since creation. */
int msm_cost_triplet_octets_prefetch(msm_cost *c, const int32_t *labeling, int32_t label, double *E);
int msm_cost_prefetch_stats(msm_cost *c, int64_t *taken, int64_t *dropped);
/* computePairwiseCost :190-226 for n (pair, labelA, labelB) queries, and the full table of
 * computePairwiseCosts :228-234: paircosts[(pair*L + labelB)*L + labelA] */
int msm_cost_pairwise_batch(msm_cost *c, const int32_t *pair, const int32_t *la, const int32_t *lb, int32_t n, double *out);
int msm_cost_pairwise_table(msm_cost *c, double *paircosts);
/* computeTripletCosts M/DiscreteCostFunction.cpp:245-253 (the MCMC optimiser's table, M/mcmc_opt.h:58): tcosts[((t - t0)*L + a)*L*L + b*L + c]
 * for the triplets t0 <= t < t1 (the whole table is T * L^3 doubles, 281 MB at ico4 / 19 labels: fetch it in ranges) */
int msm_cost_triplet_table(msm_cost *c, int32_t t0, int32_t t1, double *tcosts);
/* evaluateTotalCostSum :55-77: parts = {unary, pairwise, triplet} sums in the reference's serial order */
int msm_cost_total(msm_cost *c, const int32_t *labeling, double *total, double parts[3]);
/* Optional HIP-event timing of the sampling kernel (k_unary_rays or k_unary_samples) of each unary-table launch, recorded on the
 * context's stream.  msm_cost_kernel_times returns the durations (ms) of the last launches (most recent last; at most
 * 64 are kept) after synchronising the stream. */
int msm_cost_enable_timing(msm_cost *c, int enable);
int msm_cost_kernel_times(msm_cost *c, double *ms, int32_t cap, int32_t *n);
/* counters since creation: [0] point samples (rotate + nearest triangle + interpolate), [1] unary evals,
 * [2] triplet evals, [3] pairwise evals */
int msm_cost_counters(msm_cost *c, int64_t counters[4]);

/* ------------------------------------------------------------------------------------------------
 * groupwise registration (gMSM).  Replaces DiscreteGroupModel::setupCostFunction and its helpers
 * (M/DiscreteGroupModel.cpp:37-196) and DiscreteGroupCostFunction's evaluators
 * (M/DiscreteGroupCostFunction.cpp:26-98).  Node ids are subject * N + control point, as in the reference.
 * Subjects are independent until the pairwise costs, so a multi-GPU run shards them.
 * ---------------------------------------------------------------------------------------------- */
typedef struct msm_group msm_group;
typedef struct msm_group_params { /* set_parameters M/DiscreteCostFunction.cpp:119-133 */
    int32_t simmeasure;  /* 1 SSD, 2 correlation, 4 DICE, 5 genDICE (get_sim_for_min, M/similarities.h:48-58) */
    int32_t fixnan;      /* "fixnan": NaN cost -> FIX_NAN = 1e7 (M/reg_tools.h:31) */
    double  lambda, mu, kappa, k_exp, rexp, range;
    double  percentile;  /* "percentile" (M/DiscreteCostFunction.cpp:129): DICE threshold rank; 0 means the default 0.75 */
} msm_group_params;
msm_group *msm_group_create(msm_ctx *ctx, const msm_group_params *params, int32_t num_subjects);
msm_ctx   *msm_group_context(msm_group *g);   /* [host] the context the group was created on (its stream: msm_ctx_stream / msm_ctx_wait_stream) */
void       msm_group_destroy(msm_group *g);
/* DiscreteGroupModel::set_meshspace target_space M/DiscreteGroupModel.h:56-61, set_masks :54 (mask: V(template) or NULL) */
int msm_group_set_template(msm_group *g, msm_mesh *template_mesh, const double *mask);
/* Initialize(controlgrid) M/DiscreteGroupModel.cpp:141-161: every subject starts from this grid; builds the triplets */
int msm_group_set_controlgrid(msm_group *g, const double *xyz, const int32_t *tri, int32_t N, int32_t Tc);
/* m_datameshes[s] (reset_meshspace M/DiscreteGroupModel.h:63-65) and FEAT->get_data_matrix(s) (D x V).  The first call
 * for a subject also captures _ORIG_MESHES[s] (set_meshes M/DiscreteGroupCostFunction.h:40-46). */
int msm_group_set_subject(msm_group *g, int32_t subject, msm_mesh *data_mesh, const double *feat, int32_t D);
/* reset_CPgrid M/DiscreteGroupModel.h:67-70 */
int msm_group_reset_cpgrid(msm_group *g, int32_t subject, const double *xyz);
/* m_labels = m_samples (M/DiscreteGroupModel.cpp:177); labels[0] is the sampling-grid centre */
int msm_group_set_labels(msm_group *g, const double *labels, int32_t L);
/* The order of the pair list -- of msm_group_get_pairs and of every pair index and pair range of this interface.  0 (default): estimate_pairs' own
 * (subject A, control point, subject B: M/DiscreteGroupModel.cpp:37-55).  1: control point by control point along a space-filling curve, so that a
 * contiguous range of the list is a region of the sphere: the layout for runs that shard the list over ranks (msm_group_fusion_move_dev), whose
 * slices then read a part of every resampled map instead of whole maps.  Same set of pairs; the optimiser reads pairs[i] beside the i-th costs
 * (I/Fusion/Fusion.h:157-196).  Takes effect at the next set-up. */
int msm_group_set_pair_layout(msm_group *g, int32_t layout);
/* Who computes estimate_rotation_matrix(centre, vertex) (R/point.cpp:97-152) for the data meshes' vertices in get_patch_data (M/DiscreteGroupModel.cpp:97-105):
 * 1 (default) the host's libm as the reference calls it (the matrices are applied on the device in the reference's operation order): the rotated meshes are then
 * the reference's to the bit -- which decides the resampled values where a label carries data vertices exactly onto template vertices (a regular icosphere as the
 * template under regular data grids: the last bits of the rotation pick the triangle that "contains" such a vertex); V x 0.15 us of host time per subject and
 * set-up, beside the GPU's work.  0: the device (its own acos / sincos: the rotated meshes agree with the reference's to 1e-13). */
int msm_group_set_rotation_mode(msm_group *g, int32_t mode);
/* setupCostFunction M/DiscreteGroupModel.cpp:163-196: estimate_pairs :37-55, get_spacings :123-139, get_rotations :77-86,
 * get_patch_data :88-121 */
int msm_group_setup(msm_group *g);
/* Sharded set-up (one process per GPU, subjects split over ranks): every rank runs msm_group_setup_subjects on ITS
 * subjects (the pair list, spacings and rotations are computed for all subjects -- they only need the control grids),
 * exports them, imports the other ranks' subjects (the exchange itself is the caller's all-gather / broadcast over
 * RCCL, see newmsm_amd/dist.py), then msm_group_finalize.  msm_group_setup = all subjects + finalize.
 * F: L x D x V(template) doubles; pptr: N*L + 1; pidx: pptr[N*L] entries (query *npidx with pidx == NULL). */
int msm_group_setup_subjects(msm_group *g, const int32_t *subjects, int32_t n);
int msm_group_export_subject(msm_group *g, int32_t subject, double *F, int32_t *pptr, int32_t *pidx, int64_t cap, int64_t *npidx);
int msm_group_import_subject(msm_group *g, int32_t subject, const double *F, const int32_t *pptr, const int32_t *pidx, int64_t npidx);
/* the same exchange with the caller's buffers in DEVICE memory of this context's GPU (what an RCCL all-gather reads and writes):
 * no host copy in between.  F_dev / pptr_dev / pidx_dev as above; any of them may be NULL on export. */
int msm_group_export_subject_dev(msm_group *g, int32_t subject, double *F_dev, int32_t *pptr_dev, int32_t *pidx_dev, int64_t cap, int64_t *npidx);
int msm_group_import_subject_dev(msm_group *g, int32_t subject, const double *F_dev, const int32_t *pptr_dev, const int32_t *pidx_dev, int64_t npidx);
/* n subjects at once out of / into strided device buffers -- the send and receive buffers of ONE all-gather: subject subjects[k]'s arrays start at
 * F_dev + k * F_stride (doubles), pptr_dev + k * pptr_stride, pidx_dev + k * pidx_stride (int32 entries); npidx[k] = its index count (written on
 * export, read on import).  One range-check launch and one synchronisation per call; the imported row offsets stay on the device (the host copy is
 * fetched if msm_group_patch asks).  msm_group_setup_more_subjects: further subjects of this rank after msm_group_setup_subjects, so that a rank can
 * set up and exchange its shard in chunks (the exchange of one chunk overlapping the set-up of the next, newmsm_amd/dist.py). */
int msm_group_export_subjects_dev(msm_group *g, const int32_t *subjects, int32_t n, double *F_dev, int64_t F_stride, int32_t *pptr_dev, int64_t pptr_stride,
                                  int32_t *pidx_dev, int64_t pidx_stride, int64_t *npidx);
int msm_group_import_subjects_dev(msm_group *g, const int32_t *subjects, int32_t n, const double *F_dev, int64_t F_stride, const int32_t *pptr_dev,
                                  int64_t pptr_stride, const int32_t *pidx_dev, int64_t pidx_stride, const int64_t *npidx);
int msm_group_setup_more_subjects(msm_group *g, const int32_t *subjects, int32_t n);
int msm_group_finalize(msm_group *g);
int msm_group_sizes(msm_group *g, int32_t *nodes, int32_t *pairs, int32_t *triplets);
/* S subjects, N control points each, L labels, D feature rows, V(template): the sizes of the exchange buffers above (any pointer may be NULL) */
int msm_group_dims(msm_group *g, int32_t *S, int32_t *N, int32_t *L, int32_t *D, int32_t *Vt);
int msm_group_get_pairs(msm_group *g, int32_t *pairs /* P x 2 */);
int msm_group_get_triplets(msm_group *g, int32_t *triplets /* T x 3 */);
/* one patch (subject, control point, label): ascending template vertex ids and their D values (cap entries); *n = size */
int msm_group_patch(msm_group *g, int32_t subject, int32_t cp, int32_t label, int32_t *ids, double *data, int32_t cap, int32_t *n);
/* computePairwiseCost M/DiscreteGroupCostFunction.cpp:54-98 / computeTripletCost :26-52 for n queries */
int msm_group_pairwise_batch(msm_group *g, const int32_t *pair, const int32_t *la, const int32_t *lb, int32_t n, double *out);
int msm_group_triplet_batch(msm_group *g, const int32_t *triplet, const int32_t *la, const int32_t *lb, const int32_t *lc, int32_t n, double *out);
/* One label step of Fusion::optimize (I/Fusion/Fusion.h:157-196) in one call: labeling (S * N, by global node id) is the
 * current labeling, label the proposed one.  pair_quads[4 * p + k] = pair_data[p].buffer[k] (k = 2 * [A takes the label] +
 * [B takes it]) and triplet_octets[8 * t + k] = triplet_data[t].buffer[k] (k = 000..111); either output may be NULL.
 * Nothing but the labeling travels to the GPU. */
int msm_group_fusion_move(msm_group *g, const int32_t *labeling, int32_t label, double *pair_quads, double *triplet_octets);

/* A slice of one label step for a multi-GPU run: every rank evaluates ITS slice of the pair list and of the triplet list
 * (the loop of M/DiscreteGroupCostFunction.cpp:54-98 over N_cp * S (S - 1) / 2 pairs is what a 64-subject iteration spends its
 * time in) and leaves the results in DEVICE memory, from where the caller gathers them to the optimiser's rank (RCCL gather).
 * pairs [pair0, pair1) -> quads_dev[4 * (pair1 - pair0)], triplets [trip0, trip1) -> octets_dev[8 * (trip1 - trip0)], both in
 * the buffer order of msm_group_fusion_move. */
int msm_group_fusion_move_dev(msm_group *g, const int32_t *labeling, int32_t label, int64_t pair0, int64_t pair1, int64_t trip0, int64_t trip1,
                              double *quads_dev, double *octets_dev);
/* Optional HIP-event timing of a label step's kernels (everything msm_group_fusion_move / _dev queue on the context's stream for one
 * step: the pair passes, the kept-cost copies, the triplets), for bench.py's gmsm roofline: msm_group_move_kernels_ms returns the most recent
 * step's GPU time in milliseconds (-1 when none was timed). */
int msm_group_time_moves(msm_group *g, int enable);
int msm_group_move_kernels_ms(msm_group *g, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* MSMHIP_H */
