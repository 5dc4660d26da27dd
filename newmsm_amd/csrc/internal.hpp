// internal.hpp -- shared declarations of libmsmhip (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/msmhip.h"
#include "devbuf.hpp"
#include "geom.hpp"

namespace msm {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
// pool.cpp: device memory of the handles (size-classed free lists over hipMalloc / hipFree)
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);
void pool_trim();
size_t pool_idle_bytes();

#define MSM_HIP(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return msm::fail(MSM_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define MSM_TRY(call)               \
    do {                            \
        const int st_ = (call);     \
        if (st_) return st_;        \
    } while (0)

// ---------------------------------------------------------------- host mesh helpers (host_mesh.cpp)
struct Adjacency {
    std::vector<int32_t> nbr_ptr, nbr, tid_ptr, tid;
};
void build_adjacency(const int32_t *tri /*3 x T SoA*/, int V, int T, Adjacency &adj);
void icosphere_unit(int order, std::vector<double> &xyz_aos, std::vector<int32_t> &tri_aos);

// ---------------------------------------------------------------- octree (octree.cpp)
// Per-triangle record used by the exact test: vertices, the triangle-only half of project_point, ids.
struct alignas(128) TriRec {
    double v0[3], v1[3], v2[3];
    double s3[3];
    double d;
    int32_t id[3];
    int32_t tri;
    double reserved;
};
static_assert(sizeof(TriRec) == 128, "TriRec must be 128 bytes");

// Per-triangle record of the ray-table path, nine 16-byte pieces (144 bytes, two cache lines):
//   pieces 0-2  the three edge planes (float4: inward unit normal, acceptance threshold)      -- octree.cpp
//   pieces 3-8  the vertices v0,v1,v2 (9 doubles) and, on single-feature meshes, the feature at them (3 doubles)
// so that a wavefront can fetch the records of its 64 samples with 9 load instructions that touch ~2 lines per
// record (lanes of one instruction read neighbouring pieces) instead of 9 x 64 lines.  Built on the GPU
// (kernels.hip: k_build_raytri).
constexpr int kRayPieces = 9;

struct FlatOctree {
    // node[n].x >= 0: internal node, children are node[n].x .. +7 in (i,j,k) order.
    // node[n].x <  0: leaf with (-x - 1) entries starting at node[n].y (a multiple of 8) in leaf_tri / cone;
    //                 each leaf's entries are padded to a multiple of 8 (leaf_tri -1).
    // node[n].z: leaves with 1..64 entries own 64 sub-cell masks at mask[64 * z ..]; -1 otherwise.
    // node[n].w: depth of the node (root 0); its box has edge 202 / 2^w.
    std::vector<int4> node;
    std::vector<double4> nodebox;  // per node: lower corner (x,y,z) and edge length (w), all exact dyadics
    int nmask_blocks = 0;
    std::vector<int32_t> parent;
    std::vector<int32_t> leaf_tri;
    // Dense top of the tree: the node reached after grid_depth levels of descent (or the leaf met earlier),
    // indexed [ix][iy][iz] with G = 2^grid_depth cells per axis over (-101, 101).  Child boxes are exact
    // halvings, so the cell of a point is found arithmetically and the descent starts there.
    std::vector<int32_t> grid;
    int grid_depth = 0;
    // closed, consistently oriented, star-shaped about the origin, covering the sphere exactly once: every ray from
    // the origin meets exactly one triangle (true for the icosphere targets; false for folded meshes)
    bool simple = false;
    // Ray table of a simple surface (octree.cpp: build_ray_table); ray_G == 0 when absent.
    int ray_G = 0;                   // cube-map cells per face axis
    std::vector<int4> ray_cell;      // 6 x G x G: up to four candidate triangles per direction cell, -1 padded
    std::vector<float4> ray_edge;    // 3 per triangle: inward unit normal of the plane (origin, edge k); .w: see octree.cpp
    std::vector<int4> ray_more;      // candidates 4..7 of the cells that have more than four
    std::vector<int4> ray_excl;      // up to seven leaf boxes a query must not lie in for its triangle to be vouched for: {b0, b1, b2, count} and, for count > 3, {b3 .. b6} in the next record
    double ray_r2lo = 0, ray_r2hi = 0;  // squared radius range of the query points the table is valid for
    int64_t stats[5] = {0, 0, 0, 0, 0};
    // a tree built on the GPU (octree_kernels.hip) has no host arrays: `node` etc. stay empty and these describe it
    int dev_nnodes = 0, dev_entries = 0;
    int nnodes() const { return node.empty() ? dev_nnodes : (int)node.size(); }
    int nentries() const { return node.empty() ? dev_entries : (int)leaf_tri.size(); }
};
// builds the tree exactly as Octree::initialize_tree / add_triangle do (R/octree.cpp:42-141)
void build_octree(const double *xyz /*3 x V SoA*/, const int32_t *tri /*3 x T SoA*/, int V, int T, FlatOctree &out);
// decides FlatOctree::simple and, for a simple surface, fills the ray table of an already built tree
void build_ray_table(const double *xyz, const int32_t *tri, int V, int T, FlatOctree &tree);

// device view of a mesh's search structure
struct DevTree {
    const int4 *node;
    const int32_t *parent;
    const int32_t *leaf_tri;
    const float4 *cone;
    const TriRec *rec;
    const int32_t *grid;
    int grid_depth;  // G = 1 << grid_depth cells per axis
    int simple;  // see FlatOctree::simple
    const unsigned long long *mask;  // 64 per mask block (see FlatOctree::node), or nullptr before the masks are built
    int nnodes;
    // ray table (FlatOctree::ray_*); ray_G == 0: none
    int ray_G;
    const int4 *ray_cell;
    const float4 *ray_tri;   // kRayPieces float4 per triangle
    const int4 *ray_more, *ray_excl;
    double ray_r2lo, ray_r2hi;
};

}  // namespace msm

namespace msm {
struct Stager;
}

// ---------------------------------------------------------------- handles
struct msm_ctx {
    int device = 0;
    int num_cus = 256;  // hipDeviceAttributeMultiprocessorCount (MI355X: 256; a partitioned or smaller device has fewer): sizes the launches whose workgroups wait for one another
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int *d_status = nullptr;  // first error code raised by a kernel (atomicMin), 0 when clean
    int *h_status = nullptr;  // pinned
    // grow-only scratch of the host-array query entry points (hipMalloc / hipFree per call cost more than the queries)
    void *q_buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // queries, triangle ids, vertex ids, weights, the list of queries the direction table left open
    size_t q_cap[5] = {0, 0, 0, 0, 0};
    // pinned staging blocks of every host <-> device copy whose host side is not pinned memory of this library (stager.cpp): grow only, event fenced,
    // never moved; created and destroyed with the context
    msm::Stager *stager = nullptr;
    hipEvent_t wait_ev = nullptr;  // msm_ctx_wait_stream
    // A label step queued ahead (msm_cost_triplet_octets_prefetch) shares the stream, the status word and the mapped flags with every other call on the context:
    // the cost function that holds one (at most one per context), resolved -- taken by its matching call, or waited for and discarded -- before any other
    // entry point synchronises the stream or reads the flags (cost_cliques.cpp: drop_ctx_pending)
    struct msm_cost *pending_cost = nullptr;
    // bumped by everything a queued label step's inputs or destination depend on and that the cost function cannot see: a mesh's coordinates or features
    // changed, a pinned block released (a new block may get the old one's address).  A queued step is only taken when the epoch is the one it was queued under.
    uint64_t epoch = 0;
    // small pinned buffers for the per-label-step calls (labeling in, fusion-move energies out)
    void *io_pin = nullptr;
    void *io_dev = nullptr;  // its device address (the block is mapped), or nullptr
    size_t io_cap = 0;
    // a mapped pinned word kernels of the per-label-step calls store a raised status into (no status copy on their fast path)
    int *h_flag = nullptr;
    int *d_flag_map = nullptr;  // its device address
    std::shared_ptr<void> resample_scratch;  // api.cpp: device buffers of adaptive_weights_dev, kept between calls
    // msm_ctx_time_queries: events around the search kernel of the host-array query entry points
    hipEvent_t q_ev0 = nullptr, q_ev1 = nullptr;
    bool q_timing = false, q_timed = false;
    // scratch of the GPU octree build (octree_kernels.hip), grow only
    double *oct_box = nullptr;
    int *oct_ints = nullptr, *oct_counters = nullptr, *oct_hcounters = nullptr;
    size_t oct_cap_box = 0, oct_cap_ints = 0;
    // msm_host_alloc blocks: pinned host memory the GPU writes results into directly
    struct HostBlock {
        char *host;
        char *dev;
        size_t bytes;
        bool registered = false;  // msm_host_register: the caller's memory, unregistered (not freed) on release
    };
    std::vector<HostBlock> host_blocks;
};

namespace msm {
struct RayJob;
}

struct msm_mesh {
    msm_ctx *ctx = nullptr;
    int V = 0, T = 0, D = 0;
    std::vector<double> xyz;   // 3 x V SoA (host copy)
    std::vector<int32_t> tri;  // 3 x T SoA
    std::vector<double> feat;  // D x V host copy
    bool tree_valid = false;
    bool gpu_tree_always = false;   // build the tree on the GPU whatever the size (the lane meshes of the gMSM set-up: the host never sees their coordinates)
    bool host_xyz_stale = false;    // the coordinates were last written on the device only (group.cpp): fetched before any host-side use
    std::shared_ptr<void> oct_job;  // a GPU build that has been queued but not looked at yet (octree_kernels.hip)
    msm::FlatOctree tree;
    // device
    double *d_xyz = nullptr;   // 3 x V SoA
    int32_t *d_tri = nullptr;  // 3 x T SoA
    float4 *d_tcone = nullptr; // per-triangle bounding cone (scratch of the record kernel)
    double *d_feat = nullptr;  // V x D (vertex-major: one row per vertex for gathers)
    int4 *d_node = nullptr;
    double4 *d_nodebox = nullptr;
    unsigned long long *d_mask = nullptr;
    bool masks_valid = false;
    bool rays_valid = false;
    int32_t *d_parent = nullptr;
    int32_t *d_leaf_tri = nullptr;
    float4 *d_cone = nullptr;
    msm::TriRec *d_rec = nullptr;
    int32_t *d_grid = nullptr;
    int4 *d_ray_cell = nullptr;
    float4 *d_ray_edge = nullptr;
    float4 *d_ray_tri = nullptr;
    int4 *d_ray_more = nullptr, *d_ray_excl = nullptr;
    size_t cap_ray_more = 0, cap_ray_excl = 0;
    bool rayrec_valid = false;  // d_ray_tri matches the current coordinates and features
    size_t cap_ray_cell = 0, cap_ray_edge = 0, cap_ray_rec = 0;
    size_t cap_node = 0, cap_parent = 0, cap_box = 0, cap_leaf = 0, cap_cone = 0, cap_rec = 0, cap_grid = 0, cap_mask = 0;  // one per buffer
    msm::Adjacency adj;
    bool adj_valid = false;
    int32_t *d_tid_ptr = nullptr, *d_tid = nullptr;  // Mpoint::trID lists as CSR (unfold's fold test)
    int32_t *d_fold = nullptr;                       // [0] folded count, [1] vertices without a triangle, then V flags
    // background build of the ray table (api.cpp: ensure_rays): the job of the current tree, and finished-with jobs of
    // earlier trees that are joined when the mesh goes
    uint64_t tree_gen = 0;
    std::shared_ptr<msm::RayJob> ray_job;
    std::vector<std::shared_ptr<msm::RayJob>> stale_jobs;
};

namespace msm {
// Resampler::get_adaptive_barycentric_weights in two halves (api.cpp)
struct AdaptiveQueries {
    std::vector<int> fvid, rvid, closest;  // forward (new -> old) and reverse (old -> new) hit-triangle vertex ids, 3 x N SoA
    std::vector<double> fw, rw;            // and their projected barycentric weights
};
// directions: 1 forward (new -> old), 2 reverse (old -> new), 3 both
int adaptive_queries(msm_mesh *in_mesh, msm_mesh *new_mesh, bool with_closest, AdaptiveQueries &q, int directions = 3);
// The same weights with queries AND list surgery on the device (resample_kernels.hip; no exclusion mask): the CSR stays in HBM.
struct AdaptiveDev {
    int nOld = 0, nNew = 0;
    const int *row_ptr = nullptr, *col = nullptr;  // device; valid until the next adaptive_weights_dev on this context
    const double *val = nullptr;
};
int adaptive_weights_dev(msm_mesh *in_mesh, msm_mesh *new_mesh, AdaptiveDev &out, bool check = true, const DevTree *in_tree = nullptr);  // check = false: the caller checks the status word
// out (device, D x V(new)) = the weights applied to d_data (device, D x V(in)): barycentric_data_interpolation R/resampler.cpp:40-52
int apply_weights_dev(msm_ctx *ctx, const AdaptiveDev &w, const double *d_data, int D, double *d_out);
int ensure_adjacency_dev(msm_mesh *m);
int ensure_tree_pair(msm_mesh *a, msm_mesh *b);  // both trees; a host build of one runs while the GPU builds the other  // Mpoint::trID lists as CSR in HBM (d_tid_ptr / d_tid)
void adaptive_surgery(const AdaptiveQueries &q, int nOld, int nNew, const std::vector<double> &oldA, const std::vector<double> &newA,
                      const double *excl, std::vector<int32_t> &row_ptr, std::vector<int32_t> &col, std::vector<double> &val);
void vertex_areas_of(const double *xyz, const int32_t *tri, int V, int T, const Adjacency &a, std::vector<double> &area);
int install_coords_and_tree(msm_mesh *m, const double *xyz, FlatOctree &&tree);
int ensure_tree(msm_mesh *m);  // build + upload the search structure if stale
int gpu_build_octree(msm_mesh *m, const std::function<void()> *overlap = nullptr);
// B trees over one triangle list and B coordinate sets, built together (octree_kernels.hip: gpu_build_forest); the arrays of tree b
// start b * s_* elements into the shared buffers
struct Forest {
    int B = 0, T = 0, V = 0;
    size_t s_node = 0, s_leaf = 0, s_rec = 0, s_grid = 0;
    DevBuf<int4> node;
    DevBuf<int32_t> parent, leaf_tri, grid;
    DevBuf<double4> nodebox;
    DevBuf<float4> cone, tcone;
    DevBuf<TriRec> rec;
    DevBuf<double> box;
    DevBuf<int> ints, counters;
    int *h_counters = nullptr;  // pinned
    size_t h_counters_cap = 0;
    struct Info {
        int nnodes = 0, entries = 0, grid_depth = 0;
    };
    std::vector<Info> info;
    int last_levels = 0;  // the levels the deepest tree of the previous build needed: queued up front by the next one (no look in between)
    ~Forest() {
        if (h_counters) (void)hipHostFree(h_counters);
    }
};
// component a of vertex i of tree b at d_xyz[a * comp_stride + b * tree_stride + i]; MSM_ERR_CAPACITY: a tree outgrew its arrays
int gpu_build_forest(msm_ctx *ctx, Forest &f, const double *d_xyz, size_t comp_stride, size_t tree_stride, int V, const int32_t *d_tri, int T, int B);
DevTree forest_tree(const Forest &f, int b);
int gpu_build_octree_begin(msm_mesh *m);   // the same in two halves: queue the build ... 
int gpu_build_octree_finish(msm_mesh *m);  // ... wait for it (one build at a time per context)
int ensure_tree_begin(msm_mesh *m);        // api.cpp: starts the GPU build of an invalid tree (no-op otherwise); ensure_tree() completes it  // octree_kernels.hip: the same tree built in HBM from the mesh's device coordinates (MSM_ERR_CAPACITY: use the host build)
bool mesh_tree_on_gpu(const msm_mesh *m);  // api.cpp: is this mesh's tree built by the GPU kernels (large meshes) or on the host
int finish_tree(msm_mesh *m);       // what follows either build: validity flags, generation
int ensure_masks(msm_mesh *m);  // + the per-leaf sub-cell masks the cost kernels use (built on the GPU)
int ensure_rays(msm_mesh *m, bool wait = false);   // + the ray table of a simple surface (unary table kernels); see api.cpp
DevTree dev_tree(const msm_mesh *m);
int check_status(msm_ctx *ctx, const char *what);  // sync + read kernel status
int ctx_io_pinned(msm_ctx *ctx, size_t bytes, void **out);  // grow-only pinned scratch for small per-call transfers
// device address of [p, p + bytes) when it lies in a msm_host_alloc block of this context, else nullptr
void *ctx_mapped(msm_ctx *ctx, const void *p, size_t bytes);
int ctx_flag(msm_ctx *ctx);  // makes ctx->h_flag / d_flag_map available
int drop_ctx_pending(msm_ctx *ctx);  // cost_cliques.cpp: the label step queued ahead on this context, if any, is waited for and discarded (see msm_ctx::pending_cost)
// host -> device on the context's stream: from where it lies when src is pinned memory of this context (msm_host_alloc / msm_host_register; complete on
// return, the caller may write the array again), through the staging blocks otherwise (src consumed on return, the copy queued; nothing waits)
int upload_staged(msm_ctx *ctx, void *dst, const void *src, size_t bytes);
// stager.cpp (stage_h2d / stage_d2h are declared in devbuf.hpp)
void stager_create(msm_ctx *ctx);
void stager_destroy(msm_ctx *ctx);  // the device is idle
void stager_stats(msm_ctx *ctx, int64_t out[4]);  // blocks, bytes, blocks ever allocated, waits for a busy block
int ctx_sync(msm_ctx *ctx);         // hipStreamSynchronize(ctx->stream) + what stage_d2h fetched goes to its destinations
void stage_deliver(msm_ctx *ctx);   // the second half alone (the stream has been synchronised by other means)
void stage_forget(msm_ctx *ctx);     // drops them instead
// An entry point that fails between a device -> host copy and its synchronisation must not leave the delivery behind: a later synchronisation would write
// into a local that is gone, or into an array the caller freed after the failed call.  Every error return starts at fail() on the thread that queued the
// copy, which calls this: the deliveries still pending on the context this thread last fetched through are dropped.
void stage_on_failure();
}  // namespace msm
