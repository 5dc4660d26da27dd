// octree_kernels.hip -- newresampler::Octree (R/octree.cpp:31-141, R/node.cpp) built on the GPU, level by level.
//
// The reference inserts the triangles one by one; the tree that results is decided per node by one scan of the triangles
// whose boxes overlap it, in ascending id (octree.cpp: Builder::build_down states why): a node becomes internal iff the
// split heuristic `num_split > 0 && total_size < 3 n` holds at some insertion from the 50th on, and its children then
// receive, in order, the triangles of its WHOLE list that overlap them.  That is a level-synchronous computation:
//   k_oct_decide   a wavefront per open node scans its list (wave prefix sums of the heuristic's two running sums, stopping at
//                  the first entry that triggers the split);
//   k_oct_count    a workgroup per 256-entry chunk of a splitting node counts what each child receives from the chunk;
//   k_oct_chunk_scan a wavefront per splitting node turns its chunks' counts into prefixes (where each chunk's share of a child's
//                  list starts) and totals;
//   k_oct_scan     one workgroup per tree turns the decisions of the level into node numbers, list offsets, chunk tables and leaf
//                  slots (prefix sums in node order: the tree's numbering does not depend on scheduling);
//   k_oct_fill     a workgroup per chunk writes its entries into the children's lists (ballot-ordered compaction behind the
//                  chunk's start: ids stay ascending) or, for a leaf, into the leaf array (padded to 8 with -1).
// The levels are queued in batches without looking at the outcome in between; the build comes in two halves (begin / finish) for
// callers with several streams, and as a FOREST (gpu_build_forest): B coordinate sets over one triangle list, the tree as the
// second grid dimension of every launch -- the L rotated copies of a gMSM subject's data mesh, the S control grids of a group.
// Child boxes are exact halvings of (-101, 101), so every box is reproduced bit for bit from (lower corner, edge).
// The host build (octree.cpp) takes 3.5 ms per ico6 mesh on sixteen threads -- every iteration of a registration builds one
// for the moved source, gMSM nineteen per subject -- plus the upload of its arrays; this one leaves them where the search
// kernels read them.  tests/test_gpu_search.py compares the leaves (boxes and ordered lists) of both builds.
#include <algorithm>
#include <cstring>
#include <functional>
#include <vector>

#include "kernels.hpp"

namespace msm {

namespace {

constexpr int kWave = 64;
enum { C_NNODES = 0, C_NOPEN, C_ARENA, C_NMASK, C_MAXDEPTH, C_OVERFLOW, C_REFS, C_MAXLEAF, C_NLEAVES, C_NCHUNK, C_NCHUNK_NEXT, C_SCAN_DONE, C_SCAN_OVER,
       C_SNAP_NOPEN, C_SNAP_NNODES, C_SNAP_ARENA, C_SNAP_NMASK, C_SCAN_TICKET, C_COUNT };  // C_SNAP_*: the level's starting values, see k_oct_decide; C_SCAN_TICKET: k_oct_scan

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}

// xyz: component a of vertex i of tree b at xyz[a * comp + b * tree + i] (one mesh: comp = V, tree = 0)
__global__ __launch_bounds__(256) void k_oct_boxes(const double *__restrict__ xyz, size_t comp, size_t tree, const int32_t *__restrict__ tri, int T, double *__restrict__ box,
                                                    int *__restrict__ list, size_t s_box, size_t s_list) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    xyz += blockIdx.y * tree;
    box += blockIdx.y * s_box;
    list += blockIdx.y * s_list;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = xyz[(size_t)a * comp + tri[t]];
    for (int k = 1; k < 3; ++k)
        for (int a = 0; a < 3; ++a) {
            const double c = xyz[(size_t)a * comp + tri[(size_t)k * T + t]];
            if (c < lo[a]) lo[a] = c;
            if (c > hi[a]) hi[a] = c;
        }
    for (int a = 0; a < 3; ++a) box[(size_t)6 * t + a] = lo[a], box[(size_t)6 * t + 3 + a] = hi[a];
    list[t] = t;  // the root's list
}

constexpr int kChunk = 64;  // list entries a WAVEFRONT of k_oct_count / k_oct_fill takes (round 5: 256 per workgroup -- at the deep levels nearly every node's list is shorter than a
                            // wavefront, three of a workgroup's four wavefronts idled through two barriers per chunk, and a workgroup had one chunk's dependent loads in flight)

struct OctWork {
    const double *box;     // 6 per triangle
    int4 *node;            // the mesh's node array (FlatOctree::node layout)
    int32_t *parent;
    double4 *nodebox;      // lower corner, edge
    int *counters;
    int *open_node[2];     // node ids of the level's open nodes / of the next level's
    int *open_off[2];      // start of each open node's list
    int *open_len[2];
    int *open_chunk[2];    // first chunk of each open node
    int *chunk_open[2];    // per chunk: its open node (index into open_*), and where it starts in that node's list
    int *chunk_beg[2];
    int *list[2];          // the lists themselves
    unsigned char *flags;  // per entry of list[cur]: which children of its (splitting) node it goes to -- k_oct_count works them out from the 48-byte box, k_oct_fill reads the byte
    int *split;            // per open node
    int *cc;               // 8 per chunk: what each child receives from this chunk, then (after k_oct_chunk_scan) where the chunk's share starts
    int *ctot;             // 8 per open node: what each child of a splitting node receives in all
    int32_t *leaf_tri;     // the mesh's leaf array
    int *agg;              // k_oct_scan: 8 ints per 1024 open nodes (7 sums + the level that published them), see there
    int cap_nodes, cap_refs, cap_arena, cap_open, cap_chunks;
    // a forest (gpu_build_forest): tree blockIdx.y of the launch uses the arrays `stride` elements further on; all zero for one tree
    size_t s_box, s_node, s_cnt, s_ints, s_leaf;  // s_ints: every int work array of a tree lies in one block, the blocks s_ints apart
};

// the arrays of this workgroup's tree
__device__ __forceinline__ OctWork tree_view(OctWork w) {
    const size_t b = blockIdx.y;
    if (b == 0) return w;
    w.box += b * w.s_box;
    w.node += b * w.s_node;
    w.parent += b * w.s_node;
    w.nodebox += b * w.s_node;
    w.counters += b * w.s_cnt;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w.open_node[k] += b * w.s_ints;
        w.open_off[k] += b * w.s_ints;
        w.open_len[k] += b * w.s_ints;
        w.open_chunk[k] += b * w.s_ints;
        w.chunk_open[k] += b * w.s_ints;
        w.chunk_beg[k] += b * w.s_ints;
        w.list[k] += b * w.s_ints;
    }
    w.split += b * w.s_ints;
    w.flags += b * w.s_ints * sizeof(int);
    w.ctot += b * w.s_ints;
    w.cc += b * w.s_ints;
    w.agg += b * w.s_ints;
    w.leaf_tri += b * w.s_leaf;
    return w;
}

// a triangle's box (lower corner, upper corner) as three 16-byte loads: the six 8-byte loads the compiler makes of `bx[a]` (it only knows the array to be 8-byte
// aligned) are six L1 look-ups per lane of a divergent gather -- the boxes start on 16-byte boundaries (48-byte records in a pool block)
struct Box6 {
    double v[6];
};
__device__ __forceinline__ Box6 load_box(const double *box, int t) {
    const double2 *q = reinterpret_cast<const double2 *>(box + (size_t)6 * t);
    const double2 a = q[0], b = q[1], c = q[2];
    Box6 r;
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = b.x, r.v[3] = b.y, r.v[4] = c.x, r.v[5] = c.y;
    return r;
}

// the 8 overlap flags of a triangle box against the children of a node (Node::can_contain on each child, R/node.cpp:108-116):
// per axis the child box is the parent's [lower, middle] or [middle, upper]
__device__ __forceinline__ unsigned child_flags(const double *bx, const double lo[3], const double mid[3], const double hi[3]) {
    unsigned half[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        half[a] = (!(bx[3 + a] < lo[a] || bx[a] > mid[a]) ? 1u : 0u) | (!(bx[3 + a] < mid[a] || bx[a] > hi[a]) ? 2u : 0u);
    unsigned f = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if ((half[0] >> ((c >> 2) & 1) & 1u) && (half[1] >> ((c >> 1) & 1) & 1u) && (half[2] >> (c & 1) & 1u)) f |= 1u << c;
    return f;
}

__device__ __forceinline__ void node_box(const double4 b, double lo[3], double mid[3], double hi[3]) {
    lo[0] = b.x, lo[1] = b.y, lo[2] = b.z;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        hi[a] = lo[a] + b.w;               // exact: boxes are dyadic subdivisions of (-101, 101)
        mid[a] = (lo[a] + hi[a]) / 2.0;    // Node::bounds[a][1], R/node.cpp:40-44
    }
}

// Does the node split?  Octree::add_triangle, R/octree.cpp:65-131: running total_size / num_split over the node's list in id
// order, tested from the 50th entry on.  A wavefront per open node; the scan stops at the first entry that triggers the split
// (for the large nodes of the top levels that is within the first chunk).
template <int cur>  // which of the two open lists this level reads (a template parameter: a run-time index into the views' pointer pairs sent the kernels to scratch)
__global__ __launch_bounds__(256) void k_oct_decide(OctWork w) {
    w = tree_view(w);
    const int nopen = w.counters[C_NOPEN];
    // The level's starting values for k_oct_scan.  That kernel's last workgroup overwrites C_NOPEN / C_NNODES / C_ARENA / C_NMASK with the
    // next level's while others of its workgroups may not have started yet (two processes sharing the GPU: a workgroup scheduled late read the
    // NEW open count and went to work on a block that did not exist) -- so its workgroups read this copy, which nothing touches while it runs.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.counters[C_SNAP_NOPEN] = nopen;
        w.counters[C_SNAP_NNODES] = w.counters[C_NNODES];
        w.counters[C_SNAP_ARENA] = w.counters[C_ARENA];
        w.counters[C_SNAP_NMASK] = w.counters[C_NMASK];
        w.counters[C_SCAN_TICKET] = 0;  // k_oct_scan's workgroups draw their logical blocks of this level from here
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int o = wave; o < nopen; o += nwaves) {
        const int n = w.open_node[cur][o], off = w.open_off[cur][o], len = w.open_len[cur][o];
        // the level's chunk tables (round 5: k_oct_scan wrote them for the next level, a thread walking its node's children chunk by chunk -- with
        // 64-entry chunks the nodes of levels 2 and 3 have hundreds, and the scan took 200 us there)
        for (int j = lane, k0 = w.open_chunk[cur][o], nk = (len + kChunk - 1) / kChunk; j < nk; j += kWave) {
            w.chunk_open[cur][k0 + j] = o;
            w.chunk_beg[cur][k0 + j] = j * kChunk;
        }
        const int *list = w.list[cur] + off;
        double lo[3], mid[3], hi[3];
        node_box(w.nodebox[n], lo, mid, hi);
        bool splits = false;
        if (len >= kMaxTriangles) {
            int tot = 0, ns = 0;
            for (int base = 0; base < len && !splits; base += kWave) {
                const int i = base + lane;
                int s = 0, q = 0;
                if (i < len) {
                    const Box6 b6 = load_box(w.box, list[i]);
                    const double *bx = b6.v;
                    s = 8;
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        if ((bx[d] < mid[d]) == (bx[3 + d] < mid[d])) s >>= 1;
                    q = s != 8 ? 1 : 0;
                }
                const int ts = tot + wave_incl_scan(s, lane), nq = ns + wave_incl_scan(q, lane);
                const bool hit = i < len && i + 1 >= kMaxTriangles && nq > 0 && ts < 3 * (i + 1);
                splits = __any(hit);
                tot = __shfl(ts, kWave - 1, kWave);
                ns = __shfl(nq, kWave - 1, kWave);
            }
        }
        if (lane == 0) w.split[o] = splits ? 1 : 0;
    }
}

// per chunk of a splitting node: how many of its entries each child receives
template <int cur>  // which of the two open lists this level reads (a template parameter: a run-time index into the views' pointer pairs sent the kernels to scratch)
__global__ __launch_bounds__(256) void k_oct_count(OctWork w) {
    w = tree_view(w);
    const int nchunks = w.counters[cur ? C_NCHUNK_NEXT : C_NCHUNK];  // one count per open list: the scan of a level writes the other one
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int k = wave; k < nchunks; k += nwaves) {  // wavefront-uniform
        const int o = w.chunk_open[cur][k];
        if (!w.split[o]) continue;
        const int n = w.open_node[cur][o], beg = w.chunk_beg[cur][k], len = w.open_len[cur][o];
        const int *list = w.list[cur] + w.open_off[cur][o];
        double lo[3], mid[3], hi[3];
        node_box(w.nodebox[n], lo, mid, hi);
        const int i = beg + lane;
        unsigned f = 0u;
        if (i < len) {
            const Box6 b6 = load_box(w.box, list[i]);
            f = child_flags(b6.v, lo, mid, hi);
        }
        if (i < len) w.flags[w.open_off[cur][o] + i] = (unsigned char)f;
        int mine = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int pc = __popcll(__ballot((f >> c) & 1u));
            if (lane == c) mine = pc;
        }
        if (lane < 8) w.cc[8 * (size_t)k + lane] = mine;
    }
}

// per splitting node: its chunks' counts become exclusive prefixes (where each chunk's share of a child's list starts), the sums
// go to ctot.  A wavefront per node, 64 chunks at a time -- the root of an ico6 mesh has 320 chunks, and one thread of
// k_oct_scan walking them with dependent loads and stores took up to 100 us per level.
template <int cur>  // which of the two open lists this level reads (a template parameter: a run-time index into the views' pointer pairs sent the kernels to scratch)
__global__ __launch_bounds__(256) void k_oct_chunk_scan(OctWork w) {
    w = tree_view(w);
    const int nopen = w.counters[C_NOPEN];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int o = wave; o < nopen; o += nwaves) {
        if (!w.split[o]) continue;
        const int len = w.open_len[cur][o], k0 = w.open_chunk[cur][o], nk = (len + kChunk - 1) / kChunk;
        int carry[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int base = 0; base < nk; base += kWave) {  // uniform
            const int j = base + lane;
            int v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (j < nk) {
                const int4 a = *reinterpret_cast<const int4 *>(w.cc + 8 * (size_t)(k0 + j)), b = *reinterpret_cast<const int4 *>(w.cc + 8 * (size_t)(k0 + j) + 4);
                v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
            }
            int e[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int incl = wave_incl_scan(v[c], lane);
                e[c] = carry[c] + incl - v[c];
                carry[c] += __shfl(incl, kWave - 1, kWave);
            }
            if (j < nk) {
                *reinterpret_cast<int4 *>(w.cc + 8 * (size_t)(k0 + j)) = make_int4(e[0], e[1], e[2], e[3]);
                *reinterpret_cast<int4 *>(w.cc + 8 * (size_t)(k0 + j) + 4) = make_int4(e[4], e[5], e[6], e[7]);
            }
        }
        if (lane == 0) {
            *reinterpret_cast<int4 *>(w.ctot + 8 * (size_t)o) = make_int4(carry[0], carry[1], carry[2], carry[3]);
            *reinterpret_cast<int4 *>(w.ctot + 8 * (size_t)o + 4) = make_int4(carry[4], carry[5], carry[6], carry[7]);
        }
    }
}

// block-wide exclusive scans of kN values per thread at once (1024 threads): v[] becomes the exclusive prefixes, total[] the sums
template <int kN>
__device__ void block_excl_scan_n(int v[kN], int total[kN]) {
    __shared__ int s_w[kN][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    int incl[kN];
#pragma unroll
    for (int q = 0; q < kN; ++q) {
        incl[q] = wave_incl_scan(v[q], lane);
        if (lane == kWave - 1) s_w[q][wv] = incl[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kN; ++q) {
        int before = 0, all = 0;
        for (int k = 0; k < nw; ++k) {
            const int t = s_w[q][k];
            before += k < wv ? t : 0;
            all += t;
        }
        v[q] = incl[q] - v[q] + before;
        total[q] = all;
    }
    __syncthreads();
}

// block-wide exclusive scan of one value per thread (1024 threads), returns the exclusive prefix; *total = sum
__device__ int block_excl_scan(int v, int *total) {
    __shared__ int s_w[16];
    __shared__ int s_tot;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v, lane);
    if (lane == kWave - 1) s_w[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
            const int t = s_w[k];
            s_w[k] = acc;
            acc += t;
        }
        s_tot = acc;
    }
    __syncthreads();
    const int excl = incl - v + s_w[wv];
    *total = s_tot;
    __syncthreads();
    return excl;
}

// The level's decisions become node numbers, list offsets, chunk tables and leaf slots -- prefix sums in open-node order, so the numbering
// of the tree does not depend on scheduling.  One workgroup did this for a whole level (up to 95 us at the deep levels of an ico6 mesh:
// twelve rounds of 1024 nodes, each thread then writing its node's eight children and their chunk tables).  Now up to kScanBlocks
// workgroups per tree share the level: logical block lb = 1024 consecutive open nodes; a workgroup scans its block, publishes the block's
// seven sums (w.agg, tagged with the level), waits for the sums of the blocks before it (published before anything is waited for) and writes
// its nodes.  The last workgroup to finish adds everything up for the counters.
// A workgroup DRAWS its logical blocks from a ticket counter (round 5; until then block lb belonged to workgroup lb % gridDim.x, and a workgroup in its
// second round waited for first-round blocks of workgroups that might not have been dispatched yet -- with two to four set-up pipelines building forests
// side by side the launches are no longer resident as a whole, and a wait that ran out sent the subject down the slow per-label path, silently: ADVICE r4).
// A ticket is only ever lower than one's own if its holder drew it earlier, i.e. is running and publishes its sums before it waits for anything: the
// waits end whatever part of the launch is resident, and the numbering (prefix sums in open-node order) does not depend on who scans which block.
constexpr int kScanThreads = 256;  // threads of a k_oct_scan workgroup = open nodes of a logical block (round 5: 1024 -- sixteen wavefronts that need a whole CU's
                                   // worth of slots at once waited long for one while two set-up pipelines and their batch streams kept the CUs busy)
constexpr int kScanBlocks = 64;
template <int cur>
__global__ __launch_bounds__(kScanThreads) void k_oct_scan(OctWork w, int depth) {
    w = tree_view(w);
    const int nopen = w.counters[C_SNAP_NOPEN];  // the copies k_oct_decide made for this level (see there)
    const int nxt = cur ^ 1, tid = threadIdx.x;
    if (nopen == 0) {  // the tree is complete: the levels still queued find no chunks either
        if (blockIdx.x == 0 && tid == 0) w.counters[nxt ? C_NCHUNK_NEXT : C_NCHUNK] = 0;
        return;
    }
    const int nnodes0 = w.counters[C_SNAP_NNODES], arena0 = w.counters[C_SNAP_ARENA], nmask0 = w.counters[C_SNAP_NMASK];
    const int nlog = (nopen + kScanThreads - 1) / kScanThreads, epoch = depth + 1;
    __shared__ int s_over, s_max, s_last, s_lb;
    for (;;) {
        if (tid == 0) s_over = 0, s_max = 0, s_lb = atomicAdd(&w.counters[C_SCAN_TICKET], 1);
        __syncthreads();
        const int lb = s_lb;  // uniform
        if (lb >= nlog) break;
        const int o = lb * kScanThreads + tid;
        const bool in = o < nopen;
        const int sp = in ? w.split[o] : 0;
        const int len = in ? w.open_len[cur][o] : 0;
        const bool leaf = in && !sp;
        // what the children of a splitting node receive (k_oct_chunk_scan summed its chunks)
        int ct[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (sp) {
            const int4 a = *reinterpret_cast<const int4 *>(w.ctot + 8 * (size_t)o), b = *reinterpret_cast<const int4 *>(w.ctot + 8 * (size_t)o + 4);
            ct[0] = a.x, ct[1] = a.y, ct[2] = a.z, ct[3] = a.w, ct[4] = b.x, ct[5] = b.y, ct[6] = b.z, ct[7] = b.w;
        }
        int ctot = 0, cchunks = 0;
        for (int c = 0; c < 8; ++c) ctot += ct[c], cchunks += (ct[c] + kChunk - 1) / kChunk;
        const int hasmask = leaf && len >= 1 && len <= 64;
        int sc[7] = {sp, leaf ? ((len + 7) & ~7) : 0, hasmask, ctot, cchunks, leaf ? len : 0, leaf ? 1 : 0}, tot[7];
        block_excl_scan_n<7>(sc, tot);
        // publish this block's sums, then collect those of the blocks before it (a level of up to 1024 open nodes has neither)
        int before[7] = {0, 0, 0, 0, 0, 0, 0}, carry[7] = {0, 0, 0, 0, 0, 0, 0};
        if (nlog > 1) {
            if (tid < 7) __hip_atomic_store(w.agg + 8 * (size_t)lb + tid, tot[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(w.agg + 8 * (size_t)lb + 7, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int j = tid; j < lb; j += kScanThreads) {
            // (bounded: should a block before this one never report -- it cannot, see above -- the build gives up and the host builds the tree)
            // (polled relaxed, one acquire fence behind the wait: an acquire load per round invalidated the vector L1 of a CU this kernel shares with the
            // other streams' kernels on every round)
            for (int spin = 0; __hip_atomic_load(w.agg + 8 * (size_t)j + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin) {
                if (spin > (1 << 22)) {
                    atomicOr(&w.counters[C_SCAN_OVER], 2);  // 2: a wait that ran out (reported under MSMHIP_TIMING), 1: an array that is too small
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // pairs with the release store of the sums' tag
#pragma unroll
            for (int q = 0; q < 7; ++q) before[q] += __hip_atomic_load(w.agg + 8 * (size_t)j + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lb > 0) block_excl_scan_n<7>(before, carry);  // carry: the sums over all blocks before this one
        const int rank = carry[0] + sc[0], aoff = carry[1] + sc[1], mblk = carry[2] + sc[2], loff = carry[3] + sc[3], koff = carry[4] + sc[4];
        if (leaf) atomicMax(&s_max, len);
        if (in) {
            const int n = w.open_node[cur][o];
            if (sp) {
                const int child_base = nnodes0 + 8 * rank;
                if (child_base + 8 > w.cap_nodes || 8 * rank + 8 > w.cap_open || loff + ctot > w.cap_refs || koff + cchunks > w.cap_chunks) {
                    s_over = 1;
                } else {
                    w.node[n] = make_int4(child_base, 0, -1, depth);
                    const double4 pb = w.nodebox[n];
                    const double h = pb.w / 2.0;  // the children's edge; lower corners = the parent's lower bound or its middle
                    int lo = loff, kk = koff;
                    for (int c = 0; c < 8; ++c) {
                        const int id = child_base + c, oc = 8 * rank + c;
                        w.parent[id] = n;
                        const double cx = ((c >> 2) & 1) ? (pb.x + (pb.x + pb.w)) / 2.0 : pb.x;
                        const double cy = ((c >> 1) & 1) ? (pb.y + (pb.y + pb.w)) / 2.0 : pb.y;
                        const double cz = (c & 1) ? (pb.z + (pb.z + pb.w)) / 2.0 : pb.z;
                        w.nodebox[id] = make_double4(cx, cy, cz, h);
                        w.node[id] = make_int4(-1, 0, -1, depth + 1);  // provisional: an empty leaf (the next level decides it)
                        w.open_node[nxt][oc] = id;
                        w.open_off[nxt][oc] = lo;
                        w.open_len[nxt][oc] = ct[c];
                        w.open_chunk[nxt][oc] = kk;  // (the chunk tables themselves: the next level's k_oct_decide)
                        kk += (ct[c] + kChunk - 1) / kChunk;
                        lo += ct[c];
                    }
                }
            } else {
                if (arena0 + aoff + ((len + 7) & ~7) > w.cap_arena) s_over = 1;
                else w.node[n] = make_int4(-len - 1, arena0 + aoff, hasmask ? nmask0 + mblk : -1, depth);  // k_oct_fill finds its slot here
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (s_over) atomicOr(&w.counters[C_SCAN_OVER], 1);
            if (s_max) atomicMax(&w.counters[C_MAXLEAF], s_max);
            s_last = atomicAdd(&w.counters[C_SCAN_DONE], 1) == nlog - 1;
        }
        __syncthreads();
        if (s_last) {  // every block of the level has written its nodes: the level's totals
            int all[7] = {0, 0, 0, 0, 0, 0, 0}, sum[7];
            if (nlog > 1) {
                for (int j = tid; j < nlog; j += kScanThreads)
#pragma unroll
                    for (int q = 0; q < 7; ++q) all[q] += __hip_atomic_load(w.agg + 8 * (size_t)j + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                block_excl_scan_n<7>(all, sum);
            } else {
#pragma unroll
                for (int q = 0; q < 7; ++q) sum[q] = tot[q];
            }
            if (tid == 0) {
                w.counters[C_SCAN_DONE] = 0;
                if (const int over = __hip_atomic_load(&w.counters[C_SCAN_OVER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    w.counters[C_OVERFLOW] = over;
                    w.counters[C_NOPEN] = 0;
                    w.counters[C_NCHUNK] = w.counters[C_NCHUNK_NEXT] = 0;
                } else {
                    w.counters[C_NNODES] = nnodes0 + 8 * sum[0];
                    w.counters[C_NOPEN] = 8 * sum[0];
                    w.counters[nxt ? C_NCHUNK_NEXT : C_NCHUNK] = sum[4];  // the chunk count of the list the next level reads
                    w.counters[C_ARENA] = arena0 + sum[1];
                    w.counters[C_NMASK] = nmask0 + sum[2];
                    w.counters[C_MAXDEPTH] = depth;
                    w.counters[C_REFS] += sum[5];
                    w.counters[C_NLEAVES] += sum[6];
                }
            }
        }
    }
}

// per chunk: a splitting node's entries go to its children's lists (ballot-ordered compaction behind the chunk's start in each
// child list: ids stay ascending); a leaf's entries go to the leaf array, padded to a multiple of eight with -1
template <int cur>  // which of the two open lists this level reads (a template parameter: a run-time index into the views' pointer pairs sent the kernels to scratch)
__global__ __launch_bounds__(256) void k_oct_fill(OctWork w) {
    w = tree_view(w);
    if (w.counters[C_OVERFLOW]) return;
    const int nchunks = w.counters[cur ? C_NCHUNK_NEXT : C_NCHUNK];
    const int nxt = cur ^ 1;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int k = wave; k < nchunks; k += nwaves) {  // wavefront-uniform
        const int o = w.chunk_open[cur][k];
        const int n = w.open_node[cur][o], beg = w.chunk_beg[cur][k], len = w.open_len[cur][o];
        const int *list = w.list[cur] + w.open_off[cur][o];
        const int4 nd = w.node[n];
        const int i = beg + lane;
        if (nd.x < 0) {
            const int cnt = -nd.x - 1, padded = (cnt + 7) & ~7;
            if (i < padded) w.leaf_tri[nd.y + i] = i < cnt ? list[i] : -1;
            continue;
        }
        const int t = i < len ? list[i] : 0;
        const unsigned f = i < len ? (unsigned)w.flags[w.open_off[cur][o] + i] : 0u;  // k_oct_count's (round 5: both kernels read the entry's box, 48 bytes gathered, to get these eight bits)
        const int oc0 = nd.x - w.open_node[nxt][0];  // the children's places in the next open list (consecutive)
        int *out = w.list[nxt];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned long long bal = __ballot((f >> c) & 1u);
            if (!((f >> c) & 1u)) continue;
            const int at = w.open_off[nxt][oc0 + c] + w.cc[8 * (size_t)k + c];
            out[at + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        }
    }
}

// the state before level 0: node 0 with the cube (-101, 101) and every triangle (k_oct_boxes wrote the list), in chunks
__global__ __launch_bounds__(256) void k_oct_init(OctWork w, int T, int root_chunks) {
    w = tree_view(w);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < root_chunks) {
        w.chunk_open[0][i] = 0;
        w.chunk_beg[0][i] = i * kChunk;
    }
    if (i <= C_COUNT) w.counters[i] = i == C_NNODES || i == C_NOPEN ? 1 : i == C_NCHUNK ? root_chunks : 0;
    for (int j = i; j < 8 * (w.cap_open / kScanThreads + 2); j += gridDim.x * blockDim.x) w.agg[j] = 0;  // no level has published sums yet
    if (i == 0) {
        w.node[0] = make_int4(-1, 0, -1, 0);
        w.nodebox[0] = make_double4(-kBounds, -kBounds, -kBounds, 2 * kBounds);
        w.parent[0] = -1;
        w.open_node[0][0] = 0;
        w.open_off[0][0] = 0;
        w.open_len[0][0] = T;
        w.open_chunk[0][0] = 0;
    }
}

// the same for every tree of a forest, the depth taken from the tree's counters
__global__ __launch_bounds__(256) void k_oct_grid_forest(const int4 *__restrict__ node, size_t s_node, const int *__restrict__ counters, size_t s_cnt, int32_t *__restrict__ grid,
                                                          size_t s_grid) {
    node += blockIdx.y * s_node;
    grid += blockIdx.y * s_grid;
    const int gd = min(counters[blockIdx.y * s_cnt + C_MAXDEPTH], 6), G = 1 << gd;
    const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (size_t)G * G * G) return;
    const int iz = (int)(cell % G), iy = (int)((cell / G) % G), ix = (int)(cell / ((size_t)G * G));
    int n = 0;
    for (int d = 0; d < gd; ++d) {
        const int4 nd = node[n];
        if (nd.x < 0) break;
        const int sh = gd - 1 - d;
        n = nd.x + 4 * ((ix >> sh) & 1) + 2 * ((iy >> sh) & 1) + ((iz >> sh) & 1);
    }
    grid[cell] = n;
}

// dense top grid (FlatOctree::grid): the node a point of each depth-gd cell reaches after gd levels of descent, or the leaf met earlier
__global__ __launch_bounds__(256) void k_oct_grid(const int4 *__restrict__ node, int gd, int32_t *__restrict__ grid) {
    const int G = 1 << gd;
    const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (size_t)G * G * G) return;
    const int iz = (int)(cell % G), iy = (int)((cell / G) % G), ix = (int)(cell / ((size_t)G * G));
    int n = 0;
    for (int d = 0; d < gd; ++d) {
        const int4 nd = node[n];
        if (nd.x < 0) break;
        const int sh = gd - 1 - d;
        n = nd.x + 4 * ((ix >> sh) & 1) + 2 * ((iy >> sh) & 1) + ((iz >> sh) & 1);
    }
    grid[cell] = n;
}

}  // namespace

// Builds the search tree of m's current device coordinates into m's device arrays.  Returns MSM_OK, or MSM_ERR_CAPACITY when
// the tree outgrows the preallocated arrays (degenerate meshes: the caller falls back to the host build).
namespace {
struct OctJob {  // a build between gpu_build_octree_begin and _finish
    OctWork w;
    int cur = 0, depth = 0;
    int trees = 1;
    int *h_counters = nullptr;  // pinned, trees x (C_COUNT + 1)
};
constexpr int kMaxLevels = 24, kNextBatch = 2;
// levels queued before the first look: an icosphere of T triangles finishes at depth log4(T) - 2 (ico6: 6, ico5: 5, ico4: 4) and the loop runs once
// per depth including the last; a deeper (warped, irregular) tree gets the rest in batches of kNextBatch after a look at the counters
static int first_batch(int T) {
    int l4 = 0;
    for (long long t = T; t >= 4; t >>= 2) ++l4;
    return std::max(3, std::min(10, l4 - 1));
}
// Levels are queued in batches without looking at the outcome in between: a level with nothing open costs five empty launches, a
// look costs a round trip (first_batch sizes the first one from the number of triangles).
int queue_levels(msm_ctx *ctx, OctJob &j, int count) {
    const int upto = std::min(kMaxLevels, j.depth + count);
    const unsigned B = (unsigned)j.trees;
    const unsigned div = B > 1 ? 4 : 1;  // a forest's trees share the machine
    // k_oct_scan's workgroups wait for one another's sums; they draw their blocks from a ticket counter, so the waits end whatever part of a launch is
    // resident (see there) -- the grid is sized for one 1024-thread workgroup per CU of this device and tree, more would only queue
    const unsigned scan_blocks = std::max(1u, std::min((unsigned)kScanBlocks, 4u * (unsigned)std::max(1, ctx->num_cus) / B));
    for (; j.depth < upto; ++j.depth) {
        if (j.cur == 0) {
            hipLaunchKernelGGL(k_oct_decide<0>, dim3(512 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_count<0>, dim3(1024 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_chunk_scan<0>, dim3(256 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_scan<0>, dim3(scan_blocks, B), dim3(kScanThreads), 0, ctx->stream, j.w, j.depth);
            hipLaunchKernelGGL(k_oct_fill<0>, dim3(1024 / div, B), dim3(256), 0, ctx->stream, j.w);
        } else {
            hipLaunchKernelGGL(k_oct_decide<1>, dim3(512 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_count<1>, dim3(1024 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_chunk_scan<1>, dim3(256 / div, B), dim3(256), 0, ctx->stream, j.w);
            hipLaunchKernelGGL(k_oct_scan<1>, dim3(scan_blocks, B), dim3(kScanThreads), 0, ctx->stream, j.w, j.depth);
            hipLaunchKernelGGL(k_oct_fill<1>, dim3(1024 / div, B), dim3(256), 0, ctx->stream, j.w);
        }
        j.cur ^= 1;
    }
    MSM_HIP(hipGetLastError());
    // a forest's counters are consecutive (s_cnt = C_COUNT + 1)
    MSM_HIP(hipMemcpyAsync(j.h_counters, j.w.counters, sizeof(int) * (B == 1 ? (size_t)C_COUNT : (size_t)B * (C_COUNT + 1)), hipMemcpyDeviceToHost, ctx->stream));
    return MSM_OK;
}
}  // namespace

// The build in two halves, so that a caller with several streams (group.cpp: the lanes of the gMSM set-up) can queue the levels of
// one mesh and go on with another before looking at the outcome.  begin: everything up to and including the first batch of
// levels, nothing waited for; finish: waits, queues more levels for deeper trees, then the grid, records and cones.  One build at a
// time per context (the scratch arrays and the counters' landing place belong to the context).
int gpu_build_octree_begin(msm_mesh *m) {
    msm_ctx *ctx = m->ctx;
    const int T = m->T, V = m->V;
    const int cap_nodes = T + 64, cap_refs = 6 * T + 256, cap_arena = 8 * T + 512, cap_open = cap_nodes, cap_chunks = cap_refs / kChunk + cap_open + 64;
    MSM_HIP(hipSetDevice(ctx->device));
    if ((size_t)6 * T > ctx->oct_cap_box) {
        if (ctx->oct_box) (void)msm::pool_free(ctx->oct_box);
        ctx->oct_box = nullptr;
        ctx->oct_cap_box = (size_t)6 * T + 1024;
        MSM_HIP(msm::pool_malloc((void **)&ctx->oct_box, ctx->oct_cap_box * sizeof(double)));
    }
    const size_t need_ints = (size_t)2 * cap_refs + (size_t)8 * cap_open + (size_t)cap_open + (size_t)8 * cap_open + (size_t)4 * cap_chunks + (size_t)8 * cap_chunks + 16 +
                             8 * ((size_t)cap_open / kScanThreads + 2) + ((size_t)cap_refs + 3) / 4;
    if (need_ints > ctx->oct_cap_ints) {
        if (ctx->oct_ints) (void)msm::pool_free(ctx->oct_ints);
        ctx->oct_ints = nullptr;
        ctx->oct_cap_ints = need_ints + 4096;
        MSM_HIP(msm::pool_malloc((void **)&ctx->oct_ints, ctx->oct_cap_ints * sizeof(int)));
    }
    if (!ctx->oct_counters) {
        MSM_HIP(msm::pool_malloc((void **)&ctx->oct_counters, sizeof(int) * (C_COUNT + 1)));
        MSM_HIP(hipHostMalloc((void **)&ctx->oct_hcounters, sizeof(int) * (C_COUNT + 1)));
    }
    struct {
        double *box;
        int *ints, *counters, *h_counters;
    } s{ctx->oct_box, ctx->oct_ints, ctx->oct_counters, ctx->oct_hcounters};
    auto grow = [&](void **p, size_t &cap, size_t need, size_t elem) -> hipError_t {
        if (need <= cap && *p) return hipSuccess;
        if (*p) (void)msm::pool_free(*p);
        *p = nullptr;
        cap = need;
        return msm::pool_malloc(p, cap * elem);
    };
    MSM_HIP(grow((void **)&m->d_node, m->cap_node, cap_nodes, sizeof(int4)));
    MSM_HIP(grow((void **)&m->d_parent, m->cap_parent, cap_nodes, sizeof(int32_t)));
    MSM_HIP(grow((void **)&m->d_nodebox, m->cap_box, cap_nodes, sizeof(double4)));
    MSM_HIP(grow((void **)&m->d_leaf_tri, m->cap_leaf, cap_arena, sizeof(int32_t)));
    MSM_HIP(grow((void **)&m->d_cone, m->cap_cone, cap_arena, sizeof(float4)));
    MSM_HIP(grow((void **)&m->d_rec, m->cap_rec, (size_t)T, sizeof(TriRec)));
    MSM_HIP(grow((void **)&m->d_grid, m->cap_grid, (size_t)64 * 64 * 64, sizeof(int32_t)));

    OctWork w;
    w.box = s.box;
    w.node = m->d_node;
    w.parent = m->d_parent;
    w.nodebox = m->d_nodebox;
    w.counters = s.counters;
    int *p = s.ints;
    w.list[0] = p, p += cap_refs;
    w.list[1] = p, p += cap_refs;
    for (int k = 0; k < 2; ++k) {
        w.open_node[k] = p, p += cap_open;
        w.open_off[k] = p, p += cap_open;
        w.open_len[k] = p, p += cap_open;
        w.open_chunk[k] = p, p += cap_open;
        w.chunk_open[k] = p, p += cap_chunks;
        w.chunk_beg[k] = p, p += cap_chunks;
    }
    w.split = p, p += cap_open;
    p += (4 - ((p - s.ints) & 3)) & 3;  // 16-byte alignment of the int4 accesses below
    w.ctot = p, p += (size_t)8 * cap_open;
    w.cc = p, p += (size_t)8 * cap_chunks;
    w.agg = p, p += 8 * ((size_t)cap_open / kScanThreads + 2);
    w.flags = reinterpret_cast<unsigned char *>(p);
    w.leaf_tri = m->d_leaf_tri;
    w.cap_nodes = cap_nodes, w.cap_refs = cap_refs, w.cap_arena = cap_arena, w.cap_open = cap_open, w.cap_chunks = cap_chunks;

    // root: node 0 with the cube (-101, 101) and every triangle
    const int root_chunks = (T + kChunk - 1) / kChunk;
    w.s_box = w.s_node = w.s_cnt = w.s_ints = w.s_leaf = 0;
    hipLaunchKernelGGL(k_oct_boxes, dim3((T + 255) / 256), dim3(256), 0, ctx->stream, m->d_xyz, (size_t)V, (size_t)0, m->d_tri, T, s.box, w.list[0], (size_t)0, (size_t)0);
    hipLaunchKernelGGL(k_oct_init, dim3((std::max(root_chunks, C_COUNT + 1) + 255) / 256), dim3(256), 0, ctx->stream, w, T, root_chunks);
    auto job = std::make_shared<OctJob>();
    job->w = w;
    job->h_counters = ctx->oct_hcounters;
    m->oct_job = job;
    return queue_levels(ctx, *job, first_batch(T));
}

int gpu_build_octree_finish(msm_mesh *m) {
    msm_ctx *ctx = m->ctx;
    if (!m->oct_job) return fail(MSM_ERR_STATE, "octree build: nothing was begun");
    const std::shared_ptr<void> keep = m->oct_job;
    OctJob &job = *static_cast<OctJob *>(keep.get());
    m->oct_job.reset();
    const int T = m->T, V = m->V;
    MSM_HIP(hipSetDevice(ctx->device));
    MSM_TRY(ctx_sync(ctx));
    while (ctx->oct_hcounters[C_NOPEN] != 0 && !ctx->oct_hcounters[C_OVERFLOW] && job.depth < kMaxLevels) {
        int st = queue_levels(ctx, job, kNextBatch);
        if (st) return st;
        MSM_TRY(ctx_sync(ctx));
    }
    struct {
        int *h_counters;
    } s{ctx->oct_hcounters};
    const int *hc = s.h_counters;
    if ((hc[C_OVERFLOW] & 2) && std::getenv("MSMHIP_TIMING"))
        fprintf(stderr, "  octree build: a k_oct_scan workgroup waited in vain for an earlier one (%d CUs, not all of the launch resident?): host build instead\n", ctx->num_cus);
    if (hc[C_OVERFLOW] || hc[C_NOPEN] != 0) return MSM_ERR_CAPACITY;
    FlatOctree &o = m->tree;
    o = FlatOctree{};  // no host arrays: the tree lives on the device (dev_nnodes etc. below describe it)
    o.dev_nnodes = hc[C_NNODES];
    o.dev_entries = hc[C_ARENA];
    o.nmask_blocks = hc[C_NMASK];
    o.grid_depth = std::min(hc[C_MAXDEPTH], 6);
    o.stats[0] = hc[C_NNODES], o.stats[1] = hc[C_NLEAVES], o.stats[2] = hc[C_MAXDEPTH], o.stats[3] = hc[C_REFS], o.stats[4] = hc[C_MAXLEAF];
    const int G = 1 << o.grid_depth;
    const size_t cells = (size_t)G * G * G;
    hipLaunchKernelGGL(k_oct_grid, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, m->d_node, o.grid_depth, m->d_grid);
    MSM_HIP(hipGetLastError());
    return launch_build_recs(ctx, m->d_xyz, V, m->d_tri, T, m->d_rec, m->d_tcone, m->d_leaf_tri, o.dev_entries, m->d_cone);
}

// B trees at once: the same triangle list over B coordinate sets (gMSM: a subject's data mesh rotated to every label), every level's
// kernels launched once with the tree as the second grid dimension.  The per-label chain of ~55 launches of a few microseconds
// each becomes one chain per subject.  Each tree is numbered by prefix sums over ITS open nodes (one workgroup of k_oct_scan per
// tree), so every tree is what gpu_build_octree builds for that coordinate set, node for node.
int gpu_build_forest(msm_ctx *ctx, Forest &f, const double *d_xyz, size_t comp_stride, size_t tree_stride, int V, const int32_t *d_tri, int T, int B) {
    if (B <= 0 || T <= 0) return fail(MSM_ERR_INVALID, "gpu_build_forest: bad arguments");
    MSM_HIP(hipSetDevice(ctx->device));
    const int cap_nodes = T + 64, cap_refs = 6 * T + 256, cap_arena = 8 * T + 512, cap_open = cap_nodes, cap_chunks = cap_refs / kChunk + cap_open + 64;
    const size_t per_ints = ((size_t)2 * cap_refs + (size_t)8 * cap_open + (size_t)cap_open + (size_t)8 * cap_open + (size_t)4 * cap_chunks + (size_t)8 * cap_chunks + 16 +
                             8 * ((size_t)cap_open / kScanThreads + 2) + ((size_t)cap_refs + 3) / 4 + 3) & ~(size_t)3;
    f.B = B, f.T = T, f.V = V;
    f.s_node = (size_t)cap_nodes, f.s_leaf = (size_t)cap_arena, f.s_rec = (size_t)T, f.s_grid = (size_t)64 * 64 * 64;
    MSM_HIP(f.node.ensure(f.s_node * B));
    MSM_HIP(f.parent.ensure(f.s_node * B));
    MSM_HIP(f.nodebox.ensure(f.s_node * B));
    MSM_HIP(f.leaf_tri.ensure(f.s_leaf * B));
    MSM_HIP(f.cone.ensure(f.s_leaf * B));
    MSM_HIP(f.rec.ensure(f.s_rec * B));
    MSM_HIP(f.tcone.ensure(f.s_rec * B));
    MSM_HIP(f.grid.ensure(f.s_grid * B));
    MSM_HIP(f.box.ensure((size_t)6 * T * B));
    MSM_HIP(f.ints.ensure(per_ints * B));
    MSM_HIP(f.counters.ensure((size_t)(C_COUNT + 1) * B));
    if (f.h_counters_cap < (size_t)(C_COUNT + 1) * B) {
        if (f.h_counters) (void)hipHostFree(f.h_counters);
        f.h_counters = nullptr;
        f.h_counters_cap = (size_t)(C_COUNT + 1) * B;
        MSM_HIP(hipHostMalloc((void **)&f.h_counters, sizeof(int) * f.h_counters_cap));
    }
    OctWork w;
    w.box = f.box.p;
    w.node = f.node.p;
    w.parent = f.parent.p;
    w.nodebox = f.nodebox.p;
    w.counters = f.counters.p;
    // within a tree's block of ints the arrays lie as in gpu_build_octree_begin; the blocks of consecutive trees per_ints apart,
    // so every array's stride is per_ints
    int *p = f.ints.p;
    w.list[0] = p, p += cap_refs;
    w.list[1] = p, p += cap_refs;
    for (int k = 0; k < 2; ++k) {
        w.open_node[k] = p, p += cap_open;
        w.open_off[k] = p, p += cap_open;
        w.open_len[k] = p, p += cap_open;
        w.open_chunk[k] = p, p += cap_open;
        w.chunk_open[k] = p, p += cap_chunks;
        w.chunk_beg[k] = p, p += cap_chunks;
    }
    w.split = p, p += cap_open;
    p += (4 - ((p - f.ints.p) & 3)) & 3;
    w.ctot = p, p += (size_t)8 * cap_open;
    w.cc = p, p += (size_t)8 * cap_chunks;
    w.agg = p, p += 8 * ((size_t)cap_open / kScanThreads + 2);
    w.flags = reinterpret_cast<unsigned char *>(p);
    w.leaf_tri = f.leaf_tri.p;
    w.cap_nodes = cap_nodes, w.cap_refs = cap_refs, w.cap_arena = cap_arena, w.cap_open = cap_open, w.cap_chunks = cap_chunks;
    w.s_box = (size_t)6 * T, w.s_node = f.s_node, w.s_cnt = (size_t)(C_COUNT + 1), w.s_ints = per_ints, w.s_leaf = f.s_leaf;
    const unsigned UB = (unsigned)B;
    const int root_chunks = (T + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(k_oct_boxes, dim3((T + 255) / 256, UB), dim3(256), 0, ctx->stream, d_xyz, comp_stride, tree_stride, d_tri, T, f.box.p, w.list[0], w.s_box, w.s_ints);
    hipLaunchKernelGGL(k_oct_init, dim3((std::max(root_chunks, C_COUNT + 1) + 255) / 256, UB), dim3(256), 0, ctx->stream, w, T, root_chunks);
    OctJob job;
    job.w = w;
    job.trees = B;
    job.h_counters = f.h_counters;
    // a forest that is rebuilt (gMSM: one per set-up pipeline, subject after subject, iteration after iteration) queues what its last build needed:
    // warped ico6 meshes go two levels deeper than the regular one, and a second batch costs a round trip plus launches the GPU waits for
    int st = queue_levels(ctx, job, std::min(kMaxLevels, std::max(first_batch(T), f.last_levels)));
    if (st) return st;
    // grid, records and cones read the counters on the device: queued behind the levels without a look at the outcome (round 5: the look was a round trip
    // on the critical path of every subject of a gMSM set-up); should a tree turn out to need more levels they run again
    auto finish = [&]() -> int {
        hipLaunchKernelGGL(k_oct_grid_forest, dim3((unsigned)((f.s_grid + 255) / 256), UB), dim3(256), 0, ctx->stream, f.node.p, f.s_node, f.counters.p, (size_t)(C_COUNT + 1),
                           f.grid.p, f.s_grid);
        MSM_HIP(hipGetLastError());
        return launch_build_recs_forest(ctx, d_xyz, comp_stride, tree_stride, d_tri, T, B, f.rec.p, f.tcone.p, f.s_rec, f.leaf_tri.p, f.cone.p, f.s_leaf, f.counters.p + C_ARENA,
                                        (size_t)(C_COUNT + 1));
    };
    st = finish();
    if (st) return st;
    MSM_TRY(ctx_sync(ctx));
    bool again = false;
    auto open_somewhere = [&] {
        for (int b = 0; b < B; ++b) {
            const int *hc = f.h_counters + (size_t)b * (C_COUNT + 1);
            if (hc[C_NOPEN] != 0 && !hc[C_OVERFLOW]) return true;
        }
        return false;
    };
    while (open_somewhere() && job.depth < kMaxLevels) {
        st = queue_levels(ctx, job, kNextBatch);
        if (st) return st;
        MSM_TRY(ctx_sync(ctx));
        again = true;
    }
    f.info.assign(B, Forest::Info{});
    f.last_levels = 0;
    for (int b = 0; b < B; ++b) f.last_levels = std::max(f.last_levels, f.h_counters[(size_t)b * (C_COUNT + 1) + C_MAXDEPTH] + 1);
    for (int b = 0; b < B; ++b) {
        const int *hc = f.h_counters + (size_t)b * (C_COUNT + 1);
        if ((hc[C_OVERFLOW] & 2) && std::getenv("MSMHIP_TIMING"))
            fprintf(stderr, "  forest build: a k_oct_scan workgroup of tree %d waited in vain for an earlier one (%d CUs): per-tree builds instead\n", b, ctx->num_cus);
        if (hc[C_OVERFLOW] || hc[C_NOPEN] != 0) return MSM_ERR_CAPACITY;
        f.info[b].nnodes = hc[C_NNODES];
        f.info[b].entries = hc[C_ARENA];
        f.info[b].grid_depth = std::min(hc[C_MAXDEPTH], 6);
    }
    return again ? finish() : MSM_OK;
}

DevTree forest_tree(const Forest &f, int b) {
    DevTree t{};
    t.node = f.node.p + (size_t)b * f.s_node;
    t.parent = f.parent.p + (size_t)b * f.s_node;
    t.leaf_tri = f.leaf_tri.p + (size_t)b * f.s_leaf;
    t.cone = f.cone.p + (size_t)b * f.s_leaf;
    t.rec = f.rec.p + (size_t)b * f.s_rec;
    t.grid = f.grid.p + (size_t)b * f.s_grid;
    t.grid_depth = f.info[b].grid_depth;
    t.simple = 0;
    t.mask = nullptr;
    t.nnodes = f.info[b].nnodes;
    t.ray_G = 0;
    t.ray_cell = nullptr, t.ray_tri = nullptr, t.ray_more = nullptr, t.ray_excl = nullptr;
    t.ray_r2lo = t.ray_r2hi = 0.0;
    return t;
}

int gpu_build_octree(msm_mesh *m, const std::function<void()> *overlap) {
    int st = gpu_build_octree_begin(m);
    if (st) return st;
    if (overlap && *overlap) (*overlap)();  // host work of the caller that does not depend on this tree, while the GPU builds it
    return gpu_build_octree_finish(m);
}

}  // namespace msm
