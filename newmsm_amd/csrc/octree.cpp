// octree.cpp -- host construction of the triangle search structure.
//
// newresampler::Octree (R/octree.cpp:31-141, R/node.cpp) grows a pointer tree by inserting triangles
// one at a time: a leaf that reaches MAX_TRIANGLES entries is split when the split heuristic holds,
// and its triangles are handed down to every child whose closed box overlaps their AABB.  Which leaf
// a query lands in -- and so which triangles are candidates -- depends on that insertion history, so
// the same incremental procedure is run here, on index-based nodes, and then laid out as flat arrays
// for the GPU:
//   node[]      int4 per node (first child, or leaf entry range + mask block + depth) -> descent is arithmetic + 1 load/level
//   leaf_tri[]  triangle ids of all leaves, contiguous per leaf, in insertion (= ascending id) order, padded to x8
//   cone[]      float4 per (padded) leaf entry: conservative bounding cone of the triangle (cheap reject test; filled on the
//               GPU together with recs[], kernels.hip k_build_recs / k_expand_cones)
//   mask[]      per leaf 64 x 64-bit: which entries a query inside each of the leaf's 4x4x4 sub-cells can hit
//               (filled on the GPU, kernels.hip k_build_masks)
//   recs[]      128-byte record per triangle for the exact test
//   ray table   (simple surfaces only) cube-map of directions -> candidate triangles, plus three float edge planes
//               per triangle: decides most queries without touching the tree, with the reference's result
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <unordered_map>

#include "host_parallel.hpp"
#include "internal.hpp"
#include "search_device.hpp"  // the acceptance test of the direction table, which also compiles for the host (msm_ray_table_check)

// set-up timing to stderr when MSMHIP_TIMING is set
#define TICK(name) do { if (std::getenv("MSMHIP_TIMING")) { auto now_ = std::chrono::steady_clock::now(); fprintf(stderr, "  %s %.1f ms\n", name, std::chrono::duration<double, std::milli>(now_ - tick_).count()); tick_ = now_; } } while (0)

namespace msm {

namespace {

struct BNode {
    int first_child = -1;  // index of child (0,0,0); children are consecutive in (i,j,k) order
    int parent = -1;
    double b[3][3];  // per axis: lower, middle, upper (Node::bounds, R/node.h:42)
    std::vector<int32_t> tris;
    // running sums of the split heuristic over `tris` (the reference recomputes them over the whole leaf at every
    // insertion past MAX_TRIANGLES, R/octree.cpp:71-93; the node's midpoints never change, so they can be kept)
    int total_size = 0, num_split = 0;
};

struct Builder {
    const double *xyz;
    const int32_t *tri;
    int V, T;
    std::deque<BNode> nodes;
    std::vector<double> box;  // per triangle: lo[3], hi[3]
    const double *bxp = nullptr;  // = box.data(), or another builder's boxes (worker threads share them)

    void precompute_boxes() {
        box.resize((size_t)6 * T);
        bxp = box.data();
        parallel_chunks(T, T < 20000 ? 1 : std::min(host_workers(), 8), [&](int, int t0, int t1) {
            for (int t = t0; t < t1; ++t) {
                double *lo = &box[(size_t)6 * t], *hi = lo + 3;
                for (int a = 0; a < 3; ++a) lo[a] = hi[a] = xyz[a * V + tri[t]];
                for (int k = 1; k < 3; ++k)
                    for (int a = 0; a < 3; ++a) {
                        double c = xyz[a * V + tri[k * T + t]];
                        if (c < lo[a]) lo[a] = c;
                        if (c > hi[a]) hi[a] = c;
                    }
            }
        });
    }
    void aabb(int t, double lo[3], double hi[3]) const {
        const double *b = &bxp[(size_t)6 * t];
        for (int a = 0; a < 3; ++a) {
            lo[a] = b[a];
            hi[a] = b[3 + a];
        }
    }
    // Node::can_contain, R/node.cpp:108-116
    bool overlaps(int n, const double lo[3], const double hi[3]) const {
        const BNode &nd = nodes[n];
        for (int a = 0; a < 3; ++a)
            if (hi[a] < nd.b[a][0] || lo[a] > nd.b[a][2]) return false;
        return true;
    }
    // Node::make_children, R/node.cpp:84-106
    void split(int n) {
        int base = (int)nodes.size();
        for (int c = 0; c < 8; ++c) {
            BNode ch;
            const int oct[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
            for (int a = 0; a < 3; ++a) {
                ch.b[a][0] = nodes[n].b[a][oct[a]];
                ch.b[a][2] = nodes[n].b[a][oct[a] + 1];
                ch.b[a][1] = (ch.b[a][0] + ch.b[a][2]) / 2.0;
            }
            ch.parent = n;
            ch.tris.reserve(kMaxTriangles);
            nodes.push_back(std::move(ch));
        }
        nodes[n].first_child = base;
    }
    // Octree::add_triangle, R/octree.cpp:65-141
    void add(int n, int t, const double lo[3], const double hi[3]) {
        if (nodes[n].first_child < 0) {
            nodes[n].tris.push_back(t);
            {
                int split_size = 8;
                for (int d = 0; d < 3; ++d)  // containing_oct: coordinate < middle, R/node.cpp:70-82
                    if ((lo[d] < nodes[n].b[d][1]) == (hi[d] < nodes[n].b[d][1])) split_size >>= 1;
                nodes[n].total_size += split_size;
                if (split_size != 8) ++nodes[n].num_split;
            }
            const int num = (int)nodes[n].tris.size();
            if (num < kMaxTriangles) return;
            if (!(nodes[n].num_split > 0 && nodes[n].total_size < 3 * num)) return;
            split(n);
            std::vector<int32_t> held;
            held.swap(nodes[n].tris);  // clear_triangles() after redistribution
            for (int tt : held) {
                double tlo[3], thi[3];
                aabb(tt, tlo, thi);
                for (int c = 0; c < 8; ++c)
                    if (overlaps(nodes[n].first_child + c, tlo, thi)) add(nodes[n].first_child + c, tt, tlo, thi);
            }
        } else {
            for (int c = 0; c < 8; ++c)
                if (overlaps(nodes[n].first_child + c, lo, hi)) add(nodes[n].first_child + c, t, lo, hi);
        }
    }

    // ---- the same tree, top down ------------------------------------------------------------------------------
    // Inserting the triangles one by one is equivalent to this: a node sees, in ascending id, every triangle whose box
    // overlaps it (when a leaf splits, the triangles it holds are handed down in order and all later ones follow in
    // order; a child never sees anything else, nor in another order).  It stays a leaf unless the split heuristic
    // holds at some insertion from the MAX_TRIANGLES-th on -- the first such insertion splits it, and from then on only
    // its children matter.  So a node is decided by one scan of its list, and its children by the sub-lists of the
    // WHOLE list that overlap them: no per-triangle descent, and independent subtrees can be built on separate threads.
    // `nodes` must hold node n with its box; children are appended to `nodes`.
    void build_down(int n, std::vector<int32_t> &list) {
        int total_size = 0, num_split = 0;
        bool splits = false;
        const int len = (int)list.size();
        for (int i = 0; i < len; ++i) {
            const double *bx = &bxp[(size_t)6 * list[i]];
            int split_size = 8;
            for (int d = 0; d < 3; ++d)
                if ((bx[d] < nodes[n].b[d][1]) == (bx[3 + d] < nodes[n].b[d][1])) split_size >>= 1;
            total_size += split_size;
            if (split_size != 8) ++num_split;
            if (i + 1 >= kMaxTriangles && num_split > 0 && total_size < 3 * (i + 1)) {
                splits = true;
                break;
            }
        }
        if (!splits) {
            nodes[n].tris = std::move(list);
            return;
        }
        split(n);
        const int first = nodes[n].first_child;
        std::vector<int32_t> sub[8];
        distribute(n, list, sub);
        std::vector<int32_t>().swap(list);  // the parent's list is not needed below
        for (int c = 0; c < 8; ++c) build_down(first + c, sub[c]);
    }

    // The children's lists of a node that has just been split: child c receives, in order, every triangle of `list` whose box
    // overlaps the child's box (Node::can_contain on the child).  The eight tests of a triangle share their comparisons: per
    // axis the child box is the parent's [lower, middle] or [middle, upper], so twelve comparisons decide all eight.
    void distribute(int n, const std::vector<int32_t> &list, std::vector<int32_t> sub[8]) const {
        const BNode &nd = nodes[n];
        const size_t guess = list.size() / 5 + 8;
        for (int c = 0; c < 8; ++c) sub[c].reserve(guess);
        for (const int32_t t : list) {
            const double *bx = &bxp[(size_t)6 * t];
            unsigned half[3];  // bit 0: overlaps the lower half along the axis, bit 1: the upper half
            for (int a = 0; a < 3; ++a)
                half[a] = (!(bx[3 + a] < nd.b[a][0] || bx[a] > nd.b[a][1]) ? 1u : 0u) | (!(bx[3 + a] < nd.b[a][1] || bx[a] > nd.b[a][2]) ? 2u : 0u);
            for (int c = 0; c < 8; ++c)
                if ((half[0] >> ((c >> 2) & 1) & 1u) && (half[1] >> ((c >> 1) & 1) & 1u) && (half[2] >> (c & 1) & 1u)) sub[c].push_back(t);
        }
    }
};

inline V3 vtx(const double *xyz, int V, int i) { return mk(xyz[i], xyz[V + i], xyz[2 * V + i]); }

// True when every ray from the origin meets exactly one triangle: the mesh is a closed, consistently oriented
// 2-manifold, every face is strictly front- (or every face strictly back-) facing seen from the origin, and the faces'
// solid angles add up to one full sphere (degree of the radial projection = 1).
bool simple_star_surface(const double *xyz, const int32_t *tri, int V, int T) {
    if (T < 4) return false;
    // faces: orientation seen from the origin and solid angle, on worker threads (chunk sums added in chunk order)
    const int workers = T < 20000 ? 1 : host_workers();
    std::vector<double> part_solid(4 * (size_t)std::max(workers, 1), 0.0);
    std::vector<int> part_sign(part_solid.size(), 0);  // +1 / -1: all faces of the chunk face that way; 2: the mesh fails
    parallel_chunks(T, workers, [&](int c, int t0, int t1) {
        double solid = 0.0;
        int sign = 0;
        for (int t = t0; t < t1; ++t) {
            const int id[3] = {tri[t], tri[T + t], tri[2 * T + t]};
            bool bad = id[0] == id[1] || id[1] == id[2] || id[0] == id[2];
            for (int k = 0; k < 3; ++k) bad = bad || id[k] < 0 || id[k] >= V;
            if (bad) {
                sign = 2;
                break;
            }
            const V3 a = vtx(xyz, V, id[0]), b = vtx(xyz, V, id[1]), cc = vtx(xyz, V, id[2]);
            const double la = norm(a), lb = norm(b), lc = norm(cc);
            const double triple = dot(a, cross(b, cc));
            if (!(la > 0 && lb > 0 && lc > 0) || !std::isfinite(triple)) {
                sign = 2;
                break;
            }
            const int s = triple > 1e-12 * la * lb * lc ? 1 : (triple < -1e-12 * la * lb * lc ? -1 : 0);
            if (s == 0 || (sign != 0 && s != sign)) {  // a face seen edge-on or from behind
                sign = 2;
                break;
            }
            sign = s;
            // Van Oosterom-Strackee solid angle of the face
            const double den = la * lb * lc + dot(a, b) * lc + dot(a, cc) * lb + dot(b, cc) * la;
            solid += 2.0 * std::atan2(std::fabs(triple), den);
        }
        part_solid[c] = solid;
        part_sign[c] = sign;
    });
    double solid = 0.0;
    int sign = 0;
    for (size_t c = 0; c < part_solid.size(); ++c) {
        if (part_sign[c] == 0) continue;
        if (part_sign[c] == 2 || (sign != 0 && part_sign[c] != sign)) return false;
        sign = part_sign[c];
        solid += part_solid[c];
    }
    // directed edges u -> v bucketed by u: each must be unique (consistent orientation, manifold) and have its opposite
    // (closed surface).  Buckets hold a vertex's valence (six on an icosphere), so the checks are short scans.
    std::vector<int32_t> ptr((size_t)V + 1, 0);
    for (int k = 0; k < 3; ++k)
        for (int t = 0; t < T; ++t) ++ptr[(size_t)tri[(size_t)k * T + t] + 1];
    for (int v = 0; v < V; ++v) ptr[v + 1] += ptr[v];
    std::vector<int32_t> head((size_t)3 * T), fill(ptr.begin(), ptr.end() - 1);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) head[fill[tri[(size_t)k * T + t]]++] = tri[(size_t)((k + 1) % 3) * T + t];
    std::atomic<bool> ok{true};
    parallel_chunks(V, workers, [&](int, int v0, int v1) {
        for (int u = v0; u < v1 && ok.load(std::memory_order_relaxed); ++u)
            for (int e = ptr[u]; e < ptr[u + 1]; ++e) {
                const int v = head[e];
                for (int f = ptr[u]; f < e; ++f)
                    if (head[f] == v) ok.store(false, std::memory_order_relaxed);  // the same directed edge twice
                bool opposite = false;
                for (int f = ptr[v]; f < ptr[v + 1]; ++f) opposite = opposite || head[f] == u;
                if (!opposite) ok.store(false, std::memory_order_relaxed);
            }
    });
    if (!ok.load()) return false;
    const long E = (long)3 * T / 2;
    if ((long)V - E + (long)T != 2) return false;  // sphere topology (all vertices referenced is implied when this holds for a closed manifold)
    return std::fabs(solid - 4.0 * M_PI) < 1e-6;
}

// Per triangle: the in-plane distance from every edge of the triangle beyond which no OTHER triangle of a simple
// surface can pass the reference's inside test.  A same_side product equals 2*area*|edge|*(in-plane distance to that
// edge), so a triangle accepts points up to band = 1e-8 / (2*area*shortest edge) outside its edges; the margin is
// max(1e-5 * longest edge, 1000 * the widest band among all triangles sharing a vertex with this one).
std::vector<double> safe_margins(const double *xyz, const int32_t *tri, int V, int T) {
    std::vector<double> band(T), lmax(T), vband(V, 0.0), margin(T);
    for (int t = 0; t < T; ++t) {
        const int id[3] = {tri[t], tri[T + t], tri[2 * T + t]};
        const V3 a = vtx(xyz, V, id[0]), b = vtx(xyz, V, id[1]), c = vtx(xyz, V, id[2]);
        const double la = norm(sub(b, c)), lb = norm(sub(a, c)), lc = norm(sub(a, b));
        const double area = 0.5 * norm(cross(sub(b, a), sub(c, a)));
        lmax[t] = std::fmax(la, std::fmax(lb, lc));
        const double lmin = std::fmin(la, std::fmin(lb, lc));
        band[t] = (area > 0 && lmin > 0) ? 1e-8 / (2 * area * lmin) : HUGE_VAL;
        if (!(band[t] == band[t])) band[t] = HUGE_VAL;
        for (int k = 0; k < 3; ++k) vband[id[k]] = std::fmax(vband[id[k]], band[t]);
    }
    for (int t = 0; t < T; ++t) {
        const double widest = std::fmax(vband[tri[t]], std::fmax(vband[tri[T + t]], vband[tri[2 * T + t]]));
        const double dist = std::fmax(1e-5 * lmax[t], 1e3 * widest);
        margin[t] = (std::isfinite(dist) && dist < 0.05 * lmax[t]) ? dist : HUGE_VAL;
    }
    return margin;
}

// The leaves whose closed box meets [lo, hi] but which do not list triangle t (at most `cap` are collected; returns
// false when there are more)
bool lacking_leaves(const FlatOctree &o, int n, const double lo[3], const double hi[3], int t, int cap, std::vector<int> &lack) {
    const double4 box = o.nodebox[n];
    const double blo[3] = {box.x, box.y, box.z};
    for (int a = 0; a < 3; ++a)
        if (hi[a] < blo[a] || lo[a] > blo[a] + box.w) return true;
    const int4 nd = o.node[n];
    if (nd.x < 0) {
        const int32_t *first = o.leaf_tri.data() + nd.y, *last = first + (-nd.x - 1);
        if (std::find(first, last, t) != last) return true;
        if ((int)lack.size() >= cap) return false;
        lack.push_back(n);
        return true;
    }
    for (int c = 0; c < 8; ++c)
        if (!lacking_leaves(o, nd.x + c, lo, hi, t, cap, lack)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Ray table (simple surfaces).  On such a surface exactly one triangle t meets the ray through a query p, and
// Octree::get_closest_triangle (R/octree.cpp:156-214) returns t provided that
//   (1) t is listed in the leaf p descends to, and
//   (2) no other listed triangle passes the -1e-8 inside test,
// because a single passing triangle wins whatever its dist_to_point.  Both are settled per triangle here:
//   edge planes  the inward unit normals n_k of the planes through the origin and edge k.  The in-plane distance of
//                the projected point from edge k is at least |projection| * (p^ . n_k) >= rho * (p^ . n_k), rho = distance
//                of the triangle's plane from the origin, so  p^ . n_k >= margin / rho  for k = 0,1,2  implies (2);
//                the stored threshold adds kRayFloatAllowance = 3e-6 for the float evaluation (direction 2e-7, normals 6e-8, dot 2e-7);
//                a kernel may settle what lies inside that allowance with FP64 edge planes (search_device.hpp: ray_accepts_fp64);
//   robustness   every point of the shell kRayShell around radius 100 whose direction lies in t's spherical triangle
//                is within `sag` of the flat triangle scaled to that radius; if every leaf meeting the AABB of that
//                region lists t, (1) holds for every such p.  Up to three leaves that do not list t are recorded as
//                exclusion boxes (ray_excl): (1) then holds for every p outside those leaves, and p's leaf at depth d
//                is found arithmetically (child boxes are exact halvings).  More than kRayExclMax = 7 (two records): a threshold
//                nothing passes.  (With three, 72 of the 81 920 triangles of an ico6 sphere were unusable -- the ones whose box ends
//                on a leaf boundary with 2 x 2 leaves beyond it -- and every sample in them went to the complete search.)
// Record layout (three float4 per triangle): {n0, thr} {n1, bits of the ray_excl index or -1} {n2, 0}.
// The cube-map cells only propose candidates, likeliest first: {c0,c1,c2,c3}, or {c0,c1,c2,-2-k} with c3..c6 in
// ray_more[k]; a miss just means the complete search.
// ------------------------------------------------------------------------------------------------
constexpr double kRayShell = 1e-4;
constexpr int kRayExclMax = 7;

}  // namespace

void build_ray_table(const double *xyz, const int32_t *tri, int V, int T, FlatOctree &out) {
    auto tick_ = std::chrono::steady_clock::now();
    out.ray_G = 0;
    out.ray_cell.clear();
    out.ray_edge.clear();
    out.simple = simple_star_surface(xyz, tri, V, T);
    const char *no_table = std::getenv("MSMHIP_DISABLE_RAYTABLE");  // testing: force the complete search everywhere
    if (!out.simple || (no_table && no_table[0] == '1')) return;
    TICK("simple");
    const std::vector<double> margin = safe_margins(xyz, tri, V, T);
    TICK("margins");
    out.ray_edge.assign((size_t)3 * T, make_float4(0.f, 0.f, 0.f, 2.f));
    out.ray_excl.clear();
    out.ray_more.clear();
    out.ray_r2lo = (kRad - kRayShell) * (kRad - kRayShell);
    out.ray_r2hi = (kRad + kRayShell) * (kRad + kRayShell);
    std::vector<char> usable(T, 0);
    const int workers = host_workers();
    // exclusion records are collected per chunk and numbered afterwards (chunks are contiguous triangle ranges, so the
    // numbering does not depend on the number of threads)
    struct Chunk {
        std::vector<int4> excl;
        std::vector<int> owner;
    };
    std::vector<Chunk> chunk_out(4 * (size_t)workers + 1);
    parallel_chunks(T, workers, [&](int ci, int t_begin, int t_end) {
    std::vector<int> lack;
    Chunk &mine = chunk_out[ci];
    for (int t = t_begin; t < t_end; ++t) {
        const V3 v[3] = {vtx(xyz, V, tri[t]), vtx(xyz, V, tri[T + t]), vtx(xyz, V, tri[2 * T + t])};
        V3 s3;
        double pd;
        plane_of(v[0], v[1], v[2], s3, pd);
        const double rho = std::fabs(pd);  // distance of the triangle's plane from the origin
        if (!(margin[t] < HUGE_VAL) || !(rho > 0)) continue;
        const double thr = margin[t] / rho * (1 + 1e-6) + kRayFloatAllowance;
        if (!(thr < 0.1)) continue;
        // (1): the shell region above the triangle
        V3 u[3];
        double cosmax = 1.0;
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            const double n = norm(v[k]);
            ok = ok && n > 0;
            u[k] = scale(v[k], 1.0 / n);
        }
        if (!ok) continue;
        for (int k = 0; k < 3; ++k) cosmax = std::fmin(cosmax, dot(u[k], u[(k + 1) % 3]));
        if (!(cosmax > 0.5)) continue;
        const double rlo = kRad - kRayShell, rhi = kRad + kRayShell;
        const double sag = rhi * (1 - std::sqrt(cosmax)) * (1 + 1e-9) + 1e-9;
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (int k = 0; k < 3; ++k) {
            const double c[3] = {u[k].x, u[k].y, u[k].z};
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::fmin(lo[a], std::fmin(rlo * c[a], rhi * c[a]));
                hi[a] = std::fmax(hi[a], std::fmax(rlo * c[a], rhi * c[a]));
            }
        }
        for (int a = 0; a < 3; ++a) {
            lo[a] -= sag;
            hi[a] += sag;
        }
        lack.clear();
        if (!lacking_leaves(out, 0, lo, hi, t, kRayExclMax, lack)) continue;
        // (2): edge planes; edge k is opposite vertex k (same_side(p, v_k, v_k+1, v_k+2), R/point.cpp:41-44)
        float4 e[3];
        for (int k = 0; k < 3 && ok; ++k) {
            V3 n = cross(v[(k + 1) % 3], v[(k + 2) % 3]);
            const double nn = norm(n);
            if (!(nn > 0) || !std::isfinite(nn)) {
                ok = false;
                break;
            }
            n = scale(n, 1.0 / nn);
            if (dot(n, v[k]) < 0) n = scale(n, -1.0);
            e[k] = make_float4((float)n.x, (float)n.y, (float)n.z, (float)thr);
            if (!((double)e[k].w >= thr)) e[k].w = std::nextafterf(e[k].w, 1.f);  // round the threshold up
        }
        if (!ok) continue;
        int excl = -1;
        if (!lack.empty()) {  // leaf boxes as depth << 24 | ix << 16 | iy << 8 | iz (depth <= 8: coordinates fit 8 bits)
            int32_t bx[kRayExclMax];
            for (int j = 0; j < kRayExclMax; ++j) bx[j] = -1;
            for (size_t j = 0; j < lack.size() && ok; ++j) {
                const double4 nb = out.nodebox[lack[j]];
                const int d = out.node[lack[j]].w;
                if (d > 8) {
                    ok = false;
                    break;
                }
                const double size = 2 * kBounds / (double)(1 << d);
                const int ix = (int)std::lround((nb.x + kBounds) / size), iy = (int)std::lround((nb.y + kBounds) / size), iz = (int)std::lround((nb.z + kBounds) / size);
                bx[j] = (d << 24) | (ix << 16) | (iy << 8) | iz;
            }
            if (!ok) continue;
            excl = (int)mine.excl.size();  // chunk-local; renumbered below
            mine.excl.push_back(make_int4(bx[0], bx[1], bx[2], (int)lack.size()));
            mine.owner.push_back(t);
            if (lack.size() > 3) {  // boxes 3..6 in the record that follows
                mine.excl.push_back(make_int4(bx[3], bx[4], bx[5], bx[6]));
                mine.owner.push_back(-1);
            }
        }
        e[1].w = __builtin_bit_cast(float, (int32_t)excl);
        e[2].w = 0.f;
        for (int k = 0; k < 3; ++k) out.ray_edge[(size_t)3 * t + k] = e[k];
        usable[t] = 1;
    }
    });
    for (const Chunk &c : chunk_out) {
        const int base = (int)out.ray_excl.size();
        for (size_t k = 0; k < c.excl.size(); ++k) {
            out.ray_excl.push_back(c.excl[k]);
            if (c.owner[k] >= 0) out.ray_edge[(size_t)3 * c.owner[k] + 1].w = __builtin_bit_cast(float, (int32_t)(base + (int)k));
        }
    }
    TICK("edges+robust");
    // cube map: face f = 2*axis + (negative side); (u, v) = the two other components over |major component|
    int G = 8;
    while (G < 512 && (double)G * G * 6 < 4.0 * T) G *= 2;
    if (const char *e = std::getenv("MSMHIP_RAY_G")) G = std::max(8, std::min(1024, atoi(e)));  // experiments only
    out.ray_G = G;
    const size_t ncell = (size_t)6 * G * G;
    struct Cand {
        uint32_t cell;
        float score;
        int32_t tri;
    };
    std::vector<std::vector<Cand>> found_by(4 * (size_t)workers + 1);
    std::vector<int32_t> count(ncell + 1, 0);
    const double m = 1e-5;  // cells are grown by this much (uv units) before the overlap test
    parallel_chunks(T, workers, [&](int ci, int t_begin, int t_end) {
    std::vector<Cand> &found = found_by[ci];
    found.reserve((size_t)(t_end - t_begin) * 8);
    for (int t = t_begin; t < t_end; ++t) {
        if (!usable[t]) continue;
        const V3 v[3] = {vtx(xyz, V, tri[t]), vtx(xyz, V, tri[T + t]), vtx(xyz, V, tri[2 * T + t])};
        const double ln[3] = {norm(v[0]), norm(v[1]), norm(v[2])};
        for (int f = 0; f < 6; ++f) {
            const int a = f >> 1, bb = (a + 1) % 3, cc = (a + 2) % 3;
            const double sgn = (f & 1) ? -1.0 : 1.0;
            double pu[3], pv[3];
            bool front = true;
            for (int k = 0; k < 3; ++k) {
                const double c[3] = {v[k].x, v[k].y, v[k].z};
                const double w = sgn * c[a];
                if (!(w > 0.05 * ln[k])) {
                    front = false;
                    break;
                }
                pu[k] = c[bb] / w;
                pv[k] = c[cc] / w;
            }
            if (!front) continue;
            const double umin = std::fmin(pu[0], std::fmin(pu[1], pu[2])) - m, umax = std::fmax(pu[0], std::fmax(pu[1], pu[2])) + m;
            const double vmin = std::fmin(pv[0], std::fmin(pv[1], pv[2])) - m, vmax = std::fmax(pv[0], std::fmax(pv[1], pv[2])) + m;
            if (umax < -1 || umin > 1 || vmax < -1 || vmin > 1) continue;
            auto cellof = [&](double x) { return std::max(0, std::min(G - 1, (int)std::floor((x + 1) * 0.5 * G))); };
            const int iu0 = cellof(umin), iu1 = cellof(umax), iv0 = cellof(vmin), iv1 = cellof(vmax);
            const double gu = (pu[0] + pu[1] + pu[2]) / 3, gv = (pv[0] + pv[1] + pv[2]) / 3;
            // edge functions, oriented so that the third vertex is on the positive side
            double ex[3], ey[3], ox[3], oy[3], sg[3];
            for (int k = 0; k < 3; ++k) {
                const int p = (k + 1) % 3, q = (k + 2) % 3;
                ex[k] = pu[q] - pu[p], ey[k] = pv[q] - pv[p], ox[k] = pu[p], oy[k] = pv[p];
                sg[k] = (ex[k] * (pv[k] - oy[k]) - ey[k] * (pu[k] - ox[k])) >= 0 ? 1.0 : -1.0;
            }
            for (int iu = iu0; iu <= iu1; ++iu)
                for (int iv = iv0; iv <= iv1; ++iv) {
                    const double x0 = -1 + 2.0 * iu / G - m, x1 = -1 + 2.0 * (iu + 1) / G + m;
                    const double y0 = -1 + 2.0 * iv / G - m, y1 = -1 + 2.0 * (iv + 1) / G + m;
                    bool sep = false;  // a triangle edge with the whole (grown) cell strictly outside
                    for (int k = 0; k < 3 && !sep; ++k) {
                        auto E = [&](double x, double y) { return sg[k] * (ex[k] * (y - oy[k]) - ey[k] * (x - ox[k])); };
                        sep = E(x0, y0) < 0 && E(x1, y0) < 0 && E(x0, y1) < 0 && E(x1, y1) < 0;
                    }
                    if (sep) continue;
                    const double cu = 0.5 * (x0 + x1) - gu, cv = 0.5 * (y0 + y1) - gv;
                    const size_t cell = ((size_t)f * G + iu) * G + iv;
                    found.push_back(Cand{(uint32_t)cell, (float)(cu * cu + cv * cv), t});
                }
        }
    }
    });
    size_t nfound = 0;
    for (const auto &f : found_by) {
        nfound += f.size();
        for (const Cand &c : f) ++count[c.cell + 1];
    }
    TICK("raster");
    for (size_t c = 0; c < ncell; ++c) count[c + 1] += count[c];
    std::vector<Cand> cand(nfound);
    {
        std::vector<int32_t> fill(count.begin(), count.end() - 1);
        for (const auto &f : found_by)  // chunk order = triangle order: the same buckets as a serial pass
            for (const Cand &c : f) cand[(size_t)fill[c.cell]++] = c;
    }
    TICK("bucket");
    out.ray_cell.assign(ncell, make_int4(-1, -1, -1, -1));
    parallel_chunks((int)ncell, workers, [&](int, int c_begin, int c_end) {
        for (int c = c_begin; c < c_end; ++c)
            std::sort(cand.data() + count[c], cand.data() + count[c + 1],
                      [](const Cand &x, const Cand &y) { return x.score < y.score || (x.score == y.score && x.tri < y.tri); });
    });
    for (size_t c = 0; c < ncell; ++c) {
        const Cand *first = cand.data() + count[c], *last = cand.data() + count[c + 1];
        int32_t *slot = &out.ray_cell[c].x;
        const int ncand = (int)(last - first);
        if (ncand <= 4) {
            for (int k = 0; k < ncand; ++k) slot[k] = first[k].tri;
        } else {
            for (int k = 0; k < 3; ++k) slot[k] = first[k].tri;
            slot[3] = -2 - (int)out.ray_more.size();
            int4 more = make_int4(-1, -1, -1, -1);
            int32_t *ms = &more.x;
            for (int k = 0; k < 4 && 3 + k < ncand; ++k) ms[k] = first[3 + k].tri;
            out.ray_more.push_back(more);
        }
    }
    TICK("cells");
}

namespace {

// build_down over the whole mesh; the subtrees below the first two levels are built on worker threads, each in its own
// node store, and spliced into b.nodes afterwards (only the order of the nodes differs from the serial build).
void build_top_down(Builder &b) {
    auto tick_ = std::chrono::steady_clock::now();
    std::vector<int32_t> all(b.T);
    for (int t = 0; t < b.T; ++t) all[t] = t;
    const int workers = host_workers();
    if (workers == 1 || b.T < 20000) {  // threads cost more than they save on small meshes
        b.build_down(0, all);
        return;
    }
    struct Job {
        int node;
        std::vector<int32_t> list;
    };
    // levels 0 and 1: decide and split here, collect the (node, list) pairs of the next level instead of recursing
    std::vector<Job> frontier, next;
    frontier.push_back(Job{0, std::move(all)});
    for (int level = 0; level < 2; ++level) {
        next.clear();
        // decide every node of the level (a scan that stops at the split), create the children of those that split ...
        std::vector<std::pair<int, int>> todo;  // (job, slot in next) of the children to fill
        for (size_t jn = 0; jn < frontier.size(); ++jn) {
            Job &j = frontier[jn];
            int total_size = 0, num_split = 0;
            bool splits = false;
            const int len = (int)j.list.size();
            for (int i = 0; i < len && !splits; ++i) {
                const double *bx = &b.bxp[(size_t)6 * j.list[i]];
                int split_size = 8;
                for (int d = 0; d < 3; ++d)
                    if ((bx[d] < b.nodes[j.node].b[d][1]) == (bx[3 + d] < b.nodes[j.node].b[d][1])) split_size >>= 1;
                total_size += split_size;
                if (split_size != 8) ++num_split;
                splits = i + 1 >= kMaxTriangles && num_split > 0 && total_size < 3 * (i + 1);
            }
            if (!splits) {
                b.nodes[j.node].tris = std::move(j.list);
                continue;
            }
            b.split(j.node);
            const int first = b.nodes[j.node].first_child;
            for (int c = 0; c < 8; ++c) {
                todo.push_back({(int)jn, (int)next.size()});
                next.push_back(Job{first + c, {}});
            }
        }
        // ... and fill the children's lists (one pass over the parent's list per split node) on the worker threads
        parallel_for((int)todo.size() / 8, workers, [&](int k) {
            const Job &j = frontier[todo[8 * k].first];
            std::vector<int32_t> sub[8];
            b.distribute(j.node, j.list, sub);
            for (int c = 0; c < 8; ++c) next[todo[8 * k + c].second].list = std::move(sub[c]);
        });
        frontier.swap(next);
    }
    TICK("octree: top levels");
    // every remaining job builds its subtree in a private node store whose node 0 stands for the job's node
    std::vector<Builder> part;
    part.reserve(frontier.size());
    for (const Job &j : frontier) {
        part.push_back(Builder{b.xyz, b.tri, b.V, b.T, {}, {}, b.bxp});
        part.back().nodes.push_back(b.nodes[j.node]);
    }
    parallel_for((int)frontier.size(), workers, [&](int k) { part[k].build_down(0, frontier[k].list); });
    TICK("octree: subtrees");
    for (size_t k = 0; k < part.size(); ++k) {
        const int root = frontier[k].node, base = (int)b.nodes.size() - 1;  // local i > 0 -> base + i
        auto global = [&](int i) { return i == 0 ? root : base + i; };
        std::deque<BNode> &ln = part[k].nodes;
        b.nodes[root].tris = std::move(ln[0].tris);
        b.nodes[root].first_child = ln[0].first_child < 0 ? -1 : global(ln[0].first_child);
        for (size_t i = 1; i < ln.size(); ++i) {
            BNode nd = std::move(ln[i]);
            nd.parent = global(nd.parent);
            if (nd.first_child >= 0) nd.first_child = global(nd.first_child);
            b.nodes.push_back(std::move(nd));
        }
    }
}

}  // namespace

void build_octree(const double *xyz, const int32_t *tri, int V, int T, FlatOctree &out) {
    auto tick_ = std::chrono::steady_clock::now();
    Builder b{xyz, tri, V, T, {}, {}, nullptr};
    b.precompute_boxes();
    TICK("octree: boxes");
    BNode root;
    for (int a = 0; a < 3; ++a) {
        root.b[a][0] = -kBounds;
        root.b[a][2] = kBounds;
        root.b[a][1] = (root.b[a][0] + root.b[a][2]) / 2.0;
    }
    root.tris.reserve(kMaxTriangles);
    b.nodes.push_back(std::move(root));
    const char *incremental = std::getenv("MSMHIP_INCREMENTAL_OCTREE");  // the literal restatement, kept for cross-checks
    if (incremental && incremental[0] == '1') {
        for (int t = 0; t < T; ++t) {  // initialize_tree, R/octree.cpp:42-63
            double lo[3], hi[3];
            b.aabb(t, lo, hi);
            b.add(0, t, lo, hi);
        }
    } else {
        build_top_down(b);
    }

    TICK("octree: insertion");
    out.simple = false;  // decided together with the ray table (build_ray_table), which only the cost kernels need
    out.ray_G = 0;
    out.ray_cell.clear();
    out.ray_edge.clear();

    // Leaf entries are padded to multiples of 8 (id -1, a cone nothing passes): 8 cones = one 128-byte line.
    const int n = (int)b.nodes.size();
    out.node.resize(n);
    out.nodebox.resize(n);
    out.parent.resize(n);
    out.leaf_tri.clear();
    out.nmask_blocks = 0;
    int64_t leaves = 0, maxleaf = 0, refs = 0;
    std::vector<int> depth(n, 0);
    int maxdepth = 0;
    for (int i = 0; i < n; ++i) {
        const BNode &nd = b.nodes[i];
        out.parent[i] = nd.parent;
        if (i > 0) depth[i] = depth[nd.parent] + 1;  // children always follow their parent in the node array
        maxdepth = std::max(maxdepth, depth[i]);
        out.nodebox[i] = make_double4(nd.b[0][0], nd.b[1][0], nd.b[2][0], nd.b[0][2] - nd.b[0][0]);
        if (nd.first_child >= 0) {
            out.node[i] = make_int4(nd.first_child, 0, -1, depth[i]);
        } else {
            const int cnt = (int)nd.tris.size();
            const int mblock = (cnt >= 1 && cnt <= 64) ? out.nmask_blocks++ : -1;
            out.node[i] = make_int4(-cnt - 1, (int)out.leaf_tri.size(), mblock, depth[i]);
            for (int e = 0; e < ((cnt + 7) & ~7); ++e) {
                out.leaf_tri.push_back(e < cnt ? nd.tris[e] : -1);
            }
            ++leaves;
            refs += cnt;
            maxleaf = std::max<int64_t>(maxleaf, (int64_t)cnt);
        }
    }
    // dense top grid: every node at depth <= grid_depth that is a leaf, or sits at grid_depth, owns a cube of cells
    out.grid_depth = std::min(maxdepth, 6);
    const int G = 1 << out.grid_depth;
    out.grid.assign((size_t)G * G * G, 0);
    {
        struct Item { int node, depth, ix, iy, iz; };
        std::vector<Item> stack{{0, 0, 0, 0, 0}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            const BNode &nd = b.nodes[it.node];
            if (nd.first_child < 0 || it.depth == out.grid_depth) {
                const int span = 1 << (out.grid_depth - it.depth);
                for (int x = 0; x < span; ++x)
                    for (int y = 0; y < span; ++y)
                        for (int z = 0; z < span; ++z)
                            out.grid[((size_t)(it.ix * span + x) * G + (it.iy * span + y)) * G + (it.iz * span + z)] = it.node;
            } else {
                for (int c = 0; c < 8; ++c)
                    stack.push_back({nd.first_child + c, it.depth + 1, 2 * it.ix + ((c >> 2) & 1), 2 * it.iy + ((c >> 1) & 1), 2 * it.iz + (c & 1)});
            }
        }
    }
    TICK("octree: flatten + grid");
    out.stats[0] = n;
    out.stats[1] = leaves;
    out.stats[2] = maxdepth;
    out.stats[3] = refs;
    out.stats[4] = maxleaf;
}

}  // namespace msm

// [host] testing hook: the search tree of a mesh without a GPU -- its statistics and a signature of its leaves (box and
// triangle list of every leaf, independent of the order in which the nodes were created)
extern "C" int msm_octree_signature(const double *xyz, const int32_t *tri, int32_t V, int32_t T, int64_t stats[5], uint64_t *signature) {
    if (!xyz || !tri || V <= 0 || T <= 0) return msm::fail(MSM_ERR_INVALID, "msm_octree_signature: bad arguments");
    msm::FlatOctree o;
    msm::build_octree(xyz, tri, V, T, o);
    if (stats) std::copy(o.stats, o.stats + 5, stats);
    if (signature) {
        uint64_t sum = 0;
        for (size_t n = 0; n < o.node.size(); ++n) {
            if (o.node[n].x >= 0) continue;
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) {
                for (int k = 0; k < 8; ++k) {
                    h ^= (v >> (8 * k)) & 0xff;
                    h *= 1099511628211ull;
                }
            };
            const double4 b = o.nodebox[n];
            for (double d : {b.x, b.y, b.z, b.w}) mix((uint64_t)__builtin_bit_cast(uint64_t, d));
            for (int e = 0; e < -o.node[n].x - 1; ++e) mix((uint64_t)o.leaf_tri[o.node[n].y + e]);
            sum += h;
        }
        *signature = sum;
    }
    return MSM_OK;
}

// Testing hook, host only: the guarantee of the direction table (build_ray_table), checked point by point against the reference's own search.
// For each of `nsamples` points on the radius shell -- random directions, and points placed within 1e-10 .. 1e-3 (relative to the edge) of
// random edges and vertices, on either side -- every candidate of the point's cell that a kernel may accept (the float test, or the FP64
// re-test of a nearly accepted candidate, plus ray_vouches: the very functions the kernels call, search_device.hpp) must be THE answer of
// Octree::get_closest_triangle's first pass (R/octree.cpp:156-178): listed in the leaf the point descends to, and the only listed triangle
// that passes the -1e-8 inside test.  report: [0] points, [1] accepted by the float test, [2] accepted only by the FP64 re-test, [3] left to
// the complete search, [4] VIOLATIONS (must be 0), [5] triangles the table cannot use, [6] triangles with exclusion boxes, [7] simple surface,
// [8] exclusion boxes checked one by one (a point at the centre of the leaf a box stands for must be refused for that triangle, a violation otherwise),
// [9] sampled points whose triangle passed the edge-plane test and was refused by ray_vouches alone.
extern "C" int msm_ray_table_check(const double *xyz, const int32_t *tri, int32_t V, int32_t T, int32_t nsamples, uint64_t seed, int64_t report[10]) {
    using namespace msm;
    if (!xyz || !tri || !report || V <= 0 || T <= 0 || nsamples < 0) return fail(MSM_ERR_INVALID, "msm_ray_table_check: bad arguments");
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i)
        if (tri[i] < 0 || tri[i] >= V) return fail(MSM_ERR_INVALID, "msm_ray_table_check: triangle vertex id %d out of range [0,%d)", tri[i], V);
    FlatOctree o;
    build_octree(xyz, tri, V, T, o);
    build_ray_table(xyz, tri, V, T, o);
    for (int k = 0; k < 10; ++k) report[k] = 0;
    report[7] = o.simple ? 1 : 0;
    if (o.ray_G <= 0) return MSM_OK;
    std::vector<int> guarded;  // triangles with exclusion boxes: every fifth point is placed above one of them
    for (int t = 0; t < T; ++t) {
        if (o.ray_edge[3 * (size_t)t].w >= 2.f) ++report[5];
        else if (__builtin_bit_cast(int32_t, o.ray_edge[3 * (size_t)t + 1].w) >= 0) guarded.push_back(t);
    }
    report[6] = (int64_t)guarded.size();
    DevTree dt{};
    dt.ray_G = o.ray_G;
    dt.ray_cell = o.ray_cell.data();
    dt.ray_more = o.ray_more.data();
    dt.ray_excl = o.ray_excl.data();
    dt.ray_r2lo = o.ray_r2lo, dt.ray_r2hi = o.ray_r2hi;
    // the records themselves: every leaf that meets the shell region of a guarded triangle without listing it must be refused
    for (int t : guarded) {
        const V3 v[3] = {vtx(xyz, V, tri[t]), vtx(xyz, V, tri[T + t]), vtx(xyz, V, tri[2 * (size_t)T + t])};
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (int k = 0; k < 3; ++k) {
            const V3 u = scale(v[k], 1.0 / norm(v[k]));
            const double c[3] = {u.x, u.y, u.z};
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::fmin(lo[a], std::fmin((kRad - kRayShell) * c[a], (kRad + kRayShell) * c[a]));
                hi[a] = std::fmax(hi[a], std::fmax((kRad - kRayShell) * c[a], (kRad + kRayShell) * c[a]));
            }
        }
        double cosmax = 1.0;  // the region's box grown by the sagitta, as build_ray_table does
        for (int k = 0; k < 3; ++k) cosmax = std::fmin(cosmax, dot(scale(v[k], 1.0 / norm(v[k])), scale(v[(k + 1) % 3], 1.0 / norm(v[(k + 1) % 3]))));
        const double sag = (kRad + kRayShell) * (1 - std::sqrt(cosmax)) * (1 + 1e-9) + 1e-9;
        for (int a = 0; a < 3; ++a) lo[a] -= sag, hi[a] += sag;
        std::vector<int> lack;
        lacking_leaves(o, 0, lo, hi, t, 1 << 20, lack);
        for (int leaf : lack) {
            const double4 b = o.nodebox[leaf];
            const V3 centre = mk(b.x + 0.5 * b.w, b.y + 0.5 * b.w, b.z + 0.5 * b.w);
            ++report[8];
            if (ray_vouches(dt, o.ray_edge[3 * (size_t)t + 1], centre)) ++report[4];
        }
    }
    uint64_t rs = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() {  // uniform in [0, 1)
        rs = rs * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
    };
    auto vert = [&](int t, int k) { return vtx(xyz, V, tri[(size_t)k * T + t]); };
    std::vector<int> hits;
    for (int it = 0; it < nsamples; ++it) {
        V3 p;
        const int kind = (it % 5 == 4 && !guarded.empty()) ? 4 : it % 4;
        if (kind == 4) {  // above a triangle with exclusion boxes, towards its rim (where the shell region leaves the triangle's own box)
            const int t = guarded[(size_t)(rnd() * guarded.size()) % guarded.size()];
            double w[3] = {rnd(), rnd() * 0.05, rnd() * 0.05};
            const int k = (int)(rnd() * 3) % 3;
            std::swap(w[0], w[k]);
            const V3 a = vert(t, 0), b = vert(t, 1), c = vert(t, 2);
            p = mk(w[0] * a.x + w[1] * b.x + w[2] * c.x, w[0] * a.y + w[1] * b.y + w[2] * c.y, w[0] * a.z + w[1] * b.z + w[2] * c.z);
        } else if (kind == 0) {  // a random direction
            do p = mk(2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1);
            while (!(norm(p) > 0.1 && norm(p) <= 1.0));
        } else {  // next to an edge (kind 1, 2) or a vertex (kind 3) of a random triangle, on either side
            const int t = (int)(rnd() * T) % T, k = (int)(rnd() * 3) % 3;
            const V3 a = vert(t, k), b = vert(t, (k + 1) % 3), c = vert(t, (k + 2) % 3);
            const double along = kind == 3 ? (rnd() < 0.5 ? 0.0 : 1e-9 * rnd()) : rnd();
            const V3 on = mk(a.x + along * (b.x - a.x), a.y + along * (b.y - a.y), a.z + along * (b.z - a.z));
            const double off = (rnd() < 0.5 ? -1.0 : 1.0) * std::pow(10.0, -10.0 + 7.0 * rnd()) * (rnd() < 0.1 ? 0.0 : 1.0);
            p = mk(on.x + off * (c.x - on.x), on.y + off * (c.y - on.y), on.z + off * (c.z - on.z));
        }
        const double radius = kRad + (it % 7 == 0 ? (2 * rnd() - 1) * 0.9e-4 : 0.0);  // the shell the table vouches for: 1e-4 either side
        p = scale(normalized(p), radius);
        ++report[0];
        // the reference's first pass: descend to the leaf, every listed triangle through the inside test
        int n = 0;
        while (o.node[n].x >= 0) {
            const double4 b = o.nodebox[n];
            const double mx = (b.x + (b.x + b.w)) / 2.0, my = (b.y + (b.y + b.w)) / 2.0, mz = (b.z + (b.z + b.w)) / 2.0;
            n = o.node[n].x + 4 * (!(p.x < mx)) + 2 * (!(p.y < my)) + (!(p.z < mz));
        }
        hits.clear();
        for (int e = 0; e < -o.node[n].x - 1; ++e) {
            const int t = o.leaf_tri[o.node[n].y + e];
            const V3 a = vert(t, 0), b = vert(t, 1), c = vert(t, 2);
            if (point_in_triangle(project_point(p, a, b, c), a, b, c)) hits.push_back(t);
        }
        // what a kernel may accept
        float fx, fy, fz;
        const int4 cell = ray_cell_of(dt, p, fx, fy, fz);
        int4 more = make_int4(cell.w, -1, -1, -1);
        if (cell.x >= 0 && cell.w < -1) more = o.ray_more[-2 - cell.w];
        const int cand[7] = {cell.x, cell.y, cell.z, more.x, more.y, more.z, more.w};
        const double pn = norm(p);
        bool by_float = false, by_fp64 = false, bad = false;
        for (int k = 0; k < 7 && cell.x >= 0; ++k) {
            const int t = cand[k];
            if (t < 0) break;
            const float4 e0 = o.ray_edge[3 * (size_t)t], e1 = o.ray_edge[3 * (size_t)t + 1], e2 = o.ray_edge[3 * (size_t)t + 2];
            float least;
            const int lvl = ray_accept_level(e0, e1, e2, fx, fy, fz, least);
            bool ok = lvl == 2;
            const bool fp64 = lvl == 1 && ray_accepts_fp64(vert(t, 0), vert(t, 1), vert(t, 2), p, pn, (double)e0.w - kRayFloatAllowance + 1e-12);
            ok = ok || fp64;
            if (ok && !ray_vouches(dt, e1, p)) ++report[9];
            if (!ok || !ray_vouches(dt, e1, p)) continue;
            (lvl == 2 ? by_float : by_fp64) = true;
            if (!(hits.size() == 1 && hits[0] == t)) bad = true;
        }
        if (bad) ++report[4];
        if (by_float) ++report[1];
        else if (by_fp64) ++report[2];
        else ++report[3];
    }
    return MSM_OK;
}
