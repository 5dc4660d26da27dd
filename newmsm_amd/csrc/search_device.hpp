// search_device.hpp -- device-side nearest-triangle search (gfx950).
//
// Octree::get_closest_triangle (R/octree.cpp:156-214) decides with, in this order:
//   1. descent to the LAST child (i,j,k order) whose closed box contains the point;
//   2. over that leaf's triangles in stored order: ray-project the point on the triangle's plane
//      (project_point), keep triangles for which point_in_triangle holds (slack -1e-8), winner = first
//      strictly smaller Triangle::dist_to_point;
//   3. if none: the same test over the 8 children of the leaf's parent;
//   4. if none: the triangle owning the vertex with the smallest geodesic distance among those.
// The device code makes the same decisions with the same FP64 arithmetic.  What it changes is cost:
//   - the descent needs no node bounds: child boxes are halvings of (-101,101), recomputed on the fly
//     with the reference's own (lo+hi)/2.0, so a level costs three compares and one 8-byte load;
//   - a float bounding-cone test (conservative, see octree.cpp) discards most leaf entries before the
//     exact FP64 test, and hits are first collected in a bit mask so that the expensive exact tests of
//     the 64 lanes of a wavefront run together instead of at 34 different loop iterations;
//   - dist_to_point is only evaluated when a second triangle also passes the inside test (its value
//     cannot change the result otherwise).
#pragma once

#include "internal.hpp"

namespace msm {

struct Hit {
    int tri;
    int id0, id1, id2;
    V3 v0, v1, v2;
};

struct ScanState {
    int best;
    bool have_d;
    double bestd;
    V3 best_mp;
    Hit hit;
};

__device__ __forceinline__ void take(ScanState &s, const TriRec &r, int t) {
    s.best = t;
    s.hit.tri = t;
    s.hit.id0 = r.id[0];
    s.hit.id1 = r.id[1];
    s.hit.id2 = r.id[2];
    s.hit.v0 = mk(r.v0[0], r.v0[1], r.v0[2]);
    s.hit.v1 = mk(r.v1[0], r.v1[1], r.v1[2]);
    s.hit.v2 = mk(r.v2[0], r.v2[1], r.v2[2]);
}

// distance_to_triangle (R/octree.cpp:143-154) + the running-minimum update (:172-178) for one entry
__device__ __forceinline__ void exact_candidate(const DevTree &T, int t, const V3 &p, ScanState &s) {
    const TriRec &r = T.rec[t];
    const V3 v0 = mk(r.v0[0], r.v0[1], r.v0[2]), v1 = mk(r.v1[0], r.v1[1], r.v1[2]), v2 = mk(r.v2[0], r.v2[1], r.v2[2]);
    const V3 mp = project_with_plane(p, mk(r.s3[0], r.s3[1], r.s3[2]), r.d);
    if (!point_in_triangle(mp, v0, v1, v2)) return;
    if (s.best < 0) {
        take(s, r, t);
        s.best_mp = mp;
        s.have_d = false;
        return;
    }
    if (!s.have_d) {
        s.bestd = dist_to_point(s.best_mp, s.hit.v0, s.hit.v1, s.hit.v2);
        s.have_d = true;
    }
    const double d = dist_to_point(mp, v0, v1, v2);
    if (d > -1.0 && d < s.bestd) {
        take(s, r, t);
        s.bestd = d;
    }
}

// one leaf: cone pre-filter into a bit mask, then exact tests in entry order
__device__ __forceinline__ void scan_leaf(const DevTree &T, int beg, int cnt, const V3 &p, float fx, float fy, float fz, ScanState &s) {
    for (int base = 0; base < cnt; base += 64) {
        const int m = min(64, cnt - base);
        unsigned long long mask = 0ull;
        const float4 *cone = T.cone + beg + base;
        for (int e = 0; e < m; ++e) {
            const float4 c = cone[e];
            const float dt = c.x * fx + c.y * fy + c.z * fz;
            if (fabsf(dt) >= c.w) mask |= 1ull << e;
        }
        while (mask) {
            const int e = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            exact_candidate(T, T.leaf_tri[beg + base + e], p, s);
        }
    }
}

// Returns the triangle id (>= 0) and fills `hit`, or MSM_ERR_OUTSIDE / MSM_ERR_NOTFOUND.
__device__ __forceinline__ int find_closest_triangle(const DevTree &T, const V3 &p, Hit &hit) {
    // Node::contains_point of the root, R/node.cpp:58-68 (written so that NaN behaves as in the reference)
    if (p.x < -kBounds || p.x > kBounds || p.y < -kBounds || p.y > kBounds || p.z < -kBounds || p.z > kBounds) return MSM_ERR_OUTSIDE;
    double lx = -kBounds, hx = kBounds, ly = -kBounds, hy = kBounds, lz = -kBounds, hz = kBounds;
    int n = 0;
    int2 nd = T.node[0];
    while (nd.x >= 0) {
        const double mx = (lx + hx) / 2.0, my = (ly + hy) / 2.0, mz = (lz + hz) / 2.0;
        // the upper child's closed box [mid, hi] contains p unless p < mid; the last containing child wins
        const int cx = !(p.x < mx), cy = !(p.y < my), cz = !(p.z < mz);
        if (cx) lx = mx; else hx = mx;
        if (cy) ly = my; else hy = my;
        if (cz) lz = mz; else hz = mz;
        n = nd.x + 4 * cx + 2 * cy + cz;
        nd = T.node[n];
    }
    // direction of p in float for the cone pre-filter
    const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
    const float inv = rsqrtf(qx * qx + qy * qy + qz * qz);
    const float fx = qx * inv, fy = qy * inv, fz = qz * inv;

    ScanState s;
    s.best = -1;
    s.have_d = false;
    s.bestd = DBL_MAX;
    scan_leaf(T, nd.y, -nd.x - 1, p, fx, fy, fz, s);
    if (s.best < 0) {
        const int par = T.parent[n];
        if (par < 0) return MSM_ERR_NOTFOUND;  // the reference dereferences a null parent here
        const int first = T.node[par].x;
        for (int c = 0; c < 8; ++c) {
            const int2 sib = T.node[first + c];
            if (sib.x < 0) scan_leaf(T, sib.y, -sib.x - 1, p, fx, fy, fz, s);
        }
        if (s.best < 0) {
            // closest vertex by geodesic distance, R/octree.cpp:195-208
            double bestd = DBL_MAX;
            for (int c = 0; c < 8; ++c) {
                const int2 sib = T.node[first + c];
                if (sib.x >= 0) continue;
                for (int e = 0; e < -sib.x - 1; ++e) {
                    const int t = T.leaf_tri[sib.y + e];
                    const TriRec &r = T.rec[t];
                    const double *vv[3] = {r.v0, r.v1, r.v2};
                    for (int v = 0; v < 3; ++v) {
                        const double d = chord_to_arc(norm(sub(mk(vv[v][0], vv[v][1], vv[v][2]), p)));
                        if (d < bestd) {
                            take(s, r, t);
                            bestd = d;
                        }
                    }
                }
            }
        }
        if (s.best < 0) return MSM_ERR_NOTFOUND;
    }
    hit = s.hit;
    return s.best;
}

}  // namespace msm
