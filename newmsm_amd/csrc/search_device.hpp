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
//   - child boxes are exact halvings of (-101,101), so the first grid_depth levels of the descent are
//     one arithmetic cell lookup in a dense grid (1 load instead of a chain of dependent loads);
//   - a float bounding-cone test (conservative, see octree.cpp) discards most leaf entries before the
//     exact FP64 test, and hits are first collected in a bit mask so that the expensive exact tests of
//     the 64 lanes of a wavefront run together instead of at 34 different loop iterations;
//   - dist_to_point is only evaluated when a second triangle also passes the inside test (its value
//     cannot change the result otherwise);
//   - only the winning triangle id is carried (few live registers -> more wavefronts per SIMD); callers
//     re-read its 128-byte record, which is L1/L2 resident by then.
#pragma once

#include "internal.hpp"

namespace msm {

struct ScanState {
    int best;      // winning triangle so far, -1 = EMPTY_TRIANGLE
    int have_d;    // bestd holds dist_to_point of `best`
    double bestd;
};

__device__ __forceinline__ V3 rec_v0(const TriRec &r) { return mk(r.v0[0], r.v0[1], r.v0[2]); }
__device__ __forceinline__ V3 rec_v1(const TriRec &r) { return mk(r.v1[0], r.v1[1], r.v1[2]); }
__device__ __forceinline__ V3 rec_v2(const TriRec &r) { return mk(r.v2[0], r.v2[1], r.v2[2]); }

// distance_to_triangle (R/octree.cpp:143-154): the projected point and whether it is inside
__device__ __forceinline__ bool inside_test(const TriRec &r, const V3 &p, V3 &mp) {
    mp = project_with_plane(p, mk(r.s3[0], r.s3[1], r.s3[2]), r.d);
    return point_in_triangle(mp, rec_v0(r), rec_v1(r), rec_v2(r));
}

// Ray table (octree.cpp: build_ray_table): the direction cell of p, i.e. up to four candidate triangles, likeliest
// first; all -1 when p is off the radius shell the table vouches for (or NaN).  Float arithmetic only proposes
// candidates and accepts them with the margins built into the thresholds; it never decides between two candidates.
// (the functions of the table's acceptance test also compile for the host: octree.cpp checks the table's guarantee with them, msm_ray_table_check)
MSM_HD int ray_clamp(int i, int hi) {  // max(0, min(hi, i))
#ifdef __HIP_DEVICE_COMPILE__
    return max(0, min(hi, i));
#else
    return i < 0 ? 0 : (i > hi ? hi : i);
#endif
}
MSM_HD float ray_rsqrt(float x) {
#ifdef __HIP_DEVICE_COMPILE__
    return rsqrtf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
MSM_HD int4 ray_cell_of(const DevTree &T, const V3 &p, float &fx, float &fy, float &fz) {
    const double r2 = p.x * p.x + p.y * p.y + p.z * p.z;
    fx = fy = fz = 0.f;
    if (!(r2 >= T.ray_r2lo && r2 <= T.ray_r2hi)) return make_int4(-1, -1, -1, -1);
    const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
    const float inv = ray_rsqrt(qx * qx + qy * qy + qz * qz);
    fx = qx * inv, fy = qy * inv, fz = qz * inv;
    const float ax = fabsf(fx), ay = fabsf(fy), az = fabsf(fz);
    int face;
    float w, u, v;
    if (ax >= ay && ax >= az) {
        face = fx < 0.f ? 1 : 0, w = ax, u = fy, v = fz;
    } else if (ay >= az) {
        face = fy < 0.f ? 3 : 2, w = ay, u = fz, v = fx;
    } else {
        face = fz < 0.f ? 5 : 4, w = az, u = fx, v = fy;
    }
    const float iw = 1.0f / w, half = 0.5f * (float)T.ray_G;
    const int G = T.ray_G;
    const int iu = ray_clamp((int)((u * iw + 1.0f) * half), G - 1);
    const int iv = ray_clamp((int)((v * iw + 1.0f) * half), G - 1);
    return T.ray_cell[((size_t)face * G + iu) * G + iv];
}

// the acceptance test of one candidate: all three edge-plane products at or above the triangle's threshold (e0.w)
MSM_HD bool ray_accepts(const float4 &e0, const float4 &e1, const float4 &e2, float fx, float fy, float fz) {
    const float d0 = __builtin_fmaf(e0.z, fz, __builtin_fmaf(e0.y, fy, e0.x * fx));
    const float d1 = __builtin_fmaf(e1.z, fz, __builtin_fmaf(e1.y, fy, e1.x * fx));
    const float d2 = __builtin_fmaf(e2.z, fz, __builtin_fmaf(e2.y, fy, e2.x * fx));
    return (int)(d0 >= e0.w) & (int)(d1 >= e0.w) & (int)(d2 >= e0.w);
}

// The float test's allowance inside the stored threshold (octree.cpp: build_ray_table adds it to margin / rho: direction 2e-7, normals 6e-8, dot
// 2e-7 and slack), and the two tests that use it: level 1 of ray_accept_level ("no float evaluation of a point the FP64 test would accept lies below this") and the FP64
// test itself -- the inward unit normals of the planes through the origin and each edge from the FP64 vertices, p^ . n_k >= thr for k = 0, 1, 2
// with thr = (stored threshold) - allowance (+ 1e-12 for its own rounding), which is at least the margin / rho the proof asks for.
// 2: accepted (as ray_accepts); 1: not accepted, but no product is further below the threshold than the allowance allows; 0: neither
// (`least`: the smallest of the three products -- among nearly accepted candidates the one with the largest is the likeliest to hold the point)
MSM_HD int ray_accept_level(const float4 &e0, const float4 &e1, const float4 &e2, float fx, float fy, float fz, float &least) {
    const float d0 = __builtin_fmaf(e0.z, fz, __builtin_fmaf(e0.y, fy, e0.x * fx));
    const float d1 = __builtin_fmaf(e1.z, fz, __builtin_fmaf(e1.y, fy, e1.x * fx));
    const float d2 = __builtin_fmaf(e2.z, fz, __builtin_fmaf(e2.y, fy, e2.x * fx));
    least = fminf(d0, fminf(d1, d2));
    const float lo = e0.w - 6.5e-6f;  // (comparisons, not the minimum: a NaN product must not pass)
    if ((int)(d0 >= e0.w) & (int)(d1 >= e0.w) & (int)(d2 >= e0.w)) return 2;
    return (int)(d0 >= lo) & (int)(d1 >= lo) & (int)(d2 >= lo);
}
MSM_HD bool ray_accepts_fp64(const V3 &v0, const V3 &v1, const V3 &v2, const V3 &p, double pn, double thr) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const V3 &a = k == 0 ? v1 : (k == 1 ? v2 : v0), &b = k == 0 ? v2 : (k == 1 ? v0 : v1), &c = k == 0 ? v0 : (k == 1 ? v1 : v2);
        const V3 n = cross(a, b);  // plane through the origin and the edge opposite vertex k; turned towards vertex k
        double val = dot(n, p);
        if (dot(n, c) < 0.0) val = -val;
        ok = ok && val >= thr * norm(n) * pn;
    }
    return ok;
}

// Is the accepted triangle listed in the octree leaf p descends to?  Yes unless p lies in one of the (at most seven)
// leaf boxes recorded for the triangle (octree.cpp: build_ray_table); e1.w carries the index of that record, or -1.
MSM_HD bool ray_vouches(const DevTree &T, const float4 &e1, const V3 &p);

__device__ __forceinline__ double candidate_distance(const DevTree &T, int t, const V3 &p) {
    const TriRec &r = T.rec[t];
    V3 mp;
    inside_test(r, p, mp);
    return dist_to_point(mp, rec_v0(r), rec_v1(r), rec_v2(r));
}

// the running-minimum update of R/octree.cpp:172-178 for one entry that passed the cone test
__device__ __forceinline__ void exact_candidate(const DevTree &T, int t, const V3 &p, ScanState &s) {
    V3 mp;
    if (!inside_test(T.rec[t], p, mp)) return;
    if (s.best < 0) {  // first triangle that contains the projection: accepted whatever its distance
        s.best = t;
        s.have_d = 0;
        return;
    }
    if (!s.have_d) {
        s.bestd = candidate_distance(T, s.best, p);
        s.have_d = 1;
    }
    const double d = candidate_distance(T, t, p);
    if (d > -1.0 && d < s.bestd) {
        s.best = t;
        s.bestd = d;
    }
}

// the float cone test of one leaf entry; a filter with its own safety margin: fused multiply-adds are fine here
__device__ __forceinline__ bool cone_pass(const float4 &c, float fx, float fy, float fz) {
    const float dt = __builtin_fmaf(c.z, fz, __builtin_fmaf(c.y, fy, c.x * fx));
    return fabsf(dt) >= c.w;
}

// cone filter for 8 consecutive leaf entries (one 128-byte line); bit j set = entry j may contain p
__device__ __forceinline__ unsigned cone_batch(const float4 *__restrict__ cb, float fx, float fy, float fz) {
    float4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = cb[j];
    unsigned bits = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= (cone_pass(c[j], fx, fy, fz) ? 1u : 0u) << j;
    return bits;
}

// one leaf: cone pre-filter (8 entries per step), then exact tests in entry order
__device__ __forceinline__ void scan_leaf(const DevTree &T, int beg, int cnt, const V3 &p, float fx, float fy, float fz, ScanState &s) {
    for (int eb = 0; eb < cnt; eb += 8) {
        unsigned bits = cone_batch(T.cone + beg + eb, fx, fy, fz);
        while (bits) {
            const int e = __ffs((int)bits) - 1;
            bits &= bits - 1;
            exact_candidate(T, T.leaf_tri[beg + eb + e], p, s);
        }
    }
}

// Steps 3 and 4 of get_closest_triangle (R/octree.cpp:180-208); rare, kept out of line so that the hot
// path does not pay registers for it
__device__ __forceinline__ int fallback_search(const DevTree &T, int n, const V3 &p, float fx, float fy, float fz) {
    ScanState s;
    s.best = -1;
    s.have_d = 0;
    s.bestd = DBL_MAX;
    const int par = T.parent[n];
    if (par < 0) return MSM_ERR_NOTFOUND;  // the reference dereferences a null parent here
    const int first = T.node[par].x;
    for (int c = 0; c < 8; ++c) {
        const int4 sib = T.node[first + c];
        if (sib.x < 0) scan_leaf(T, sib.y, -sib.x - 1, p, fx, fy, fz, s);
    }
    if (s.best >= 0) return s.best;
    // closest vertex by geodesic distance, R/octree.cpp:195-208
    double bestd = DBL_MAX;
    for (int c = 0; c < 8; ++c) {
        const int4 sib = T.node[first + c];
        if (sib.x >= 0) continue;
        for (int e = 0; e < -sib.x - 1; ++e) {
            const int t = T.leaf_tri[sib.y + e];  // e < count: never a padding entry
            const TriRec &r = T.rec[t];
            for (int v = 0; v < 3; ++v) {
                const double *vv = v == 0 ? r.v0 : (v == 1 ? r.v1 : r.v2);
                const double d = chord_to_arc(norm(sub(mk(vv[0], vv[1], vv[2]), p)));
                if (d < bestd) {
                    s.best = t;
                    bestd = d;
                }
            }
        }
    }
    return s.best >= 0 ? s.best : MSM_ERR_NOTFOUND;
}

// index of the grid cell along one axis: the number of cell boundaries b_j = -101 + j*h that are <= p,
// i.e. exactly the upper/lower choices the reference's descent makes with its (lo+hi)/2.0 midpoints
// (h = 202/G and all b_j are exact in FP64)
MSM_HD int grid_axis(double p, int G, double h) {
    int i = (int)((p + kBounds) * (1.0 / h));  // estimate; the two comparisons below make it exact
    i = ray_clamp(i, G - 1);
    if (i + 1 < G && !(p < -kBounds + (i + 1) * h)) ++i;
    else if (i > 0 && p < -kBounds + i * h) --i;
    return (p == p) ? i : G - 1;  // NaN: every child "contains" it, the last one wins
}

MSM_HD bool ray_vouches(const DevTree &T, const float4 &e1, const V3 &p) {
    const int k = __builtin_bit_cast(int, e1.w);
    if (k < 0) return true;
    const int4 b = T.ray_excl[k];
    auto lies_in = [&](int box) {
        const int d = box >> 24, G = 1 << d;
        const double h = 2 * kBounds / G;
        return grid_axis(p.x, G, h) == ((box >> 16) & 0xff) && grid_axis(p.y, G, h) == ((box >> 8) & 0xff) && grid_axis(p.z, G, h) == (box & 0xff);
    };
    const int first[3] = {b.x, b.y, b.z};
    bool clear = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j >= b.w) continue;
        clear = clear && !lies_in(first[j]);
    }
    if (b.w > 3) {  // boxes 3 .. 6 in the next record (a triangle in 1 000 of an icosphere)
        const int4 b2 = T.ray_excl[k + 1];
        const int rest[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (3 + j >= b.w) continue;
            clear = clear && !lies_in(rest[j]);
        }
    }
    return clear;
}

// Ray-table search for one point: the triangle the reference's search returns, or -1 when the table cannot vouch for
// it (then the caller runs find_closest_triangle).  Same test as k_unary_rays (unary_kernels.hip).
__device__ __forceinline__ int ray_find(const DevTree &T, const V3 &p) {
    if (T.ray_G <= 0) return -1;
    float fx, fy, fz;
    const int4 c = ray_cell_of(T, p, fx, fy, fz);
    if (c.x < 0) return -1;
    int4 mo = make_int4(c.w, -1, -1, -1);
    if (c.w < -1) mo = T.ray_more[-2 - c.w];
#pragma unroll 1
    for (int k = 0; k < 7; ++k) {
        const int ck = k == 0 ? c.x : (k == 1 ? c.y : (k == 2 ? c.z : (k == 3 ? mo.x : (k == 4 ? mo.y : (k == 5 ? mo.z : mo.w)))));
        if (ck < 0) break;
        const float4 *r = T.ray_tri + (size_t)kRayPieces * ck;
        const float4 e0 = r[0], e1 = r[1], e2 = r[2];
        if (ray_accepts(e0, e1, e2, fx, fy, fz)) return ray_vouches(T, e1, p) ? ck : -1;
    }
    return -1;
}

// Returns the triangle id (>= 0), or MSM_ERR_OUTSIDE / MSM_ERR_NOTFOUND.
__device__ __forceinline__ int find_closest_triangle(const DevTree &T, const V3 &p, bool allow_fallback = true) {
    // Node::contains_point of the root, R/node.cpp:58-68 (written so that NaN behaves as in the reference)
    if (p.x < -kBounds || p.x > kBounds || p.y < -kBounds || p.y > kBounds || p.z < -kBounds || p.z > kBounds) return MSM_ERR_OUTSIDE;
    const int G = 1 << T.grid_depth;
    const double h = 2 * kBounds / G;
    const int ix = grid_axis(p.x, G, h), iy = grid_axis(p.y, G, h), iz = grid_axis(p.z, G, h);
    int n = T.grid[((size_t)ix * G + iy) * G + iz];
    int4 nd = T.node[n];
    if (nd.x >= 0) {  // deeper than the grid: continue the reference's descent from this cell's box
        double lx = -kBounds + ix * h, hx = -kBounds + (ix + 1) * h;
        double ly = -kBounds + iy * h, hy = -kBounds + (iy + 1) * h;
        double lz = -kBounds + iz * h, hz = -kBounds + (iz + 1) * h;
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
    }
    // direction of p in float for the cone pre-filter
    const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
    const float inv = rsqrtf(qx * qx + qy * qy + qz * qz);
    const float fx = qx * inv, fy = qy * inv, fz = qz * inv;

    ScanState s;
    s.best = -1;
    s.have_d = 0;
    s.bestd = DBL_MAX;
    scan_leaf(T, nd.y, -nd.x - 1, p, fx, fy, fz, s);
    if (s.best >= 0) return s.best;

    // rare: nothing in the leaf contains the projection
    if (!allow_fallback) return MSM_ERR_NOTFOUND;  // profiling variants only
    return fallback_search(T, n, p, fx, fy, fz);
}


__device__ __forceinline__ bool outside_root(const V3 &p) {
    return p.x < -kBounds || p.x > kBounds || p.y < -kBounds || p.y > kBounds || p.z < -kBounds || p.z > kBounds;
}


// locate the octree leaf of p (same decisions as find_closest_triangle) and the sub-cell of p inside the leaf's
// box: the box is cut 4x4x4 and the three cut positions per axis are exact dyadics, so plain comparisons place p
// in the same (closed) sub-cell the mask builder reasoned about.
__device__ __forceinline__ int4 locate_leaf(const DevTree &T, const V3 &p, int &subcell) {
    const int G = 1 << T.grid_depth;
    const double h = 2 * kBounds / G;
    const int ix = grid_axis(p.x, G, h), iy = grid_axis(p.y, G, h), iz = grid_axis(p.z, G, h);
    int4 nd = T.node[T.grid[((size_t)ix * G + iy) * G + iz]];
    double lx, ly, lz, size;
    if (nd.x < 0) {  // a leaf at depth w <= grid_depth: its box is the depth-w cell above this grid cell
        const int up = T.grid_depth - nd.w;
        size = ldexp(2 * kBounds, -nd.w);  // 202 / 2^w, exact
        lx = -kBounds + (ix >> up) * size;
        ly = -kBounds + (iy >> up) * size;
        lz = -kBounds + (iz >> up) * size;
    } else {
        lx = -kBounds + ix * h;
        ly = -kBounds + iy * h;
        lz = -kBounds + iz * h;
        double hx = lx + h, hy = ly + h, hz = lz + h;
        while (nd.x >= 0) {
            const double mx = (lx + hx) / 2.0, my = (ly + hy) / 2.0, mz = (lz + hz) / 2.0;
            const int cx = !(p.x < mx), cy = !(p.y < my), cz = !(p.z < mz);
            if (cx) lx = mx; else hx = mx;
            if (cy) ly = my; else hy = my;
            if (cz) lz = mz; else hz = mz;
            nd = T.node[nd.x + 4 * cx + 2 * cy + cz];
        }
        size = hx - lx;
    }
    const double q = size / 4;
    const int sx = (p.x >= lx + q) + (p.x >= lx + 2 * q) + (p.x >= lx + 3 * q);
    const int sy = (p.y >= ly + q) + (p.y >= ly + 2 * q) + (p.y >= ly + 3 * q);
    const int sz = (p.z >= lz + q) + (p.z >= lz + 2 * q) + (p.z >= lz + 3 * q);
    subcell = 16 * sx + 4 * sy + sz;
    return nd;
}


// ------------------------------------------------------------------------------------------------
// The search of one point by the eight lanes of a group (lane & 7; all 64 lanes of the wavefront must call this
// together, groups with valid == false just ride along).  The lanes share the point and split the set bits of the
// leaf's sub-cell mask (cone test + exact inside test each).  Exactly one containing triangle = the reference's answer
// (R/octree.cpp:166-178: a single passing triangle wins whatever its distance); several are resolved in entry order
// with dist_to_point evaluated by the owner lanes (:172-178).  Returns the triangle, or kGroupUndecided when the
// complete search is needed (nothing in the leaf, no masks, outside the root, NaN).  The value is the same in all
// lanes of the group.  With a lane per point the exact tests of 64 lanes end up at 64 different loop positions and
// run one after the other (32 us for 2 % of an ico6 table); what counts for these short lists is the length of one
// wavefront's dependent chain.
// ------------------------------------------------------------------------------------------------
constexpr int kGroupUndecided = -100;

__device__ __forceinline__ int group8_find(const DevTree &T, bool valid, const V3 &p, int lane) {
    const int sub = lane & 7;
    int4 leaf = make_int4(-1, 0, -1, 0);
    unsigned long long mm = 0ull;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    bool serial = !valid;
    if (valid) {
        if (outside_root(p) || !(p.x == p.x && p.y == p.y && p.z == p.z) || !T.mask) {
            serial = true;
        } else {
            int subcell;
            leaf = locate_leaf(T, p, subcell);
            if (leaf.z < 0) {
                serial = true;  // empty or oversized leaf: no masks
            } else {
                mm = T.mask[(size_t)leaf.z * 64 + subcell];
                const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
                const float inv = rsqrtf(qx * qx + qy * qy + qz * qz);
                fx = qx * inv, fy = qy * inv, fz = qz * inv;
            }
        }
    }
    unsigned long long hits = 0ull;  // entries whose triangle contains the projection (same in all lanes of a group)
    // lane `sub` of the group takes the sub-th, (sub+8)-th, ... set bit of the mask: ceil(popcount / 8) rounds
    unsigned long long mine = serial ? 0ull : mm;
    for (int k = 0; k < sub; ++k) mine &= mine - 1ull;
    while (__any(mine != 0ull)) {
        bool hit = false;
        int e = 0;
        if (mine) {
            e = __ffsll((long long)mine) - 1;
            const float4 c = T.cone[leaf.y + e];
            const float dt = fabsf(__builtin_fmaf(c.z, fz, __builtin_fmaf(c.y, fy, c.x * fx)));
            if (dt >= c.w) {
                V3 mp;
                hit = inside_test(T.rec[T.leaf_tri[leaf.y + e]], p, mp);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) mine &= mine - 1ull;  // my next bit is eight set bits further
        }
        const unsigned long long bal = __ballot(hit);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ek = __shfl(e, (lane & ~7) | k, 64);
            if ((bal >> ((lane & ~7) | k)) & 1ull) hits |= 1ull << ek;
        }
    }
    int win = -1;
    double bestd = 0.0;
    unsigned long long rest = (!serial && __popcll(hits) >= 2) ? hits : 0ull;
    while (__any(rest != 0ull)) {
        const int e = rest ? __ffsll((long long)rest) - 1 : 0;
        double d = 0.0;
        if (rest && (e & 7) == sub) d = candidate_distance(T, T.leaf_tri[leaf.y + e], p);
        d = __shfl(d, (lane & ~7) | (e & 7), 64);
        if (rest) {
            if (win < 0 || (d > -1.0 && d < bestd)) {
                win = e;
                bestd = d;
            }
            rest &= rest - 1ull;
        }
    }
    if (serial) return kGroupUndecided;
    if (__popcll(hits) == 1) return T.leaf_tri[leaf.y + __ffsll((long long)hits) - 1];
    if (win >= 0) return T.leaf_tri[leaf.y + win];
    return kGroupUndecided;
}

// ------------------------------------------------------------------------------------------------
// The COMPLETE search of one point by a group of G = 4 or 8 neighbouring lanes (lane & (G - 1)); all 64 lanes of the
// wavefront must call this together, groups with valid == false ride along.  Unlike group8_find it needs no sub-cell
// masks (a tree built a moment ago has none) and ends with the reference's answer in every case, without a serial path:
//   descent (arithmetic, as find_closest_triangle) -> the leaf's cone filter, G entries per load round, two rounds in
//   flight -> the entries that pass are dealt out to the lanes in entry order (about 2.4 per query: one round of exact
//   FP64 tests) -> the containing triangles are taken in entry order as R/octree.cpp:166-178 does: the first one, then
//   every strictly closer one (dist_to_point is only evaluated when a second triangle shows up; every lane of the group
//   computes it -- same instructions, same value, no exchange) -> none: the same over the sibling leaves with one running
//   minimum (:180-193), then the nearest vertex of their triangles (:195-208: entries split over the lanes, first minimum
//   in the reference's visiting order).
// With a lane per query the 34 cone tests and the exact tests of 64 lanes run at 64 different loop positions (75 us for
// the 40 962 queries of an ico6 mesh); what counts for such a launch is the length of one wavefront's dependent chain.
// `out.compute(rec, p, projected p)` is called by the lane that finds a containing triangle (the record and the projection
// are in its registers then); on return `owner` is true in exactly one lane of every valid group -- a lane whose `out`
// belongs to the returned triangle (for an error code: the group's first lane).  The result is the same in all lanes
// of a group: a triangle id, MSM_ERR_OUTSIDE or MSM_ERR_NOTFOUND.
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ unsigned or_group(unsigned v) {
    static_assert(G == 4 || G == 8, "groups of 4 or 8 lanes");
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    if (G == 8) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror: lane i <-> 7 - i of each eight
    return v;
}
template <int G>
__device__ __forceinline__ unsigned long long or_group64(unsigned long long v) {
    return ((unsigned long long)or_group<G>((unsigned)(v >> 32)) << 32) | or_group<G>((unsigned)v);
}

// out of line: asin inlined three times costs the search kernels 34 registers for a path that almost never runs
__device__ __noinline__ double arc_of_chord_slow(double c) { return chord_to_arc(c); }

template <int G, class Payload>
__device__ __forceinline__ int group_search(const DevTree &T, bool valid, const V3 &p, int lane, Payload &out, bool &owner) {
    const int sl = lane & (G - 1);
    int n = 0, res = MSM_ERR_NOTFOUND;
    int4 nd = make_int4(-1, 0, -1, 0);  // an empty leaf
    float fx = 0.f, fy = 0.f, fz = 0.f;
    bool scan = false;
    owner = false;
    if (valid) {
        if (outside_root(p)) {
            res = MSM_ERR_OUTSIDE;
        } else {  // the descent of find_closest_triangle
            const int Gd = 1 << T.grid_depth;
            const double h = 2 * kBounds / Gd;
            const int ix = grid_axis(p.x, Gd, h), iy = grid_axis(p.y, Gd, h), iz = grid_axis(p.z, Gd, h);
            n = T.grid[((size_t)ix * Gd + iy) * Gd + iz];
            nd = T.node[n];
            if (nd.x >= 0) {
                double lx = -kBounds + ix * h, hx = -kBounds + (ix + 1) * h;
                double ly = -kBounds + iy * h, hy = -kBounds + (iy + 1) * h;
                double lz = -kBounds + iz * h, hz = -kBounds + (iz + 1) * h;
                while (nd.x >= 0) {
                    const double mx = (lx + hx) / 2.0, my = (ly + hy) / 2.0, mz = (lz + hz) / 2.0;
                    const int cx = !(p.x < mx), cy = !(p.y < my), cz = !(p.z < mz);
                    if (cx) lx = mx; else hx = mx;
                    if (cy) ly = my; else hy = my;
                    if (cz) lz = mz; else hz = mz;
                    n = nd.x + 4 * cx + 2 * cy + cz;
                    nd = T.node[n];
                }
            }
            const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
            const float inv = rsqrtf(qx * qx + qy * qy + qz * qz);
            fx = qx * inv, fy = qy * inv, fz = qz * inv;
            scan = true;
        }
    }
    int win = -1, myt = -2, sib0 = 0;
    bool have_d = false;
    double bestd = DBL_MAX;
    // pass 0: the leaf of p; passes 1..8 (only for groups that found nothing): the children of its parent, R/octree.cpp:180-193
    for (int pass = 0; pass <= 8; ++pass) {
        if (pass == 1) {
            const int par = scan && win < 0 ? T.parent[n] : -1;  // no parent: the reference dereferences a null pointer here; MSM_ERR_NOTFOUND
            scan = par >= 0;
            if (!__any(scan)) break;
            sib0 = scan ? T.node[par].x : 0;
        }
        if (pass >= 1) nd = scan ? T.node[sib0 + pass - 1] : make_int4(-1, 0, -1, 0);
        const int cnt = scan && nd.x < 0 ? -nd.x - 1 : 0;
        // 64 entries at a time (a leaf has more only when the split heuristic refused to split it: degenerate meshes)
        for (int eb = 0; __any(eb < cnt); eb += 64) {
            // --- cone filter: entry eb + G * round + sl; lists are padded to multiples of eight with cones nothing passes
            const int left = min(cnt - eb, 64), rounds = left > 0 ? (left + G - 1) / G : 0;
            const float4 *cb = T.cone + nd.y + eb;
            unsigned long long cand = 0ull;
            for (int r = 0; __any(r < rounds); r += 2) {
                float4 c0 = make_float4(0.f, 0.f, 0.f, 2.f), c1 = c0;
                if (r < rounds) c0 = cb[G * r + sl];
                if (r + 1 < rounds) c1 = cb[G * (r + 1) + sl];
                if (cone_pass(c0, fx, fy, fz)) cand |= 1ull << (G * r + sl);
                if (cone_pass(c1, fx, fy, fz)) cand |= 1ull << (G * (r + 1) + sl);
            }
            cand = or_group64<G>(cand);
            // --- exact tests: the k-th candidate in entry order goes to lane k % G in round k / G
            const int ncand = __popcll(cand);
            unsigned long long hits = 0ull;
            unsigned onehit = 0u;
            for (int r = 0; __any(r * G < ncand); ++r) {
                unsigned long long hb = 0ull;
                unsigned ht = 0u;
                const int want = r * G + sl;
                if (want < ncand) {
                    unsigned long long c = cand;
                    for (int k = 0; k < want; ++k) c &= c - 1ull;
                    const int e = __ffsll((long long)c) - 1;
                    const int t = T.leaf_tri[nd.y + eb + e];
                    const TriRec &rec = T.rec[t];
                    V3 mp;
                    if (inside_test(rec, p, mp)) {
                        hb = 1ull << e;
                        ht = (unsigned)t + 1u;
                        myt = t;
                        out.compute(rec, p, mp);
                    }
                }
                hits |= or_group64<G>(hb);
                onehit |= or_group<G>(ht);
            }
            // --- the containing triangles in entry order (every lane of the group does the same)
            if (win < 0 && __popcll(hits) == 1) {
                win = (int)onehit - 1;
                hits = 0ull;
            }
            while (__any(hits != 0ull)) {
                if (hits) {
                    const int t = T.leaf_tri[nd.y + eb + __ffsll((long long)hits) - 1];
                    hits &= hits - 1ull;
                    if (win < 0) {
                        win = t;
                    } else {
                        double d = 0.0;
                        for (int k = have_d ? 1 : 0; k < 2; ++k) {
                            d = candidate_distance(T, k == 0 ? win : t, p);
                            if (k == 0) bestd = d;
                        }
                        have_d = true;
                        if (d > -1.0 && d < bestd) {
                            win = t;
                            bestd = d;
                        }
                    }
                }
            }
        }
        if (pass == 0 && !__any(scan && win < 0)) break;
    }
    // --- the closest vertex among the siblings' triangles by geodesic distance (R/octree.cpp:195-208); scan: the group has a parent
    if (__any(scan && win < 0)) {
        const bool need = scan && win < 0;
        double vd = DBL_MAX;
        long long vord = 0x7fffffffffffffffll;
        int vt = -1;
        for (int c = 0; c < 8; ++c) {
            const int4 sib = need ? T.node[sib0 + c] : make_int4(0, 0, 0, 0);
            const int cnt = sib.x < 0 ? -sib.x - 1 : 0;
            for (int e0 = 0; __any(e0 < cnt); e0 += G) {
                const int e = e0 + sl;
                if (e < cnt) {
                    const int t = T.leaf_tri[sib.y + e];
                    const TriRec &r = T.rec[t];
                    for (int v = 0; v < 3; ++v) {
                        const double *vv = v == 0 ? r.v0 : (v == 1 ? r.v1 : r.v2);
                        const double d = arc_of_chord_slow(norm(sub(mk(vv[0], vv[1], vv[2]), p)));
                        if (d < vd) {  // within a lane the visiting order is ascending: the first minimum stays
                            vd = d;
                            vord = ((long long)c << 40) | ((long long)e << 2) | v;
                            vt = t;
                        }
                    }
                }
            }
        }
        for (int off = 1; off < G; off <<= 1) {  // the smallest distance, ties to the earliest in the reference's order
            const double od = __shfl_xor(vd, off, 64);
            const long long oo = __shfl_xor(vord, off, 64);
            const int ot = __shfl_xor(vt, off, 64);
            if (od < vd || (od == vd && oo < vord)) vd = od, vord = oo, vt = ot;
        }
        if (need) win = vt;
    }
    int t = res;
    if (win >= 0) {
        t = win;
        owner = myt == win;
    }
    // the winner's lane went on to another entry that also passed, the winner came from the nearest-vertex step, or an error: the first lane
    const bool orphan = valid && or_group<G>(owner ? 1u : 0u) == 0u;
    if (__any(orphan)) {
        if (orphan && sl == 0) {
            if (t >= 0) {
                const TriRec &rec = T.rec[t];
                V3 mp;
                inside_test(rec, p, mp);
                out.compute(rec, p, mp);
            }
            owner = true;
        }
    }
    return t;
}

}  // namespace msm
