// kernels.hip -- hand-written gfx950 kernels of libmsmhip and their launch wrappers.
//
// All arithmetic that decides an index (which triangle, which patch) is FP64 in the reference's own
// operation order (this file is compiled with -ffp-contract=off).  No MFMA: the path is gather /
// compare / short reductions.  Wavefront = 64 lanes throughout.
#include <climits>
#include <cstring>
#include <algorithm>

#include "kernels.hpp"
#include "search_device.hpp"

namespace msm {

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Sub-cell masks.  A leaf's box is cut into 4x4x4 sub-cells; bit e of mask[sub-cell] is set unless no point of
// that (closed) sub-cell can pass entry e's cone filter -- and with it the reference's inside test, which the
// cone contains.  One 64-lane workgroup per node, one lane per sub-cell.  All directions of points in a box lie
// within beta of the direction of its centre c, sin(beta) = half-diagonal / |c|; with theta the angle between c
// and the cone axis, the largest |cos| a point of the box can reach is cos(max(theta-beta,0)) on the axis side
// and -cos(min(theta+beta,pi)) on the antipodal side.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_build_masks(DevTree T, const double4 *__restrict__ nodebox, unsigned long long *__restrict__ mask) {
    const int n = blockIdx.x;
    const int4 nd = T.node[n];
    if (nd.x >= 0 || nd.z < 0) return;
    const int cnt = -nd.x - 1;
    const double4 box = nodebox[n];
    const double q = box.w / 4;
    const int s = threadIdx.x, sx = s >> 4, sy = (s >> 2) & 3, sz = s & 3;
    const double cx = box.x + (sx + 0.5) * q, cy = box.y + (sy + 0.5) * q, cz = box.z + (sz + 0.5) * q;
    const double cn = sqrt(cx * cx + cy * cy + cz * cz);
    const double rb = 0.5 * q * 1.7320508075688774 * (1 + 1e-9) + 1e-9;
    unsigned long long m = 0ull;
    if (!(cn > rb)) {
        m = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);  // the box reaches the origin: any direction
    } else {
        const double sb = rb / cn, cb = sqrt(1 - sb * sb);
        const double ux = cx / cn, uy = cy / cn, uz = cz / cn;
        for (int e = 0; e < cnt; ++e) {
            const float4 c = T.cone[nd.y + e];
            const double thr = (double)c.w - 1e-6;  // the device evaluates the cone test in float
            bool hit = thr <= 0.0;
            if (!hit) {
                const double an = sqrt((double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z);
                double ct = (ux * c.x + uy * c.y + uz * c.z) / an;
                ct = ct > 1 ? 1 : (ct < -1 ? -1 : ct);
                const double stt = sqrt(1 - ct * ct);
                // cos(theta - beta) and -cos(theta + beta), saturated at 1 when the axis (or its antipode) is inside the cap
                const double near = (ct >= cb) ? 1.0 : ct * cb + stt * sb;
                const double far = (-ct >= cb) ? 1.0 : -(ct * cb - stt * sb);
                hit = near >= thr || far >= thr;
            }
            if (hit) m |= 1ull << e;
        }
    }
    mask[(size_t)nd.z * 64 + s] = m;
}

int launch_build_masks(msm_ctx *ctx, const DevTree &T, const double4 *d_nodebox, unsigned long long *d_mask) {
    hipLaunchKernelGGL(k_build_masks, dim3(T.nnodes), dim3(64), 0, ctx->stream, T, d_nodebox, d_mask);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// Conservative bounding cone of the set of directions for which the reference's inside test
// (project_point + point_in_triangle with its -1e-8 slack, R/point.cpp:36-60) can succeed.
// A point passes same_side for edge e iff its in-plane signed distance from the edge line exceeds
// -1e-8 / (2 * area * |e|); the accepted region is the triangle grown by that band.
__device__ float4 bounding_cone(const V3 &a, const V3 &b, const V3 &c, double plane_d) {
    const float4 always = make_float4(0.f, 0.f, 1.f, -2.f);
    V3 axis = mk(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z);
    double an = norm(axis);
    if (!(an > 1e-12) || !isfinite(an)) return always;
    axis = scale(axis, 1.0 / an);
    double rho = 0.0;
    for (const V3 *v : {&a, &b, &c}) {
        double n = norm(*v);
        if (!(n > 1e-12)) return always;
        double cs = dot(axis, *v) / n;
        cs = cs > 1 ? 1 : (cs < -1 ? -1 : cs);
        rho = fmax(rho, acos(cs));
    }
    const double la = norm(sub(b, c)), lb = norm(sub(a, c)), lc = norm(sub(a, b));
    const double area = 0.5 * norm(cross(sub(b, a), sub(c, a)));
    const double lmin = fmin(la, fmin(lb, lc)), lmax = fmax(la, fmax(lb, lc));
    if (!(area > 0) || !(lmin > 0)) return always;
    // widest band over the three edges, plus rounding noise of the reference's own cross/dot products
    const double band = 1e-8 / (2 * area * lmin) + 1e-9 * (1 + lmax);
    // an offset polygon's corner moves by band / sin(angle/2); smallest interior angle from the altitude
    // sin(smallest interior angle) = 2*area / (product of its two sides) >= 2*area / lmax^2
    const double half = 0.5 * asin(fmin(1.0, 2 * area / (lmax * lmax)));
    const double reach = band / fmax(sin(half), 1e-300);
    const double h = fabs(plane_d);  // distance of the triangle's plane from the origin
    if (!(h > 0) || !(reach / h < 0.25) || !isfinite(reach)) return always;
    const double rho2 = rho + 1.0001 * asin(reach / h) + 1e-7;
    if (!(rho2 < 1.5)) return always;
    // 2e-6 absorbs float rounding of the query direction and of the dot product
    return make_float4((float)axis.x, (float)axis.y, (float)axis.z, (float)(cos(rho2) - 2e-6));
}



// The per-triangle records of the exact test (vertices, the triangle-only half of project_point, ids) and the bounding
// cones, from the mesh as it sits in HBM: 13.7 MB at ico6 that used to be computed on the host and copied over.
__global__ __launch_bounds__(256) void k_build_recs(const double *__restrict__ xyz, int V, const int32_t *__restrict__ tri, int T, TriRec *__restrict__ recs,
                                                     float4 *__restrict__ tcone) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int i0 = tri[t], i1 = tri[T + t], i2 = tri[2 * T + t];
    const V3 v0 = mk(xyz[i0], xyz[V + i0], xyz[2 * V + i0]), v1 = mk(xyz[i1], xyz[V + i1], xyz[2 * V + i1]), v2 = mk(xyz[i2], xyz[V + i2], xyz[2 * V + i2]);
    TriRec r;
    r.v0[0] = v0.x, r.v0[1] = v0.y, r.v0[2] = v0.z;
    r.v1[0] = v1.x, r.v1[1] = v1.y, r.v1[2] = v1.z;
    r.v2[0] = v2.x, r.v2[1] = v2.y, r.v2[2] = v2.z;
    V3 s3;
    plane_of(v0, v1, v2, s3, r.d);  // distance_to_triangle calls project_point(pt, v0, v1, v2), R/octree.cpp:149
    r.s3[0] = s3.x, r.s3[1] = s3.y, r.s3[2] = s3.z;
    r.id[0] = i0, r.id[1] = i1, r.id[2] = i2;
    r.tri = t;
    r.reserved = 0.0;
    recs[t] = r;
    tcone[t] = bounding_cone(v0, v1, v2, r.d);
}

// cone per (padded) leaf entry; padding entries get a cone nothing passes (|dot| <= 1 < 2)
__global__ __launch_bounds__(256) void k_expand_cones(const int32_t *__restrict__ leaf_tri, int n, const float4 *__restrict__ tcone, float4 *__restrict__ cone) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int t = leaf_tri[e];
    cone[e] = t >= 0 ? tcone[t] : make_float4(0.f, 0.f, 0.f, 2.f);
}

// the same for every tree of a forest (blockIdx.y)
__global__ __launch_bounds__(256) void k_build_recs_forest(const double *__restrict__ xyz, size_t comp, size_t tree, const int32_t *__restrict__ tri, int T,
                                                            TriRec *__restrict__ recs, float4 *__restrict__ tcone, size_t s_rec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    xyz += blockIdx.y * tree;
    recs += blockIdx.y * s_rec;
    tcone += blockIdx.y * s_rec;
    const int i0 = tri[t], i1 = tri[T + t], i2 = tri[2 * T + t];
    const V3 v0 = mk(xyz[i0], xyz[comp + i0], xyz[2 * comp + i0]), v1 = mk(xyz[i1], xyz[comp + i1], xyz[2 * comp + i1]), v2 = mk(xyz[i2], xyz[comp + i2], xyz[2 * comp + i2]);
    TriRec r;
    r.v0[0] = v0.x, r.v0[1] = v0.y, r.v0[2] = v0.z;
    r.v1[0] = v1.x, r.v1[1] = v1.y, r.v1[2] = v1.z;
    r.v2[0] = v2.x, r.v2[1] = v2.y, r.v2[2] = v2.z;
    V3 s3;
    plane_of(v0, v1, v2, s3, r.d);
    r.s3[0] = s3.x, r.s3[1] = s3.y, r.s3[2] = s3.z;
    r.id[0] = i0, r.id[1] = i1, r.id[2] = i2;
    r.tri = t;
    r.reserved = 0.0;
    recs[t] = r;
    tcone[t] = bounding_cone(v0, v1, v2, r.d);
}
__global__ __launch_bounds__(256) void k_expand_cones_forest(const int32_t *__restrict__ leaf_tri, size_t s_leaf, const int *__restrict__ entries, size_t entries_stride,
                                                              const float4 *__restrict__ tcone, size_t s_rec, float4 *__restrict__ cone) {
    const int n = entries[blockIdx.y * entries_stride];
    leaf_tri += blockIdx.y * s_leaf;
    cone += blockIdx.y * s_leaf;
    tcone += blockIdx.y * s_rec;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int t = leaf_tri[e];
        cone[e] = t >= 0 ? tcone[t] : make_float4(0.f, 0.f, 0.f, 2.f);
    }
}
int launch_build_recs_forest(msm_ctx *ctx, const double *d_xyz, size_t comp_stride, size_t tree_stride, const int32_t *d_tri, int T, int B, TriRec *d_rec, float4 *d_tcone,
                             size_t s_rec, const int32_t *d_leaf_tri, float4 *d_cone, size_t s_leaf, const int *d_entries, size_t entries_stride) {
    if (T <= 0 || B <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_build_recs_forest, dim3((T + 255) / 256, (unsigned)B), dim3(256), 0, ctx->stream, d_xyz, comp_stride, tree_stride, d_tri, T, d_rec, d_tcone, s_rec);
    hipLaunchKernelGGL(k_expand_cones_forest, dim3(512, (unsigned)B), dim3(256), 0, ctx->stream, d_leaf_tri, s_leaf, d_entries, entries_stride, d_tcone, s_rec, d_cone);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_build_recs(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, TriRec *d_rec, float4 *d_tcone, const int32_t *d_leaf_tri,
                      int nentries, float4 *d_cone) {
    if (T > 0) hipLaunchKernelGGL(k_build_recs, dim3((T + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, V, d_tri, T, d_rec, d_tcone);
    MSM_HIP(hipGetLastError());
    if (nentries > 0) hipLaunchKernelGGL(k_expand_cones, dim3((nentries + 255) / 256), dim3(256), 0, ctx->stream, d_leaf_tri, nentries, d_tcone, d_cone);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_build_raytri(const TriRec *__restrict__ rec, const float4 *__restrict__ edge, int T,
                                                       const double *__restrict__ feat1, int D, float4 *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const TriRec &r = rec[t];
    float4 *o = out + (size_t)kRayPieces * t;
    for (int k = 0; k < 3; ++k) o[k] = edge[3 * (size_t)t + k];
    double d[12];
    for (int k = 0; k < 3; ++k) {
        d[k] = r.v0[k];
        d[3 + k] = r.v1[k];
        d[6 + k] = r.v2[k];
        d[9 + k] = feat1 ? feat1[(size_t)r.id[k] * D] : 0.0;  // feature row 1 (vertex-major storage: column 0)
    }
    double2 *od = reinterpret_cast<double2 *>(o + 3);
    for (int k = 0; k < 6; ++k) od[k] = make_double2(d[2 * k], d[2 * k + 1]);
}

__global__ __launch_bounds__(256) void k_copy_to_mapped(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n2, const double *__restrict__ src1,
                                                         double *__restrict__ dst1, const int *__restrict__ status, int *__restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) dst[i] = src[i];
    if (i == 0) {
        if (src1) *dst1 = *src1;  // the odd element
        const int st = *status;
        if (st != 0) __hip_atomic_store(flags, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_copy_to_mapped(msm_ctx *ctx, const double *d_src, double *mapped_dst, size_t n, int *flags_mapped) {
    const size_t n2 = n / 2;
    const bool odd = (n & 1) != 0;
    hipLaunchKernelGGL(k_copy_to_mapped, dim3((unsigned)((std::max<size_t>(n2, 1) + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const double2 *>(d_src), reinterpret_cast<double2 *>(mapped_dst), n2, odd ? d_src + n - 1 : nullptr,
                       odd ? mapped_dst + n - 1 : nullptr, ctx->d_status, flags_mapped);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_build_raytri(msm_ctx *ctx, const TriRec *d_rec, const float4 *d_edge, int T, const double *d_feat1, int D, float4 *d_out) {
    if (T <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_build_raytri, dim3((T + 255) / 256), dim3(256), 0, ctx->stream, d_rec, d_edge, T, d_feat1, D, d_out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// msm_query_triangles (Resampler::get_barycentric_weights, R/resampler.cpp:142-167): G lanes per query point
// (search_device.hpp: group_search).  The lane whose exact test finds the containing triangle has the record and
// the projected point in registers and computes the weights right there (calc_barycentric_weights' projection IS
// the inside test's: the record's plane is plane_of(v0, v1, v2) in the same arithmetic).
// ------------------------------------------------------------------------------------------------
struct QueryPayload {
    int mode;
    int id0, id1, id2;
    double wa, wb, wc;
    __device__ __forceinline__ void compute(const TriRec &r, const V3 &p, const V3 &mp) {
        id0 = r.id[0], id1 = r.id[1], id2 = r.id[2];
        // calc_barycentric_weights projects the query first (R/triangle.cpp:130); barycentric_interpolation does not
        area_weights(rec_v0(r), rec_v1(r), rec_v2(r), mode == MSM_WEIGHTS_PROJECTED ? mp : p, wa, wb, wc);
    }
};

template <int G>
__global__ __launch_bounds__(256) void k_query(DevTree T, const double *__restrict__ q, int N, int *__restrict__ tri_id,
                                                int *__restrict__ vid, double *__restrict__ w, int mode, int *status) {
    constexpr int per_block = 256 / G;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * per_block; base < N; base += gridDim.x * per_block) {  // block-uniform: every lane takes part in the group's exchanges
        const int i = base + threadIdx.x / G;
        const bool valid = i < N;
        const V3 p = valid ? mk(q[i], q[N + i], q[2 * (size_t)N + i]) : mk(0.0, 0.0, 0.0);
        QueryPayload out;
        out.mode = mode;
        bool owner;
        const int t = group_search<G>(T, valid, p, lane, out, owner);
        if (!owner) continue;
        if (tri_id) tri_id[i] = t;
        if (t < 0) {
            raise_status(status, t);
            if (vid) vid[i] = vid[N + i] = vid[2 * (size_t)N + i] = -1;
            if (w) w[i] = w[N + i] = w[2 * (size_t)N + i] = 0.0;
            continue;
        }
        if (vid) vid[i] = out.id0, vid[N + i] = out.id1, vid[2 * (size_t)N + i] = out.id2;
        if (w) w[i] = out.wa, w[N + i] = out.wb, w[2 * (size_t)N + i] = out.wc;
    }
}

// sphere_project_warp / surface_resample (R/resampler.cpp:284-302,311-328): the query's barycentric weights in its triangle of `from`
// (get_barycentric_weights: a std::map per query -- ascending vertex id, a later duplicate overwrites) applied to the coordinates `to` of the
// same vertices, summed in map order; to_sphere: renormalised to radius 100 (:324-325).  The lane that finds the triangle does all of it.
struct WarpPayload {
    const double *to;  // 3 x V SoA
    int V, to_sphere;
    double x, y, z;
    __device__ __forceinline__ void compute(const TriRec &r, const V3 &, const V3 &mp) {
        double wt[3];
        area_weights(rec_v0(r), rec_v1(r), rec_v2(r), mp, wt[0], wt[1], wt[2]);
        int n = 0, kid[3];
        double kw[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int id = r.id[j];
            int pos = 0;
            while (pos < n && kid[pos] < id) ++pos;
            if (pos < n && kid[pos] == id) {
                kw[pos] = wt[j];
                continue;
            }
            for (int s = n; s > pos; --s) kid[s] = kid[s - 1], kw[s] = kw[s - 1];
            kid[pos] = id, kw[pos] = wt[j];
            ++n;
        }
        V3 p = mk(0.0, 0.0, 0.0);
        for (int j = 0; j < n; ++j) {
            p.x += to[kid[j]] * kw[j];
            p.y += to[(size_t)V + kid[j]] * kw[j];
            p.z += to[2 * (size_t)V + kid[j]] * kw[j];
        }
        if (to_sphere) p = scale(normalized(p), 100);
        x = p.x, y = p.y, z = p.z;
    }
};

// q and out may be the same array (a mesh warped in place): a query's coordinates are read before its result is written, by its own group only
template <int G>
__global__ __launch_bounds__(256) void k_warp(DevTree T, const double *q, int N, const double *__restrict__ to, int V, int to_sphere, double *out, int *status) {
    constexpr int per_block = 256 / G;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * per_block; base < N; base += gridDim.x * per_block) {
        const int i = base + threadIdx.x / G;
        const bool valid = i < N;
        const V3 p = valid ? mk(q[i], q[N + i], q[2 * (size_t)N + i]) : mk(0.0, 0.0, 0.0);
        WarpPayload wp;
        wp.to = to, wp.V = V, wp.to_sphere = to_sphere;
        wp.x = wp.y = wp.z = 0.0;
        bool owner;
        const int t = group_search<G>(T, valid, p, lane, wp, owner);
        if (!owner) continue;
        if (t < 0) {
            raise_status(status, t);
            continue;  // the coordinates stay what they were; the call fails
        }
        out[i] = wp.x, out[N + i] = wp.y, out[2 * (size_t)N + i] = wp.z;
    }
}

// Octree::get_closest_vertex_ID, R/octree.cpp:216-233
struct ClosestVertexPayload {
    int best;
    __device__ __forceinline__ void compute(const TriRec &r, const V3 &p, const V3 &) {
        double dist = DBL_MAX;
        const V3 vv[3] = {rec_v0(r), rec_v1(r), rec_v2(r)};
        const int ids[3] = {r.id[0], r.id[1], r.id[2]};
        best = 0;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double d = norm(sub(p, vv[v]));
            if (d < dist) {
                best = ids[v];
                dist = d;
            }
        }
    }
};

template <int G>
__global__ __launch_bounds__(256) void k_closest_vertex(DevTree T, const double *__restrict__ q, int N, int *__restrict__ out, int *status) {
    constexpr int per_block = 256 / G;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * per_block; base < N; base += gridDim.x * per_block) {
        const int i = base + threadIdx.x / G;
        const bool valid = i < N;
        const V3 p = valid ? mk(q[i], q[N + i], q[2 * (size_t)N + i]) : mk(0.0, 0.0, 0.0);
        ClosestVertexPayload cv;
        cv.best = 0;
        bool owner;
        const int t = group_search<G>(T, valid, p, lane, cv, owner);
        if (!owner) continue;
        if (t < 0) raise_status(status, t);
        out[i] = t < 0 ? t : cv.best;
    }
}

// ------------------------------------------------------------------------------------------------
// smooth_data, R/resampler.cpp:168-230: Gaussian smoothing of the data over the geodesic neighbourhood of each
// vertex.  The reference tests every vertex pair (N^2 = 1.7e9 at ico6) serially per output vertex; here a wavefront
// owns an output vertex and sweeps the unit vectors of all vertices (precomputed once with Point::normalize's
// arithmetic, so the membership test (actual | ref) >= cos(ang) sees the reference's bits), compacts the members in
// ascending order into LDS and then sums weights and features over that list in the reference's order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unit_vectors(const double *__restrict__ xyz, int N, double *__restrict__ unit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const V3 u = normalized(mk(xyz[i], xyz[N + i], xyz[2 * N + i]));
    unit[i] = u.x;
    unit[N + i] = u.y;
    unit[2 * N + i] = u.z;
}

constexpr int kSmoothList = 512;  // neighbours per LDS pass of one wavefront

// A centre's thresholds.  lo / hi: the band around range * MAXSEP inside which the host decides (see above).  dlo2 / dhi2: squared chords that settle a point
// without the square root and the asin() of the reference's test (round 5: those two were most of the kernel -- some 25 chunks of 64 points survive the
// pruning per centre, ten of them with a point in range): arc(d) = 2 R asin(d / 2R) is increasing, so a chord below 2 R sin(lo / 2R) (1 - 1e-9) has its arc below
// lo and one above 2 R sin(hi / 2R) (1 + 1e-9) has it above hi, with nine orders of magnitude between the margin and the rounding of either libm; the points
// in between -- ties, in practice none -- take the reference's arithmetic as before.
struct RangeBand {
    double lo, hi, dlo2, dhi2;
};
__device__ __forceinline__ RangeBand range_band(double thr) {
    RangeBand b;
    const double slack = fabs(thr) * 1e-11;
    b.lo = thr - slack, b.hi = thr + slack;
    b.dlo2 = -1.0, b.dhi2 = INFINITY;  // nothing settled early (a NaN or non-positive threshold, a range beyond a quarter of the sphere's girth)
    const double a = b.lo / (2 * kRad), e = b.hi / (2 * kRad);
    if (a > 0.0 && e < 1.5) {
        const double s = 2 * kRad * sin(a) * (1 - 1e-9), t = 2 * kRad * sin(e) * (1 + 1e-9);
        b.dlo2 = s * s, b.dhi2 = t * t;
    }
    return b;
}
// 0 out, 1 in, 2 undecided (within_controlpt_range, M/DiscreteCostFunction.cpp:102-107)
__device__ __forceinline__ int range_state(const V3 c, const V3 p, const RangeBand &b) {
    const V3 v = sub(c, p);
    const double d2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (d2 < b.dlo2) return 1;
    if (d2 > b.dhi2) return 0;
    const double d = norm(v);
    if (d > b.hi) return 0;  // arc >= chord, so chord > hi is certainly out of range
    const double arc = chord_to_arc(d);
    return (arc < b.lo) ? 1 : ((arc > b.hi) ? 0 : 2);
}

__global__ __launch_bounds__(256) void k_chunk_bounds(const double *__restrict__ src, int Nsrc, double4 *__restrict__ cb);

__global__ __launch_bounds__(256) void k_smooth(const double *__restrict__ unit, int N, const int *__restrict__ cv, const double *__restrict__ data,
                                                int Vorig, int D, double sigma, double cosang, const double *__restrict__ excl,
                                                double *__restrict__ out, double *__restrict__ excl_out, const double4 *__restrict__ cb, int *status) {
    __shared__ int s_n[4][kSmoothList];
    __shared__ double s_w[4][kSmoothList];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    int *ln = s_n[wave];
    double *lw = s_w[wave];
    const int c = cv[i];
    if (c < 0 || c >= N) {  // the reference indexes sphLow with the id found in orig's octree (:185)
        if (lane == 0) raise_status(status, c < 0 ? c : MSM_ERR_INVALID);
        for (int d = lane; d < D; d += 64) out[(size_t)d * N + i] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    if (lane == 0 && excl_out) excl_out[i] = 0.0;
    for (int d = lane; d < D; d += 64) out[(size_t)d * N + i] = 0.0;
    if (excl && !(excl[c] > 0)) return;  // :200: excluded centre: data stays 0
    const V3 ref = mk(unit[c], unit[N + c], unit[2 * N + c]);
    const double gain = 1 / sqrt(2 * M_PI * sigma * sigma);
    double SUM = 0.0, excl_sum = 0.0;
    double acc = 0.0;  // lane d < D accumulates feature d (more than 64 features: extra passes below)
    int count = 0;     // members in the LDS list (wavefront-uniform)
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wavefront's LDS writes have landed
        // weights in list order on every lane (same values everywhere), then the features
        for (int k = 0; k < count; ++k) {
            double w = lw[k];
            excl_sum += w;
            if (excl) w = excl[ln[k]] * w;
            SUM += w;
            for (int d = lane; d < D; d += 64) {
                const double add = data[(size_t)d * Vorig + ln[k]] * w;
                if (d < 64) acc += add;
                else out[(size_t)d * N + i] += add;  // rare: more than 64 features
            }
        }
        count = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // The sweep is pruned without changing what it finds or the order it finds it in: the unit vectors are taken 64 consecutive ids at a
    // time with a bounding ball (k_chunk_bounds); (a | ref) <= (centre | ref) + radius for every member a, so a chunk whose bound stays
    // below cos(ang) holds no member and is skipped (a NaN anywhere keeps the chunk).  The lanes test 64 chunks at once; the surviving
    // ones -- a few per cent on ico-derived meshes, whose numbering keeps neighbours close -- are swept in ascending order as before.
    const int nchunks = (N + 63) >> 6;
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
        bool cand = false;
        if (c0 + lane < nchunks) {
            const double4 b = cb[c0 + lane];
            cand = !(b.x * ref.x + b.y * ref.y + b.z * ref.z + b.w < cosang);
        }
        unsigned long long todo = __ballot(cand);
        while (todo) {
            const int n0 = (c0 + __ffsll((long long)todo) - 1) << 6;
            todo &= todo - 1ull;
            const int n = n0 + lane;
            bool in = false;
            double chord = 0.0;
            if (n < N) {
                const V3 a = mk(unit[n], unit[N + n], unit[2 * N + n]);
                in = dot(a, ref) >= cosang;
                if (in) chord = norm(sub(ref, a));
            }
            const unsigned long long bal = __ballot(in);
            if (!bal) continue;
            const int add = __popcll(bal);
            if (count + add > kSmoothList) flush();
            if (in) {
                const double g = 2 * kRad * asin(chord / (2 * kRad));
                const int at = count + __popcll(bal & ((1ull << lane) - 1ull));
                ln[at] = n;
                lw[at] = gain * exp(-(g * g) / (2 * sigma * sigma));
            }
            count += add;
        }
    }
    flush();
    if (lane == 0 && excl && excl_out && excl_sum != 0.0) excl_out[i] = SUM / excl_sum;
    for (int d = lane; d < D; d += 64) {
        double v = d < 64 ? acc : out[(size_t)d * N + i];
        if (SUM != 0.0) v /= SUM;
        out[(size_t)d * N + i] = v;
    }
}

int launch_smooth(msm_ctx *ctx, const double *d_xyz, int N, double *d_unit, const int *d_cv, const double *d_data, int Vorig, int D, double sigma,
                  double cosang, const double *d_excl, double *d_out, double *d_excl_out) {
    if (N <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_unit_vectors, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, N, d_unit);
    MSM_HIP(hipGetLastError());
    // the chunks' bounding balls live behind the unit vectors in the caller's buffer (3 N doubles rounded up to a 32-byte boundary + 4 per chunk)
    double4 *cb = reinterpret_cast<double4 *>(d_unit + smooth_bounds_offset(N));
    hipLaunchKernelGGL(k_chunk_bounds, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, d_unit, N, cb);
    MSM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_smooth, dim3((N + 3) / 4), dim3(256), 0, ctx->stream, d_unit, N, d_cv, d_data, Vorig, D, sigma, cosang, d_excl, d_out,
                       d_excl_out, cb, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// get_source_data range test (within_controlpt_range, M/DiscreteCostFunction.cpp:102-107): one
// workgroup per control point sweeps all source vertices and writes the in-range ones, in ascending
// order, to the control point's slot.  The geodesic distance needs asin(); device and host libm may
// differ in the last bit, and this workload has exact ties (a control point's farthest neighbour
// sits exactly at range*MAXSEP), so entries within 1e-11 of the threshold are only flagged (bit 31)
// and the host decides them with its own libm, like the reference would.
//
// The sweep is pruned without changing what it finds: the source vertices are taken in chunks of 64 consecutive ids, each with a
// bounding ball (k_chunk_bounds); a chunk whose ball lies farther from the control point than the threshold chord cannot hold
// an in-range vertex (arc >= chord >= distance to the ball) and is skipped.  Ico-derived meshes number neighbouring vertices
// close together, so a control point looks at a few per cent of the chunks.  The surviving chunks are visited in ascending
// order, four at a time (one per wavefront), which keeps the slot order ascending.
__global__ __launch_bounds__(256) void k_chunk_bounds(const double *__restrict__ src, int Nsrc, double4 *__restrict__ cb) {
    const int lane = threadIdx.x & 63, ch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nchunks = (Nsrc + 63) >> 6;
    if (ch >= nchunks) return;
    const int i = ch * 64 + lane;
    const bool in = i < Nsrc;
    const V3 p = in ? mk(src[i], src[Nsrc + i], src[2 * (size_t)Nsrc + i]) : mk(0, 0, 0);
    const double n = (double)min(64, Nsrc - ch * 64);
    const V3 c = mk(wave_sum(p.x) / n, wave_sum(p.y) / n, wave_sum(p.z) / n);
    double r = in ? norm(sub(p, c)) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(r, off, 64);
        r = (o > r || o != o) ? o : r;  // a NaN coordinate makes the whole chunk a candidate
    }
    if (lane == 0) cb[ch] = make_double4(c.x, c.y, c.z, r * (1 + 1e-12) + 1e-12);
}

__global__ __launch_bounds__(256) void k_range(const double *__restrict__ cp, int Ncp, const double *__restrict__ src, int Nsrc,
                                                const double4 *__restrict__ cb, const double *__restrict__ maxsep, double range, int cap,
                                                uint32_t *__restrict__ slots, int *__restrict__ counts, int *__restrict__ nflag) {
    const int k = blockIdx.x;
    if (k >= Ncp) return;
    const V3 c = mk(cp[k], cp[Ncp + k], cp[2 * Ncp + k]);
    const RangeBand band = range_band(range * maxsep[k]);
    const double reach = band.hi * (1 + 1e-12) + 1e-12;  // a chunk is skipped only if even its ball's nearest point is beyond hi
    const int nchunks = (Nsrc + 63) >> 6;
    constexpr int kTile = 1024;
    __shared__ int cand[kTile];
    __shared__ int wave_count[4];
    __shared__ int running, ncand;
    if (threadIdx.x == 0) running = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tile = 0; tile < nchunks; tile += kTile) {
        __syncthreads();
        if (threadIdx.x == 0) ncand = 0;
        __syncthreads();
        for (int part = 0; part < kTile && tile + part < nchunks; part += 256) {  // uniform
            const int ch = tile + part + threadIdx.x;
            bool keep = false;
            if (ch < nchunks) {
                const double4 b = cb[ch];
                keep = !(norm(sub(c, mk(b.x, b.y, b.z))) - b.w > reach);
            }
            const unsigned long long ball = __ballot(keep);
            if (lane == 0) wave_count[wave] = __popcll(ball);
            __syncthreads();
            if (keep) {
                int pos = ncand + __popcll(ball & ((1ull << lane) - 1));
                for (int wv = 0; wv < wave; ++wv) pos += wave_count[wv];
                cand[pos] = ch;
            }
            __syncthreads();
            if (threadIdx.x == 0) ncand += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            __syncthreads();
        }
        const int nc = ncand;
        for (int it = 0; it < nc; it += 4) {  // uniform
            const int ci = it + wave;
            const int i = ci < nc ? cand[ci] * 64 + lane : Nsrc;
            const int state = i < Nsrc ? range_state(c, mk(src[i], src[Nsrc + i], src[2 * (size_t)Nsrc + i]), band) : 0;
            const unsigned long long ball = __ballot(state != 0);
            if (lane == 0) wave_count[wave] = __popcll(ball);
            __syncthreads();
            if (state != 0) {
                int pos = running + __popcll(ball & ((1ull << lane) - 1));
                for (int wv = 0; wv < wave; ++wv) pos += wave_count[wv];
                if (pos < cap) slots[(size_t)k * cap + pos] = (uint32_t)i | (state == 2 ? 0x80000000u : 0u);
                if (state == 2) atomicAdd(nflag, 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) running += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            __syncthreads();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[k] = running;
}

__device__ __forceinline__ int grid_cell(double x, const RangeGrid &g) {
    const double f = floor((x - g.origin) * g.inv_h);  // monotone in x: a point within R of m lies in a cell between those of m - R and m + R
    return f < 0.0 ? 0 : (f >= (double)g.G ? g.G - 1 : (int)f);
}
__global__ __launch_bounds__(256) void k_range_grid_count(const double *__restrict__ src, int Nsrc, RangeGrid g, int *__restrict__ count, int *__restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nsrc) return;
    const double x = src[i], y = src[Nsrc + i], z = src[2 * (size_t)Nsrc + i];
    if (!(fabs(x) <= 1e300 && fabs(y) <= 1e300 && fabs(z) <= 1e300)) {  // NaN or infinite: such a vertex is a candidate of every centre (k_range flags it)
        *bad = 1;
        return;
    }
    atomicAdd(&count[(grid_cell(x, g) * g.G + grid_cell(y, g)) * g.G + grid_cell(z, g)], 1);
}
__global__ __launch_bounds__(256) void k_range_grid_fill(const double *__restrict__ src, int Nsrc, RangeGrid g, int *__restrict__ cursor, int *__restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nsrc || *g.bad) return;
    const int cell = (grid_cell(src[i], g) * g.G + grid_cell(src[Nsrc + i], g)) * g.G + grid_cell(src[2 * (size_t)Nsrc + i], g);
    ids[g.start[cell] + atomicAdd(&cursor[cell], 1)] = i;
}

// The same test for centres that come in clusters (gMSM: the L candidate positions of a control point, consecutive in the list and within a label
// radius of one another).  k_range spends most of its time on what the cluster has in common: every centre's workgroup tests all the chunks' balls
// (641 of them on an ico6 template, for a patch that touches ten) and takes three block barriers per four chunks.  Here a workgroup takes a cluster of
// up to 64 consecutive centres: the balls are tested ONCE against the cluster's own bounding ball, the survivors (ascending) are kept in LDS, and each
// wavefront then owns centres -- for a centre it tests the listed balls 64 at a time (a lane each) and sweeps the chunks that pass, its running count in
// a register: no barrier after the list is made.  Pruning is conservative at both stages (a chunk is dropped only if no point of its ball can be in
// range), so rows, order and flags are k_range's.
__global__ __launch_bounds__(256) void k_range_cluster(const double *__restrict__ cp, int Ncp, int cluster, const double *__restrict__ src, int Nsrc,
                                                        const double4 *__restrict__ cb, const double *__restrict__ maxsep, double range, int cap,
                                                        uint32_t *__restrict__ slots, int *__restrict__ counts, int *__restrict__ nflag, RangeGrid grid) {
    const int k0 = blockIdx.x * cluster, n = min(cluster, Ncp - k0);  // cluster <= 64
    if (n <= 0) return;
    constexpr int kTile = 1024;
    __shared__ int cand[kTile];
    __shared__ double s_px[kTile], s_py[kTile], s_pz[kTile];  // the grid path: the candidates' coordinates
    __shared__ int s_n;
    __shared__ int wave_count[4];
    __shared__ int ncand;
    __shared__ int s_run[64];
    __shared__ double s_ball[4];  // the cluster's bounding ball: centre, radius + the largest reach of its centres
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const bool in = lane < n;
        const V3 c = in ? mk(cp[k0 + lane], cp[Ncp + k0 + lane], cp[2 * (size_t)Ncp + k0 + lane]) : mk(0, 0, 0);
        const double thr0 = in ? range * maxsep[k0 + lane] : 0.0, hi0 = thr0 + fabs(thr0) * 1e-11;  // as below, per centre
        const double inv = 1.0 / (double)n;
        const V3 m = mk(wave_sum(c.x) * inv, wave_sum(c.y) * inv, wave_sum(c.z) * inv);
        double r = in ? norm(sub(c, m)) : 0.0, reach = in ? hi0 * (1 + 1e-12) + 1e-12 : -1.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(r, off, 64), q = __shfl_xor(reach, off, 64);
            r = (o > r || o != o) ? o : r;  // a NaN centre makes every chunk a candidate (the comparisons below are all false)
            reach = (q > reach || q != q) ? q : reach;
        }
        if (lane == 0) s_ball[0] = m.x, s_ball[1] = m.y, s_ball[2] = m.z, s_ball[3] = (r + reach) * (1 + 1e-12) + 1e-12;
        s_run[lane] = 0;
    }
    __syncthreads();
    const V3 bc = mk(s_ball[0], s_ball[1], s_ball[2]);
    const double breach = s_ball[3];
    // With a grid over the source vertices (round 5): consecutive ids are NOT neighbours on an icosphere -- the ball of 64 consecutive vertices of an ico6
    // mesh has a radius of 30 on a sphere of radius 100, 66 of 641 chunks pass a centre's test and 4 200 points are looked at for a patch of 60 -- so the
    // candidates come from the cells the cluster's ball touches (some 350 vertices), are sorted by id once for the cluster and tested by every centre
    // from LDS.  Same tests on the same points in the same order: rows, order and flags are the sweep's, which stays for what does not fit (more than
    // kTile candidates, a grid with a non-finite vertex, a NaN centre).
    if (grid.start && !*grid.bad && fabs(bc.x) <= 1e300 && fabs(bc.y) <= 1e300 && fabs(bc.z) <= 1e300 && breach >= 0.0 && breach <= 1e300) {  // uniform
        const int G = grid.G;
        const int x0 = grid_cell(bc.x - breach, grid), x1 = grid_cell(bc.x + breach, grid), y0 = grid_cell(bc.y - breach, grid), y1 = grid_cell(bc.y + breach, grid);
        const int z0 = grid_cell(bc.z - breach, grid), z1 = grid_cell(bc.z + breach, grid);
        const int ny = y1 - y0 + 1, nz = z1 - z0 + 1, ncell = (x1 - x0 + 1) * ny * nz;
        if (ncell <= 4096) {
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
            for (int t = threadIdx.x; t < ncell; t += 256) {
                const int cell = ((x0 + t / (ny * nz)) * G + y0 + (t / nz) % ny) * G + z0 + t % nz;
                const int b = grid.start[cell], len = grid.start[cell + 1] - b;
                if (len > 0) {
                    const int at = atomicAdd(&s_n, len);
                    for (int j = 0; j < len && at + j < kTile; ++j) cand[at + j] = grid.ids[b + j];
                }
            }
            __syncthreads();
            const int nc = s_n;
            if (nc <= kTile) {  // uniform
                int P = 64;
                while (P < nc) P <<= 1;
                for (int t = nc + threadIdx.x; t < P; t += 256) cand[t] = INT_MAX;
                __syncthreads();
                for (int kk = 2; kk <= P; kk <<= 1)  // ascending ids (bitonic: the cells come in no order)
                    for (int j = kk >> 1; j > 0; j >>= 1) {
                        for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                            const int i = 2 * t - (t & (j - 1)), l = i + j;
                            const int a = cand[i], b = cand[l];
                            if ((a > b) == ((i & kk) == 0)) cand[i] = b, cand[l] = a;
                        }
                        __syncthreads();
                    }
                for (int t = threadIdx.x; t < nc; t += 256) {
                    const int i = cand[t];
                    s_px[t] = src[i], s_py[t] = src[Nsrc + i], s_pz[t] = src[2 * (size_t)Nsrc + i];
                }
                __syncthreads();
                for (int t = wave; t < n; t += 4) {  // wavefront-uniform: this wavefront's centres
                    const int k = k0 + t;
                    const V3 c = mk(cp[k], cp[Ncp + k], cp[2 * (size_t)Ncp + k]);
                    const RangeBand band = range_band(range * maxsep[k]);
                    int run = 0;
                    for (int base = 0; base < nc; base += 64) {  // uniform
                        const int ci = base + lane;
                        const int i = ci < nc ? cand[ci] : 0;
                        const int state = ci < nc ? range_state(c, mk(s_px[ci], s_py[ci], s_pz[ci]), band) : 0;
                        const unsigned long long hit = __ballot(state != 0);
                        if (state != 0) {
                            const int pos = run + __popcll(hit & ((1ull << lane) - 1));
                            if (pos < cap) slots[(size_t)k * cap + pos] = (uint32_t)i | (state == 2 ? 0x80000000u : 0u);
                            if (state == 2) atomicAdd(nflag, 1);
                        }
                        run += __popcll(hit);
                    }
                    if (lane == 0) counts[k] = run;
                }
                return;
            }
        }
    }
    const int nchunks = (Nsrc + 63) >> 6;
    for (int tile = 0; tile < nchunks; tile += kTile) {
        __syncthreads();  // the previous tile's list is no longer read
        if (threadIdx.x == 0) ncand = 0;
        __syncthreads();
        for (int part = 0; part < kTile && tile + part < nchunks; part += 256) {  // uniform
            const int ch = tile + part + threadIdx.x;
            bool keep = false;
            if (ch < nchunks) {
                const double4 b = cb[ch];
                keep = !(norm(sub(bc, mk(b.x, b.y, b.z))) - b.w > breach);
            }
            const unsigned long long ball = __ballot(keep);
            if (lane == 0) wave_count[wave] = __popcll(ball);
            __syncthreads();
            if (keep) {
                int pos = ncand + __popcll(ball & ((1ull << lane) - 1));
                for (int wv = 0; wv < wave; ++wv) pos += wave_count[wv];
                cand[pos] = ch;
            }
            __syncthreads();
            if (threadIdx.x == 0) ncand += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            __syncthreads();
        }
        const int nc = ncand;
        for (int t = wave; t < n; t += 4) {  // wavefront-uniform: this wavefront's centres
            const int k = k0 + t;
            const V3 c = mk(cp[k], cp[Ncp + k], cp[2 * (size_t)Ncp + k]);
            const RangeBand band = range_band(range * maxsep[k]);
            const double reach = band.hi * (1 + 1e-12) + 1e-12;
            int run = s_run[t];
            for (int base = 0; base < nc; base += 64) {  // uniform
                const int ci = base + lane;
                bool keep = false;
                int mych = 0;
                if (ci < nc) {
                    mych = cand[ci];
                    const double4 b = cb[mych];
                    keep = !(norm(sub(c, mk(b.x, b.y, b.z))) - b.w > reach);
                }
                unsigned long long todo = __ballot(keep);
                while (todo) {  // uniform: the chunks that pass, ascending, four at a time (their twelve loads in flight together: one chunk per round was a chain
                                // of dependent loads, 25 rounds per centre)
                    constexpr int kU = 4;
                    int i[kU];
                    V3 q[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const bool any = todo != 0;
                        const int j = any ? __ffsll((long long)todo) - 1 : 0;
                        if (any) todo &= todo - 1;
                        i[u] = any ? __shfl(mych, j, 64) * 64 + lane : Nsrc;
                        q[u] = i[u] < Nsrc ? mk(src[i[u]], src[Nsrc + i[u]], src[2 * (size_t)Nsrc + i[u]]) : mk(0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int state = i[u] < Nsrc ? range_state(c, q[u], band) : 0;
                        const unsigned long long hit = __ballot(state != 0);
                        if (state != 0) {
                            const int pos = run + __popcll(hit & ((1ull << lane) - 1));
                            if (pos < cap) slots[(size_t)k * cap + pos] = (uint32_t)i[u] | (state == 2 ? 0x80000000u : 0u);
                            if (state == 2) atomicAdd(nflag, 1);
                        }
                        run += __popcll(hit);
                    }
                }
            }
            if (lane == 0) s_run[t] = run;  // only this wavefront reads it again (next tile)
            if (tile + kTile >= nchunks && lane == 0) counts[k] = run;
        }
    }
}

// slot rows -> one contiguous list (CSR offsets pptr already scanned from the counts); only used when no entry was flagged
// (pidx_cap: the list's capacity -- a caller that compacts before it has looked at the counts sizes the list from the previous subject's and
// checks afterwards; rows beyond `cap` entries were not written by k_range either)
__global__ __launch_bounds__(256) void k_patch_compact(const uint32_t *__restrict__ slots, int cap, const int32_t *__restrict__ pptr, int M,
                                                        int32_t *__restrict__ pidx, size_t pidx_cap) {
    const int lane = threadIdx.x & 63, k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= M) return;
    const int b = pptr[k], n = min(pptr[k + 1] - b, cap);
    for (int j = lane; j < n; j += 64)
        if ((size_t)(b + j) < pidx_cap) pidx[b + j] = (int32_t)(slots[(size_t)k * cap + j] & 0x7fffffffu);
}

// ------------------------------------------------------------------------------------------------
// The rotation every (control point, label) evaluation starts with:
// estimate_rotation_matrix(_CPgrid[node], ROTATIONS[node] * labels[label]), M/DiscreteCostFunction.cpp:380.
// Kept out of the table kernels: acos/sincos would otherwise set their register budget.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_label_rotations(const double *__restrict__ cp, int N, const double *__restrict__ rot,
                                                          const double *__restrict__ labels, int L, double *__restrict__ rnl,
                                                          double *__restrict__ moved_out, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int node = i / L, l = i - node * L;
    const V3 lab = mk(labels[l], labels[L + l], labels[2 * L + l]);
    const V3 moved = rotate(rot + 9 * (size_t)node, lab);
    if (moved_out) {
        moved_out[3 * (size_t)i] = moved.x;
        moved_out[3 * (size_t)i + 1] = moved.y;
        moved_out[3 * (size_t)i + 2] = moved.z;
    }
    double R[9];
    if (!rotation_matrix(mk(cp[node], cp[N + node], cp[2 * N + node]), moved, R)) raise_status(status, MSM_ERR_ROTATION);
#pragma unroll
    for (int k = 0; k < 9; ++k) rnl[9 * (size_t)i + k] = R[k];
}

int launch_label_rotations(msm_ctx *ctx, const double *d_cp, int N, const double *d_rot, const double *d_labels, int L, double *d_rnl,
                           double *d_moved) {
    hipLaunchKernelGGL(k_label_rotations, dim3((N * L + 255) / 256), dim3(256), 0, ctx->stream, d_cp, N, d_rot, d_labels, L, d_rnl, d_moved,
                       ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// check_for_intersections, M/reg_tools.cpp:118-129: a vertex is folded when the normal of its first triangle and the
// normal of any of its triangles have a dot product <= 0.5.  One thread per vertex; the normals are recomputed per
// vertex (Triangle::normal, same operation order as the host pass that moves the folded vertices).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_detect(const double *__restrict__ xyz, int V, const int32_t *__restrict__ tri, int T,
                                                      const int32_t *__restrict__ tid_ptr, const int32_t *__restrict__ tid, int32_t *__restrict__ fold) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    auto normal_of = [&](int t) {
        const int a = tri[t], b = tri[T + t], c = tri[2 * T + t];
        return tri_normal(mk(xyz[a], xyz[V + a], xyz[2 * V + a]), mk(xyz[b], xyz[V + b], xyz[2 * V + b]), mk(xyz[c], xyz[V + c], xyz[2 * V + c]));
    };
    const int b = tid_ptr[i], e = tid_ptr[i + 1];
    int folded = 0;
    if (b == e) {
        atomicAdd(fold + 1, 1);
    } else {
        const V3 n0 = normal_of(tid[b]);
        for (int k = b; k < e; ++k) folded |= dot(n0, normal_of(tid[k])) <= 0.5;
        if (folded) atomicAdd(fold, 1);
    }
    fold[2 + i] = folded;
}

int launch_fold_detect(msm_ctx *ctx, const double *d_xyz, int V, const int32_t *d_tri, int T, const int32_t *d_tid_ptr, const int32_t *d_tid, int32_t *d_fold) {
    MSM_HIP(hipMemsetAsync(d_fold, 0, 2 * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_fold_detect, dim3((V + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, V, d_tri, T, d_tid_ptr, d_tid, d_fold);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline int grid_for(int n, int block, int cap) {
    int g = (n + block - 1) / block;
    return g < 1 ? 1 : (g > cap ? cap : g);
}

// lanes per query of the search kernels (search_device.hpp: group_search).  Eight while all wavefronts of the launch are resident at once
// (4 per SIMD at the kernels' 124 registers: 32 768 queries) -- the launch is then one dependent chain, which eight lanes keep shortest
// (2 562 queries: 7.4 us against 9.2 with four) --, four beyond that, where instruction issue counts as well (40 962 queries on an ico6
// tree: 13.3 us against 17.2).  MSMHIP_QUERY_LANES=4|8 forces one.
// The same queries against a target that has a direction table (a simple surface: every regular icosphere, a group's template): a lane per query looks its
// direction's cell up, tests the (at most seven) candidate triangles' edge planes in float and, when the table vouches for the accepted one (ray_find: the
// triangle the reference's search returns, provably), computes the weights from the triangle's record exactly as the searching lane of k_query does -- a
// chain of two dependent loads instead of six.  What the table cannot settle (a point within the float test's allowance of an edge, a triangle with
// exclusion boxes: a fraction of a percent) is listed and searched by k_query_open, eight lanes per query.  Same triangles, same weights, bit for bit
// (tests/test_gpu_search.py: test_query_through_the_direction_table).  Used where one target serves very many queries: the reverse queries of gMSM's
// get_patch_data (L x V rotated vertices against the template per subject, M/DiscreteGroupModel.cpp:105 -> R/resampler.cpp:77).
__global__ __launch_bounds__(256) void k_query_rays(DevTree T, const double *__restrict__ q, int N, int *__restrict__ tri_id, int *__restrict__ vid,
                                                     double *__restrict__ w, int mode, int *__restrict__ open_list, int *__restrict__ open_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const V3 p = mk(q[i], q[N + i], q[2 * (size_t)N + i]);
    const int t = ray_find(T, p);
    if (t < 0) {
        open_list[atomicAdd(open_count, 1)] = i;
        return;
    }
    const TriRec &r = T.rec[t];
    V3 mp;
    (void)inside_test(r, p, mp);  // the projection onto the triangle's plane (the table has vouched for the test's outcome)
    QueryPayload out;
    out.mode = mode;
    out.compute(r, p, mp);
    if (tri_id) tri_id[i] = t;
    if (vid) vid[i] = out.id0, vid[N + i] = out.id1, vid[2 * (size_t)N + i] = out.id2;
    if (w) w[i] = out.wa, w[N + i] = out.wb, w[2 * (size_t)N + i] = out.wc;
}

// the listed queries through the complete search (k_query's body at the listed indices; the list's order does not matter: a query's result is its own)
template <int G>
__global__ __launch_bounds__(256) void k_query_open(DevTree T, const double *__restrict__ q, int N, const int *__restrict__ open_list, const int *__restrict__ open_count,
                                                     int *__restrict__ tri_id, int *__restrict__ vid, double *__restrict__ w, int mode, int *status) {
    constexpr int per_block = 256 / G;
    const int lane = threadIdx.x & 63, n = *open_count;
    for (int base = blockIdx.x * per_block; base < n; base += gridDim.x * per_block) {  // block-uniform
        const int j = base + threadIdx.x / G;
        const bool valid = j < n;
        const int i = valid ? open_list[j] : 0;
        const V3 p = valid ? mk(q[i], q[N + i], q[2 * (size_t)N + i]) : mk(0.0, 0.0, 0.0);
        QueryPayload out;
        out.mode = mode;
        bool owner;
        const int t = group_search<G>(T, valid, p, lane, out, owner);
        if (!owner) continue;
        if (tri_id) tri_id[i] = t;
        if (t < 0) {
            raise_status(status, t);
            if (vid) vid[i] = vid[N + i] = vid[2 * (size_t)N + i] = -1;
            if (w) w[i] = w[N + i] = w[2 * (size_t)N + i] = 0.0;
            continue;
        }
        if (vid) vid[i] = out.id0, vid[N + i] = out.id1, vid[2 * (size_t)N + i] = out.id2;
        if (w) w[i] = out.wa, w[N + i] = out.wb, w[2 * (size_t)N + i] = out.wc;
    }
}

int query_lanes(long long N) {
    static const int forced = [] {
        const char *e = std::getenv("MSMHIP_QUERY_LANES");
        return e ? std::atoi(e) : 0;
    }();
    if (forced == 4 || forced == 8) return forced;
    return N <= 32768 ? 8 : 4;
}

int launch_query(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode) {
    if (N <= 0) return MSM_OK;
    if (query_lanes(N) == 4)
        hipLaunchKernelGGL(k_query<4>, dim3(grid_for(N, 64, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_tri, d_vid, d_w, mode, ctx->d_status);
    else
        hipLaunchKernelGGL(k_query<8>, dim3(grid_for(N, 32, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_tri, d_vid, d_w, mode, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// d_open: N + 1 ints of scratch (the list of unsettled queries and, at d_open[N], their count); a target without a table: the complete search
int launch_query_rays(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode, int *d_open) {
    if (N <= 0) return MSM_OK;
    static const bool off = [] { const char *e = std::getenv("MSMHIP_QUERY_RAYS"); return e && std::strcmp(e, "off") == 0; }();
    if (T.ray_G <= 0 || !d_open || off) return launch_query(ctx, T, d_q, N, d_tri, d_vid, d_w, mode);
    MSM_HIP(hipMemsetAsync(d_open + N, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_query_rays, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, T, d_q, N, d_tri, d_vid, d_w, mode, d_open, d_open + N);
    hipLaunchKernelGGL(k_query_open<8>, dim3(256), dim3(256), 0, ctx->stream, T, d_q, N, d_open, d_open + N, d_tri, d_vid, d_w, mode, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_warp(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, const double *d_to, int V, bool to_sphere, double *d_out) {
    if (N <= 0) return MSM_OK;
    if (query_lanes(N) == 4)
        hipLaunchKernelGGL(k_warp<4>, dim3(grid_for(N, 64, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_to, V, to_sphere ? 1 : 0, d_out, ctx->d_status);
    else
        hipLaunchKernelGGL(k_warp<8>, dim3(grid_for(N, 32, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_to, V, to_sphere ? 1 : 0, d_out, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_closest_vertex(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_out) {
    if (N <= 0) return MSM_OK;
    if (query_lanes(N) == 4)
        hipLaunchKernelGGL(k_closest_vertex<4>, dim3(grid_for(N, 64, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_out, ctx->d_status);
    else
        hipLaunchKernelGGL(k_closest_vertex<8>, dim3(grid_for(N, 32, 32768)), dim3(256), 0, ctx->stream, T, d_q, N, d_out, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side order of the points of each patch (unary kernels): Morton order of their positions, ties by id, so that the
// lanes of a wavefront sample neighbouring places of the target.  A wavefront per patch ranks its entries by the 64-bit key
// (code << 32 | id); the keys of a patch are distinct, so an entry's place is the number of smaller keys.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}
__global__ __launch_bounds__(256) void k_morton_codes(const double *__restrict__ xyz, int n, uint32_t *__restrict__ code) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double u = (xyz[(size_t)a * n + v] + kBounds) / (2 * kBounds);
        q[a] = (uint32_t)fmax(0.0, fmin(1023.0, u == u ? u * 1024.0 : 0.0));
    }
    code[v] = spread10(q[0]) << 2 | spread10(q[1]) << 1 | spread10(q[2]);
}
__global__ __launch_bounds__(64) void k_sort_patches(const int32_t *__restrict__ pptr, int ngroups, const int32_t *__restrict__ pidx,
                                                      const uint32_t *__restrict__ code, int32_t *__restrict__ sorted) {
    const int g = blockIdx.x;
    if (g >= ngroups) return;
    const int b = pptr[g], len = pptr[g + 1] - b;
    for (int i = threadIdx.x; i < len; i += 64) {
        const int32_t id = pidx[b + i];
        const unsigned long long mine = (unsigned long long)code[id] << 32 | (uint32_t)id;
        int rank = 0;
        for (int j = 0; j < len; ++j) {
            const int32_t o = pidx[b + j];
            rank += ((unsigned long long)code[o] << 32 | (uint32_t)o) < mine;
        }
        sorted[b + rank] = id;
    }
}

int launch_sort_patches(msm_ctx *ctx, const double *d_xyz, int n, const int32_t *d_pptr, int ngroups, const int32_t *d_pidx, uint32_t *d_code,
                        int32_t *d_sorted) {
    if (n <= 0 || ngroups <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_morton_codes, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, n, d_code);
    hipLaunchKernelGGL(k_sort_patches, dim3(ngroups), dim3(64), 0, ctx->stream, d_pptr, ngroups, d_pidx, d_code, d_sorted);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_range_grid_build(msm_ctx *ctx, const double *d_src, int Nsrc, int G, double origin, double inv_h, int *d_start, int *d_cursor, int *d_ids, int *d_bad, int *d_tmp) {
    if (Nsrc <= 0 || G <= 0) return MSM_OK;
    const size_t cells = (size_t)G * G * G;
    MSM_HIP(hipMemsetAsync(d_start, 0, sizeof(int) * (cells + 1), ctx->stream));
    MSM_HIP(hipMemsetAsync(d_cursor, 0, sizeof(int) * cells, ctx->stream));
    MSM_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
    RangeGrid g;
    g.start = d_start, g.ids = d_ids, g.bad = d_bad, g.G = G, g.origin = origin, g.inv_h = inv_h;
    hipLaunchKernelGGL(k_range_grid_count, dim3((Nsrc + 255) / 256), dim3(256), 0, ctx->stream, d_src, Nsrc, g, d_start, d_bad);
    MSM_HIP(hipGetLastError());
    int st = launch_scan_exclusive(ctx, d_start, (int)cells, d_tmp);
    if (st) return st;
    hipLaunchKernelGGL(k_range_grid_fill, dim3((Nsrc + 255) / 256), dim3(256), 0, ctx->stream, d_src, Nsrc, g, d_cursor, d_ids);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_range(msm_ctx *ctx, const double *d_cp, int Ncp, const double *d_src, int Nsrc, const double *d_maxsep, double range,
                 int cap, uint32_t *d_slots, int *d_counts, double4 *d_chunk_bounds, int *d_nflag, int cluster, const RangeGrid *grid) {
    if (Ncp <= 0) return MSM_OK;
    const int nchunks = (Nsrc + 63) / 64;
    MSM_HIP(hipMemsetAsync(d_nflag, 0, sizeof(int), ctx->stream));
    if (nchunks > 0) hipLaunchKernelGGL(k_chunk_bounds, dim3((nchunks + 3) / 4), dim3(256), 0, ctx->stream, d_src, Nsrc, d_chunk_bounds);
    static const bool cluster_off = [] { const char *e = std::getenv("MSMHIP_RANGE_CLUSTER"); return e && std::strcmp(e, "off") == 0; }();
    if (cluster > 1 && nchunks > 0 && !cluster_off) {
        const int per = cluster <= 64 ? cluster : 32;  // centres that are consecutive in the list are neighbours either way
        static const bool grid_off = [] { const char *e = std::getenv("MSMHIP_RANGE_GRID"); return e && std::strcmp(e, "off") == 0; }();
        hipLaunchKernelGGL(k_range_cluster, dim3((Ncp + per - 1) / per), dim3(256), 0, ctx->stream, d_cp, Ncp, per, d_src, Nsrc, d_chunk_bounds, d_maxsep, range, cap, d_slots,
                           d_counts, d_nflag, grid && !grid_off ? *grid : RangeGrid{});
    } else
        hipLaunchKernelGGL(k_range, dim3(Ncp), dim3(256), 0, ctx->stream, d_cp, Ncp, d_src, Nsrc, d_chunk_bounds, d_maxsep, range, cap, d_slots, d_counts, d_nflag);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_patch_compact(msm_ctx *ctx, const uint32_t *d_slots, int cap, const int32_t *d_pptr, int M, int32_t *d_pidx, size_t pidx_cap) {
    if (M <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_patch_compact, dim3((M + 3) / 4), dim3(256), 0, ctx->stream, d_slots, cap, d_pptr, M, d_pidx, pidx_cap);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
