// geom.hpp -- point / triangle geometry shared by host set-up code and the gfx950 kernels.
//
// The functions follow newMSM's arithmetic operation by operation (R/point.cpp, R/triangle.cpp under
// /root/reference/libraries/msm-newresampler/src) and this library is built with -ffp-contract=off,
// so +,-,*,/ and sqrt give the same FP64 bits as the reference's CPU build.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#define MSM_HD __host__ __device__ __forceinline__

namespace msm {

constexpr double kEps = 1e-8;        // EPSILON, R/point.h:31
constexpr double kRad = 100.0;       // RAD, R/point.h:32
constexpr double kBounds = 101.0;    // MESH_BOUNDS, R/octree.h:37
constexpr double kRayFloatAllowance = 3e-6;  // direction table: what the float evaluation of an edge-plane product may be off by (octree.cpp: build_ray_table)
constexpr int kMaxTriangles = 50;    // MAX_TRIANGLES, R/node.h:33

struct V3 {
    double x, y, z;
};

MSM_HD V3 mk(double x, double y, double z) { return V3{x, y, z}; }
MSM_HD V3 sub(const V3 &a, const V3 &b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MSM_HD V3 scale(const V3 &a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
// operator| (dot), R/point.cpp:174-176
MSM_HD double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// operator* (cross), R/point.cpp:178-183: the Y term is written b.x*a.z - b.z*a.x
MSM_HD V3 cross(const V3 &a, const V3 &b) {
    return V3{a.y * b.z - a.z * b.y, b.x * a.z - b.z * a.x, a.x * b.y - b.x * a.y};
}
MSM_HD double norm(const V3 &a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// Point::normalize, R/point.cpp:26-34
MSM_HD V3 normalized(V3 a) {
    double n = norm(a);
    if (n > kEps) {
        a.x /= n;
        a.y /= n;
        a.z /= n;
    }
    return a;
}
// operator*(Matrix, Point), R/point.cpp:207-213 (row-major R)
MSM_HD V3 rotate(const double *R, const V3 &v) {
    return V3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z};
}

// compute_area, R/point.cpp:68-75
MSM_HD double compute_area(const V3 &v0, const V3 &v1, const V3 &v2) {
    return 0.5 * norm(cross(sub(v1, v0), sub(v2, v0)));
}
// Triangle::normal, R/triangle.cpp:45-50
MSM_HD V3 tri_normal(const V3 &v0, const V3 &v1, const V3 &v2) { return normalized(cross(sub(v2, v0), sub(v1, v0))); }
// Triangle::calc_area, R/triangle.cpp:52-55
MSM_HD double tri_area(const V3 &v0, const V3 &v1, const V3 &v2) { return 0.5 * norm(cross(sub(v2, v0), sub(v1, v0))); }

// The per-triangle part of project_point (R/point.cpp:46-60): unit plane normal s3 and s3.v1.
// Depends on the triangle only, so it is computed once per mesh instead of once per test.
MSM_HD void plane_of(const V3 &v1, const V3 &v2, const V3 &v3, V3 &s3, double &d) {
    V3 s1 = normalized(sub(v3, v1));
    V3 s2 = normalized(sub(v2, v1));
    s3 = normalized(cross(s1, s2));
    d = dot(s3, v1);
}
// the per-query part of project_point: vb * ((s3|v1) / (s3|vb))
MSM_HD V3 project_with_plane(const V3 &vb, const V3 &s3, double d) {
    double si = d / dot(s3, vb);
    return scale(vb, si);
}
MSM_HD V3 project_point(const V3 &vb, const V3 &v1, const V3 &v2, const V3 &v3) {
    V3 s3;
    double d;
    plane_of(v1, v2, v3, s3, d);
    return project_with_plane(vb, s3, d);
}

// same_side, R/point.cpp:36-39
MSM_HD bool same_side(const V3 &p1, const V3 &p2, const V3 &a, const V3 &b) {
    V3 ba = sub(b, a);
    return dot(cross(ba, sub(p1, a)), cross(ba, sub(p2, a))) > -kEps;
}
// point_in_triangle, R/point.cpp:41-44
MSM_HD bool point_in_triangle(const V3 &p, const V3 &a, const V3 &b, const V3 &c) {
    return same_side(p, a, b, c) && same_side(p, b, c, a) && same_side(p, c, a, b);
}

// Triangle::dist_to_point, R/triangle.cpp:85-122
MSM_HD double dist_to_point(const V3 &x0, const V3 &x1, const V3 &x2, const V3 &x3) {
    double dmin = DBL_MAX, d;
    V3 u = sub(x2, x1);
    if (dot(sub(x0, x1), u) > 0 && dot(sub(x0, x2), u) < 0) {
        d = norm(cross(sub(x0, x1), sub(x0, x2))) / norm(sub(x2, x1));
        if (d < dmin) dmin = d;
    }
    u = sub(x3, x1);
    if (dot(sub(x0, x1), u) > 0 && dot(sub(x0, x3), u) < 0) {
        d = norm(cross(sub(x0, x1), sub(x0, x3))) / norm(sub(x3, x1));
        if (d < dmin) dmin = d;
    }
    u = sub(x3, x2);
    if (dot(sub(x0, x2), u) > 0 && dot(sub(x0, x3), u) < 0) {
        d = norm(cross(sub(x0, x2), sub(x0, x3))) / norm(sub(x3, x2));
        if (d < dmin) dmin = d;
    }
    d = norm(sub(x0, x1));
    if (d < dmin) dmin = d;
    d = norm(sub(x0, x2));
    if (d < dmin) dmin = d;
    d = norm(sub(x0, x3));
    if (d < dmin) dmin = d;
    return dmin;
}

// the normalised sub-triangle areas used by calc_barycentric_weights (R/triangle.cpp:132-140, given the
// projected point) and by barycentric_interpolation / barycentric (:147-154, given the raw point)
MSM_HD void area_weights(const V3 &v1, const V3 &v2, const V3 &v3, const V3 &p, double &wa, double &wb, double &wc) {
    double Aa = compute_area(p, v2, v3);
    double Ab = compute_area(p, v1, v3);
    double Ac = compute_area(p, v1, v2);
    double A = Aa + Ab + Ac;
    wa = Aa / A;
    wb = Ab / A;
    wc = Ac / A;
}

// estimate_rotation_matrix, R/point.cpp:97-152.  Returns false for the reference's exception.
MSM_HD bool rotation_matrix(V3 ci, V3 index, double *R) {
    ci = normalized(ci);
    index = normalized(index);
    double cd = dot(ci, index);
    double theta = acos(cd);
    if (theta > M_PI) return false;
    V3 cr = normalized(cross(ci, index));
    if (fabs(1 - cd) < kEps) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else if (norm(cr) < kEps) {
        R[0] = -1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = -1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = -1;
    } else {
        const double c[3] = {cr.x, cr.y, cr.z};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (fabs(-1 - cd) < kEps) {
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) R[3 * r + q] = 2 * (c[r] * c[q]) - I[3 * r + q];
        } else {
            const double u[9] = {0, -cr.z, cr.y, cr.z, 0, -cr.x, -cr.y, cr.x, 0};
            // g++ (the reference's compiler) merges sin(theta)/cos(theta) into one sincos() call, whose
            // results can differ from separate sin()/cos() in the last bit; do the same explicitly
            double st, ct;
            sincos(theta, &st, &ct);
            const double omc = 1 - ct;
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) {
                    double s = 0.0;
                    for (int k = 0; k < 3; ++k) s += u[3 * r + k] * u[3 * k + q];
                    R[3 * r + q] = (I[3 * r + q] + u[3 * r + q] * st) + omc * s;
                }
        }
    }
    return true;
}

// geodesic distance from a chord, e.g. R/octree.cpp:200, M/DiscreteCostFunction.cpp:104-106
MSM_HD double chord_to_arc(double chord) { return 2 * kRad * asin(chord / (2 * kRad)); }

}  // namespace msm
