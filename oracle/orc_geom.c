/*
 * orc_geom.c -- oracle restatement of newMSM point / triangle geometry.  TEST INFRASTRUCTURE ONLY
 * (see msm_oracle.h).  Parity unpinned: pinned by structural statistics only.
 *
 * Follows R/point.cpp and R/triangle.cpp of rbesenczi/newMSM operation by operation so that FP64
 * results are bit-identical to the reference when compiled without FMA contraction.
 */
#include "orc_internal.h"

/* Point::normalize, R/point.cpp:26-34: only divides when the norm exceeds EPSILON */
void orc_normalize(double v[3]) {
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n > ORC_EPSILON) {
        v[0] /= n;
        v[1] /= n;
        v[2] /= n;
    }
}

/* same_side, R/point.cpp:36-39 */
int orc_same_side(const double p1[3], const double p2[3], const double a[3], const double b[3]) {
    double ba[3], pa[3], qa[3], c1[3], c2[3];
    v_sub(b, a, ba);
    v_sub(p1, a, pa);
    v_sub(p2, a, qa);
    v_cross(ba, pa, c1);
    v_cross(ba, qa, c2);
    return v_dot(c1, c2) > -ORC_EPSILON;
}

/* point_in_triangle, R/point.cpp:41-44 */
int orc_point_in_triangle(const double p[3], const double a[3], const double b[3], const double c[3]) {
    return orc_same_side(p, a, b, c) && orc_same_side(p, b, c, a) && orc_same_side(p, c, a, b);
}

/* project_point, R/point.cpp:46-60: ray from the origin through vb meets the plane of (v1,v2,v3) */
void orc_project_point(const double vb[3], const double v1[3], const double v2[3], const double v3[3], double out[3]) {
    double s1[3], s2[3], s3[3];
    v_sub(v3, v1, s1);
    orc_normalize(s1);
    v_sub(v2, v1, s2);
    orc_normalize(s2);
    v_cross(s1, s2, s3);
    orc_normalize(s3);
    double si = v_dot(s3, v1) / v_dot(s3, vb);
    out[0] = vb[0] * si;
    out[1] = vb[1] * si;
    out[2] = vb[2] * si;
}

/* compute_area, R/point.cpp:68-75 */
double orc_compute_area(const double v0[3], const double v1[3], const double v2[3]) {
    double a[3], b[3], c[3];
    v_sub(v1, v0, a);
    v_sub(v2, v0, b);
    v_cross(a, b, c);
    return 0.5 * v_norm(c);
}

/* Triangle::normal, R/triangle.cpp:45-50: (v2-v0) x (v1-v0), normalised */
void orc_tri_normal(const double v0[3], const double v1[3], const double v2[3], double out[3]) {
    double a[3], b[3];
    v_sub(v2, v0, a);
    v_sub(v1, v0, b);
    v_cross(a, b, out);
    orc_normalize(out);
}

/* Triangle::calc_area, R/triangle.cpp:52-55 */
double orc_tri_calc_area(const double v0[3], const double v1[3], const double v2[3]) {
    double a[3], b[3], c[3];
    v_sub(v2, v0, a);
    v_sub(v1, v0, b);
    v_cross(a, b, c);
    return 0.5 * v_norm(c);
}

/* one edge term of Triangle::dist_to_point, R/triangle.cpp:97-113 */
static void edge_term(const double x0[3], const double xa[3], const double xb[3], double *dmin) {
    double u[3], pa[3], pb[3], c[3];
    v_sub(xb, xa, u);
    v_sub(x0, xa, pa);
    v_sub(x0, xb, pb);
    if (v_dot(pa, u) > 0 && v_dot(pb, u) < 0) {
        v_cross(pa, pb, c);
        double d = v_norm(c) / v_norm(u);
        if (d < *dmin) *dmin = d;
    }
}

/* Triangle::dist_to_point, R/triangle.cpp:85-122: min distance to the three edges and vertices */
double orc_dist_to_point(const double x0[3], const double x1[3], const double x2[3], const double x3[3]) {
    double dmin = DBL_MAX, d, t[3];
    edge_term(x0, x1, x2, &dmin);
    edge_term(x0, x1, x3, &dmin);
    edge_term(x0, x2, x3, &dmin);
    v_sub(x0, x1, t);
    d = v_norm(t);
    if (d < dmin) dmin = d;
    v_sub(x0, x2, t);
    d = v_norm(t);
    if (d < dmin) dmin = d;
    v_sub(x0, x3, t);
    d = v_norm(t);
    if (d < dmin) dmin = d;
    return dmin;
}

/* calc_barycentric_weights, R/triangle.cpp:124-143 (query is ray-projected first) */
void orc_calc_barycentric_weights(const double v1[3], const double v2[3], const double v3[3], const double vref[3], double w[3]) {
    double pp[3];
    orc_project_point(vref, v1, v2, v3, pp);
    double Aa = orc_compute_area(pp, v2, v3);
    double Ab = orc_compute_area(pp, v1, v3);
    double Ac = orc_compute_area(pp, v1, v2);
    double A = Aa + Ab + Ac;
    w[0] = Aa / A;
    w[1] = Ab / A;
    w[2] = Ac / A;
}

/* barycentric_interpolation, R/triangle.cpp:145-157 (query used as given) */
double orc_barycentric_interpolation(const double v1[3], const double v2[3], const double v3[3], const double vref[3],
                                     double a1, double a2, double a3) {
    double Aa = orc_compute_area(vref, v2, v3);
    double Ab = orc_compute_area(vref, v1, v3);
    double Ac = orc_compute_area(vref, v1, v2);
    double A = Aa + Ab + Ac;
    Aa = Aa / A;
    Ab = Ab / A;
    Ac = Ac / A;
    return Aa * a1 + Ab * a2 + Ac * a3;
}

/* barycentric, R/triangle.cpp:159-172 */
void orc_barycentric_point(const double v1[3], const double v2[3], const double v3[3], const double vref[3],
                           const double a1[3], const double a2[3], const double a3[3], double out[3]) {
    double Aa = orc_compute_area(vref, v2, v3);
    double Ab = orc_compute_area(vref, v1, v3);
    double Ac = orc_compute_area(vref, v1, v2);
    double A = Aa + Ab + Ac;
    Aa = Aa / A;
    Ab = Ab / A;
    Ac = Ac / A;
    /* va1 * Aa + va2 * Ab + va3 * Ac, left to right */
    for (int k = 0; k < 3; ++k) out[k] = a1[k] * Aa + a2[k] * Ab + a3[k] * Ac;
}

/* estimate_rotation_matrix, R/point.cpp:97-152.  Row-major R taking unit(ci) onto unit(index). */
int orc_rotation_matrix(const double ci_in[3], const double index_in[3], double R[9]) {
    double ci[3] = {ci_in[0], ci_in[1], ci_in[2]};
    double ix[3] = {index_in[0], index_in[1], index_in[2]};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    orc_normalize(ci);
    orc_normalize(ix);
    double cdot = v_dot(ci, ix);
    double theta = acos(cdot);
    if (theta > M_PI) return -1;
    double cr[3];
    v_cross(ci, ix, cr);
    orc_normalize(cr);
    if (fabs(1 - cdot) < ORC_EPSILON) {
        memcpy(R, I, sizeof(I));
    } else if (v_norm(cr) < ORC_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = -I[k];
    } else {
        double u[9] = {0, -cr[2], cr[1], cr[2], 0, -cr[0], -cr[1], cr[0], 0};
        if (fabs(-1 - cdot) < ORC_EPSILON) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R[3 * r + c] = 2 * (cr[r] * cr[c]) - I[3 * r + c];
        } else {
            /* R = I + u*sin(theta) + (1-cos(theta))*(u*u), evaluated left to right per element */
            double uu[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < 3; ++k) s += u[3 * r + k] * u[3 * k + c];
                    uu[3 * r + c] = s;
                }
            double st = sin(theta), omc = 1 - cos(theta);
            for (int k = 0; k < 9; ++k) R[k] = (I[k] + u[k] * st) + omc * uu[k];
        }
    }
    return 0;
}
