/*
 * orc_regtools.c -- oracle restatement of the mesh utilities of newMSM that bracket the label-cost path each
 * iteration: unfold (M/reg_tools.cpp:59-178) and variance_normalise (:804-843).
 * TEST INFRASTRUCTURE ONLY (see msm_oracle.h).  Parity unpinned: the reference ships no vectors for these and
 * cannot be built in this image.
 */
#include "orc_internal.h"

/* computeNormal2EdgeOfTriangle, M/reg_tools.cpp:59-78 */
static void normal_to_edge(const double v0[3], const double v1[3], const double v2[3], double out[3]) {
    double s1[3], s2[3], n[3];
    v_sub(v2, v0, s1);
    v_sub(v1, v0, s2);
    if (v_norm(s1) > 1e-10) orc_normalize(s1);
    else s1[0] = s1[1] = s1[2] = 0.0;
    if (v_norm(s2) > 1e-10) orc_normalize(s2);
    else s2[0] = s2[1] = s2[2] = 0.0;
    v_cross(s1, s2, n);
    if (v_norm(n) > 1e-10) orc_normalize(n);
    else n[0] = n[1] = n[2] = 0.0;
    v_cross(s2, n, out);
    if (v_dot(s1, out) < 0) {
        out[0] = out[0] * -1;
        out[1] = out[1] * -1;
        out[2] = out[2] * -1;
    }
}

/* computeGradientOfBarycentricTriangle, :80-93: norm2edge * 0.5 * base, evaluated left to right */
static void area_gradient(const double v0[3], const double v1[3], const double v2[3], double out[3]) {
    double e[3], n[3];
    normal_to_edge(v0, v1, v2, n);
    v_sub(v1, v0, e);
    const double base = v_norm(e);
    for (int a = 0; a < 3; ++a) out[a] = (n[a] * 0.5) * base;
}

/* spatialgradient, :95-116 */
static void spatial_gradient(const orc_mesh *m, int index, double grad[3]) {
    const double *ci = m->xyz + 3 * (size_t)index;
    grad[0] = grad[1] = grad[2] = 0.0;
    for (int e = m->tid_ptr[index]; e < m->tid_ptr[index + 1]; ++e) {
        const int *t = m->tri + 3 * (size_t)m->tid[e];
        const double *v0 = m->xyz + 3 * (size_t)t[0], *v1 = m->xyz + 3 * (size_t)t[1], *v2 = m->xyz + 3 * (size_t)t[2];
        double d[3], dA[3];
        v_sub(ci, v0, d);
        if (v_norm(d) == 0) {
            area_gradient(v1, v2, v0, dA);
        } else {
            v_sub(ci, v1, d);
            if (v_norm(d) == 0) area_gradient(v2, v0, v1, dA);
            else area_gradient(v0, v1, v2, dA);
        }
        for (int a = 0; a < 3; ++a) grad[a] = grad[a] + dA[a];
    }
}

static void normal_of(const orc_mesh *m, int t, double n[3]) {
    const int *v = m->tri + 3 * (size_t)t;
    orc_tri_normal(m->xyz + 3 * (size_t)v[0], m->xyz + 3 * (size_t)v[1], m->xyz + 3 * (size_t)v[2], n);
}

/* check_for_intersections, :118-129: the normal of the vertex's first triangle against all of its triangles */
static int is_folded(const orc_mesh *m, int ind) {
    double n0[3], n[3];
    normal_of(m, m->tid[m->tid_ptr[ind]], n0);
    for (int e = m->tid_ptr[ind]; e < m->tid_ptr[ind + 1]; ++e) {
        normal_of(m, m->tid[e], n);
        if (v_dot(n0, n) <= 0.5) return 1;
    }
    return 0;
}

/* unfold, :131-178.  Returns the number of passes that moved vertices (0: the mesh was not folded), -1 if a vertex has
 * no triangle (the reference throws in get_triangle_from_vertex).  *first_folded (optional): folded vertices found by
 * the first pass. */
int orc_unfold(orc_mesh *m, double rad, int *first_folded) {
    int *folded = (int *)malloc(sizeof(int) * (size_t)m->V);
    double *grads = (double *)malloc(sizeof(double) * 3 * (size_t)m->V);
    int it = 0;
    if (first_folded) *first_folded = 0;
    for (int i = 0; i < m->V; ++i)
        if (m->tid_ptr[i] == m->tid_ptr[i + 1]) {
            free(folded);
            free(grads);
            return -1;
        }
    for (;;) {
        int nf = 0;
        for (int i = 0; i < m->V; ++i)
            if (is_folded(m, i)) folded[nf++] = i;
        if (it == 0 && first_folded) *first_folded = nf;
        if (nf == 0) break;
        for (int k = 0; k < nf; ++k) spatial_gradient(m, folded[k], grads + 3 * (size_t)k);
        for (int k = 0; k < nf; ++k) {
            double step = 1.0, ci[3], pp[3];
            double *x = m->xyz + 3 * (size_t)folded[k];
            const double *g = grads + 3 * (size_t)k;
            memcpy(ci, x, sizeof(ci));
            do {
                for (int a = 0; a < 3; ++a) pp[a] = ci[a] - g[a] * step;
                orc_normalize(pp);
                for (int a = 0; a < 3; ++a) x[a] = pp[a] * rad;
                step *= 0.5;
            } while (is_folded(m, folded[k]) && step > 1e-3);
            for (int a = 0; a < 3; ++a) x[a] = pp[a] * rad;
        }
        it++;
        if (it == 1000) break;
    }
    free(folded);
    free(grads);
    return it;
}

/* variance_normalise, :804-843: per feature row the running (Welford) mean and variance over the vertices that are not
 * excluded, in vertex order; the row is centred and, when the variance is positive, divided by its square root.
 * data D x V row-major, in place; excl (length V, > 0 keeps the vertex) may be NULL. */
void orc_variance_normalise(double *data, int D, int V, const double *excl) {
    for (int d = 0; d < D; ++d) {
        double *row = data + (size_t)d * V;
        double mean = 0.0, var = 0.0;
        long n = 0;
        for (int i = 0; i < V; ++i) {
            if (excl && !(excl[i] > 0.0)) continue;
            const double delta = row[i] - mean;
            mean += delta / (double)(n + 1);
            var += delta * (row[i] - mean);
            ++n;
        }
        /* _data[i].size()-1 is unsigned: an empty row divides by 2^64-1, a single value by zero */
        var /= n == 0 ? 18446744073709551615.0 : (double)(n - 1);
        for (int i = 0; i < V; ++i) {
            if (excl && !(excl[i] > 0.0)) continue;
            row[i] -= mean;
            if (var > 0.0) row[i] /= sqrt(var);
        }
    }
}
