/*
 * orc_resample.c -- oracle restatement of newMSM's resampler (R/resampler.cpp).
 * TEST INFRASTRUCTURE ONLY (see msm_oracle.h).  Parity unpinned: pinned by structural statistics only.
 *
 * The reference keeps weights in std::map<int,double>: iteration (and therefore every summation)
 * runs in ascending vertex id, and a repeated id overwrites.  The small sorted lists below keep
 * exactly that order.
 */
#include "orc_internal.h"

/* a std::map<int,double> with few entries */
typedef struct {
    int n, cap;
    int *key;
    double *val;
} wmap;

static void wmap_set(wmap *m, int k, double v) {
    int i = 0;
    while (i < m->n && m->key[i] < k) ++i;
    if (i < m->n && m->key[i] == k) {
        m->val[i] = v;
        return;
    }
    if (m->n == m->cap) {
        m->cap = m->cap ? 2 * m->cap : 4;
        m->key = (int *)realloc(m->key, sizeof(int) * m->cap);
        m->val = (double *)realloc(m->val, sizeof(double) * m->cap);
    }
    memmove(&m->key[i + 1], &m->key[i], sizeof(int) * (m->n - i));
    memmove(&m->val[i + 1], &m->val[i], sizeof(double) * (m->n - i));
    m->key[i] = k;
    m->val[i] = v;
    m->n++;
}

static void wmap_free(wmap *m, int count) {
    for (int i = 0; i < count; ++i) {
        free(m[i].key);
        free(m[i].val);
    }
    free(m);
}

/* Resampler::get_barycentric_weights, R/resampler.cpp:142-167 */
static int bary_weights(const orc_octree *t, const double *q, int N, int *tri_id, int *vid, double *w, int raw) {
    const orc_mesh *m = t->mesh;
    int status = 0;
    for (int k = 0; k < N; ++k) {
        const double *ci = &q[3 * k];
        int tr = orc_octree_closest_triangle(t, ci, NULL);
        if (tri_id) tri_id[k] = tr;
        if (tr < 0) {
            if (!status) status = tr;
            for (int j = 0; j < 3; ++j) {
                vid[3 * k + j] = -1;
                w[3 * k + j] = 0.0;
            }
            continue;
        }
        const int *n = &m->tri[3 * tr];
        const double *v0 = &m->xyz[3 * n[0]], *v1 = &m->xyz[3 * n[1]], *v2 = &m->xyz[3 * n[2]];
        for (int j = 0; j < 3; ++j) vid[3 * k + j] = n[j];
        if (!raw)
            orc_calc_barycentric_weights(v0, v1, v2, ci, &w[3 * k]);
        else {
            /* the normalised areas inside barycentric_interpolation, R/triangle.cpp:147-154 */
            double Aa = orc_compute_area(ci, v1, v2), Ab = orc_compute_area(ci, v0, v2), Ac = orc_compute_area(ci, v0, v1);
            double A = Aa + Ab + Ac;
            w[3 * k] = Aa / A;
            w[3 * k + 1] = Ab / A;
            w[3 * k + 2] = Ac / A;
        }
    }
    return status;
}

int orc_barycentric_weights(const orc_octree *t, const double *q, int N, int *tri_id, int *vid, double *w) {
    return bary_weights(t, q, N, tri_id, vid, w, 0);
}
int orc_barycentric_weights_raw(const orc_octree *t, const double *q, int N, int *tri_id, int *vid, double *w) {
    return bary_weights(t, q, N, tri_id, vid, w, 1);
}

/* Resampler::get_adaptive_barycentric_weights, R/resampler.cpp:72-140 (single thread order) */
long orc_adaptive_barycentric_weights(const orc_mesh *in_mesh, const orc_mesh *new_mesh, const double *excl,
                                      int *row_ptr, int *col, double *val) {
    int nOld = in_mesh->V, nNew = new_mesh->V;
    orc_octree *tree_in = orc_octree_build(in_mesh);
    orc_octree *tree_new = orc_octree_build(new_mesh);
    int *fvid = (int *)malloc(sizeof(int) * 3 * nNew), *rvid = (int *)malloc(sizeof(int) * 3 * nOld);
    double *fw = (double *)malloc(sizeof(double) * 3 * nNew), *rw = (double *)malloc(sizeof(double) * 3 * nOld);
    long result = -1;
    wmap *forward = (wmap *)calloc(nNew, sizeof(wmap)), *reorder = (wmap *)calloc(nNew, sizeof(wmap));
    wmap *adapt = (wmap *)calloc(nNew, sizeof(wmap));
    double *newA = (double *)calloc(nNew, sizeof(double)), *oldA = (double *)calloc(nOld, sizeof(double));
    double *corr = (double *)calloc(nOld, sizeof(double));
    char *active = (char *)calloc(nNew, 1);

    if (bary_weights(tree_in, new_mesh->xyz, nNew, NULL, fvid, fw, 0) < 0) goto done;   /* forward, :75 */
    if (bary_weights(tree_new, in_mesh->xyz, nOld, NULL, rvid, rw, 0) < 0) goto done;   /* reverse, :78 */
    for (int k = 0; k < nNew; ++k)
        for (int j = 0; j < 3; ++j) wmap_set(&forward[k], fvid[3 * k + j], fw[3 * k + j]);
    for (int o = 0; o < nOld; ++o) { /* :91-97, iterating reverse[o] in ascending key */
        oldA[o] = orc_mesh_vertex_area(in_mesh, o);
        wmap tmp = {0, 0, NULL, NULL};
        for (int j = 0; j < 3; ++j) wmap_set(&tmp, rvid[3 * o + j], rw[3 * o + j]);
        for (int e = 0; e < tmp.n; ++e) wmap_set(&reorder[tmp.key[e]], o, tmp.val[e]);
        free(tmp.key);
        free(tmp.val);
    }
    for (int k = 0; k < nNew; ++k) { /* :99-118 */
        if (excl) {
            int cv = orc_octree_closest_vertex(tree_in, &new_mesh->xyz[3 * k]);
            if (cv < 0 || excl[cv] == 0) continue;
        }
        active[k] = 1;
        newA[k] = orc_mesh_vertex_area(new_mesh, k);
        const wmap *src = (reorder[k].n <= forward[k].n) ? &forward[k] : &reorder[k];
        for (int e = 0; e < src->n; ++e) wmap_set(&adapt[k], src->key[e], src->val[e]);
        for (int e = 0; e < adapt[k].n; ++e) {
            adapt[k].val[e] *= newA[k];
            corr[adapt[k].key[e]] += adapt[k].val[e];
        }
    }
    for (int k = 0; k < nNew; ++k) { /* :120-137 */
        if (!active[k]) continue;
        double wsum = 0.0;
        for (int e = 0; e < adapt[k].n; ++e) {
            adapt[k].val[e] *= oldA[adapt[k].key[e]] / corr[adapt[k].key[e]];
            wsum += adapt[k].val[e];
        }
        if (wsum != 0.0)
            for (int e = 0; e < adapt[k].n; ++e) adapt[k].val[e] /= wsum;
    }
    result = 0;
    for (int k = 0; k < nNew; ++k) {
        if (row_ptr) row_ptr[k] = (int)result;
        for (int e = 0; e < adapt[k].n; ++e, ++result)
            if (col) {
                col[result] = adapt[k].key[e];
                val[result] = adapt[k].val[e];
            }
    }
    if (row_ptr) row_ptr[nNew] = (int)result;
done:
    wmap_free(forward, nNew);
    wmap_free(reorder, nNew);
    wmap_free(adapt, nNew);
    free(fvid); free(rvid); free(fw); free(rw); free(newA); free(oldA); free(corr); free(active);
    orc_octree_destroy(tree_in);
    orc_octree_destroy(tree_new);
    return result;
}

/* Resampler::barycentric_data_interpolation, R/resampler.cpp:40-52 */
void orc_apply_weights(const int *row_ptr, const int *col, const double *val, int Nnew,
                       const double *data, int D, int Vin, const double *excl, double *out) {
    for (int d = 0; d < D; ++d)
        for (int k = 0; k < Nnew; ++k) {
            double acc = 0.0;
            for (int e = row_ptr[k]; e < row_ptr[k + 1]; ++e)
                if (!excl || excl[col[e]] != 0) acc += data[(long)d * Vin + col[e]] * val[e];
            out[(long)d * Nnew + k] = acc;
        }
}

/* metric_resample, R/resampler.cpp:304-309 */
int orc_metric_resample(const orc_mesh *in_mesh, const double *data, int D, const orc_mesh *new_mesh, double *out) {
    return orc_metric_resample_excl(in_mesh, data, D, new_mesh, NULL, out, NULL);
}

/* barycentric_data_interpolation with EXCL, R/resampler.cpp:30-70: the mask enters the weights (:38), the sums (:45-47) and is
 * itself resampled (:54-67) */
int orc_metric_resample_excl(const orc_mesh *in_mesh, const double *data, int D, const orc_mesh *new_mesh, const double *excl, double *out,
                             double *excl_out) {
    long nnz = orc_adaptive_barycentric_weights(in_mesh, new_mesh, excl, NULL, NULL, NULL);
    if (nnz < 0) return -1;
    int *rp = (int *)malloc(sizeof(int) * (new_mesh->V + 1)), *col = (int *)malloc(sizeof(int) * (nnz + 1));
    double *val = (double *)malloc(sizeof(double) * (nnz + 1));
    orc_adaptive_barycentric_weights(in_mesh, new_mesh, excl, rp, col, val);
    orc_apply_weights(rp, col, val, new_mesh->V, data, D, in_mesh->V, excl, out);
    if (excl && excl_out)
        for (int k = 0; k < new_mesh->V; ++k) {
            double acc = 0.0;
            for (int e = rp[k]; e < rp[k + 1]; ++e)
                if (excl[col[e]] != 0) acc += excl[col[e]] * val[e];
            excl_out[k] = acc;
        }
    free(rp); free(col); free(val);
    return 0;
}

/* create_exclusion, R/mesh.cpp:1257-1273 */
void orc_create_exclusion(const double *data, int D, int V, double thrl, double thru, double *excl) {
    for (int i = 0; i < V; ++i) {
        excl[i] = 0.0;
        for (int d = 0; d < D; ++d)
            if (!(data[(long)d * V + i] >= (thrl - ORC_EPSILON) && data[(long)d * V + i] <= (thru + ORC_EPSILON))) {
                excl[i] = 1.0;
                break;
            }
    }
}

/* nearest_neighbour_interpolation with EXCL, R/resampler.cpp:232-258 */
int orc_nearest_neighbour_excl(const orc_mesh *orig, const double *data, int D, const double *q, int N, const double *excl, double *out,
                               double *excl_out) {
    orc_octree *t = orc_octree_build(orig);
    int st = 0;
    for (int i = 0; i < N; ++i) {
        int cv = orc_octree_closest_vertex(t, &q[3 * i]);
        if (cv < 0) { st = cv; break; }
        if (excl_out) excl_out[i] = 0.0;
        for (int d = 0; d < D; ++d) out[(long)d * N + i] = 0.0;
        if (!excl || excl[cv] != 0) {
            if (excl && excl_out) excl_out[i] = excl[cv];
            for (int d = 0; d < D; ++d) out[(long)d * N + i] = data[(long)d * orig->V + cv];
        }
    }
    orc_octree_destroy(t);
    return st;
}

/* sphere_project_warp, R/resampler.cpp:311-328 */
int orc_sphere_project_warp(double *sphere, int N, const orc_mesh *from, const double *to_xyz) {
    orc_octree *t = orc_octree_build(from);
    int *vid = (int *)malloc(sizeof(int) * 3 * N);
    double *w = (double *)malloc(sizeof(double) * 3 * N);
    int st = bary_weights(t, sphere, N, NULL, vid, w, 0);
    if (st == 0)
        for (int i = 0; i < N; ++i) {
            wmap mp = {0, 0, NULL, NULL};
            for (int j = 0; j < 3; ++j) wmap_set(&mp, vid[3 * i + j], w[3 * i + j]);
            double p[3] = {0, 0, 0};
            for (int e = 0; e < mp.n; ++e)
                for (int a = 0; a < 3; ++a) p[a] += to_xyz[3 * mp.key[e] + a] * mp.val[e];
            orc_normalize(p);
            for (int a = 0; a < 3; ++a) sphere[3 * i + a] = p[a] * 100;
            free(mp.key);
            free(mp.val);
        }
    free(vid); free(w);
    orc_octree_destroy(t);
    return st;
}

/* nearest_neighbour_interpolation, R/resampler.cpp:232-258 (no exclusion mask) */
int orc_nearest_neighbour(const orc_mesh *orig, const double *data, int D, const double *q, int N, double *out) {
    orc_octree *t = orc_octree_build(orig);
    int st = 0;
    for (int i = 0; i < N; ++i) {
        int cv = orc_octree_closest_vertex(t, &q[3 * i]);
        if (cv < 0) { st = cv; break; }
        for (int d = 0; d < D; ++d) out[(long)d * N + i] = data[(long)d * orig->V + cv];
    }
    orc_octree_destroy(t);
    return st;
}

/* smooth_data, R/resampler.cpp:168-230, index for index: the octree is built over `orig`, but the closest vertex id it
 * returns is then used to index sphLow's coordinates (:185) and the neighbour loop runs over sphLow's vertices while
 * reading orig's data (:210) -- the function is meant for orig and sphLow being the same mesh (its callers pass
 * that).  excl (optional, >= max(V) values) is the EXCL mesh's data; excl_out (optional, sphLow->V) receives the
 * smoothed mask.  check_scale (R/mesh.cpp:1198-1208) is the caller's job here.  Returns 0, or a negative octree
 * error, or -3 when a closest vertex id does not exist in sphLow. */
int orc_smooth_data(const orc_mesh *orig, const double *data, int D, const orc_mesh *sphLow, double sigma, const double *excl, double *out,
                    double *excl_out) {
    const int N = sphLow->V;
    const double ang = 4 * asin(sigma / (2 * ORC_RAD));
    const double cosang = cos(ang);
    orc_octree *t = orc_octree_build(orig);
    int st = 0;
    for (long k = 0; k < (long)D * N; ++k) out[k] = 0.0;
    /* the reference's loop over the output vertices is an OpenMP loop too (:179); every vertex is computed as in the serial code */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (excl_out) excl_out[i] = 0.0;
        const int cv = orc_octree_closest_vertex(t, &sphLow->xyz[3 * i]);
        if (cv < 0 || cv >= N) {
#pragma omp critical
            if (st == 0) st = cv < 0 ? cv : -3;
            continue;
        }
        double ref[3] = {sphLow->xyz[3 * cv], sphLow->xyz[3 * cv + 1], sphLow->xyz[3 * cv + 2]};
        orc_normalize(ref);
        double SUM = 0.0, excl_sum = 0.0;
        if (!excl || excl[cv] > 0) {
            for (int n = 0; n < N; ++n) {
                double actual[3] = {sphLow->xyz[3 * n], sphLow->xyz[3 * n + 1], sphLow->xyz[3 * n + 2]};
                orc_normalize(actual);
                if (!((actual[0] * ref[0] + actual[1] * ref[1] + actual[2] * ref[2]) >= cosang)) continue;
                const double dx = ref[0] - actual[0], dy = ref[1] - actual[1], dz = ref[2] - actual[2];
                const double chord = sqrt(dx * dx + dy * dy + dz * dz);
                const double g = 2 * ORC_RAD * asin(chord / (2 * ORC_RAD));
                double weight = (1 / sqrt(2 * M_PI * sigma * sigma)) * exp(-(g * g) / (2 * sigma * sigma));
                excl_sum += weight;
                if (excl) weight = excl[n] * weight;
                SUM += weight;
                for (int d = 0; d < D; ++d) out[(long)d * N + i] += data[(long)d * orig->V + n] * weight;
            }
            if (excl_sum != 0.0 && excl && excl_out) excl_out[i] = SUM / excl_sum;
            for (int d = 0; d < D; ++d)
                if (SUM != 0.0) out[(long)d * N + i] /= SUM;
        }
    }
    orc_octree_destroy(t);
    return st;
}
