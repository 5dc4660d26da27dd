/*
 * orc_mesh.c -- oracle restatement of newMSM's icosphere generation and mesh adjacency.
 * TEST INFRASTRUCTURE ONLY (see msm_oracle.h).  Parity unpinned: pinned by structural statistics only.
 */
#include "orc_internal.h"

/* ------------------------------------------------------------------ icosphere */

void orc_icosphere_counts(int order, int *V, int *T) {
    long t = 20, v = 12;
    for (int i = 0; i < order; ++i) {
        v = v + (t * 3) / 2; /* one new vertex per edge, E = 3T/2 */
        t *= 4;
    }
    *V = (int)v;
    *T = (int)t;
}

/* Point operator==, R/point.cpp:239-243 */
static int pt_equal(const double *a, const double *b) {
    return fabs(a[0] - b[0]) < ORC_EPSILON && fabs(a[1] - b[1]) < ORC_EPSILON && fabs(a[2] - b[2]) < ORC_EPSILON;
}

typedef struct {
    long key;
    int val;
} edge_slot;

static unsigned long hash_key(long k) {
    unsigned long x = (unsigned long)k;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdUL;
    x ^= x >> 33;
    return x;
}

/*
 * retessellate, R/mesh.cpp:910-1005.  For every old triangle (v0,v1,v2): midpoints pt0=(v1,v2),
 * pt2=(v0,v1), pt1=(v0,v2); each is looked up among the points added so far (tolerance equality,
 * last match wins) and otherwise created, numbered in the order p0,p1,p2; four children
 * (p2,p0,p1) (p1,v0,p2) (p0,v2,p1) (p2,v1,p0); finally every vertex is re-normalised.
 */
static void retessellate(double *xyz, int *V, int *tri, int *T, int literal) {
    int oldT = *T, oldV = *V, nv = *V;
    int *old = (int *)malloc(sizeof(int) * 3 * oldT);
    memcpy(old, tri, sizeof(int) * 3 * oldT);
    long cap = 1;
    while (cap < 4L * oldT) cap <<= 1;
    edge_slot *tab = NULL;
    if (!literal) {
        tab = (edge_slot *)malloc(sizeof(edge_slot) * cap);
        for (long i = 0; i < cap; ++i) tab[i].key = -1;
    }
    int nt = 0;
    for (int t = 0; t < oldT; ++t) {
        int v[3] = {old[3 * t], old[3 * t + 1], old[3 * t + 2]};
        /* edge (a,b) of midpoint m: m0 <-> (v1,v2), m1 <-> (v0,v2), m2 <-> (v0,v1) */
        int ea[3] = {v[1], v[0], v[0]}, eb[3] = {v[2], v[2], v[1]};
        double mid[3][3];
        int p[3] = {-1, -1, -1};
        for (int m = 0; m < 3; ++m)
            for (int k = 0; k < 3; ++k) mid[m][k] = (xyz[3 * ea[m] + k] + xyz[3 * eb[m] + k]) / 2;
        if (literal) {
            for (int a = oldV; a < nv; ++a)
                for (int m = 0; m < 3; ++m)
                    if (pt_equal(mid[m], &xyz[3 * a])) p[m] = a;
        } else {
            for (int m = 0; m < 3; ++m) {
                int lo = ea[m] < eb[m] ? ea[m] : eb[m], hi = ea[m] < eb[m] ? eb[m] : ea[m];
                long key = (long)lo * (long)(1L << 31) + hi;
                unsigned long h = hash_key(key) & (unsigned long)(cap - 1);
                while (tab[h].key != -1 && tab[h].key != key) h = (h + 1) & (unsigned long)(cap - 1);
                if (tab[h].key == key) p[m] = tab[h].val;
            }
        }
        for (int m = 0; m < 3; ++m)
            if (p[m] < 0) {
                p[m] = nv;
                memcpy(&xyz[3 * nv], mid[m], sizeof(double) * 3);
                if (!literal) {
                    int lo = ea[m] < eb[m] ? ea[m] : eb[m], hi = ea[m] < eb[m] ? eb[m] : ea[m];
                    long key = (long)lo * (long)(1L << 31) + hi;
                    unsigned long h = hash_key(key) & (unsigned long)(cap - 1);
                    while (tab[h].key != -1) h = (h + 1) & (unsigned long)(cap - 1);
                    tab[h].key = key;
                    tab[h].val = nv;
                }
                ++nv;
            }
        int c[4][3] = {{p[2], p[0], p[1]}, {p[1], v[0], p[2]}, {p[0], v[2], p[1]}, {p[2], v[1], p[0]}};
        for (int q = 0; q < 4; ++q, ++nt) memcpy(&tri[3 * nt], c[q], sizeof(int) * 3);
    }
    for (int i = 0; i < nv; ++i) orc_normalize(&xyz[3 * i]);
    *V = nv;
    *T = nt;
    free(old);
    free(tab);
}

/* make_mesh_from_icosa, R/mesh.cpp:1111-1196 */
int orc_icosphere(int order, int literal, double *xyz, int *tri) {
    const double tau = 0.8506508084, one = 0.5257311121;
    /* ZA ZB ZC ZD YA YB YC YD XA XB XC XD = 0..11 */
    const double base[12][3] = {{tau, one, 0},  {-tau, one, 0}, {-tau, -one, 0}, {tau, -one, 0},
                                {one, 0, tau},  {one, 0, -tau}, {-one, 0, -tau}, {-one, 0, tau},
                                {0, tau, one},  {0, -tau, one}, {0, -tau, -one}, {0, tau, -one}};
    enum { ZA, ZB, ZC, ZD, YA, YB, YC, YD, XA, XB, XC, XD };
    const int faces[20][3] = {{YD, XA, YA}, {XB, YD, YA}, {XD, YC, YB}, {YC, XC, YB}, {ZD, YA, ZA},
                              {YB, ZD, ZA}, {ZB, YD, ZC}, {YC, ZB, ZC}, {XD, ZA, XA}, {ZB, XD, XA},
                              {ZD, XC, XB}, {XC, ZC, XB}, {ZA, YA, XA}, {YB, ZA, XD}, {ZD, XB, YA},
                              {XC, ZD, YB}, {ZB, XA, YD}, {XD, ZB, YC}, {XB, ZC, YD}, {ZC, XC, YC}};
    int V = 12, T = 20;
    memcpy(xyz, base, sizeof(base));
    for (int t = 0; t < 20; ++t) { /* swap_orientation: vertices 1 and 2 exchanged */
        tri[3 * t] = faces[t][0];
        tri[3 * t + 1] = faces[t][2];
        tri[3 * t + 2] = faces[t][1];
    }
    for (int io = 0; io < order; ++io) retessellate(xyz, &V, tri, &T, literal);
    return 0;
}

/* true_rescale, R/mesh.cpp:1210-1219 */
void orc_true_rescale(double *xyz, int V, double rad) {
    for (int i = 0; i < V; ++i) {
        double *p = &xyz[3 * i];
        orc_normalize(p);
        p[0] = p[0] * rad;
        p[1] = p[1] * rad;
        p[2] = p[2] * rad;
    }
}

/* ------------------------------------------------------------------ resample_anatomy (aMSM, --regoption=5) */

/* a growing list of ints with std::vector's insert-at-the-front, as M/mesh_registration.cpp:277-279 uses it */
typedef struct {
    int *v;
    int n, cap;
} int_list;
static void list_reserve(int_list *l, int want) {
    if (want <= l->cap) return;
    while (l->cap < want) l->cap = l->cap ? 2 * l->cap : 8;
    l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap);
}
static void list_push_back(int_list *l, int x) {
    list_reserve(l, l->n + 1);
    l->v[l->n++] = x;
}
static void list_insert_front(int_list *l, const int *src, int count) { /* vector::insert(begin(), first, last) */
    list_reserve(l, l->n + count);
    memmove(l->v + count, l->v, sizeof(int) * (size_t)l->n);
    memcpy(l->v, src, sizeof(int) * (size_t)count);
    l->n += count;
}
static void list_assign(int_list *dst, const int_list *src) {
    dst->n = 0;
    list_reserve(dst, src->n);
    memcpy(dst->v, src->v, sizeof(int) * (size_t)src->n);
    dst->n = src->n;
}

/*
 * Mesh_registration::resample_anatomy, M/mesh_registration.cpp:250-332 (without its surface_resample call :323), statement by statement:
 * ANAT_ico = control_grid; `levels` x retessellate(ANAT_ico, FACE_neighbours_tmp) -- R/mesh.cpp:1007-1109: old_tr_nbours[t] = the four children of t in
 * the order they are numbered --; from the second pass on the lists are merged by inserting the children of every entry at the FRONT (:268-283);
 * true_rescale (:299); baryweights[id] = calc_barycentric_weights(...) in the loop order of :303-321 (a later assignment replaces the map).
 * Sizes: Va = N + (vertices added), Ta = Tc * 4^levels; the caller allocates from orc_resample_anatomy_sizes.  literal: the O(V^2) duplicate search.
 */
void orc_resample_anatomy_sizes(int N, int Tc, int levels, int *Va, int *Ta) {
    long v = N, t = Tc;
    for (int i = 0; i < levels; ++i) {
        v += 3 * t / 2;
        t *= 4;
    }
    *Va = (int)v;
    *Ta = (int)t;
}
int orc_resample_anatomy_grid(const double *cp_xyz, int N, const int *cp_tri, int Tc, int levels, double rad, int literal, double *axyz, int *atri, int *w_ptr,
                              int *w_cp, double *w_val, int *face_ptr, int *face_idx) {
    int V = N, T = Tc;
    memcpy(axyz, cp_xyz, sizeof(double) * 3 * (size_t)N);
    memcpy(atri, cp_tri, sizeof(int) * 3 * (size_t)Tc);
    int_list *nb = (int_list *)calloc((size_t)Tc, sizeof(int_list));   /* ANAT_to_CPgrid_neighbours */
    int_list *fn = (int_list *)calloc((size_t)Tc, sizeof(int_list));   /* FACE_neighbours */
    if (levels > 0) {
        for (int i = 0; i < levels; ++i) {
            const int oldT = T;
            retessellate(axyz, &V, atri, &T, literal);
            /* FACE_neighbours_tmp[t] = {4t, 4t+1, 4t+2, 4t+3}: tot_triangles counts up through the old triangles in order (R/mesh.cpp:1083-1094) */
            if (i > 0) {
                for (int j = 0; j < Tc; ++j) {
                    nb[j].n = 0; /* ANAT_to_CPgrid_neighbours.clear(); push_back(tmp) */
                    for (int k = 0; k < fn[j].n; ++k) {
                        const int f = fn[j].v[k];
                        if (f < 0 || f >= oldT) return -1;
                        const int kids[4] = {4 * f, 4 * f + 1, 4 * f + 2, 4 * f + 3};
                        list_insert_front(&nb[j], kids, 4);
                    }
                }
                for (int j = 0; j < Tc; ++j) list_assign(&fn[j], &nb[j]);
            } else {
                for (int j = 0; j < Tc; ++j) {
                    fn[j].n = 0;
                    for (int c = 0; c < 4; ++c) list_push_back(&fn[j], 4 * j + c);
                }
            }
        }
        for (int j = 0; j < Tc; ++j)
            if (nb[j].n == 0) list_assign(&nb[j], &fn[j]); /* one increase in resolution: the result of the retessellation directly (:284-286) */
    } else {
        for (int i = 0; i < Tc; ++i) list_push_back(&nb[i], i);
    }
    orc_true_rescale(axyz, V, rad);
    /* baryweights: std::map<int,double> per vertex, here three (id, weight) slots kept sorted by id */
    int *cnt = (int *)calloc((size_t)V, sizeof(int));
    int *ids = (int *)malloc(sizeof(int) * 3 * (size_t)V);
    double *ws = (double *)malloc(sizeof(double) * 3 * (size_t)V);
    for (int i = 0; i < Tc; ++i) {
        const int id[3] = {cp_tri[3 * i], cp_tri[3 * i + 1], cp_tri[3 * i + 2]};
        const double *v0 = &cp_xyz[3 * id[0]], *v1 = &cp_xyz[3 * id[1]], *v2 = &cp_xyz[3 * id[2]];
        for (int jj = 0; jj < nb[i].n; ++jj) {
            const int j = nb[i].v[jj];
            for (int k = 0; k < 3; ++k) {
                const int a = atri[3 * j + k];
                double w[3];
                orc_calc_barycentric_weights(v0, v1, v2, &axyz[3 * a], w);
                /* weights[n1] = ..; weights[n2] = ..; weights[n3] = ..  into a fresh map */
                int n = 0, mid[3];
                double mw[3];
                for (int q = 0; q < 3; ++q) {
                    int at = -1;
                    for (int r = 0; r < n; ++r)
                        if (mid[r] == id[q]) at = r;
                    if (at >= 0) {
                        mw[at] = w[q];
                        continue;
                    }
                    int pos = n;
                    while (pos > 0 && mid[pos - 1] > id[q]) {
                        mid[pos] = mid[pos - 1];
                        mw[pos] = mw[pos - 1];
                        --pos;
                    }
                    mid[pos] = id[q];
                    mw[pos] = w[q];
                    ++n;
                }
                cnt[a] = n;
                for (int r = 0; r < n; ++r) {
                    ids[3 * a + r] = mid[r];
                    ws[3 * a + r] = mw[r];
                }
            }
        }
    }
    w_ptr[0] = 0;
    for (int a = 0; a < V; ++a) {
        for (int r = 0; r < cnt[a]; ++r) {
            w_cp[w_ptr[a] + r] = ids[3 * a + r];
            w_val[w_ptr[a] + r] = ws[3 * a + r];
        }
        w_ptr[a + 1] = w_ptr[a] + cnt[a];
    }
    face_ptr[0] = 0;
    for (int i = 0; i < Tc; ++i) {
        memcpy(face_idx + face_ptr[i], nb[i].v, sizeof(int) * (size_t)nb[i].n);
        face_ptr[i + 1] = face_ptr[i] + nb[i].n;
    }
    for (int j = 0; j < Tc; ++j) {
        free(nb[j].v);
        free(fn[j].v);
    }
    free(nb);
    free(fn);
    free(cnt);
    free(ids);
    free(ws);
    return 0;
}

/* ------------------------------------------------------------------ mesh */

static int has_nbr(const int *list, int n, int v) {
    for (int i = 0; i < n; ++i)
        if (list[i] == v) return 1;
    return 0;
}

/* adjacency exactly as repeated Mesh::push_triangle would build it, R/mesh.cpp:115-134 */
orc_mesh *orc_mesh_create(const double *xyz, int V, const int *tri, int T) {
    orc_mesh *m = (orc_mesh *)calloc(1, sizeof(orc_mesh));
    m->V = V;
    m->T = T;
    m->xyz = (double *)malloc(sizeof(double) * 3 * V);
    memcpy(m->xyz, xyz, sizeof(double) * 3 * V);
    m->tri = (int *)malloc(sizeof(int) * 3 * T);
    memcpy(m->tri, tri, sizeof(int) * 3 * T);
    m->tarea = (double *)malloc(sizeof(double) * T);
    m->tid_ptr = (int *)calloc(V + 1, sizeof(int));
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) m->tid_ptr[tri[3 * t + k] + 1]++;
    for (int v = 0; v < V; ++v) m->tid_ptr[v + 1] += m->tid_ptr[v];
    m->tid = (int *)malloc(sizeof(int) * (m->tid_ptr[V] > 0 ? m->tid_ptr[V] : 1));
    int *fill = (int *)calloc(V, sizeof(int));
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            int v = tri[3 * t + k];
            m->tid[m->tid_ptr[v] + fill[v]++] = t;
        }
    /* neighbour capacity: at most 2 per incident triangle */
    int *cap_ptr = (int *)calloc(V + 1, sizeof(int));
    for (int v = 0; v < V; ++v) cap_ptr[v + 1] = cap_ptr[v] + 2 * (m->tid_ptr[v + 1] - m->tid_ptr[v]);
    int *tmp = (int *)malloc(sizeof(int) * (cap_ptr[V] > 0 ? cap_ptr[V] : 1));
    memset(fill, 0, sizeof(int) * V);
    for (int t = 0; t < T; ++t) {
        const int *n = &tri[3 * t];
        static const int order[6][2] = {{0, 1}, {0, 2}, {1, 0}, {1, 2}, {2, 0}, {2, 1}};
        for (int q = 0; q < 6; ++q) {
            int a = n[order[q][0]], b = n[order[q][1]];
            if (!has_nbr(&tmp[cap_ptr[a]], fill[a], b)) tmp[cap_ptr[a] + fill[a]++] = b;
        }
    }
    m->nbr_ptr = (int *)calloc(V + 1, sizeof(int));
    for (int v = 0; v < V; ++v) m->nbr_ptr[v + 1] = m->nbr_ptr[v] + fill[v];
    m->nbr = (int *)malloc(sizeof(int) * (m->nbr_ptr[V] > 0 ? m->nbr_ptr[V] : 1));
    for (int v = 0; v < V; ++v) memcpy(&m->nbr[m->nbr_ptr[v]], &tmp[cap_ptr[v]], sizeof(int) * fill[v]);
    free(tmp);
    free(cap_ptr);
    free(fill);
    for (int t = 0; t < T; ++t)
        m->tarea[t] = orc_tri_calc_area(&m->xyz[3 * tri[3 * t]], &m->xyz[3 * tri[3 * t + 1]], &m->xyz[3 * tri[3 * t + 2]]);
    return m;
}

void orc_mesh_destroy(orc_mesh *m) {
    if (!m) return;
    free(m->xyz);
    free(m->tri);
    free(m->tarea);
    free(m->nbr_ptr);
    free(m->nbr);
    free(m->tid_ptr);
    free(m->tid);
    free(m);
}

void orc_mesh_set_coords(orc_mesh *m, const double *xyz, int refresh_areas) {
    memcpy(m->xyz, xyz, sizeof(double) * 3 * m->V);
    if (refresh_areas)
        for (int t = 0; t < m->T; ++t)
            m->tarea[t] = orc_tri_calc_area(&m->xyz[3 * m->tri[3 * t]], &m->xyz[3 * m->tri[3 * t + 1]], &m->xyz[3 * m->tri[3 * t + 2]]);
}

int orc_mesh_nvertices(const orc_mesh *m) { return m->V; }
int orc_mesh_ntriangles(const orc_mesh *m) { return m->T; }
const double *orc_mesh_coords(const orc_mesh *m) { return m->xyz; }
const int *orc_mesh_triangles(const orc_mesh *m) { return m->tri; }

void orc_mesh_adjacency(const orc_mesh *m, const int **nbr_ptr, const int **nbr, const int **tid_ptr, const int **tid) {
    *nbr_ptr = m->nbr_ptr;
    *nbr = m->nbr;
    *tid_ptr = m->tid_ptr;
    *tid = m->tid;
}

/* compute_vertex_area, R/mesh.cpp:1275-1283: mean of the adjacent triangles' cached areas */
double orc_mesh_vertex_area(const orc_mesh *m, int v) {
    double sum = 0;
    for (int i = m->tid_ptr[v]; i < m->tid_ptr[v + 1]; ++i) sum += m->tarea[m->tid[i]];
    return sum / (m->tid_ptr[v + 1] - m->tid_ptr[v]);
}

/* Mesh::calculate_MaxVD, R/mesh.cpp:263-277 */
double orc_mesh_max_vd(const orc_mesh *m) {
    double best = -DBL_MAX;
    for (int i = 0; i < m->V; ++i)
        for (int j = m->nbr_ptr[i]; j < m->nbr_ptr[i + 1]; ++j) {
            double d[3];
            v_sub(&m->xyz[3 * i], &m->xyz[3 * m->nbr[j]], d);
            double dist = 2 * ORC_RAD * asin(v_norm(d) / (2 * ORC_RAD));
            if (dist > best) best = dist;
        }
    return best;
}

/* Mesh::calculate_MeanVD, R/mesh.cpp:279-297 */
double orc_mesh_mean_vd(const orc_mesh *m) {
    int k = 0;
    double kr = 0.0;
    for (int i = 0; i < m->V; ++i)
        for (int j = m->nbr_ptr[i]; j < m->nbr_ptr[i + 1]; ++j) {
            double d[3];
            v_sub(&m->xyz[3 * m->nbr[j]], &m->xyz[3 * i], d);
            k++;
            kr += v_norm(d);
        }
    return kr / k;
}
