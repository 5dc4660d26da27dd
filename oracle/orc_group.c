/*
 * orc_group.c -- oracle restatement of newMSM's groupwise (gMSM) model and cost function
 * (M/DiscreteGroupModel.cpp, M/DiscreteGroupCostFunction.cpp).  TEST INFRASTRUCTURE ONLY (see msm_oracle.h).
 * Parity unpinned: pinned by structural statistics only.
 */
#include "orc_internal.h"

struct orc_group {
    orc_group_params p;
    int S;
    const orc_mesh *tmpl;
    double *mask;
    int N, Tc;            /* control grid size */
    int *cp_tri;          /* Tc x 3 */
    orc_mesh **cpmesh;    /* per subject control grid (own copies) */
    const orc_mesh **data;
    double **feat;        /* per subject D x V(data) */
    double **orig;        /* _ORIG_MESHES[s] coords */
    int D;
    double *labels;
    int L;
    double subcorr;
    /* setup products */
    int *pairs, P;
    int *triplets, T;
    double *rot;          /* S*N x 9 */
    double **spacing;     /* per subject N */
    double **F;           /* per (s,l): D x V_tmpl resampled features */
    int **pptr, **pidx;   /* per subject: CSR over (v*L + l) of template vertex ids */
    int nthreads;         /* OpenMP threads of get_patch_data's loop over the subjects (0 / 1: serial) */
};

orc_group *orc_group_create(const orc_group_params *p, int S) {
    orc_group *g = (orc_group *)calloc(1, sizeof(orc_group));
    g->p = *p;
    g->S = S;
    g->cpmesh = (orc_mesh **)calloc(S, sizeof(orc_mesh *));
    g->data = (const orc_mesh **)calloc(S, sizeof(orc_mesh *));
    g->feat = (double **)calloc(S, sizeof(double *));
    g->orig = (double **)calloc(S, sizeof(double *));
    g->spacing = (double **)calloc(S, sizeof(double *));
    g->pptr = (int **)calloc(S, sizeof(int *));
    g->pidx = (int **)calloc(S, sizeof(int *));
    return g;
}

static void free_setup(orc_group *g) {
    free(g->pairs); g->pairs = NULL;
    free(g->rot); g->rot = NULL;
    for (int s = 0; s < g->S; ++s) {
        free(g->spacing[s]); g->spacing[s] = NULL;
        free(g->pptr[s]); g->pptr[s] = NULL;
        free(g->pidx[s]); g->pidx[s] = NULL;
    }
    if (g->F) {
        for (int k = 0; k < g->S * g->L; ++k) free(g->F[k]);
        free(g->F);
        g->F = NULL;
    }
}

void orc_group_destroy(orc_group *g) {
    if (!g) return;
    free_setup(g);
    for (int s = 0; s < g->S; ++s) {
        orc_mesh_destroy(g->cpmesh[s]);
        free(g->feat[s]);
        free(g->orig[s]);
    }
    free(g->cpmesh); free((void *)g->data); free(g->feat); free(g->orig); free(g->spacing); free(g->pptr); free(g->pidx);
    free(g->mask); free(g->cp_tri); free(g->labels); free(g->triplets);
    free(g);
}

void orc_group_set_template(orc_group *g, const orc_mesh *tmpl, const double *mask) {
    g->tmpl = tmpl;
    free(g->mask);
    g->mask = NULL;
    if (mask) {
        g->mask = (double *)malloc(sizeof(double) * tmpl->V);
        memcpy(g->mask, mask, sizeof(double) * tmpl->V);
    }
}

/* Initialize, M/DiscreteGroupModel.cpp:141-161 + estimate_triplets :57-75 */
void orc_group_set_controlgrid(orc_group *g, const orc_mesh *cp) {
    g->N = cp->V;
    g->Tc = cp->T;
    free(g->cp_tri);
    g->cp_tri = (int *)malloc(sizeof(int) * 3 * cp->T);
    memcpy(g->cp_tri, cp->tri, sizeof(int) * 3 * cp->T);
    for (int s = 0; s < g->S; ++s) {
        orc_mesh_destroy(g->cpmesh[s]);
        g->cpmesh[s] = orc_mesh_create(cp->xyz, cp->V, cp->tri, cp->T);
    }
    g->subcorr = 0.1 * g->S; /* set_meshes, M/DiscreteGroupCostFunction.h:45 */
    g->T = g->S * cp->T;
    free(g->triplets);
    g->triplets = (int *)malloc(sizeof(int) * 3 * g->T);
    for (int s = 0; s < g->S; ++s)
        for (int t = 0; t < cp->T; ++t) {
            int v[3] = {cp->tri[3 * t] + s * g->N, cp->tri[3 * t + 1] + s * g->N, cp->tri[3 * t + 2] + s * g->N};
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2 - a; ++b)
                    if (v[b] > v[b + 1]) { int q = v[b]; v[b] = v[b + 1]; v[b + 1] = q; }
            memcpy(&g->triplets[3 * (s * cp->T + t)], v, sizeof(v));
        }
}

void orc_group_set_subject(orc_group *g, int s, const orc_mesh *data, const double *feat, int D) {
    g->data[s] = data;
    g->D = D;
    free(g->feat[s]);
    g->feat[s] = (double *)malloc(sizeof(double) * D * data->V);
    memcpy(g->feat[s], feat, sizeof(double) * D * data->V);
    if (!g->orig[s]) {
        g->orig[s] = (double *)malloc(sizeof(double) * 3 * data->V);
        memcpy(g->orig[s], data->xyz, sizeof(double) * 3 * data->V);
    }
}

void orc_group_reset_cpgrid(orc_group *g, int s, const double *xyz) { orc_mesh_set_coords(g->cpmesh[s], xyz, 1); }

void orc_group_set_labels(orc_group *g, const double *labels, int L) {
    free_setup(g); /* sizes depend on L */
    free(g->labels);
    g->L = L;
    g->labels = (double *)malloc(sizeof(double) * 3 * L);
    memcpy(g->labels, labels, sizeof(double) * 3 * L);
}

/* setupCostFunction, M/DiscreteGroupModel.cpp:163-196 */
int orc_group_setup(orc_group *g) {
    const int S = g->S, N = g->N, L = g->L, Vt = g->tmpl->V, D = g->D;
    const double *centre = g->labels; /* m_samples[0] is the sampling-grid centre */
    free_setup(g);
    /* estimate_pairs :37-55 */
    g->P = N * S * (S - 1) / 2;
    g->pairs = (int *)malloc(sizeof(int) * 2 * (g->P > 0 ? g->P : 1));
    {
        orc_octree **trees = (orc_octree **)calloc(S, sizeof(orc_octree *));
        for (int s = 0; s < S; ++s) trees[s] = orc_octree_build(g->cpmesh[s]);
        int pair = 0, bad = 0;
        for (int a = 0; a < S; ++a)
            for (int v = 0; v < N; ++v)
                for (int b = a + 1; b < S; ++b) {
                    int cv = orc_octree_closest_vertex(trees[b], &g->cpmesh[a]->xyz[3 * v]);
                    if (cv < 0) bad = cv;
                    g->pairs[2 * pair] = a * N + v;
                    g->pairs[2 * pair + 1] = b * N + cv;
                    pair++;
                }
        for (int s = 0; s < S; ++s) orc_octree_destroy(trees[s]);
        free(trees);
        if (bad) return bad;
    }
    /* get_spacings :123-139, get_rotations :77-86 */
    g->rot = (double *)malloc(sizeof(double) * 9 * S * N);
    for (int s = 0; s < S; ++s) {
        double mvd;
        g->spacing[s] = (double *)malloc(sizeof(double) * N);
        orc_cp_spacings(g->cpmesh[s], g->spacing[s], &mvd);
        orc_cp_rotations(centre, g->cpmesh[s]->xyz, N, &g->rot[9 * (size_t)s * N]);
    }
    /* get_patch_data :88-121; the reference's loop over the subjects is an OpenMP loop (:92, num_threads(_nthreads)): orc_group_set_threads */
    g->F = (double **)calloc((size_t)S * L, sizeof(double *));
    int failed = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g->nthreads > 0 ? g->nthreads : 1)
    for (int s = 0; s < S; ++s) {
        const orc_mesh *dm = g->data[s];
        long cap = 256L * N * L, n = 0;
        g->pptr[s] = (int *)malloc(sizeof(int) * ((size_t)N * L + 1));
        g->pidx[s] = (int *)malloc(sizeof(int) * cap);
        for (int l = 0; l < L; ++l) {
            double *xyz = (double *)malloc(sizeof(double) * 3 * dm->V);
            memcpy(xyz, dm->xyz, sizeof(double) * 3 * dm->V);
            if (l > 0)
                for (int i = 0; i < dm->V; ++i) { /* rigid rotation of every data vertex by the label's displacement */
                    double R[9];
                    orc_rotation_matrix(centre, &dm->xyz[3 * i], R);
                    m_apply(R, &g->labels[3 * l], &xyz[3 * i]);
                }
            orc_mesh *rot_mesh = orc_mesh_create(xyz, dm->V, dm->tri, dm->T);
            free(xyz);
            double *F = (double *)malloc(sizeof(double) * (size_t)D * Vt);
            int st = orc_metric_resample(rot_mesh, g->feat[s], D, g->tmpl, F);
            orc_mesh_destroy(rot_mesh);
            g->F[(size_t)s * L + l] = F;
            if (st) {
#pragma omp critical
                if (!failed) failed = st;
            }
        }
        /* patch membership: template vertices in range of the rotated control point, ascending id */
        for (int v = 0; v < N; ++v)
            for (int l = 0; l < L; ++l) {
                double rcp[3];
                m_apply(&g->rot[9 * ((size_t)s * N + v)], &g->labels[3 * l], rcp);
                g->pptr[s][v * L + l] = (int)n;
                for (int i = 0; i < Vt; ++i) {
                    double d[3];
                    v_sub(rcp, &g->tmpl->xyz[3 * i], d);
                    if ((2 * ORC_RAD * asin(v_norm(d) / (2 * ORC_RAD))) < g->p.range * g->spacing[s][v]) {
                        if (n == cap) {
                            cap *= 2;
                            g->pidx[s] = (int *)realloc(g->pidx[s], sizeof(int) * cap);
                        }
                        g->pidx[s][n++] = i;
                    }
                }
            }
        g->pptr[s][N * L] = (int)n;
    }
    return failed;
}

void orc_group_set_threads(orc_group *g, int nthreads) { g->nthreads = nthreads; }

/* n evaluations each, spread over threads as Fusion::optimize spreads its pair and triplet loops (I/Fusion/Fusion.h:164, :181): the CPU leg of bench.py's
 * gMSM object */
void orc_group_pairwise_batch(orc_group *g, const int *pair, const int *la, const int *lb, int n, double *out, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) out[i] = orc_group_pairwise(g, pair[i], la[i], lb[i]);
}
void orc_group_triplet_batch(orc_group *g, const int *t, const int *la, const int *lb, const int *lc, int n, double *out, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) out[i] = orc_group_triplet(g, t[i], la[i], lb[i], lc[i]);
}

void orc_group_sizes(const orc_group *g, int *nodes, int *pairs, int *triplets) {
    *nodes = g->S * g->N;
    *pairs = g->P;
    *triplets = g->T;
}
const int *orc_group_pairs(const orc_group *g) { return g->pairs; }
const int *orc_group_triplets(const orc_group *g) { return g->triplets; }

int orc_group_patch(const orc_group *g, int s, int v, int l, int *ids, double *data, int cap) {
    const int beg = g->pptr[s][v * g->L + l], n = g->pptr[s][v * g->L + l + 1] - beg;
    const double *F = g->F[(size_t)s * g->L + l];
    for (int i = 0; i < n && i < cap; ++i) {
        ids[i] = g->pidx[s][beg + i];
        for (int d = 0; d < g->D; ++d) data[(size_t)i * g->D + d] = F[(size_t)d * g->tmpl->V + ids[i]];
    }
    return n;
}

/* DiscreteGroupCostFunction::computePairwiseCost, M/DiscreteGroupCostFunction.cpp:54-98 */
double orc_group_pairwise(orc_group *g, int pair, int la, int lb) {
    const int N = g->N, L = g->L, D = g->D, Vt = g->tmpl->V;
    const int sa = g->pairs[2 * pair] / N, sb = g->pairs[2 * pair + 1] / N;
    const int na = g->pairs[2 * pair] - sa * N, nb = g->pairs[2 * pair + 1] - sb * N;
    const int ba = g->pptr[sa][na * L + la], ea = g->pptr[sa][na * L + la + 1];
    const int bb = g->pptr[sb][nb * L + lb], eb = g->pptr[sb][nb * L + lb + 1];
    const double *FA = g->F[(size_t)sa * L + la], *FB = g->F[(size_t)sb * L + lb];
    int cap = ea - ba, n = 0;
    int *ids = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    for (int i = ba, j = bb; i < ea; ++i) { /* both lists ascend: map::find as a merge */
        while (j < eb && g->pidx[sb][j] < g->pidx[sa][i]) ++j;
        if (j < eb && g->pidx[sb][j] == g->pidx[sa][i]) ids[n++] = g->pidx[sa][i];
    }
    double *A = (double *)malloc(sizeof(double) * 3 * (n > 0 ? n : 1)), *B = A + n, *W = B + n;
    for (int i = 0; i < n; ++i) W[i] = g->mask ? fabs(g->mask[ids[i]]) : 1.0;
    double cost = 0.0;
    /* the reference reads patch_data_A[0].size() even for an empty intersection (undefined); we report NaN */
    if (n == 0) cost = NAN;
    for (int d = 0; d < D && n > 0; ++d) {
        for (int i = 0; i < n; ++i) {
            A[i] = FA[(size_t)d * Vt + ids[i]];
            B[i] = FB[(size_t)d * Vt + ids[i]];
        }
        cost += orc_sim_for_min(g->p.simmeasure, A, B, W, n, g->p.percentile);
    }
    if (n > 0) cost /= D;
    free(ids);
    free(A);
    if (g->p.fixnan && isnan(cost)) return 1e7; /* FIX_NAN */
    return cost;
}

/* DiscreteGroupCostFunction::computeTripletCost, M/DiscreteGroupCostFunction.cpp:26-52 */
double orc_group_triplet(orc_group *g, int t, int la, int lb, int lc) {
    const int N = g->N;
    const int s = t / g->Tc;
    const int *id = &g->triplets[3 * t];
    const int lab[3] = {la, lb, lc};
    double r[3][3], cur[3][3], org[3][3], nd[3], nc[3];
    for (int k = 0; k < 3; ++k) {
        const int v = id[k] - s * N;
        m_apply(&g->rot[9 * (size_t)id[k]], &g->labels[3 * lab[k]], r[k]);
        memcpy(cur[k], &g->cpmesh[s]->xyz[3 * v], sizeof(double) * 3);
        memcpy(org[k], &g->orig[s][3 * v], sizeof(double) * 3);
    }
    orc_tri_normal(r[0], r[1], r[2], nd);
    orc_tri_normal(cur[0], cur[1], cur[2], nc);
    if (v_dot(nd, nc) < 0.0) return ORC_FOLDING;
    double e = orc_triangular_strain(org, r, g->p.mu, g->p.kappa, g->p.k_exp);
    if (g->p.fixnan && isnan(e)) return 1e7;
    return g->subcorr * g->p.lambda * pow(e, g->p.rexp);
}
