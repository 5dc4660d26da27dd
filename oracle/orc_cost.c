/*
 * orc_cost.c -- oracle restatement of newMSM's discrete label-cost evaluation
 * (M/DiscreteCostFunction.cpp, M/similarities.cpp, M/reg_tools.cpp strain, M/DiscreteModel.cpp glue).
 * TEST INFRASTRUCTURE ONLY (see msm_oracle.h).  Parity unpinned: pinned by structural statistics only.
 *
 * NEWMAT (Armadillo behind FSL's armawrap) is not available, so 2x2 inverse / 3x3 determinant are
 * written out with the textbook cofactor formulas; they can differ from the reference's LAPACK
 * route in the last bits.
 */
#include "orc_internal.h"

/* ------------------------------------------------------------------ similarity */

/* sparsesimkernel::corr (weighted), M/similarities.cpp:129-158 */
double orc_corr_weighted(const double *A, const double *B, const double *w, int n) {
    double prod = 0.0, varA = 0.0, varB = 0.0, meanA = 0.0, meanB = 0.0, sum = 0.0;
    for (int i = 0; i < n; ++i) sum += w[i];
    for (int i = 0; i < n; ++i) {
        meanA += w[i] * A[i];
        meanB += w[i] * B[i];
    }
    if (sum > 0.0) {
        meanA /= sum;
        meanB /= sum;
    }
    for (int s = 0; s < n; ++s) {
        prod += w[s] * (A[s] - meanA) * (B[s] - meanB);
        varA += w[s] * (A[s] - meanA) * (A[s] - meanA);
        varB += w[s] * (B[s] - meanB) * (B[s] - meanB);
    }
    if (sum > 0.0) {
        prod /= sum;
        varA /= sum;
        varB /= sum;
    }
    if (varA == 0.0 || varB == 0.0) return 0.0;
    return prod / (sqrt(varA) * sqrt(varB));
}

/* sparsesimkernel::SSD (weighted), M/similarities.cpp:179-188 */
double orc_ssd_weighted(const double *A, const double *B, const double *w, int n) {
    double prod = 0.0;
    for (int i = 0; i < n; ++i) prod += w[i] * (A[i] - B[i]) * (A[i] - B[i]);
    return sqrt(prod) / n;
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static void dice_counts(const double *A, const double *B, int n, double percentile, int *sa, int *sb, int *common) {
    int idx = (int)floor(percentile * n);
    double *As = (double *)malloc(sizeof(double) * n), *Bs = (double *)malloc(sizeof(double) * n);
    memcpy(As, A, sizeof(double) * n);
    memcpy(Bs, B, sizeof(double) * n);
    qsort(As, n, sizeof(double), cmp_double);
    qsort(Bs, n, sizeof(double), cmp_double);
    int size_A = n, size_B = n, c = 0;
    for (int i = 0; i < n; ++i) {
        int ov = 1;
        if (A[i] < As[idx]) {
            size_A--;
            ov = 0;
        }
        if (B[i] < Bs[idx]) {
            size_B--;
            ov = 0;
        }
        c += ov;
    }
    *sa = size_A;
    *sb = size_B;
    *common = c;
    free(As);
    free(Bs);
}

/* sparsesimkernel::DICE, M/similarities.cpp:201-226 */
double orc_dice(const double *A, const double *B, int n, double percentile) {
    int sa, sb, c;
    dice_counts(A, B, n, percentile, &sa, &sb, &c);
    return 1.0 - ((2.0 * c) / (sa + sb));
}

/* sparsesimkernel::genDICE, M/similarities.cpp:228-253 */
double orc_gendice(const double *A, const double *B, int n, double percentile) {
    int sa, sb, c;
    dice_counts(A, B, n, percentile, &sa, &sb, &c);
    return 1.0 - (2.0 * (((c / pow(sb, 2))) / ((sa + sb) / pow(sb, 2))));
}

/* sparsesimkernel::get_sim_for_min, M/similarities.h:48-58 */
double orc_sim_for_min(int sim, const double *A, const double *B, const double *w, int n, double percentile) {
    if (sim == 1) return orc_ssd_weighted(A, B, w, n);
    if (sim == 2) return 1 - (1 + orc_corr_weighted(A, B, w, n)) * 0.5;
    if (sim == 4) return orc_dice(A, B, n, percentile);
    if (sim == 5) return orc_gendice(A, B, n, percentile);
    return NAN;
}

/* ------------------------------------------------------------------ strain */

static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* triangle_strain, M/reg_tools.cpp:551-646 (principal-strain side output omitted) */
double orc_triangle_strain(const double AA[3][2], const double BB[3][2], double MU, double KAPPA, double k_exp) {
    double c0 = AA[1][0] - AA[0][0], c1 = AA[1][1] - AA[0][1];
    double c4 = AA[2][0] - AA[0][0], c5 = AA[2][1] - AA[0][1];
    double c0c = BB[1][0] - BB[0][0], c1c = BB[1][1] - BB[0][1];
    double c4c = BB[2][0] - BB[0][0], c5c = BB[2][1] - BB[0][1];
    /* Edges = [c0 c4; c1 c5], edges = [c0c c4c; c1c c5c]; F = edges * Edges^-1 */
    double det = c0 * c5 - c4 * c1;
    double i00 = c5 / det, i01 = -c4 / det, i10 = -c1 / det, i11 = c0 / det;
    double F00 = c0c * i00 + c4c * i10, F01 = c0c * i01 + c4c * i11;
    double F10 = c1c * i00 + c5c * i10, F11 = c1c * i01 + c5c * i11;
    /* F3D = [F 0; 0 0 1]; F3D_2 = F3D^T F3D */
    double G[9] = {F00 * F00 + F10 * F10, F00 * F01 + F10 * F11, 0,
                   F01 * F00 + F11 * F10, F01 * F01 + F11 * F11, 0,
                   0, 0, 1};
    double I1 = G[0] + G[4] + G[8];
    double I3 = det3(G);
    double J = sqrt(I3);
    double I1st = (I1 - 1.0) / J;
    double R;
    if (I1st <= 2)
        R = 1.0;
    else
        R = 0.5 * (I1st + sqrt(I1st * I1st - 4));
    double Rs = pow(R, k_exp), Js = pow(J, k_exp);
    return 0.5 * (MU * (Rs + 1.0 / Rs - 2) + KAPPA * (Js + 1.0 / Js - 2));
}

/* calculate_tri(const Point&), M/reg_tools.cpp:267-313 */
static void calc_tri(const double a[3], double e1[3], double e2[3]) {
    double b[3] = {1.0, 0.0, 0.0}, c[3];
    v_cross(a, b, c);
    double len = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    if (len == 0.0) {
        b[0] = 0.0;
        b[1] = 1.0;
        b[2] = 0.0;
        v_cross(a, b, c);
        len = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    }
    len = sqrt(len);
    if (len == 0.0) len = 1;
    e1[0] = c[0] / len;
    e1[1] = c[1] / len;
    e1[2] = c[2] / len;
    v_cross(a, c, b);
    len = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    if (len == 0) len = 1;
    e2[0] = b[0] / len;
    e2[1] = b[1] / len;
    e2[2] = b[2] / len;
}

/* calculate_triangular_strain(Triangle,Triangle,...), M/reg_tools.cpp:698-743 */
double orc_triangular_strain(const double o[3][3], const double f[3][3], double mu, double kappa, double k_exp) {
    double nO[3], nF[3], e1[3], e2[3], t1[3], t2[3];
    orc_tri_normal(o[0], o[1], o[2], nO);
    orc_tri_normal(f[0], f[1], f[2], nF);
    calc_tri(nO, e1, e2);
    calc_tri(nF, t1, t2);
    /* TRANS = [e1 e2 nO] as columns; swap the first two columns if det < 0 (:712-719) */
    double TR[9] = {e1[0], e2[0], nO[0], e1[1], e2[1], nO[1], e1[2], e2[2], nO[2]};
    const double *c1 = e1, *c2 = e2;
    if (det3(TR) < 0) {
        c1 = e2;
        c2 = e1;
        double TS[9] = {e2[0], e1[0], nO[0], e2[1], e1[1], nO[1], e2[2], e1[2], nO[2]};
        memcpy(TR, TS, sizeof(TR));
    }
    /* the second test re-checks TRANS (not TRANS2), :721 -- kept as is */
    const double *d1 = t1, *d2 = t2;
    if (det3(TR) < 0) {
        d1 = t2;
        d2 = t1;
    }
    double A2[3][2], B2[3][2];
    for (int i = 0; i < 3; ++i) { /* ORIG2D = ORIG3D * TRANS: row i = (v.c1, v.c2, v.n) */
        A2[i][0] = o[i][0] * c1[0] + o[i][1] * c1[1] + o[i][2] * c1[2];
        A2[i][1] = o[i][0] * c2[0] + o[i][1] * c2[1] + o[i][2] * c2[2];
        B2[i][0] = f[i][0] * d1[0] + f[i][1] * d1[1] + f[i][2] * d1[2];
        B2[i][1] = f[i][0] * d2[0] + f[i][1] * d2[1] + f[i][2] * d2[2];
    }
    return orc_triangle_strain(A2, B2, mu, kappa, k_exp);
}

/* ------------------------------------------------------------------ discrete model host logic */

/* NonLinearSRegDiscreteModel::Initialize, M/DiscreteModel.cpp:72-89 */
void orc_cp_spacings(const orc_mesh *cp, double *maxsep, double *mvdmax) {
    for (int k = 0; k < cp->V; ++k) {
        maxsep[k] = 0;
        for (int j = cp->nbr_ptr[k]; j < cp->nbr_ptr[k + 1]; ++j) {
            double d[3];
            v_sub(&cp->xyz[3 * k], &cp->xyz[3 * cp->nbr[j]], d);
            double dist = 2 * ORC_RAD * asin(v_norm(d) / (2 * ORC_RAD));
            if (dist > maxsep[k]) maxsep[k] = dist;
        }
    }
    *mvdmax = orc_mesh_max_vd(cp);
}

/* std::map<double,Point> */
typedef struct {
    int n, cap;
    double *key;
    double *pt;
} dmap;

static void dmap_set(dmap *m, double k, const double p[3]) {
    int i = 0;
    while (i < m->n && m->key[i] < k) ++i;
    if (i < m->n && m->key[i] == k) {
        memcpy(&m->pt[3 * i], p, sizeof(double) * 3);
        return;
    }
    if (m->n == m->cap) {
        m->cap = m->cap ? 2 * m->cap : 32;
        m->key = (double *)realloc(m->key, sizeof(double) * m->cap);
        m->pt = (double *)realloc(m->pt, sizeof(double) * 3 * m->cap);
    }
    memmove(&m->key[i + 1], &m->key[i], sizeof(double) * (m->n - i));
    memmove(&m->pt[3 * (i + 1)], &m->pt[3 * i], sizeof(double) * 3 * (m->n - i));
    m->key[i] = k;
    memcpy(&m->pt[3 * i], p, sizeof(double) * 3);
    m->n++;
}

/* Initialize_sampling_grid + label_sampling_grid, M/DiscreteModel.cpp:110-190 */
int orc_label_sampling_grid(const orc_mesh *sg, double dist, int abs_is_int, int *centroid_out,
                            double *samples_out, int *nsamples, double *bary_out, int *nbary, int maxn) {
    int centroid = -1;
    for (int i = 0; i < sg->V; ++i)
        if (sg->nbr_ptr[i + 1] - sg->nbr_ptr[i] == 6) {
            centroid = i;
            break;
        }
    if (centroid < 0) centroid = 0; /* m_centroid keeps its default 0 */
    if (centroid_out) *centroid_out = centroid;
    dmap samples = {0, 0, NULL, NULL}, barys = {0, 0, NULL, NULL};
    char *found = (char *)calloc(sg->V, 1), *found_tr = (char *)calloc(sg->T, 1);
    int *get = (int *)malloc(sizeof(int) * sg->V * 8), *next = (int *)malloc(sizeof(int) * sg->V * 8);
    int nget = 0, nnext = 0;
    const double *centre = &sg->xyz[3 * centroid];
    get[nget++] = centroid;
    while (nget > 0) {
        for (int g = 0; g < nget; ++g) {
            int gn = get[g];
            for (int j = sg->nbr_ptr[gn]; j < sg->nbr_ptr[gn + 1]; ++j) {
                int v = sg->nbr[j];
                double d[3];
                v_sub(&sg->xyz[3 * v], centre, d);
                double distance = v_norm(d);
                if (distance <= dist && !found[v] && v != centroid) {
                    dmap_set(&samples, distance, &sg->xyz[3 * v]);
                    next[nnext++] = v;
                    found[v] = 1;
                }
            }
            for (int j = sg->tid_ptr[gn]; j < sg->tid_ptr[gn + 1]; ++j) {
                int t = sg->tid[j];
                const double *v1 = &sg->xyz[3 * sg->tri[3 * t]], *v2 = &sg->xyz[3 * sg->tri[3 * t + 1]], *v3 = &sg->xyz[3 * sg->tri[3 * t + 2]];
                double bary[3] = {(v1[0] + v2[0] + v3[0]) / 3, (v1[1] + v2[1] + v3[1]) / 3, (v1[2] + v2[2] + v3[2]) / 3};
                orc_normalize(bary);
                for (int a = 0; a < 3; ++a) bary[a] = bary[a] * ORC_RAD;
                double bc[3];
                v_sub(bary, centre, bc);
                double distance = v_norm(bc);
                if (distance <= dist && v_norm(bc) > 0 && !found_tr[t]) {
                    for (int e = 0; e < barys.n; ++e) {
                        double oc[3];
                        v_sub(&barys.pt[3 * e], centre, oc);
                        double q = 1 - (v_dot(bc, oc) / (v_norm(bc) * v_norm(oc)));
                        double aq = abs_is_int ? (double)abs((int)q) : fabs(q);
                        if (aq < 1e-2) found_tr[t] = 1;
                    }
                    if (!found_tr[t]) dmap_set(&barys, distance, bary);
                    found_tr[t] = 1;
                }
            }
        }
        memcpy(get, next, sizeof(int) * nnext);
        nget = nnext;
        nnext = 0;
    }
    int ns = 1 + samples.n, nb = 1 + barys.n, rc = 0;
    if (ns > maxn || nb > maxn)
        rc = -1;
    else {
        memcpy(samples_out, centre, sizeof(double) * 3);
        memcpy(samples_out + 3, samples.pt, sizeof(double) * 3 * samples.n);
        memcpy(bary_out, centre, sizeof(double) * 3);
        memcpy(bary_out + 3, barys.pt, sizeof(double) * 3 * barys.n);
    }
    *nsamples = ns;
    *nbary = nb;
    free(samples.key); free(samples.pt); free(barys.key); free(barys.pt);
    free(found); free(found_tr); free(get); free(next);
    return rc;
}

/* rescale_sampling_grid, M/DiscreteModel.cpp:192-214; samples[0] is the centre */
void orc_rescale_sampling_grid(const double *samples, int n, double *scale, double *labels) {
    const double *centre = samples;
    if (*scale >= 0.25) {
        for (int i = 0; i < n; ++i) {
            double p[3];
            for (int a = 0; a < 3; ++a) p[a] = centre[a] + (centre[a] - samples[3 * i + a]) * (*scale);
            orc_normalize(p);
            for (int a = 0; a < 3; ++a) labels[3 * i + a] = p[a] * 100;
        }
    } else {
        *scale = 1;
        memcpy(labels, samples, sizeof(double) * 3 * n);
    }
    *scale *= 0.8;
}

/* get_rotations, M/DiscreteModel.cpp:310-319 */
void orc_cp_rotations(const double centre[3], const double *cp, int N, double *rot) {
    for (int k = 0; k < N; ++k) orc_rotation_matrix(centre, &cp[3 * k], &rot[9 * k]);
}

static void sort3(int v[3]) {
    int t;
    if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
    if (v[1] > v[2]) { t = v[1]; v[1] = v[2]; v[2] = t; }
    if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
}

/* estimate_triplets, M/DiscreteModel.cpp:291-308 */
void orc_estimate_triplets(const orc_mesh *cp, int *triplets) {
    for (int i = 0; i < cp->T; ++i) {
        int v[3] = {cp->tri[3 * i], cp->tri[3 * i + 1], cp->tri[3 * i + 2]};
        sort3(v);
        memcpy(&triplets[3 * i], v, sizeof(v));
    }
}

/* estimate_pairs, M/DiscreteModel.cpp:271-289 */
int orc_estimate_pairs(const orc_mesh *cp, int *pairs) {
    int pair = 0;
    for (int i = 0; i < cp->V; ++i)
        for (int j = cp->nbr_ptr[i]; j < cp->nbr_ptr[i + 1]; ++j)
            if (cp->nbr[j] > i) {
                if (pairs) {
                    pairs[2 * pair] = i;
                    pairs[2 * pair + 1] = cp->nbr[j];
                }
                pair++;
            }
    return pair;
}

/* ------------------------------------------------------------------ cost function */

struct orc_cost {
    orc_cost_params p;
    const orc_mesh *target;
    const orc_octree *ttree;
    const orc_mesh *source; /* _SOURCE (connectivity + current coords) */
    double *orig_xyz;       /* _ORIG coords captured at set_meshes */
    int norig;
    const orc_mesh *cpgrid; /* _CPgrid */
    double *ocp_xyz;        /* _oCPgrid coords */
    int D;
    const double *src_feat, *ref_feat;
    const double *cfw;
    int cfw_rows;
    double *maxsep;
    double mvdmax;
    double *labels;
    int L;
    double *rot;
    int *triplets, T;
    int *pairs, P;
    /* get_source_data products */
    int ngroups;
    int *grp_ptr, *grp_idx;
    double *absw;
    long samples;
    /* scratch */
    double *tgt; /* target data for one group: maxgroup x D */
    int maxgroup;
    /* anatomical regularisation (regoption 4/5): set_anatomical + set_anatomical_neighbourhood,
     * M/DiscreteCostFunction.h:160-170 */
    const orc_mesh *asphere;      /* _TARGEThi: the anatomical-resolution sphere (aICO) */
    const orc_octree *anattree;   /* Octree(_TARGEThi), M/DiscreteCostFunction.h:162-163 */
    const double *atarget_xyz;    /* _aTARGET coordinates, vertex ids of asphere */
    const orc_mesh *asource;      /* _aSOURCE */
    const int *aw_ptr, *aw_cp;    /* _ANATbaryweights: per _aSOURCE vertex (control point id, weight), ascending ids */
    const double *aw_val;
    const int *af_ptr, *af_idx;   /* NEARESTFACES: per triplet the _aSOURCE faces */
};

orc_cost *orc_cost_create(const orc_cost_params *p) {
    orc_cost *c = (orc_cost *)calloc(1, sizeof(orc_cost));
    c->p = *p;
    return c;
}

void orc_cost_destroy(orc_cost *c) {
    if (!c) return;
    free(c->orig_xyz); free(c->ocp_xyz); free(c->maxsep); free(c->labels); free(c->rot);
    free(c->triplets); free(c->pairs); free(c->grp_ptr); free(c->grp_idx); free(c->absw); free(c->tgt);
    free(c);
}

/* set_meshes, M/DiscreteCostFunction.h:196-198 */
void orc_cost_set_meshes(orc_cost *c, const orc_mesh *target, const orc_octree *ttree, const orc_mesh *source, const orc_mesh *cpgrid) {
    c->target = target;
    c->ttree = ttree;
    c->source = source;
    c->cpgrid = cpgrid;
    free(c->orig_xyz);
    c->norig = source->V;
    c->orig_xyz = (double *)malloc(sizeof(double) * 3 * source->V);
    memcpy(c->orig_xyz, source->xyz, sizeof(double) * 3 * source->V);
    free(c->ocp_xyz);
    c->ocp_xyz = (double *)malloc(sizeof(double) * 3 * cpgrid->V);
    memcpy(c->ocp_xyz, cpgrid->xyz, sizeof(double) * 3 * cpgrid->V);
}
void orc_cost_reset_source(orc_cost *c, const orc_mesh *source) { c->source = source; }
void orc_cost_reset_cpgrid(orc_cost *c, const orc_mesh *cpgrid) { c->cpgrid = cpgrid; }
void orc_cost_set_features(orc_cost *c, const double *src_feat, const double *ref_feat, int D) {
    c->src_feat = src_feat;
    c->ref_feat = ref_feat;
    c->D = D;
}
void orc_cost_set_cfweight(orc_cost *c, const double *w, int rows) {
    c->cfw = w;
    c->cfw_rows = w ? rows : 0;
}
void orc_cost_set_spacings(orc_cost *c, const double *maxsep, double mvdmax) {
    free(c->maxsep);
    c->maxsep = (double *)malloc(sizeof(double) * c->cpgrid->V);
    memcpy(c->maxsep, maxsep, sizeof(double) * c->cpgrid->V);
    c->mvdmax = mvdmax;
}
void orc_cost_set_labels(orc_cost *c, const double *labels, int L, const double *rot) {
    free(c->labels);
    free(c->rot);
    c->L = L;
    c->labels = (double *)malloc(sizeof(double) * 3 * L);
    memcpy(c->labels, labels, sizeof(double) * 3 * L);
    c->rot = (double *)malloc(sizeof(double) * 9 * c->cpgrid->V);
    memcpy(c->rot, rot, sizeof(double) * 9 * c->cpgrid->V);
}
void orc_cost_set_triplets(orc_cost *c, const int *triplets, int T) {
    free(c->triplets);
    c->T = T;
    c->triplets = (int *)malloc(sizeof(int) * 3 * (T > 0 ? T : 1));
    memcpy(c->triplets, triplets, sizeof(int) * 3 * T);
}
void orc_cost_set_pairs(orc_cost *c, const int *pairs, int P) {
    free(c->pairs);
    c->P = P;
    c->pairs = (int *)malloc(sizeof(int) * 2 * (P > 0 ? P : 1));
    memcpy(c->pairs, pairs, sizeof(int) * 2 * P);
}

/* cfweight(row d (0-based), vertex i): all ones unless supplied (M/mesh_registration.cpp:234-238) */
static double cfw_at(const orc_cost *c, int d, int i) {
    if (!c->cfw) return 1.0;
    return c->cfw[(long)d * c->source->V + i];
}
static int cfw_nrows(const orc_cost *c) { return c->cfw ? c->cfw_rows : 1; }

/* within_controlpt_range, M/DiscreteCostFunction.cpp:102-107 */
static int within_range(const orc_cost *c, int cp, int src) {
    double d[3];
    v_sub(&c->cpgrid->xyz[3 * cp], &c->source->xyz[3 * src], d);
    return (2 * ORC_RAD * asin(v_norm(d) / (2 * ORC_RAD))) < c->p.range * c->maxsep[cp];
}

/* resample_weights, M/DiscreteCostFunction.cpp:303-323 */
static int resample_weights(orc_cost *c) {
    int ns = c->source->V;
    double *mw = (double *)malloc(sizeof(double) * ns);
    for (int k = 0; k < ns; ++k) {
        double best = -DBL_MAX;
        for (int j = 0; j < cfw_nrows(c); ++j)
            if (cfw_at(c, j, k) > best) best = cfw_at(c, j, k);
        mw[k] = best;
    }
    free(c->absw);
    c->absw = (double *)malloc(sizeof(double) * c->cpgrid->V);
    int st = orc_metric_resample(c->source, mw, 1, c->cpgrid, c->absw);
    free(mw);
    return st;
}

/* initialize() + get_source_data() of the five cost-function classes:
 * Univariate :334-351, Multivariate :393-416, Patchwise :629-650 group by control point (range test);
 * HOUnivariate :468-485, HOMultivariate :541-563 group by closest control-grid triangle. */
int orc_cost_get_source_data(orc_cost *c) {
    int ns = c->source->V, ncp = c->cpgrid->V;
    int ho = (c->p.kind == ORC_HO_UNIVARIATE || c->p.kind == ORC_HO_MULTIVARIATE);
    int ng = ho ? c->cpgrid->T : ncp;
    free(c->grp_ptr);
    free(c->grp_idx);
    c->ngroups = ng;
    c->grp_ptr = (int *)calloc(ng + 1, sizeof(int));
    int *grp_of = NULL;
    if (ho) {
        orc_octree *cpt = orc_octree_build(c->cpgrid);
        grp_of = (int *)malloc(sizeof(int) * ns);
        for (int i = 0; i < ns; ++i) {
            grp_of[i] = orc_octree_closest_triangle(cpt, &c->source->xyz[3 * i], NULL);
            if (grp_of[i] < 0) {
                orc_octree_destroy(cpt);
                free(grp_of);
                return grp_of[i];
            }
            c->grp_ptr[grp_of[i] + 1]++;
        }
        orc_octree_destroy(cpt);
        for (int g = 0; g < ng; ++g) c->grp_ptr[g + 1] += c->grp_ptr[g];
        c->grp_idx = (int *)malloc(sizeof(int) * (ns > 0 ? ns : 1));
        int *fill = (int *)calloc(ng, sizeof(int));
        for (int i = 0; i < ns; ++i) c->grp_idx[c->grp_ptr[grp_of[i]] + fill[grp_of[i]]++] = i;
        free(fill);
        free(grp_of);
    } else {
        long cap = 128L * ncp, n = 0;
        c->grp_idx = (int *)malloc(sizeof(int) * cap);
        for (int k = 0; k < ncp; ++k) {
            c->grp_ptr[k] = (int)n;
            for (int i = 0; i < ns; ++i)
                if (within_range(c, k, i)) {
                    if (n == cap) {
                        cap *= 2;
                        c->grp_idx = (int *)realloc(c->grp_idx, sizeof(int) * cap);
                    }
                    c->grp_idx[n++] = i;
                }
        }
        c->grp_ptr[ncp] = (int)n;
    }
    c->maxgroup = 1;
    for (int g = 0; g < ng; ++g)
        if (c->grp_ptr[g + 1] - c->grp_ptr[g] > c->maxgroup) c->maxgroup = c->grp_ptr[g + 1] - c->grp_ptr[g];
    free(c->tgt);
    c->tgt = (double *)malloc(sizeof(double) * (long)c->maxgroup * (c->D > 0 ? c->D : 1));
    return resample_weights(c);
}

void orc_cost_patches(const orc_cost *c, const int **ptr, const int **idx, int *ngroups) {
    *ptr = c->grp_ptr;
    *idx = c->grp_idx;
    *ngroups = c->ngroups;
}
const double *orc_cost_absolute_weights(const orc_cost *c) { return c->absw; }
long orc_cost_samples(const orc_cost *c) { return c->samples; }

/* the shared body of the get_target_data() variants: closest target triangle of p and the
 * barycentric interpolation of all D reference features (query NOT projected), e.g. :361-375 */
static int sample_target(const orc_cost *c, const double p[3], double *out /* D */, long *nsamples) {
    int tr = orc_octree_closest_triangle(c->ttree, p, NULL);
    if (tr < 0) return tr;
    const orc_mesh *m = c->target;
    const int *n = &m->tri[3 * tr];
    const double *v0 = &m->xyz[3 * n[0]], *v1 = &m->xyz[3 * n[1]], *v2 = &m->xyz[3 * n[2]];
    for (int d = 0; d < c->D; ++d) {
        const double *ref = &c->ref_feat[(long)d * m->V];
        out[d] = orc_barycentric_interpolation(v0, v1, v2, p, ref[n[0]], ref[n[1]], ref[n[2]]);
    }
    ++*nsamples;
    return 0;
}

/* similarity of one group given its target data laid out [point][D] in c->tgt */
static double group_similarity(const orc_cost *c, int g, const double *tgt) {
    int beg = c->grp_ptr[g], n = c->grp_ptr[g + 1] - beg, D = c->D, ns = c->source->V;
    const int *idx = &c->grp_idx[beg];
    int kind = c->p.kind;
    if (kind == ORC_UNIVARIATE || kind == ORC_HO_UNIVARIATE) {
        /* _sourcedata/_weights hold feature row 1 and cfweight row 1, :343-347 / :477-481 */
        double *A = (double *)malloc(sizeof(double) * 3 * (n > 0 ? n : 1)), *B = A + n, *W = B + n;
        for (int i = 0; i < n; ++i) {
            A[i] = c->src_feat[idx[i]];
            B[i] = tgt[(long)i * D];
            W[i] = cfw_nrows(c) >= 1 ? cfw_at(c, 0, idx[i]) : 1.0;
        }
        double s = orc_sim_for_min(c->p.simmeasure, A, B, W, n, c->p.percentile);
        free(A);
        return s;
    }
    if (kind == ORC_MULTIVARIATE || kind == ORC_HO_MULTIVARIATE) {
        /* mean over patch points of a D-long feature-vector similarity, :449-457 / :607-617 */
        double cost = 0.0;
        double *A = (double *)malloc(sizeof(double) * 3 * D), *B = A + D, *W = B + D;
        for (int i = 0; i < n; ++i) {
            for (int d = 0; d < D; ++d) {
                A[d] = c->src_feat[(long)d * ns + idx[i]];
                B[d] = tgt[(long)i * D + d];
                W[d] = cfw_nrows(c) >= d + 1 ? cfw_at(c, d, idx[i]) : 1.0;
            }
            cost += orc_sim_for_min(c->p.simmeasure, A, B, W, D, c->p.percentile);
        }
        free(A);
        if (n > 0) cost /= n;
        return cost;
    }
    /* patchwise: per channel a patch similarity, averaged over channels, :681-691 */
    double cost = 0.0;
    double *A = (double *)malloc(sizeof(double) * 3 * (n > 0 ? n : 1)), *B = A + n, *W = B + n;
    for (int d = 0; d < D; ++d) {
        for (int i = 0; i < n; ++i) {
            A[i] = c->src_feat[(long)d * ns + idx[i]];
            B[i] = tgt[(long)i * D + d];
            W[i] = cfw_nrows(c) >= 1 ? cfw_at(c, 0, idx[i]) : 1.0;
        }
        cost += orc_sim_for_min(c->p.simmeasure, A, B, W, n, c->p.percentile);
    }
    free(A);
    return cost / D;
}

/* computeUnaryCost: Univariate :378-383, Multivariate :444-458, Patchwise :680-692; HO variants return 0 */
static double unary_eval(const orc_cost *c, int node, int label, double *tgt, long *nsamples) {
    if (c->p.kind == ORC_HO_UNIVARIATE || c->p.kind == ORC_HO_MULTIVARIATE) return 0.0;
    double newcp[3], R[9];
    m_apply(&c->rot[9 * node], &c->labels[3 * label], newcp);
    orc_rotation_matrix(&c->cpgrid->xyz[3 * node], newcp, R);
    int beg = c->grp_ptr[node], n = c->grp_ptr[node + 1] - beg;
    for (int i = 0; i < n; ++i) {
        double p[3];
        m_apply(R, &c->source->xyz[3 * c->grp_idx[beg + i]], p);
        if (sample_target(c, p, &tgt[(long)i * c->D], nsamples) < 0) return NAN;
    }
    return c->absw[node] * group_similarity(c, node, tgt);
}

double orc_cost_unary(orc_cost *c, int node, int label) { return unary_eval(c, node, label, c->tgt, &c->samples); }

/* computeUnaryCosts, M/DiscreteCostFunction.cpp:236-243 */
void orc_cost_unary_table(orc_cost *c, double *U) {
    int N = c->cpgrid->V;
    for (int j = 0; j < c->L; ++j)
        for (int k = 0; k < N; ++k) U[(long)j * N + k] = orc_cost_unary(c, k, j);
}

/* computeUnaryCosts with the reference's OpenMP loop (parallel over nodes inside the label loop,
 * M/DiscreteCostFunction.cpp:238-242); used by bench.py's cpu_baseline leg */
void orc_cost_unary_table_omp(orc_cost *c, double *U, int nthreads) {
    int N = c->cpgrid->V;
    long total = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
    {
        double *tgt = (double *)malloc(sizeof(double) * (long)c->maxgroup * (c->D > 0 ? c->D : 1));
        long mine = 0;
        for (int j = 0; j < c->L; ++j) {
#pragma omp for
            for (int k = 0; k < N; ++k) U[(long)j * N + k] = unary_eval(c, k, j, tgt, &mine);
        }
        total += mine;
        free(tgt);
    }
    c->samples += total;
}

/* triplet_likelihood of the HO classes: get_target_data :487-518 / :565-599, likelihood :520-531 / :601-618 */
static double triplet_likelihood(const orc_cost *c, int t, const double n0[3], const double n1[3], const double n2[3], double *tgt, long *nsamples) {
    if (c->p.kind != ORC_HO_UNIVARIATE && c->p.kind != ORC_HO_MULTIVARIATE) return 0.0;
    const int *id = &c->triplets[3 * t];
    const double *cp0 = &c->cpgrid->xyz[3 * id[0]], *cp1 = &c->cpgrid->xyz[3 * id[1]], *cp2 = &c->cpgrid->xyz[3 * id[2]];
    int beg = c->grp_ptr[t], n = c->grp_ptr[t + 1] - beg;
    for (int i = 0; i < n; ++i) {
        double sp[3], tmp[3];
        orc_project_point(&c->source->xyz[3 * c->grp_idx[beg + i]], cp0, cp1, cp2, sp);
        orc_barycentric_point(cp0, cp1, cp2, sp, n0, n1, n2, tmp);
        orc_normalize(tmp);
        for (int a = 0; a < 3; ++a) tmp[a] *= ORC_RAD;
        if (sample_target(c, tmp, &tgt[(long)i * c->D], nsamples) < 0) return NAN;
    }
    double sim = group_similarity(c, t, tgt);
    return (c->absw[id[0]] + c->absw[id[1]] + c->absw[id[2]]) / 3.0 * sim;
}

void orc_cost_set_anatomical(orc_cost *c, const orc_mesh *sphere, const orc_octree *sphere_tree, const double *atarget_xyz,
                             const orc_mesh *asource, const int *w_ptr, const int *w_cp, const double *w_val, const int *face_ptr,
                             const int *face_idx) {
    c->asphere = sphere;
    c->anattree = sphere_tree;
    c->atarget_xyz = atarget_xyz;
    c->asource = asource;
    c->aw_ptr = w_ptr;
    c->aw_cp = w_cp;
    c->aw_val = w_val;
    c->af_ptr = face_ptr;
    c->af_idx = face_idx;
}

/* deform_anatomy, M/DiscreteCostFunction.cpp:255-301, for one vertex of one anatomical face (the reference's
 * moved/transformed maps only cache this per evaluation).  id / moved: the triplet's control points and their
 * proposed positions; a control point outside the triplet enters as a default-constructed Point (0,0,0) through
 * std::map::operator[] (:269).  Returns 0, or -1 when the search fails (the reference then substitutes a zero
 * triangle, :272-278, and its weights are NaN). */
static int deform_vertex(const orc_cost *c, int tindex, const int id[3], double moved[3][3], double out[3]) {
    double np[3] = {0.0, 0.0, 0.0};
    for (int j = c->aw_ptr[tindex]; j < c->aw_ptr[tindex + 1]; ++j) {
        const int cp = c->aw_cp[j];
        const double w = c->aw_val[j];
        double v[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < 3; ++k)
            if (id[k] == cp) memcpy(v, moved[k], sizeof(v));
        for (int a = 0; a < 3; ++a) np[a] += v[a] * w;
    }
    const int t = orc_octree_closest_triangle(c->anattree, np, NULL);
    if (t < 0) {
        out[0] = out[1] = out[2] = NAN;
        return -1;
    }
    const int *n = &c->asphere->tri[3 * t];
    double w[3];
    orc_calc_barycentric_weights(&c->asphere->xyz[3 * n[0]], &c->asphere->xyz[3 * n[1]], &c->asphere->xyz[3 * n[2]], np, w);
    /* std::map<int,double> weight: iterated in ascending vertex id (:290-291) */
    int order[3] = {0, 1, 2};
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (n[order[b]] < n[order[a]]) {
                int tmp = order[a];
                order[a] = order[b];
                order[b] = tmp;
            }
    out[0] = out[1] = out[2] = 0.0;
    for (int q = 0; q < 3; ++q) {
        const int vid = n[order[q]];
        for (int a = 0; a < 3; ++a) out[a] += c->atarget_xyz[3 * vid + a] * w[order[q]];
    }
    return 0;
}

/* computeTripletCost, M/DiscreteCostFunction.cpp:135-188 (regoption 2/3: spherical strain; 4/5: anatomical strain) */
static double triplet_eval(const orc_cost *c, int t, int la, int lb, int lc, double *tgt, long *nsamples) {
    const int *id = &c->triplets[3 * t];
    double r[3][3], cur[3][3], org[3][3], nd[3], nc[3];
    m_apply(&c->rot[9 * id[0]], &c->labels[3 * la], r[0]);
    m_apply(&c->rot[9 * id[1]], &c->labels[3 * lb], r[1]);
    m_apply(&c->rot[9 * id[2]], &c->labels[3 * lc], r[2]);
    for (int k = 0; k < 3; ++k) {
        memcpy(cur[k], &c->cpgrid->xyz[3 * id[k]], sizeof(double) * 3);
        memcpy(org[k], &c->orig_xyz[3 * id[k]], sizeof(double) * 3);
    }
    orc_tri_normal(r[0], r[1], r[2], nd);
    orc_tri_normal(cur[0], cur[1], cur[2], nc);
    if (v_dot(nd, nc) < 0.0) return ORC_FOLDING * c->p.lambda;
    double likelihood = triplet_likelihood(c, t, r[0], r[1], r[2], tgt, nsamples);
    double cost = 0.0;
    if (c->p.rmode == 2 || c->p.rmode == 3) {
        cost = orc_triangular_strain(org, r, c->p.mu, c->p.kappa, c->p.k_exp);
    } else if ((c->p.rmode == 4 || c->p.rmode == 5) && c->asource) { /* :169-182 */
        const int beg = c->af_ptr[t], nf = c->af_ptr[t + 1] - beg;
        for (int n = 0; n < nf; ++n) {
            const int *fv = &c->asource->tri[3 * c->af_idx[beg + n]];
            double o[3][3], d[3][3];
            for (int k = 0; k < 3; ++k) {
                memcpy(o[k], &c->asource->xyz[3 * fv[k]], sizeof(double) * 3);
                deform_vertex(c, fv[k], id, r, d[k]);
            }
            cost += orc_triangular_strain(o, d, c->p.mu, c->p.kappa, c->p.k_exp);
        }
        cost = cost / (double)nf;
    } else {
        return NAN;
    }
    return likelihood + c->p.lambda * pow(cost, c->p.rexp);
}

double orc_cost_triplet(orc_cost *c, int t, int la, int lb, int lc) { return triplet_eval(c, t, la, lb, lc, c->tgt, &c->samples); }

/* the eight computeTripletCost calls per triplet of one fusion move, I/Fusion/Fusion.h:181-196 (the reference's
 * OpenMP loop over the triplets; every thread evaluates whole triplets): E[8*t + k], k = 000..111 with bit order
 * (A,B,C), 0 = labeling[node], 1 = label */
void orc_cost_triplet_octets(orc_cost *c, const int *labeling, int label, double *E, int nthreads) {
    long total = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
    {
        double *tgt = (double *)malloc(sizeof(double) * (long)(c->maxgroup > 0 ? c->maxgroup : 1) * (c->D > 0 ? c->D : 1));
        long mine = 0;
#pragma omp for schedule(static)
        for (int t = 0; t < c->T; ++t) {
            const int *id = &c->triplets[3 * t];
            const int a = labeling[id[0]], b = labeling[id[1]], cc = labeling[id[2]];
            double *e = &E[8 * (long)t];
            e[0] = triplet_eval(c, t, a, b, cc, tgt, &mine);
            e[1] = triplet_eval(c, t, a, b, label, tgt, &mine);
            e[2] = triplet_eval(c, t, a, label, cc, tgt, &mine);
            e[3] = triplet_eval(c, t, a, label, label, tgt, &mine);
            e[4] = triplet_eval(c, t, label, b, cc, tgt, &mine);
            e[5] = triplet_eval(c, t, label, b, label, tgt, &mine);
            e[6] = triplet_eval(c, t, label, label, cc, tgt, &mine);
            e[7] = triplet_eval(c, t, label, label, label, tgt, &mine);
        }
        total += mine;
        free(tgt);
    }
    c->samples += total;
}

/* computePairwiseCost, M/DiscreteCostFunction.cpp:190-226, restated without mutating the CP grid */
double orc_cost_pairwise(orc_cost *c, int pair, int la, int lb) {
    int a = c->pairs[2 * pair], b = c->pairs[2 * pair + 1];
    const orc_mesh *g = c->cpgrid;
    const double *v0 = &g->xyz[3 * a], *v1 = &g->xyz[3 * b];
    double na[3], nb[3], R1[9], R2[9], Rd[9];
    m_apply(&c->rot[9 * a], &c->labels[3 * la], na);
    m_apply(&c->rot[9 * b], &c->labels[3 * lb], nb);
    orc_rotation_matrix(v0, na, R1);
    orc_rotation_matrix(v1, nb, R2);
    for (int r = 0; r < 3; ++r) /* R_diff = R1^T R2 */
        for (int q = 0; q < 3; ++q) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += R1[3 * k + r] * R2[3 * k + q];
            Rd[3 * r + q] = s;
        }
    double trace = Rd[0] + Rd[4] + Rd[8];
    const double theta_MVD = 2 * asin(c->mvdmax / (2 * ORC_RAD));
    const double theta = acos((trace - 1) / 2);
    double cost = 0.0;
    if (fabs(1 - (trace - 1) / 2) > ORC_EPSILON) {
        /* folding test over the triangles adjacent to the FIRST node only, :205-211 */
        for (int j = g->tid_ptr[a]; j < g->tid_ptr[a + 1]; ++j) {
            int t = g->tid[j];
            double no[3], nn[3], p[3][3];
            const int *n = &g->tri[3 * t];
            orc_tri_normal(&c->ocp_xyz[3 * n[0]], &c->ocp_xyz[3 * n[1]], &c->ocp_xyz[3 * n[2]], no);
            for (int k = 0; k < 3; ++k) {
                const double *src = (n[k] == a) ? na : (n[k] == b) ? nb : &g->xyz[3 * n[k]];
                memcpy(p[k], src, sizeof(double) * 3);
            }
            orc_tri_normal(p[0], p[1], p[2], nn);
            if (v_dot(no, nn) < 0.0) return ORC_FOLDING;
        }
        if (c->p.rexp == 1)
            cost = c->p.lambda * ((sqrt(2) * theta) / theta_MVD);
        else
            cost = c->p.lambda * pow(((sqrt(2) * theta) / theta_MVD), c->p.rexp);
    }
    return cost;
}

/* evaluateTotalCostSum, M/DiscreteCostFunction.cpp:55-77 */
double orc_cost_total(orc_cost *c, const int *labeling, double parts[3]) {
    double u = 0.0, pw = 0.0, tc = 0.0;
    for (int i = 0; i < c->cpgrid->V; ++i) u += orc_cost_unary(c, i, labeling[i]);
    for (int p = 0; p < c->P; ++p) pw += orc_cost_pairwise(c, p, labeling[c->pairs[2 * p]], labeling[c->pairs[2 * p + 1]]);
    for (int t = 0; t < c->T; ++t)
        tc += orc_cost_triplet(c, t, labeling[c->triplets[3 * t]], labeling[c->triplets[3 * t + 1]], labeling[c->triplets[3 * t + 2]]);
    if (parts) {
        parts[0] = u;
        parts[1] = pw;
        parts[2] = tc;
    }
    return u + pw + tc;
}

/* computePairwiseCosts, M/DiscreteCostFunction.cpp:228-234: paircosts[(pair * L + labelB) * L + labelA] = computePairwiseCost(pair, labelA, labelB).
 * The reference runs this serially because its computePairwiseCost moves the control grid in place; the restatement is functional, so the
 * pairs are spread over threads. */
void orc_cost_pairwise_table(orc_cost *c, double *out) {
    const int L = c->L;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < c->P; ++p)
        for (int j = 0; j < L; ++j)
            for (int k = 0; k < L; ++k) out[((size_t)p * L + k) * L + j] = orc_cost_pairwise(c, p, j, k);
}

/* computeTripletCosts, M/DiscreteCostFunction.cpp:245-253: tcosts[t][a][b][c] for t0 <= t < t1 */
void orc_cost_triplet_table(orc_cost *c, int t0, int t1, double *out) {
    const int L = c->L;
    long total = 0;
#pragma omp parallel reduction(+ : total)
    {
        double *tgt = (double *)malloc(sizeof(double) * (long)(c->maxgroup > 0 ? c->maxgroup : 1) * (c->D > 0 ? c->D : 1));
        long mine = 0;
#pragma omp for schedule(dynamic, 4)
        for (int t = t0; t < t1; ++t)
            for (int a = 0; a < L; ++a)
                for (int b = 0; b < L; ++b)
                    for (int l = 0; l < L; ++l) out[(((size_t)(t - t0) * L + a) * L + b) * L + l] = triplet_eval(c, t, a, b, l, tgt, &mine);
        total += mine;
        free(tgt);
    }
    c->samples += total;
}
