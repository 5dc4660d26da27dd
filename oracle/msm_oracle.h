/*
 * msm_oracle.h -- CPU restatement of newMSM's hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle for msm-mi355x.  It restates, in plain C and in the
 * reference's own arithmetic order (FP64, no FMA contraction), the algorithms on the hot path of
 * rbesenczi/newMSM: octree nearest-triangle search, barycentric / adaptive-barycentric resampling
 * and the discrete unary / pairwise / triplet label-cost evaluation.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/libraries/, R/ =
 * msm-newresampler/src, M/ = msm-newmeshreg/src, I/ = msm-newmeshreg/include).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (newmsm_amd/, libmsmhip.so) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, golden vectors or fixtures
 *  and it cannot be built in this image (every translation unit includes FSL's
 * armawrap/newmat.h, miscmaths, newmesh/giftiInterface.h or utils/options.h, none of which exist
 * here and none of which may be replaced by stand-ins).  The oracle is therefore pinned only by
 * (a) the structural statistics SURVEY.md section 8 recorded from the reference's own code
 * (icosphere sizes, ico6 octree node/leaf/reference counts, triangle tests per query, patch sizes,
 * label count) -- checked in tests/test_oracle_pins.py -- and (b) line-by-line citation.
 *
 * Conventions: points are double[N][3] (row-major N x 3), triangles int[T][3], feature matrices
 * D x V row-major (the reference's pvalues[dim][vertex]).
 */
#ifndef MSM_ORACLE_H
#define MSM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_RAD 100.0        /* R/point.h:32 */
#define ORC_EPSILON 1e-8     /* R/point.h:31 */
#define ORC_FOLDING 1e7      /* M/reg_tools.h:30 */
#define ORC_MAX_TRIANGLES 50 /* R/node.h:33 */
#define ORC_MESH_BOUNDS 101  /* R/octree.h:37 */

/* ------------------------------------------------------------------ geometry (R/point.cpp, R/triangle.cpp) */
void   orc_normalize(double v[3]);
int    orc_same_side(const double p1[3], const double p2[3], const double a[3], const double b[3]);
int    orc_point_in_triangle(const double p[3], const double a[3], const double b[3], const double c[3]);
void   orc_project_point(const double vb[3], const double v1[3], const double v2[3], const double v3[3], double out[3]);
double orc_compute_area(const double v0[3], const double v1[3], const double v2[3]);
double orc_dist_to_point(const double x0[3], const double x1[3], const double x2[3], const double x3[3]);
void   orc_tri_normal(const double v0[3], const double v1[3], const double v2[3], double out[3]);
void   orc_calc_barycentric_weights(const double v1[3], const double v2[3], const double v3[3], const double vref[3], double w[3]);
double orc_barycentric_interpolation(const double v1[3], const double v2[3], const double v3[3], const double vref[3],
                                     double a1, double a2, double a3);
void   orc_barycentric_point(const double v1[3], const double v2[3], const double v3[3], const double vref[3],
                             const double a1[3], const double a2[3], const double a3[3], double out[3]);
/* row-major 3x3; returns 0, or -1 for the reference's "angle greater than pi" exception */
int    orc_rotation_matrix(const double ci[3], const double index[3], double R[9]);

/* ------------------------------------------------------------------ icosphere + mesh (R/mesh.cpp) */
/* vertex / triangle counts of make_mesh_from_icosa(order) */
void orc_icosphere_counts(int order, int *V, int *T);
/* make_mesh_from_icosa(order): unit sphere, re-normalised after every subdivision exactly like the
 * reference.  literal_search=1 uses the reference's O(V^2) tolerance search for duplicate midpoints,
 * 0 uses an edge hash (identical output while distinct midpoints are > 1e-8 apart). */
int  orc_icosphere(int order, int literal_search, double *xyz, int *tri);
/* true_rescale (R/mesh.cpp:1210) */
void orc_true_rescale(double *xyz, int V, double rad);
/* Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) without its surface_resample call: the control grid (AoS) retessellated `levels`
 * times with the face neighbourhoods kept and merged as the reference merges them, rescaled to rad; _ANATbaryweights and NEARESTFACES as CSR.
 * axyz 3 x Va (AoS), atri 3 x Ta, w_ptr Va + 1, w_cp / w_val 3 x Va, face_ptr Tc + 1, face_idx Ta (sizes: orc_resample_anatomy_sizes). */
void orc_resample_anatomy_sizes(int N, int Tc, int levels, int *Va, int *Ta);
int  orc_resample_anatomy_grid(const double *cp_xyz, int N, const int *cp_tri, int Tc, int levels, double rad, int literal_search, double *axyz, int *atri,
                               int *w_ptr, int *w_cp, double *w_val, int *face_ptr, int *face_idx);

typedef struct orc_mesh orc_mesh;
/* builds adjacency in Mesh::push_triangle order and caches triangle areas from the given coords */
orc_mesh *orc_mesh_create(const double *xyz, int V, const int *tri, int T);
void      orc_mesh_destroy(orc_mesh *m);
/* Mesh::set_coord for all vertices; refresh_areas=1 mimics a Mesh copy (triangles rebuilt) */
void      orc_mesh_set_coords(orc_mesh *m, const double *xyz, int refresh_areas);
int       orc_mesh_nvertices(const orc_mesh *m);
int       orc_mesh_ntriangles(const orc_mesh *m);
const double *orc_mesh_coords(const orc_mesh *m);
const int    *orc_mesh_triangles(const orc_mesh *m);
/* neighbour / adjacent-triangle lists in reference order (CSR) */
void      orc_mesh_adjacency(const orc_mesh *m, const int **nbr_ptr, const int **nbr, const int **tid_ptr, const int **tid);
double    orc_mesh_vertex_area(const orc_mesh *m, int v);       /* compute_vertex_area R/mesh.cpp:1275 */
double    orc_mesh_max_vd(const orc_mesh *m);                    /* calculate_MaxVD R/mesh.cpp:263 */
double    orc_mesh_mean_vd(const orc_mesh *m);                   /* calculate_MeanVD R/mesh.cpp:279 */

/* ------------------------------------------------------------------ octree (R/octree.cpp, R/node.cpp) */
typedef struct orc_octree orc_octree;
orc_octree *orc_octree_build(const orc_mesh *m);   /* the tree keeps a pointer to m (coords are read live, like shared_ptr<Mpoint>) */
void        orc_octree_destroy(orc_octree *t);
/* stats[0]=nodes [1]=leaves [2]=max depth (root=0) [3]=triangle references [4]=max triangles in a leaf */
void        orc_octree_stats(const orc_octree *t, long stats[5]);
/* get_closest_triangle: returns triangle id, -1 if pt is outside the root box, -2 if nothing found
 * (the two cases where the reference throws).  *ntests (optional) accumulates distance_to_triangle calls. */
int         orc_octree_closest_triangle(const orc_octree *t, const double pt[3], long *ntests);
int         orc_octree_closest_vertex(const orc_octree *t, const double pt[3]);

/* ------------------------------------------------------------------ resampler (R/resampler.cpp) */
/* get_barycentric_weights (:142-167): per query the hit triangle, its 3 vertex ids (triangle order)
 * and calc_barycentric_weights; returns 0 or the (negative) status of the first failing query */
int orc_barycentric_weights(const orc_octree *t, const double *q, int N, int *tri_id, int *vid, double *w);
/* same query, raw weights of barycentric_interpolation (query not projected) */
int orc_barycentric_weights_raw(const orc_octree *t, const double *q, int N, int *tri_id, int *vid, double *w);
/* get_adaptive_barycentric_weights (:72-140), serial order.  excl (length in-mesh V) may be NULL.
 * Output CSR rows sorted by column (std::map order).  Call with col==NULL to size: returns nnz. */
long orc_adaptive_barycentric_weights(const orc_mesh *in_mesh, const orc_mesh *new_mesh, const double *excl,
                                      int *row_ptr, int *col, double *val);
/* barycentric_data_interpolation (:30-70) given the CSR weights; data D x Vin -> out D x Vnew */
void orc_apply_weights(const int *row_ptr, const int *col, const double *val, int Nnew,
                       const double *data, int D, int Vin, const double *excl, double *out);
/* metric_resample = adaptive weights + apply (excl NULL) */
int  orc_metric_resample(const orc_mesh *in_mesh, const double *data, int D, const orc_mesh *new_mesh, double *out);
/* the same with EXCL (values on in_mesh, 0 = excluded); excl_out (optional, V(new)) = the mask the reference writes back */
int  orc_metric_resample_excl(const orc_mesh *in_mesh, const double *data, int D, const orc_mesh *new_mesh, const double *excl, double *out,
                              double *excl_out);
/* create_exclusion (R/mesh.cpp:1257-1273) */
void orc_create_exclusion(const double *data, int D, int V, double thrl, double thru, double *excl);
/* nearest_neighbour_interpolation with EXCL (:232-258) */
int  orc_nearest_neighbour_excl(const orc_mesh *orig, const double *data, int D, const double *q, int N, const double *excl, double *out,
                                double *excl_out);
/* sphere_project_warp (:311-328): sphere[N] moved through from->to */
int  orc_sphere_project_warp(double *sphere, int N, const orc_mesh *from, const double *to_xyz);
/* smooth_data (:168-230): Gaussian smoothing over the geodesic neighbourhood (see orc_resample.c for the index quirks) */
int  orc_smooth_data(const orc_mesh *orig, const double *data, int D, const orc_mesh *sphLow, double sigma, const double *excl, double *out,
                     double *excl_out);
/* nearest_neighbour_interpolation (:232-258) without exclusion */
int  orc_nearest_neighbour(const orc_mesh *orig, const double *data, int D, const double *q, int N, double *out);

/* ------------------------------------------------------------------ mesh utilities around the path (M/reg_tools.cpp) */
/* unfold (:131-178) on the mesh's coordinates, in place.  Returns the number of passes that moved vertices (0: not
 * folded), -1 if a vertex has no triangle.  *first_folded (optional): folded vertices found by the first pass. */
int  orc_unfold(orc_mesh *m, double rad, int *first_folded);
/* variance_normalise (:804-843): data D x V in place, excl (length V, > 0 keeps) may be NULL */
void orc_variance_normalise(double *data, int D, int V, const double *excl);

/* ------------------------------------------------------------------ similarity (M/similarities.cpp) */
double orc_corr_weighted(const double *A, const double *B, const double *w, int n);
double orc_ssd_weighted(const double *A, const double *B, const double *w, int n);
double orc_dice(const double *A, const double *B, int n, double percentile);
double orc_gendice(const double *A, const double *B, int n, double percentile);
/* get_sim_for_min (M/similarities.h:48-58); sim 1=SSD 2=corr 4=DICE 5=genDICE */
double orc_sim_for_min(int sim, const double *A, const double *B, const double *w, int n, double percentile);

/* ------------------------------------------------------------------ strain (M/reg_tools.cpp) */
double orc_triangle_strain(const double A2d[3][2], const double B2d[3][2], double mu, double kappa, double k_exp);
double orc_triangular_strain(const double o[3][3], const double f[3][3], double mu, double kappa, double k_exp);

/* ------------------------------------------------------------------ discrete model host logic (M/DiscreteModel.cpp) */
/* Initialize :77-89: per-CP max neighbour spacing and MVDmax */
void orc_cp_spacings(const orc_mesh *cp, double *maxsep, double *mvdmax);
/* Initialize_sampling_grid + label_sampling_grid :110-190.  sg = sampling grid (radius 100).
 * Returns counts; samples/barycentres must hold up to 3*maxn doubles. abs_is_int selects the
 * int abs() overload at :170 (SURVEY hard parts). */
int  orc_label_sampling_grid(const orc_mesh *sg, double max_dist, int abs_is_int, int *centroid,
                             double *samples, int *nsamples, double *barycentres, int *nbary, int maxn);
/* rescale_sampling_grid :192-214 (scale is read and updated) */
void orc_rescale_sampling_grid(const double *samples, int n, double *scale, double *labels);
/* get_rotations :310-319: ROT[k] = R(centre -> CP[k]), row-major 9 doubles per CP */
void orc_cp_rotations(const double centre[3], const double *cp, int N, double *rot);
/* estimate_triplets :291-308 / estimate_pairs :271-289 */
void orc_estimate_triplets(const orc_mesh *cp, int *triplets);
int  orc_estimate_pairs(const orc_mesh *cp, int *pairs /* may be NULL to count */);

/* ------------------------------------------------------------------ cost function (M/DiscreteCostFunction.cpp) */
typedef struct orc_cost orc_cost;
enum { ORC_UNIVARIATE = 0, ORC_MULTIVARIATE = 1, ORC_PATCHWISE = 2, ORC_HO_UNIVARIATE = 3, ORC_HO_MULTIVARIATE = 4 };

typedef struct {
    int    kind;          /* ORC_* */
    int    simmeasure;    /* 1 SSD, 2 corr, 4 DICE, 5 genDICE */
    int    rmode;         /* regularisermode: 1 pairwise, 2/3 strain */
    double lambda, mu, kappa, k_exp, rexp, range, percentile;
} orc_cost_params;

orc_cost *orc_cost_create(const orc_cost_params *p);
void      orc_cost_destroy(orc_cost *c);
/* set_meshes: target mesh + its octree, source mesh (with triangles: resample_weights needs them),
 * CP grid mesh; ORIG source / original CP grid are captured at this call like set_meshes() does. */
void orc_cost_set_meshes(orc_cost *c, const orc_mesh *target, const orc_octree *ttree,
                         const orc_mesh *source, const orc_mesh *cpgrid);
/* set_anatomical + set_anatomical_neighbourhood (M/DiscreteCostFunction.h:160-170) for regoption 4/5: the
 * anatomical-resolution sphere _TARGEThi with its octree, _aTARGET coordinates (AoS, the sphere's vertex ids),
 * _aSOURCE, _ANATbaryweights as CSR over _aSOURCE vertices (control point ids ascending), NEARESTFACES as CSR over
 * triplets.  Pointers are borrowed. */
void orc_cost_set_anatomical(orc_cost *c, const orc_mesh *sphere, const orc_octree *sphere_tree, const double *atarget_xyz,
                             const orc_mesh *asource, const int *w_ptr, const int *w_cp, const double *w_val, const int *face_ptr,
                             const int *face_idx);
void orc_cost_reset_source(orc_cost *c, const orc_mesh *source);
void orc_cost_reset_cpgrid(orc_cost *c, const orc_mesh *cpgrid);
/* featurespace: input (source) D x Nsrc and reference (target) D x Ntgt */
void orc_cost_set_features(orc_cost *c, const double *src_feat, const double *ref_feat, int D);
/* cost-function weighting rows x Nsrc (rows == 1 or D); NULL = the reference's all-ones default */
void orc_cost_set_cfweight(orc_cost *c, const double *w, int rows);
void orc_cost_set_spacings(orc_cost *c, const double *maxsep, double mvdmax);
void orc_cost_set_labels(orc_cost *c, const double *labels, int L, const double *rot /* Ncp x 9 */);
void orc_cost_set_triplets(orc_cost *c, const int *triplets, int T);
void orc_cost_set_pairs(orc_cost *c, const int *pairs, int P);
/* initialize() + get_source_data(): patches (or per-triangle bins for HO) + AbsoluteWeights */
int  orc_cost_get_source_data(orc_cost *c);
/* patch listing: ptr has (Ncp or Ntri)+1 entries */
void orc_cost_patches(const orc_cost *c, const int **ptr, const int **idx, int *ngroups);
const double *orc_cost_absolute_weights(const orc_cost *c);

double orc_cost_unary(orc_cost *c, int node, int label);
void   orc_cost_unary_table(orc_cost *c, double *U /* L x N, label*N+node */);
void   orc_cost_unary_table_omp(orc_cost *c, double *U, int nthreads);
double orc_cost_triplet(orc_cost *c, int triplet, int la, int lb, int lc);
/* one fusion move's octets, I/Fusion/Fusion.h:181-196: E[8*t + k], k = 000..111 (A,B,C), OpenMP over triplets */
void   orc_cost_triplet_octets(orc_cost *c, const int *labeling, int label, double *E, int nthreads);
/* computeTripletCosts (:245-253): out[(t - t0) x L x L x L] */
void   orc_cost_triplet_table(orc_cost *c, int t0, int t1, double *out);
double orc_cost_pairwise(orc_cost *c, int pair, int la, int lb);
void   orc_cost_pairwise_table(orc_cost *c, double *out /* P x L x L, [(pair * L + labelB) * L + labelA], :228-234 */);
double orc_cost_total(orc_cost *c, const int *labeling, double parts[3]);
/* number of patch point samples evaluated so far (for throughput accounting) */
long   orc_cost_samples(const orc_cost *c);

/* ------------------------------------------------------------------ groupwise (gMSM): M/DiscreteGroupModel.cpp, M/DiscreteGroupCostFunction.cpp */
typedef struct orc_group orc_group;
typedef struct {
    int    simmeasure, fixnan;
    double lambda, mu, kappa, k_exp, rexp, range;
    double percentile; /* DICE threshold rank, sparsesimkernel::percentile (M/similarities.h:68; 0.75 by default) */
} orc_group_params;
orc_group *orc_group_create(const orc_group_params *p, int num_subjects);
void       orc_group_destroy(orc_group *g);
/* set_meshspace(target_space, ...) :  the template mesh every subject is resampled to */
void orc_group_set_template(orc_group *g, const orc_mesh *tmpl, const double *mask /* V_tmpl or NULL */);
/* Initialize(controlgrid) :141-161: every subject starts from this control grid */
void orc_group_set_controlgrid(orc_group *g, const orc_mesh *cp);
/* data mesh of one subject (m_datameshes[s], reset_meshspace) with its features D x V; the first call per subject also
 * captures _ORIG_MESHES[s] (set_meshes) */
void orc_group_set_subject(orc_group *g, int s, const orc_mesh *data, const double *feat, int D);
void orc_group_reset_cpgrid(orc_group *g, int s, const double *xyz);
void orc_group_set_labels(orc_group *g, const double *labels, int L);
/* setupCostFunction :163-196: estimate_pairs, get_spacings, get_rotations, get_patch_data */
int  orc_group_setup(orc_group *g);
void orc_group_sizes(const orc_group *g, int *nodes, int *pairs, int *triplets);
const int *orc_group_pairs(const orc_group *g);
const int *orc_group_triplets(const orc_group *g);
/* patch (subject, control point, label): template vertex ids (ascending) and their D values; returns the count */
int  orc_group_patch(const orc_group *g, int s, int v, int l, int *ids, double *data, int cap);
double orc_group_pairwise(orc_group *g, int pair, int la, int lb);   /* DiscreteGroupCostFunction.cpp:54-98 */
double orc_group_triplet(orc_group *g, int t, int la, int lb, int lc); /* :26-52 */
void orc_group_set_threads(orc_group *g, int nthreads); /* get_patch_data's OpenMP loop over the subjects, M/DiscreteGroupModel.cpp:92 */
/* n evaluations each over OpenMP threads, as Fusion::optimize's pair and triplet loops run them (I/Fusion/Fusion.h:164,181) */
void orc_group_pairwise_batch(orc_group *g, const int *pair, const int *la, const int *lb, int n, double *out, int nthreads);
void orc_group_triplet_batch(orc_group *g, const int *t, const int *la, const int *lb, const int *lc, int n, double *out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
