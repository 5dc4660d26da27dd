/*
 * abi_demo.c -- drives libmsmhip through include/msmhip.h from plain C (no Python, no HIP headers): the way a
 * newMSM maintainer's adapter would.  `abi_demo host` needs no GPU; `abi_demo gpu` runs a tiny
 * resampling + unary-table pass and checks invariants.  Exit code 0 = all checks passed.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msmhip.h"

#define CHECK(cond, ...)                                      \
    do {                                                      \
        if (!(cond)) {                                        \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                     \
            fprintf(stderr, " [%s]\n", msm_last_error());     \
            return 1;                                         \
        }                                                     \
    } while (0)

static int host_checks(void) {
    int32_t V, T;
    CHECK(msm_abi_version() == MSM_ABI_VERSION, "abi version");
    CHECK(msm_icosphere_counts(4, &V, &T) == MSM_OK && V == 2562 && T == 5120, "ico4 counts");
    CHECK(msm_icosphere_counts(42, &V, &T) == MSM_ERR_INVALID && strlen(msm_last_error()) > 0, "bad order must fail with a message");
    msm_icosphere_counts(3, &V, &T);
    double *xyz = malloc(sizeof(double) * 3 * V);
    int32_t *tri = malloc(sizeof(int32_t) * 3 * T);
    CHECK(msm_icosphere(3, 100.0, xyz, tri) == MSM_OK, "icosphere");
    for (int i = 0; i < V; ++i) {
        double r = sqrt(xyz[i] * xyz[i] + xyz[V + i] * xyz[V + i] + xyz[2 * V + i] * xyz[2 * V + i]);
        CHECK(fabs(r - 100.0) < 1e-9, "radius of vertex %d", i);
    }
    double *maxsep = malloc(sizeof(double) * V), mvd;
    CHECK(msm_cp_spacings(xyz, tri, V, T, maxsep, &mvd) == MSM_OK && mvd > 0, "spacings");
    int32_t *trip = malloc(sizeof(int32_t) * 3 * T);
    CHECK(msm_estimate_triplets(tri, T, trip) == MSM_OK, "triplets");
    for (int t = 0; t < T; ++t) CHECK(trip[3 * t] < trip[3 * t + 1] && trip[3 * t + 1] < trip[3 * t + 2], "triplet %d not ascending", t);
    CHECK(msm_estimate_pairs(tri, V, T, NULL) == 3 * T / 2, "edge count");
    free(xyz); free(tri); free(maxsep); free(trip);
    return 0;
}

static int gpu_checks(void) {
    CHECK(msm_device_count() > 0, "no GPU visible");
    msm_ctx *ctx = msm_ctx_create(0);
    CHECK(ctx != NULL, "context");
    int32_t V, T, Vc, Tc;
    msm_icosphere_counts(4, &V, &T);
    msm_icosphere_counts(2, &Vc, &Tc);
    double *xyz = malloc(sizeof(double) * 3 * V), *cxyz = malloc(sizeof(double) * 3 * Vc);
    int32_t *tri = malloc(sizeof(int32_t) * 3 * T), *ctri = malloc(sizeof(int32_t) * 3 * Tc);
    msm_icosphere(4, 100.0, xyz, tri);
    msm_icosphere(2, 100.0, cxyz, ctri);
    msm_mesh *target = msm_mesh_create(ctx, xyz, V, tri, T), *source = msm_mesh_create(ctx, xyz, V, tri, T), *cp = msm_mesh_create(ctx, cxyz, Vc, ctri, Tc);
    CHECK(target && source && cp, "meshes");
    /* query the control points against the target: weights are a partition of unity, ids valid */
    int32_t *tid = malloc(sizeof(int32_t) * Vc), *vid = malloc(sizeof(int32_t) * 3 * Vc);
    double *w = malloc(sizeof(double) * 3 * Vc);
    CHECK(msm_query_triangles(target, cxyz, Vc, tid, vid, w, MSM_WEIGHTS_PROJECTED) == MSM_OK, "query");
    for (int i = 0; i < Vc; ++i) {
        CHECK(tid[i] >= 0 && tid[i] < T, "triangle id");
        CHECK(fabs(w[i] + w[Vc + i] + w[2 * Vc + i] - 1.0) < 1e-12, "weights of query %d", i);
        /* ico2 vertices are ico4 vertices: the hit triangle must own the vertex with weight ~1 */
        double wmax = fmax(w[i], fmax(w[Vc + i], w[2 * Vc + i]));
        CHECK(wmax > 1.0 - 1e-9, "query %d should sit on a vertex", i);
    }
    /* a point outside the root box reports the reference's exception */
    double far[3] = {0.0, 0.0, 250.0};
    int32_t one;
    CHECK(msm_query_triangles(target, far, 1, &one, NULL, NULL, MSM_WEIGHTS_RAW) == MSM_ERR_OUTSIDE && one == MSM_ERR_OUTSIDE, "outside point");
    CHECK(strstr(msm_last_error(), "bounding box") != NULL, "error text");
    /* identical feature on both sides and the zero displacement label: correlation 1 -> cost 0 */
    double *feat = malloc(sizeof(double) * V);
    for (int i = 0; i < V; ++i) feat[i] = sin(xyz[i] / 20.0) + cos(xyz[V + i] / 15.0);
    CHECK(msm_mesh_set_features(target, feat, 1) == MSM_OK, "target features");
    msm_cost_params p = {MSM_COST_UNIVARIATE, 2, 3, 0, 0.1, 0.1, 10.0, 2.0, 2.0, 1.0, 0.75};
    msm_cost *c = msm_cost_create(ctx, &p);
    CHECK(c != NULL, "cost");
    CHECK(msm_cost_get_source_data(c) == MSM_ERR_STATE, "call order must be enforced");
    CHECK(msm_cost_set_meshes(c, target, source, cp) == MSM_OK, "set_meshes");
    CHECK(msm_cost_set_source_features(c, feat, 1) == MSM_OK, "source features");
    double *maxsep = malloc(sizeof(double) * Vc), mvd;
    msm_cp_spacings(cxyz, ctri, Vc, Tc, maxsep, &mvd);
    CHECK(msm_cost_set_spacings(c, maxsep, mvd) == MSM_OK, "spacings");
    enum { CAP = 64 };
    double samples[3 * CAP], bary[3 * CAP], labels[3 * CAP];
    int32_t ns, nb;
    CHECK(msm_label_sampling_grid(4, 0.5 * mvd, 0, CAP, samples, &ns, bary, &nb) == MSM_OK && ns > 1, "labels");
    for (int k = 0; k < 3; ++k) memcpy(labels + k * ns, samples + k * CAP, sizeof(double) * ns);  /* compact 3 x CAP -> 3 x ns */
    const double centre[3] = {labels[0], labels[ns], labels[2 * ns]};
    double *rot = malloc(sizeof(double) * 9 * Vc);
    CHECK(msm_cp_rotations(centre, cxyz, Vc, rot) == MSM_OK, "rotations");
    CHECK(msm_cost_set_labels(c, labels, ns, rot) == MSM_OK, "set_labels");
    CHECK(msm_cost_get_source_data(c) == MSM_OK, "get_source_data");
    double *U = malloc(sizeof(double) * ns * Vc);
    CHECK(msm_cost_unary_table(c, U) == MSM_OK, "unary table");
    for (int n = 0; n < Vc; ++n) CHECK(fabs(U[n]) < 1e-9, "label 0 (no move) of node %d costs %.3e, expected 0", n, U[n]);
    double worst = 0;
    for (int i = 0; i < ns * Vc; ++i) {
        CHECK(U[i] == U[i] && U[i] >= -1e-12 && U[i] <= 1.0 + 1e-9, "cost %d out of [0,1]", i);
        if (U[i] > worst) worst = U[i];
    }
    CHECK(worst > 1e-6, "moving away from the identity must cost something");
    printf("abi_demo gpu: %d queries, unary table %d x %d, max cost %.4f\n", Vc, ns, Vc, worst);
    msm_cost_destroy(c);
    msm_mesh_destroy(target); msm_mesh_destroy(source); msm_mesh_destroy(cp);
    msm_ctx_destroy(ctx);
    free(xyz); free(cxyz); free(tri); free(ctri); free(tid); free(vid); free(w); free(feat); free(maxsep); free(rot); free(U);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_demo host|gpu\n"); return 2; }
    if (host_checks()) return 1;
    if (!strcmp(argv[1], "gpu") && gpu_checks()) return 1;
    printf("abi_demo %s: ok\n", argv[1]);
    return 0;
}
