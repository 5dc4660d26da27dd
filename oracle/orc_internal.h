/* orc_internal.h -- shared internals of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see msm_oracle.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msm_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Point operators, R/point.cpp:174-197 */
static inline double v_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v_sub(const double a[3], const double b[3], double o[3]) {
    o[0] = a[0] - b[0];
    o[1] = a[1] - b[1];
    o[2] = a[2] - b[2];
}
/* operator*(Point,Point): note the Y component is written v2.X*v1.Z - v2.Z*v1.X */
static inline void v_cross(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = b[0] * a[2] - b[2] * a[0];
    double z = a[0] * b[1] - b[0] * a[1];
    o[0] = x;
    o[1] = y;
    o[2] = z;
}
static inline double v_norm(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
/* operator*(Matrix,Point), R/point.cpp:207-213 (row-major M) */
static inline void m_apply(const double M[9], const double v[3], double o[3]) {
    double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x;
    o[1] = y;
    o[2] = z;
}

double orc_tri_calc_area(const double v0[3], const double v1[3], const double v2[3]);

struct orc_mesh {
    int V, T;
    double *xyz;   /* V x 3 */
    int *tri;      /* T x 3 */
    double *tarea; /* Triangle::area cached at construction (R/triangle.cpp:27-43) */
    int *nbr_ptr, *nbr; /* Mpoint::nID in push order */
    int *tid_ptr, *tid; /* Mpoint::trID in push order */
};

typedef struct orc_node {
    struct orc_node *child[8]; /* index 4*i+2*j+k <-> children[i][j][k] */
    struct orc_node *parent;
    double bounds[3][3];
    int is_leaf;
    int *tris;
    int ntris, cap;
} orc_node;

struct orc_octree {
    const orc_mesh *mesh;
    orc_node *root;
};

#endif
