/*
 * orc_octree.c -- oracle restatement of newMSM's octree nearest-triangle search.
 * TEST INFRASTRUCTURE ONLY (see msm_oracle.h).  Parity unpinned: pinned by structural statistics only
 * (ico6: 14 281 nodes / 12 496 leaves / depth 6 / 176 096 references).
 *
 * A pointer tree grown by inserting the triangles one at a time, as R/octree.cpp does; the shape of
 * the tree (and with it the candidate set of every query) depends on that insertion history.
 */
#include "orc_internal.h"

/* Node::Node(lower,upper), R/node.cpp:42-54 and Node::Node(), :30-40 */
static orc_node *node_new(void) {
    orc_node *n = (orc_node *)calloc(1, sizeof(orc_node));
    n->is_leaf = 1;
    n->cap = ORC_MAX_TRIANGLES;
    n->tris = (int *)malloc(sizeof(int) * n->cap);
    return n;
}

static void node_free(orc_node *n) {
    if (!n) return;
    for (int c = 0; c < 8; ++c) node_free(n->child[c]);
    free(n->tris);
    free(n);
}

static void node_push(orc_node *n, int t) {
    if (n->ntris == n->cap) {
        n->cap *= 2;
        n->tris = (int *)realloc(n->tris, sizeof(int) * n->cap);
    }
    n->tris[n->ntris++] = t;
}

/* Node::make_children, R/node.cpp:84-106 */
static void node_make_children(orc_node *n) {
    n->is_leaf = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                orc_node *c = node_new();
                int oct[3] = {i, j, k};
                for (int a = 0; a < 3; ++a) {
                    c->bounds[a][0] = n->bounds[a][oct[a]];
                    c->bounds[a][2] = n->bounds[a][oct[a] + 1];
                    c->bounds[a][1] = (c->bounds[a][0] + c->bounds[a][2]) / 2.0;
                }
                c->parent = n;
                n->child[4 * i + 2 * j + k] = c;
            }
}

/* Node::can_contain, R/node.cpp:108-116 */
static int node_can_contain(const orc_node *n, const double lo[3], const double hi[3]) {
    for (int a = 0; a < 3; ++a)
        if (hi[a] < n->bounds[a][0] || lo[a] > n->bounds[a][2]) return 0;
    return 1;
}

/* Node::contains_point, R/node.cpp:58-68 */
static int node_contains_point(const orc_node *n, const double p[3]) {
    for (int a = 0; a < 3; ++a) {
        if (p[a] < n->bounds[a][0]) return 0;
        if (p[a] > n->bounds[a][2]) return 0;
    }
    return 1;
}

static void tri_aabb(const orc_mesh *m, int t, double lo[3], double hi[3]) {
    const double *v0 = &m->xyz[3 * m->tri[3 * t]];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = v0[a];
    for (int k = 1; k < 3; ++k) {
        const double *v = &m->xyz[3 * m->tri[3 * t + k]];
        for (int a = 0; a < 3; ++a) {
            if (v[a] < lo[a]) lo[a] = v[a];
            if (v[a] > hi[a]) hi[a] = v[a];
        }
    }
}

/* Octree::add_triangle, R/octree.cpp:65-141 */
static void add_triangle(const orc_mesh *m, orc_node *node, int t, const double lo[3], const double hi[3]) {
    if (node->is_leaf) {
        node_push(node, t);
        int num = node->ntris;
        if (num >= ORC_MAX_TRIANGLES) {
            int total_size = 0, num_split = 0;
            for (int i = 0; i < num; ++i) {
                double tlo[3], thi[3];
                tri_aabb(m, node->tris[i], tlo, thi);
                int split = 8;
                /* containing_oct (R/node.cpp:70-82): point[i] < bounds[i][1] */
                for (int d = 0; d < 3; ++d)
                    if ((tlo[d] < node->bounds[d][1]) == (thi[d] < node->bounds[d][1])) split >>= 1;
                total_size += split;
                if (split != 8) ++num_split;
            }
            if (num_split > 0 && total_size < 3 * num) {
                node_make_children(node);
                for (int d = 0; d < num; ++d) {
                    double tlo[3], thi[3];
                    int tt = node->tris[d];
                    tri_aabb(m, tt, tlo, thi);
                    for (int c = 0; c < 8; ++c)
                        if (node_can_contain(node->child[c], tlo, thi)) add_triangle(m, node->child[c], tt, tlo, thi);
                }
                node->ntris = 0; /* clear_triangles */
            }
        }
    } else {
        for (int c = 0; c < 8; ++c)
            if (node_can_contain(node->child[c], lo, hi)) add_triangle(m, node->child[c], t, lo, hi);
    }
}

/* Octree::Octree + initialize_tree, R/octree.cpp:31-63 */
orc_octree *orc_octree_build(const orc_mesh *m) {
    orc_octree *t = (orc_octree *)calloc(1, sizeof(orc_octree));
    t->mesh = m;
    t->root = node_new();
    for (int a = 0; a < 3; ++a) {
        t->root->bounds[a][0] = -ORC_MESH_BOUNDS;
        t->root->bounds[a][2] = ORC_MESH_BOUNDS;
        t->root->bounds[a][1] = (t->root->bounds[a][0] + t->root->bounds[a][2]) / 2.0;
    }
    for (int i = 0; i < m->T; ++i) {
        double lo[3], hi[3];
        tri_aabb(m, i, lo, hi);
        add_triangle(m, t->root, i, lo, hi);
    }
    return t;
}

void orc_octree_destroy(orc_octree *t) {
    if (!t) return;
    node_free(t->root);
    free(t);
}

static void stats_rec(const orc_node *n, int depth, long s[5]) {
    s[0]++;
    if (depth > s[2]) s[2] = depth;
    if (n->is_leaf) {
        s[1]++;
        s[3] += n->ntris;
        if (n->ntris > s[4]) s[4] = n->ntris;
    } else
        for (int c = 0; c < 8; ++c) stats_rec(n->child[c], depth + 1, s);
}

void orc_octree_stats(const orc_octree *t, long s[5]) {
    memset(s, 0, sizeof(long) * 5);
    stats_rec(t->root, 0, s);
}

/* Octree::distance_to_triangle, R/octree.cpp:143-154 */
static double distance_to_triangle(const orc_mesh *m, const double pt[3], int t) {
    const double *v0 = &m->xyz[3 * m->tri[3 * t]], *v1 = &m->xyz[3 * m->tri[3 * t + 1]], *v2 = &m->xyz[3 * m->tri[3 * t + 2]];
    double mp[3];
    orc_project_point(pt, v0, v1, v2, mp);
    if (orc_point_in_triangle(mp, v0, v1, v2)) return orc_dist_to_point(mp, v0, v1, v2);
    return -1.0; /* NOT_IN_TRIANGLE */
}

/* Octree::get_closest_triangle, R/octree.cpp:156-214 */
int orc_octree_closest_triangle(const orc_octree *tr, const double pt[3], long *ntests) {
    const orc_mesh *m = tr->mesh;
    if (!node_contains_point(tr->root, pt)) return -1;
    int closest = -1; /* EMPTY_TRIANGLE */
    double best = DBL_MAX;
    const orc_node *cur = tr->root;
    /* the range-for over children is bound to the node entered with; the LAST containing child wins */
    while (!cur->is_leaf) {
        const orc_node *at = cur;
        for (int c = 0; c < 8; ++c)
            if (node_contains_point(at->child[c], pt)) cur = at->child[c];
        if (cur == at) return -2; /* cannot happen: children tile the parent */
    }
    for (int i = 0; i < cur->ntris; ++i) {
        double d = distance_to_triangle(m, pt, cur->tris[i]);
        if (ntests) ++*ntests;
        if (d > -1.0 && d < best) {
            closest = cur->tris[i];
            best = d;
        }
    }
    if (closest == -1 && cur->parent) {
        best = DBL_MAX;
        for (int c = 0; c < 8; ++c) {
            const orc_node *o = cur->parent->child[c];
            for (int i = 0; i < o->ntris; ++i) {
                double d = distance_to_triangle(m, pt, o->tris[i]);
                if (ntests) ++*ntests;
                if (d > -1.0 && d < best) {
                    closest = o->tris[i];
                    best = d;
                }
            }
        }
    }
    if (closest == -1 && cur->parent) {
        best = DBL_MAX;
        for (int c = 0; c < 8; ++c) {
            const orc_node *o = cur->parent->child[c];
            for (int i = 0; i < o->ntris; ++i)
                for (int v = 0; v < 3; ++v) {
                    double d[3];
                    v_sub(&m->xyz[3 * m->tri[3 * o->tris[i] + v]], pt, d);
                    double dist = 2 * ORC_RAD * asin(v_norm(d) / (2 * ORC_RAD));
                    if (dist < best) {
                        closest = o->tris[i];
                        best = dist;
                    }
                }
        }
    }
    if (closest == -1) return -2;
    return closest;
}

/* Octree::get_closest_vertex_ID, R/octree.cpp:216-233 */
int orc_octree_closest_vertex(const orc_octree *tr, const double pt[3]) {
    int t = orc_octree_closest_triangle(tr, pt, NULL);
    if (t < 0) return t;
    const orc_mesh *m = tr->mesh;
    double dist = DBL_MAX;
    int best = 0;
    for (int v = 0; v < 3; ++v) {
        double d[3];
        v_sub(pt, &m->xyz[3 * m->tri[3 * t + v]], d);
        double cd = v_norm(d);
        if (cd < dist) {
            best = m->tri[3 * t + v];
            dist = cd;
        }
    }
    return best;
}
