// host_mesh.cpp -- [host] entry points of msmhip.h: icosphere generation, mesh adjacency and the small
// pieces of NonLinearSRegDiscreteModel that prepare inputs for the device path.  No GPU needed.
//
// Vertex / triangle / neighbour numbering is observable in newMSM's outputs, so these routines
// reproduce the reference's order exactly (file:line cited per function, paths under
// /root/reference/libraries/).
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "internal.hpp"

namespace msm {

thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    stage_on_failure();  // stager.cpp: what this thread fetched and has not delivered yet is dropped (its destination may not outlive the failed call)
    return code;
}

// One 1->4 subdivision, R/mesh.cpp:910-1005.  The reference finds already-created edge midpoints by
// an O(V^2) tolerance search over the points added in this pass; midpoints of one edge are computed
// from the same two operands either way round, so they are bit-identical and an edge-keyed hash map
// finds the same point (distinct edges of an icosphere have midpoints far more than 1e-8 apart).
// New points are numbered in the order (v1,v2) (v0,v2) (v0,v1) per triangle, and every vertex is
// re-normalised at the end of the pass, as in the reference.
static void subdivide(std::vector<double> &xyz, std::vector<int32_t> &tri) {
    const size_t oldT = tri.size() / 3;
    std::vector<int32_t> next;
    next.reserve(oldT * 12);
    std::unordered_map<uint64_t, int32_t> midpoint;
    midpoint.reserve(oldT * 2);
    auto key = [](int32_t a, int32_t b) { return (uint64_t)std::min(a, b) << 32 | (uint32_t)std::max(a, b); };
    for (size_t t = 0; t < oldT; ++t) {
        const int32_t v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
        const int32_t ea[3] = {v1, v0, v0}, eb[3] = {v2, v2, v1};
        int32_t p[3];
        bool fresh[3];
        for (int m = 0; m < 3; ++m) {
            auto it = midpoint.find(key(ea[m], eb[m]));
            fresh[m] = it == midpoint.end();
            p[m] = fresh[m] ? -1 : it->second;
        }
        for (int m = 0; m < 3; ++m)
            if (fresh[m]) {
                p[m] = (int32_t)(xyz.size() / 3);
                for (int k = 0; k < 3; ++k) xyz.push_back((xyz[3 * ea[m] + k] + xyz[3 * eb[m] + k]) / 2);
                midpoint.emplace(key(ea[m], eb[m]), p[m]);
            }
        const int32_t kids[12] = {p[2], p[0], p[1], p[1], v0, p[2], p[0], v2, p[1], p[2], v1, p[0]};
        next.insert(next.end(), kids, kids + 12);
    }
    tri.swap(next);
    for (size_t i = 0; i < xyz.size() / 3; ++i) {
        V3 n = normalized(mk(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        xyz[3 * i] = n.x;
        xyz[3 * i + 1] = n.y;
        xyz[3 * i + 2] = n.z;
    }
}

// make_mesh_from_icosa, R/mesh.cpp:1111-1196 (AoS output, unit sphere)
void icosphere_unit(int order, std::vector<double> &xyz, std::vector<int32_t> &tri) {
    const double t = 0.8506508084, o = 0.5257311121;
    // ZA ZB ZC ZD YA YB YC YD XA XB XC XD
    const double base[36] = {t, o, 0, -t, o, 0, -t, -o, 0, t, -o, 0, o, 0, t, o, 0, -t, -o, 0, -t, -o, 0, t, 0, t, o, 0, -t, o, 0, -t, -o, 0, t, -o};
    enum { ZA, ZB, ZC, ZD, YA, YB, YC, YD, XA, XB, XC, XD };
    const int32_t faces[60] = {YD, XA, YA, XB, YD, YA, XD, YC, YB, YC, XC, YB, ZD, YA, ZA, YB, ZD, ZA, ZB, YD, ZC, YC, ZB, ZC, XD, ZA, XA, ZB, XD, XA,
                               ZD, XC, XB, XC, ZC, XB, ZA, YA, XA, YB, ZA, XD, ZD, XB, YA, XC, ZD, YB, ZB, XA, YD, XD, ZB, YC, XB, ZC, YD, ZC, XC, YC};
    xyz.assign(base, base + 36);
    tri.resize(60);
    for (int f = 0; f < 20; ++f) {  // swap_orientation(): second and third vertex exchanged
        tri[3 * f] = faces[3 * f];
        tri[3 * f + 1] = faces[3 * f + 2];
        tri[3 * f + 2] = faces[3 * f + 1];
    }
    for (int i = 0; i < order; ++i) subdivide(xyz, tri);
}

// Mesh::push_triangle applied to every triangle in order, R/mesh.cpp:115-134
// Mesh::initialize's neighbour and triangle lists (R/mesh.cpp: push_triangle): per vertex the triangles in ascending id, and the
// neighbours in first-seen order -- vertex n[0] of a triangle meets n[1] then n[2], n[1] meets n[0] then n[2], n[2] meets n[0] then
// n[1].  Flat arrays only (a counting sort for the triangles, then one walk over each vertex's triangles for the neighbours): the
// first version kept two std::vectors per vertex and took 2-3 ms at ico6, which every fresh mesh of a registration level paid.
void build_adjacency(const int32_t *tri, int V, int T, Adjacency &adj) {
    adj.tid_ptr.assign((size_t)V + 1, 0);
    for (int k = 0; k < 3; ++k)
        for (int t = 0; t < T; ++t) adj.tid_ptr[(size_t)tri[(size_t)k * T + t] + 1]++;
    for (int v = 0; v < V; ++v) adj.tid_ptr[v + 1] += adj.tid_ptr[v];
    adj.tid.resize((size_t)adj.tid_ptr[V]);
    std::vector<int32_t> fill(adj.tid_ptr.begin(), adj.tid_ptr.end() - 1);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            const int32_t v = tri[(size_t)k * T + t];
            // a vertex listed twice in one (degenerate) triangle is listed twice here too, as push_back did
            adj.tid[(size_t)fill[v]++] = t;
        }
    // neighbours: at most two new ones per incident triangle
    adj.nbr_ptr.assign((size_t)V + 1, 0);
    std::vector<int32_t> scratch(2 * adj.tid.size());
    std::vector<int32_t> count(V, 0);
    for (int v = 0; v < V; ++v) {
        int32_t *list = scratch.data() + 2 * (size_t)adj.tid_ptr[v];
        int n = 0;
        int32_t last_t = -1;
        for (int j = adj.tid_ptr[v]; j < adj.tid_ptr[v + 1]; ++j) {
            const int32_t t = adj.tid[j];
            if (t == last_t) continue;  // degenerate triangle: handled once, for each position below
            last_t = t;
            const int32_t nn[3] = {tri[t], tri[(size_t)T + t], tri[2 * (size_t)T + t]};
            for (int k = 0; k < 3; ++k) {
                if (nn[k] != v) continue;
                for (int q = 0; q < 3; ++q) {
                    if (q == k) continue;
                    const int32_t o = nn[q];
                    bool seen = false;
                    for (int i = 0; i < n && !seen; ++i) seen = list[i] == o;
                    if (!seen) list[n++] = o;
                }
            }
        }
        count[v] = n;
        adj.nbr_ptr[v + 1] = adj.nbr_ptr[v] + n;
    }
    adj.nbr.resize((size_t)adj.nbr_ptr[V]);
    for (int v = 0; v < V; ++v) std::copy_n(scratch.data() + 2 * (size_t)adj.tid_ptr[v], count[v], adj.nbr.begin() + adj.nbr_ptr[v]);
}

static inline V3 pt(const double *xyz, int V, int i) { return mk(xyz[i], xyz[V + i], xyz[2 * V + i]); }

}  // namespace msm

using namespace msm;

// every [host] entry point that takes a triangle list checks it before indexing with it
static int check_triangles(const char *who, const int32_t *tri, int32_t V, int32_t T) {
    if (!tri || V <= 0 || T <= 0) return fail(MSM_ERR_INVALID, "%s: bad arguments", who);
    for (int64_t i = 0; i < 3 * (int64_t)T; ++i)
        if (tri[i] < 0 || tri[i] >= V) return fail(MSM_ERR_INVALID, "%s: triangle vertex id %d out of range [0,%d)", who, tri[i], V);
    return MSM_OK;
}

extern "C" {

int msm_abi_version(void) { return MSM_ABI_VERSION; }
const char *msm_last_error(void) { return g_error.c_str(); }

int msm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int msm_icosphere_counts(int order, int32_t *V, int32_t *T) {
    if (order < 0 || order > 9) return fail(MSM_ERR_INVALID, "icosphere order %d out of range [0,9]", order);
    int64_t v = 12, t = 20;
    for (int i = 0; i < order; ++i) {
        v += 3 * t / 2;
        t *= 4;
    }
    if (V) *V = (int32_t)v;
    if (T) *T = (int32_t)t;
    return MSM_OK;
}

int msm_icosphere(int order, double radius, double *xyz, int32_t *tri) {
    int32_t V, T;
    int st = msm_icosphere_counts(order, &V, &T);
    if (st) return st;
    if (!xyz || !tri) return fail(MSM_ERR_INVALID, "msm_icosphere: null output");
    // the unit icosphere of an order is a constant: a registration asks for the same few orders at every level (data grid, control grid,
    // sampling grid, project_CPgrid), 2.4 ms each at order 6 -- kept per order for the life of the process
    static std::mutex mu;
    static std::map<int, std::shared_ptr<const std::pair<std::vector<double>, std::vector<int32_t>>>> kept;
    std::shared_ptr<const std::pair<std::vector<double>, std::vector<int32_t>>> unit;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = kept.find(order);
        if (it != kept.end()) unit = it->second;
    }
    if (!unit) {
        auto fresh = std::make_shared<std::pair<std::vector<double>, std::vector<int32_t>>>();
        icosphere_unit(order, fresh->first, fresh->second);
        unit = fresh;
        std::lock_guard<std::mutex> lock(mu);
        kept.emplace(order, unit);
    }
    const std::vector<double> &p = unit->first;
    const std::vector<int32_t> &f = unit->second;
    for (int i = 0; i < V; ++i) {
        V3 c = mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        if (radius > 0) c = scale(normalized(c), radius);  // true_rescale, R/mesh.cpp:1210-1219
        xyz[i] = c.x;
        xyz[V + i] = c.y;
        xyz[2 * V + i] = c.z;
    }
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) tri[k * T + t] = f[3 * t + k];
    return MSM_OK;
}

// Mesh_registration::resample_anatomy, M/mesh_registration.cpp:250-332, without its surface_resample call.
//   ANAT_ico   the control grid retessellated `levels` times: retessellate(mesh, old_tr_nbours) R/mesh.cpp:1007-1109 numbers the children of triangle t
//              4t .. 4t+3 and re-normalises EVERY vertex to the unit sphere; true_rescale to `rad` afterwards (:299)
//   NEARESTFACES[i]  the faces of ANAT_ico under control triangle i: after the first pass [4i, 4i+1, 4i+2, 4i+3]; every further pass replaces each
//              entry by its four children, inserting each entry's children at the FRONT of the new list (:268-283) -- so the order of a list
//              reverses block-wise with every pass.  It is the summation order of the mean strain (computeTripletCost, M/DiscreteCostFunction.cpp:169-182)
//   _ANATbaryweights[v]  calc_barycentric_weights (R/triangle.cpp:124-143) of ANAT_ico vertex v in the control triangle of the LAST (i, face, corner) that
//              names it in the loop :303-321 (std::map assignment replaces the row); keys ascending (std::map order)
int msm_resample_anatomy_grid(const double *cp_xyz, int32_t N, const int32_t *cp_tri, int32_t Tc, int32_t levels, double rad, int32_t *Va_out, int32_t *Ta_out,
                              double *axyz, int32_t *atri, int32_t *w_ptr, int32_t *w_cp, double *w_val, int32_t *face_ptr, int32_t *face_idx) {
    if (!cp_xyz || !cp_tri || N <= 0 || Tc <= 0 || levels < 0 || levels > 8) return fail(MSM_ERR_INVALID, "msm_resample_anatomy_grid: bad arguments");
    if (int st = check_triangles("msm_resample_anatomy_grid", cp_tri, N, Tc)) return st;
    std::vector<double> xyz(3 * (size_t)N);
    std::vector<int32_t> tri(3 * (size_t)Tc);
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = cp_xyz[(size_t)k * N + i];
    for (int t = 0; t < Tc; ++t)
        for (int k = 0; k < 3; ++k) tri[3 * (size_t)t + k] = cp_tri[(size_t)k * Tc + t];
    std::vector<std::vector<int32_t>> faces((size_t)Tc);
    for (int i = 0; i < Tc; ++i) faces[(size_t)i].push_back(i);  // levels == 0: each control triangle is its own face (:287-293)
    for (int pass = 0; pass < levels; ++pass) {
        subdivide(xyz, tri);  // children of triangle t: 4t .. 4t+3 (tot_triangles counts up through the loop over the old triangles)
        for (auto &list : faces) {
            std::vector<int32_t> next;
            next.reserve(list.size() * 4);
            if (pass == 0) {  // FACE_neighbours = FACE_neighbours_tmp: in order
                for (int32_t f : list)
                    for (int c = 0; c < 4; ++c) next.push_back(4 * f + c);
            } else {          // insert(begin, children of list[k]) for k = 0, 1, ...: the last entry's children come first
                for (size_t k = list.size(); k-- > 0;)
                    for (int c = 0; c < 4; ++c) next.push_back(4 * list[k] + c);
            }
            list.swap(next);
        }
    }
    const int32_t Va = (int32_t)(xyz.size() / 3), Ta = (int32_t)(tri.size() / 3);
    if (Va_out) *Va_out = Va;
    if (Ta_out) *Ta_out = Ta;
    if (!axyz && !atri && !w_ptr && !w_cp && !w_val && !face_ptr && !face_idx) return MSM_OK;  // the sizing call
    if (!axyz || !atri || !w_ptr || !w_cp || !w_val || !face_ptr || !face_idx) return fail(MSM_ERR_INVALID, "msm_resample_anatomy_grid: all outputs or none");
    for (int32_t i = 0; i < Va; ++i) {  // true_rescale(ANAT_ico, RAD)
        const V3 c = scale(normalized(mk(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2])), rad);
        xyz[3 * (size_t)i] = c.x, xyz[3 * (size_t)i + 1] = c.y, xyz[3 * (size_t)i + 2] = c.z;
    }
    struct Row {
        int n = 0;
        int32_t id[3];
        double w[3];
    };
    std::vector<Row> rows((size_t)Va);
    for (int i = 0; i < Tc; ++i) {
        const int32_t id[3] = {cp_tri[i], cp_tri[(size_t)Tc + i], cp_tri[2 * (size_t)Tc + i]};
        V3 v[3];
        for (int k = 0; k < 3; ++k) v[k] = mk(cp_xyz[id[k]], cp_xyz[(size_t)N + id[k]], cp_xyz[2 * (size_t)N + id[k]]);
        for (int32_t j : faces[(size_t)i])
            for (int k = 0; k < 3; ++k) {
                const int32_t a = tri[3 * (size_t)j + k];
                const V3 ci = mk(xyz[3 * (size_t)a], xyz[3 * (size_t)a + 1], xyz[3 * (size_t)a + 2]);
                const V3 pp = project_point(ci, v[0], v[1], v[2]);
                const double Aa = compute_area(pp, v[1], v[2]), Ab = compute_area(pp, v[0], v[2]), Ac = compute_area(pp, v[0], v[1]);
                const double A = Aa + Ab + Ac;
                const double w[3] = {Aa / A, Ab / A, Ac / A};
                std::map<int32_t, double> m;  // weights[n1] = ..; weights[n2] = ..; weights[n3] = ..: a repeated id keeps the later value
                for (int q = 0; q < 3; ++q) m[id[q]] = w[q];
                Row &r = rows[(size_t)a];
                r.n = 0;
                for (const auto &kv : m) {
                    r.id[r.n] = kv.first;
                    r.w[r.n] = kv.second;
                    ++r.n;
                }
            }
    }
    w_ptr[0] = 0;
    for (int32_t a = 0; a < Va; ++a) {
        const Row &r = rows[(size_t)a];
        for (int q = 0; q < r.n; ++q) {
            w_cp[w_ptr[a] + q] = r.id[q];
            w_val[w_ptr[a] + q] = r.w[q];
        }
        w_ptr[a + 1] = w_ptr[a] + r.n;
    }
    face_ptr[0] = 0;
    for (int i = 0; i < Tc; ++i) {
        std::copy(faces[(size_t)i].begin(), faces[(size_t)i].end(), face_idx + face_ptr[i]);
        face_ptr[i + 1] = face_ptr[i] + (int32_t)faces[(size_t)i].size();
    }
    for (int32_t i = 0; i < Va; ++i)
        for (int k = 0; k < 3; ++k) axyz[(size_t)k * Va + i] = xyz[3 * (size_t)i + k];
    for (int32_t t = 0; t < Ta; ++t)
        for (int k = 0; k < 3; ++k) atri[(size_t)k * Ta + t] = tri[3 * (size_t)t + k];
    return MSM_OK;
}

int msm_mesh_adjacency(const int32_t *tri, int32_t V, int32_t T, int32_t *nbr_ptr, int32_t *nbr, int32_t *tid_ptr, int32_t *tid) {
    if (int st = check_triangles("msm_mesh_adjacency", tri, V, T)) return st;
    Adjacency a;
    build_adjacency(tri, V, T, a);
    if (nbr_ptr) std::copy(a.nbr_ptr.begin(), a.nbr_ptr.end(), nbr_ptr);
    if (tid_ptr) std::copy(a.tid_ptr.begin(), a.tid_ptr.end(), tid_ptr);
    if (nbr) std::copy(a.nbr.begin(), a.nbr.end(), nbr);
    if (tid) std::copy(a.tid.begin(), a.tid.end(), tid);
    return MSM_OK;
}

int msm_vertex_areas(const double *xyz, const int32_t *tri, int32_t V, int32_t T, double *area) {
    if (!xyz || !area) return fail(MSM_ERR_INVALID, "msm_vertex_areas: bad arguments");
    if (int st = check_triangles("msm_vertex_areas", tri, V, T)) return st;
    Adjacency a;
    build_adjacency(tri, V, T, a);
    std::vector<double> ta(T);
    for (int t = 0; t < T; ++t) ta[t] = tri_area(pt(xyz, V, tri[t]), pt(xyz, V, tri[T + t]), pt(xyz, V, tri[2 * T + t]));
    for (int v = 0; v < V; ++v) {
        double sum = 0;
        for (int j = a.tid_ptr[v]; j < a.tid_ptr[v + 1]; ++j) sum += ta[a.tid[j]];
        area[v] = sum / (a.tid_ptr[v + 1] - a.tid_ptr[v]);
    }
    return MSM_OK;
}

int msm_cp_spacings(const double *xyz, const int32_t *tri, int32_t V, int32_t T, double *maxsep, double *mvdmax) {
    if (!xyz || !maxsep || !mvdmax) return fail(MSM_ERR_INVALID, "msm_cp_spacings: null argument");
    if (int st = check_triangles("msm_cp_spacings", tri, V, T)) return st;
    Adjacency a;
    build_adjacency(tri, V, T, a);
    double best = -DBL_MAX;  // calculate_MaxVD starts from lowest()
    for (int k = 0; k < V; ++k) {
        maxsep[k] = 0;
        V3 cp = pt(xyz, V, k);
        for (int j = a.nbr_ptr[k]; j < a.nbr_ptr[k + 1]; ++j) {
            double dist = chord_to_arc(norm(sub(cp, pt(xyz, V, a.nbr[j]))));
            if (dist > maxsep[k]) maxsep[k] = dist;
            if (dist > best) best = dist;
        }
    }
    *mvdmax = best;
    return MSM_OK;
}

int msm_label_sampling_grid(int sg_order, double max_dist, int abs_is_int, int32_t cap, double *samples, int32_t *nsamples,
                            double *barycentres, int32_t *nbarycentres) {
    int32_t V, T;
    int st = msm_icosphere_counts(sg_order, &V, &T);
    if (st) return st;
    std::vector<double> g(3 * (size_t)V);
    std::vector<int32_t> f(3 * (size_t)T);
    msm_icosphere(sg_order, kRad, g.data(), f.data());
    Adjacency a;
    build_adjacency(f.data(), V, T, a);
    int centroid = 0;  // first vertex with six neighbours, M/DiscreteModel.cpp:115-120
    for (int i = 0; i < V; ++i)
        if (a.nbr_ptr[i + 1] - a.nbr_ptr[i] == 6) {
            centroid = i;
            break;
        }
    const V3 centre = pt(g.data(), V, centroid);
    // std::map<double,Point>: ascending distance, equal keys overwrite
    std::map<double, V3> smp, bar;
    std::vector<char> found(V, 0), found_tr(T, 0);
    std::vector<int> ring{centroid}, next;
    while (!ring.empty()) {
        for (int gn : ring) {
            for (int j = a.nbr_ptr[gn]; j < a.nbr_ptr[gn + 1]; ++j) {
                int v = a.nbr[j];
                V3 s = pt(g.data(), V, v);
                double distance = norm(sub(s, centre));
                if (distance <= max_dist && !found[v] && v != centroid) {
                    smp[distance] = s;
                    next.push_back(v);
                    found[v] = 1;
                }
            }
            for (int j = a.tid_ptr[gn]; j < a.tid_ptr[gn + 1]; ++j) {
                int t = a.tid[j];
                V3 v1 = pt(g.data(), V, f[t]), v2 = pt(g.data(), V, f[T + t]), v3 = pt(g.data(), V, f[2 * T + t]);
                V3 b = normalized(mk((v1.x + v2.x + v3.x) / 3, (v1.y + v2.y + v3.y) / 3, (v1.z + v2.z + v3.z) / 3));
                b = scale(b, kRad);
                V3 bc = sub(b, centre);
                double distance = norm(bc);
                if (distance <= max_dist && norm(bc) > 0 && !found_tr[t]) {
                    for (auto &e : bar) {
                        V3 oc = sub(e.second, centre);
                        double q = 1 - (dot(bc, oc) / (norm(bc) * norm(oc)));
                        double aq = abs_is_int ? (double)std::abs((int)q) : std::fabs(q);
                        if (aq < 1e-2) found_tr[t] = 1;
                    }
                    if (!found_tr[t]) bar[distance] = b;
                    found_tr[t] = 1;
                }
            }
        }
        ring.swap(next);
        next.clear();
    }
    const int ns = 1 + (int)smp.size(), nb = 1 + (int)bar.size();
    if (nsamples) *nsamples = ns;
    if (nbarycentres) *nbarycentres = nb;
    if (ns > cap || nb > cap) return fail(MSM_ERR_CAPACITY, "label buffers hold %d entries, need %d / %d", cap, ns, nb);
    auto emit = [&](double *out, const std::map<double, V3> &m) {
        if (!out) return;
        int i = 0;
        out[0] = centre.x, out[cap] = centre.y, out[2 * cap] = centre.z;
        for (auto &e : m) {
            ++i;
            out[i] = e.second.x, out[cap + i] = e.second.y, out[2 * cap + i] = e.second.z;
        }
    };
    emit(samples, smp);
    emit(barycentres, bar);
    return MSM_OK;
}

int msm_rescale_sampling_grid(const double *samples, int32_t n, double *sc, double *labels) {
    if (!samples || !sc || !labels || n <= 0) return fail(MSM_ERR_INVALID, "msm_rescale_sampling_grid: bad arguments");
    const V3 centre = pt(samples, n, 0);
    if (*sc >= 0.25) {
        for (int i = 0; i < n; ++i) {
            V3 s = pt(samples, n, i);
            V3 p = mk(centre.x + (centre.x - s.x) * (*sc), centre.y + (centre.y - s.y) * (*sc), centre.z + (centre.z - s.z) * (*sc));
            p = scale(normalized(p), 100);
            labels[i] = p.x, labels[n + i] = p.y, labels[2 * n + i] = p.z;
        }
    } else {
        *sc = 1;
        std::copy(samples, samples + 3 * (size_t)n, labels);
    }
    *sc *= 0.8;
    return MSM_OK;
}

int msm_rotation_matrix(const double ci[3], const double index[3], double R[9]) {
    if (!rotation_matrix(mk(ci[0], ci[1], ci[2]), mk(index[0], index[1], index[2]), R))
        return fail(MSM_ERR_ROTATION, "rotation angle is greater than 90 degrees");
    return MSM_OK;
}

int msm_cp_rotations(const double centre[3], const double *cp, int32_t N, double *rot) {
    if (!centre || !cp || !rot) return fail(MSM_ERR_INVALID, "msm_cp_rotations: null argument");
    for (int k = 0; k < N; ++k)
        if (!rotation_matrix(mk(centre[0], centre[1], centre[2]), pt(cp, N, k), rot + 9 * (size_t)k))
            return fail(MSM_ERR_ROTATION, "rotation angle is greater than 90 degrees");
    return MSM_OK;
}

int msm_estimate_triplets(const int32_t *tri, int32_t T, int32_t *triplets) {
    if (!tri || !triplets) return fail(MSM_ERR_INVALID, "msm_estimate_triplets: null argument");
    for (int t = 0; t < T; ++t) {
        int32_t v[3] = {tri[t], tri[T + t], tri[2 * T + t]};
        std::sort(v, v + 3);
        std::copy(v, v + 3, triplets + 3 * (size_t)t);
    }
    return MSM_OK;
}

int msm_estimate_pairs(const int32_t *tri, int32_t V, int32_t T, int32_t *pairs) {
    if (int st = check_triangles("msm_estimate_pairs", tri, V, T)) return st;
    Adjacency a;
    build_adjacency(tri, V, T, a);
    int n = 0;
    for (int i = 0; i < V; ++i)
        for (int j = a.nbr_ptr[i]; j < a.nbr_ptr[i + 1]; ++j)
            if (a.nbr[j] > i) {
                if (pairs) {
                    pairs[2 * n] = i;
                    pairs[2 * n + 1] = a.nbr[j];
                }
                ++n;
            }
    return n;
}

}  // extern "C"
