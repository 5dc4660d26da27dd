// msmhip.hpp -- C++ host side above the C ABI of msmhip.h (header only, C++17, no HIP headers).
//
// newMSM is C++; this is the host-language mirror of the interfaces its optimisers and drivers use for the hot
// path, with the reference's names and argument meaning, so that an adapter inside newMSM is a
// matter of converting NEWMAT / newresampler containers to the plain vectors used here:
//
//   msmhip::Mesh                         newresampler::Mesh as the path sees it (coordinates, triangles, pvalues)
//                                        + newresampler::Octree (built behind the handle)          R/mesh.h, R/octree.h:48-52
//   msmhip::get_barycentric_weights ...  the free functions of R/resampler.h:38-53
//   msmhip::DiscreteCostFunction         DiscreteCostFunction's evaluator interface (M/DiscreteCostFunction.h:32-80) with
//                                        the NonLinearSRegDiscreteCostFunction setters (:82-224); `kind` selects which of
//                                        the five subclasses (:226-283) it stands for
//   msmhip::DiscreteGroupModel           DiscreteGroupModel + DiscreteGroupCostFunction (M/DiscreteGroupModel.h:37-108)
//   msmhip::FusionModel / GroupFusionModel   newmeshreg::DiscreteModel AS THE UNMODIFIED OPTIMISERS CALL IT (M/DiscreteModel.h:31-88):
//                                        per-clique evaluators that are thread safe and O(1) -- Fusion::optimize's OpenMP
//                                        loops (I/Fusion/Fusion.h:148-196) turn into ONE ABI call per label step behind them
//
// Conventions: points are AoS (x0 y0 z0 x1 ...) as in newresampler::Point containers and are transposed to the
// ABI's SoA here; triangles are AoS (3 ids per triangle); data matrices are row-major D x V like
// newresampler::Mesh::pvalues.  Errors are thrown as msmhip::Error carrying the reference's exception text.
#pragma once

#include <atomic>
#include <cstdint>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "msmhip.h"

namespace msmhip {

struct Error : std::runtime_error {  // MeshregException / MeshException (what() prints the message)
    int code;
    Error(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};
inline void check(int st) {
    if (st != MSM_OK) throw Error(st, msm_last_error());
}

using Points = std::vector<double>;     // AoS, 3 per point
using Triangles = std::vector<int32_t>; // AoS, 3 per triangle
using Matrix = std::vector<double>;     // row-major rows x cols

inline std::vector<double> to_soa(const Points &p) {
    const size_t n = p.size() / 3;
    std::vector<double> s(3 * n);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) s[a * n + i] = p[3 * i + a];
    return s;
}
inline Points to_aos(const std::vector<double> &s) {
    const size_t n = s.size() / 3;
    Points p(3 * n);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) p[3 * i + a] = s[a * n + i];
    return p;
}
inline std::vector<int32_t> tri_to_soa(const Triangles &t) {
    const size_t n = t.size() / 3;
    std::vector<int32_t> s(3 * n);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) s[a * n + i] = t[3 * i + a];
    return s;
}

// ---------------------------------------------------------------- host-side model helpers (no GPU)
// make_mesh_from_icosa + true_rescale, R/mesh.cpp:1111-1219
inline std::pair<Points, Triangles> make_mesh_from_icosa(int order, double radius = MSM_RAD) {
    int32_t V, T;
    check(msm_icosphere_counts(order, &V, &T));
    std::vector<double> xyz(3 * (size_t)V);
    std::vector<int32_t> tri(3 * (size_t)T);
    check(msm_icosphere(order, radius, xyz.data(), tri.data()));
    Triangles t(3 * (size_t)T);
    for (int i = 0; i < T; ++i)
        for (int a = 0; a < 3; ++a) t[3 * (size_t)i + a] = tri[(size_t)a * T + i];
    return {to_aos(xyz), t};
}
// NonLinearSRegDiscreteModel::Initialize, M/DiscreteModel.cpp:72-89: MAXSEP per control point and MVDmax
inline std::pair<std::vector<double>, double> cp_spacings(const Points &xyz, const Triangles &tri) {
    const int32_t V = (int32_t)(xyz.size() / 3), T = (int32_t)(tri.size() / 3);
    std::vector<double> ms(V);
    double mvd = 0;
    check(msm_cp_spacings(to_soa(xyz).data(), tri_to_soa(tri).data(), V, T, ms.data(), &mvd));
    return {ms, mvd};
}
// get_rotations, M/DiscreteModel.cpp:310-319 (row-major 3x3 per control point)
inline std::vector<double> cp_rotations(const double centre[3], const Points &cp) {
    const int32_t N = (int32_t)(cp.size() / 3);
    std::vector<double> rot(9 * (size_t)N);
    check(msm_cp_rotations(centre, to_soa(cp).data(), N, rot.data()));
    return rot;
}
// Initialize_sampling_grid + label_sampling_grid, M/DiscreteModel.cpp:110-190: {samples, barycentres}; samples[0] is the centre
inline std::pair<Points, Points> label_sampling_grid(int sg_order, double max_dist, bool abs_is_int = false) {
    const int32_t cap = 4096;
    std::vector<double> s(3 * (size_t)cap), b(3 * (size_t)cap);
    int32_t ns = 0, nb = 0;
    check(msm_label_sampling_grid(sg_order, max_dist, abs_is_int ? 1 : 0, cap, s.data(), &ns, b.data(), &nb));
    Points samples(3 * (size_t)ns), bary(3 * (size_t)nb);
    for (int i = 0; i < ns; ++i)
        for (int a = 0; a < 3; ++a) samples[3 * (size_t)i + a] = s[(size_t)a * cap + i];
    for (int i = 0; i < nb; ++i)
        for (int a = 0; a < 3; ++a) bary[3 * (size_t)i + a] = b[(size_t)a * cap + i];
    return {samples, bary};
}
// rescale_sampling_grid, M/DiscreteModel.cpp:192-214: the labels of this iteration; `scale` is m_scale (read and updated)
inline Points rescale_sampling_grid(const Points &samples, double &scale) {
    const int32_t n = (int32_t)(samples.size() / 3);
    std::vector<double> out(3 * (size_t)n);
    check(msm_rescale_sampling_grid(to_soa(samples).data(), n, &scale, out.data()));
    return to_aos(out);
}
// estimate_triplets / estimate_pairs, M/DiscreteModel.cpp:271-308
inline std::vector<int32_t> estimate_triplets(const Triangles &tri) {
    const int32_t T = (int32_t)(tri.size() / 3);
    std::vector<int32_t> out(3 * (size_t)T);
    check(msm_estimate_triplets(tri_to_soa(tri).data(), T, out.data()));
    return out;
}
inline std::vector<int32_t> estimate_pairs(const Triangles &tri, int V) {
    const int32_t T = (int32_t)(tri.size() / 3);
    const auto s = tri_to_soa(tri);
    const int n = msm_estimate_pairs(s.data(), V, T, nullptr);
    if (n < 0) check(n);
    std::vector<int32_t> out(2 * (size_t)n);
    const int m = msm_estimate_pairs(s.data(), V, T, out.data());
    if (m < 0) check(m);
    return out;
}

// ---------------------------------------------------------------- context and mesh
class Context {
public:
    explicit Context(int device = 0) : h_(msm_ctx_create(device)) {
        if (!h_) throw Error(MSM_ERR_NOGPU, msm_last_error());
    }
    ~Context() { msm_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    msm_ctx *handle() const { return h_; }
    void synchronize() { check(msm_ctx_synchronize(h_)); }

private:
    msm_ctx *h_;
};

class Mesh {
public:
    Mesh(Context &ctx, const Points &xyz, const Triangles &tri) : V_((int)(xyz.size() / 3)), T_((int)(tri.size() / 3)) {
        h_ = msm_mesh_create(ctx.handle(), to_soa(xyz).data(), V_, tri_to_soa(tri).data(), T_);
        if (!h_) throw Error(MSM_ERR_INVALID, msm_last_error());
    }
    ~Mesh() { msm_mesh_destroy(h_); }
    Mesh(const Mesh &) = delete;
    Mesh &operator=(const Mesh &) = delete;
    int nvertices() const { return V_; }
    int ntriangles() const { return T_; }
    void set_coords(const Points &xyz) { check(msm_mesh_update_coords(h_, to_soa(xyz).data())); }  // Mesh::set_coord for every vertex
    Points get_coords() const {
        std::vector<double> s(3 * (size_t)V_);
        check(msm_mesh_get_coords(h_, s.data()));
        return to_aos(s);
    }
    // target-side search structures now instead of in the background (msm_mesh_prepare_search)
    bool prepare_search(bool wait = true) {
        int32_t ready = 0;
        check(msm_mesh_prepare_search(h_, wait ? 1 : 0, &ready));
        return ready != 0;
    }
    void set_pvalues(const Matrix &data) { check(msm_mesh_set_features(h_, data.data(), (int32_t)(data.size() / V_))); }  // D x V
    msm_mesh *handle() const { return h_; }

private:
    msm_mesh *h_;
    int V_, T_;
};

// ---------------------------------------------------------------- resampler (R/resampler.h:38-53)
struct BarycentricWeights {  // per query: the hit triangle, its vertex ids (triangle order) and weights
    std::vector<int32_t> triangle, ids;
    std::vector<double> weights;
};
// Octree::get_closest_triangle + Resampler::get_barycentric_weights, R/octree.cpp:156-214, R/resampler.cpp:142-167
inline BarycentricWeights get_barycentric_weights(Mesh &target, const Points &q, bool raw_weights = false) {
    const int32_t N = (int32_t)(q.size() / 3);
    BarycentricWeights r;
    r.triangle.resize(N);
    std::vector<int32_t> ids(3 * (size_t)N);
    std::vector<double> w(3 * (size_t)N);
    check(msm_query_triangles(target.handle(), to_soa(q).data(), N, r.triangle.data(), ids.data(), w.data(),
                              raw_weights ? MSM_WEIGHTS_RAW : MSM_WEIGHTS_PROJECTED));
    r.ids.resize(3 * (size_t)N);
    r.weights.resize(3 * (size_t)N);
    for (int i = 0; i < N; ++i)
        for (int a = 0; a < 3; ++a) {
            r.ids[3 * (size_t)i + a] = ids[(size_t)a * N + i];
            r.weights[3 * (size_t)i + a] = w[(size_t)a * N + i];
        }
    return r;
}
struct SparseWeights {  // std::vector<std::map<int,double>> as CSR, columns ascending
    std::vector<int32_t> row_ptr, col;
    std::vector<double> val;
};
// Resampler::get_adaptive_barycentric_weights, R/resampler.cpp:72-140
inline SparseWeights get_adaptive_barycentric_weights(Mesh &in_mesh, Mesh &new_mesh, const std::vector<double> *excl = nullptr) {
    SparseWeights w;
    int64_t nnz = 0;
    const double *pe = excl ? excl->data() : nullptr;
    check(msm_adaptive_barycentric_weights(in_mesh.handle(), new_mesh.handle(), pe, nullptr, nullptr, nullptr, 0, &nnz));
    w.row_ptr.resize((size_t)new_mesh.nvertices() + 1);
    w.col.resize((size_t)nnz);
    w.val.resize((size_t)nnz);
    check(msm_adaptive_barycentric_weights(in_mesh.handle(), new_mesh.handle(), pe, w.row_ptr.data(), w.col.data(), w.val.data(), nnz, &nnz));
    return w;
}
// metric_resample, R/resampler.cpp:304-309: data D x V(in) -> D x V(ref)
inline Matrix metric_resample(Mesh &in_mesh, const Matrix &data, Mesh &ref, std::vector<double> *EXCL = nullptr) {
    const int32_t D = (int32_t)(data.size() / in_mesh.nvertices());
    Matrix out((size_t)D * ref.nvertices());
    std::vector<double> eo(EXCL ? ref.nvertices() : 0);
    check(msm_metric_resample(in_mesh.handle(), data.data(), D, ref.handle(), EXCL ? EXCL->data() : nullptr, out.data(), EXCL ? eo.data() : nullptr));
    if (EXCL) *EXCL = eo;  // the reference writes the resampled mask back (:66)
    return out;
}
// surface_resample (R/resampler.cpp:284-302) / project_anatomical_mesh (:260-282): the coordinates `coords` given on the vertices of `from`, carried to
// the points q by barycentric weights
inline Points barycentric_coords_resample(Mesh &from, const Points &coords, const Points &q) {
    std::vector<double> out(q.size());
    check(msm_barycentric_coords_resample(from.handle(), to_soa(coords).data(), to_soa(q).data(), (int32_t)(q.size() / 3), out.data()));
    return to_aos(out);
}
// Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) without its surface_resample call [host]: see msm_resample_anatomy_grid
struct AnatomyGrid {
    Points sphere_xyz;       // ANAT_ico, radius rad
    Triangles sphere_tri;
    SparseWeights weights;   // _ANATbaryweights
    std::vector<int32_t> face_ptr, face_idx;  // NEARESTFACES
};
inline AnatomyGrid resample_anatomy_grid(const Points &cp_xyz, const Triangles &cp_tri, int levels, double rad = 100.0) {
    const int32_t N = (int32_t)(cp_xyz.size() / 3), Tc = (int32_t)(cp_tri.size() / 3);
    const std::vector<double> x = to_soa(cp_xyz);
    const std::vector<int32_t> t = tri_to_soa(cp_tri);
    int32_t Va = 0, Ta = 0;
    check(msm_resample_anatomy_grid(x.data(), N, t.data(), Tc, levels, rad, &Va, &Ta, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    std::vector<double> ax(3 * (size_t)Va), wv(3 * (size_t)Va);
    std::vector<int32_t> at(3 * (size_t)Ta), wp((size_t)Va + 1), wc(3 * (size_t)Va);
    AnatomyGrid g;
    g.face_ptr.resize((size_t)Tc + 1);
    g.face_idx.resize((size_t)Ta);
    check(msm_resample_anatomy_grid(x.data(), N, t.data(), Tc, levels, rad, &Va, &Ta, ax.data(), at.data(), wp.data(), wc.data(), wv.data(), g.face_ptr.data(),
                                    g.face_idx.data()));
    g.sphere_xyz = to_aos(ax);
    g.sphere_tri.resize(3 * (size_t)Ta);
    for (int32_t k = 0; k < Ta; ++k)
        for (int c = 0; c < 3; ++c) g.sphere_tri[3 * (size_t)k + c] = at[(size_t)c * Ta + k];
    wc.resize((size_t)wp.back());
    wv.resize((size_t)wp.back());
    g.weights.row_ptr = wp, g.weights.col = wc, g.weights.val = wv;
    return g;
}
// sphere_project_warp, R/resampler.cpp:311-328: `sphere` is carried through the deformation from -> to
inline Points sphere_project_warp(const Points &sphere, Mesh &from, const Points &to) {
    std::vector<double> s = to_soa(sphere);
    check(msm_sphere_project_warp(from.handle(), to_soa(to).data(), s.data(), (int32_t)(sphere.size() / 3)));
    return to_aos(s);
}
// the same for the coordinates a mesh already holds, in place on the device (msm_mesh_sphere_project_warp): SPH_reg of run_discrete_opt
inline void sphere_project_warp(Mesh &sphere, Mesh &from, const Points &to) {
    check(msm_mesh_sphere_project_warp(sphere.handle(), from.handle(), to_soa(to).data()));
}
// smooth_data, R/resampler.cpp:168-230
inline Matrix smooth_data(Mesh &orig, const Matrix &data, Mesh &sphLow, double sigma, std::vector<double> *EXCL = nullptr) {
    const int32_t D = (int32_t)(data.size() / orig.nvertices());
    Matrix out((size_t)D * sphLow.nvertices());
    std::vector<double> eo(EXCL ? sphLow.nvertices() : 0);
    check(msm_smooth_data(orig.handle(), data.data(), D, sphLow.handle(), sigma, EXCL ? EXCL->data() : nullptr, out.data(), EXCL ? eo.data() : nullptr));
    if (EXCL) *EXCL = eo;  // the reference writes the smoothed mask back (:225)
    return out;
}
// nearest_neighbour_interpolation, R/resampler.cpp:232-258
inline Matrix nearest_neighbour_interpolation(Mesh &orig, const Matrix &data, const Points &q, std::vector<double> *EXCL = nullptr) {
    const int32_t D = (int32_t)(data.size() / orig.nvertices()), N = (int32_t)(q.size() / 3);
    Matrix out((size_t)D * N);
    std::vector<double> eo(EXCL ? N : 0);
    check(msm_nearest_neighbour(orig.handle(), data.data(), D, to_soa(q).data(), N, EXCL ? EXCL->data() : nullptr, out.data(), EXCL ? eo.data() : nullptr));
    if (EXCL) *EXCL = eo;
    return out;
}
// create_exclusion, R/mesh.cpp:1257-1273
inline std::vector<double> create_exclusion(const Matrix &data, int V, double thrl, double thru) {
    std::vector<double> excl((size_t)V);
    check(msm_create_exclusion(data.data(), (int32_t)(data.size() / V), V, thrl, thru, excl.data()));
    return excl;
}

// ---------------------------------------------------------------- around one iteration (M/reg_tools.h, M/mcmc_opt.h)
// unfold(SOURCE, verbosity), M/reg_tools.cpp:131-178, on the mesh's current coordinates; returns the passes that moved vertices
inline int unfold(Mesh &SOURCE, double RAD = 100.0, int *first_folded = nullptr) {
    int32_t passes = 0, first = 0;
    check(msm_mesh_unfold(SOURCE.handle(), RAD, &passes, &first));
    if (first_folded) *first_folded = first;
    return passes;
}
// variance_normalise(DATA, EXCL), M/reg_tools.cpp:804-843: DATA D x V in place
inline void variance_normalise(Matrix &DATA, int V, const std::vector<double> *EXCL = nullptr) {
    check(msm_variance_normalise(DATA.data(), (int32_t)(DATA.size() / V), V, EXCL ? EXCL->data() : nullptr));
}
// MCMC::optimise, M/mcmc_opt.h:31-134, over the tables of getUnaryCosts() (L x N) and getTCosts() (T x L x L x L)
inline void mcmc_optimise(const Matrix &unary_costs, const Matrix &tcosts, const std::vector<int32_t> &triplets, int num_nodes, int num_labels,
                          double dist_param, int mciters, uint64_t seed, std::vector<int32_t> &labeling) {
    check(msm_mcmc_optimise(unary_costs.data(), tcosts.data(), triplets.data(), num_nodes, num_labels, (int32_t)(triplets.size() / 3), dist_param,
                            mciters, seed, labeling.data()));
}

// The stand-in for the binary solve of one label step of Fusion::optimize (msm_fusion_icm_step: iterated conditional modes, NOT ELC +
// FastPD): x[node] = 1 where the proposed label is taken.  unary2 N x 2 (may be empty: no unary costs), quads P x 4, octets T x 8.
inline std::vector<int32_t> fusion_icm_step(int num_nodes, const std::vector<double> &unary2, const double *quads, const std::vector<int32_t> &pairs,
                                            const double *octets, const std::vector<int32_t> &triplets, int max_passes = 5) {
    std::vector<int32_t> x((size_t)num_nodes, 0);
    check(msm_fusion_icm_step(unary2.empty() ? nullptr : unary2.data(), quads, pairs.data(), (int32_t)(pairs.size() / 2), octets, triplets.data(),
                              (int32_t)(triplets.size() / 3), num_nodes, max_passes, x.data()));
    return x;
}

// The stand-in for FPD::FastPD on the multi-label pairwise MRF of --regoption=1 (msm_pairwise_icm: iterated conditional modes): unary L x N,
// paircosts P x L x L as computePairwiseCosts fills it, labeling in (the start) and out.
inline void pairwise_icm(const Matrix &unary_costs, const Matrix &paircosts, const std::vector<int32_t> &pairs, int num_nodes, int num_labels,
                         std::vector<int32_t> &labeling, int max_passes = 100) {
    check(msm_pairwise_icm(unary_costs.data(), paircosts.data(), pairs.data(), num_nodes, num_labels, (int32_t)(pairs.size() / 2), max_passes, labeling.data()));
}

// ---------------------------------------------------------------- discrete cost function
struct Parameters {  // what set_parameters reads from the myparam map, M/DiscreteCostFunction.cpp:119-133
    int kind = MSM_COST_UNIVARIATE;
    int simmeasure = 2, regularisermode = 3;
    double lambda = 0.1, shearmodulus = 0.1, bulkmodulus = 10.0, kexponent = 2.0, exponent = 2.0, range = 1.0, percentile = 0.75;
};

class DiscreteCostFunction {
public:
    DiscreteCostFunction(Context &ctx, const Parameters &P) {
        msm_cost_params p{P.kind, P.simmeasure, P.regularisermode, 0, P.lambda, P.shearmodulus, P.bulkmodulus, P.kexponent, P.exponent, P.range, P.percentile};
        h_ = msm_cost_create(ctx.handle(), &p);
        if (!h_) throw Error(MSM_ERR_INVALID, msm_last_error());
    }
    ~DiscreteCostFunction() { msm_cost_destroy(h_); }
    DiscreteCostFunction(const DiscreteCostFunction &) = delete;
    DiscreteCostFunction &operator=(const DiscreteCostFunction &) = delete;

    // ---- what the model hands over (NonLinearSRegDiscreteCostFunction, M/DiscreteCostFunction.h:139-224)
    void set_meshes(Mesh &target, Mesh &source, Mesh &GRID) {
        N_ = GRID.nvertices();
        check(msm_cost_set_meshes(h_, target.handle(), source.handle(), GRID.handle()));
    }
    void reset_source(Mesh &source) { check(msm_cost_reset_source(h_, source.handle())); }
    void reset_CPgrid(Mesh &grid) { check(msm_cost_reset_cpgrid(h_, grid.handle())); }
    void set_featurespace(const Matrix &source_features, int D) { check(msm_cost_set_source_features(h_, source_features.data(), D)); }
    void set_dataaffintyweighting(const Matrix &W, int rows) { check(msm_cost_set_cfweight(h_, W.data(), rows)); }
    void set_spacings(const std::vector<double> &MAXSEP, double MVDmax) { check(msm_cost_set_spacings(h_, MAXSEP.data(), MVDmax)); }
    void set_labels(const Points &labels, const std::vector<double> &ROT) {
        L_ = (int)(labels.size() / 3);
        check(msm_cost_set_labels(h_, to_soa(labels).data(), L_, ROT.data()));
    }
    void setTriplets(const std::vector<int32_t> &t) {
        T_ = (int)(t.size() / 3);
        check(msm_cost_set_triplets(h_, t.data(), T_));
    }
    void setPairs(const std::vector<int32_t> &p) {
        P_ = (int)(p.size() / 2);
        check(msm_cost_set_pairs(h_, p.data(), P_));
    }
    // set_anatomical + set_anatomical_neighbourhood (:160-170); weights / faces as CSR (see msmhip.h)
    void set_anatomical(Mesh &targetS, const Points &targetA, const Points &sourceA, const Triangles &sourceA_tri, const SparseWeights &weights,
                        const std::vector<int32_t> &face_ptr, const std::vector<int32_t> &face_idx) {
        check(msm_cost_set_anatomical(h_, targetS.handle(), to_soa(targetA).data(), to_soa(sourceA).data(), (int32_t)(sourceA.size() / 3),
                                      tri_to_soa(sourceA_tri).data(), (int32_t)(sourceA_tri.size() / 3), weights.row_ptr.data(), weights.col.data(),
                                      weights.val.data(), face_ptr.data(), face_idx.data()));
    }
    // DiscreteCostFunction::initialize, M/DiscreteCostFunction.cpp:27-53: allocates and zero-fills the tables
    void initialize(int numNodes, int numLabels, int numPairs, int numTriplets) {
        (void)numTriplets;
        unarycosts.assign((size_t)numNodes * numLabels, 0.0);
        paircosts.assign((size_t)numPairs * numLabels * numLabels, 0.0);
    }
    void get_source_data() { check(msm_cost_get_source_data(h_)); }

    // ---- evaluators (M/DiscreteCostFunction.h:48-59)
    void computeUnaryCosts() {  // unarycosts[label * numNodes + node], :236-243
        unarycosts.resize((size_t)N_ * L_);
        check(msm_cost_unary_table(h_, unarycosts.data()));
    }
    double computeUnaryCost(int node, int label) {
        double v;
        const int32_t n = node, l = label;
        check(msm_cost_unary_batch(h_, &n, &l, 1, &v));
        return v;
    }
    void computePairwiseCosts() {  // paircosts[(pair * L + labelB) * L + labelA], :228-234
        paircosts.resize((size_t)P_ * L_ * L_);
        check(msm_cost_pairwise_table(h_, paircosts.data()));
    }
    double computePairwiseCost(int pair, int labelA, int labelB) {
        double v;
        const int32_t p = pair, a = labelA, b = labelB;
        check(msm_cost_pairwise_batch(h_, &p, &a, &b, 1, &v));
        return v;
    }
    double computeTripletCost(int triplet, int labelA, int labelB, int labelC) {
        double v;
        const int32_t t = triplet, a = labelA, b = labelB, c = labelC;
        check(msm_cost_triplet_batch(h_, &t, &a, &b, &c, 1, &v));
        return v;
    }
    // computeTripletCosts, :245-253: tcosts[t][a][b][c] flat (the table MCMC::optimise reads)
    std::vector<double> computeTripletCosts() {
        std::vector<double> tcosts((size_t)T_ * L_ * L_ * L_);
        if (!tcosts.empty()) check(msm_cost_triplet_table(h_, 0, T_, tcosts.data()));
        return tcosts;
    }
    // the batched forms the optimisers' loops collapse to (I/Fusion/Fusion.h:138-196)
    std::vector<double> computeTripletCost(const std::vector<int32_t> &t, const std::vector<int32_t> &a, const std::vector<int32_t> &b,
                                           const std::vector<int32_t> &c) {
        std::vector<double> out(t.size());
        check(msm_cost_triplet_batch(h_, t.data(), a.data(), b.data(), c.data(), (int32_t)t.size(), out.data()));
        return out;
    }
    std::vector<double> tripletOctets(const std::vector<int32_t> &labeling, int label) {  // E[8 t + k], k = 000..111
        std::vector<double> E(8 * (size_t)T_);
        check(msm_cost_triplet_octets(h_, labeling.data(), label, E.data()));
        return E;
    }
    double evaluateTotalCostSum(const std::vector<int32_t> &labeling) {  // :55-77
        double total, parts[3];
        check(msm_cost_total(h_, labeling.data(), &total, parts));
        return total;
    }
    std::vector<double> AbsoluteWeights() {  // resample_weights, :303-323
        std::vector<double> w(N_);
        check(msm_cost_absolute_weights(h_, w.data()));
        return w;
    }
    int getNumNodes() const { return N_; }
    int getNumLabels() const { return L_; }
    msm_cost *handle() const { return h_; }

    std::vector<double> unarycosts, paircosts;  // the tables FastPD / MCMC read (M/DiscreteCostFunction.h:35-38)

private:
    msm_cost *h_ = nullptr;
    int N_ = 0, L_ = 0, T_ = 0, P_ = 0;
};

// ---------------------------------------------------------------- groupwise model (gMSM)
struct GroupParameters {
    int simmeasure = 2;
    bool fixnan = false;
    double lambda = 0.1, shearmodulus = 0.1, bulkmodulus = 10.0, kexponent = 2.0, exponent = 2.0, range = 1.0;
    double percentile = 0.75;
};

// DiscreteGroupModel (the optimisers' DiscreteModel) + DiscreteGroupCostFunction, M/DiscreteGroupModel.h:37-108
class DiscreteGroupModel {
public:
    DiscreteGroupModel(Context &ctx, const GroupParameters &P, int num_subjects) {
        msm_group_params p{P.simmeasure, P.fixnan ? 1 : 0, P.lambda, P.shearmodulus, P.bulkmodulus, P.kexponent, P.exponent, P.range, P.percentile};
        h_ = msm_group_create(ctx.handle(), &p, num_subjects);
        if (!h_) throw Error(MSM_ERR_INVALID, msm_last_error());
    }
    ~DiscreteGroupModel() { msm_group_destroy(h_); }
    DiscreteGroupModel(const DiscreteGroupModel &) = delete;
    DiscreteGroupModel &operator=(const DiscreteGroupModel &) = delete;

    void set_meshspace(Mesh &target_space, const std::vector<double> *mask = nullptr) {  // + set_masks
        check(msm_group_set_template(h_, target_space.handle(), mask ? mask->data() : nullptr));
    }
    void Initialize(const Points &controlgrid, const Triangles &tri) {  // M/DiscreteGroupModel.cpp:141-161
        check(msm_group_set_controlgrid(h_, to_soa(controlgrid).data(), tri_to_soa(tri).data(), (int32_t)(controlgrid.size() / 3), (int32_t)(tri.size() / 3)));
    }
    void reset_meshspace(Mesh &source, const Matrix &features, int D, int num) { check(msm_group_set_subject(h_, num, source.handle(), features.data(), D)); }
    void reset_CPgrid(const Points &grid, int num) { check(msm_group_reset_cpgrid(h_, num, to_soa(grid).data())); }
    void set_labels(const Points &labels) { check(msm_group_set_labels(h_, to_soa(labels).data(), (int32_t)(labels.size() / 3))); }
    // 0: the reference's list order; 1: control point by control point (the layout for lists sharded over ranks), msm_group_set_pair_layout
    void set_pair_layout(int layout) { check(msm_group_set_pair_layout(h_, layout)); }
    // 1 (default): the data vertices' rotation matrices from the host's libm, as the reference; 0: the device computes them (msm_group_set_rotation_mode)
    void set_rotation_mode(int mode) { check(msm_group_set_rotation_mode(h_, mode)); }
    void setupCostFunction() { check(msm_group_setup(h_)); }  // :163-196
    int getNumNodes() {
        int32_t n, p, t;
        check(msm_group_sizes(h_, &n, &p, &t));
        return n;
    }
    std::vector<int32_t> getPairs() {
        int32_t n, p, t;
        check(msm_group_sizes(h_, &n, &p, &t));
        std::vector<int32_t> out(2 * (size_t)p);
        check(msm_group_get_pairs(h_, out.data()));
        return out;
    }
    std::vector<int32_t> getTriplets() {
        int32_t n, p, t;
        check(msm_group_sizes(h_, &n, &p, &t));
        std::vector<int32_t> out(3 * (size_t)t);
        check(msm_group_get_triplets(h_, out.data()));
        return out;
    }
    // one label step of Fusion::optimize (I/Fusion/Fusion.h:157-196): pair_data[p].buffer[0..3] and triplet_data[t].buffer[0..7]
    void fusionMove(const std::vector<int32_t> &labeling, int label, std::vector<double> &pair_quads, std::vector<double> &triplet_octets) {
        int32_t nodes = 0, pairs = 0, triplets = 0;
        check(msm_group_sizes(h_, &nodes, &pairs, &triplets));
        pair_quads.resize(4 * (size_t)pairs);
        triplet_octets.resize(8 * (size_t)triplets);
        check(msm_group_fusion_move(h_, labeling.data(), label, pair_quads.data(), triplet_octets.data()));
    }
    std::vector<double> computePairwiseCost(const std::vector<int32_t> &pair, const std::vector<int32_t> &a, const std::vector<int32_t> &b) {
        std::vector<double> out(pair.size());
        check(msm_group_pairwise_batch(h_, pair.data(), a.data(), b.data(), (int32_t)pair.size(), out.data()));
        return out;
    }
    std::vector<double> computeTripletCost(const std::vector<int32_t> &t, const std::vector<int32_t> &a, const std::vector<int32_t> &b,
                                           const std::vector<int32_t> &c) {
        std::vector<double> out(t.size());
        check(msm_group_triplet_batch(h_, t.data(), a.data(), b.data(), c.data(), (int32_t)t.size(), out.data()));
        return out;
    }
    msm_group *handle() const { return h_; }

private:
    msm_group *h_ = nullptr;
};

// ---------------------------------------------------------------- the optimisers' view: newmeshreg::DiscreteModel
// Fusion::optimize (I/Fusion/Fusion.h:122-244) asks for costs one clique at a time from OpenMP threads:
//     2 N  computeUnaryCost(node, labeling[node] | label)                            :148-155
//     4 P  computePairwiseCost(pair, labeling[A] | label, labeling[B] | label)       :164-174
//     8 T  computeTripletCost(triplet, ... the eight combinations ...)               :181-196
// per label step; FPD::FastPD reads getUnaryCosts() and calls computePairwiseCost on demand (I/FastPD/FastPD.h:126,213,224).
// The classes below answer those calls without an edit of Fusion.h / FastPD.h: the first clique call of a label step finds
// that its labels are not in the cached step, takes the lock, evaluates the WHOLE step with one ABI call (the proposed label
// is the argument that differs from the model's labeling -- or, for the all-current combination Fusion asks first, the label
// the unary sweep has just proposed) and every further call of the step, from any thread, is a table look-up under a shared
// lock.  Correctness never depends on that guess: a call is served from the cache only if each of its labels equals the
// cached labeling's or the cached proposed label at its node; anything else (total-cost sweeps with arbitrary labels, another
// optimiser) is evaluated on its own.
namespace detail {
// pinned, GPU-mapped memory for the per-step buffers (msm_host_alloc): the kernels write the costs where the optimiser reads them
class HostBuffer {
public:
    HostBuffer() = default;
    HostBuffer(const HostBuffer &) = delete;
    HostBuffer &operator=(const HostBuffer &) = delete;
    ~HostBuffer() { release(); }
    double *ensure(msm_ctx *ctx, size_t n) {
        if (n > cap_) {
            release();
            ctx_ = ctx;
            p_ = static_cast<double *>(msm_host_alloc(ctx, sizeof(double) * (n ? n : 1)));
            if (!p_) throw Error(MSM_ERR_HIP, msm_last_error());
            cap_ = n;
        }
        return p_;
    }
    double *data() const { return p_; }

private:
    void release() {
        if (p_) msm_host_free(ctx_, p_);
        p_ = nullptr;
        cap_ = 0;
    }
    msm_ctx *ctx_ = nullptr;
    double *p_ = nullptr;
    size_t cap_ = 0;
};
}  // namespace detail

#ifndef MSMHIP_STEP_COSTS_DEFINED
#define MSMHIP_STEP_COSTS_DEFINED
// The costs of one label step as the kernels deliver them (also declared by msmhip_fusion.hpp, whose fusion_optimize reads them)
struct StepCosts {
    const double *unary_table = nullptr;     // unarycosts[label * num_nodes + node], or null: every unary cost is 0
    const double *pair_quads = nullptr;      // [4 p + k], k = 00 01 10 11 (I/Fusion/Fusion.h:170-173)
    const double *triplet_octets = nullptr;  // [8 t + k], k = 000 .. 111 (I/Fusion/Fusion.h:188-195)
};
#endif

struct FusionCounters {
    std::atomic<long> step_calls{0};    // whole label steps evaluated (one ABI call each)
    std::atomic<long> served{0};        // clique costs answered from a cached step / table
    std::atomic<long> single_calls{0};  // clique costs that had to be evaluated on their own
};

// Pairwise registration: NonLinearSRegDiscreteModel (M/DiscreteModel.h:90-190) over a msmhip::DiscreteCostFunction.
class FusionModel {
public:
    // costfct: fully set up (meshes, features, labels, cliques, get_source_data()); the model keeps the labeling
    FusionModel(Context &ctx, DiscreteCostFunction &costfct, const std::vector<int32_t> &triplets, const std::vector<int32_t> &pairs)
        : ctx_(ctx.handle()), cf_(costfct), triplets_(triplets), pairs_(pairs), labeling_(costfct.getNumNodes(), 0) {}

    int getNumNodes() const { return cf_.getNumNodes(); }
    int getNumLabels() const { return cf_.getNumLabels(); }
    int getNumPairs() const { return (int)(pairs_.size() / 2); }
    int getNumTriplets() const { return (int)(triplets_.size() / 3); }
    int *getLabeling() { return labeling_.data(); }  // mutable, owned by the model (M/DiscreteModel.h:48)
    const int *getPairs() const { return pairs_.data(); }
    const int *getTriplets() const { return triplets_.data(); }

    // setupCostFunction's tail (M/DiscreteModel.cpp:216-262): new labels / grid -> the cached step is void; the tables the
    // optimisers read (unarycosts for FastPD / MCMC and the 2 N look-ups of Fusion; paircosts for FastPD's PAIR macro) are
    // computed once here
    void setupCostFunction(bool with_pair_table = false) {
        std::unique_lock<std::shared_mutex> wr(mu_);
        step_valid_ = false;
        hint_.store(-1, std::memory_order_relaxed);
        cf_.computeUnaryCosts();
        have_pairs_ = false;
        if (with_pair_table && getNumPairs() > 0) {
            cf_.computePairwiseCosts();
            have_pairs_ = true;
        }
    }
    const double *getUnaryCosts() const { return cf_.unarycosts.data(); }  // unarycosts[label * numNodes + node]

    double computeUnaryCost(int node, int label) {  // I/Fusion/Fusion.h:151-152
        if (label != labeling_[(size_t)node]) hint_.store(label, std::memory_order_relaxed);  // the label this step proposes
        counters.served.fetch_add(1, std::memory_order_relaxed);
        return cf_.unarycosts[(size_t)label * getNumNodes() + node];
    }
    double computePairwiseCost(int pair, int labelA, int labelB) {  // regoption 1: I/FastPD/FastPD.h:213,224 / I/Fusion/Fusion.h:170-173
        if (have_pairs_) {
            counters.served.fetch_add(1, std::memory_order_relaxed);
            return cf_.paircosts[((size_t)pair * getNumLabels() + labelB) * getNumLabels() + labelA];
        }
        std::unique_lock<std::shared_mutex> wr(mu_);
        counters.single_calls.fetch_add(1, std::memory_order_relaxed);
        return cf_.computePairwiseCost(pair, labelA, labelB);
    }
    double computeTripletCost(int triplet, int labelA, int labelB, int labelC) {  // I/Fusion/Fusion.h:188-195
        const int32_t *n = &triplets_[3 * (size_t)triplet];
        double v;
        {
            std::shared_lock<std::shared_mutex> rd(mu_);
            if (lookup(triplet, n, labelA, labelB, labelC, v)) return v;
        }
        std::unique_lock<std::shared_mutex> wr(mu_);
        if (lookup(triplet, n, labelA, labelB, labelC, v)) return v;  // another thread has evaluated the step meanwhile
        int label = labelA != labeling_[n[0]] ? labelA : (labelB != labeling_[n[1]] ? labelB : (labelC != labeling_[n[2]] ? labelC : hint_.load(std::memory_order_relaxed)));
        const bool fits = label >= 0 && label < getNumLabels() && (labelA == labeling_[n[0]] || labelA == label) && (labelB == labeling_[n[1]] || labelB == label) &&
                          (labelC == labeling_[n[2]] || labelC == label);
        if (fits) {  // a new label step: all 8 T costs with one call, into the buffer the kernel writes directly
            step_lab_ = labeling_;
            step_label_ = label;
            evaluate_step(label);
            step_valid_ = true;
            counters.step_calls.fetch_add(1, std::memory_order_relaxed);
            if (lookup(triplet, n, labelA, labelB, labelC, v)) return v;
        }
        counters.single_calls.fetch_add(1, std::memory_order_relaxed);  // not part of a fusion move: on its own
        return cf_.computeTripletCost(triplet, labelA, labelB, labelC);
    }
    double evaluateTotalCostSum() {  // M/DiscreteCostFunction.cpp:55-77 over the model's labeling
        std::unique_lock<std::shared_mutex> wr(mu_);
        return cf_.evaluateTotalCostSum(labeling_);
    }
    // A whole label step for a caller that reads buffers instead of asking clique by clique (msmhip::fusion_optimize): one ABI call for the
    // 8 T triplet costs; the 4 P pairwise costs of regoption 1 are read out of the pair table.  Valid until the next labelStep / setup.
    StepCosts labelStep(int label) {
        std::unique_lock<std::shared_mutex> wr(mu_);
        const int L = getNumLabels(), P = getNumPairs(), T = getNumTriplets();
        if (label < 0 || label >= L) throw Error(MSM_ERR_INVALID, "labelStep: label out of range");
        StepCosts out;
        out.unary_table = cf_.unarycosts.data();
        step_lab_ = labeling_;
        step_label_ = label;
        if (P > 0) {
            if (!have_pairs_) {
                cf_.computePairwiseCosts();
                have_pairs_ = true;
            }
            step_quads_.resize(4 * (size_t)P);
            for (int p = 0; p < P; ++p) {
                const int a = labeling_[(size_t)pairs_[2 * (size_t)p]], b = labeling_[(size_t)pairs_[2 * (size_t)p + 1]];
                const double *tab = &cf_.paircosts[(size_t)p * L * L];  // [labelB * L + labelA]
                double *q = &step_quads_[4 * (size_t)p];
                q[0] = tab[(size_t)b * L + a];
                q[1] = tab[(size_t)label * L + a];
                q[2] = tab[(size_t)b * L + label];
                q[3] = tab[(size_t)label * L + label];
            }
            out.pair_quads = step_quads_.data();
        }
        if (T > 0) {
            out.triplet_octets = evaluate_step(label);
            step_valid_ = true;
        }
        counters.step_calls.fetch_add(1, std::memory_order_relaxed);
        return out;
    }
    FusionCounters counters;
    bool speculate = true;  // queue the next label step's evaluations while the optimiser solves the current one (msm_cost_triplet_octets_prefetch)

private:
    // The 8 T costs of the step (step_lab_, label) into one of two buffers used in turn, then -- a hint -- the NEXT step of Fusion's sweep over the labels
    // (I/Fusion/Fusion.h:136-140: label + 1, wrapping into the second sweep) queued into the other for the labeling as it stands: while ELC + FastPD solve this
    // step the GPU would idle, and a step that accepts no proposal (most of a converging level) leaves the next step's evaluations exactly these.  A step
    // whose labeling did change is evaluated afresh; the buffer a solve is reading is never the one being written.
    double *evaluate_step(int label) {
        const size_t n = 8 * triplets_.size() / 3;
        double *E = octets_[turn_].ensure(ctx_, n);
        check(msm_cost_triplet_octets(cf_.handle(), step_lab_.data(), label, E));
        cur_ = E;
        turn_ ^= 1;
        if (speculate && getNumLabels() > 1) {
            double *next = octets_[turn_].ensure(ctx_, n);
            check(msm_cost_triplet_octets_prefetch(cf_.handle(), step_lab_.data(), (label + 1) % getNumLabels(), next));
        }
        return E;
    }
    bool lookup(int t, const int32_t *n, int la, int lb, int lc, double &v) {
        if (!step_valid_) return false;
        int k = 0;
        const int arg[3] = {la, lb, lc};
        for (int j = 0; j < 3; ++j) {
            const int cur = step_lab_[(size_t)n[j]];
            if (arg[j] == cur) k = 2 * k;  // also when cur == the proposed label: both combinations cost the same
            else if (arg[j] == step_label_) k = 2 * k + 1;
            else return false;
        }
        v = cur_[8 * (size_t)t + k];
        counters.served.fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    msm_ctx *ctx_;
    DiscreteCostFunction &cf_;
    std::vector<int32_t> triplets_, pairs_, labeling_;
    std::shared_mutex mu_;
    std::atomic<int> hint_{-1};
    bool step_valid_ = false, have_pairs_ = false;
    std::vector<int32_t> step_lab_;
    std::vector<double> step_quads_;
    int step_label_ = -1;
    detail::HostBuffer octets_[2];
    double *cur_ = nullptr;
    int turn_ = 0;
};

// Groupwise registration: DiscreteGroupModel (M/DiscreteGroupModel.h:37-108) as Fusion::optimize calls it.  computeUnaryCost is 0
// (M/DiscreteGroupCostFunction.h), pairs and triplets of a label step come from one msm_group_fusion_move.
class GroupFusionModel {
public:
    GroupFusionModel(Context &ctx, DiscreteGroupModel &model) : ctx_(ctx.handle()), m_(model) { refresh_sizes(); }
    void setupCostFunction() {  // M/DiscreteGroupModel.cpp:163-196
        std::unique_lock<std::shared_mutex> wr(mu_);
        m_.setupCostFunction();
        refresh_sizes();
        pairs_ = m_.getPairs();
        triplets_ = m_.getTriplets();
        step_valid_ = false;
        hint_.store(-1, std::memory_order_relaxed);
    }
    int getNumNodes() const { return nodes_; }
    int getNumPairs() const { return (int)(pairs_.size() / 2); }
    int getNumTriplets() const { return (int)(triplets_.size() / 3); }
    int *getLabeling() { return labeling_.data(); }
    const int *getPairs() const { return pairs_.data(); }
    const int *getTriplets() const { return triplets_.data(); }

    double computeUnaryCost(int node, int label) {
        if (label != labeling_[(size_t)node]) hint_.store(label, std::memory_order_relaxed);
        return 0.0;
    }
    int getNumLabels() const { return labels_; }
    // DiscreteCostFunction::evaluateTotalCostSum (M/DiscreteCostFunction.cpp:55-77) over the model's labeling: pairs, then triplets, each
    // summed serially in clique order (the unary costs of the group model are 0)
    double evaluateTotalCostSum() {
        std::unique_lock<std::shared_mutex> wr(mu_);
        const size_t P = pairs_.size() / 2, T = triplets_.size() / 3;
        std::vector<int32_t> id(std::max(P, T)), la(std::max(P, T)), lb(std::max(P, T)), lc(T);
        double total = 0.0;
        if (P) {
            for (size_t p = 0; p < P; ++p) {
                id[p] = (int32_t)p;
                la[p] = labeling_[(size_t)pairs_[2 * p]];
                lb[p] = labeling_[(size_t)pairs_[2 * p + 1]];
            }
            std::vector<double> c(P);
            check(msm_group_pairwise_batch(m_.handle(), id.data(), la.data(), lb.data(), (int32_t)P, c.data()));
            for (size_t p = 0; p < P; ++p) total += c[p];
        }
        if (T) {
            for (size_t t = 0; t < T; ++t) {
                id[t] = (int32_t)t;
                la[t] = labeling_[(size_t)triplets_[3 * t]];
                lb[t] = labeling_[(size_t)triplets_[3 * t + 1]];
                lc[t] = labeling_[(size_t)triplets_[3 * t + 2]];
            }
            std::vector<double> c(T);
            check(msm_group_triplet_batch(m_.handle(), id.data(), la.data(), lb.data(), lc.data(), (int32_t)T, c.data()));
            for (size_t t = 0; t < T; ++t) total += c[t];
        }
        return total;
    }
    // a whole label step for msmhip::fusion_optimize: one msm_group_fusion_move into the pinned buffers
    StepCosts labelStep(int label) {
        std::unique_lock<std::shared_mutex> wr(mu_);
        if (label < 0 || label >= labels_) throw Error(MSM_ERR_INVALID, "labelStep: label out of range");
        step_lab_ = labeling_;
        step_label_ = label;
        double *q = quads_.ensure(ctx_, 4 * pairs_.size() / 2), *o = octets_.ensure(ctx_, 8 * triplets_.size() / 3);
        check(msm_group_fusion_move(m_.handle(), step_lab_.data(), label, q, o));
        step_valid_ = true;
        counters.step_calls.fetch_add(1, std::memory_order_relaxed);
        StepCosts out;
        out.pair_quads = q;
        out.triplet_octets = o;
        return out;
    }
    double computePairwiseCost(int pair, int labelA, int labelB) {  // M/DiscreteGroupCostFunction.cpp:54-98
        const int32_t *n = &pairs_[2 * (size_t)pair];
        const int arg[2] = {labelA, labelB};
        double v;
        {
            std::shared_lock<std::shared_mutex> rd(mu_);
            if (lookup(quads_.data(), 4, pair, n, arg, 2, v)) return v;
        }
        std::unique_lock<std::shared_mutex> wr(mu_);
        if (lookup(quads_.data(), 4, pair, n, arg, 2, v)) return v;
        if (new_step(n, arg, 2) && lookup(quads_.data(), 4, pair, n, arg, 2, v)) return v;
        counters.single_calls.fetch_add(1, std::memory_order_relaxed);
        return m_.computePairwiseCost({pair}, {labelA}, {labelB})[0];
    }
    double computeTripletCost(int triplet, int labelA, int labelB, int labelC) {  // M/DiscreteGroupCostFunction.cpp:26-52
        const int32_t *n = &triplets_[3 * (size_t)triplet];
        const int arg[3] = {labelA, labelB, labelC};
        double v;
        {
            std::shared_lock<std::shared_mutex> rd(mu_);
            if (lookup(octets_.data(), 8, triplet, n, arg, 3, v)) return v;
        }
        std::unique_lock<std::shared_mutex> wr(mu_);
        if (lookup(octets_.data(), 8, triplet, n, arg, 3, v)) return v;
        if (new_step(n, arg, 3) && lookup(octets_.data(), 8, triplet, n, arg, 3, v)) return v;
        counters.single_calls.fetch_add(1, std::memory_order_relaxed);
        return m_.computeTripletCost({triplet}, {labelA}, {labelB}, {labelC})[0];
    }
    FusionCounters counters;

private:
    void refresh_sizes() {
        nodes_ = m_.getNumNodes();
        labeling_.resize((size_t)nodes_, 0);
        int32_t S = 0, N = 0, L = 0, D = 0, Vt = 0;
        if (msm_group_dims(m_.handle(), &S, &N, &L, &D, &Vt) == MSM_OK) labels_ = L;
    }
    bool lookup(const double *buf, int width, int clique, const int32_t *n, const int *arg, int arity, double &v) {
        if (!step_valid_) return false;
        int k = 0;
        for (int j = 0; j < arity; ++j) {
            const int cur = step_lab_[(size_t)n[j]];
            if (arg[j] == cur) k = 2 * k;
            else if (arg[j] == step_label_) k = 2 * k + 1;
            else return false;
        }
        v = buf[(size_t)width * clique + k];
        counters.served.fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    bool new_step(const int32_t *n, const int *arg, int arity) {  // called with the lock held
        int label = hint_.load(std::memory_order_relaxed);
        for (int j = arity - 1; j >= 0; --j)
            if (arg[j] != labeling_[(size_t)n[j]]) label = arg[j];
        if (label < 0) return false;
        for (int j = 0; j < arity; ++j)
            if (arg[j] != labeling_[(size_t)n[j]] && arg[j] != label) return false;
        step_lab_ = labeling_;
        step_label_ = label;
        double *q = quads_.ensure(ctx_, 4 * pairs_.size() / 2), *o = octets_.ensure(ctx_, 8 * triplets_.size() / 3);
        check(msm_group_fusion_move(m_.handle(), step_lab_.data(), label, q, o));
        step_valid_ = true;
        counters.step_calls.fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    msm_ctx *ctx_;
    DiscreteGroupModel &m_;
    int nodes_ = 0, labels_ = 0;
    std::vector<int32_t> pairs_, triplets_, labeling_;
    std::shared_mutex mu_;
    std::atomic<int> hint_{-1};
    bool step_valid_ = false;
    std::vector<int32_t> step_lab_;
    int step_label_ = -1;
    detail::HostBuffer quads_, octets_;
};

}  // namespace msmhip
