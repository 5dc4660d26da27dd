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
//
// Conventions: points are AoS (x0 y0 z0 x1 ...) as in newresampler::Point containers and are transposed to the
// ABI's SoA here; triangles are AoS (3 ids per triangle); data matrices are row-major D x V like
// newresampler::Mesh::pvalues.  Errors are thrown as msmhip::Error carrying the reference's exception text.
#pragma once

#include <cstdint>
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
// sphere_project_warp, R/resampler.cpp:311-328: `sphere` is carried through the deformation from -> to
inline Points sphere_project_warp(const Points &sphere, Mesh &from, const Points &to) {
    std::vector<double> s = to_soa(sphere);
    check(msm_sphere_project_warp(from.handle(), to_soa(to).data(), s.data(), (int32_t)(sphere.size() / 3)));
    return to_aos(s);
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

}  // namespace msmhip
