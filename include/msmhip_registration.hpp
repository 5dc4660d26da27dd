// msmhip_registration.hpp -- one resolution level of a discrete registration driven through msmhip.hpp, the way newMSM's
// callers drive the path with the Monte Carlo optimiser:
//     Mesh_registration::run_discrete_opt             M/mesh_registration.cpp:164-232
//     NonLinearSRegDiscreteModel::Initialize           M/DiscreteModel.cpp:63-108
//     NonLinearSRegDiscreteModel::setupCostFunction    M/DiscreteModel.cpp:216-262
//     NonLinearSRegDiscreteModel::applyLabeling        M/DiscreteModel.cpp:264-269
//     MCMC::optimise                                   M/mcmc_opt.h:31-134
// Header only, C++17, no HIP headers.  newmsm_amd/registration.py is the same loop in Python (used by the parity tests, which
// also run it over the oracle); tests/test_cpp_host.py checks that the two give identical results.
#ifndef MSMHIP_REGISTRATION_HPP
#define MSMHIP_REGISTRATION_HPP

#include "msmhip.hpp"

namespace msmhip {

struct LevelOptions {
    int sg_order = -1;  // sampling grid resolution; cp_order + 2 when negative
    int iters = 3;      // --it
    int mciters = 200;  // --mciters
    double mcparam = 0.8;
    uint64_t seed = 0;  // iteration i draws from std::mt19937(seed + i); the reference seeds from std::random_device
    double labeldist = 0.5;
    bool rescale_labels = false;
    Parameters cost;    // kind, similarity measure, regulariser
};

struct LevelResult {
    Points sph_reg, cpgrid;
    std::vector<double> energies;
    std::vector<std::vector<int32_t>> labelings;
};

// m_CPgrid.set_coord(i, m_ROT[i] * m_labels[labeling[i]]), operator*(Matrix, Point) R/point.cpp:207-213
inline Points apply_labeling(const std::vector<double> &ROT, const Points &labels, const std::vector<int32_t> &labeling) {
    const size_t N = labeling.size();
    Points out(3 * N);
    for (size_t i = 0; i < N; ++i) {
        const double *R = &ROT[9 * i], *v = &labels[3 * (size_t)labeling[i]];
        for (int r = 0; r < 3; ++r) out[3 * i + r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
    }
    return out;
}

// target / source: the reference and the moving sphere at this level's data resolution with their D x V features;
// sph_reg: the current registered position of the source sphere; cp_start: the control grid after warp_CPgrid, or null
inline LevelResult run_discrete_opt(Context &ctx, const Points &target_xyz, const Triangles &target_tri, const Matrix &ref_feat,
                                    const Points &source_xyz, const Triangles &source_tri, const Matrix &src_feat, int D, Points sph_reg,
                                    int cp_order, const LevelOptions &o, const Points *cp_start = nullptr) {
    // ---- initialize_level / Initialize(CONTROL)
    auto [cp_xyz, cp_tri] = make_mesh_from_icosa(cp_order);
    Mesh TARGET(ctx, target_xyz, target_tri), SOURCE(ctx, source_xyz, source_tri), CPGRID(ctx, cp_xyz, cp_tri);
    TARGET.set_pvalues(ref_feat);
    auto [MAXSEP, MVDmax] = cp_spacings(cp_xyz, cp_tri);
    auto [samples, barycentres] = label_sampling_grid(o.sg_order < 0 ? cp_order + 2 : o.sg_order, o.labeldist * MVDmax);
    const double centre[3] = {samples[0], samples[1], samples[2]};
    const std::vector<int32_t> triplets = estimate_triplets(cp_tri);
    DiscreteCostFunction costfct(ctx, o.cost);
    costfct.set_meshes(TARGET, SOURCE, CPGRID);  // _ORIG, _oCPgrid
    costfct.set_featurespace(src_feat, D);
    costfct.set_spacings(MAXSEP, MVDmax);
    const int N = (int)(cp_xyz.size() / 3);
    int m_iter = 1;
    double m_scale = 1.0;
    if (cp_start) cp_xyz = *cp_start;
    LevelResult res;
    for (int it = 0; it < o.iters; ++it) {
        // ---- reset_meshspace + setupCostFunction
        SOURCE.set_coords(sph_reg);
        costfct.reset_source(SOURCE);
        CPGRID.set_coords(cp_xyz);
        costfct.reset_CPgrid(CPGRID);
        const std::vector<double> ROT = cp_rotations(centre, cp_xyz);
        Points labels;
        if (o.rescale_labels) labels = rescale_sampling_grid(samples, m_scale);
        else labels = (m_iter % 2 == 0) ? samples : barycentres;
        costfct.set_labels(labels, ROT);
        costfct.get_source_data();
        costfct.setTriplets(triplets);
        ++m_iter;
        // ---- MCMC: computeUnaryCosts, computeTripletCosts, optimise
        costfct.computeUnaryCosts();
        const std::vector<double> tcosts = costfct.computeTripletCosts();
        std::vector<int32_t> labeling((size_t)N, 0);  // resetLabeling
        mcmc_optimise(costfct.unarycosts, tcosts, triplets, N, (int)(labels.size() / 3), o.mcparam, o.mciters, o.seed + (uint64_t)it, labeling);
        res.energies.push_back(costfct.evaluateTotalCostSum(labeling));
        res.labelings.push_back(labeling);
        // ---- applyLabeling, warp the source through the control grid's move, unfold both (:219-230)
        const Points moved = apply_labeling(ROT, labels, labeling);
        sph_reg = sphere_project_warp(sph_reg, CPGRID, moved);  // CPGRID still holds the previous grid
        CPGRID.set_coords(moved);
        unfold(CPGRID);
        cp_xyz = CPGRID.get_coords();
        SOURCE.set_coords(sph_reg);
        unfold(SOURCE);
        sph_reg = SOURCE.get_coords();
    }
    res.sph_reg = sph_reg;
    res.cpgrid = cp_xyz;
    return res;
}

}  // namespace msmhip

#endif
