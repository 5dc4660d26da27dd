// msmhip_group_registration.hpp -- a groupwise (gMSM) registration with the host side in C++: the caller loops of
//     Group_Mesh_registration::run_multiresolutions   M/mesh_registration.cpp:30-50 over the overrides of M/group_mesh_registration.cpp:26-133
//     Group_Mesh_registration::run_discrete_opt       M/group_mesh_registration.cpp:70-118
//     DiscreteGroupModel::Initialize / setupCostFunction / applyLabeling   M/DiscreteGroupModel.cpp:145-197, .h:74-78
//     Fusion::optimize (the label loop)               I/Fusion/Fusion.h:136-229
// over msmhip::DiscreteGroupModel (msmhip.hpp) -- the C++ twin of newmsm_amd/group_registration.py: the same library calls in the same order
// (tests/test_cpp_host.py compares the two: identical labelings, energies and coordinates).  The binary solve of a label step (ELC + FastPD:
// licence-restricted, FSL-bound) is the stand-in of msm_fusion_icm_step, as in msmhip_registration.hpp.  Header only; needs msmhip.hpp.
#ifndef MSMHIP_GROUP_REGISTRATION_HPP
#define MSMHIP_GROUP_REGISTRATION_HPP

#include "msmhip_config.hpp"
#include "msmhip_registration.hpp"

namespace msmhip {

struct GroupLevelOptions {
    int iters = 2;
    int sg_order = -1;       // --SGgrid (m_SGres); < 0: two above the control grid
    int icm_passes = 5;      // the stand-in solve
    double labeldist = 0.5;  // _labeldist, M/DiscreteModel.h:167
    GroupParameters cost;    // --simval --lambda --fixnan --shearmod --bulkmod --k_exponent --regexp --cprange --percentile
};

struct GroupLevelResult {
    std::vector<Points> sph_regs, cpgrids;        // per subject: the registered data grid and the control grid at the end of the level
    std::vector<double> energies;                 // per iteration
    std::vector<std::vector<int32_t>> labelings;  // per iteration, S x N
};

// One resolution level.  feats: per subject D x V(data grid); sph_regs: per subject the registered data grid so far; cps_start (optional): the control
// grids the level starts from (warp_CPgrid of the previous level's warp, M/DiscreteGroupModel.h:69-72); mask (optional, V(template)): --mask.
inline GroupLevelResult run_group_discrete_opt(Context &ctx, const Points &template_xyz, const Triangles &template_tri, const Points &data_xyz,
                                               const Triangles &data_tri, const std::vector<Matrix> &feats, int D, std::vector<Points> sph_regs, int cp_order,
                                               const GroupLevelOptions &o, const std::vector<Points> *cps_start = nullptr,
                                               const std::vector<double> *mask = nullptr, PhaseClock *clock = nullptr) {
    const int S = (int)feats.size();
    if (S < 1 || (int)sph_regs.size() != S) throw Error(MSM_ERR_INVALID, "run_group_discrete_opt: one feature matrix and one registered sphere per subject");
    auto [cp_xyz0, cp_tri] = make_mesh_from_icosa(cp_order);
    const int N = (int)(cp_xyz0.size() / 3);
    auto [MAXSEP, MVDmax] = cp_spacings(cp_xyz0, cp_tri);
    (void)MAXSEP;
    auto [samples, barycentres] = label_sampling_grid(o.sg_order < 0 ? cp_order + 2 : o.sg_order, o.labeldist * MVDmax);  // m_labels = m_samples in every iteration, :176
    (void)barycentres;
    const double centre[3] = {samples[0], samples[1], samples[2]};
    const int L = (int)(samples.size() / 3);
    DiscreteGroupModel g(ctx, o.cost, S);
    Mesh TEMPLATE(ctx, template_xyz, template_tri);
    g.set_meshspace(TEMPLATE, mask);
    g.Initialize(cp_xyz0, cp_tri);
    std::vector<std::unique_ptr<Mesh>> meshes;
    for (int s = 0; s < S; ++s) {
        meshes.emplace_back(new Mesh(ctx, data_xyz, data_tri));
        g.reset_meshspace(*meshes.back(), feats[(size_t)s], D, s);  // set_meshspace: the original data meshes
    }
    std::vector<Points> cps((size_t)S, cp_xyz0);
    if (cps_start) cps = *cps_start;
    std::vector<Points> prev = cps;  // previous_controlgrids = model->get_CPgrid(subject), :75-78
    GroupLevelResult res;
    double energy = 0.0;
    std::vector<double> quads, octets;
    for (int it = 0; it < o.iters; ++it) {
        for (int s = 0; s < S; ++s) {
            // iteration 0 runs on the level's ORIGINAL data grid for every subject (set_meshspace(target_space, SPH_orig, S), M/group_mesh_registration.cpp:54),
            // also at later levels: project_CPgrid carries the previous warp to ALL_SPH_REG and the control grids (warp_CPgrid), not to m_datameshes;
            // reset_meshspace(ALL_SPH_REG[subject]) comes at the end of an iteration (:114)
            if (it > 0) {
                meshes[(size_t)s]->set_coords(sph_regs[(size_t)s]);
                g.reset_meshspace(*meshes[(size_t)s], feats[(size_t)s], D, s);
            }
            g.reset_CPgrid(cps[(size_t)s], s);
        }
        g.set_labels(samples);
        PhaseClock::timed(clock, "setup", [&] { g.setupCostFunction(); });
        const std::vector<int32_t> pairs = g.getPairs(), triplets = g.getTriplets();
        std::vector<int32_t> labeling((size_t)S * N, 0);  // resetLabeling
        const std::vector<double> no_unary;
        auto differs = [&](int label) {
            for (int32_t l : labeling)
                if (l != label) return true;
            return false;
        };
        for (int step = 0; step < 2 * L; ++step) {  // two sweeps over the labels, I/Fusion/Fusion.h:136-138
            const int label = step % L;
            if (!differs(label)) continue;
            PhaseClock::timed(clock, "fusion_moves", [&] { g.fusionMove(labeling, label, quads, octets); });
            const std::vector<int32_t> x =
                PhaseClock::timed(clock, "optimiser", [&] { return fusion_icm_step(S * N, no_unary, quads.data(), pairs, octets.data(), triplets, o.icm_passes); });
            for (size_t i = 0; i < labeling.size(); ++i)
                if (x[i] == 1 && labeling[i] != label) labeling[i] = label;
        }
        PhaseClock::timed(clock, "total_cost", [&] { g.fusionMove(labeling, 0, quads, octets); });
        double newenergy = 0.0, part = 0.0;  // evaluateTotalCostSum: pairs, then triplets, at the labeling
        for (size_t p = 0; p < quads.size(); p += 4) newenergy += quads[p];
        for (size_t t = 0; t < octets.size(); t += 8) part += octets[t];
        newenergy += part;
        res.energies.push_back(newenergy);
        res.labelings.push_back(labeling);
        if (it > 1 && energy - newenergy < newenergy * 0.01) break;  // :91-98
        for (int s = 0; s < S; ++s) {  // applyLabeling + the per-subject tail of the loop, :104-115
            const std::vector<double> ROT = cp_rotations(centre, cps[(size_t)s]);
            const std::vector<int32_t> mine(labeling.begin() + (size_t)s * N, labeling.begin() + (size_t)(s + 1) * N);
            Mesh moved(ctx, apply_labeling(ROT, samples, mine), cp_tri);
            PhaseClock::timed(clock, "unfold", [&] { return unfold(moved); });
            const Points new_cp = moved.get_coords();
            Mesh before(ctx, prev[(size_t)s], cp_tri);
            const Points sph = PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(sph_regs[(size_t)s], before, new_cp); });
            meshes[(size_t)s]->set_coords(sph);
            PhaseClock::timed(clock, "unfold", [&] { return unfold(*meshes[(size_t)s]); });
            sph_regs[(size_t)s] = meshes[(size_t)s]->get_coords();
            prev[(size_t)s] = new_cp;
            cps[(size_t)s] = new_cp;
        }
        energy = newenergy;
    }
    res.sph_regs = std::move(sph_regs);
    res.cpgrids = std::move(cps);
    return res;
}

struct GroupLevelSpec {
    int data_order = 5, cp_order = 2;
    double sigma_in = 0.0;
    GroupLevelOptions options;
};

struct GroupMultiresResult {
    std::vector<Points> sphere_regs;                 // per subject: the input sphere moved through the final warp ("sphere-<i>.reg", :120-125)
    std::vector<std::vector<Points>> level_regs;     // per level, per subject: the registered data grid ("sphere-<i>.LR.reg" = the last level's)
    std::vector<std::vector<double>> energies;       // per level, per iteration
    std::vector<std::vector<int32_t>> labelings;     // every iteration's labeling, level after level
};

// Group_Mesh_registration::run_multiresolutions without file I/O:
//   per level  initialize_level (:26-57): featurespace::initialise over all subjects (M/featurespace.cpp:39-86: metric_resample onto the level's
//              icosphere, smooth_data with --sigma_in, variance_normalise), the control grid, the model over (template, data grid, S);
//              evaluate (:59-68): level 1 starts every subject on the data grid, later levels carry each subject's warp to the new data grid and
//              control grid (project_CPgrid with the subject's index, M/mesh_registration.cpp:131-162), then run_discrete_opt (:70-118);
//   at the end transform (:120-125).
// meshes: per subject (xyz, tri), spheres of radius 100; datas: per subject D x V(mesh).
inline GroupMultiresResult run_group_multiresolutions(Context &ctx, const std::vector<std::pair<Points, Triangles>> &meshes, const std::vector<Matrix> &datas, int D,
                                                      const Points &template_xyz, const Triangles &template_tri, const std::vector<GroupLevelSpec> &levels,
                                                      bool varnorm, const std::vector<double> *mask = nullptr, PhaseClock *clock = nullptr) {
    const int S = (int)meshes.size();
    if ((int)datas.size() != S) throw Error(MSM_ERR_INVALID, "featurespace::Initialize do not have the same number of datasets and surface meshes");  // M/featurespace.cpp:43-44
    if (levels.empty()) throw Error(MSM_ERR_INVALID, "run_group_multiresolutions: no DISCRETE level");
    std::vector<std::unique_ptr<Mesh>> in_mesh;
    for (int s = 0; s < S; ++s) in_mesh.emplace_back(new Mesh(ctx, meshes[(size_t)s].first, meshes[(size_t)s].second));
    GroupMultiresResult res;
    std::vector<Points> prev_regs;
    int prev_order = -1;
    for (const GroupLevelSpec &lv : levels) {
        auto [ico_xyz, ico_tri] = make_mesh_from_icosa(lv.data_order);
        Mesh ico(ctx, ico_xyz, ico_tri);
        std::vector<Matrix> feats;
        for (int s = 0; s < S; ++s) {
            Matrix f = PhaseClock::timed(clock, "metric_resample", [&] { return metric_resample(*in_mesh[(size_t)s], datas[(size_t)s], ico); });
            if (lv.sigma_in > 0.0) f = PhaseClock::timed(clock, "smooth_data", [&] { return smooth_data(ico, f, ico, lv.sigma_in); });
            if (varnorm) variance_normalise(f, ico.nvertices());
            feats.push_back(std::move(f));
        }
        std::vector<Points> sph, cps_start;
        if (prev_regs.empty()) {
            sph.assign((size_t)S, ico_xyz);  // ALL_SPH_REG.resize(num_subjects, SPH_orig), :60-61
        } else {
            auto [prev_xyz, prev_tri] = make_mesh_from_icosa(prev_order);
            Mesh prev_ico(ctx, prev_xyz, prev_tri);
            auto [cp_xyz, cp_tri] = make_mesh_from_icosa(lv.cp_order);
            for (int s = 0; s < S; ++s) {
                const Points &in_xyz = meshes[(size_t)s].first;
                const Points incurrent = PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(in_xyz, prev_ico, prev_regs[(size_t)s]); });
                Mesh moved(ctx, PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(ico_xyz, *in_mesh[(size_t)s], incurrent); }), ico_tri);
                Mesh cpm(ctx, PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(cp_xyz, *in_mesh[(size_t)s], incurrent); }), cp_tri);  // warp_CPgrid
                PhaseClock::timed(clock, "unfold", [&] { return unfold(cpm); });
                PhaseClock::timed(clock, "unfold", [&] { return unfold(moved); });
                cps_start.push_back(cpm.get_coords());
                sph.push_back(moved.get_coords());
            }
        }
        GroupLevelResult r = run_group_discrete_opt(ctx, template_xyz, template_tri, ico_xyz, ico_tri, feats, D, sph, lv.cp_order, lv.options,
                                                    cps_start.empty() ? nullptr : &cps_start, mask, clock);
        res.labelings.insert(res.labelings.end(), r.labelings.begin(), r.labelings.end());
        res.energies.push_back(r.energies);
        res.level_regs.push_back(r.sph_regs);
        prev_regs = std::move(r.sph_regs);
        prev_order = lv.data_order;
    }
    auto [last_xyz, last_tri] = make_mesh_from_icosa(levels.back().data_order);
    Mesh last(ctx, last_xyz, last_tri);
    for (int s = 0; s < S; ++s)
        res.sphere_regs.push_back(PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(meshes[(size_t)s].first, last, prev_regs[(size_t)s]); }));
    return res;
}

// The levels of a --groupwise run from a configuration (msmhip_config.hpp: parse_config), as Group_Mesh_registration reads it: every level DISCRETE
// ("AFFINE/RIGID registration is not supported in groupwise mode.", M/group_mesh_registration.cpp:29-30), the optimiser HOCR (:87); the model has its own
// regulariser, so --regoption is not looked at (the twin of newmsm_amd/config.py: levels_from_config(..., groupwise=True)).
inline std::vector<GroupLevelSpec> group_levels_from_config(const Config &c, bool *varnorm = nullptr) {
    if (c.IN || c.INc) throw ConfigError("--IN / --INc (histogram matching through FSL's MISCMATHS::Histogram, M/reg_tools.cpp:745-802) is not available");
    if (c.excl) throw ConfigError("--excl (exclusion masks from the cut thresholds) is not wired into the level loop");
    for (const std::string &m : c.opt)
        if (m == "RIGID" || m == "AFFINE") throw ConfigError("AFFINE/RIGID registration is not supported in groupwise mode.");
    if (c.dopt != "HOCR") throw ConfigError("Groupwise mode is only supported in the HOCR version of MSM.");
    if (varnorm) *varnorm = c.VN;
    std::vector<GroupLevelSpec> levels;
    for (size_t i = 0; i < c.opt.size(); ++i) {
        if (c.opt[i] != "DISCRETE") continue;
        GroupLevelSpec lv;
        lv.data_order = c.datagrid[i], lv.cp_order = c.CPgrid[i], lv.sigma_in = c.sigma_in[i];
        GroupLevelOptions &o = lv.options;
        o.sg_order = c.SGgrid[i], o.iters = c.it[i];
        o.cost.simmeasure = c.simval[i], o.cost.fixnan = c.fixnan, o.cost.lambda = c.lambda[i];
        o.cost.shearmodulus = c.shearmod, o.cost.bulkmodulus = c.bulkmod, o.cost.kexponent = c.k_exponent, o.cost.exponent = c.regexp;
        o.cost.range = c.cprange, o.cost.percentile = c.percentile;
        levels.push_back(lv);
    }
    return levels;
}

}  // namespace msmhip

#endif
