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

#include <chrono>
#include <map>
#include <memory>
#include <string>

#include "msmhip.hpp"

namespace msmhip {

// wall-clock seconds and call counts per phase of the loops below, under the names newmsm_amd/registration.py uses ("get_source_data",
// "unary_table", "fusion_moves", "optimiser", "total_cost", "sphere_project_warp", "unfold", "metric_resample", "smooth_data", ...)
struct PhaseClock {
    std::map<std::string, double> seconds;
    std::map<std::string, long> calls;
    template <class Fn>
    static auto timed(PhaseClock *c, const char *name, Fn &&fn) {
        if (!c) return fn();
        const auto t0 = std::chrono::steady_clock::now();
        struct Stop {
            PhaseClock *c;
            const char *name;
            std::chrono::steady_clock::time_point t0;
            ~Stop() {
                c->seconds[name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                ++c->calls[name];
            }
        } stop{c, name, t0};
        return fn();
    }
};

struct LevelOptions {
    int sg_order = -1;  // sampling grid resolution; cp_order + 2 when negative
    int iters = 3;      // --it
    int mciters = 200;  // --mciters
    double mcparam = 0.8;
    uint64_t seed = 0;  // iteration i draws from std::mt19937(seed + i); the reference seeds from std::random_device
    double labeldist = 0.5;
    bool rescale_labels = false;
    Parameters cost;    // kind, similarity measure, regulariser
    // false: MCMC::optimise over the unary and T x L^3 triplet tables.  true: the label loop of Fusion::optimize (I/Fusion/Fusion.h:136-229),
    // per label step ONE msm_cost_triplet_octets for the 8 T combinations -- how --dopt=HOCR drives the path -- with a stand-in for the
    // licence-restricted binary solve (msm_fusion_icm_step, icm_passes passes): the path is exercised as HOCR would, the result is not HOCR's
    bool fusion = false;
    int icm_passes = 5;
    // true: --regoption=1 as --dopt=FastPD drives it (M/mesh_registration.cpp:182-188): the model lists pairs instead of triplets, per iteration
    // computeUnaryCosts + computePairwiseCosts, then a stand-in for FPD::FastPD(model, 100) (msm_pairwise_icm)
    bool pairwise = false;
    int anat_order = -1;  // --anatgrid of this level (regularisermode 4 / 5, aMSM); cp_order + 2 when negative
    // the fusion loop queues the next label step's evaluations while the host solves the current one (msm_cost_triplet_octets_prefetch: a hint, the
    // results do not depend on it)
    bool speculate = true;
    // measurement only: when set, the cost function records HIP events around its kernels (msm_cost_enable_timing) and the duration of every fusion
    // move's kernel (ms) is appended here -- one event query per move, so a run with this set is not the one whose wall clock is reported
    std::vector<double> *move_kernel_ms = nullptr;
};

// the anatomical surfaces of a --regoption=5 (aMSM) run: V x 3 on the vertices of the input / reference SPHERE (MESHES[0] / MESHES[1]), whose handles
// the level loop searches (set_anatomical, M/mesh_registration.cpp:437-441)
struct Anatomy {
    Mesh *in_sphere = nullptr, *ref_sphere = nullptr;
    const Points *in_anat = nullptr, *ref_anat = nullptr;
};

// --inweight / --refweight brought to a level's data grid (downsample_cfweighting, M/mesh_registration.cpp:334-350: nearest neighbour): rows x V(data grid)
struct Weighting {
    const Matrix *in_weight = nullptr, *ref_weight = nullptr;
    int in_rows = 0, ref_rows = 0;
};

// Mesh_registration::combine_costfunction_weighting, M/mesh_registration.cpp:849-869: the mean of the two weightings over the rows both have; the rows
// only the larger one has are kept.  a: ra x V, b: rb x V.
inline Matrix combine_costfunction_weighting(const Matrix &a, int ra, const Matrix &b, int rb, int *rows) {
    Matrix out = ra >= rb ? a : b;
    const size_t V = ra > 0 ? a.size() / (size_t)ra : 0, n = (size_t)std::min(ra, rb) * V;
    for (size_t i = 0; i < n; ++i) out[i] = (a[i] + b[i]) / 2.0;
    *rows = std::max(ra, rb);
    return out;
}

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
                                    int cp_order, const LevelOptions &o, const Points *cp_start = nullptr, PhaseClock *clock = nullptr,
                                    const Anatomy *anat = nullptr, const Weighting *weights = nullptr) {
    // ---- initialize_level / Initialize(CONTROL)
    auto [cp_xyz, cp_tri] = make_mesh_from_icosa(cp_order);
    Mesh TARGET(ctx, target_xyz, target_tri), SOURCE(ctx, source_xyz, source_tri), CPGRID(ctx, cp_xyz, cp_tri);
    TARGET.set_pvalues(ref_feat);
    auto [MAXSEP, MVDmax] = cp_spacings(cp_xyz, cp_tri);
    auto [samples, barycentres] = label_sampling_grid(o.sg_order < 0 ? cp_order + 2 : o.sg_order, o.labeldist * MVDmax);
    const double centre[3] = {samples[0], samples[1], samples[2]};
    if (o.pairwise && o.cost.regularisermode != 1) throw Error(MSM_ERR_INVALID, "MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ");
    const std::vector<int32_t> triplets = o.pairwise ? std::vector<int32_t>() : estimate_triplets(cp_tri);
    const std::vector<int32_t> pairs = o.pairwise ? estimate_pairs(cp_tri, (int)(cp_xyz.size() / 3)) : std::vector<int32_t>();
    DiscreteCostFunction costfct(ctx, o.cost);
    costfct.set_meshes(TARGET, SOURCE, CPGRID);  // _ORIG, _oCPgrid
    costfct.set_featurespace(src_feat, D);
    costfct.set_spacings(MAXSEP, MVDmax);
    std::unique_ptr<Mesh> anat_sphere;
    if (o.cost.regularisermode == 4 || o.cost.regularisermode == 5) {  // initialize_level, M/mesh_registration.cpp:91-104
        if (o.cost.regularisermode == 4)
            throw Error(MSM_ERR_INVALID, "--regoption 4 has been removed from newMSM. Use --regoption 3 for spherical mesh regularisation or --regoption 5 for "
                                         "anatomical mesh regularisation.");
        if (!anat || !anat->in_sphere || !anat->ref_sphere || !anat->in_anat || !anat->ref_anat)
            throw Error(MSM_ERR_INVALID, "--regoption 5 requires anatomical meshes. Use --regoption 3 for spherical mesh regularisation or provide anatomical meshes.");
        if (o.pairwise) throw Error(MSM_ERR_INVALID, "MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ");
        // resample_anatomy (:250-332): ANAT_ico with NEARESTFACES and _ANATbaryweights, both anatomies resampled onto it
        const AnatomyGrid grid = resample_anatomy_grid(cp_xyz, cp_tri, std::max(0, (o.anat_order < 0 ? cp_order + 2 : o.anat_order) - cp_order));
        const Points anat_orig = PhaseClock::timed(clock, "surface_resample", [&] { return barycentric_coords_resample(*anat->in_sphere, *anat->in_anat, grid.sphere_xyz); });
        const Points anat_target = PhaseClock::timed(clock, "surface_resample", [&] { return barycentric_coords_resample(*anat->ref_sphere, *anat->ref_anat, grid.sphere_xyz); });
        anat_sphere.reset(new Mesh(ctx, grid.sphere_xyz, grid.sphere_tri));
        costfct.setTriplets(triplets);  // NEARESTFACES is indexed by triplet = control triangle
        costfct.set_anatomical(*anat_sphere, anat_target, anat_orig, grid.sphere_tri, grid.weights, grid.face_ptr, grid.face_idx);
    }
    const int N = (int)(cp_xyz.size() / 3);
    if (o.move_kernel_ms) check(msm_cost_enable_timing(costfct.handle(), 1));
    int m_iter = 1;
    double m_scale = 1.0;
    if (cp_start) cp_xyz = *cp_start;
    LevelResult res;
    detail::HostBuffer octets[2];  // used in turn: a step's costs are read by its solve while the next step may already be written into the other
    for (int it = 0; it < o.iters; ++it) {
        // ---- reset_meshspace + setupCostFunction
        SOURCE.set_coords(sph_reg);
        if (weights && weights->in_weight && weights->ref_weight) {  // setupCostFunctionWeighting(combine_weighting()), M/mesh_registration.cpp:171,334-350
            const Matrix resampled = PhaseClock::timed(clock, "metric_resample", [&] { return metric_resample(TARGET, *weights->ref_weight, SOURCE); });
            int rows = 0;
            const Matrix W = combine_costfunction_weighting(*weights->in_weight, weights->in_rows, resampled, weights->ref_rows, &rows);
            costfct.set_dataaffintyweighting(W, rows);
        }
        costfct.reset_source(SOURCE);
        CPGRID.set_coords(cp_xyz);
        costfct.reset_CPgrid(CPGRID);
        const std::vector<double> ROT = cp_rotations(centre, cp_xyz);
        Points labels;
        if (o.rescale_labels) labels = rescale_sampling_grid(samples, m_scale);
        else labels = (m_iter % 2 == 0) ? samples : barycentres;
        costfct.set_labels(labels, ROT);
        PhaseClock::timed(clock, "get_source_data", [&] { costfct.get_source_data(); });
        if (o.pairwise) costfct.setPairs(pairs);
        else costfct.setTriplets(triplets);
        ++m_iter;
        PhaseClock::timed(clock, "unary_table", [&] { costfct.computeUnaryCosts(); });
        std::vector<int32_t> labeling((size_t)N, 0);  // resetLabeling
        const int L = (int)(labels.size() / 3), T = (int)(triplets.size() / 3);
        if (o.fusion) {  // ---- Fusion::optimize: two sweeps over the labels, a fusion move per label step
            double *Ebuf[2] = {octets[0].ensure(ctx.handle(), 8 * (size_t)T), octets[1].ensure(ctx.handle(), 8 * (size_t)T)};  // pinned, GPU-mapped
            int turn = 0;
            std::vector<double> unary2(2 * (size_t)N);
            const std::vector<int32_t> no_pairs;
            auto differs = [&](int label) {
                for (int i = 0; i < N; ++i)
                    if (labeling[(size_t)i] != label) return true;
                return false;
            };
            for (int step = 0; step < 2 * L; ++step) {
                const int label = step % L;
                if (!differs(label)) continue;
                double *E = Ebuf[turn];
                turn ^= 1;
                PhaseClock::timed(clock, "fusion_moves", [&] { check(msm_cost_triplet_octets(costfct.handle(), labeling.data(), label, E)); });
                if (o.move_kernel_ms) {
                    double ms = 0.0;
                    int32_t got = 0;
                    check(msm_cost_kernel_times(costfct.handle(), &ms, 1, &got));
                    if (got == 1) o.move_kernel_ms->push_back(ms);
                }
                if (o.speculate && !o.move_kernel_ms) {  // the next step that will be evaluated if this solve changes nothing: queued while the host solves
                    for (int nxt = step + 1; nxt < 2 * L; ++nxt)
                        if (differs(nxt % L)) {
                            PhaseClock::timed(clock, "fusion_prefetch", [&] { check(msm_cost_triplet_octets_prefetch(costfct.handle(), labeling.data(), nxt % L, Ebuf[turn])); });
                            break;
                        }
                }
                for (int i = 0; i < N; ++i) {
                    unary2[2 * (size_t)i] = costfct.unarycosts[(size_t)labeling[(size_t)i] * N + i];
                    unary2[2 * (size_t)i + 1] = costfct.unarycosts[(size_t)label * N + i];
                }
                const std::vector<int32_t> x =
                    PhaseClock::timed(clock, "optimiser", [&] { return fusion_icm_step(N, unary2, nullptr, no_pairs, E, triplets, o.icm_passes); });
                for (int i = 0; i < N; ++i)
                    if (x[(size_t)i] == 1 && labeling[(size_t)i] != label) labeling[(size_t)i] = label;
            }
        } else if (o.pairwise) {  // ---- FastPD: computeUnaryCosts, computePairwiseCosts, the solve
            PhaseClock::timed(clock, "pairwise_table", [&] { costfct.computePairwiseCosts(); });
            PhaseClock::timed(clock, "optimiser", [&] { pairwise_icm(costfct.unarycosts, costfct.paircosts, pairs, N, L, labeling, 100); });
        } else {  // ---- MCMC: computeUnaryCosts, computeTripletCosts, optimise
            const std::vector<double> tcosts = PhaseClock::timed(clock, "triplet_table", [&] { return costfct.computeTripletCosts(); });
            PhaseClock::timed(clock, "optimiser", [&] { mcmc_optimise(costfct.unarycosts, tcosts, triplets, N, L, o.mcparam, o.mciters, o.seed + (uint64_t)it, labeling); });
        }
        res.energies.push_back(PhaseClock::timed(clock, "total_cost", [&] { return costfct.evaluateTotalCostSum(labeling); }));
        res.labelings.push_back(labeling);
        // ---- applyLabeling, warp the source through the control grid's move, unfold both (:219-230)
        const Points moved = apply_labeling(ROT, labels, labeling);
        // SOURCE holds sph_reg since the top of the iteration; CPGRID still holds the previous grid
        PhaseClock::timed(clock, "sphere_project_warp", [&] { sphere_project_warp(SOURCE, CPGRID, moved); });
        CPGRID.set_coords(moved);
        PhaseClock::timed(clock, "unfold", [&] { return unfold(CPGRID); });
        cp_xyz = CPGRID.get_coords();
        PhaseClock::timed(clock, "unfold", [&] { return unfold(SOURCE); });
        sph_reg = SOURCE.get_coords();
    }
    res.sph_reg = sph_reg;
    res.cpgrid = cp_xyz;
    return res;
}

// one resolution level of a schedule (msmhip_config.hpp: levels_from_config fills these from a configuration file)
struct LevelSpec {
    int data_order = 5, cp_order = 2;
    double sigma_in = 0.0, sigma_ref = 0.0;
    LevelOptions options;
};

struct MultiresResult {
    Points sphere_reg;                            // the input sphere moved through the final warp ("sphere.reg", M/mesh_registration.cpp:352-356)
    std::vector<Points> level_reg;                // the registered data grid of every level
    std::vector<std::vector<double>> energies;    // per level, per iteration
    std::vector<std::vector<int32_t>> labelings;  // every iteration's labeling, level after level
};

// Mesh_registration::run_multiresolutions (M/mesh_registration.cpp:30-50) for DISCRETE levels without file I/O -- the C++ twin of
// newmsm_amd/registration.py: run_multiresolution (same calls in the same order; tests/test_cpp_host.py compares the two):
//   per level  featurespace::initialise (M/featurespace.cpp:39-86: metric_resample of both data sets onto the level's icosphere, smooth_data,
//              variance_normalise), project_CPgrid (M/mesh_registration.cpp:131-162: the warp of the previous level carried to the new data
//              grid and control grid, unfold) and run_discrete_opt;
//   at the end transform (:352-356).
// in_* / ref_*: the input and reference spheres (radius 100) with their D x V data.
inline MultiresResult run_multiresolutions(Context &ctx, const Points &in_xyz, const Triangles &in_tri, const Matrix &in_data, const Points &ref_xyz,
                                           const Triangles &ref_tri, const Matrix &ref_data, int D, const std::vector<LevelSpec> &levels, bool varnorm,
                                           PhaseClock *clock = nullptr, const Points *in_anat = nullptr, const Points *ref_anat = nullptr,
                                           const Matrix *in_cfweight = nullptr, int in_cfrows = 0, const Matrix *ref_cfweight = nullptr, int ref_cfrows = 0) {
    if (levels.empty()) throw Error(MSM_ERR_INVALID, "run_multiresolutions: no DISCRETE level");
    if ((in_anat != nullptr) != (ref_anat != nullptr)) throw Error(MSM_ERR_INVALID, "Error: must supply both anatomical meshes or none");  // CLI/newmsm.cpp:41-43
    if (in_anat && (in_anat->size() != in_xyz.size() || ref_anat->size() != ref_xyz.size()))
        throw Error(MSM_ERR_INVALID, "MeshREG ERROR:: input/reference anatomical mesh resolution is inconsistent with input/reference spherical mesh resolution.");
    Mesh in_mesh(ctx, in_xyz, in_tri), ref_mesh(ctx, ref_xyz, ref_tri);
    Anatomy anatomy;
    if (in_anat) anatomy.in_sphere = &in_mesh, anatomy.ref_sphere = &ref_mesh, anatomy.in_anat = in_anat, anatomy.ref_anat = ref_anat;
    MultiresResult res;
    Points sph_reg_prev;
    int prev_order = -1;
    for (const LevelSpec &lv : levels) {
        auto [ico_xyz, ico_tri] = make_mesh_from_icosa(lv.data_order);
        Mesh ico(ctx, ico_xyz, ico_tri);
        Matrix feats[2];
        for (int k = 0; k < 2; ++k) {
            Mesh &mesh = k == 0 ? in_mesh : ref_mesh;
            const Matrix &data = k == 0 ? in_data : ref_data;
            const double sigma = k == 0 ? lv.sigma_in : lv.sigma_ref;
            Matrix f = PhaseClock::timed(clock, "metric_resample", [&] { return metric_resample(mesh, data, ico); });
            if (sigma > 0.0) f = PhaseClock::timed(clock, "smooth_data", [&] { return smooth_data(ico, f, ico, sigma); });
            if (varnorm) variance_normalise(f, ico.nvertices());
            feats[k] = std::move(f);
        }
        Points sph_in, cp_start;
        bool have_cp_start = false;
        if (sph_reg_prev.empty()) {
            sph_in = ico_xyz;  // level 1, no transformed mesh: project_CPgrid only unfolds the (regular) data grid
        } else {
            auto [prev_xyz, prev_tri] = make_mesh_from_icosa(prev_order);
            Mesh prev(ctx, prev_xyz, prev_tri);
            const Points incurrent = PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(in_xyz, prev, sph_reg_prev); });
            sph_in = PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(ico_xyz, in_mesh, incurrent); });
            auto [cp_xyz, cp_tri] = make_mesh_from_icosa(lv.cp_order);
            Mesh cpm(ctx, PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(cp_xyz, in_mesh, incurrent); }), cp_tri);  // warp_CPgrid
            PhaseClock::timed(clock, "unfold", [&] { return unfold(cpm); });
            cp_start = cpm.get_coords();
            have_cp_start = true;
        }
        {
            Mesh moved(ctx, sph_in, ico_tri);
            PhaseClock::timed(clock, "unfold", [&] { return unfold(moved); });
            sph_in = moved.get_coords();
        }
        Matrix w_in, w_ref;
        Weighting weights;
        if (in_cfweight && ref_cfweight) {  // downsample_cfweighting: both weightings onto the level's data grid by nearest neighbour
            w_in = nearest_neighbour_interpolation(in_mesh, *in_cfweight, ico_xyz);
            w_ref = nearest_neighbour_interpolation(ref_mesh, *ref_cfweight, ico_xyz);
            weights.in_weight = &w_in, weights.ref_weight = &w_ref, weights.in_rows = in_cfrows, weights.ref_rows = ref_cfrows;
        }
        LevelResult r = run_discrete_opt(ctx, ico_xyz, ico_tri, feats[1], ico_xyz, ico_tri, feats[0], D, sph_in, lv.cp_order, lv.options,
                                         have_cp_start ? &cp_start : nullptr, clock, in_anat ? &anatomy : nullptr, in_cfweight && ref_cfweight ? &weights : nullptr);
        res.labelings.insert(res.labelings.end(), r.labelings.begin(), r.labelings.end());
        res.energies.push_back(r.energies);
        res.level_reg.push_back(r.sph_reg);
        sph_reg_prev = std::move(r.sph_reg);
        prev_order = lv.data_order;
    }
    auto [last_xyz, last_tri] = make_mesh_from_icosa(levels.back().data_order);
    Mesh last(ctx, last_xyz, last_tri);
    res.sphere_reg = PhaseClock::timed(clock, "sphere_project_warp", [&] { return sphere_project_warp(in_xyz, last, sph_reg_prev); });
    return res;
}

}  // namespace msmhip

#endif
