"""One resolution level of a groupwise (gMSM) registration, as the reference's caller drives the hot path:

    Group_Mesh_registration::run_discrete_opt           M/group_mesh_registration.cpp:70-118
    DiscreteGroupModel::Initialize / setupCostFunction  M/DiscreteGroupModel.cpp:145-197
    DiscreteGroupModel::applyLabeling                   M/DiscreteGroupModel.h:74-78
    Fusion::optimize (the label loop)                   I/Fusion/Fusion.h:136-229

Per iteration: every subject's registered sphere and control grid go back into the model, setupCostFunction (pairs between subjects,
rotations, get_patch_data), two sweeps over the labels -- per label step ONE fusion move for all 4 P inter-subject pair costs and 8 T
triplet costs --, applyLabeling, then per subject: unfold the moved control grid, carry the data sphere through the move
(sphere_project_warp), unfold it.  The loop is caller logic; what it calls goes through `ops` (ProductGroupOps below over the C ABI; the parity
tests pass an oracle-backed object with the same methods).  The binary solve of a label step (ELC + FastPD: licence-restricted, FSL-bound)
is replaced by a stand-in, as in registration.run_discrete_level(optimiser="fusion"): the run exercises the path as HOCR drives it, its
result is not the reference's optimum."""
import numpy as np

from . import api
from .registration import ProductOps, apply_labeling


class ProductGroupOps(ProductOps):
    def __init__(self, ctx, host_rotations=True):
        """host_rotations (the library's default): the rotation matrices of the data meshes' vertices from the host's libm, as the reference computes them
        (DiscreteGroupCostFunction.set_rotation_mode) -- on regular icospheres (template and subjects' spheres alike, as gMSM's scripts set them up) a label
        carries data vertices exactly onto template vertices and the last bits of the rotation decide the resampled values there; False: the device computes them"""
        super().__init__(ctx)
        self.host_rotations = host_rotations

    def group(self, S, simmeasure, lambda_, fixnan, **params):
        """params: mu, kappa, k_exp, rexp, range_, percentile (--shearmod --bulkmod --k_exponent --regexp --cprange --percentile) where they differ from the
        reference's defaults"""
        g = api.DiscreteGroupCostFunction(self.ctx, S, simmeasure=simmeasure, lambda_=lambda_, fixnan=fixnan, **params)
        g.set_rotation_mode(g.HOST_ROTATIONS if self.host_rotations else g.DEVICE_ROTATIONS)
        return _ProductGroup(g)


class _ProductGroup:
    def __init__(self, g):
        self.g = g

    def set_template(self, mesh, mask=None):
        self.g.set_template(mesh, mask)

    def initialize(self, cp_mesh, cp_xyz, cp_tri):
        self.g.Initialize(cp_xyz, cp_tri)

    def set_subject(self, s, mesh, feat):
        self.g.reset_meshspace(s, mesh, feat)

    def reset_cpgrid(self, s, xyz):
        self.g.reset_CPgrid(s, xyz)

    def set_labels(self, labels):
        self.g.set_labels(labels)

    def setup(self):
        self.g.setupCostFunction()

    def pairs(self):
        return self.g.getPairs()

    def triplets(self):
        return self.g.getTriplets()

    def fusion_move(self, labeling, label):
        return self.g.fusionMove(labeling, label)


def run_group_level(ops, template_xyz, template_tri, data_xyz, data_tri, feats, sph_regs, cp_order, *, iters=2, simmeasure=2, lambda_=0.1,
                    labeldist=0.5, icm_passes=5, timings=None, fixnan=True, sg_order=None, cps_start=None, mask=None, cost_params=None):
    """feats: S x D x V data on the subjects' data grid (data_xyz, data_tri: the regular sphere the features live on); sph_regs: S x V x 3,
    every subject's registered sphere so far.  sg_order: the sampling grid's resolution (--SGgrid, m_SGres; two above the control grid when not
    given); cps_start (S x N x 3, optional): the control grids the level starts from (warp_CPgrid of the previous level's warp,
    M/DiscreteGroupModel.h:69-72; the regular grid when not given); mask (V(template), optional): --mask, the weights of the common template
    vertices in a pair cost (M/DiscreteGroupCostFunction.cpp:77); cost_params: the strain and similarity parameters of ops.group.  fixnan (--fixnan, M/DiscreteGroupCostFunction.cpp:50,96): the cost of two patches without a
    common template vertex is 1e7 instead of NaN -- with NaN costs the stand-in solve never moves a node of such a pair.  Returns (sph_regs, control grids S x N x 3, energies per iteration, labelings)."""
    import time

    clock = timings if timings is not None else {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        out = fn(*a)
        clock[name] = clock.get(name, 0.0) + time.perf_counter() - t0
        return out

    S = len(feats)
    cp_xyz0, cp_tri = ops.icosphere(cp_order)
    N = len(cp_xyz0)
    cp_mesh0 = ops.mesh(cp_xyz0, cp_tri)
    _, mvdmax = ops.cp_spacings(cp_mesh0, cp_xyz0, cp_tri)
    samples, _ = ops.label_sampling_grid(cp_order + 2 if sg_order is None else sg_order, labeldist * mvdmax)  # m_labels = m_samples in every iteration, :176
    g = ops.group(S, simmeasure, lambda_, fixnan, **(cost_params or {}))
    template = ops.mesh(template_xyz, template_tri)
    if mask is None:
        g.set_template(template)
    else:
        g.set_template(template, np.ascontiguousarray(mask, dtype=np.float64))
    g.initialize(cp_mesh0, cp_xyz0, cp_tri)
    meshes = [ops.mesh(data_xyz, data_tri) for _ in range(S)]
    for s in range(S):
        g.set_subject(s, meshes[s], feats[s])  # set_meshspace: the original data meshes
    sph_regs = [np.array(x, dtype=np.float64) for x in sph_regs]
    cps = [np.array(cp_xyz0 if cps_start is None else cps_start[s], dtype=np.float64) for s in range(S)]
    prev = [np.array(c) for c in cps]  # previous_controlgrids = model->get_CPgrid(subject), :75-78
    energies, labelings, energy = [], [], 0.0
    for it in range(iters):
        for s in range(S):
            # The model's data meshes at iteration 0 are the level's ORIGINAL data grid for every subject (set_meshspace(target_space, SPH_orig, S) in
            # initialize_level, M/group_mesh_registration.cpp:54) -- also at levels after the first, where project_CPgrid has carried the previous level's warp
            # to ALL_SPH_REG and to the model's control grids (warp_CPgrid) but NOT to m_datameshes; reset_meshspace(ALL_SPH_REG[subject]) only comes at the
            # end of an iteration (:114).  So the first get_patch_data of a later level rotates and resamples the unwarped data grid against warped control
            # grids: what newmsm does, reproduced here (until round 5 iteration 0 started from the projected spheres).
            if it > 0:
                ops.set_coords(meshes[s], sph_regs[s])
                g.set_subject(s, meshes[s], feats[s])  # reset_meshspace at the end of the previous iteration, :114
            g.reset_cpgrid(s, cps[s])
        g.set_labels(samples)
        timed("setup", g.setup)
        pairs, triplets = g.pairs(), g.triplets()
        labeling = np.zeros(S * N, dtype=np.int32)  # resetLabeling
        for sweep in range(2):
            for label in range(len(samples)):
                if not np.any(labeling != label):
                    continue
                quads, octets = timed("fusion_moves", g.fusion_move, labeling, label)
                x = timed("optimiser", ops.fusion_step, S * N, octets, triplets, icm_passes, quads, pairs)
                labeling = np.where((x == 1) & (labeling != label), label, labeling).astype(np.int32)
        quads, octets = timed("total_cost", g.fusion_move, labeling, 0)
        newenergy = float(np.sum(quads[:, 0]) + np.sum(octets[:, 0]))  # evaluateTotalCostSum: pairs, then triplets, at the labeling
        energies.append(newenergy)
        labelings.append(labeling)
        if it > 1 and energy - newenergy < newenergy * 0.01:  # :91-98
            break
        for s in range(S):  # applyLabeling + the per-subject tail of the loop, :104-115
            rot = ops.cp_rotations(samples[0], cps[s])
            moved = ops.mesh(apply_labeling(rot, samples, labeling[s * N:(s + 1) * N]), cp_tri)
            timed("unfold", ops.unfold, moved)
            new_cp = ops.coords(moved)
            sph = timed("sphere_project_warp", ops.sphere_project_warp, sph_regs[s], ops.mesh(prev[s], cp_tri), new_cp)
            ops.set_coords(meshes[s], sph)
            timed("unfold", ops.unfold, meshes[s])
            sph_regs[s] = ops.coords(meshes[s])
            prev[s], cps[s] = new_cp, new_cp
        energy = newenergy
    return np.stack(sph_regs), np.stack(cps), energies, labelings


def run_group_multiresolution(ops, meshes, datas, template_xyz, template_tri, levels, *, mask=None, varnorm=False, fixnan=False, timings=None,
                              labelings_out=None, **level_kw):
    """Group_Mesh_registration::run_multiresolutions (M/mesh_registration.cpp:30-50 over the overrides of M/group_mesh_registration.cpp) without
    file I/O:

    per level  initialize_level (:26-57): featurespace::initialise over all subjects (M/featurespace.cpp:39-86: every subject's data onto the
               level's icosphere by metric_resample, smooth_data with the level's --sigma_in, variance_normalise), the control grid, the
               model over (template, data grid, S);
               evaluate (:59-68): level 1 starts every subject on the data grid; later levels carry each subject's warp to the new data grid
               and control grid (project_CPgrid with the subject's index, M/mesh_registration.cpp:131-162), then run_discrete_opt (:70-118);
    at the end transform (:120-125): every subject's input sphere moved through its final warp ("sphere-<i>.reg").

    meshes: per subject (xyz, tri), spheres of radius 100; datas: per subject D x V(mesh); template_*: the sphere the patches are compared on
    (--template); levels: dicts as config.levels_from_config builds them (data_order, cp_order, sg_order, sigma_in, iters, simmeasure,
    cost_params["lambda_"]); mask: V(template) weights (--mask).  Returns (registered input spheres, per-level S x V x 3 registered data grids,
    per-level energies)."""
    import time

    clock = timings if timings is not None else {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        out = fn(*a)
        clock[name] = clock.get(name, 0.0) + time.perf_counter() - t0
        return out

    S = len(meshes)
    if len(datas) != S:
        raise ValueError("featurespace::Initialize do not have the same number of datasets and surface meshes")  # M/featurespace.cpp:43-44
    in_xyz = [np.asarray(m[0], dtype=np.float64) for m in meshes]
    in_mesh = [ops.mesh(in_xyz[s], meshes[s][1]) for s in range(S)]
    prev_regs, prev_order, regs, all_energies = None, None, [], []
    for lv in levels:
        ico_xyz, ico_tri = ops.icosphere(lv["data_order"])
        ico = ops.mesh(ico_xyz, ico_tri)
        feats = []
        for s in range(S):
            f = timed("metric_resample", ops.metric_resample, in_mesh[s], datas[s], ico)
            if lv.get("sigma_in", 0.0) > 0.0:
                f = timed("smooth_data", ops.smooth_data, ico, f, lv["sigma_in"])
            if varnorm:
                f = ops.variance_normalise(f)
            feats.append(f)
        cps_start = None
        if prev_regs is None:
            sph = [ico_xyz for _ in range(S)]  # ALL_SPH_REG.resize(num_subjects, SPH_orig), :60-61
        else:
            prev_xyz, prev_tri = ops.icosphere(prev_order)
            prev_ico = ops.mesh(prev_xyz, prev_tri)
            cp_xyz, cp_tri = ops.icosphere(lv["cp_order"])
            sph, cps_start = [], []
            for s in range(S):
                incurrent = timed("sphere_project_warp", ops.sphere_project_warp, in_xyz[s], prev_ico, prev_regs[s])
                moved = ops.mesh(timed("sphere_project_warp", ops.sphere_project_warp, ico_xyz, in_mesh[s], incurrent), ico_tri)
                cpm = ops.mesh(timed("sphere_project_warp", ops.sphere_project_warp, cp_xyz, in_mesh[s], incurrent), cp_tri)  # warp_CPgrid
                timed("unfold", ops.unfold, cpm)
                timed("unfold", ops.unfold, moved)
                cps_start.append(ops.coords(cpm))
                sph.append(ops.coords(moved))
        kw = dict(level_kw)
        kw.update({k: lv[k] for k in ("sg_order", "iters", "simmeasure") if k in lv})
        if "cost_params" in lv:
            kw["cost_params"] = {k: v for k, v in lv["cost_params"].items() if k != "lambda_"}
            if "lambda_" in lv["cost_params"]:
                kw["lambda_"] = lv["cost_params"]["lambda_"]
        out, _, energies, labelings = run_group_level(ops, template_xyz, template_tri, ico_xyz, ico_tri, np.stack(feats), sph, lv["cp_order"], cps_start=cps_start,
                                                      mask=mask, fixnan=fixnan, timings=clock, **kw)
        if labelings_out is not None:
            labelings_out.extend(labelings)
        regs.append(out)
        all_energies.append(energies)
        prev_regs, prev_order = out, lv["data_order"]
    last_xyz, last_tri = ops.icosphere(levels[-1]["data_order"])
    last = ops.mesh(last_xyz, last_tri)
    sphere_regs = [timed("sphere_project_warp", ops.sphere_project_warp, in_xyz[s], last, prev_regs[s]) for s in range(S)]
    return sphere_regs, regs, all_energies
