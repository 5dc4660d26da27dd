"""Group_Mesh_registration restated statement by statement over the ORACLE's primitives (test infrastructure: only tests import this).

newmsm_amd/group_registration.py (and its C++ twin include/msmhip_group_registration.hpp) are the caller logic of a --groupwise run re-organised around
an `ops` object; the parity tests run that SAME driver over the HIP path and over the oracle, so a mistake in the driver's own bookkeeping cancels out
(until round 5 iteration 0 of every level after the first ran on the projected spheres where newmsm runs it on the level's original data
grid).  This file keeps the reference's own state -- MESHES, SPH_orig, ALL_SPH_REG, the model's m_datameshes / m_controlmeshes -- and its own order of
statements, one method per reference function:

    run_multiresolutions   M/mesh_registration.cpp:30-50
    initialize_level       M/group_mesh_registration.cpp:26-57 (featurespace::initialise M/featurespace.cpp:39-86, DiscreteGroupModel::set_meshspace /
                           Initialize M/DiscreteGroupModel.h:53-58, M/DiscreteGroupModel.cpp:146-166)
    evaluate               M/group_mesh_registration.cpp:59-68
    project_CPgrid         M/mesh_registration.cpp:131-162 (level > 1 branch; warp_CPgrid M/DiscreteGroupModel.h:69-72)
    run_discrete_opt       M/group_mesh_registration.cpp:70-118 (setupCostFunction M/DiscreteGroupModel.cpp:168-199, applyLabeling M/DiscreteGroupModel.h:74-78,
                           the label loop of Fusion::optimize I/Fusion/Fusion.h:136-229 with the stand-in solve every end-to-end run here uses)
    transform              M/group_mesh_registration.cpp:120-125

recentre() of the regular spheres (a shift of ~1e-15) is not applied, as in the drivers under test."""
import numpy as np

from newmsm_amd import api
from newmsm_amd.registration import apply_labeling
from oracle import oracle as O

RAD = 100.0


class LiteralGroupModel:
    """DiscreteGroupModel's state as the driver sees it; the cost side (pairs, spacings, rotations, patch data, clique costs) is the oracle's group"""

    def __init__(self, S, simmeasure, lambda_, fixnan, labeldist, sg_order, cost_params):
        self.S, self.labeldist, self.sg_order = S, labeldist, sg_order
        self.og = O.Group(S, simmeasure=simmeasure, lambda_=lambda_, fixnan=fixnan, **cost_params)
        self.keep = []

    def set_featurespace(self, DATA):
        self.DATA = DATA

    def set_meshspace(self, target, source_xyz, source_tri, num):  # M/DiscreteGroupModel.h:53-58
        self.target_space = target
        self.tri = source_tri
        self.m_datameshes = [np.array(source_xyz) for _ in range(num)]  # m_datameshes.resize(num, source)

    def set_masks(self, mask):
        self.mask = mask

    def Initialize(self, control_xyz, control_tri):  # M/DiscreteGroupModel.cpp:146-166
        self.cp_tri = control_tri
        self.m_controlmeshes = [np.array(control_xyz) for _ in range(self.S)]
        self.N = len(control_xyz)
        control = O.Mesh(control_xyz, control_tri)
        _, mvdmax = O.cp_spacings(control)
        self.m_maxs_dist = self.labeldist * mvdmax
        _, self.m_samples, _ = O.label_sampling_grid(O.Mesh(*O.icosphere(self.sg_order)), self.m_maxs_dist)  # Initialize_sampling_grid
        self.og.set_template(self.target_space, getattr(self, "mask", None))
        self.og.set_controlgrid(control)
        self.keep.append(control)
        # costfct->set_meshes(m_datameshes, controlgrid, num): the original data meshes and control grid (the strain's reference configuration)
        for s in range(self.S):
            m = O.Mesh(self.m_datameshes[s], self.tri)
            self.og.set_subject(s, m, self.DATA[s])
            self.keep.append(m)

    def reset_meshspace(self, source_xyz, num):  # M/DiscreteGroupModel.h:60-62
        self.m_datameshes[num] = np.array(source_xyz)

    def reset_CPgrid(self, grid_xyz, num):
        self.m_controlmeshes[num] = np.array(grid_xyz)

    def get_CPgrid(self, num):
        return np.array(self.m_controlmeshes[num])

    def warp_CPgrid(self, start_mesh, end_xyz, num):  # M/DiscreteGroupModel.h:69-72
        cp = O.Mesh(O.sphere_project_warp(self.m_controlmeshes[num], start_mesh, end_xyz), self.cp_tri)
        O.unfold(cp, RAD)
        self.m_controlmeshes[num] = np.array(cp.xyz)

    def setupCostFunction(self):  # M/DiscreteGroupModel.cpp:168-199
        self.labeling = np.zeros(self.S * self.N, dtype=np.int32)  # resetLabeling
        for s in range(self.S):
            m = O.Mesh(self.m_datameshes[s], self.tri)
            self.og.set_subject(s, m, self.DATA[s])  # get_patch_data reads m_datameshes[subject]
            self.keep.append(m)
            self.og.reset_cpgrid(s, self.m_controlmeshes[s])
        self.m_labels = self.m_samples
        self.og.set_labels(self.m_labels)
        self.og.setup()  # estimate_pairs, get_spacings, get_rotations, get_patch_data

    def applyLabeling(self):  # M/DiscreteGroupModel.h:74-78
        for s in range(self.S):
            rot = O.cp_rotations(self.m_labels[0], self.m_controlmeshes[s])  # m_ROT of this iteration's set-up
            self.m_controlmeshes[s] = apply_labeling(rot, self.m_labels, self.labeling[s * self.N:(s + 1) * self.N])

    # the evaluators Fusion::optimize calls
    def pair_quads(self, label):
        pr = self.og.pairs()
        lab = self.labeling
        return pr, np.array([[self.og.pairwise(p, int(lab[a]), int(lab[b])), self.og.pairwise(p, int(lab[a]), label), self.og.pairwise(p, label, int(lab[b])),
                              self.og.pairwise(p, label, label)] for p, (a, b) in enumerate(pr)]).reshape(len(pr), 4)

    def triplet_octets(self, label):
        tr = self.og.triplets()
        out = np.empty((len(tr), 8))
        for t, nodes in enumerate(tr):
            cur = [int(self.labeling[v]) for v in nodes]
            for k in range(8):  # 000 .. 111, I/Fusion/Fusion.h:188-195
                out[t, k] = self.og.triplet(t, *[label if k >> (2 - j) & 1 else cur[j] for j in range(3)])
        return tr, out


class GroupMeshRegistrationLiteral:
    def __init__(self, meshes, datas, template_xyz, template_tri, levels, mask=None, varnorm=False, fixnan=False, labeldist=0.5, icm_passes=5):
        self.MESHES = [(np.asarray(x, dtype=np.float64), t) for x, t in meshes]
        self.DATAlist = datas
        self.target_space = O.Mesh(template_xyz, template_tri)
        self.levels, self.mask, self._varnorm, self.fixnan, self.labeldist, self.icm_passes = levels, mask, varnorm, fixnan, labeldist, icm_passes
        self.num_subjects = len(meshes)
        self.labelings, self.energies = [], []

    def run_multiresolutions(self):  # M/mesh_registration.cpp:30-50
        for i in range(len(self.levels)):
            self.level = i + 1
            self.initialize_level(i)
            self.evaluate()
        return self.transform()

    def initialize_level(self, current_lvl):  # M/group_mesh_registration.cpp:26-57
        lv = self.levels[current_lvl]
        self.lv = lv
        ico_xyz, ico_tri = O.icosphere(lv["data_order"])  # featurespace::initialise, M/featurespace.cpp:39-86
        icotmp = O.Mesh(ico_xyz, ico_tri)
        DATA = []
        for i in range(self.num_subjects):
            tmp = O.metric_resample(O.Mesh(*self.MESHES[i]), self.DATAlist[i], icotmp)
            if lv.get("sigma_in", 0.0) > 0.0:
                tmp = O.smooth_data(icotmp, tmp, icotmp, lv["sigma_in"])
            DATA.append(tmp)
        if self._varnorm:
            DATA = [O.variance_normalise(d) for d in DATA]
        self.SPH_orig, self.SPH_tri = ico_xyz, ico_tri
        control_xyz, control_tri = O.icosphere(lv["cp_order"])
        cp = dict(lv.get("cost_params", {}))
        lam = cp.pop("lambda_", 0.1)
        self.model = LiteralGroupModel(self.num_subjects, lv.get("simmeasure", 2), lam, self.fixnan, self.labeldist, lv.get("sg_order", lv["cp_order"] + 2), cp)
        self.model.set_featurespace(DATA)
        self.model.set_meshspace(self.target_space, self.SPH_orig, self.SPH_tri, self.num_subjects)
        if self.mask is not None:
            self.model.set_masks(self.mask)
        self.model.Initialize(control_xyz, control_tri)

    def evaluate(self):  # M/group_mesh_registration.cpp:59-68
        if self.level == 1:
            self.ALL_SPH_REG = [np.array(self.SPH_orig) for _ in range(self.num_subjects)]
        else:
            for subject in range(self.num_subjects):
                self.ALL_SPH_REG[subject] = self.project_CPgrid(self.SPH_orig, self.ALL_SPH_REG[subject], subject)
        self.run_discrete_opt()

    def project_CPgrid(self, SPH_in, REG, num):  # M/mesh_registration.cpp:131-162, level > 1
        ico_xyz, ico_tri = O.icosphere(self.prev_order)  # make_mesh_from_icosa(REG.get_resolution())
        icotmp = O.Mesh(ico_xyz, ico_tri)
        inorig = O.Mesh(*self.MESHES[num])
        incurrent = O.sphere_project_warp(self.MESHES[num][0], icotmp, REG)
        moved = O.Mesh(O.sphere_project_warp(SPH_in, inorig, incurrent), self.SPH_tri)
        self.model.warp_CPgrid(inorig, incurrent, num)
        O.unfold(moved, RAD)
        return np.array(moved.xyz)

    def run_discrete_opt(self):  # M/group_mesh_registration.cpp:70-118
        model, S = self.model, self.num_subjects
        energy = newenergy = 0.0
        previous_controlgrids = [model.get_CPgrid(s) for s in range(S)]
        level_energies = []
        for it in range(self.lv.get("iters", 2)):
            model.setupCostFunction()
            newenergy = self.fusion_optimize(model)
            level_energies.append(newenergy)
            self.labelings.append(np.array(model.labeling))
            if it > 1 and energy - newenergy < newenergy * 0.01:
                break
            model.applyLabeling()
            for subject in range(S):
                transformed_controlgrid = O.Mesh(model.get_CPgrid(subject), model.cp_tri)
                O.unfold(transformed_controlgrid, RAD)
                new_cp = np.array(transformed_controlgrid.xyz)
                sph = O.Mesh(O.sphere_project_warp(self.ALL_SPH_REG[subject], O.Mesh(previous_controlgrids[subject], model.cp_tri), new_cp), self.SPH_tri)
                O.unfold(sph, RAD)
                self.ALL_SPH_REG[subject] = np.array(sph.xyz)
                previous_controlgrids[subject] = new_cp
                model.reset_CPgrid(new_cp, subject)
                model.reset_meshspace(self.ALL_SPH_REG[subject], subject)
            energy = newenergy
        self.energies.append(level_energies)
        self.prev_order = self.lv["data_order"]

    def fusion_optimize(self, model):  # I/Fusion/Fusion.h:122-244 (the binary solve of a step: the stand-in every end-to-end run here uses)
        for _sweep in range(2):
            for label in range(len(model.m_labels)):
                if not np.any(model.labeling != label):  # sumlabeldiff == 0
                    continue
                pairs, quads = model.pair_quads(label)
                triplets, octets = model.triplet_octets(label)
                x = api.fusion_icm_step(model.S * model.N, octets, triplets, self.icm_passes, quads=quads, pairs=pairs)
                model.labeling = np.where((x == 1) & (model.labeling != label), label, model.labeling).astype(np.int32)
        _, quads = model.pair_quads(0)  # evaluateTotalCostSum: pairs, then triplets, at the labeling
        _, octets = model.triplet_octets(0)
        return float(np.sum(quads[:, 0]) + np.sum(octets[:, 0]))

    def transform(self):  # M/group_mesh_registration.cpp:120-125
        last = O.Mesh(self.SPH_orig, self.SPH_tri)
        return [O.sphere_project_warp(self.MESHES[s][0], last, self.ALL_SPH_REG[s]) for s in range(self.num_subjects)]
