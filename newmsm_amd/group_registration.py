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
    def group(self, S, simmeasure, lambda_, fixnan):
        return _ProductGroup(api.DiscreteGroupCostFunction(self.ctx, S, simmeasure=simmeasure, lambda_=lambda_, fixnan=fixnan))


class _ProductGroup:
    def __init__(self, g):
        self.g = g

    def set_template(self, mesh):
        self.g.set_template(mesh, None)

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
                    labeldist=0.5, icm_passes=5, timings=None, fixnan=True):
    """feats: S x D x V data on the subjects' data grid (data_xyz, data_tri: the regular sphere the features live on); sph_regs: S x V x 3,
    every subject's registered sphere so far.  fixnan (--fixnan, M/DiscreteGroupCostFunction.cpp:50,96): the cost of two patches without a
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
    samples, _ = ops.label_sampling_grid(cp_order + 2, labeldist * mvdmax)  # m_labels = m_samples in every iteration, :176
    g = ops.group(S, simmeasure, lambda_, fixnan)
    template = ops.mesh(template_xyz, template_tri)
    g.set_template(template)
    g.initialize(cp_mesh0, cp_xyz0, cp_tri)
    meshes = [ops.mesh(data_xyz, data_tri) for _ in range(S)]
    for s in range(S):
        g.set_subject(s, meshes[s], feats[s])  # set_meshspace: the original data meshes
    sph_regs = [np.array(x, dtype=np.float64) for x in sph_regs]
    cps = [np.array(cp_xyz0) for _ in range(S)]
    prev = [np.array(cp_xyz0) for _ in range(S)]
    energies, labelings, energy = [], [], 0.0
    for it in range(iters):
        for s in range(S):
            ops.set_coords(meshes[s], sph_regs[s])
            g.set_subject(s, meshes[s], feats[s])  # reset_meshspace
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
