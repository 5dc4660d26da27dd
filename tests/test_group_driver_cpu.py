"""The caller logic of a --groupwise run (newmsm_amd/group_registration.py) against a statement-by-statement restatement of Group_Mesh_registration
(tests/group_driver_literal.py), both over the oracle's primitives: no GPU.  What it pins is the driver's OWN bookkeeping -- which mesh the model holds in
which iteration (iteration 0 of every level after the first runs on the level's ORIGINAL data grid, M/group_mesh_registration.cpp:54 and :114),
the order of warp / unfold / reset calls -- which the parity tests cannot see because they run one driver over two sets of ops."""
import numpy as np

from newmsm_amd import api, group_registration, synthetic
from oracle import oracle as O

from group_driver_literal import GroupMeshRegistrationLiteral
from helpers import OracleOps


def test_group_driver_equals_the_literal_restatement_over_two_levels():
    S, D = 2, 2
    xyz, tri = O.icosphere(3)
    meshes, datas = [], []
    for s in range(S):
        sph = synthetic.known_warp(xyz, seed=60 + s, rot_deg=1.5 + s, amp=0.8)  # the subjects' input spheres: irregular
        meshes.append((sph, tri))
        datas.append(synthetic.features(synthetic.known_warp(xyz, seed=80 + s, rot_deg=4.0 + 3 * s, amp=2.0), D, seed=7))
    txyz, ttri = O.icosphere(3)
    levels = [dict(data_order=3, cp_order=1, sg_order=3, sigma_in=2.0, iters=2, simmeasure=2, cost_params=dict(lambda_=0.01)),
              dict(data_order=4, cp_order=2, sg_order=4, sigma_in=1.0, iters=3, simmeasure=2, cost_params=dict(lambda_=0.01))]
    labs = []
    regs, level_regs, energies = group_registration.run_group_multiresolution(OracleOps(api.mcmc_optimise), meshes, datas, txyz, ttri, levels, varnorm=True, fixnan=True,
                                                                              labelings_out=labs)
    lit = GroupMeshRegistrationLiteral(meshes, datas, txyz, ttri, levels, varnorm=True, fixnan=True)
    want = lit.run_multiresolutions()
    assert len(labs) == len(lit.labelings) and len(labs) >= 4
    for k, (a, b) in enumerate(zip(labs, lit.labelings)):
        assert np.array_equal(a, b), "labeling of iteration %d differs" % k
    for a, b in zip(energies, lit.energies):
        assert np.allclose(a, b, rtol=1e-12)
    for s in range(S):
        assert np.array_equal(regs[s], want[s])  # the same primitives in the same order: bit for bit
    n1 = len(energies[0])
    assert np.any(labs[0] != 0) and np.any(labs[n1] != 0), "both levels should move control points in their first iteration"
    # the statement the ADVICE was about matters in this run: the projected spheres the second level starts from are NOT its original data grid
    assert max(np.abs(level_regs[0][s] - O.icosphere(3)[0]).max() for s in range(S)) > 1e-3
