"""Wall-clock of one resolution level of a pairwise registration on the MI355X path (BASELINE.json: "wall-clock per ico6
pairwise reg"), with the Monte Carlo optimiser of the reference (M/mcmc_opt.h) as the caller:

    python tools/time_registration.py [data_order cp_order iters mciters]

Prints the per-phase split: the path (get_source_data, unary table, triplet table, warp, unfold) against the optimiser."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import newmsm_amd as M  # noqa: E402
from newmsm_amd import registration, synthetic  # noqa: E402

data_order, cp_order, iters, mciters = (int(a) for a in (sys.argv[1:5] + ["6", "4", "3", "50"][len(sys.argv) - 1:]))
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(data_order)
ref = synthetic.features(xyz, 1, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 1, 7)
ops = registration.ProductOps(ctx)
for rep in range(2):  # the first pass pays for library load and allocations
    clock = {}
    t0 = time.perf_counter()
    reg, cp, energies, labelings = registration.run_discrete_level(ops, xyz, tri, ref, xyz, tri, src, xyz, cp_order, iters=iters, mciters=mciters,
                                                                   mcparam=0.3, seed=1, timings=clock, cost_params=dict(lambda_=0.01))
    wall = time.perf_counter() - t0
L = len(M.label_sampling_grid(cp_order + 2, 0.5 * M.cp_spacings(*M.make_mesh_from_icosa(cp_order))[1])[0])
print(json.dumps(dict(data_order=data_order, cp_order=cp_order, iterations=iters, mciters=mciters, labels_even_iterations=L,
                      wall_s=round(wall, 4), phases_s={k: round(v, 4) for k, v in clock.items()},
                      path_s=round(sum(v for k, v in clock.items() if k != "optimiser"), 4), energies=[round(e, 4) for e in energies])))
