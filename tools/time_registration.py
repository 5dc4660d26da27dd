"""Wall-clock of a pairwise registration on the MI355X path (BASELINE.json: "wall-clock per ico6 pairwise reg"), driven by
the reference's caller loop (newmsm_amd/registration.py) with its Monte Carlo optimiser (M/mcmc_opt.h):

    python tools/time_registration.py [iters mciters]

Schedule: three DISCRETE levels as in config/basic_configs (data grids ico4/5/6, control grids ico2/3/4, sampling grids
ico4/5/6, sigma 4/2/1), input and reference spheres ico6 with one feature (sulc-like).  Prints the per-phase split of the
second run (the first pays for library load and allocations): the path against the optimiser."""
import json
import sys
import time

sys.path.insert(0, ".")
import newmsm_amd as M  # noqa: E402
from newmsm_amd import registration, synthetic  # noqa: E402

iters, mciters = (int(a) for a in (sys.argv[1:3] + ["3", "50"][len(sys.argv) - 1:]))
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(6)
ref = synthetic.features(xyz, 1, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 1, 7)
levels = [dict(data_order=4, cp_order=2, sigma_in=4.0, sigma_ref=4.0), dict(data_order=5, cp_order=3, sigma_in=2.0, sigma_ref=2.0),
          dict(data_order=6, cp_order=4, sigma_in=1.0, sigma_ref=1.0)]
ops = registration.ProductOps(ctx)
for rep in range(2):
    clock = {}
    t0 = time.perf_counter()
    reg, regs, energies = registration.run_multiresolution(ops, xyz, tri, src, xyz, tri, ref, levels, varnorm=True, timings=clock, iters=iters,
                                                           mciters=mciters, mcparam=0.8, seed=1, cost_params=dict(lambda_=0.1))
    wall = time.perf_counter() - t0
print(json.dumps(dict(levels=[(lv["data_order"], lv["cp_order"]) for lv in levels], iterations_per_level=iters, mciters=mciters,
                      wall_s=round(wall, 4), phases_s={k: round(v, 4) for k, v in sorted(clock.items())},
                      path_s=round(sum(v for k, v in clock.items() if k != "optimiser"), 4),
                      energies=[[round(e, 3) for e in lv] for lv in energies])))
