"""tests/time_registration_cpu.py -- the three-level registration of tools/time_registration.py driven over the CPU oracle
(OpenMP where the oracle has it) and over the MI355X path on the same inputs: the end-to-end counterpart of bench.py's
cpu_baseline.  A script, run by hand on a GPU box (it lives under tests/ because it uses the oracle):

    python tests/time_registration_cpu.py [iters mciters]  ->  one JSON line
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import registration, synthetic  # noqa: E402
from tests.helpers import OracleOps  # noqa: E402

iters, mciters = (int(a) for a in (sys.argv[1:3] + ["3", "50"][len(sys.argv) - 1:]))
xyz, tri = M.make_mesh_from_icosa(6)
ref = synthetic.features(xyz, 1, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 1, 7)
levels = [dict(data_order=4, cp_order=2, sigma_in=4.0, sigma_ref=4.0), dict(data_order=5, cp_order=3, sigma_in=2.0, sigma_ref=2.0),
          dict(data_order=6, cp_order=4, sigma_in=1.0, sigma_ref=1.0)]
kw = dict(varnorm=True, iters=iters, mciters=mciters, mcparam=0.8, seed=1, cost_params=dict(lambda_=0.1))
out = {}
results = {}
for name, ops in (("gpu", registration.ProductOps(M.Context(0))), ("cpu_port", OracleOps(M.mcmc_optimise))):
    reps = 2 if name == "gpu" else 1
    for rep in range(reps):
        clock = {}
        t0 = time.perf_counter()
        results[name] = registration.run_multiresolution(ops, xyz, tri, src, xyz, tri, ref, levels, timings=clock, **kw)
        wall = time.perf_counter() - t0
    out[name] = dict(wall_s=round(wall, 3), path_s=round(sum(v for k, v in clock.items() if k != "optimiser"), 3),
                     phases_s={k: round(v, 3) for k, v in sorted(clock.items())})
d = np.abs(results["gpu"][0] - results["cpu_port"][0]).max()
out["max_abs_coordinate_difference_mm"] = float(d)
out["path_speedup"] = round(out["cpu_port"]["path_s"] / out["gpu"]["path_s"], 1)
out["cpu_threads"] = int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count()
print(json.dumps(out))
