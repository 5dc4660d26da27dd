"""Wall-clock of an HCP MSMAll-shaped pairwise registration (BASELINE config 3; north star: ">= 50x the reference multicore-CPU wall-clock on
a full ico6 HCP MSMAll pairwise registration at 1 GPU") on the MI355X path and, with "cpu", on the CPU port (oracle/, OpenMP) -- the same caller
loop, the same schedule (newmsm_amd.registration.hcp_msmall_levels: config/HCP_multimodal_alignment/MSMAllStrainFinalconf1to1_1to3_2), the same
stand-in for the licence-restricted binary solve, synthetic ico6 spheres with 32 features:

    python tools/time_registration_msmall.py [gpu|cpu|both] [it1 it2 it3]

Prints one JSON line per leg and, for "both", the ratios and the largest angle between the two registered spheres."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import registration, synthetic  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "gpu"
its = tuple(int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (10, 15, 15)
xyz, tri = M.make_mesh_from_icosa(6)
ref = synthetic.features(xyz, 32, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 32, 7)
out = {}


def run(ops, reps):
    for _ in range(reps):
        clock = {}
        t0 = time.perf_counter()
        reg, _, energies = registration.run_multiresolution(ops, xyz, tri, src, xyz, tri, ref, registration.hcp_msmall_levels(its), varnorm=True, timings=clock)
        wall = time.perf_counter() - t0
    path = sum(v for k, v in clock.items() if k != "optimiser")
    return reg, dict(wall_s=round(wall, 4), path_s=round(path, 4), phases_s={k: round(v, 4) for k, v in sorted(clock.items())}, iterations=its,
                     final_energy=[round(e[-1], 6) for e in energies])


if which in ("gpu", "both"):
    reg_g, out["gpu"] = run(registration.ProductOps(M.Context(0)), 2)
    print(json.dumps({"leg": "MI355X path", **out["gpu"]}), flush=True)
if which in ("cpu", "both"):
    from newmsm_amd import dist as D

    os.environ.setdefault("MSM_ORACLE_THREADS", str(D.host_cores()))
    from tests import helpers  # the checker, here as the CPU baseline

    reg_c, out["cpu"] = run(helpers.OracleOps(M.mcmc_optimise), 1)
    out["cpu"]["threads"] = helpers.ORACLE_THREADS
    print(json.dumps({"leg": "CPU port (oracle/, OpenMP)", **out["cpu"]}), flush=True)
if which == "both":
    ua, ub = reg_g / np.linalg.norm(reg_g, axis=1, keepdims=True), reg_c / np.linalg.norm(reg_c, axis=1, keepdims=True)
    ang = 2.0 * np.arcsin(np.minimum(1.0, 0.5 * np.linalg.norm(ua - ub, axis=1)))
    print(json.dumps({"wall_ratio": round(out["cpu"]["wall_s"] / out["gpu"]["wall_s"], 1), "path_ratio": round(out["cpu"]["path_s"] / out["gpu"]["path_s"], 1),
                      "max_angle_rad": float(ang.max())}))
