"""tests/fuzz_group_multires.py [seed] [count] -- random multi-level groupwise registrations (Group_Mesh_registration::run_multiresolutions,
M/group_mesh_registration.cpp:26-133) through newmsm_amd/group_registration.py over the MI355X path and over the oracle: subjects on irregular spheres of
their own and an irregular template (every third run: the regular icosphere for all of them), two levels of random resolutions, smoothing, variance normalisation, masks, similarity measures.  The labelings of every
iteration must be identical and the registered spheres within 1e-4 rad (north_star).  A script, run by hand on a GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from helpers import OracleOps, angles  # noqa: E402
from newmsm_amd import group_registration as GR  # noqa: E402
from newmsm_amd import synthetic  # noqa: E402

ctx = M.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad, t0 = 0, time.time()
for k in range(n):
    S, D = int(rng.integers(2, 4)), int(rng.integers(1, 3))
    mesh_order = int(rng.choice([3, 4]))
    xyz, tri = M.make_mesh_from_icosa(mesh_order)
    regular = bool(rng.integers(0, 3) == 0)  # every third: the regular icosphere as the template and as every subject's sphere, as gMSM's scripts set a run up --
    if regular:                               # labels then carry data vertices exactly onto template vertices (exact with the host's rotation matrices, the default)
        txyz, meshes = xyz, [(xyz, tri) for _ in range(S)]
    else:
        txyz = synthetic.known_warp(xyz, seed=int(rng.integers(1, 10**6)), rot_deg=float(rng.uniform(2, 9)), amp=float(rng.uniform(0.5, 2.0)))
        meshes = [(synthetic.known_warp(xyz, seed=int(rng.integers(1, 10**6)), rot_deg=float(rng.uniform(0, 2)), amp=float(rng.uniform(0.3, 1.5))), tri) for _ in range(S)]
    datas = [synthetic.features(synthetic.known_warp(meshes[s][0], seed=int(rng.integers(1, 10**6)), rot_deg=3.0, amp=2.0), D, seed=5) for s in range(S)]
    sim = int(rng.choice([1, 2]))
    lam = float(rng.choice([1e-3, 1e-2, 0.1]))
    levels = []
    for lv in range(2):
        data_order = int(rng.integers(2, mesh_order + 1)) if lv == 0 else mesh_order
        cp_order = int(rng.integers(1, max(2, data_order - 1)))
        levels.append(dict(data_order=data_order, cp_order=cp_order, sg_order=cp_order + int(rng.integers(1, 3)), iters=int(rng.integers(1, 3)), simmeasure=sim,
                           sigma_in=float(rng.choice([0.0, 2.0, 4.0])), cost_params=dict(lambda_=lam, mu=0.4, kappa=1.6)))
    mask = (rng.random(len(xyz)) > 0.2).astype(np.float64) if rng.integers(0, 2) else None
    kw = dict(mask=mask, varnorm=bool(rng.integers(0, 2)), fixnan=True)
    lg, lw = [], []
    got = GR.run_group_multiresolution(GR.ProductGroupOps(ctx), meshes, datas, txyz, tri, levels, labelings_out=lg, **kw)
    want = GR.run_group_multiresolution(OracleOps(M.mcmc_optimise), meshes, datas, txyz, tri, levels, labelings_out=lw, **kw)
    same = len(lg) == len(lw) and all(np.array_equal(a, b) for a, b in zip(lg, lw))
    ang = max(float(angles(got[0][s], want[0][s]).max()) for s in range(S))
    ok = same and ang <= 1e-4
    bad += 0 if ok else 1
    print("ok" if ok else "MISMATCH", k, "regular" if regular else "irregular", "S=%d D=%d mesh=ico%d levels=%s sim=%d lambda=%g mask=%s vn=%s: labelings %s, %.1e rad, %d labels taken" % (
        S, D, mesh_order, [(l["data_order"], l["cp_order"], l["sg_order"], l["iters"], l["sigma_in"]) for l in levels], sim, lam, mask is not None, kw["varnorm"],
        "identical" if same else "DIFFER", ang, sum(int(np.count_nonzero(l)) for l in lg)), flush=True)
print("fuzz_group_multires: %d configs, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
