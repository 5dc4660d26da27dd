"""tests/fuzz_registration.py [seed] [count] -- random one-level registrations (resolutions, cost classes, label schedules,
optimiser seeds) driven over the HIP path and over the oracle by the same caller loop: labelings must be identical and
registered coordinates within the north star's 1e-4 rad (they agree to ~1e-12 mm unless a 1e-16 difference of two costs
flips a near-tie of the optimiser).  A script, run by hand on a GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import registration, synthetic  # noqa: E402
from tests.helpers import OracleOps  # noqa: E402

ctx = M.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad, flips, t0 = 0, 0, time.time()
for k in range(n):
    data_order = int(rng.choice([3, 4]))
    cp_order = int(rng.integers(1, data_order - 1))
    kind = str(rng.choice(["univariate", "multivariate", "patchwise", "ho_univariate", "ho_multivariate"]))
    D = 1 if kind in ("univariate", "ho_univariate") else int(rng.integers(2, 5) if kind != "ho_multivariate" else rng.choice([2, 5, 13, 20]))
    # the triclique classes are driven as --dopt=HOCR drives them: the label loop of Fusion over fusion moves (stand-in binary solve); the others
    # by either optimiser
    optimiser = "fusion" if kind.startswith("ho_") or rng.integers(0, 2) == 0 else "mcmc"
    xyz, tri = M.make_mesh_from_icosa(data_order)
    seed = int(rng.integers(1, 10**6))
    ref = synthetic.features(xyz, D, seed)
    src = synthetic.features(synthetic.known_warp(xyz, seed=seed + 2, rot_deg=float(rng.uniform(1, 5)), amp=float(rng.uniform(0.5, 3))), D, seed)
    kw = dict(cp_order=cp_order, iters=int(rng.integers(1, 4)), mciters=int(rng.integers(5, 60)), mcparam=float(rng.uniform(0.2, 0.9)), seed=seed, kind=kind,
              rescale_labels=bool(rng.integers(0, 2)), simmeasure=int(rng.choice([1, 2])), cost_params=dict(lambda_=float(rng.uniform(0.01, 0.3))),
              optimiser=optimiser)
    if kind.startswith("ho_"):
        kw["cost_params"].update(lambda_=float(rng.uniform(0.001, 0.05)), mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    pops, oops = registration.ProductOps(ctx), OracleOps(M.mcmc_optimise)
    akw_p, akw_o = {}, {}
    if optimiser == "fusion" and kind in ("univariate", "ho_univariate", "ho_multivariate") and rng.integers(0, 3) == 0:
        # round 4: --regoption=5 (aMSM): the level prepares the anatomical regulariser (resample_anatomy) from two synthetic anatomical surfaces
        kw.update(rmode=5, iters=min(kw["iters"], 2))
        kw["cost_params"].update(lambda_=float(rng.uniform(0.005, 0.05)), mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
        order = int(rng.integers(cp_order, min(cp_order + 3, data_order + 1)))
        ian, ran = synthetic.anatomy(xyz, seed=seed + 5, base=60.0), synthetic.anatomy(xyz, seed=seed + 6, base=62.0)
        akw_p = dict(anat=dict(order=order, in_anat=ian, ref_anat=ran, in_mesh=pops.mesh(xyz, tri), ref_mesh=pops.mesh(xyz, tri)))
        akw_o = dict(anat=dict(order=order, in_anat=ian, ref_anat=ran, in_mesh=oops.mesh(xyz, tri), ref_mesh=oops.mesh(xyz, tri)))
        kind = kind + " aMSM(anat ico%d)" % order
    got = registration.run_discrete_level(pops, xyz, tri, ref, xyz, tri, src, xyz, **kw, **akw_p)
    want = registration.run_discrete_level(oops, xyz, tri, ref, xyz, tri, src, xyz, **kw, **akw_o)
    same = all(np.array_equal(a, b) for a, b in zip(got[3], want[3]))
    ua, ub = got[0] / 100.0, want[0] / 100.0
    ang = float(np.max(2 * np.arcsin(np.minimum(1.0, 0.5 * np.linalg.norm(ua - ub, axis=1)))))
    ok = ang <= 1e-4
    flips += not same
    bad += not ok
    print("ok" if ok and same else ("FLIP" if ok else "MISMATCH"), k, kind, D, data_order, cp_order, "max angle %.2e rad" % ang, {a: kw[a] for a in ("iters", "mciters", "rescale_labels", "simmeasure", "optimiser")}, flush=True)
print("fuzz_registration: %d configs, %d beyond 1e-4 rad, %d with a different labeling, %.0f s" % (n, bad, flips, time.time() - t0))
