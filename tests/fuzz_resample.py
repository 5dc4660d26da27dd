"""tests/fuzz_resample.py [seed] [count] -- random resampler configurations through the HIP path and the oracle: searches and
weights (both modes), adaptive-barycentric CSR with and without an exclusion mask, metric_resample, sphere_project_warp,
nearest-neighbour interpolation, smooth_data and unfold on randomly warped / jittered / folded spheres.  Everything but
smooth_data (exp / asin on the device) must be bit-exact.  A script, run by hand on a GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402

ctx = M.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad, t0 = 0, time.time()


def shape(xyz, kind, seed):
    if kind == 1:
        return synthetic.known_warp(xyz, seed=seed, rot_deg=float(rng.uniform(0, 5)), amp=float(rng.uniform(0.2, 2.0)))
    if kind == 2:
        out = xyz + np.random.default_rng(seed).normal(scale=float(rng.uniform(0.05, 0.8)), size=xyz.shape)
        return out / np.linalg.norm(out, axis=1, keepdims=True) * 100.0
    return xyz


for k in range(n):
    oa, ob = int(rng.choice([2, 3, 4, 5])), int(rng.choice([2, 3, 4]))
    axyz, atri = M.make_mesh_from_icosa(oa)
    bxyz, btri = M.make_mesh_from_icosa(ob)
    ka, kb = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    axyz, bxyz = shape(axyz, ka, int(rng.integers(1, 10**6))), shape(bxyz, kb, int(rng.integers(1, 10**6)))
    D = int(rng.integers(1, 4))
    data = synthetic.features(axyz, D, int(rng.integers(1, 1000)))
    excl = (rng.random(len(axyz)) > float(rng.uniform(0.05, 0.5))).astype(float) if rng.integers(0, 2) else None
    ma, mb, oma, omb = M.Mesh(ctx, axyz, atri), M.Mesh(ctx, bxyz, btri), O.Mesh(axyz, atri), O.Mesh(bxyz, btri)
    what = []
    try:
        q = synthetic.random_sphere_points(500, seed=int(rng.integers(1, 10**6)))
        for mode, raw in ((M.WEIGHTS_PROJECTED, False), (M.WEIGHTS_RAW, True)):
            st, t, vid, w = ma.query_triangles(q, mode=mode, check_status=False)
            ost, ot, ovid, ow = O.Octree(oma).barycentric_weights(q, raw=raw)
            if not (np.array_equal(t, ot) and np.array_equal(vid, ovid) and np.array_equal(w, ow, equal_nan=True)):
                what.append("query/%d" % mode)
        rp, col, val = M.get_adaptive_barycentric_weights(ma, mb, excl)
        orp, ocol, oval = O.adaptive_barycentric_weights(oma, omb, excl)
        if not (np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(val, oval, equal_nan=True)):
            what.append("adaptive weights")
        if excl is None:
            if not np.array_equal(M.metric_resample(ma, data, mb), O.metric_resample(oma, data, omb), equal_nan=True):
                what.append("metric_resample")
            if not np.array_equal(M.nearest_neighbour_interpolation(ma, data, bxyz), O.nearest_neighbour(oma, data, bxyz)):
                what.append("nearest neighbour")
        else:
            g, gm = M.metric_resample(ma, data, mb, excl=excl)
            w_, wm = O.metric_resample_excl(oma, data, omb, excl)
            if not (np.array_equal(g, w_, equal_nan=True) and np.array_equal(gm, wm, equal_nan=True)):
                what.append("metric_resample + mask")
            g, gm = M.nearest_neighbour_interpolation(ma, data, bxyz, excl=excl)
            w_, wm = O.nearest_neighbour_excl(oma, data, bxyz, excl)
            if not (np.array_equal(g, w_) and np.array_equal(gm, wm)):
                what.append("nearest neighbour + mask")
        to = synthetic.known_warp(axyz, seed=int(rng.integers(1, 10**6)), rot_deg=2.0, amp=0.8)
        if not np.array_equal(M.sphere_project_warp(bxyz, ma, to), O.sphere_project_warp(bxyz, oma, to)):
            what.append("sphere_project_warp")
        if oa <= 4:
            sig = float(rng.uniform(2.0, 12.0))
            if not np.allclose(M.smooth_data(ma, data, ma, sig), O.smooth_data(oma, data, oma, sig), rtol=1e-11, atol=1e-13, equal_nan=True):
                what.append("smooth_data")
        bad_xyz = bxyz.copy()
        nbr_ptr, nbr, _, _ = M.mesh_adjacency(btri, len(bxyz))
        for v in rng.choice(len(bxyz), int(rng.integers(1, 12)), replace=False):
            nb = nbr[nbr_ptr[v] + rng.integers(0, nbr_ptr[v + 1] - nbr_ptr[v])]
            p = bxyz[nb] + float(rng.uniform(1.05, 1.6)) * (bxyz[nb] - bxyz[v])
            bad_xyz[v] = p * 100.0 / np.linalg.norm(p)
        mb.set_coords(bad_xyz)
        omb.set_coords(bad_xyz)
        if not (mb.unfold() == O.unfold(omb) and np.array_equal(mb.get_coords(), omb.xyz)):
            what.append("unfold")
    except (M.MsmError, RuntimeError) as e:
        print("   (error on both sides expected for this input: %s)" % str(e)[:70])
    if what:
        bad += 1
    print("MISMATCH" if what else "ok", k, "ico%d(%d) -> ico%d(%d) D=%d excl=%s %s" % (oa, ka, ob, kb, D, excl is not None, what), flush=True)
print("fuzz_resample: %d configs, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
