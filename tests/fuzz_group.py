"""tests/fuzz_group.py [seed] [count] -- random groupwise (gMSM) configurations through the HIP path and the oracle: pair lists,
patch index sets, inter-subject pairwise costs (all four similarity measures, with and without a mask) and strain triplets.
A script, run by hand on a GPU box (round 1: 120 configurations, no mismatch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402

ctx = M.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad, t0 = 0, time.time()
for k in range(n):
    S = int(rng.integers(2, 5))
    data_order = int(rng.choice([3, 4]))
    cp_order = int(rng.integers(1, data_order - 1))
    big = rng.integers(0, 5) == 0
    if big:  # round 4: a control grid far coarser than the template -- patches of hundreds to thousands of entries (beyond the register
        cp_order, data_order = int(rng.integers(0, 2)), int(rng.choice([4, 5]))  # rounds and the membership bits of k_group_pairwise), DICE left to LDS-sized patches
        S = 2
    os.environ["MSMHIP_GROUP_PAIR_LANES"] = str(rng.choice([16, 32]))  # read when the set-up is finalised
    D = int(rng.integers(1, 4))
    sim = int(rng.choice([1, 2, 4, 5]))
    if big:
        sim = int(rng.choice([1, 2]))  # (DICE packs a patch's common entries into LDS: at most 2 560 entries, a loud MSM_ERR_CAPACITY beyond)
    mask = bool(rng.integers(0, 2))
    pct = float(rng.uniform(0.2, 0.9))
    amp, rot = float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.0, 4.0))
    dxyz, dtri = M.make_mesh_from_icosa(data_order)
    cxyz, ctri = M.make_mesh_from_icosa(cp_order)
    _, mvd = M.cp_spacings(cxyz, ctri)
    samples, _ = M.label_sampling_grid(cp_order + 2, float(rng.uniform(0.3, 0.6)) * mvd)
    mk = np.cos(dxyz[:, 0] / 30.0) if mask else None
    g = M.DiscreteGroupCostFunction(ctx, S, simmeasure=sim, lambda_=0.2, percentile=pct)
    og = O.Group(S, simmeasure=sim, lambda_=0.2, percentile=pct)
    tm, otm = M.Mesh(ctx, dxyz, dtri), O.Mesh(dxyz, dtri)
    g.set_template(tm, mk)
    og.set_template(otm, mk)
    g.Initialize(cxyz, ctri)
    og.set_controlgrid(O.Mesh(cxyz, ctri))
    keep = []
    for s in range(S):
        seed = int(rng.integers(1, 10**6))
        sph = synthetic.known_warp(dxyz, seed=seed, rot_deg=rot + s, amp=amp)
        feat = synthetic.features(synthetic.known_warp(dxyz, seed=seed + 1, rot_deg=2.0, amp=1.0), D, seed=5)
        regular, om = M.Mesh(ctx, dxyz, dtri), O.Mesh(dxyz, dtri)
        g.reset_meshspace(s, regular, feat)
        og.set_subject(s, om, feat)
        regular.set_coords(sph)
        om.set_coords(sph)
        g.reset_meshspace(s, regular, feat)
        og.set_subject(s, om, feat)
        cp_s = synthetic.known_warp(cxyz, seed=seed, rot_deg=rot + s, amp=amp)
        g.reset_CPgrid(s, cp_s)
        og.reset_cpgrid(s, cp_s)
        keep += [regular, om]
    g.set_labels(samples)
    og.set_labels(samples)
    g.setupCostFunction()
    og.setup()
    why = []
    ok = np.array_equal(g.getPairs(), og.pairs()) and np.array_equal(g.getTriplets(), og.triplets())
    if not ok:
        why.append("pair / triplet lists")
    for s, v, l in zip(rng.integers(0, S, 6), rng.integers(0, len(cxyz), 6), rng.integers(0, g.L, 6)):
        same = np.array_equal(g.patch(s, v, l)[0], og.patch(s, v, l)[0])
        if not same:
            why.append("patch ids of (%d, %d, %d): %d against %d entries" % (s, v, l, len(g.patch(s, v, l)[0]), len(og.patch(s, v, l)[0])))
        ok = ok and same
    p, la, lb = (rng.integers(0, g.P, 150).astype(np.int32), rng.integers(0, g.L, 150).astype(np.int32), rng.integers(0, g.L, 150).astype(np.int32))
    got, want = g.computePairwiseCost(p, la, lb), np.array([og.pairwise(*q) for q in zip(p, la, lb)])
    fin = np.isfinite(want)
    pair_ok = np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], want[fin], rtol=1e-9, atol=1e-11)
    if not pair_ok:
        differs = (np.isfinite(got) != fin) | (fin & ~np.isclose(got, want, rtol=1e-9, atol=1e-11))
        bad_i = np.nonzero(differs)[0]
        # A correlation over one or two common template vertices is rounding noise in the reference itself (with one vertex the weighted mean w a / w equals a or
        # misses it by an ulp, so the "variance" is 0 or 1e-34 and the cost 0.5 or 0 / 1): the resampled maps of the two sides agree to 1e-12, not to the bit, and
        # may land on different sides.  Such queries are reported, not counted.
        pr_, N_ = g.getPairs(), len(cxyz)
        real = []
        for i in bad_i:
            na, nb = int(pr_[p[i], 0]), int(pr_[p[i], 1])
            oa, _ = og.patch(na // N_, na % N_, int(la[i]))
            ob, _ = og.patch(nb // N_, nb % N_, int(lb[i]))
            ncommon = len(np.intersect1d(oa, ob))
            if ncommon > 2 or sim != 2:
                real.append((int(p[i]), int(la[i]), int(lb[i]), ncommon, float(got[i]), float(want[i])))
        if real:
            why.append("pair costs: %d of 150 differ, e.g. (pair, la, lb, common vertices, got, want) = %r" % (len(real), real[0]))
        else:
            print("   (%d pair cost(s) over at most two common vertices differ: ill-conditioned in the reference itself, not counted)" % len(bad_i), flush=True)
            pair_ok = True
    ok = ok and pair_ok
    t, a, b, c = (rng.integers(0, g.T, 100).astype(np.int32), *[rng.integers(0, g.L, 100).astype(np.int32) for _ in range(3)])
    got, want = g.computeTripletCost(t, a, b, c), np.array([og.triplet(*q) for q in zip(t, a, b, c)])
    if not np.allclose(got, want, rtol=1e-9, atol=1e-11):
        why.append("triplet costs")
    ok = ok and np.allclose(got, want, rtol=1e-9, atol=1e-11)
    # label steps as Fusion makes them, labels revisited (second sweep) with the labeling changing in between: the step's kept
    # (current, current) and (label, label) costs against the explicit batch evaluation of all 4 P combinations
    lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    pr, pp, kk = g.getPairs(), np.repeat(np.arange(g.P, dtype=np.int32), 4), np.tile(np.arange(4), g.P)
    l1, l2 = (int(x) for x in rng.integers(0, g.L, 2))
    for label in (l1, l2, l1, l2):
        quads, _ = g.fusionMove(lab, label)
        la = np.where(kk & 2, label, lab[pr[pp, 0]]).astype(np.int32)
        lb = np.where(kk & 1, label, lab[pr[pp, 1]]).astype(np.int32)
        step_ok = np.array_equal(quads.ravel(), g.computePairwiseCost(pp, la, lb), equal_nan=True)
        if not step_ok:
            why.append("label step %d against the batch evaluation" % label)
        ok = ok and step_ok
        lab = np.where(rng.random(g.num_nodes) < 0.2, label, lab).astype(np.int32)
    if rng.integers(0, 3) == 0:
        # round 4: the pair list control point by control point (msm_group_set_pair_layout, the layout of sharded runs): the same pairs, and label steps
        # -- first visits, a changed labeling, second visits with the kept costs -- equal to the reference order's under the permutation, bit for bit
        ref_pairs = pr.copy()
        steps, lab2 = [], lab.copy()
        for label in (l1, l2, l1):
            steps.append((lab2.copy(), label))
            lab2 = np.where(rng.random(g.num_nodes) < 0.2, label, lab2).astype(np.int32)
        want_q = [np.array(g.fusionMove(*st)[0]) for st in steps]
        g.set_pair_layout(g.CP_MAJOR)
        g.setupCostFunction()
        where = {(int(a), int(b)): i for i, (a, b) in enumerate(ref_pairs)}
        try:
            pos = np.array([where[(int(a), int(b))] for a, b in g.getPairs()])
            lay_ok = sorted(pos.tolist()) == list(range(g.P))
        except KeyError:
            lay_ok = False
        if lay_ok:
            for st, wq in zip(steps, want_q):
                lay_ok = lay_ok and np.array_equal(g.fusionMove(*st)[0], wq[pos], equal_nan=True)
        if not lay_ok:
            why.append("control-point-major pair list")
        ok = ok and lay_ok
    if not ok:
        bad += 1
    print("ok" if ok else "MISMATCH", k, "S=%d data=%d cp=%d D=%d sim=%d mask=%s pct=%.2f lanes=%s" % (S, data_order, cp_order, D, sim, mask, pct, os.environ["MSMHIP_GROUP_PAIR_LANES"]),
          "; ".join(why), flush=True)
print("fuzz_group: %d configs, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
