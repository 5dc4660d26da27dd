"""tests/fuzz_parity.py [seed] [count] -- random configurations (resolutions, warps, target shapes, cost classes, similarity
measures) through the HIP path and the oracle; prints every mismatch.  Run on a GPU box; used to look for rare parity
failures beyond what tests/ samples (round 1: 2 250 configurations, with the direction table built up front and in the background, none)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem
from tests.helpers import oracle_cost
ctx = M.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for k in range(n):
    data_order = int(rng.choice([3, 4, 5]))
    cp_order = int(rng.integers(0, data_order - 1))
    shape = int(rng.integers(0, 4))
    kind = str(rng.choice(["univariate", "multivariate", "patchwise", "ho_univariate", "ho_multivariate"]))
    D = 1 if "uni" in kind else int(rng.integers(2, 41))
    rmode = int(rng.choice([3, 3, 2, 1])) if not kind.startswith("ho") else 3
    lam = float(rng.uniform(0.01, 0.5))
    kw = dict(seed=int(rng.integers(1, 10**6)), warp_amp=float(rng.uniform(0.0, 1.2)), warp_rot=float(rng.uniform(0.0, 4.0)),
              labeldist=float(rng.uniform(0.3, 0.7)), rescale=bool(rng.integers(0, 2)))
    if shape == 1: kw["target_warp"] = float(rng.uniform(0.5, 3.0))
    elif shape == 2: kw["target_radial"] = float(10 ** rng.uniform(-5, -2.5))
    elif shape == 3: kw["target_noise"] = float(rng.uniform(0.1, 1.0))
    sim = int(rng.choice([1, 2, 4, 5]))
    try:
        inp = problem.pairwise_inputs(data_order, cp_order, D=D, **kw)
        cf, keep = problem.build_cost(ctx, inp, kind=kind, simmeasure=sim, rmode=rmode, lambda_=lam)
        cf.get_source_data()
        oc = oracle_cost(inp, kind, simmeasure=sim, rmode=rmode, lambda_=lam)
        if rmode != 1: oc.set_pairs(np.zeros((0, 2), dtype=np.int32))   # the model holds pairs or triplets, never both (M/DiscreteModel.cpp:99-102)
        else: oc.set_triplets(np.zeros((0, 3), dtype=np.int32))
        oc.get_source_data()
        if kind.startswith("ho"):
            T, L = cf.T, cf.L
            q = [rng.integers(0, T, 150).astype(np.int32)] + [rng.integers(0, L, 150).astype(np.int32) for _ in range(3)]
            got = cf.computeTripletCost(*q); want = np.array([oc.triplet(*r) for r in zip(*q)])
            # a whole label step of Fusion (I/Fusion/Fusion.h:181-196): the fused fusion-move kernel (direction-table targets, bins of
            # <= 128 points) or the split path, against the oracle's replay of the 8 T calls
            lab = rng.integers(0, L, cf.N).astype(np.int32)
            label = int(rng.integers(0, L))
            E, Eo = cf.tripletOctets(lab, label), oc.triplet_octets(lab, label, threads=8)
            fin = np.isfinite(Eo)
            if not (np.array_equal(np.isfinite(E), fin) and np.allclose(E[fin], Eo[fin], rtol=1e-9, atol=1e-11)):
                print("   octets differ: max %.3e" % np.nanmax(np.abs(E - Eo)))
                got = np.full_like(want, np.inf)  # counted as a mismatch below
        else:
            got, want = cf.computeUnaryCosts(), oc.unary_table()
        both = np.isfinite(want)
        ok = np.array_equal(np.isfinite(got), both) and np.allclose(got[both], want[both], rtol=1e-9, atol=1e-11)
        if ok and not kind.startswith("ho"):  # the regulariser of this configuration: pair or triplet cliques, and the total
            L = cf.L
            if rmode == 1:
                q = [rng.integers(0, cf.P, 100).astype(np.int32)] + [rng.integers(0, L, 100).astype(np.int32) for _ in range(2)]
                g2, w2 = cf.computePairwiseCost(*q), np.array([oc.pairwise(*r) for r in zip(*q)])
            else:
                q = [rng.integers(0, cf.T, 100).astype(np.int32)] + [rng.integers(0, L, 100).astype(np.int32) for _ in range(3)]
                g2, w2 = cf.computeTripletCost(*q), np.array([oc.triplet(*r) for r in zip(*q)])
            lab = rng.integers(0, L, cf.N).astype(np.int32)
            tg, to = cf.evaluateTotalCostSum(lab), oc.total(lab)
            ok_clique = np.allclose(g2, w2, rtol=1e-9, atol=1e-11, equal_nan=True)
            ok_total = abs(tg[0] - to[0]) <= 1e-9 * abs(to[0]) + 1e-11
            ok = ok_clique and ok_total
            if not ok:
                print("   rmode %d lambda %.3f: cliques %s (max diff %.3e), total %s: %r vs %r" % (rmode, lam, ok_clique, np.nanmax(np.abs(g2 - w2)), ok_total, tg, to))
    except M.MsmError as e:
        ok = "octree" in str(e) or "bounding box" in str(e)   # the reference throws on these inputs too
        print("   (error: %s)" % str(e)[:80])
    if not ok:
        bad += 1
        print("MISMATCH", k, kind, D, sim, data_order, cp_order, kw, flush=True)
    else:
        print("ok", k, kind, D, sim, data_order, cp_order, shape, flush=True)
print("fuzz: %d configs, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
