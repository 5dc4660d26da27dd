"""tools/time_fusion_moves.py -- wall time of successive fusion-move (triplet octet) calls of the HO classes."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem
ctx = M.Context(0)
for kind, D in (("ho_univariate", 1), ("ho_multivariate", 32), ("ho_univariate", 1)):
    inp = problem.pairwise_inputs(6, 4, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind, rmode=3); cf.get_source_data()
    lab = np.random.default_rng(0).integers(0, cf.L, cf.N).astype(np.int32)
    ts = []
    for k in range(8):
        t = time.perf_counter(); cf.tripletOctets(lab, 3 + (k % 3)); ts.append((time.perf_counter() - t) * 1e3)
    print(kind, D, " ".join("%.2f" % x for x in ts))
