"""tools/time_cost_kinds.py -- wall time per call of the other cost classes at ico6 / ico4 (multivariate / patchwise tables,
triclique and strain fusion moves, pairwise table), including host transfers."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem
ctx = M.Context(0)
def timeit(f, n=5):
    f(); ctx.synchronize()
    t=time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize()
    return (time.perf_counter()-t)/n*1e3
for kind, D in (("multivariate", 32), ("patchwise", 32), ("multivariate", 3)):
    inp = problem.pairwise_inputs(6, 4, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind); cf.get_source_data(); keep["target"].prepare_search(wait=True)  # steady state: direction table in place
    ms = timeit(lambda: cf.computeUnaryCosts_async())
    print('%s D=%d unary table %.3f ms -> %.1f M evals/s' % (kind, D, ms, 48678/ms/1e3))
for kind, D in (("ho_univariate", 1), ("ho_multivariate", 32)):
    inp = problem.pairwise_inputs(6, 4, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind, rmode=3); 
    t=time.perf_counter(); cf.get_source_data(); print(kind, 'get_source_data %.1f ms' % ((time.perf_counter()-t)*1e3))
    lab = np.random.default_rng(0).integers(0, cf.L, cf.N).astype(np.int32)
    ms = timeit(lambda: cf.tripletOctets(lab, 3))
    print('%s D=%d triplet octets (8 x %d evals) %.3f ms -> %.2f M evals/s' % (kind, D, cf.T, ms, 8*cf.T/ms/1e3))
inp = problem.pairwise_inputs(6, 4, D=1)
cf, keep = problem.build_cost(ctx, inp, kind="univariate", rmode=3); cf.get_source_data()
lab = np.zeros(cf.N, dtype=np.int32)
print('strain octets %.3f ms' % timeit(lambda: cf.tripletOctets(lab, 3)))
cp, keep2 = problem.build_cost(ctx, inp, kind="univariate", rmode=1)
print('pairwise table (P x L x L = %d) %.3f ms' % (cp.P*cp.L*cp.L, timeit(lambda: cp.computePairwiseCosts(), 3)))
