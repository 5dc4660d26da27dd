"""tools/time_fusion_move.py KIND D [calls [data_order cp_order]] -- wall time per msm_cost_triplet_octets call (one label step of Fusion, I/Fusion/Fusion.h:181-196)
at ico6 / ico4 for an HO cost class; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem

kind, D = sys.argv[1], int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = M.Context(0)
do, co = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (6, 4)
inp = problem.pairwise_inputs(do, co, D=D)
cf, keep = problem.build_cost(ctx, inp, kind=kind, rmode=3, lambda_=0.01, mu=0.4, kappa=1.6)
cf.get_source_data()
rng = np.random.default_rng(0)
lab = rng.integers(0, cf.L, cf.N).astype(np.int32)
ref = cf.tripletOctets(lab, 3).copy()
for name, out in (("pageable E", None), ("msm_host_alloc E", ctx.host_array((cf.T, 8)))):
    for _ in range(3):
        got = cf.tripletOctets(lab, 3, out)
    assert os.environ.get("MSM_NOCHECK") or np.array_equal(got, ref)
    cf.enable_timing(True)
    ts = []
    for i in range(calls):
        t = time.perf_counter()
        cf.tripletOctets(lab, i % cf.L, out)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e6
    kt = cf.kernel_times() * 1e3
    cf.enable_timing(False)
    print("%s D=%d, %s: %d calls, per call median %.1f us, mean %.1f us, min %.1f us -> %.1f M evals/s; kernels (events) median %.1f us" % (
        kind, D, name, calls, np.median(ts), ts.mean(), ts.min(), 8 * cf.T / np.median(ts), np.median(kt) if len(kt) else float("nan")))
