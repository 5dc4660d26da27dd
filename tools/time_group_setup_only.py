"""tools/time_group_setup_only.py [cp_major] -- eight setupCostFunction calls of the 64-subject group at ico6 / ico4 (the first allocates), nothing else: the
workload of profiles/r5_setup_*.  cp_major: the pair list control point by control point, as every launched run has it (dist.sharded_group_setup)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import newmsm_amd as M
from newmsm_amd import problem

ctx = M.Context(0)
g, keep = problem.build_group(ctx, 64, 6, 4, D=2)
if len(sys.argv) > 1 and sys.argv[1] == "cp_major":
    g.set_pair_layout(g.CP_MAJOR)
for r in range(8):
    t0 = time.perf_counter()
    g.setupCostFunction()
    print("setup %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
