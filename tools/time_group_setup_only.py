import sys,time,os
sys.path.insert(0,".")
import newmsm_amd as M
from newmsm_amd import problem
ctx=M.Context(0)
g,keep=problem.build_group(ctx,64,6,4,D=2)
for r in range(8):
    t0=time.perf_counter(); g.setupCostFunction(); print("setup %.1f ms"%((time.perf_counter()-t0)*1e3),flush=True)
