"""tools/time_iteration.py [kind [data_order cp_order [D]]] -- per-iteration set-up of a pairwise registration (default ico6 / ico4): reset_source (new source coordinates) + get_source_data + table"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem, synthetic
ctx = M.Context(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "univariate"
do, co = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 4)
inp = problem.pairwise_inputs(do, co, D=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
cf, keep = problem.build_cost(ctx, inp, kind=kind, rmode=3)
cf.get_source_data(); cf.computeUnaryCosts()
src, cpg = keep["source"], keep["cpgrid"]
for it in range(4):
    xyz = synthetic.known_warp(inp["source_orig_xyz"], seed=100 + it, rot_deg=2.0, amp=0.6)
    cxyz = synthetic.known_warp(inp["cp_orig_xyz"], seed=100 + it, rot_deg=2.0, amp=0.6)
    t0 = time.perf_counter(); src.set_coords(xyz); cpg.set_coords(cxyz); cf.reset_source(src); cf.reset_CPgrid(cpg); t1 = time.perf_counter()
    cf.set_labels(inp["labels"], M.cp_rotations(inp["samples"][0], cxyz)); t2 = time.perf_counter()
    cf.get_source_data(); t3 = time.perf_counter()
    U = cf.computeUnaryCosts(); t4 = time.perf_counter()
    print("iter %d: reset %.1f ms, labels %.1f ms, get_source_data %.1f ms, unary table (first after reset, incl. D2H) %.2f ms" % (it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
