"""tools/time_warp.py -- sphere_project_warp of the 40 962 data vertices through a control grid that has just moved (a new search tree per call, as in
every iteration of a registration: M/mesh_registration.cpp:224), control grids ico2 / ico3 / ico4; MSMHIP_OCTREE=host|gpu forces where the tree is built"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import synthetic
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(6)
for order in (2, 3, 4):
    lo, ltri = M.make_mesh_from_icosa(order)
    low = M.Mesh(ctx, lo, ltri)
    moved = [synthetic.known_warp(lo, seed=3 + k, rot_deg=1.0, amp=0.3) for k in range(4)]
    ts = []
    for k in range(12):
        low.set_coords(moved[k % 4])
        t0 = time.perf_counter()
        M.sphere_project_warp(xyz, low, moved[(k + 1) % 4])
        ts.append(time.perf_counter() - t0)
    print("sphere_project_warp ico6 through a moved ico%d grid: median %.2f ms (first %.2f, min %.2f)" % (order, np.median(ts[2:]) * 1e3, ts[0] * 1e3, min(ts) * 1e3), flush=True)
