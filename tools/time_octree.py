import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import newmsm_amd as M
from newmsm_amd import synthetic
xyz, tri = M.make_mesh_from_icosa(6)
w = synthetic.known_warp(xyz, seed=3, rot_deg=2.0, amp=0.6)
for rep in range(3):
    t0 = time.perf_counter(); s = M.octree_signature(w, tri); dt = time.perf_counter() - t0
    print("octree_signature %.2f ms" % (dt * 1e3), s[0] if isinstance(s, tuple) else s)
