"""metric_resample / sphere_project_warp / smooth_data wall time at ico6 (per call, after warm-up)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import synthetic
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(6)
lo, ltri = M.make_mesh_from_icosa(4)
warped = synthetic.known_warp(xyz, seed=3, rot_deg=2.0, amp=0.6)
data = synthetic.features(xyz, 2, 7)
src = M.Mesh(ctx, warped, tri); ico = M.Mesh(ctx, xyz, tri); low = M.Mesh(ctx, lo, ltri)
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
def fresh():
    src.set_coords(warped)   # new coordinates -> new octree, as in every iteration
    return M.metric_resample(src, data, ico)
print("metric_resample ico6 -> ico6 (fresh source octree each call): %.2f ms" % t(fresh))
print("metric_resample ico6 -> ico6 (trees kept):                   %.2f ms" % t(lambda: M.metric_resample(src, data, ico)))
print("metric_resample ico6 -> ico4:                                %.2f ms" % t(lambda: M.metric_resample(src, data, low)))
print("sphere_project_warp ico6 through ico4 grid:                  %.2f ms" % t(lambda: M.sphere_project_warp(xyz, low, lo)))
