"""first-use costs of a fresh ico6 mesh: creation, first query (tree build), first unfold (adjacency), first metric_resample onto it"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import synthetic
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(6)
src = M.Mesh(ctx, xyz, tri); data = synthetic.features(xyz, 1, 3)
q = xyz[:1000].copy()
def t(label, fn):
    t0 = time.perf_counter(); out = fn(); print("%-46s %.2f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True); return out
for rep in range(3):
    print("rep", rep)
    m = t("Mesh(ico6)", lambda: M.Mesh(ctx, synthetic.known_warp(xyz, seed=rep, rot_deg=1.0, amp=0.5), tri))
    t("first query (tree)", lambda: m.query_triangles(q))
    t("second query", lambda: m.query_triangles(q))
    t("first unfold (adjacency)", lambda: m.unfold())
    t("second unfold", lambda: m.unfold())
    t("metric_resample src -> m (first)", lambda: M.metric_resample(src, data, m))
    t("metric_resample src -> m (second)", lambda: M.metric_resample(src, data, m))
    t("sphere_project_warp (first)", lambda: M.sphere_project_warp(xyz, m, xyz))
    t("sphere_project_warp (second)", lambda: M.sphere_project_warp(xyz, m, xyz))
