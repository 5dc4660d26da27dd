"""tools/time_query_kernel.py [reps] -- the GPU time of msm_query_triangles' kernel by HIP events on the context's stream (msm_ctx_time_queries), the three cases of
tools/time_query.py; no profiler attached."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = M.Context(0)
xyz6, tri6 = M.make_mesh_from_icosa(6)
xyz4, tri4 = M.make_mesh_from_icosa(4)
warped6 = synthetic.known_warp(xyz6, seed=3, rot_deg=2.0, amp=0.6)
warped4 = synthetic.known_warp(xyz4, seed=3, rot_deg=2.0, amp=0.6)
cases = [("40962 queries on a warped ico6 mesh", M.Mesh(ctx, warped6, tri6), xyz6), ("40962 queries on a warped ico4 mesh", M.Mesh(ctx, warped4, tri4), xyz6),
         ("2562 queries on a warped ico6 mesh", M.Mesh(ctx, warped6, tri6), xyz4)]
ctx.time_queries(True)
for name, mesh, q in cases:
    for _ in range(3):
        mesh.query_triangles(q)
    ks = []
    for _ in range(reps):
        mesh.query_triangles(q)
        ks.append(ctx.query_kernel_ms() * 1e3)
    print("%-40s kernel %6.2f us median, %6.2f us fastest (HIP events)" % (name, float(np.median(ks)), min(ks)), flush=True)
