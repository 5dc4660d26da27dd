"""tools/time_query.py [reps] -- msm_query_triangles (Octree::get_closest_triangle + calc_barycentric_weights, R/octree.cpp:156-214,
R/resampler.cpp:142-167) at the sizes a registration meets; wall time per call includes the copies of the host-array entry point, so run it
under `rocprofv3 --kernel-trace --stats` for the kernel's own time (tools/collect_query_profile.sh).  MSMHIP_QUERY_LANES=4|8."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import newmsm_amd as M  # noqa: E402
from newmsm_amd import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = M.Context(0)
xyz6, tri6 = M.make_mesh_from_icosa(6)
xyz4, tri4 = M.make_mesh_from_icosa(4)
warped6 = synthetic.known_warp(xyz6, seed=3, rot_deg=2.0, amp=0.6)
warped4 = synthetic.known_warp(xyz4, seed=3, rot_deg=2.0, amp=0.6)
cases = [("40962 queries on a warped ico6 mesh", M.Mesh(ctx, warped6, tri6), xyz6), ("40962 queries on a warped ico4 mesh", M.Mesh(ctx, warped4, tri4), xyz6),
         ("2562 queries on a warped ico6 mesh", M.Mesh(ctx, warped6, tri6), xyz4)]
for name, mesh, q in cases:
    mesh.query_triangles(q)
    t0 = time.perf_counter()
    for _ in range(reps):
        st, tri, vid, w = mesh.query_triangles(q)
    dt = (time.perf_counter() - t0) / reps
    print("%-40s %8.1f us per call (host arrays in and out), %.1f M queries/s" % (name, dt * 1e6, len(q) / dt / 1e6), flush=True)
