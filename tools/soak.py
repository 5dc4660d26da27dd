"""tools/soak.py [cycles] -- repeats whole workflows (a three-level pairwise registration; a gMSM group: build, set-up twice, label steps,
destroy) and prints the device memory in use after every cycle: it must level off (the handles' memory pool keeps idle buffers up to
MSMHIP_POOL_MB; nothing may grow with the number of cycles)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import newmsm_amd as M
from newmsm_amd import problem, registration, synthetic

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = M.Context(0)
xyz, tri = M.make_mesh_from_icosa(6)
ref = synthetic.features(xyz, 1, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 1, 7)
levels = [dict(data_order=4, cp_order=2, sigma_in=4.0, sigma_ref=4.0), dict(data_order=5, cp_order=3, sigma_in=2.0, sigma_ref=2.0),
          dict(data_order=6, cp_order=4, sigma_in=1.0, sigma_ref=1.0)]
ops = registration.ProductOps(ctx)
rng = np.random.default_rng(0)
used = []
for c in range(cycles):
    t0 = time.perf_counter()
    registration.run_multiresolution(ops, xyz, tri, src, xyz, tri, ref, levels, varnorm=True, iters=2, mciters=20, mcparam=0.8, seed=1, cost_params=dict(lambda_=0.1))
    S = 8 + 4 * (c % 3)  # group sizes vary from cycle to cycle
    g, keep = problem.build_group(ctx, S, 5, 3, D=2)
    g.setupCostFunction(); g.setupCostFunction()
    lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    for l in range(4):
        q, o = g.fusionMove(lab, l)
        lab = np.where(rng.random(g.num_nodes) < 0.1, rng.integers(0, g.L, g.num_nodes), lab).astype(np.int32)
    g.close(); del g, keep; gc.collect()
    free, total = torch.cuda.mem_get_info(0)
    used.append((total - free) / 2**20)
    print("cycle %2d: %.2f s, device memory in use %.0f MiB" % (c, time.perf_counter() - t0, used[-1]), flush=True)
tail = used[len(used) // 2:]
print("second half: min %.0f max %.0f MiB -> %s" % (min(tail), max(tail), "level" if max(tail) - min(tail) < 64 else "GROWING"))
