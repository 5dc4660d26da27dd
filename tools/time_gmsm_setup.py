"""gMSM per-iteration set-up time (get_patch_data for S subjects) at BASELINE config 5 shape: ico6 data / ico4 control grid."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import synthetic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data_order, cp_order, D = 6, 4, 2
ctx = M.Context(0)
dxyz, dtri = M.make_mesh_from_icosa(data_order)
cxyz, ctri = M.make_mesh_from_icosa(cp_order)
_, mvd = M.cp_spacings(cxyz, ctri)
samples, _ = M.label_sampling_grid(cp_order + 2, 0.5 * mvd)
g = M.DiscreteGroupCostFunction(ctx, S, simmeasure=2, lambda_=0.2)
tm = M.Mesh(ctx, dxyz, dtri)
g.set_template(tm, None)
g.Initialize(cxyz, ctri)
keep = []
for s in range(S):
    sph = synthetic.known_warp(dxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)
    feat = synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=2.0, amp=1.0), D, seed=5)
    regular = M.Mesh(ctx, dxyz, dtri)
    g.reset_meshspace(s, regular, feat)
    regular.set_coords(sph)
    g.reset_meshspace(s, regular, feat)
    g.reset_CPgrid(s, synthetic.known_warp(cxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5))
    keep.append(regular)
g.set_labels(samples)
for rep in range(2):
    t0 = time.perf_counter()
    g.setupCostFunction()
    dt = time.perf_counter() - t0
    print("setupCostFunction: S=%d L=%d D=%d  %.2f s  (%.3f s per subject, %d resamples)  threads=%s" % (S, len(samples), D, dt, dt / S, S * len(samples), os.environ.get("MSMHIP_HOST_THREADS", "auto")), flush=True)
# pairwise costs throughput
rng = np.random.default_rng(1)
if g.P == 0: sys.exit(0)
n = 200000
p = rng.integers(0, g.P, n).astype(np.int32); la = rng.integers(0, g.L, n).astype(np.int32); lb = rng.integers(0, g.L, n).astype(np.int32)
g.computePairwiseCost(p[:1000], la[:1000], lb[:1000])
t0 = time.perf_counter(); g.computePairwiseCost(p, la, lb); dt = time.perf_counter() - t0
print("group pairwise: %d evals in %.3f s = %.2f M evals/s (incl. host transfers)" % (n, dt, n / dt / 1e6))
# one label step of Fusion::optimize: 4P pairwise + 8T triplet evaluations from the labeling alone
lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
g.fusionMove(lab, 3)
t0 = time.perf_counter(); quads, octets = g.fusionMove(lab, 5); dt = time.perf_counter() - t0
print("fusion move: %d pairwise + %d triplet evals in %.4f s = %.1f M evals/s (incl. the copy of the results to the host)" % (quads.size, octets.size, dt, (quads.size + octets.size) / dt / 1e6))
