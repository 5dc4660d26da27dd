"""What ONE rank of a W-rank gMSM iteration does, timed on one GPU: the one-GPU cost of every term of the 8-GPU projection
except the wire time of the collectives.  Rank `r` of `W` at S subjects, ico<data> / ico<cp>:

  common      the part every rank repeats (estimate_pairs, spacings, rotations, ROT x label)            msm_group_setup_subjects, first part
  subjects    get_patch_data of the rank's S / W subjects                                                msm_group_setup_subjects, second part
  export      the rank's subjects into the send buffers (device-to-device)                               msm_group_export_subject_dev
  import      the other ranks' subjects out of the receive buffers (device-to-device + range checks)     msm_group_import_subject_dev
  finalize    pointer tables, patch statistics                                                           msm_group_finalize
  step        a label step on the rank's slice of the pair / triplet lists into pinned host memory       msm_group_fusion_move_dev

usage: time_group_rank.py [S] [W] [r] [data_order cp_order]     (MSMHIP_TIMING=1 prints the library's own phase timings; LAYOUT=0: the pair list in the
reference's order instead of control-point major, the layout dist.sharded_group_setup chooses for W > 1)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import newmsm_amd as M
from newmsm_amd import dist as D
from newmsm_amd import problem

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
do, co = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (6, 4)
ctx = M.Context(0)
mine = list(D.shard(S, r, W))
others = [s for s in range(S) if s not in mine]

# the whole group once (stands in for the other ranks): its subjects exported into device tensors = what the all-gathers deliver
LAYOUT = int(os.environ.get("LAYOUT", "1"))
full, keep_full = problem.build_group(ctx, S, do, co, D=2)
full.set_pair_layout(LAYOUT)
full.setupCostFunction()
L, Dm, V, Mrows = full.L, full.D, full._keep["template"].V, full.N * full.L + 1
counts = [full.subject_index_count(s) for s in range(S)]
imax = max(counts)
F = torch.zeros((S, L, Dm, V), dtype=torch.float64, device="cuda:0")
pp = torch.zeros((S, Mrows), dtype=torch.int32, device="cuda:0")
pi = torch.zeros((S, imax), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()  # torch's fills are complete before the library writes into the tensors from its own streams
for s in range(S):
    full.export_subject_dev(s, F[s].data_ptr(), pp[s].data_ptr(), pi[s].data_ptr(), imax)
torch.cuda.synchronize()

g, keep = problem.build_group(ctx, S, do, co, D=2, subjects=mine)
g.set_pair_layout(LAYOUT)
sendF = torch.zeros((len(mine), L, Dm, V), dtype=torch.float64, device="cuda:0")
sendpp = torch.zeros((len(mine), Mrows), dtype=torch.int32, device="cuda:0")
sendpi = torch.zeros((len(mine), imax), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()  # torch's fills are complete before the library writes into the tensors from its own streams


CHUNKS = int(os.environ.get("CHUNKS", "2"))


def iteration(report):
    """the calls of dist.sharded_group_setup on the nccl path, without the collectives: the shard set up and exported in CHUNKS pieces, the other
    ranks' subjects imported a (piece, rank) at a time"""
    t = {"set-up (common + subjects)": 0.0, "export": 0.0}
    bounds = D._chunk_bounds(len(mine), CHUNKS)
    for n, (k0, k1) in enumerate(bounds):
        part = mine[k0:k1]
        t0 = time.perf_counter()
        (g.setup_subjects if n == 0 else g.setup_more_subjects)(part)
        t["set-up (common + subjects)"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        g.export_subjects_dev(part, sendF[k0].data_ptr(), L * Dm * V, sendpp[k0].data_ptr(), Mrows, sendpi[k0].data_ptr(), imax)
        t["export"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    for k0, k1 in bounds:
        for rr in range(W):
            if rr == r:
                continue
            theirs = list(D.shard(S, rr, W))[k0:k1]
            s0 = theirs[0]
            g.import_subjects_dev(theirs, F[s0].data_ptr(), L * Dm * V, pp[s0].data_ptr(), Mrows, pi[s0].data_ptr(), imax, [counts[s] for s in theirs])
    t["import"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.finalize()
    t["finalize"] = time.perf_counter() - t0
    if report:
        print("rank %d of %d, S = %d, ico%d / ico%d: %d subjects of its own, set up and exchanged in %d piece(s); pair list %s" % (r, W, S, do, co, len(mine), len(bounds), "control-point major" if LAYOUT else "in the reference's order"))
        for k, v in t.items():
            print("  %-28s %7.2f ms" % (k, v * 1e3))
        print("  %-28s %7.2f ms   (all-gather payload of the group: %.2f GB)" % ("set-up, this rank", sum(t.values()) * 1e3, (F.numel() * 8 + pp.numel() * 4 + pi.numel() * 4) / 1e9), flush=True)
    return t


iteration(False)
iteration(True)
iteration(True)

# a label step on this rank's slice, delivered into pinned host memory at the slice's position (what SharedStepBuffer holds)
pr, tr = D.shard(g.P, r, W), D.shard(g.T, r, W)
out_q, out_t = ctx.host_array((g.P, 4)), ctx.host_array((g.T, 8))
rng = np.random.default_rng(3)
lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
labs = [lab]
for _ in range(12):
    labs.append(np.where(rng.random(g.num_nodes) < 0.10, rng.integers(0, g.L, g.num_nodes), labs[-1]).astype(np.int32))


def step(i, label):
    g.fusionMove_dev(labs[i % len(labs)], label, (pr.start, pr.stop), (tr.start, tr.stop), out_q.ctypes.data + 8 * 4 * pr.start, out_t.ctypes.data + 8 * 8 * tr.start)


for i in range(3):
    step(i, 1)
g.time_moves(True)
for sweep in ("first visit of a label", "second visit"):
    ts, ks = [], []
    for i in range(6):
        t0 = time.perf_counter()
        step(i, 2 + i)
        ts.append(time.perf_counter() - t0)
        ks.append(g.move_kernels_ms())
    print("  label step, slice 1/%d (%d pairs, %d triplets), %s: %.2f ms per call, %.2f ms of kernels" % (W, len(pr), len(tr), sweep, np.median(ts) * 1e3, np.median(ks)), flush=True)
g.time_moves(False)
# the same slice through dist.ShardedMove's shared-memory transport with one rank (begin / publish / wait + the views)
if os.environ.get("SHM", "1") == "1":
    mover = D.ShardedMove(g, D.Comm(), transport="shm")
    for i in range(3):
        mover.move(labs[i], 1)
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        mover.move(labs[i % len(labs)], 2 + i % 6)
        ts.append(time.perf_counter() - t0)
    print("  ShardedMove (shm transport, one rank = the WHOLE step): %.2f ms per step" % (np.median(ts) * 1e3), flush=True)
    mover.close()
