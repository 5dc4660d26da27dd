"""one label step of Fusion for a group at ico6 / ico4: kernels only (results left in HBM), the C call that delivers them to
the host, and dist.ShardedMove at world size 1.  usage: time_group_step.py [S] [data_order cp_order]
environment: CHANGE=1 a tenth of the nodes changes its label between steps (as in bench.py), TORCH_STREAM=1 the context on a torch stream"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import newmsm_amd as M
from newmsm_amd import problem, dist as D
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
do, co = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 4)
if os.environ.get("TORCH_STREAM"):  # the context on a stream torch created, as bench.py has it
    _ts = torch.cuda.Stream()
    ctx = M.Context(0, stream=_ts.cuda_stream)
else:
    ctx = M.Context(0)
g, keep = problem.build_group(ctx, S, do, co, D=2, subjects=list(range(S)), template_order=os.environ.get("TEMPLATE_ORDER"))
t0 = time.perf_counter(); g.setupCostFunction(); print("set-up %.3f s (%.1f ms per subject)" % (time.perf_counter() - t0, (time.perf_counter() - t0) / S * 1e3), flush=True)
rng = np.random.default_rng(3)
lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
buf = torch.zeros(4 * g.P + 8 * g.T, dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()  # the fill is complete before the library writes into the tensor from its own stream
def dev_step(label):
    g.fusionMove_dev(lab, label, (0, g.P), (0, g.T), buf.data_ptr(), buf.data_ptr() + 8 * 4 * g.P)
    torch.cuda.synchronize()
pinned = (ctx.host_array((g.P, 4)), ctx.host_array((g.T, 8)))
if os.environ.get("CHANGE"):  # a tenth of the nodes changes its label between steps, as in bench.py
    _labs = [lab]
    for _ in range(12):
        _labs.append(np.where(rng.random(g.num_nodes) < 0.10, rng.integers(0, g.L, g.num_nodes), _labs[-1]).astype(np.int32))
    _k = [0]
    def _next():
        _k[0] += 1
        return _labs[_k[0] % len(_labs)]
    _fm = g.fusionMove
    g.fusionMove = lambda l_, label, out=None: _fm(_next(), label, out=out) if out is not None else _fm(_next(), label)
for name, fn in (("kernels (results in HBM)", dev_step), ("msm_group_fusion_move (host arrays)", lambda l: g.fusionMove(lab, l)),
                 ("msm_group_fusion_move (msm_host_alloc arrays)", lambda l: g.fusionMove(lab, l, out=pinned)),
                 ("ShardedMove.move, world 1", None)):
    if fn is None:
        mover = D.ShardedMove(g, D.Comm(None, "nccl", "cuda:0", 0, 1))
        fn = lambda l: mover.move(lab, l)
    fn(1)
    t0 = time.perf_counter()
    for i in range(5): fn(2 + i)
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for i in range(5): fn(2 + i)   # the same labels again: the second sweep of Fusion ((label, label) costs kept from the first)
    dt2 = (time.perf_counter() - t0) / 5
    print("%-48s %.2f ms per step (second visit of a label: %.2f ms): %.0f M pair + triplet evals/s" % (name, dt * 1e3, dt2 * 1e3, (4 * g.P + 8 * g.T) / dt / 1e6), flush=True)
