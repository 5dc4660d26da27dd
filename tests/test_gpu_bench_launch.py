"""`python bench.py --gpus N` as the driver types it: the bench starts its own N ranks (torch.distributed.run as a child process), the line's
n_gpus is what the communicator counted, and the north-star collective (the group-mean template all-reduce) is in the line.  The test box has
one GPU, so the two ranks share it and talk over gloo (MSM_BENCH_REHEARSAL=1): everything but the transport of the collectives is the N > 1 path
(sharded set-up + all-gathers, label steps by clique through the shared step buffer)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_gpus_2_through_the_self_launch(built):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MSM_BENCH_REHEARSAL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--subjects", "16"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["metric"] == "label-cost evals/sec" and line["scaling"] == "weak" and line["value"] > 0
    g = line["gmsm"]
    assert "error" not in g, g
    assert g["subjects"] == 16 and len(g["levels"]) == 3 and g["subjects_per_hour"] > 0 and g["scaling"] == "strong"
    t = g["template_allreduce"]
    assert "error" not in t, t
    assert t["ranks"] == 2 and t["sums_correct"] and t["bytes"] == 8 * (7 * 40962 + 1) and t["us"] > 0


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_template_allreduce(built):
    """the N = 1 line carries gmsm.template_allreduce through a one-rank RCCL communicator (headline only otherwise: --mode gmsm on a small group)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "gmsm", "--steps", "2", "--subjects", "8"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1])
    t = line["gmsm"]["template_allreduce"]
    assert line["n_gpus"] == 1 and "error" not in t, t
    assert t["ranks"] == 1 and t["sums_correct"] and "nccl" in t["collective"]
