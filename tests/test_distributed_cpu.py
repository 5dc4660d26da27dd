"""The N > 1 path on CPU: two gloo ranks (world_size 2) exercise sharding, the max-over-ranks timing reduction
and the group template all-reduce that the GPU box runs over RCCL."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

from newmsm_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partitions_every_subject_once():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in D.shard(n, r, world)]
            assert got == list(range(n))
            sizes = [len(D.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_template_update_matches_numpy():
    rng = np.random.default_rng(0)
    spheres = rng.normal(size=(5, 42, 3))
    feats = rng.normal(size=(5, 2, 42))
    out = D.group_template_update(spheres, feats)
    m = spheres.mean(axis=0)
    assert np.allclose(out["template"], m / np.linalg.norm(m, axis=1, keepdims=True) * 100.0)
    assert np.allclose(out["mean"], feats.mean(axis=0)) and np.allclose(out["stdev"], feats.std(axis=0))
    assert out["n_subjects"] == 5


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from newmsm_amd import dist as D
    rank, local_rank, world = D.env()
    dist = D.init("gloo")
    rng = np.random.default_rng(123)
    spheres = rng.normal(size=(7, 162, 3)); feats = rng.normal(size=(7, 3, 162))
    mine = list(D.shard(7, rank, world))
    out = D.group_template_update(spheres[mine], feats[mine], dist)
    tmax = D.max_over_ranks(1.0 + rank, dist)
    ref = D.group_template_update(spheres, feats)
    ok = all(np.allclose(out[k], ref[k], rtol=1e-12, atol=1e-12) for k in ("template", "mean", "stdev"))
    dist.barrier()
    print(json.dumps({"rank": rank, "ok": bool(ok), "n": out["n_subjects"], "tmax": tmax, "mine": mine}))
    dist.destroy_process_group()
""")


def test_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(eval(so.strip().splitlines()[-1].replace("true", "True").replace("false", "False")))
    assert all(o["ok"] for o in outs)
    assert all(o["n"] == 7 and o["tmax"] == 2.0 for o in outs)
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(7))


GROUP_WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from newmsm_amd import dist as D

    class FakeGroup:
        \"\"\"stands in for DiscreteGroupCostFunction: per-subject products are deterministic functions of the subject id\"\"\"
        L, D, N = 3, 2, 4                                    # 12 patch rows per subject
        class _T: V = 42
        _keep = {"template": _T}
        def __init__(self): self.have = {}; self.final = False
        def subject_index_count(self, s): return len(self.have[s][2])
        @staticmethod
        def make(s):
            rng = np.random.default_rng(100 + s)
            n = int(rng.integers(5, 40))
            pptr = np.sort(rng.integers(0, n, 12)).astype(np.int32); pptr[0] = 0; pptr = np.append(pptr, n).astype(np.int32)
            return rng.normal(size=(3, 2, 42)), pptr, rng.integers(0, 42, n).astype(np.int32)
        def setup_subjects(self, subjects):
            for s in subjects: self.have[s] = self.make(s)
        def export_subject(self, s): return self.have[s]
        def import_subject(self, s, F, pptr, pidx): self.have[s] = (np.array(F), np.array(pptr), np.array(pidx))
        def finalize(self): self.final = True

    rank, _, world = D.env()
    dist = D.init("gloo")
    g = FakeGroup()
    mine = D.sharded_group_setup(g, 5, dist)
    ok = g.final and sorted(g.have) == list(range(5))
    for s in range(5):
        F, pp, pi = FakeGroup.make(s)
        ok = ok and np.array_equal(g.have[s][0], F) and np.array_equal(g.have[s][1], pp) and np.array_equal(g.have[s][2], pi)
    dist.barrier()
    print(json.dumps({"rank": rank, "ok": bool(ok), "mine": mine}))
    dist.destroy_process_group()
""")


def test_sharded_group_exchange_two_gloo_ranks(tmp_path):
    script = tmp_path / "gworker.py"
    script.write_text(GROUP_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(eval(so.strip().splitlines()[-1].replace("true", "True").replace("false", "False")))
    assert all(o["ok"] for o in outs)
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(5))


SHM_WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from newmsm_amd import dist as D

    rank, _, world = D.env()
    dist = D.init("gloo")
    n = 1000
    buf = D.SharedStepBuffer(dist, n, dst=0)
    lo, hi = D.shard(n, rank, world).start, D.shard(n, rank, world).stop
    ok = D.same_node(dist)
    for step in range(1, 8):                                       # seven steps over two alternating slots
        buf.begin(step)                                            # producers wait for the consumer to release the slot
        slot = buf.data[(step - 1) %% buf.slots]
        slot[lo:hi] = 1000.0 * step + np.arange(lo, hi)            # this rank's slice, at its final position
        buf.publish(step)
        if rank == 0:
            got = buf.wait(step)
            ok = ok and np.array_equal(got, 1000.0 * step + np.arange(n))
    dist.barrier()
    ok = ok and not os.path.exists(buf.path)                       # unlinked once everyone had it mapped
    print(json.dumps({"rank": rank, "ok": bool(ok)}))
    dist.destroy_process_group()
""")


def test_shared_step_buffer_two_ranks(tmp_path):
    """dist.SharedStepBuffer (the single-node transport of ShardedMove): slices written by two processes arrive at rank 0 without
    a collective; the backing file is gone from /dev/shm as soon as both have mapped it."""
    script = tmp_path / "shm.py"
    script.write_text(SHM_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        assert json.loads(so.strip().splitlines()[-1])["ok"]


def _bench(argv, env=None, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` -- the shape of the driver's N = 1 command with N > 1 -- launches its own ranks (torch.distributed.run as a child),
    and n_gpus is what an all-reduce of ones over the communicator says.  --dry-launch stops after the communicator exists (no GPU here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rc, line, err = _bench(["--gpus", "2", "--dry-launch"], env)
    assert rc == 0, err[-2000:]
    assert line["dry_launch"] and line["self_launched"] and line["n_gpus"] == 2 and line["world"] == 2 and line["every_rank_reported"]
    t = line["template_allreduce"]  # the north-star collective ran over the same communicator
    assert t["ranks"] == 2 and t["sums_correct"] and t["bytes"] == 8 * (7 * 2562 + 1) and t["us"] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
    """under a launcher that started another number of ranks than --gpus says the bench ends non-zero instead of misreporting n_gpus"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    rc, line, err = _bench(["--gpus", "2", "--dry-launch"], env)
    assert rc != 0 and line is None and "--gpus 2" in err
