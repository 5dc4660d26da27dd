"""One-process-per-GPU plumbing (torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU).

The pairwise path does not shard: ranks are replicas.  Groupwise (gMSM) work shards by subject, and the one
exchange is the template update -- an all-reduce(sum) of per-rank accumulators (the
reference does this step with files + wb_command, gMSM_scripts/run_gMSM.sh:66-139).
"""
import os

import numpy as np


def env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def host_cores():
    """CPU cores this process may actually use (cgroup quota and affinity aware)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def init(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns torch.distributed or None."""
    rank, local_rank, world = env()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def shard(n_items, rank, world):
    """Contiguous, balanced shard of range(n_items) (subjects) for this rank."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def _tensor(x, device):
    import torch

    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=device)


def all_reduce_sum(arr, dist=None, device="cpu"):
    """Sum of a numpy array over all ranks (identity without a process group)."""
    if dist is None:
        return np.array(arr, dtype=np.float64)
    t = _tensor(arr, device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def max_over_ranks(value, dist=None, device="cpu"):
    if dist is None:
        return float(value)
    t = _tensor([value], device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def group_template_update(local_spheres, local_features=None, dist=None, device="cpu", radius=100.0):
    """Template update of a groupwise run: every rank holds the registered spheres (n_local x V x 3) and
    resampled feature maps (n_local x D x V) of ITS subjects.  One all-reduce(sum) of the accumulators
    [sum xyz | sum f | sum f^2 | count] yields, on every rank, the mean sphere (re-projected to `radius`), the
    group mean and the group standard deviation of the features -- what run_gMSM.sh computes with
    `wb_command -surface-average` and `-metric-reduce MEAN / STDEV`."""
    local_spheres = np.asarray(local_spheres, dtype=np.float64)
    V = local_spheres.shape[1]
    parts = [local_spheres.sum(axis=0).ravel()]
    D = 0
    if local_features is not None:
        local_features = np.asarray(local_features, dtype=np.float64)
        D = local_features.shape[1]
        parts += [local_features.sum(axis=0).ravel(), (local_features ** 2).sum(axis=0).ravel()]
    parts.append(np.array([float(local_spheres.shape[0])]))
    acc = all_reduce_sum(np.concatenate(parts), dist, device)
    n = acc[-1]
    mean_xyz = acc[: 3 * V].reshape(V, 3) / n
    norm = np.linalg.norm(mean_xyz, axis=1, keepdims=True)
    template = np.where(norm > 1e-8, mean_xyz / np.maximum(norm, 1e-300) * radius, mean_xyz)
    out = {"template": template, "n_subjects": int(round(n))}
    if D:
        s1 = acc[3 * V: 3 * V + D * V].reshape(D, V)
        s2 = acc[3 * V + D * V: 3 * V + 2 * D * V].reshape(D, V)
        mean = s1 / n
        out["mean"] = mean
        out["stdev"] = np.sqrt(np.maximum(s2 / n - mean ** 2, 0.0))
    return out


def sharded_group_setup(group, n_subjects, dist=None, device="cpu"):
    """Groupwise set-up with the subjects sharded over the ranks.

    `group` offers setup_subjects(list), export_subject(s) -> (F, pptr, pidx), import_subject(s, F, pptr, pidx) and
    finalize() (newmsm_amd.DiscreteGroupCostFunction).  Every rank runs the expensive per-subject work -- one rigid
    rotation + adaptive-barycentric resample per label -- for ITS subjects only; the resampled feature maps and
    patch lists are then broadcast from their owner (RCCL over xGMI with the nccl backend; F is L x D x V doubles,
    12.5 MB per subject at ico6 / 19 labels / D = 2) so that every rank can evaluate any inter-subject pair."""
    rank, _, world = env() if dist is not None else (0, 0, 1)
    mine = list(shard(n_subjects, rank, world))
    group.setup_subjects(mine)
    if dist is not None and world > 1:
        import torch

        for s in range(n_subjects):
            owner = next(r for r in range(world) if s in shard(n_subjects, r, world))
            if owner == rank:
                F, pptr, pidx = group.export_subject(s)
                meta = torch.tensor([F.size, len(pptr), len(pidx)] + list(F.shape), dtype=torch.int64, device=device)
            else:
                meta = torch.zeros(6, dtype=torch.int64, device=device)
            dist.broadcast(meta, src=owner)
            nF, npp, npi, L, D, V = (int(x) for x in meta.cpu().tolist())
            tF = _tensor(F.ravel(), device) if owner == rank else torch.zeros(nF, dtype=torch.float64, device=device)
            tI = (torch.as_tensor(np.concatenate([pptr, pidx]).astype(np.int32), device=device) if owner == rank
                  else torch.zeros(npp + npi, dtype=torch.int32, device=device))
            dist.broadcast(tF, src=owner)
            dist.broadcast(tI, src=owner)
            if owner != rank:
                ints = tI.cpu().numpy()
                group.import_subject(s, tF.cpu().numpy().reshape(L, D, V), ints[:npp], ints[npp:])
    group.finalize()
    return mine
