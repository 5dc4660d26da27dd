#!/usr/bin/env python3
"""bench.py -- label-cost evals/sec of the MI355X hot path (BASELINE.json metric).

One "step" = one computeUnaryCosts() pass (the full unary label-cost table: every control point x every
label) of BASELINE config 2: pairwise sulc (D = 1) registration, ico6 data grid (40 962 vertices),
ico4 control grid (2 562 nodes), 19 labels -> 48 678 evals = 3.18 M point samples per step.  All inputs
are resident in HBM before the timed region starts.

Multi-GPU (--gpus N, one process per GPU under torch.distributed.run): a pairwise registration does not
shard; ranks are independent replicas working on different synthetic subjects --
no data-path collective, weak scaling.  value = evals of all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes(samples, evals, D):
    """per point sample 116 + 32*D bytes (source coord 24 + hit-triangle vertex ids 12 +
    3 vertex coords 72 + 3*D reference values 24*D + source feature 8*D + weight 8); per eval 72 (rotation) + 8 (output)."""
    return samples * (116 + 32 * D) + evals * 80


DOMINANT_KERNEL = "msm::k_unary_rays"  # the sampling kernel of a simple-surface target (newmsm_amd/csrc/unary_kernels.hip)
PMC_PROFILE = os.path.join("profiles", "r1_n_unary_pmc.json")  # tools/collect_profile.sh on this workload


def kernel_algorithmic_bytes(samples, evals, D):
    """The dominant kernel's share of the section 8(d) figure: it reads the source coordinate (24), the hit triangle's
    vertex ids (12) and coordinates (72) and, for D = 1, the three reference values (24) of every point sample, and the
    rotation (72) of every eval.  The moving feature, its weight and the cost output belong to the reduction kernel."""
    return samples * (108 + (24 if D == 1 else 0)) + evals * 72


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate --pmc runs,
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the default workload."""
    if (args.data_order, args.cp_order, args.dims) != (6, 4, 1):
        return None
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            return json.load(f)["kernels"][DOMINANT_KERNEL]["hbm_traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(inp, kind, threads):
    """The oracle (CPU restatement of the reference algorithm, OpenMP over control points like
    M/DiscreteCostFunction.cpp:238-242) on the same workload.  Checker / baseline only."""
    from tests.helpers import oracle_cost

    oc = oracle_cost(inp, kind)
    oc.get_source_data()
    t0 = time.perf_counter()
    U = oc.unary_table(threads=threads)  # also warms the caches
    first = time.perf_counter() - t0
    reps = max(1, min(200, int(12.0 / max(first, 1e-3))))  # about 12 s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        U = oc.unary_table(threads=threads)
    dt = time.perf_counter() - t0
    return U, reps * U.size / dt, dt, reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--data-order", type=int, default=6)
    ap.add_argument("--cp-order", type=int, default=4)
    ap.add_argument("--dims", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal of the N > 1 path on a one-GPU box: all ranks share cuda:0 and talk over gloo
    rehearse = os.environ.get("MSM_BENCH_REHEARSAL") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))  # nccl = RCCL over xGMI
    reduce_device = "cpu" if rehearse else "cuda"

    import __graft_entry__ as g

    if rank == 0:
        g.build()
    if dist is not None:
        dist.barrier()
    import newmsm_amd as M
    from newmsm_amd import problem

    kind = "univariate" if args.dims == 1 else "multivariate"
    stream = torch.cuda.Stream()
    ctx = M.Context(device_index, stream=stream.cuda_stream)
    # one synthetic subject per rank (different warp / feature phase)
    inp = problem.pairwise_inputs(args.data_order, args.cp_order, D=args.dims, seed=1234 + 17 * rank)
    cf, keep = problem.build_cost(ctx, inp, kind=kind)
    cf.get_source_data()
    # steady state of many tables against one target: have the target's direction table in place before the first step
    # (by default it is built in the background while the complete search serves the first tables)
    keep["target"].prepare_search(wait=True)
    ptr, _ = cf.patches()
    evals = cf.L * cf.N
    samples = cf.L * int(ptr[-1])

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cf.enable_timing(True)  # HIP events around the dominant kernel of every launch, on the launch stream
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            cf.computeUnaryCosts_async()
        ctx.synchronize()
        sync_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            cf.computeUnaryCosts_async()
        e1.record(stream)
        ctx.synchronize()
        sync_all()
        wall = time.perf_counter() - t0
    step_ms = e0.elapsed_time(e1) / args.steps  # HIP events on the launch stream: whole step (all kernels of one table)
    kt = cf.kernel_times()[-min(args.steps, 64):]
    kernel_ms = float(np.mean(kt))  # k_unary_samples alone (the dominant kernel), last <= 64 launches of the timed region
    U = cf.getUnaryCosts()
    if not np.isfinite(U).all() and not os.environ.get("MSM_BENCH_NOCHECK"):
        raise SystemExit("non-finite unary costs")

    if dist is not None:
        t = torch.tensor([wall], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        k = torch.tensor([kernel_ms], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    if rank == 0:
        value = world * evals * args.steps / wall
        abytes = algorithmic_bytes(samples, evals, args.dims)
        kbytes = kernel_algorithmic_bytes(samples, evals, args.dims)
        achieved = kbytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "label-cost evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "pairwise sulc (D=%d) unary label-cost table, ico%d data grid / ico%d control grid, %d labels, "
                            "%d evals = %d point samples per step (BASELINE config 2)" % (args.dims, args.data_order, args.cp_order, cf.L, evals, samples),
                "kind": kind, "simmeasure": "correlation", "replicas": world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args), "kernel": DOMINANT_KERNEL,
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": kbytes,
                # the whole table (sampling + fix-up + reduction kernels, HIP events around one step) against the full 8(d) figure
                "step_ms_events": step_ms, "algorithmic_bytes_per_table": abytes,
                "table_achieved": abytes / (step_ms * 1e-3) / 1e9, "table_frac": abytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            from newmsm_amd.dist import host_cores

            threads = host_cores()  # cgroup / affinity aware: the GPU box gives one GPU's share of the host
            Uo, rate, dt, reps = cpu_baseline(inp, kind, threads)
            out["cpu_baseline"] = {
                "value": rate, "unit": "evals/s", "cores": threads, "kind": "port",
                "sample": "%d full unary tables (%d evals each, %.1f s in total) of the same workload, OpenMP over control points" % (reps, Uo.size, dt),
                "max_abs_diff_vs_gpu": float(np.max(np.abs(Uo - U))),
            }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
