"""One-process-per-GPU plumbing (torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU).

The pairwise path does not shard: ranks are replicas.  Groupwise (gMSM) work shards twice:

  set-up      by SUBJECT: every rank runs get_patch_data (M/DiscreteGroupModel.cpp:88-121: a rigid rotation + an adaptive-
              barycentric resample per label) for its subjects only; the resampled feature maps and patch lists of all
              subjects are then all-gathered (three collectives for the whole group, device buffer to device buffer with
              the nccl backend) so that every rank can evaluate any inter-subject pair;
  evaluation  by CLIQUE: in every label step of Fusion (I/Fusion/Fusion.h:157-196) rank r evaluates its contiguous slice of
              the pair list (M/DiscreteGroupCostFunction.cpp:54-98 over N_cp * S (S - 1) / 2 pairs) and of the triplet list;
              the slices are gathered to the optimiser's rank;
  template    the group-mean update -- what gMSM_scripts/run_gMSM.sh:66-139 does with files and wb_command -- is one
              all-reduce(sum) of per-rank accumulators.

`Comm` carries the process group and the device its tensors live on: cuda:<local rank> for nccl (RCCL cannot reduce host
tensors), cpu for gloo (the world-size-2 rehearsal on CPU / on a one-GPU box).
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


class Comm:
    """torch.distributed process group + the device collectives run on.  `dist` is None for a single process."""

    def __init__(self, dist=None, backend=None, device="cpu", rank=0, world=1):
        self.dist, self.backend, self.device, self.rank, self.world = dist, backend, device, rank, world

    @property
    def on_gpu(self):
        return self.backend == "nccl"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.on_gpu:  # nothing of this communicator is still queued or running on the device when its buffers go
                import torch

                torch.cuda.synchronize()
            self.dist.destroy_process_group()
            self.dist = None

    destroy_process_group = close

    def all_gather(self, out, inp):
        """out[r] = rank r's inp (out: world x inp.shape), one collective.  Which form of the collective is used is decided once per
        communicator from the backend (every rank decides the same way); an error of the collective itself -- a communicator fault, a
        time-out, mismatched shapes -- propagates instead of turning into a different collective on this rank only."""
        if self.flat_all_gather:
            self.dist.all_gather_into_tensor(out, inp)
        else:
            self.dist.all_gather(list(out.unbind(0)), inp)

    @property
    def flat_all_gather(self):
        # nccl (RCCL) has the flat form; gloo of this torch build raises for it
        return self.backend == "nccl" and hasattr(self.dist, "all_gather_into_tensor")


def _timeout():
    """a collective that does not complete within MSMHIP_DIST_TIMEOUT_S (default 300 s) ends the job instead of hanging it"""
    import datetime

    return datetime.timedelta(seconds=float(os.environ.get("MSMHIP_DIST_TIMEOUT_S", "300")))


def init(backend=None, device_index=None):
    """Initialise the default process group when WORLD_SIZE > 1 (a single process gets a Comm without one).
    backend None: nccl when a GPU is visible, else gloo.  device_index: the GPU of this rank (default LOCAL_RANK)."""
    rank, local_rank, world = env()
    if world <= 1 and backend is None:
        return Comm()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        dev = local_rank if device_index is None else device_index
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev), rank=rank, world_size=world, timeout=_timeout())
        return Comm(dist, "nccl", "cuda:%d" % dev, rank, world)
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=_timeout())
    return Comm(dist, backend, "cpu", rank, world)


def _comm(c):
    """accepts a Comm, None, or (older call sites) a bare torch.distributed module initialised with gloo"""
    if c is None:
        return Comm()
    if isinstance(c, Comm):
        return c
    return Comm(c, c.get_backend(), "cpu" if c.get_backend() == "gloo" else "cuda:%d" % env()[1], c.get_rank(), c.get_world_size())


def shard(n_items, rank, world):
    """Contiguous, balanced shard of range(n_items) (subjects, pairs, triplets) for this rank."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def all_reduce_sum(arr, comm=None):
    """Sum of a numpy array over all ranks (identity without a process group)."""
    c = _comm(comm)
    if c.dist is None:
        return np.array(arr, dtype=np.float64)
    import torch

    t = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float64), device=c.device)
    c.dist.all_reduce(t, op=c.dist.ReduceOp.SUM)
    return t.cpu().numpy()


def count_ranks(comm=None):
    """how many ranks the communicator really has: an all-reduce(sum) of ones -- what RCCL saw, not what the environment says"""
    return int(round(float(all_reduce_sum(np.ones(1), comm)[0])))


def max_over_ranks(value, comm=None):
    c = _comm(comm)
    if c.dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=c.device)
    c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX)
    return float(t.item())


def group_template_update(local_spheres, local_features=None, comm=None, radius=100.0):
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
    acc = all_reduce_sum(np.concatenate(parts), comm)
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


def _chunk_bounds(nmax, chunks):
    """local index ranges [k0, k1) of the chunks a shard of up to nmax subjects is set up and exchanged in (the same on every rank)"""
    chunks = max(1, min(int(chunks), nmax)) if nmax > 0 else 1
    edges = [(i * nmax) // chunks for i in range(chunks + 1)]
    return [(edges[i], edges[i + 1]) for i in range(chunks) if edges[i + 1] > edges[i]]


def sharded_group_setup(group, n_subjects, comm=None, chunks=None, pair_layout=None):
    """Groupwise set-up with the subjects sharded over the ranks.

    `group` is a newmsm_amd.DiscreteGroupCostFunction.  Every rank runs the expensive per-subject work (get_patch_data,
    M/DiscreteGroupModel.cpp:88-121) for ITS subjects only; all-gathers then move every subject's resampled feature maps F (L x D x V
    doubles, 12.5 MB per subject at ico6 / 19 labels / D = 2), patch row pointers and patch index lists to every rank.

    nccl backend: the buffers are torch tensors on the GPU that libmsmhip fills and reads with device-to-device copies, a chunk of subjects
    per call (msm_group_export_subjects_dev / msm_group_import_subjects_dev: one synchronisation per call, range checks on the device), and
    the shard is set up and exchanged in `chunks` pieces (default 2 from four subjects per rank on; MSMHIP_GROUP_CHUNKS): the all-gathers of
    one piece are issued asynchronously (RCCL's own stream) and run over xGMI while the next piece is set up; everything is waited for once,
    before the imports.  gloo (CPU rehearsal): one piece, host tensors through the host entry points.

    pair_layout: the order of the group's pair list (DiscreteGroupCostFunction.set_pair_layout).  None: control-point major whenever the run has a
    process group -- ShardedMove gives every rank a contiguous slice of the list, which is then a region of the sphere (an eighth of every resampled map)
    instead of eight subjects' rows of it (all of their maps and most of everyone else's) -- WHATEVER the number of ranks, one included: the optimiser
    consumes pairs and their costs in list order (the order of its sums, and with near-ties its decisions), so a launched run must not depend on how many
    ranks it was given (until round 5 one rank kept the reference's order and N > 1 did not).  A plain process without a process group keeps
    the group's current layout (the reference's order by default)."""
    c = _comm(comm)
    if pair_layout is None and c.dist is not None and hasattr(group, "set_pair_layout"):
        pair_layout = group.CP_MAJOR
    if pair_layout is not None:
        group.set_pair_layout(pair_layout)
    mine = list(shard(n_subjects, c.rank, c.world))
    # one rank has nobody to exchange with (as include/msmhip_rccl.hpp: `if (c.world() > 1)`): until round 5 a launched one-rank run exported its 64 subjects into
    # torch tensors and all-gathered them to itself, 1.65 GB twice and the set-up in two pieces -- 11 ms of bench.py's 104.5.  MSMHIP_DIST_EXCHANGE=always keeps
    # that path (tests/test_gpu_group.py drives the device-resident exchange over RCCL with the one rank a one-GPU box allows).
    if c.dist is None or (c.world == 1 and os.environ.get("MSMHIP_DIST_EXCHANGE") != "always"):
        group.setup_subjects(mine)
        group.finalize()
        return mine
    import torch

    L, D, V, M = group.L, group.D, group._keep["template"].V, group.N * group.L + 1
    nmax = max(len(shard(n_subjects, r, c.world)) for r in range(c.world))
    device_path = hasattr(group, "export_subjects_dev") and torch.cuda.is_available()
    if not device_path:  # host tensors through the host entry points (CPU tests with a stand-in group)
        group.setup_subjects(mine)
        counts = torch.zeros(nmax, dtype=torch.int64)
        if mine:
            counts[: len(mine)] = torch.tensor([group.subject_index_count(s) for s in mine], dtype=torch.int64)
        all_counts = torch.zeros((c.world, nmax), dtype=torch.int64)
        c.all_gather(all_counts, counts)
        all_counts = all_counts.numpy()
        imax = int(all_counts.max())
        F = torch.zeros((nmax, L, D, V), dtype=torch.float64)
        pp = torch.zeros((nmax, M), dtype=torch.int32)
        pi = torch.zeros((nmax, max(imax, 1)), dtype=torch.int32)
        for k, s in enumerate(mine):
            f, p, i = group.export_subject(s)
            F[k] = torch.from_numpy(f)
            pp[k] = torch.from_numpy(p)
            pi[k, : len(i)] = torch.from_numpy(i)
        aF = torch.empty((c.world,) + tuple(F.shape), dtype=F.dtype)
        app = torch.empty((c.world,) + tuple(pp.shape), dtype=pp.dtype)
        api = torch.empty((c.world,) + tuple(pi.shape), dtype=pi.dtype)
        c.all_gather(aF, F)
        c.all_gather(app, pp)
        c.all_gather(api, pi)
        for r in range(c.world):
            if r == c.rank:
                continue
            for k, s in enumerate(shard(n_subjects, r, c.world)):
                n = int(all_counts[r, k])
                group.import_subject(s, aF[r, k].numpy(), app[r, k].numpy(), api[r, k, :n].numpy())
        group.finalize()
        return mine

    # device buffers that libmsmhip fills and reads; with gloo (the rehearsal of the N > 1 path on a one-GPU box) the collectives themselves go
    # through host copies of them, everything else -- pieces, batched export / import, range checks -- is the nccl path
    dev = c.device if c.on_gpu else "cuda:%d" % torch.cuda.current_device()
    if chunks is None:
        chunks = int(os.environ.get("MSMHIP_GROUP_CHUNKS", "2" if nmax >= 4 else "1"))
    pending = []  # per piece: (k0, k1, all_counts, aF, app, api, [work handles], send buffers kept alive)
    first = True
    for k0, k1 in _chunk_bounds(nmax, chunks):
        part = mine[k0:k1]
        if first:
            group.setup_subjects(part)  # includes the part every rank repeats (estimate_pairs, spacings, rotations)
            first = False
        elif part:
            group.setup_more_subjects(part)
        n = k1 - k0
        counts = torch.zeros(n, dtype=torch.int64)
        if part:
            counts[: len(part)] = torch.tensor([group.subject_index_count(s) for s in part], dtype=torch.int64)
        counts = counts.to(c.device)
        all_counts = torch.zeros((c.world, n), dtype=torch.int64, device=c.device)
        c.all_gather(all_counts, counts)  # small and blocking: sizes the index buffers of this piece
        all_counts = all_counts.cpu().numpy()
        imax = max(int(all_counts.max()), 1)
        F = torch.empty((n, L, D, V), dtype=torch.float64, device=dev)
        pp = torch.zeros((n, M), dtype=torch.int32, device=dev)
        pi = torch.zeros((n, imax), dtype=torch.int32, device=dev)
        if len(part) < n:
            F[len(part):].zero_()  # a rank with fewer subjects than the largest shard sends defined padding
        # torch's fills above run on torch's current stream, which libmsmhip's streams (created non-blocking) do not order against: the library's stream waits
        # for them (msm_ctx_wait_stream: an event, no host wait) BEFORE it writes into the same tensors -- a fill that lands after the export wipes exported
        # rows (seen with four ranks sharing one GPU: one run in seven delivered zeroed patch lists to every rank).  include/msmhip.h: "Stream contract".
        _order_after_torch(group.ctx, torch)
        if part:
            group.export_subjects_dev(part, F.data_ptr(), L * D * V, pp.data_ptr(), M, pi.data_ptr(), imax)  # synchronises libmsmhip's stream
        aF = torch.empty((c.world,) + tuple(F.shape), dtype=F.dtype, device=dev)
        app = torch.empty((c.world,) + tuple(pp.shape), dtype=pp.dtype, device=dev)
        api = torch.empty((c.world,) + tuple(pi.shape), dtype=pi.dtype, device=dev)
        works = []
        for o, i in ((aF, F), (app, pp), (api, pi)):
            if c.on_gpu:
                works.append(c.dist.all_gather_into_tensor(o, i, async_op=True))
            else:
                host = torch.empty(o.shape, dtype=o.dtype)
                c.all_gather(host, i.cpu())
                o.copy_(host)
        pending.append((k0, k1, all_counts, aF, app, api, works, (F, pp, pi)))
    for _, _, _, _, _, _, works, _ in pending:
        for w in works:
            w.wait()
    # work.wait() has made torch's current stream wait for the collectives; libmsmhip's stream is not one torch orders against: it waits for that stream in turn
    # before the imports read the gathered buffers
    _order_after_torch(group.ctx, torch)
    for k0, k1, all_counts, aF, app, api, _, _ in pending:
        imax = api.shape[2]
        for r in range(c.world):
            if r == c.rank:
                continue
            theirs = list(shard(n_subjects, r, c.world))[k0:k1]
            if theirs:
                group.import_subjects_dev(theirs, aF[r].data_ptr(), L * D * V, app[r].data_ptr(), M, api[r].data_ptr(), imax, all_counts[r, : len(theirs)])
    group.finalize()
    return mine


_SEGMENTS = [0]


def same_node(comm):
    """do all ranks of `comm` run on one node (so that they can share host memory)?"""
    c = _comm(comm)
    if c.dist is None or c.world == 1:
        return True
    lw = os.environ.get("LOCAL_WORLD_SIZE")
    if lw is not None:
        return int(lw) == c.world
    names = [None] * c.world
    c.dist.all_gather_object(names, os.uname().nodename)
    return len(set(names)) == 1


def _order_after_torch(ctx, torch):
    """libmsmhip's stream of `ctx` waits for everything torch's current stream holds now (msm_ctx_wait_stream; the caller's side of the stream contract of
    the ..._dev entry points, include/msmhip.h).  No-op without a GPU (gloo rehearsals on the CPU never reach device buffers)."""
    if torch.cuda.is_available():
        ctx.wait_stream(torch.cuda.current_stream().cuda_stream)


class SharedStepBuffer:
    """Result buffers of the ranks of ONE node in a POSIX shared-memory file that every rank maps: `slots` alternating buffers of
    `n` doubles and one progress counter per rank.  The ranks deliver their slices of a step straight into the consumer's address
    space (each GPU writes over its own PCIe link, in parallel) and publish a counter; the consumer waits on the counters -- no
    collective and no copy on the data path.  Flow control: begin(step) on the consumer releases the slot of step - slots (with
    two slots the results of a step stay valid until the next-but-one begin), and holds every producer until then.

    The file is created by rank `dst`, opened by the others after a barrier, and unlinked once everyone has it mapped."""

    HEADER = 4096

    def __init__(self, comm, n, dst=0, slots=2):
        import mmap

        c = self.c = _comm(comm)
        self.n, self.dst, self.slots = int(n), dst, slots
        self.stride = (self.n * 8 + 4095) // 4096 * 4096
        self.nbytes = self.HEADER + slots * self.stride
        _SEGMENTS[0] += 1
        self.path = "/dev/shm/msmhip-%d-%s-%d" % (os.getuid(), os.environ.get("MASTER_PORT", "0"), _SEGMENTS[0])
        if c.rank == dst:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, self.nbytes)
        c.barrier()
        if c.rank != dst:
            fd = os.open(self.path, os.O_RDWR)
        self.mm = mmap.mmap(fd, self.nbytes)
        os.close(fd)
        c.barrier()
        if c.rank == dst:
            os.unlink(self.path)  # stays alive through the mappings; nothing is left behind if a rank dies from here on
        self.counters = np.frombuffer(self.mm, dtype=np.int64, count=c.world, offset=0)
        self.released = np.frombuffer(self.mm, dtype=np.int64, count=1, offset=8 * c.world)  # steps <= released[0] may be overwritten
        self.data = [np.frombuffer(self.mm, dtype=np.float64, count=self.n, offset=self.HEADER + k * self.stride) for k in range(slots)]
        import ctypes

        self.address = ctypes.addressof(ctypes.c_char.from_buffer(self.mm))

    def slot_address(self, k):
        return self.address + self.HEADER + k * self.stride

    def _spin(self, done, what, timeout_s=600.0):
        import time

        t0 = time.monotonic()
        spins = 0
        while not done():
            spins += 1
            if spins > 2000:
                time.sleep(0.00005)
                if time.monotonic() - t0 > timeout_s:
                    raise TimeoutError("SharedStepBuffer: " + what())

    # The counters are written with release and read with acquire semantics (msm_store_release_i64 / msm_load_acquire_i64 of the
    # library: __atomic builtins): a producer's slice is visible before its counter says so, the consumer's reads of the slot cannot
    # move before its look at the counters, and the consumer's release of a slot orders its reads of that slot before a producer's
    # next writes.  (Plain numpy stores would do on x86's total store order only.)
    def _store(self, view, index, value):
        from ._lib import lib

        lib().msm_store_release_i64(view.ctypes.data + 8 * index, int(value))

    def _load(self, view, index=0):
        from ._lib import lib

        return int(lib().msm_load_acquire_i64(view.ctypes.data + 8 * index))

    def begin(self, step):
        """before this rank writes its slice of `step` (1, 2, ...): the slot must be free"""
        if self.c.rank == self.dst:
            self._store(self.released, 0, max(self._load(self.released), step - self.slots))
        self._spin(lambda: self._load(self.released) >= step - self.slots, lambda: "the consumer has not released the slot of step %d" % step)

    def publish(self, step):
        """this rank's slice of `step` (1, 2, ...) is complete in slot (step - 1) % slots"""
        self._store(self.counters, self.c.rank, step)

    def wait(self, step):
        """the consumer: block until every rank has published `step`; returns the slot's array"""
        from ._lib import lib

        L, addr, n = lib(), self.counters.ctypes.data, self.c.world
        self._spin(lambda: int(L.msm_min_acquire_i64(addr, n)) >= step,
                   lambda: "step %d not delivered by ranks %s" % (step, np.nonzero(self.counters < step)[0].tolist()))
        return self.data[(step - 1) % self.slots]


class ShardedMove:
    """One label step of Fusion for the group with the cliques sharded over the ranks (see the module docstring).

    move(labeling, label) evaluates this rank's slice of the 4 P pair costs and 8 T triplet costs and delivers all slices to
    rank `dst`, which gets (pair_quads P x 4, triplet_octets T x 8) exactly as msm_group_fusion_move returns them; the other
    ranks get (None, None).  Transports:
      "local"   one process: the results are copied into pinned arrays of this process (Context.host_array);
      "shm"     the ranks of one node (the 8 GPUs of an MI355X box): every rank's GPU copies its slice over its own PCIe link
                into a shared-memory buffer at the slice's final position (SharedStepBuffer) -- the optimiser is host code, so
                the host is where the costs are needed; no collective, no concatenation.  The arrays returned are views of
                that buffer: valid until the next-but-one move() (the producers are held back until then);
      "gather"  ranks on several nodes: slices padded to the largest, one gather to `dst`; with the nccl backend the kernels write
                into the tensor the gather sends."""

    def __init__(self, group, comm=None, dst=0, transport=None):
        import torch

        self.g, self.c, self.dst = group, _comm(comm), dst
        c = self.c
        self.prange = [shard(group.P, r, c.world) for r in range(c.world)]
        self.trange = [shard(group.T, r, c.world) for r in range(c.world)]
        self.pmax = max(len(r) for r in self.prange)
        self.tmax = max(len(r) for r in self.trange)
        if transport is None:
            transport = "local" if c.world == 1 else ("shm" if same_node(c) and hasattr(group, "ctx") else "gather")
        self.transport = transport
        self.step = 0
        self.out = None
        if transport == "local":
            self.out = (group.ctx.host_array((group.P, 4)), group.ctx.host_array((group.T, 8)))
            return
        if transport == "shm":
            self.off_t = (4 * group.P + 1) // 2 * 2  # doubles; keeps the triplet part 16-byte aligned
            self.shared = SharedStepBuffer(c, self.off_t + 8 * group.T, dst)
            ok = 1
            try:
                group.ctx.register_host(self.shared.address, self.shared.nbytes)
            except Exception:  # the driver cannot pin this mapping: every rank falls back to the gather together
                ok = 0
            if c.dist is not None and c.world > 1:
                flags = [None] * c.world
                c.dist.all_gather_object(flags, ok)
                ok = min(flags)
            if ok:
                return
            if self.shared is not None:
                try:
                    group.ctx.unregister_host(self.shared.address)
                except Exception:
                    pass
            self.shared = None
            self.transport = transport = "gather"
        dev = c.device if c.on_gpu else "cpu"
        self.send = torch.zeros(4 * self.pmax + 8 * self.tmax, dtype=torch.float64, device=dev)
        self.recv = torch.zeros((c.world, 4 * self.pmax + 8 * self.tmax), dtype=torch.float64, device=dev) if (c.rank == dst and c.dist is not None) else None
        self.gpu_scratch = None
        if not c.on_gpu:  # rehearsal: the kernels still need device buffers; results are staged through the host
            self.gpu_scratch = (torch.zeros(4 * self.pmax + 8 * self.tmax, dtype=torch.float64, device="cuda:%d" % torch.cuda.current_device())
                                if torch.cuda.is_available() else None)
        if torch.cuda.is_available():
            # the zero fills of the tensors above run on torch's current stream; libmsmhip's kernels write into them from a stream that does not order against
            # it: that stream waits for them before the first move (a late fill wiped the first results written -- the kept (label, label) costs of a rank's slice)
            _order_after_torch(group.ctx, torch)

    def close(self):
        if self.transport == "shm" and self.shared is not None:
            self.g.ctx.unregister_host(self.shared.address)
            self.shared = None
        if self.transport == "local" and self.out is not None:  # 4 P + 8 T doubles of pinned memory (186 MB at 64 subjects)
            for arr in self.out:
                self.g.ctx.release_host_array(arr)
            self.out = None

    def move(self, labeling, label):
        c, g = self.c, self.g
        pr, tr = self.prange[c.rank], self.trange[c.rank]
        self.step += 1
        if self.transport == "local":
            return g.fusionMove(labeling, label, out=self.out)
        if self.transport == "shm":
            self.shared.begin(self.step)
            base = self.shared.slot_address((self.step - 1) % self.shared.slots)
            g.fusionMove_dev(labeling, label, (pr.start, pr.stop), (tr.start, tr.stop), base + 8 * 4 * pr.start, base + 8 * (self.off_t + 8 * tr.start))
            self.shared.publish(self.step)  # fusionMove_dev returned: this rank's copies have completed
            if c.rank != self.dst:
                return None, None
            buf = self.shared.wait(self.step)
            return buf[: 4 * g.P].reshape(g.P, 4), buf[self.off_t: self.off_t + 8 * g.T].reshape(g.T, 8)
        buf = self.send if c.on_gpu else self.gpu_scratch
        g.fusionMove_dev(labeling, label, (pr.start, pr.stop), (tr.start, tr.stop), buf.data_ptr(), buf.data_ptr() + 8 * 4 * self.pmax)
        if not c.on_gpu:
            self.send.copy_(buf)  # device -> host (gloo gathers host tensors)
        if c.dist is None:
            allbuf = self.send.reshape(1, -1)
        else:
            c.dist.gather(self.send, list(self.recv.unbind(0)) if c.rank == self.dst else None, dst=self.dst)
            if c.rank != self.dst:
                return None, None
            allbuf = self.recv
        host = allbuf.cpu().numpy()
        quads = np.concatenate([host[r, : 4 * len(self.prange[r])] for r in range(c.world)]).reshape(g.P, 4)
        octets = np.concatenate([host[r, 4 * self.pmax: 4 * self.pmax + 8 * len(self.trange[r])] for r in range(c.world)]).reshape(g.T, 8)
        return quads, octets
