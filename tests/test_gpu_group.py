"""Groupwise (gMSM) path on the GPU against the oracle: pairs, patches, inter-subject pairwise costs and
per-subject strain triplets for a small synthetic group (3 subjects, ico4 data / ico2 control grid / ico4 template).

The per-label rigid rotation of the data mesh goes through acos/sincos on the device, so the resampled feature
maps agree to ~1e-12 rather than bit for bit; patch index sets and pair lists are exact."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import synthetic
from oracle import oracle as O

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-11


def build(ctx, S=3, data_order=4, cp_order=2, D=2, mask=False, sim=2, percentile=0.75, subject_orders=None, label_order_offset=2):
    dxyz, dtri = M.make_mesh_from_icosa(data_order)
    cxyz, ctri = M.make_mesh_from_icosa(cp_order)
    txyz, ttri = dxyz, dtri  # template space = a regular sphere at data resolution
    _, mvd = M.cp_spacings(cxyz, ctri)
    samples, _ = M.label_sampling_grid(cp_order + label_order_offset, 0.5 * mvd)
    mk = (np.cos(txyz[:, 0] / 30.0) if mask else None)
    g = M.DiscreteGroupCostFunction(ctx, S, simmeasure=sim, lambda_=0.2, percentile=percentile)
    og = O.Group(S, simmeasure=sim, lambda_=0.2, percentile=percentile)
    tm = M.Mesh(ctx, txyz, ttri)
    g.set_template(tm, mk)
    otm = O.Mesh(txyz, ttri)
    og.set_template(otm, mk)
    g.Initialize(cxyz, ctri)
    og.set_controlgrid(O.Mesh(cxyz, ctri))
    keep = [tm, otm]
    txyz0, ttri0 = dxyz, dtri
    for s in range(S):
        if subject_orders is not None:  # subjects on data meshes of their own (different sizes: the set-up's scratch meshes follow)
            dxyz, dtri = (txyz0, ttri0) if subject_orders[s] == data_order else M.make_mesh_from_icosa(subject_orders[s])
        sph = synthetic.known_warp(dxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)   # this subject's registered sphere so far
        feat = synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=2.0, amp=1.0), D, seed=5)
        regular = M.Mesh(ctx, dxyz, dtri)
        g.reset_meshspace(s, regular, feat)        # first call: _ORIG_MESHES = the regular sphere
        regular.set_coords(sph)
        g.reset_meshspace(s, regular, feat)
        om = O.Mesh(dxyz, dtri)
        og.set_subject(s, om, feat)
        om.set_coords(sph)
        og.set_subject(s, om, feat)
        cp_s = synthetic.known_warp(cxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)
        g.reset_CPgrid(s, cp_s)
        og.reset_cpgrid(s, cp_s)
        keep += [regular, om]
    g.set_labels(samples)
    og.set_labels(samples)
    g.setupCostFunction()
    og.setup()
    return g, og, keep


def test_group_structure_and_patches(ctx):
    g, og, _ = build(ctx)
    assert (g.num_nodes, g.P, g.T) == (og.num_nodes, og.P, og.T) == (3 * 162, 162 * 3, 3 * 320)
    assert np.array_equal(g.getPairs(), og.pairs())
    assert np.array_equal(g.getTriplets(), og.triplets())
    rng = np.random.default_rng(0)
    for s, v, l in zip(rng.integers(0, 3, 25), rng.integers(0, 162, 25), rng.integers(0, g.L, 25)):
        ids, data = g.patch(s, v, l)
        oids, odata = og.patch(s, v, l)
        assert np.array_equal(ids, oids)
        assert np.allclose(data, odata, rtol=1e-10, atol=1e-11)


def test_group_with_68_labels(ctx):
    """A finer sampling grid than the reference's 19 labels (68 per control point): the set-up's label-batched launches, the range test's clusters of 32 centres
    (a control point's labels no longer fit one cluster of 64), patches, pair costs and a label step against the oracle."""
    g, og, _ = build(ctx, S=2, data_order=4, cp_order=2, label_order_offset=3)
    assert g.L == og.L == 68
    rng = np.random.default_rng(5)
    for s, v, l in zip(rng.integers(0, 2, 30), rng.integers(0, 162, 30), rng.integers(0, g.L, 30)):
        ids, data = g.patch(s, v, l)
        oids, odata = og.patch(s, v, l)
        assert np.array_equal(ids, oids)
        assert np.allclose(data, odata, rtol=1e-10, atol=1e-11)
    n = 400
    p, la, lb = rng.integers(0, g.P, n).astype(np.int32), rng.integers(0, g.L, n).astype(np.int32), rng.integers(0, g.L, n).astype(np.int32)
    assert np.allclose(g.computePairwiseCost(p, la, lb), og.pairwise_batch(p, la, lb), rtol=RTOL, atol=ATOL, equal_nan=True)


def test_group_subjects_on_different_data_meshes(ctx):
    """get_patch_data per subject resamples THAT subject's data mesh onto the template: subjects need not share a mesh.  The lanes of
    the set-up keep scratch meshes per topology and rebuild them when the next subject's differs."""
    g, og, _ = build(ctx, subject_orders=[4, 3, 4])
    rng = np.random.default_rng(5)
    for s, v, l in zip(rng.integers(0, 3, 30), rng.integers(0, 162, 30), rng.integers(0, g.L, 30)):
        ids, data = g.patch(s, v, l)
        oids, odata = og.patch(s, v, l)
        assert np.array_equal(ids, oids)
        assert np.allclose(data, odata, rtol=1e-10, atol=1e-11)
    p = rng.integers(0, g.P, 300).astype(np.int32)
    la, lb = rng.integers(0, g.L, 300).astype(np.int32), rng.integers(0, g.L, 300).astype(np.int32)
    got, want = g.computePairwiseCost(p, la, lb), np.array([og.pairwise(*q) for q in zip(p, la, lb)])
    both = np.isnan(want) & np.isnan(got)
    assert np.allclose(got[~both], want[~both], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("lanes", [None, 16, 32])
@pytest.mark.parametrize("mask,sim", [(False, 2), (True, 2), (False, 1)])
def test_group_pairwise_costs(ctx, monkeypatch, mask, sim, lanes):
    # lanes: a quarter or half a wavefront per pair cost (k_group_pairwise<false, 16 | 32>); None: chosen from the group's patch sizes
    if lanes is not None:
        monkeypatch.setenv("MSMHIP_GROUP_PAIR_LANES", str(lanes))  # read when the set-up is finalised
    g, og, _ = build(ctx, mask=mask, sim=sim)
    rng = np.random.default_rng(1)
    p = rng.integers(0, g.P, 400).astype(np.int32)
    la = rng.integers(0, g.L, 400).astype(np.int32)
    lb = rng.integers(0, g.L, 400).astype(np.int32)
    got = g.computePairwiseCost(p, la, lb)
    want = np.array([og.pairwise(*q) for q in zip(p, la, lb)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True), np.nanmax(np.abs(got - want))
    assert np.isfinite(want).sum() > 300


@pytest.mark.parametrize("lanes", [16, 32])
@pytest.mark.parametrize("sim", [2, 1])
def test_group_pairwise_costs_of_patches_beyond_the_membership_bits(ctx, monkeypatch, lanes, sim):
    """An ico5 template under an ico0 control grid: patches of several thousand template vertices -- beyond the 64 rounds x 16 | 32 lanes of
    membership bits k_group_pairwise keeps per query (1025..2048 entries under 16 lanes used to alias bits silently, more than 2048
    was MSM_ERR_CAPACITY).  The reference has no limit (M/DiscreteGroupCostFunction.cpp:54-98 walks a std::map)."""
    monkeypatch.setenv("MSMHIP_GROUP_PAIR_LANES", str(lanes))
    g, og, _ = build(ctx, S=2, data_order=5, cp_order=0, sim=sim)
    sizes = [len(g.patch(s, v, l)[0]) for s in range(2) for v in range(12) for l in (0, 3)]
    assert max(sizes) > 2048 and min(sizes) > 1024, (min(sizes), max(sizes))
    rng = np.random.default_rng(11)
    p = rng.integers(0, g.P, 60).astype(np.int32)
    la, lb = rng.integers(0, g.L, 60).astype(np.int32), rng.integers(0, g.L, 60).astype(np.int32)
    got = g.computePairwiseCost(p, la, lb)
    want = np.array([og.pairwise(*q) for q in zip(p, la, lb)])
    assert np.isfinite(want).sum() > 40  # (two patches without a common template vertex: NaN on both sides)
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True), np.nanmax(np.abs(got - want))


@pytest.mark.parametrize("sim,percentile", [(4, 0.75), (5, 0.75), (4, 0.3)])
def test_group_pairwise_dice(ctx, sim, percentile):
    """DICE / genDICE over the common entries of the two patches (get_sim_for_min, similarities.h:53-56): the costs are
    ratios of counts, so they match the oracle exactly unless a resampled value sits within 1e-12 of a threshold."""
    g, og, _ = build(ctx, sim=sim, percentile=percentile, mask=True)
    rng = np.random.default_rng(2)
    p = rng.integers(0, g.P, 400).astype(np.int32)
    la = rng.integers(0, g.L, 400).astype(np.int32)
    lb = rng.integers(0, g.L, 400).astype(np.int32)
    got = g.computePairwiseCost(p, la, lb)
    want = np.array([og.pairwise(*q) for q in zip(p, la, lb)])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    assert ok.sum() > 300 and np.unique(want[ok]).size > 20
    assert np.array_equal(got[ok], want[ok]), np.abs(got[ok] - want[ok]).max()


def test_group_rejects_bad_similarity(ctx):
    with pytest.raises(M.MsmError, match="Unknown similarity metric"):
        M.DiscreteGroupCostFunction(ctx, 2, simmeasure=3)
    with pytest.raises(M.MsmError, match="Percentile"):
        M.DiscreteGroupCostFunction(ctx, 2, simmeasure=4, percentile=1.0)


@pytest.mark.parametrize("lanes", [16, 32])
def test_group_fusion_move(ctx, monkeypatch, lanes):
    """one label step of Fusion::optimize in one call: the four pair costs and eight triplet costs per clique, in the buffer
    order of Fusion.h:170-173,188-195, equal to the explicit batches (same kernels) and to the oracle; with a quarter and with
    half a wavefront per pair cost"""
    monkeypatch.setenv("MSMHIP_GROUP_PAIR_LANES", str(lanes))
    g, og, _ = build(ctx, mask=True)
    rng = np.random.default_rng(5)
    labeling = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    label = 7
    quads, octets = g.fusionMove(labeling, label)
    pairs, trips = g.getPairs(), g.getTriplets()
    p = np.repeat(np.arange(g.P, dtype=np.int32), 4)
    k = np.tile(np.arange(4), g.P)
    la = np.where(k & 2, label, labeling[pairs[p, 0]]).astype(np.int32)
    lb = np.where(k & 1, label, labeling[pairs[p, 1]]).astype(np.int32)
    assert np.array_equal(quads.ravel(), g.computePairwiseCost(p, la, lb), equal_nan=True)
    t = np.repeat(np.arange(g.T, dtype=np.int32), 8)
    k = np.tile(np.arange(8), g.T)
    lab3 = [np.where(k >> (2 - j) & 1, label, labeling[trips[t, j]]).astype(np.int32) for j in range(3)]
    assert np.array_equal(octets.ravel(), g.computeTripletCost(t, *lab3))
    for i in rng.integers(0, 4 * g.P, 60):
        want = og.pairwise(int(p[i]), int(la[i]), int(lb[i]))
        assert (np.isnan(want) and np.isnan(quads.ravel()[i])) or abs(quads.ravel()[i] - want) <= ATOL + RTOL * abs(want)
    for i in rng.integers(0, 8 * g.T, 60):
        want = og.triplet(int(t[i]), int(lab3[0][i]), int(lab3[1][i]), int(lab3[2][i]))
        assert abs(octets.ravel()[i] - want) <= ATOL + RTOL * abs(want)


def test_group_eight_subjects_ico5(ctx):
    """a larger group (BASELINE config 5 at its middle level: 8 subjects, ico5 data / ico3 control grid): pair list, patch
    lists, inter-subject costs and a whole label step of Fusion against the oracle"""
    g, og, _ = build(ctx, S=8, data_order=5, cp_order=3, D=2)
    assert (g.num_nodes, g.P, g.T) == (8 * 642, 642 * 28, 8 * 1280)
    assert np.array_equal(g.getPairs(), og.pairs())
    rng = np.random.default_rng(21)
    for s, v, l in zip(rng.integers(0, 8, 12), rng.integers(0, 642, 12), rng.integers(0, g.L, 12)):
        ids, data = g.patch(s, v, l)
        oids, odata = og.patch(s, v, l)
        assert np.array_equal(ids, oids) and np.allclose(data, odata, rtol=1e-10, atol=1e-11)
    labeling = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    label = 11
    quads, octets = g.fusionMove(labeling, label)
    pairs, trips = g.getPairs(), g.getTriplets()
    assert np.isfinite(quads).mean() > 0.9
    for p in rng.integers(0, g.P, 300):
        for k in range(4):
            la = label if k & 2 else int(labeling[pairs[p, 0]])
            lb = label if k & 1 else int(labeling[pairs[p, 1]])
            want = og.pairwise(int(p), la, lb)
            got = quads[p, k]
            assert (np.isnan(want) and np.isnan(got)) or abs(got - want) <= ATOL + RTOL * abs(want), (p, k, got, want)
    for t in rng.integers(0, g.T, 200):
        for k in range(8):
            lab3 = [label if k >> (2 - j) & 1 else int(labeling[trips[t, j]]) for j in range(3)]
            want = og.triplet(int(t), *lab3)
            assert abs(octets[t, k] - want) <= ATOL + RTOL * abs(want)


def test_group_triplet_costs(ctx):
    g, og, _ = build(ctx, D=1)
    rng = np.random.default_rng(2)
    t = rng.integers(0, g.T, 800).astype(np.int32)
    la, lb, lc = (rng.integers(0, g.L, 800).astype(np.int32) for _ in range(3))
    got = g.computeTripletCost(t, la, lb, lc)
    want = np.array([og.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL)
    assert (want == 1e7).any()


SHARDED_WORKER = '''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import newmsm_amd as M
from newmsm_amd import dist as D, synthetic

rank, _, world = D.env()
dist = D.init("gloo")          # two processes share the single GPU of the test box; on the 8-GPU node this is "nccl"
ctx = M.Context(0)
S, Dm = 4, 2
dxyz, dtri = M.make_mesh_from_icosa(3)
cxyz, ctri = M.make_mesh_from_icosa(1)
_, mvd = M.cp_spacings(cxyz, ctri)
samples, _ = M.label_sampling_grid(3, 0.5 * mvd)
def make():
    g = M.DiscreteGroupCostFunction(ctx, S, lambda_=0.2)
    tm = M.Mesh(ctx, dxyz, dtri); g.set_template(tm); g.Initialize(cxyz, ctri); keep = [tm]
    for s in range(S):   # every rank registers every subject (small); only the per-label resampling is sharded
        m = M.Mesh(ctx, dxyz, dtri)
        feat = synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=2.0, amp=1.0), Dm, seed=5)
        g.reset_meshspace(s, m, feat)
        m.set_coords(synthetic.known_warp(dxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)); g.reset_meshspace(s, m, feat)
        g.reset_CPgrid(s, synthetic.known_warp(cxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)); keep.append(m)
    g.set_labels(samples)
    return g, keep
g, keep = make()
mine = D.sharded_group_setup(g, S, dist)
rng = np.random.default_rng(3)
p = rng.integers(0, g.P, 300).astype(np.int32); la = rng.integers(0, g.L, 300).astype(np.int32); lb = rng.integers(0, g.L, 300).astype(np.int32)
g1, keep1 = make(); g1.setupCostFunction()
# a launched run's pair list is control-point major whatever the number of ranks (a contiguous slice = a region of the sphere; the optimiser's sums must not
# depend on the rank count): the same pairs as the reference's list in another order -- `pos`: where pair i of the sharded group's list sits in the plain group's
pairs_s, pairs_1 = g.getPairs(), g1.getPairs()
where = {(int(a), int(b)): i for i, (a, b) in enumerate(pairs_1)}
pos = np.array([where[(int(a), int(b))] for a, b in pairs_s], dtype=np.int64)
layout_ok = (not np.array_equal(pos, np.arange(g.P))) and len(set(pos.tolist())) == g.P
sharded = g.computePairwiseCost(p, la, lb)
single = g1.computePairwiseCost(pos[p].astype(np.int32), la, lb)
# a label step with the pair and triplet lists sharded over the two ranks, gathered on rank 0 (M/DiscreteGroupCostFunction.cpp:54-98)
lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
move_ok, transports, move_bad = True, [], []
for transport in (None, "gather"):   # None: the ranks share a node -> every rank's GPU writes its slice into shared host memory
    sm = D.ShardedMove(g, dist, transport=transport)
    transports.append(sm.transport)
    for step, label in enumerate((4, 1, 7)):   # three steps: both alternating buffers of the shared transport are reused
        lab_s = (lab + step) %% g.L
        q, o = sm.move(lab_s, label)
        q1, o1 = g1.fusionMove(lab_s, label)
        if rank == 0:
            same = bool(np.array_equal(q, q1[pos], equal_nan=True) and np.array_equal(o, o1))
            if not same:   # what differs, for the assertion message
                dq = ~((q == q1[pos]) | (np.isnan(q) & np.isnan(q1[pos])))
                move_bad.append((sm.transport, step, int(dq.sum()), np.argwhere(dq)[:4].tolist(), int((o != o1).sum()), float(np.nanmax(np.abs(np.where(dq, q - q1[pos], 0.0))))))
            move_ok = move_ok and same
    sm.close()
# the single-process transport: pinned arrays of this process
q, o = D.ShardedMove(g1, None).move(lab, 4)
q1, o1 = g1.fusionMove(lab, 4)
move_ok = move_ok and bool(np.array_equal(q, q1, equal_nan=True) and np.array_equal(o, o1)) and transports == ["shm", "gather"]
tmpl = D.group_template_update(np.stack([keep[1 + s].get_coords() for s in mine]), None, dist)
dist.barrier()
print(json.dumps({"rank": rank, "mine": mine, "equal": bool(np.array_equal(sharded, single, equal_nan=True)) and layout_ok, "finite": int(np.isfinite(single).sum()), "move_ok": move_ok, "move_bad": move_bad,
                  "template_radius_ok": bool(np.allclose(np.linalg.norm(tmpl["template"], axis=1), 100.0)), "n": tmpl["n_subjects"]}))
dist.destroy_process_group()
'''


def run_sharded_workers(tmp_path, world, S, port):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sharded.py"
    script.write_text((SHARDED_WORKER % root).replace("S, Dm = 4, 2", "S, Dm = %d, 2" % S))
    # MSMHIP_GROUP_CHUNKS=2: each rank sets up and exchanges its subjects in two pieces (msm_group_setup_more_subjects, a gather per piece, the
    # batched device export / import of dist.sharded_group_setup) -- with gloo the gathers go through host copies, the rest is the nccl path
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), MSMHIP_GROUP_CHUNKS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=500)
        assert pr.returncode == 0, se[-3000:]
        outs.append(eval(so.strip().splitlines()[-1].replace("true", "True").replace("false", "False")))
    assert all(o["equal"] and o["move_ok"] and o["template_radius_ok"] and o["n"] == S and o["finite"] > 200 for o in outs), outs
    return outs


def test_sharded_group_two_ranks_match_single_rank(tmp_path):
    outs = run_sharded_workers(tmp_path, 2, 4, 29561)
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == [0, 1, 2, 3]


def test_sharded_group_four_ranks_uneven_shards(tmp_path):
    """four ranks (processes sharing the test box's GPU, gloo between them) over six subjects: shards of 2, 2, 1, 1 -- ranks with fewer subjects than
    the largest shard send padding, the second piece of the short shards is empty --, the pair list control point by control point and cut in four,
    both transports of a label step: every rank's group equals a single rank's, every delivered step bit for bit"""
    outs = run_sharded_workers(tmp_path, 4, 6, 29567)
    assert [o["mine"] for o in sorted(outs, key=lambda o: o["rank"])] == [[0, 1], [2, 3], [4], [5]]


NCCL_WORKER = SHARDED_WORKER.replace('dist = D.init("gloo")', 'dist = D.init("nccl", 0)').replace('for transport in (None, "gather"):', 'for transport in ("gather", "gather"):') \
    .replace('transports == ["shm", "gather"]', 'transports == ["gather", "gather"] and dist.on_gpu')


def test_sharded_group_one_rank_over_rccl(tmp_path):
    """The same worker with the nccl backend (RCCL) and ONE rank: the test box has one GPU, so this is as far as the device-resident
    exchange can be driven here -- subjects exported into torch tensors on the GPU, all_gather_into_tensor of float64 / int32 device
    buffers, the label step written by the kernels into the tensor that dist.gather sends, the template all-reduce on the GPU."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sharded_nccl.py"
    script.write_text(NCCL_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MSMHIP_DIST_EXCHANGE="always")  # (a one-rank run skips the exchange by default: nobody to exchange with)
    pr = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
    assert pr.returncode == 0, pr.stderr[-3000:]
    o = eval(pr.stdout.strip().splitlines()[-1].replace("true", "True").replace("false", "False"))
    assert o["equal"] and o["move_ok"] and o["template_radius_ok"] and o["n"] == 4 and o["finite"] > 200 and o["mine"] == [0, 1, 2, 3], o


EXCHANGE_WORKER = '''
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, %r)
import newmsm_amd as M
from newmsm_amd import dist as D, synthetic

ctx = M.Context(0)
S, Dm = 4, 2
dxyz, dtri = M.make_mesh_from_icosa(3)
cxyz, ctri = M.make_mesh_from_icosa(1)
_, mvd = M.cp_spacings(cxyz, ctri)
samples, _ = M.label_sampling_grid(3, 0.5 * mvd)
def make():
    g = M.DiscreteGroupCostFunction(ctx, S, lambda_=0.2)
    tm = M.Mesh(ctx, dxyz, dtri); g.set_template(tm); g.Initialize(cxyz, ctri); keep = [tm]
    for s in range(S):
        m = M.Mesh(ctx, dxyz, dtri)
        feat = synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=2.0, amp=1.0), Dm, seed=5)
        g.reset_meshspace(s, m, feat)
        m.set_coords(synthetic.known_warp(dxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)); g.reset_meshspace(s, m, feat)
        g.reset_CPgrid(s, synthetic.known_warp(cxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5)); keep.append(m)
    g.set_labels(samples)
    return g, keep
# two "ranks" in one process: each sets up its shard, exports it into device tensors (what the all-gather would carry) and
# imports the other's from them
ranks = [make(), make()]
shards = [[0, 1], [2, 3]]
V, M_ = len(dxyz), len(cxyz) * len(samples) + 1
wire = {}
for (g, _), mine in zip(ranks, shards):
    g.setup_subjects(mine)
    for s in mine:
        n = g.subject_index_count(s)
        F = torch.zeros((g.L, g.D, V), dtype=torch.float64, device="cuda:0")
        pp = torch.zeros(M_, dtype=torch.int32, device="cuda:0")
        pi = torch.zeros(n + 7, dtype=torch.int32, device="cuda:0")   # a padded slot, as in the padded all-gather
        torch.cuda.synchronize()   # torch's fills are complete before the library writes into the tensors from its own (non-blocking) streams
        g.export_subject_dev(s, F.data_ptr(), pp.data_ptr(), pi.data_ptr(), n + 7)
        wire[s] = (F, pp, pi, n)
torch.cuda.synchronize()
if BATCHED:  # the strided send / receive buffers of one all-gather: a shard per call (msm_group_export_subjects_dev / msm_group_import_subjects_dev)
    for (g, _), mine in zip(ranks, shards):
        other = [s for s in range(S) if s not in mine]
        src = ranks[0][0] if other[0] in shards[0] else ranks[1][0]
        imax = max(src.subject_index_count(s) for s in other) + 5
        F = torch.zeros((len(other), g.L, g.D, V), dtype=torch.float64, device="cuda:0")
        pp = torch.zeros((len(other), M_ + 3), dtype=torch.int32, device="cuda:0")    # strides larger than the rows: padded slots
        pi = torch.full((len(other), imax), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        counts = src.export_subjects_dev(other, F.data_ptr(), g.L * g.D * V, pp.data_ptr(), M_ + 3, pi.data_ptr(), imax)
        assert [int(c) for c in counts] == [src.subject_index_count(s) for s in other]
        g.import_subjects_dev(other, F.data_ptr(), g.L * g.D * V, pp.data_ptr(), M_ + 3, pi.data_ptr(), imax, counts)
        bad = pi.clone(); bad[0, 0] = V + 5                                             # a vertex id out of range must be refused, not indexed with
        torch.cuda.synchronize()
        try:
            g.import_subjects_dev(other, F.data_ptr(), g.L * g.D * V, pp.data_ptr(), M_ + 3, bad.data_ptr(), imax, counts)
            raise SystemExit("an out-of-range template vertex id was imported")
        except M.MsmError as e:
            assert "inconsistent" in str(e), str(e)
        g.finalize()
else:
    for (g, _), mine in zip(ranks, shards):
        for s in range(S):
            if s not in mine:
                F, pp, pi, n = wire[s]
                g.import_subject_dev(s, F.data_ptr(), pp.data_ptr(), pi.data_ptr(), n)
        g.finalize()
g1, keep1 = make(); g1.setupCostFunction()
rng = np.random.default_rng(3)
p = rng.integers(0, g1.P, 400).astype(np.int32); la = rng.integers(0, g1.L, 400).astype(np.int32); lb = rng.integers(0, g1.L, 400).astype(np.int32)
single = g1.computePairwiseCost(p, la, lb)
lab = rng.integers(0, g1.L, g1.num_nodes).astype(np.int32)
q1, o1 = g1.fusionMove(lab, 5)
ok = []
for g, _ in ranks:
    q, o = g.fusionMove(lab, 5)
    ok.append(bool(np.array_equal(g.computePairwiseCost(p, la, lb), single, equal_nan=True) and np.array_equal(q, q1, equal_nan=True) and np.array_equal(o, o1)))
    f0, p0, i0 = g.export_subject(3)   # what arrived for (or was computed for) subject 3 equals the full set-up's
    f1, p1, i1 = g1.export_subject(3)
    ok.append(bool(np.array_equal(f0, f1) and np.array_equal(p0, p1) and np.array_equal(i0, i1)))
print(json.dumps({"ok": ok, "finite": int(np.isfinite(single).sum())}))
'''


@pytest.mark.parametrize("batched", [False, True])
def test_device_resident_exchange_between_two_shards(tmp_path, batched):
    """msm_group_export_subject_dev -> device buffers -> msm_group_import_subject_dev, the path an RCCL all-gather takes between two GPUs,
    driven between two shards of one process on the one GPU of the test box: both shards end up equal to the unsharded set-up.
    batched: a whole shard per call through strided buffers (msm_group_export_subjects_dev / msm_group_import_subjects_dev), the row offsets of
    the imported subjects left on the device until msm_group_export_subject asks for them; a corrupted index list is refused."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "exchange.py"
    script.write_text("BATCHED = %s\n" % batched + EXCHANGE_WORKER % root)
    pr = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
    assert pr.returncode == 0, pr.stderr[-3000:]
    o = eval(pr.stdout.strip().splitlines()[-1].replace("true", "True").replace("false", "False"))
    assert o["ok"] == [True] * 4 and o["finite"] > 200, o


def test_group_full_size_properties_64_subjects_ico6(ctx):
    """BASELINE config 5 at full size (64 subjects, ico6 data / ico4 control grid, 19 labels; the oracle takes minutes per subject
    here, so this is checked through size-independent properties): a whole label step (20.7 M pair + 2.6 M triplet costs, processed
    control point by control point in four pieces, delivered into pinned memory behind the kernels) equals the explicit batch
    evaluators on sampled cliques, slices of the step equal the whole, no cost is left unwritten, and a second set-up
    (pipelined, forest build, six lanes) reproduces the first bit for bit."""
    from newmsm_amd import problem

    S = 64
    g, keep = problem.build_group(ctx, S, 6, 4, D=2)
    g.setupCostFunction()
    assert (g.num_nodes, g.P, g.T) == (S * 2562, 2562 * S * (S - 1) // 2, S * 5120)
    rng = np.random.default_rng(9)
    labeling = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    label = 11
    out = (ctx.host_array((g.P, 4)), ctx.host_array((g.T, 8)))
    out[0][:] = -7.0
    out[1][:] = -7.0
    quads, octets = g.fusionMove(labeling, label, out=out)
    assert not (quads == -7.0).any() and not (octets == -7.0).any()
    pairs, trips = g.getPairs(), g.getTriplets()
    p = rng.integers(0, g.P, 20000).astype(np.int32)
    k = rng.integers(0, 4, 20000)
    la = np.where(k & 2, label, labeling[pairs[p, 0]]).astype(np.int32)
    lb = np.where(k & 1, label, labeling[pairs[p, 1]]).astype(np.int32)
    assert np.array_equal(quads[p, k], g.computePairwiseCost(p, la, lb), equal_nan=True)
    t = rng.integers(0, g.T, 20000).astype(np.int32)
    k = rng.integers(0, 8, 20000)
    lab3 = [np.where(k >> (2 - j) & 1, label, labeling[trips[t, j]]).astype(np.int32) for j in range(3)]
    assert np.array_equal(octets[t, k], g.computeTripletCost(t, *lab3))
    finite = np.isfinite(quads)
    assert finite.mean() > 0.99 and (quads[finite] >= 0).all() and (quads[finite] <= 1.0 + 1e-12).all()  # 1 - (1 + r) / 2
    # a rank's slice of the step (pairs and triplets of the middle eighth), delivered into pinned memory as the ranks of a node do
    p0, p1, t0, t1 = 3 * g.P // 8, 4 * g.P // 8, 3 * g.T // 8, 4 * g.T // 8
    part = ctx.host_array((4 * (p1 - p0) + 8 * (t1 - t0),))
    part[:] = -7.0
    g.fusionMove_dev(labeling, label, (p0, p1), (t0, t1), part.ctypes.data, part.ctypes.data + 8 * 4 * (p1 - p0))
    assert np.array_equal(part[: 4 * (p1 - p0)].reshape(-1, 4), quads[p0:p1], equal_nan=True)
    assert np.array_equal(part[4 * (p1 - p0):].reshape(-1, 8), octets[t0:t1])
    # the set-up is deterministic: the second one (buffers allocated, pipeline warm) gives the same step
    first = quads.copy(), octets.copy()
    g.setupCostFunction()
    q2, o2 = g.fusionMove(labeling, label, out=out)
    assert np.array_equal(q2, first[0], equal_nan=True) and np.array_equal(o2, first[1])
    g.close()


def test_group_label_steps_reuse_the_current_current_costs(ctx):
    """A run of label steps as Fusion makes them: the labeling changes at some nodes between steps.  The (current, current) cost of a
    pair is kept from the previous step and reused when neither node changed its label; every step must equal the step evaluated
    from scratch (explicit batches), including the first, a step with an unchanged labeling, one after a new set-up and a sliced
    step in between (which has a cache of its own extent)."""
    g, og, _ = build(ctx, S=4, data_order=4, cp_order=2, D=2)
    rng = np.random.default_rng(17)
    pairs = g.getPairs()
    p = np.repeat(np.arange(g.P, dtype=np.int32), 4)
    k = np.tile(np.arange(4), g.P)

    def from_scratch(labeling, label):
        la = np.where(k & 2, label, labeling[pairs[p, 0]]).astype(np.int32)
        lb = np.where(k & 1, label, labeling[pairs[p, 1]]).astype(np.int32)
        return g.computePairwiseCost(p, la, lb).reshape(g.P, 4)

    labeling = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    for step in range(8):
        label = int(rng.integers(0, g.L))
        quads, _ = g.fusionMove(labeling, label)
        assert np.array_equal(quads, from_scratch(labeling, label), equal_nan=True), step
        if step == 3:
            g.setupCostFunction()  # new patches: nothing kept applies
        if step == 5:  # a slice of a step in between
            import ctypes
            part = ctx.host_array((4 * (g.P // 2),))
            g.fusionMove_dev(labeling, label, (0, g.P // 2), (0, 0), part.ctypes.data, 0)
            assert np.array_equal(part.reshape(-1, 4), quads[: g.P // 2], equal_nan=True)
        if step != 2:  # step 2 -> 3: the same labeling again
            change = rng.random(g.num_nodes) < 0.15
            labeling = np.where(change, rng.integers(0, g.L, g.num_nodes), labeling).astype(np.int32)


def test_group_second_sweep_takes_the_proposed_proposed_costs_from_the_first(ctx):
    """Fusion makes two sweeps over the labels (I/Fusion/Fusion.h:136-138).  The (label, label) cost of a pair depends on the label alone, so
    the second sweep's come from the first; the labeling keeps changing in between.  Every step of both sweeps must equal the step
    evaluated from scratch -- and after a set-up with different patches (a subject's control grid moved) nothing kept may survive."""
    g, og, keep = build(ctx, S=4, data_order=4, cp_order=2, D=2)
    rng = np.random.default_rng(23)
    pairs = g.getPairs()
    p = np.repeat(np.arange(g.P, dtype=np.int32), 4)
    k = np.tile(np.arange(4), g.P)

    def from_scratch(labeling, label):
        la = np.where(k & 2, label, labeling[pairs[p, 0]]).astype(np.int32)
        lb = np.where(k & 1, label, labeling[pairs[p, 1]]).astype(np.int32)
        return g.computePairwiseCost(p, la, lb).reshape(g.P, 4)

    labeling = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
    first = {}
    for sweep in range(2):
        for label in range(g.L):
            quads, _ = g.fusionMove(labeling, label)
            assert np.array_equal(quads, from_scratch(labeling, label), equal_nan=True), (sweep, label)
            if sweep == 0:
                first[label] = quads[:, 3].copy()
            else:
                assert np.array_equal(quads[:, 3], first[label], equal_nan=True)
            change = rng.random(g.num_nodes) < 0.2
            labeling = np.where(change, label, labeling).astype(np.int32)
    # new patches: subject 1's control grid somewhere else
    cxyz, _ = M.make_mesh_from_icosa(2)
    g.reset_CPgrid(1, synthetic.known_warp(cxyz, seed=777, rot_deg=3.0, amp=0.8))
    g.setupCostFunction()
    pairs = g.getPairs()
    changed = 0
    for label in (3, 0, 11):
        quads, _ = g.fusionMove(labeling, label)
        assert np.array_equal(quads, from_scratch(labeling, label), equal_nan=True), label
        changed += int((~np.isclose(quads[:, 3], first[label], equal_nan=True)).sum())
    assert changed > 0  # the set-up did change these costs: a stale table would have been seen
    g.close()


def test_group_level_loop_matches_oracle(ctx):
    """Group_Mesh_registration::run_discrete_opt (M/group_mesh_registration.cpp:70-118) over three subjects: per iteration set-up, two sweeps of
    fusion moves over all labels (4 P inter-subject pair costs + 8 T triplets per step, one call each on the MI355X path), applyLabeling,
    unfold / warp / unfold per subject.  The same caller loop and the same stand-in binary solve drive the oracle: same labelings in every
    step's outcome, registered spheres within the north star's 1e-4 rad."""
    from helpers import OracleOps
    from newmsm_amd import group_registration as GR

    S, D = 3, 2
    dxyz, dtri = M.make_mesh_from_icosa(3)
    feats = np.stack([synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=3.0, amp=2.0), D, seed=5) for s in range(S)])
    sph = np.stack([dxyz for _ in range(S)])
    kw = dict(cp_order=1, iters=2, lambda_=1e-3)  # a weak regulariser: on this coarse grid a label is a large move, and nothing moves at 0.05
    got = GR.run_group_level(GR.ProductGroupOps(ctx), dxyz, dtri, dxyz, dtri, feats, sph, **kw)
    want = GR.run_group_level(OracleOps(M.mcmc_optimise), dxyz, dtri, dxyz, dtri, feats, sph, **kw)
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)
    assert any(l.any() for l in got[3])  # labels were taken: the subjects moved
    assert np.allclose(got[2], want[2], rtol=1e-9, equal_nan=True)
    ua, ub = got[0] / np.linalg.norm(got[0], axis=2, keepdims=True), want[0] / np.linalg.norm(want[0], axis=2, keepdims=True)
    ang = 2.0 * np.arcsin(np.minimum(1.0, 0.5 * np.linalg.norm(ua - ub, axis=2)))
    assert ang.max() <= 1e-4 and np.abs(got[0] - want[0]).max() < 1e-9 and np.abs(got[1] - want[1]).max() < 1e-9
    assert np.abs(got[0] - sph).max() > 1e-3


def test_group_multiresolution_matches_oracle(ctx):
    """Group_Mesh_registration::run_multiresolutions (M/mesh_registration.cpp:30-50 over M/group_mesh_registration.cpp:26-133) without the files: two
    levels over three subjects with a --mask on the template -- per level featurespace::initialise over all subjects (metric_resample, smooth_data,
    variance_normalise), from level 2 on project_CPgrid per subject (the previous level's warp carried to the new data grid and control grid), the
    level's iterations, at the end every subject's input sphere through its final warp.  The same caller loop over the oracle: same labelings in every
    iteration of both levels, energies to 1e-9, registered spheres within the north star's 1e-4 rad.

    The template and the subjects' spheres are irregular (smoothly warped icospheres), a subject's own sphere each; the regular case -- where a label
    carries data vertices exactly onto template vertices and the last bits of the rotation decide the resampled values -- is
    test_host_rotations_on_regular_spheres."""
    from helpers import OracleOps, angles
    from newmsm_amd import group_registration as GR

    S, D = 3, 2
    xyz, tri = M.make_mesh_from_icosa(4)
    txyz = synthetic.known_warp(xyz, seed=33, rot_deg=7.0, amp=1.5)
    meshes = [(synthetic.known_warp(xyz, seed=40 + s, rot_deg=0.0, amp=1.0), tri) for s in range(S)]
    datas = [synthetic.features(synthetic.known_warp(meshes[s][0], seed=90 + s, rot_deg=3.0, amp=2.0), D, seed=5) for s in range(S)]
    levels = [dict(data_order=3, cp_order=1, sg_order=3, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=2.0),
              dict(data_order=4, cp_order=2, sg_order=4, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=0.0)]
    mask = (np.random.default_rng(1).random(len(xyz)) > 0.2).astype(np.float64)
    lg, lw, t = [], [], {}
    kw = dict(mask=mask, varnorm=True, fixnan=True)
    got = GR.run_group_multiresolution(GR.ProductGroupOps(ctx), meshes, datas, txyz, tri, levels, labelings_out=lg, timings=t, **kw)
    want = GR.run_group_multiresolution(OracleOps(M.mcmc_optimise), meshes, datas, txyz, tri, levels, labelings_out=lw, **kw)
    assert len(lg) == len(lw) == 4
    for a, b in zip(lg, lw):
        assert np.array_equal(a, b)
    assert any(l.any() for l in lg[:2]) and any(l.any() for l in lg[2:])  # labels were taken at both levels
    for a, b in zip(got[2], want[2]):
        assert np.allclose(a, b, rtol=1e-9)
    for s in range(S):
        assert angles(got[0][s], want[0][s]).max() <= 1e-4 and np.abs(got[0][s] - want[0][s]).max() < 1e-8
        assert angles(got[0][s], meshes[s][0]).max() > 1e-3
    for a, b in zip(got[1], want[1]):
        assert np.abs(a - b).max() < 1e-8
    assert got[1][0].shape == (S, 642, 3) and got[1][1].shape == (S, 2562, 3)
    assert {"metric_resample", "smooth_data", "sphere_project_warp", "unfold", "setup", "fusion_moves"} <= set(t)


def test_pair_layout_control_point_major(ctx):
    """msm_group_set_pair_layout: the pair list control point by control point (the layout of runs that shard the list over ranks) holds the pairs of
    estimate_pairs (M/DiscreteGroupModel.cpp:37-55) in another order -- every control point's S (S - 1) / 2 subject pairs together --, and every
    evaluator addressed through it gives bit for bit what the reference's order gives for the same pair: single costs, whole label steps (first
    visit, a changed labeling, second visit: the kept-cost caches follow the layout), slices of a step; back to the reference's order on request."""
    g, og, keep = build(ctx, S=4, data_order=4, cp_order=2)
    ref_pairs = g.getPairs().copy()
    assert np.array_equal(ref_pairs, og.pairs())
    rng = np.random.default_rng(11)
    labs = [rng.integers(0, g.L, g.num_nodes).astype(np.int32)]
    labs.append(np.where(rng.random(g.num_nodes) < 0.2, rng.integers(0, g.L, g.num_nodes), labs[0]).astype(np.int32))
    steps = [(labs[0], 3), (labs[1], 5), (labs[1], 3)]
    want = [tuple(np.array(x) for x in g.fusionMove(l, k)) for l, k in steps]
    g.set_pair_layout(g.CP_MAJOR)
    g.setupCostFunction()
    pairs = g.getPairs()
    where = {(int(a), int(b)): i for i, (a, b) in enumerate(ref_pairs)}
    pos = np.array([where[(int(a), int(b))] for a, b in pairs])
    assert sorted(pos.tolist()) == list(range(g.P)) and not np.array_equal(pos, np.arange(g.P))
    per_cp = 4 * 3 // 2  # the subject pairs of one control point are neighbours in the list
    first = pairs[:, 0] % g.N
    assert all(len(set(first[i:i + per_cp].tolist())) == 1 for i in range(0, g.P, per_cp)) and len(set(first[::per_cp].tolist())) == g.N
    for (l, k), (q0, o0) in zip(steps, want):
        q, o = g.fusionMove(l, k)
        assert np.array_equal(q, q0[pos], equal_nan=True) and np.array_equal(o, o0)
    p = rng.integers(0, g.P, 200).astype(np.int32)
    la, lb = rng.integers(0, g.L, 200).astype(np.int32), rng.integers(0, g.L, 200).astype(np.int32)
    got = g.computePairwiseCost(p, la, lb)
    assert np.allclose(got, [og.pairwise(int(pos[i]), int(a), int(b)) for i, a, b in zip(p, la, lb)], rtol=RTOL, atol=ATOL, equal_nan=True)
    # a slice of the list = a range of control points: the second half of a step, delivered into pinned host memory at the slice's position
    half = (g.P // 2, g.P)
    out_q, out_t = ctx.host_array((g.P, 4)), ctx.host_array((g.T, 8))
    out_q[:] = -1.0
    g.fusionMove_dev(labs[1], 5, half, (0, g.T), out_q.ctypes.data + 32 * half[0], out_t.ctypes.data)
    assert np.array_equal(out_q[half[0]:], want[1][0][pos][half[0]:], equal_nan=True) and (out_q[: half[0]] == -1.0).all() and np.array_equal(out_t, want[1][1])
    g.set_pair_layout(g.REFERENCE_ORDER)
    g.setupCostFunction()
    assert np.array_equal(g.getPairs(), ref_pairs)
    q, o = g.fusionMove(*steps[0])
    assert np.array_equal(q, want[0][0], equal_nan=True) and np.array_equal(o, want[0][1])
    with pytest.raises(M.MsmError):
        g.set_pair_layout(2)


def test_host_rotations_on_regular_spheres(ctx):
    """msm_group_set_rotation_mode(1), the default: the data vertices' rotation matrices from the host's libm, as the reference computes them (R/point.cpp:97-152 called from
    M/DiscreteGroupModel.cpp:97-105), applied on the device.  On REGULAR icospheres -- the template and every subject's sphere, as gMSM's own scripts set a run up --
    a label carries data vertices exactly onto template vertices, and the last bits of the rotation decide which triangle around such a vertex "contains" it (the
    device's acos / sincos: a few resampled values differ from the reference's by up to 1e-2).  With the host's matrices every resampled
    value of every (subject, control point, label) patch is the oracle's, and the two-level registration of test_group_multiresolution_matches_oracle on regular
    spheres takes the oracle's decisions in every label step."""
    from helpers import OracleOps, angles
    from newmsm_amd import group_registration as GR

    g, og, keep = build(ctx, S=2, data_order=3, cp_order=1)
    worst_device = 0.0
    for mode in (g.DEVICE_ROTATIONS, g.HOST_ROTATIONS):
        g.set_rotation_mode(mode)
        g.setupCostFunction()
        worst = 0.0
        for s in range(2):
            for v in range(g.N):
                for l in range(g.L):
                    ids, data = g.patch(s, v, l)
                    oids, odata = og.patch(s, v, l)
                    assert np.array_equal(ids, oids)
                    if len(ids):
                        worst = max(worst, float(np.abs(data - odata).max()))
        if mode == g.DEVICE_ROTATIONS:
            worst_device = worst
        else:
            assert worst < 1e-12, worst
    # (build() warps the subjects' spheres: the device's rotations agree to 1e-9 there; the regular case follows)
    assert worst_device < 1e-9
    S, D = 3, 2
    xyz, tri = M.make_mesh_from_icosa(4)
    datas = [synthetic.features(synthetic.known_warp(xyz, seed=90 + s, rot_deg=3.0, amp=2.0), D, seed=5) for s in range(S)]
    levels = [dict(data_order=3, cp_order=1, sg_order=3, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=2.0),
              dict(data_order=4, cp_order=2, sg_order=4, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=0.0)]
    mask = (np.random.default_rng(1).random(len(xyz)) > 0.2).astype(np.float64)
    lg, lw = [], []
    kw = dict(mask=mask, varnorm=True, fixnan=True)
    got = GR.run_group_multiresolution(GR.ProductGroupOps(ctx), [(xyz, tri)] * S, datas, xyz, tri, levels, labelings_out=lg, **kw)
    want = GR.run_group_multiresolution(OracleOps(M.mcmc_optimise), [(xyz, tri)] * S, datas, xyz, tri, levels, labelings_out=lw, **kw)
    assert len(lg) == len(lw) == 4 and all(np.array_equal(a, b) for a, b in zip(lg, lw)) and any(l.any() for l in lg[2:])
    for a, b in zip(got[2], want[2]):
        assert np.allclose(a, b, rtol=1e-9)
    for s in range(S):
        assert angles(got[0][s], want[0][s]).max() <= 1e-4 and np.abs(got[0][s] - want[0][s]).max() < 1e-8


@pytest.mark.gpu
def test_patch_lists_do_not_depend_on_how_the_range_test_finds_its_candidates():
    """The range test of gMSM's patch lists (get_patch_data, M/DiscreteGroupModel.cpp:109-117) draws its candidates from a grid over the template, sorted by id per
    control point (round 5); what does not fit takes the sweep over 64-id chunks, and MSMHIP_RANGE_CLUSTER=off the kernel that takes every centre on its own.  The
    three must give the same rows in the same order: one digest over the exported maps, row offsets and index lists (ico5 / ico3: 12 k centres per subject; ico5 / ico1:
    patches of a thousand entries, beyond the grid path's candidate list)."""
    import os
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the last two: 61 and 217 labels per control point -- clusters of up to 64 centres, and a control point's centres in clusters of 32)
    for sizes in (("4", "5", "3"), ("2", "5", "1"), ("2", "4", "2", "3"), ("2", "3", "1", "4")):
        seen = set()
        for env in ({}, {"MSMHIP_RANGE_GRID": "off"}, {"MSMHIP_RANGE_CLUSTER": "off"}):
            r = subprocess.run([_sys.executable, os.path.join(root, "tools", "group_patch_digest.py"), *sizes], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            line = [x for x in r.stdout.splitlines() if x.startswith("digest ")]
            assert line, r.stdout
            seen.add(line[0])
        assert len(seen) == 1, seen
