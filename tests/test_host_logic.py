"""libmsmhip's [host] entry points against the oracle, bit for bit (no GPU needed)."""
import numpy as np
import pytest

import newmsm_amd as M
from oracle import oracle as O


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5, 6])
def test_icosphere_matches_oracle(built, order):
    a = M.make_mesh_from_icosa(order)
    b = O.icosphere(order)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_unit_icosphere(built):
    a = M.make_mesh_from_icosa(3, radius=None)
    b = O.icosphere(3, radius=None)
    assert np.array_equal(a[0], b[0])


@pytest.mark.parametrize("order", [2, 4])
def test_adjacency_areas_spacings(built, order):
    xyz, tri = M.make_mesh_from_icosa(order)
    om = O.Mesh(xyz, tri)
    for p, q in zip(M.mesh_adjacency(tri, len(xyz)), om.adjacency()):
        assert np.array_equal(p, q)
    assert np.array_equal(M.vertex_areas(xyz, tri), om.vertex_areas())
    ms, mvd = M.cp_spacings(xyz, tri)
    oms, omvd = O.cp_spacings(om)
    assert np.array_equal(ms, oms) and mvd == omvd
    assert np.array_equal(M.estimate_triplets(tri), O.estimate_triplets(om))
    assert np.array_equal(M.estimate_pairs(tri, len(xyz)), O.estimate_pairs(om))


@pytest.mark.parametrize("cp_order,sg_order", [(2, 4), (3, 5), (4, 6)])
def test_label_grid(built, cp_order, sg_order):
    xyz, tri = M.make_mesh_from_icosa(cp_order)
    _, mvd = M.cp_spacings(xyz, tri)
    sg = O.Mesh(*O.icosphere(sg_order))
    for abs_is_int in (False, True):
        s, b = M.label_sampling_grid(sg_order, 0.5 * mvd, abs_is_int=abs_is_int)
        _, os_, ob = O.label_sampling_grid(sg, 0.5 * mvd, abs_is_int=abs_is_int)
        assert np.array_equal(s, os_) and np.array_equal(b, ob)
    scale = 1.0
    oscale = 1.0
    for _ in range(9):  # crosses the 0.25 reset
        l, scale = M.rescale_sampling_grid(s, scale)
        ol, oscale = O.rescale_sampling_grid(os_, oscale)
        assert np.array_equal(l, ol) and scale == oscale
    assert np.array_equal(M.cp_rotations(s[0], xyz), O.cp_rotations(os_[0], xyz))


def test_rotation_matrix_special_cases(built):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, b = rng.normal(size=3), rng.normal(size=3)
        assert np.array_equal(M.estimate_rotation_matrix(a, b), O.rotation_matrix(a, b))
    a = np.array([0.3, -0.2, 0.9])
    assert np.array_equal(M.estimate_rotation_matrix(a, 3 * a), np.eye(3))          # identity branch
    assert np.array_equal(M.estimate_rotation_matrix(a, -a), O.rotation_matrix(a, -a))  # antipodal branch


@pytest.mark.parametrize("order,shape", [(3, "regular"), (5, "regular"), (5, "warped"), (5, "jittered"), (6, "regular")])
def test_top_down_octree_equals_incremental_insertion(built, monkeypatch, order, shape):
    # the library builds the reference's tree top down (a node is decided by one scan of the triangles overlapping it, in
    # id order) and on worker threads; the literal one-by-one insertion (R/octree.cpp:42-141) must give the same leaves
    import newmsm_amd as M
    from newmsm_amd import synthetic

    xyz, tri = M.make_mesh_from_icosa(order)
    if shape == "warped":
        xyz = synthetic.known_warp(xyz, seed=3, rot_deg=4.0, amp=3.0)
    elif shape == "jittered":
        rng = np.random.default_rng(4)
        xyz = xyz + rng.normal(scale=0.8, size=xyz.shape)
    fast = M.octree_signature(xyz, tri)
    monkeypatch.setenv("MSMHIP_HOST_THREADS", "1")
    serial = M.octree_signature(xyz, tri)
    monkeypatch.setenv("MSMHIP_INCREMENTAL_OCTREE", "1")
    literal = M.octree_signature(xyz, tri)
    assert fast == serial == literal
    if shape == "regular" and order == 6:
        assert fast[0] == dict(nodes=14281, leaves=12496, depth=6, refs=176096, max_leaf=49)  # SURVEY.md section 8 [probe]


@pytest.mark.parametrize("order,shape", [(3, "regular"), (4, "regular"), (5, "warped"), (5, "radial"), (6, "regular"), (4, "jittered")])
def test_direction_table_accepts_only_the_reference_answer(built, order, shape):
    # The table the cost kernels search simple-surface targets with (csrc/octree.cpp: build_ray_table), against the first pass of
    # Octree::get_closest_triangle (R/octree.cpp:156-178) on the host: at random directions and at points placed 1e-10 .. 1e-3 of an edge
    # away from random edges and vertices, whatever a kernel may accept -- the float edge-plane test, or the FP64 re-test of a nearly
    # accepted candidate, and ray_vouches; the very functions the kernels call -- must be the one triangle listed in the point's leaf
    # that passes the inside test.  Every exclusion box is checked too: a point in the leaf it stands for must be refused.
    import newmsm_amd as M
    from newmsm_amd import api, synthetic

    xyz, tri = M.make_mesh_from_icosa(order)
    if shape == "warped":
        xyz = synthetic.known_warp(xyz, seed=3, rot_deg=5.0, amp=2.0)
    elif shape == "radial":  # off the sphere by up to 1e-3: still one triangle per ray
        rng = np.random.default_rng(6)
        xyz = xyz * (1.0 + 1e-5 * rng.normal(size=(len(xyz), 1)))
    elif shape == "jittered":  # folds: not a simple surface, no table
        rng = np.random.default_rng(4)
        xyz = xyz + rng.normal(scale=2.0, size=xyz.shape)
    rep = api.ray_table_check(xyz, tri, 60000, seed=order)
    if shape == "jittered":
        assert not rep["simple"] and rep["points"] == 0
        return
    assert rep["simple"] and rep["points"] == 60000
    assert rep["violations"] == 0, rep
    assert rep["by_float"] > 20000 and rep["by_fp64"] > 500 and rep["open"] > 0   # all three outcomes occur
    assert rep["boxes_checked"] >= rep["with_exclusions"] > 0
    if shape == "regular":
        assert rep["unusable"] == 0   # up to seven exclusion boxes per triangle: with three, 24 / 72 triangles of ico4 / ico6 were left out


def test_variance_normalise_matches_oracle(built):
    """variance_normalise (M/reg_tools.cpp:804-843): the serial mean / variance recurrence, with and without exclusion"""
    rng = np.random.default_rng(5)
    data = rng.normal(2.0, 3.0, size=(3, 642))
    data[2] = 7.0  # zero variance: centred, not scaled
    excl = (rng.random(642) > 0.25).astype(float)
    for e in (None, excl):
        got, want = M.variance_normalise(data, e), O.variance_normalise(data, e)
        assert np.array_equal(got, want)
        keep = np.ones(642, bool) if e is None else e > 0
        assert np.allclose(got[:2, keep].mean(axis=1), 0, atol=1e-12) and np.allclose(got[:2, keep].std(axis=1, ddof=1), 1)
        assert np.array_equal(got[2, keep], np.zeros(keep.sum())) and np.array_equal(got[:, ~keep], data[:, ~keep])


def mcmc_problem(N, L, triplets, seed):
    rng = np.random.default_rng(seed)
    return rng.random((L, N)), rng.random((len(triplets), L, L, L)), np.asarray(triplets, dtype=np.int32)


def mcmc_energy(U, tc, tr, lab):
    e = sum(tc[t, lab[a], lab[b], lab[c]] + (U[lab[a], a] + U[lab[b], b] + U[lab[c], c]) / 3.0 for t, (a, b, c) in enumerate(tr))
    return float(e)


def test_mcmc_optimise_properties(built):
    """MCMC::optimise (M/mcmc_opt.h:31-134): each visit keeps the cheapest of the eight label combinations of one triplet, so
    with disjoint triplets the energy over the triplets never rises; the run is a pure function of the seed."""
    tr = np.arange(30, dtype=np.int32).reshape(10, 3)
    U, tc, tr = mcmc_problem(30, 7, tr, seed=1)
    lab0 = np.zeros(30, dtype=np.int32)
    assert np.array_equal(M.mcmc_optimise(U, tc, tr, lab0, iters=0), lab0)
    prev, energies = lab0, [mcmc_energy(U, tc, tr, lab0)]
    for sweeps in (1, 3, 10, 200):
        lab = M.mcmc_optimise(U, tc, tr, lab0, mcparam=0.3, iters=sweeps, seed=11)
        assert lab.min() >= 0 and lab.max() < 7
        energies.append(mcmc_energy(U, tc, tr, lab))
        prev = lab
    assert all(b <= a + 1e-15 for a, b in zip(energies, energies[1:])) and energies[-1] < energies[0]
    assert np.array_equal(prev, M.mcmc_optimise(U, tc, tr, lab0, mcparam=0.3, iters=200, seed=11))
    # one triplet, two labels: the single visit that proposes label 1 must pick the arg-min of the 8 combinations (first on ties)
    U1, tc1, tr1 = mcmc_problem(3, 2, [[0, 1, 2]], seed=3)
    combos = [tc1[0, a, b, c] + (U1[a, 0] + U1[b, 1] + U1[c, 2]) / 3.0 for a in (0, 1) for b in (0, 1) for c in (0, 1)]
    best = int(np.argmin(combos))
    lab = M.mcmc_optimise(U1, tc1, tr1, np.zeros(3, np.int32), mcparam=0.5, iters=50, seed=0)
    assert list(lab) == [best >> 2 & 1, best >> 1 & 1, best & 1]
    with pytest.raises(M.MsmError):
        M.mcmc_optimise(U, tc, tr, np.full(30, 7, np.int32))


def test_adjacency_flat_build_matches_the_literal_lists_on_irregular_and_degenerate_input():
    """msm_mesh_adjacency (flat counting-sort build) against Mesh::initialize's literal push_back logic: shuffled triangle order and
    triangles that list a vertex twice or three times (the literal code then records the triangle, and the vertex as its own
    neighbour, as often as it is listed)."""
    def literal(tri, V):
        nb, tr = [[] for _ in range(V)], [[] for _ in range(V)]
        for t, n in enumerate(tri):
            for k in range(3):
                tr[n[k]].append(t)
            for a, b in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
                if n[b] not in nb[n[a]]:
                    nb[n[a]].append(n[b])
        return nb, tr

    rng = np.random.default_rng(1)
    for order in (1, 2, 3):
        xyz, tri = M.make_mesh_from_icosa(order)
        V = len(xyz)
        extra = rng.integers(0, V, (5, 3))
        extra[0, 1] = extra[0, 0]
        extra[1, 2] = extra[1, 0]
        extra[2, :] = extra[2, 0]
        tri2 = np.vstack([tri[rng.permutation(len(tri))], extra]).astype(np.int32)
        nbr_ptr, nbr, tid_ptr, tid = M.mesh_adjacency(tri2, V)
        nb, tr = literal(tri2, V)
        for v in range(V):
            assert list(nbr[nbr_ptr[v]:nbr_ptr[v + 1]]) == nb[v]
            assert list(tid[tid_ptr[v]:tid_ptr[v + 1]]) == tr[v]


def test_fusion_icm_step_against_a_plain_python_version(built):
    """msm_fusion_icm_step (the stand-in binary solve that lets tests and tools drive the fusion-move path end to end) against the
    same rule written out in Python: ascending node order, strict improvement, passes until nothing changes"""
    rng = np.random.default_rng(12)
    for trial in range(8):
        N = int(rng.integers(5, 60))
        T = int(rng.integers(0, 3 * N))
        P = int(rng.integers(0, 3 * N)) if trial % 2 else 0
        tr = np.sort(np.stack([rng.choice(N, 3, replace=False) for _ in range(T)]).reshape(T, 3), axis=1).astype(np.int32) if T else np.zeros((0, 3), np.int32)
        pr = np.stack([rng.choice(N, 2, replace=False) for _ in range(P)]).reshape(P, 2).astype(np.int32) if P else np.zeros((0, 2), np.int32)
        u2 = rng.normal(size=(N, 2)) if trial != 5 else None
        oc = rng.normal(size=(T, 8))
        qd = rng.normal(size=(P, 4))
        if trial == 3 and T:
            oc[rng.integers(0, T), :] = np.nan  # a failed evaluation: comparisons with NaN never flip a node
        passes = int(rng.integers(1, 6))
        got = M.fusion_icm_step(u2 if u2 is not None else N, oc, tr, passes, quads=qd, pairs=pr)
        x = np.zeros(N, dtype=np.int32)
        for _ in range(passes):
            changed = False
            for v in range(N):
                e = [u2[v, 0], u2[v, 1]] if u2 is not None else [0.0, 0.0]
                for p in range(P):
                    if v in pr[p]:
                        for val in (0, 1):
                            xa = val if pr[p, 0] == v else x[pr[p, 0]]
                            xb = val if pr[p, 1] == v else x[pr[p, 1]]
                            e[val] += qd[p, 2 * int(xa) + int(xb)]
                for t in range(T):
                    if v in tr[t]:
                        for val in (0, 1):
                            bits = 0
                            for q in range(3):
                                xv = val if tr[t, q] == v else x[tr[t, q]]
                                bits |= int(xv) << (2 - q)
                            e[val] += oc[t, bits]
                if e[1 - x[v]] < e[x[v]]:
                    x[v] = 1 - x[v]
                    changed = True
            if not changed:
                break
        assert np.array_equal(got, x), trial
