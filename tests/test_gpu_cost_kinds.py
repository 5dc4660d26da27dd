"""The other cost-function classes and the clique costs on the GPU against the oracle.

Tolerances: these costs go through acos/asin/sincos/pow (device libm vs glibc) and, for the strain energy, a
2x2 inverse and 3x3 determinant that the oracle writes out with cofactors; rtol 1e-9 / atol 1e-11."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem
from tests.helpers import oracle_cost

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-11


def pair(ctx, inp, kind, **kw):
    cf, keep = problem.build_cost(ctx, inp, kind=kind, **kw)
    cf.get_source_data()
    oc = oracle_cost(inp, kind, **kw)
    oc.get_source_data()
    return cf, oc, keep


@pytest.mark.parametrize("kind", ["multivariate", "patchwise"])
@pytest.mark.parametrize("sim", [2, 1])
def test_unary_table_feature_kinds(ctx, kind, sim):
    inp = problem.pairwise_inputs(5, 3, D=4)
    cf, oc, _ = pair(ctx, inp, kind, simmeasure=sim)
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert np.isfinite(U).all()
    assert np.allclose(U, Uo, rtol=RTOL, atol=ATOL), np.max(np.abs(U - Uo))


def test_multivariate_with_per_dimension_weights(ctx):
    inp = problem.pairwise_inputs(4, 2, D=3)
    rng = np.random.default_rng(4)
    w = rng.uniform(0.1, 1.0, size=(3, len(inp["source_xyz"])))
    cf, _ = problem.build_cost(ctx, inp, kind="multivariate")
    cf.set_dataaffintyweighting(w)
    cf.get_source_data()
    oc = oracle_cost(inp, "multivariate")
    oc.set_cfweight(w)
    oc.get_source_data()
    assert np.array_equal(cf.absolute_weights(), oc.absolute_weights())
    assert np.allclose(cf.computeUnaryCosts(), oc.unary_table(), rtol=RTOL, atol=ATOL)


def test_multivariate_d32_full_size_spot_check(ctx):
    # BASELINE config 3 shape (MSMAll-like, 32 feature dimensions) at ico6 / ico4, checked on a sample of evaluations
    inp = problem.pairwise_inputs(6, 4, D=32)
    cf, oc, _ = pair(ctx, inp, "multivariate")
    U = cf.computeUnaryCosts()
    assert U.shape == (19, 2562) and np.isfinite(U).all()
    rng = np.random.default_rng(2)
    for n, l in zip(rng.integers(0, 2562, 12), rng.integers(0, 19, 12)):
        assert abs(U[l, n] - oc.unary(n, l)) <= ATOL + RTOL * abs(U[l, n])


def random_queries(rng, n, count, L, k):
    return [rng.integers(0, count, n).astype(np.int32)] + [rng.integers(0, L, n).astype(np.int32) for _ in range(k)]


def test_triplet_strain_costs(ctx):
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.2, mu=0.1, kappa=10.0, k_exp=2.0, rexp=2.0)
    rng = np.random.default_rng(0)
    t, la, lb, lc = random_queries(rng, 3000, cf.T, cf.L, 3)
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))
    assert (want >= 1e6).sum() > 0, "the sample should include folded moves"
    assert np.array_equal(got >= 1e6, want >= 1e6)
    # the zero labelling costs exactly the strain of the current grid; label 0 is the centre
    lab0 = np.zeros(cf.N, dtype=np.int32)
    E = cf.tripletOctets(lab0, 0)
    assert np.allclose(E, E[:, :1], rtol=0, atol=0)  # all eight combinations coincide when label == current


def test_triplet_octets_match_fusion_order(ctx):
    inp = problem.pairwise_inputs(4, 2, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, rexp=1.0, k_exp=1.5)
    rng = np.random.default_rng(1)
    labeling = rng.integers(0, cf.L, cf.N).astype(np.int32)
    label = 5
    E = cf.tripletOctets(labeling, label)
    trip = inp["triplets"]
    for t in rng.integers(0, cf.T, 60):
        a, b, c = labeling[trip[t]]
        combos = [(a, b, c), (a, b, label), (a, label, c), (a, label, label), (label, b, c), (label, b, label), (label, label, c), (label, label, label)]
        want = [oc.triplet(t, *q) for q in combos]  # order 000..111 of I/Fusion/Fusion.h:188-195
        assert np.allclose(E[t], want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 3)])
def test_triclique_likelihood(ctx, kind, D):
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, oc, _ = pair(ctx, inp, kind, rmode=3, lambda_=0.1)
    ptr, idx = cf.patches()
    optr, oidx = oc.patches()
    assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx)  # bins per control-grid triangle
    assert np.array_equal(cf.absolute_weights(), oc.absolute_weights())
    assert np.all(cf.computeUnaryCosts() == 0.0)  # the HO classes' computeUnaryCost returns 0
    rng = np.random.default_rng(3)
    t, la, lb, lc = random_queries(rng, 1500, cf.T, cf.L, 3)
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))


def test_pairwise_costs(ctx):
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=1, lambda_=0.3, rexp=2.0)
    rng = np.random.default_rng(5)
    p, la, lb = random_queries(rng, 4000, cf.P, cf.L, 2)
    got = cf.computePairwiseCost(p, la, lb)
    want = np.array([oc.pairwise(*q) for q in zip(p, la, lb)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))
    same = la == lb  # identical labels on both ends: equal rotations are not guaranteed, but label 0 (no move) is free
    zero = cf.computePairwiseCost(p[:50], np.zeros(50, np.int32), np.zeros(50, np.int32))
    assert np.all(zero == np.array([oc.pairwise(q, 0, 0) for q in p[:50]]))
    # full table layout: paircosts[(pair*L + labelB)*L + labelA]
    tab = cf.computePairwiseCosts().reshape(cf.P, cf.L, cf.L)
    assert np.allclose(tab[p, lb, la], got, rtol=0, atol=0)


def test_pairwise_rexp_one_and_folding(ctx):
    inp = problem.pairwise_inputs(4, 2, D=1, labeldist=1.5)  # long moves: some fold the neighbouring triangles
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=1, lambda_=1.0, rexp=1.0)
    rng = np.random.default_rng(6)
    p, la, lb = random_queries(rng, 3000, cf.P, cf.L, 2)
    got = cf.computePairwiseCost(p, la, lb)
    want = np.array([oc.pairwise(*q) for q in zip(p, la, lb)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL)
    assert (want == 1e7).any()


@pytest.mark.parametrize("rmode", [1, 3])
def test_total_cost_sum(ctx, rmode):
    inp = problem.pairwise_inputs(4, 2, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=rmode)
    if rmode == 1:
        cf.setPairs(inp["pairs"])
        oc.set_triplets(np.zeros((0, 3), dtype=np.int32))
    else:
        oc.set_pairs(np.zeros((0, 2), dtype=np.int32))
    rng = np.random.default_rng(7)
    labeling = rng.integers(0, cf.L, cf.N).astype(np.int32)
    tot, parts = cf.evaluateTotalCostSum(labeling)
    otot, oparts = oc.total(labeling)
    assert np.allclose(parts, oparts, rtol=RTOL, atol=ATOL) and abs(tot - otot) <= ATOL + RTOL * abs(otot)


@pytest.mark.parametrize("kind,D,rmode", [("univariate", 1, 5), ("ho_univariate", 1, 4)])
def test_anatomical_strain_regoption5(ctx, kind, D, rmode):
    # aMSM: computeTripletCost :169-182 with deform_anatomy :255-301 (one search on the anatomical sphere per face vertex)
    import oracle.oracle as O

    inp = problem.pairwise_inputs(4, 2, D=D)
    an = problem.anatomical_inputs(ctx, inp)
    cf, oc, keep = pair(ctx, inp, kind, rmode=rmode, lambda_=0.05, mu=0.4, kappa=1.6, k_exp=2.0, rexp=1.5)
    sphere = M.Mesh(ctx, an["sphere_xyz"], an["sphere_tri"])
    cf.set_anatomical(sphere, an["atarget_xyz"], an["asource_xyz"], an["sphere_tri"], an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    osphere = O.Mesh(an["sphere_xyz"], an["sphere_tri"])
    otree = O.Octree(osphere)
    oasrc = O.Mesh(an["asource_xyz"], an["sphere_tri"])
    oc.set_anatomical(osphere, otree, an["atarget_xyz"], oasrc, an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    assert np.diff(an["face_ptr"]).min() >= 1
    rng = np.random.default_rng(8)
    t, la, lb, lc = random_queries(rng, 600, cf.T, cf.L, 3)
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.isfinite(want).all() and np.ptp(want) > 0
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))
    # the fusion-move octets use the same evaluator
    labeling = rng.integers(0, cf.L, cf.N).astype(np.int32)
    E = cf.tripletOctets(labeling, 3)
    tt = 17
    ids = inp["triplets"][tt]
    k = 5  # (A,B,C) = (1,0,1)
    want_k = oc.triplet(tt, 3, int(labeling[ids[1]]), 3)
    assert abs(E[tt, k] - want_k) <= ATOL + RTOL * abs(want_k)


def test_anatomical_strain_needs_its_inputs(ctx):
    inp = problem.pairwise_inputs(4, 2, D=1)
    cf, _ = problem.build_cost(ctx, inp, kind="univariate", rmode=5)
    with pytest.raises(M.MsmError, match="anatomical"):
        cf.computeTripletCost(np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32))


def test_triplet_table_matches_on_demand_evaluation(ctx):
    # computeTripletCosts (the MCMC optimiser's tcosts[t][a][b][c], M/DiscreteCostFunction.cpp:245-253) in triplet ranges
    inp = problem.pairwise_inputs(4, 2, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.2)
    tab = cf.computeTripletCosts(5, 9)
    assert tab.shape == (4, cf.L, cf.L, cf.L)
    rng = np.random.default_rng(11)
    t, la, lb, lc = random_queries(rng, 300, 4, cf.L, 3)
    assert np.array_equal(tab[t, la, lb, lc], cf.computeTripletCost(t + 5, la, lb, lc))
    for q in range(0, 300, 25):
        want = oc.triplet(int(t[q]) + 5, int(la[q]), int(lb[q]), int(lc[q]))
        assert abs(tab[t[q], la[q], lb[q], lc[q]] - want) <= ATOL + RTOL * abs(want)
    with pytest.raises(M.MsmError):
        cf.computeTripletCosts(0, cf.T + 1)


@pytest.mark.parametrize("data_order,cp_order", [(5, 1), (5, 0)])
def test_triclique_large_bins(ctx, data_order, cp_order):
    # coarse control grids (first levels of a multiresolution run): 128 and 512 source vertices per control triangle;
    # 4- and 16-lane evaluation groups, LDS slices of one bin each
    inp = problem.pairwise_inputs(data_order, cp_order, D=1, warp_amp=0.3, warp_rot=1.0)
    cf, oc, _ = pair(ctx, inp, "ho_univariate", rmode=3, lambda_=0.1)
    ptr, _ = cf.patches()
    assert np.diff(ptr).max() > (256 if cp_order == 0 else 100)
    rng = np.random.default_rng(13)
    t, la, lb, lc = random_queries(rng, 120, cf.T, cf.L, 3)
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))


@pytest.mark.parametrize("data_order,kind,D,least,squeeze", [(5, "ho_multivariate", 9, 1024, 0.0), (6, "ho_univariate", 1, 1500, 0.0), (7, "ho_univariate", 1, 16384, 0.0),
                                                             (7, "ho_univariate", 1, 32768, 3.0), (6, "ho_multivariate", 3, 16384, 6.0)])
def test_triclique_bins_beyond_1024_points(ctx, data_order, kind, D, least, squeeze):
    """An ico0 control grid under fine data (tests/fuzz_parity.py found the first case: 1 052 source vertices under one control triangle was
    MSM_ERR_CAPACITY): 64 lanes per evaluation up to 4 096 points, the whole workgroup with an LDS slice up to 16 384, and beyond that (round 4:
    the order-7 case puts 16 436 vertices under one triangle, and a source sphere whose vertices crowd towards one pole more than 32 768) the
    values of an evaluation live in HBM: the reference pushes a bin into a std::vector and has no limit (M/DiscreteCostFunction.cpp:468-485).
    Single evaluations and a whole label step."""
    kw = dict(seed=86634, warp_amp=0.3482850938627505, warp_rot=3.350507727100841, labeldist=0.33807290039714544, rescale=False)
    inp = problem.pairwise_inputs(data_order, 0, D=D, **kw)
    if squeeze > 0.0:  # the polar angle about the middle of a control triangle shrinks, theta -> pi (theta / pi)^squeeze: a bijection of the sphere
        e = inp["cp_xyz"][inp["cp_tri"][0]].mean(axis=0)   # that crowds most source vertices under that triangle
        e = e / np.linalg.norm(e)
        u = inp["source_xyz"] / 100.0
        c = np.clip(u @ e, -1.0, 1.0)
        theta = np.arccos(c)
        tang = u - c[:, None] * e
        tang = tang / np.maximum(np.linalg.norm(tang, axis=1, keepdims=True), 1e-300)
        th2 = np.pi * (theta / np.pi) ** squeeze
        inp["source_xyz"] = 100.0 * (np.cos(th2)[:, None] * e + np.sin(th2)[:, None] * tang)
    cf, oc, _ = pair(ctx, inp, kind, simmeasure=1 if D == 1 else 2, rmode=3, lambda_=0.1)
    ptr, _ = cf.patches()
    assert np.diff(ptr).max() > least, np.diff(ptr).max()
    rng = np.random.default_rng(77)
    big = int(np.argmax(np.diff(ptr)))  # the triangle with the largest bin is among the queries
    n = 40 if data_order < 7 and squeeze == 0.0 else 10
    t, la, lb, lc = random_queries(rng, n, cf.T, cf.L, 3)
    t[0] = big
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.isfinite(want).any()
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True), np.nanmax(np.abs(got - want))
    if data_order < 7:
        lab = rng.integers(0, cf.L, cf.N).astype(np.int32)
        E, Eo = cf.tripletOctets(lab, 5), oc.triplet_octets(lab, 5, threads=8)
        assert np.allclose(E, Eo, rtol=RTOL, atol=ATOL, equal_nan=True), np.nanmax(np.abs(E - Eo))
        oc.set_pairs(np.zeros((0, 2), dtype=np.int32))  # the model holds pairs or triplets, never both
        tot, otot = cf.evaluateTotalCostSum(lab)[0], oc.total(lab)[0]
        assert abs(tot - otot) <= 1e-9 * abs(otot) + 1e-11


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 3)])
def test_triclique_with_many_labels(ctx, kind, D):
    """≈ 80 labels (finer sampling grid): on-demand triplet costs and the eight-combination fusion move against the oracle"""
    inp = problem.pairwise_inputs(4, 2, D=D, sg_order=5, rescale=False)
    assert len(inp["labels"]) > 60
    cf, oc, _ = pair(ctx, inp, kind, rmode=3, lambda_=0.1)
    rng = np.random.default_rng(4)
    t, la, lb, lc = random_queries(rng, 600, cf.T, cf.L, 3)
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.max(np.abs(got - want))
    labeling = rng.integers(0, cf.L, cf.N).astype(np.int32)
    E = cf.tripletOctets(labeling, 57).reshape(cf.T, 8)
    for tt in rng.integers(0, cf.T, 25):
        ids = inp["triplets"][tt]
        for k in range(8):
            lab = [57 if k >> (2 - j) & 1 else int(labeling[ids[j]]) for j in range(3)]
            w = oc.triplet(int(tt), *lab)
            assert abs(E[tt, k] - w) <= ATOL + RTOL * abs(w)


# ------------------------------------------------------------------------------------------------------------------------
# BASELINE config 2's --dopt=HOCR label step at its own size: the strain-only fusion move (k_triplet_octets_packed: the
# labeling in the kernel arguments -- 2 562 bytes of the 2 816 the argument block holds --, the costs written into mapped
# pinned memory) over all T x 8 evaluations against the oracle's replay of I/Fusion/Fusion.h:181-196.
# ------------------------------------------------------------------------------------------------------------------------
def fusion_labelings(cf, seed):
    rng = np.random.default_rng(seed)
    return [(np.zeros(cf.N, dtype=np.int32), int(rng.integers(1, cf.L))), (rng.integers(0, cf.L, cf.N).astype(np.int32), int(rng.integers(0, cf.L)))]


def check_octets_all_ways(ctx, cf, oc, monkeypatch, seed):
    for labeling, label in fusion_labelings(cf, seed):
        want = oc.triplet_octets(labeling, label, threads=8)
        folded = want >= 1e6 * oc.params.lambda_
        got = {"pageable": np.array(cf.tripletOctets(labeling, label)),          # staging block + memcpy
               "mapped": np.array(cf.tripletOctets(labeling, label, ctx.host_array((cf.T, 8))))}  # the kernel writes the caller's array
        monkeypatch.setenv("MSMHIP_OCTETS", "copy")  # the round-1 route: labeling and costs by copy commands, k_triplet_octets
        got["copy"] = np.array(cf.tripletOctets(labeling, label))
        monkeypatch.delenv("MSMHIP_OCTETS")
        for how, E in got.items():
            assert E.shape == (cf.T, 8) and np.isfinite(E).all(), how
            assert np.allclose(E, want, rtol=RTOL, atol=ATOL), (how, np.abs(E - want).max())
            assert np.array_equal(E >= 1e6 * oc.params.lambda_, folded), how
        assert np.array_equal(got["pageable"], got["mapped"]) and np.array_equal(got["pageable"], got["copy"])  # one evaluator, three deliveries
    return folded


def test_strain_only_fusion_move_config2_full_size(ctx, monkeypatch):
    inp = problem.pairwise_inputs(6, 4, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.1, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    assert (cf.N, cf.T, cf.L) == (2562, 5120, 19)
    check_octets_all_ways(ctx, cf, oc, monkeypatch, seed=61)
    # long label moves: folded proposals (the 1e7 * lambda sentinel of computeTripletCost :141-146) must be flagged identically
    inp = problem.pairwise_inputs(6, 4, D=1, labeldist=2.5)
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.1, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    folded = check_octets_all_ways(ctx, cf, oc, monkeypatch, seed=62)
    assert folded.any() and not folded.all()


def test_anatomical_strain_fusion_move_full_size(ctx, monkeypatch):
    """regoption 5 (aMSM) through the same packed label step at ico6 data / ico4 control grid / ico6 anatomical sphere"""
    import oracle.oracle as O

    inp = problem.pairwise_inputs(6, 4, D=1)
    an = problem.anatomical_inputs(ctx, inp)
    cf, oc, keep = pair(ctx, inp, "univariate", rmode=5, lambda_=0.05, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    sphere = M.Mesh(ctx, an["sphere_xyz"], an["sphere_tri"])
    cf.set_anatomical(sphere, an["atarget_xyz"], an["asource_xyz"], an["sphere_tri"], an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    osphere = O.Mesh(an["sphere_xyz"], an["sphere_tri"])
    oc.set_anatomical(osphere, O.Octree(osphere), an["atarget_xyz"], O.Mesh(an["asource_xyz"], an["sphere_tri"]), an["w_ptr"], an["w_cp"], an["w_val"],
                      an["face_ptr"], an["face_idx"])
    check_octets_all_ways(ctx, cf, oc, monkeypatch, seed=63)


def test_strain_only_fusion_move_beyond_the_kernel_argument_block(ctx, monkeypatch):
    """N > 2 816 control points (the labeling no longer fits the kernel arguments) and L > 256 labels (a label no longer fits a byte):
    both fall back to the labeling in device memory -- same costs"""
    inp = problem.pairwise_inputs(5, 5, D=1)  # 10 242 control points
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.1, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    assert cf.N > 2816
    check_octets_all_ways(ctx, cf, oc, monkeypatch, seed=64)
    inp = problem.pairwise_inputs(4, 2, D=1, sg_order=6, rescale=False)
    assert len(inp["labels"]) > 256
    cf, oc, _ = pair(ctx, inp, "univariate", rmode=3, lambda_=0.1, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    check_octets_all_ways(ctx, cf, oc, monkeypatch, seed=65)
