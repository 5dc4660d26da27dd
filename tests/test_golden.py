"""Committed vectors (tests/golden/oracle_small.npz, made by tests/golden/make_golden.py from the ORACLE -- the
reference ships none).  CPU: the oracle still reproduces them.  GPU: the product reproduces them."""
import os

import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem
from oracle import oracle as O
from tests.helpers import oracle_cost

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz"))


def inputs():
    inp = problem.pairwise_inputs(4, 2, D=3)
    inp1 = dict(inp, src_feat=inp["src_feat"][:1], ref_feat=inp["ref_feat"][:1], D=1)
    return inp, inp1


def test_oracle_reproduces_golden(built):
    xyz, tri = O.icosphere(3)
    t = O.Octree(O.Mesh(xyz, tri))
    _, tid, vid, w = t.barycentric_weights(G["g1_q"])
    assert np.array_equal(tid, G["g1_tri"]) and np.array_equal(vid, G["g1_vid"]) and np.array_equal(w, G["g1_w"])
    rp, col, val = O.adaptive_barycentric_weights(O.Mesh(G["g2_xin"], tri), O.Mesh(*O.icosphere(2)))
    assert np.array_equal(rp, G["g2_rp"]) and np.array_equal(col, G["g2_col"]) and np.array_equal(val, G["g2_val"])
    inp, inp1 = inputs()
    oc = oracle_cost(inp1, "univariate", rmode=3)
    oc.get_source_data()
    assert np.array_equal(oc.patches()[1], G["g3_idx"]) and np.array_equal(oc.unary_table(), G["g3_unary"])
    assert np.array_equal(np.array([oc.triplet(*r) for r in G["g4_tq"]]), G["g4_triplet"])
    assert np.array_equal(np.array([O.triangular_strain(a, b, 0.1, 10.0, 2.0) for a, b in G["g5_tris"]]), G["g5_strain"])


@pytest.mark.gpu
def test_gpu_reproduces_golden(ctx):
    xyz, tri = M.make_mesh_from_icosa(3)
    _, tid, vid, w = M.Mesh(ctx, xyz, tri).query_triangles(G["g1_q"])
    assert np.array_equal(tid, G["g1_tri"]) and np.array_equal(vid, G["g1_vid"]) and np.array_equal(w, G["g1_w"])
    _, _, _, wr = M.Mesh(ctx, xyz, tri).query_triangles(G["g1_q"], mode=M.WEIGHTS_RAW)
    assert np.array_equal(wr, G["g1_wraw"])
    rp, col, val = M.get_adaptive_barycentric_weights(M.Mesh(ctx, G["g2_xin"], tri), M.Mesh(ctx, *M.make_mesh_from_icosa(2)))
    assert np.array_equal(rp, G["g2_rp"]) and np.array_equal(col, G["g2_col"]) and np.array_equal(val, G["g2_val"])
    inp, inp1 = inputs()
    cf, _ = problem.build_cost(ctx, inp1, kind="univariate", rmode=3)
    cf.get_source_data()
    ptr, idx = cf.patches()
    assert np.array_equal(ptr, G["g3_ptr"]) and np.array_equal(idx, G["g3_idx"]) and np.array_equal(cf.absolute_weights(), G["g3_absw"])
    assert np.allclose(cf.computeUnaryCosts(), G["g3_unary"], rtol=1e-10, atol=1e-12)
    tq = G["g4_tq"].astype(np.int32)
    assert np.allclose(cf.computeTripletCost(tq[:, 0], tq[:, 1], tq[:, 2], tq[:, 3]), G["g4_triplet"], rtol=1e-9, atol=1e-11)
    cm, _ = problem.build_cost(ctx, inp, kind="multivariate")
    cm.get_source_data()
    assert np.allclose(cm.computeUnaryCosts(), G["g3_unary_mv"], rtol=1e-9, atol=1e-11)
    cp, _ = problem.build_cost(ctx, inp1, kind="univariate", rmode=1)
    pq = G["g4_pq"].astype(np.int32)
    assert np.allclose(cp.computePairwiseCost(pq[:, 0], pq[:, 1], pq[:, 2]), G["g4_pairwise"], rtol=1e-9, atol=1e-11)


# ---- second file: later additions (DICE, triclique likelihood, anatomical strain, smooth_data)
G2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_more.npz"))


def _anatomy():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.anatomy_case()


def test_oracle_reproduces_golden_more(built):
    _, inp1 = inputs()
    od = oracle_cost(inp1, "univariate", simmeasure=4, percentile=0.6)
    od.get_source_data()
    assert np.array_equal(od.unary_table(), G2["g6_dice_unary"])
    oh = oracle_cost(inp1, "ho_univariate", rmode=3, lambda_=0.1)
    oh.get_source_data()
    assert np.array_equal(np.array([oh.triplet(*r) for r in G2["g7_tq"]]), G2["g7_triclique"])
    an = _anatomy()
    oa = oracle_cost(inp1, "univariate", rmode=5, lambda_=0.05, mu=0.4, kappa=1.6, rexp=1.5)
    sphere = O.Mesh(an["sphere_xyz"], an["sphere_tri"])
    asrc = O.Mesh(an["asource_xyz"], an["sphere_tri"])
    oa.set_anatomical(sphere, O.Octree(sphere), an["atarget_xyz"], asrc, an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    assert np.array_equal(np.array([oa.triplet(*r) for r in G2["g8_tq"]]), G2["g8_anat_triplet"])
    xyz, tri = O.icosphere(3)
    m = O.Mesh(xyz, tri)
    assert np.array_equal(O.smooth_data(m, G2["g9_data"], m, 12.0), G2["g9_smooth"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_more(ctx):
    _, inp1 = inputs()
    cd, _ = problem.build_cost(ctx, inp1, kind="univariate", simmeasure=4, percentile=0.6)
    cd.get_source_data()
    assert np.array_equal(cd.computeUnaryCosts(), G2["g6_dice_unary"])  # counting measure on bit-exact samples
    ch, _ = problem.build_cost(ctx, inp1, kind="ho_univariate", rmode=3, lambda_=0.1)
    ch.get_source_data()
    tq = G2["g7_tq"].astype(np.int32)
    assert np.allclose(ch.computeTripletCost(tq[:, 0], tq[:, 1], tq[:, 2], tq[:, 3]), G2["g7_triclique"], rtol=1e-9, atol=1e-11)
    an = _anatomy()
    ca, _ = problem.build_cost(ctx, inp1, kind="univariate", rmode=5, lambda_=0.05, mu=0.4, kappa=1.6, rexp=1.5)
    sphere = M.Mesh(ctx, an["sphere_xyz"], an["sphere_tri"])
    ca.set_anatomical(sphere, an["atarget_xyz"], an["asource_xyz"], an["sphere_tri"], an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    assert np.allclose(ca.computeTripletCost(tq[:, 0], tq[:, 1], tq[:, 2], tq[:, 3]), G2["g8_anat_triplet"], rtol=1e-9, atol=1e-11)
    xyz, tri = M.make_mesh_from_icosa(3)
    m = M.Mesh(ctx, xyz, tri)
    assert np.allclose(M.smooth_data(m, G2["g9_data"], m, 12.0), G2["g9_smooth"], rtol=1e-12, atol=1e-14)


G3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_around.npz"))


def _folded():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.folded_ico3()


def test_oracle_reproduces_golden_around(built):
    xyz, tri, bad = _folded()
    m = O.Mesh(bad, tri)
    assert tuple(G3["h1_counts"]) == O.unfold(m) and np.array_equal(m.xyz, G3["h1_unfolded"])
    assert np.array_equal(O.variance_normalise(G3["h2_data"], G3["h2_keep"]), G3["h2_normed"])
    assert np.array_equal(M.variance_normalise(G3["h2_data"], G3["h2_keep"]), G3["h2_normed"])  # [host] entry point, no GPU
    assert np.array_equal(M.create_exclusion(G3["h2_data"], -0.5, 1e9), G3["h3_excl"])
    x2, t2 = O.icosphere(2)
    res, mask = O.metric_resample_excl(O.Mesh(xyz, tri), G3["h2_data"], O.Mesh(x2, t2), G3["h3_excl"])
    assert np.array_equal(res, G3["h3_resampled"], equal_nan=True) and np.array_equal(mask, G3["h3_mask"], equal_nan=True)


@pytest.mark.gpu
def test_gpu_reproduces_golden_around(ctx):
    xyz, tri, bad = _folded()
    m = M.Mesh(ctx, bad, tri)
    assert m.unfold() == tuple(G3["h1_counts"]) and np.array_equal(m.get_coords(), G3["h1_unfolded"])
    x2, t2 = M.make_mesh_from_icosa(2)
    src, dst = M.Mesh(ctx, xyz, tri), M.Mesh(ctx, x2, t2)
    res, mask = M.metric_resample(src, G3["h2_data"], dst, excl=G3["h3_excl"])
    assert np.array_equal(res, G3["h3_resampled"], equal_nan=True) and np.array_equal(mask, G3["h3_mask"], equal_nan=True)
    nn, nmask = M.nearest_neighbour_interpolation(src, G3["h2_data"], G3["h3_q"], excl=G3["h3_excl"])
    assert np.array_equal(nn, G3["h3_nn"]) and np.array_equal(nmask, G3["h3_nnmask"])
