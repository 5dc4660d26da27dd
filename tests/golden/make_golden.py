#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_small.npz.

The reference ships no golden vectors and cannot be built here, so these vectors are
ORACLE outputs (oracle/, the C restatement) on small seeded inputs, frozen so that (a) an accidental change of
the oracle is caught on CPU and (b) the GPU path can be checked against committed numbers.  They are not
reference outputs and do not pin the oracle to the reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from newmsm_amd import problem, synthetic  # noqa: E402  (host-only helpers, no GPU)
from oracle import oracle as O  # noqa: E402
from tests.helpers import oracle_cost  # noqa: E402


def main():
    out = {}
    # G1: octree queries on ico3, both weight modes
    xyz, tri = O.icosphere(3)
    q = np.concatenate([synthetic.random_sphere_points(300, seed=1), xyz[::7] @ synthetic.rotation([1, 2, 3], 3.0).T])
    t = O.Octree(O.Mesh(xyz, tri))
    st, tid, vid, w = t.barycentric_weights(q)
    _, _, _, wr = t.barycentric_weights(q, raw=True)
    out.update(g1_q=q, g1_tri=tid, g1_vid=vid, g1_w=w, g1_wraw=wr)
    # G2: adaptive barycentric CSR ico3 -> ico2 (warped input)
    xin = synthetic.known_warp(xyz, seed=21, rot_deg=5.0, amp=1.0)
    x2, t2 = O.icosphere(2)
    rp, col, val = O.adaptive_barycentric_weights(O.Mesh(xin, tri), O.Mesh(x2, t2))
    out.update(g2_xin=xin, g2_rp=rp, g2_col=col, g2_val=val)
    # G3/G4: unary tables (univariate, multivariate D=3) and clique costs on ico4 data / ico2 control grid
    inp = problem.pairwise_inputs(4, 2, D=3)
    inp1 = dict(inp, src_feat=inp["src_feat"][:1], ref_feat=inp["ref_feat"][:1], D=1)
    oc = oracle_cost(inp1, "univariate", rmode=3)
    oc.get_source_data()
    ptr, idx = oc.patches()
    out.update(g3_ptr=ptr, g3_idx=idx, g3_absw=oc.absolute_weights(), g3_unary=oc.unary_table())
    om = oracle_cost(inp, "multivariate")
    om.get_source_data()
    out.update(g3_unary_mv=om.unary_table())
    rng = np.random.default_rng(0)
    tq = np.stack([rng.integers(0, oc.T, 200), rng.integers(0, oc.L, 200), rng.integers(0, oc.L, 200), rng.integers(0, oc.L, 200)], 1)
    out.update(g4_tq=tq, g4_triplet=np.array([oc.triplet(*r) for r in tq]))
    op = oracle_cost(inp1, "univariate", rmode=1)
    pq = np.stack([rng.integers(0, op.P, 200), rng.integers(0, op.L, 200), rng.integers(0, op.L, 200)], 1)
    out.update(g4_pq=pq, g4_pairwise=np.array([op.pairwise(*r) for r in pq]))
    # G5: strain energy of random triangle pairs
    tris = rng.normal(size=(100, 2, 3, 3)) + np.array([100.0, 0, 0])
    out.update(g5_tris=tris, g5_strain=np.array([O.triangular_strain(a, b, 0.1, 10.0, 2.0) for a, b in tris]))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"), **out)
    print("wrote oracle_small.npz:", {k: v.shape for k, v in out.items()})
    more(inp, inp1)


def anatomy_case():
    """Inputs of the anatomical-strain vectors (shared with tests/test_golden.py)."""
    from tests.helpers import oracle_anatomy_by_queries

    cxyz, ctri, axyz, atri, w_ptr, w_cp, w_val, face_ptr, face_idx = oracle_anatomy_by_queries(2, 4)
    rs = 60.0 + 6.0 * synthetic.smooth_feature(axyz, 0, 99) + 3.0 * synthetic.smooth_feature(axyz, 1, 99)
    rt = 62.0 + 5.0 * synthetic.smooth_feature(axyz, 2, 100) + 4.0 * synthetic.smooth_feature(axyz, 0, 101)
    d = axyz / synthetic.RAD
    return dict(sphere_xyz=axyz, sphere_tri=atri, asource_xyz=d * rs[:, None], atarget_xyz=d * rt[:, None], w_ptr=w_ptr, w_cp=w_cp, w_val=w_val,
                face_ptr=face_ptr, face_idx=face_idx)


def more(inp, inp1):
    """Second file (later additions): DICE table, triclique likelihood, anatomical strain, smooth_data."""
    out = {}
    rng = np.random.default_rng(7)
    od = oracle_cost(inp1, "univariate", simmeasure=4, percentile=0.6)
    od.get_source_data()
    out["g6_dice_unary"] = od.unary_table()
    oh = oracle_cost(inp1, "ho_univariate", rmode=3, lambda_=0.1)
    oh.get_source_data()
    tq = np.stack([rng.integers(0, oh.T, 150), rng.integers(0, oh.L, 150), rng.integers(0, oh.L, 150), rng.integers(0, oh.L, 150)], 1)
    out.update(g7_tq=tq, g7_triclique=np.array([oh.triplet(*r) for r in tq]))
    an = anatomy_case()
    oa = oracle_cost(inp1, "univariate", rmode=5, lambda_=0.05, mu=0.4, kappa=1.6, rexp=1.5)
    sphere = O.Mesh(an["sphere_xyz"], an["sphere_tri"])
    asrc = O.Mesh(an["asource_xyz"], an["sphere_tri"])
    oa.set_anatomical(sphere, O.Octree(sphere), an["atarget_xyz"], asrc, an["w_ptr"], an["w_cp"], an["w_val"], an["face_ptr"], an["face_idx"])
    out.update(g8_tq=tq, g8_anat_triplet=np.array([oa.triplet(*r) for r in tq]))
    xyz, tri = O.icosphere(3)
    data = np.stack([synthetic.smooth_feature(xyz, 0), rng.normal(size=len(xyz))])
    m = O.Mesh(xyz, tri)
    out.update(g9_data=data, g9_smooth=O.smooth_data(m, data, m, 12.0))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_more.npz"), **out)
    print("wrote oracle_more.npz:", {k: v.shape for k, v in out.items()})


def folded_ico3():
    """an ico3 sphere with five vertices pushed across a neighbour (shared with tests/test_golden.py)"""
    xyz, tri = O.icosphere(3)
    m = O.Mesh(xyz, tri)
    nbr_ptr, nbr, _, _ = m.adjacency()
    bad = xyz.copy()
    for v in (17, 101, 230, 400, 601):
        n = nbr[nbr_ptr[v] + 1]
        p = xyz[n] + 1.3 * (xyz[n] - xyz[v])
        bad[v] = p * 100.0 / np.linalg.norm(p)
    return xyz, tri, bad


def around():
    """Third file: what brackets the path each iteration -- unfold, variance_normalise, exclusion masks."""
    out = {}
    xyz, tri, bad = folded_ico3()
    m = O.Mesh(bad, tri)
    passes, first = O.unfold(m)
    out.update(h1_counts=np.array([passes, first]), h1_unfolded=m.xyz)
    rng = np.random.default_rng(11)
    data = rng.normal(1.0, 2.0, size=(2, len(xyz)))
    keep = (rng.random(len(xyz)) > 0.3).astype(float)
    out.update(h2_data=data, h2_keep=keep, h2_normed=O.variance_normalise(data, keep))
    excl = O.create_exclusion(data, -0.5, 1e9)
    x2, t2 = O.icosphere(2)
    res, mask = O.metric_resample_excl(O.Mesh(xyz, tri), data, O.Mesh(x2, t2), excl)
    q = synthetic.random_sphere_points(200, seed=3)
    nn, nmask = O.nearest_neighbour_excl(O.Mesh(xyz, tri), data, q, excl)
    out.update(h3_excl=excl, h3_resampled=res, h3_mask=mask, h3_q=q, h3_nn=nn, h3_nnmask=nmask)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_around.npz"), **out)
    print("wrote oracle_around.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
    around()
