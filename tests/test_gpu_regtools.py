"""unfold (M/reg_tools.cpp:131-178) through the C ABI against the oracle: the GPU fold test and the host repair pass use
the reference's FP64 operation order, so the unfolded coordinates must be bit-exact."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import synthetic
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def folded_sphere(order, nfold, seed, push=1.3):
    """regular sphere with `nfold` vertices pushed across one of their neighbours (adjacent picks included)"""
    xyz, tri = M.make_mesh_from_icosa(order)
    nbr_ptr, nbr, _, _ = M.mesh_adjacency(tri, len(xyz))
    rng = np.random.default_rng(seed)
    out = xyz.copy()
    for v in rng.choice(len(xyz), nfold, replace=False):
        n = nbr[nbr_ptr[v] + rng.integers(0, nbr_ptr[v + 1] - nbr_ptr[v])]
        p = xyz[n] + push * (xyz[n] - xyz[v])
        out[v] = p * 100.0 / np.linalg.norm(p)
    return xyz, tri, out


def test_unfold_leaves_a_regular_mesh_alone(ctx):
    xyz, tri = M.make_mesh_from_icosa(5)
    w = synthetic.known_warp(xyz, seed=3, rot_deg=5.0, amp=1.0)
    m = M.Mesh(ctx, w, tri)
    assert m.unfold() == (0, 0)
    assert np.array_equal(m.get_coords(), w)
    om = O.Mesh(w, tri)
    assert O.unfold(om) == (0, 0)


@pytest.mark.parametrize("order,nfold,seed", [(3, 1, 0), (3, 12, 1), (4, 60, 2), (5, 400, 3)])
def test_unfold_matches_oracle(ctx, order, nfold, seed):
    xyz, tri, bad = folded_sphere(order, nfold, seed)
    m = M.Mesh(ctx, bad, tri)
    passes, first = m.unfold()
    om = O.Mesh(bad, tri)
    opasses, ofirst = O.unfold(om)
    assert first == ofirst and first >= nfold
    assert passes == opasses and passes >= 1
    got = m.get_coords()
    assert np.array_equal(got, om.xyz)  # bit-exact
    assert np.allclose(np.linalg.norm(got, axis=1), 100.0, atol=1e-9)
    assert (np.abs(got - bad).max(axis=1) > 0).sum() >= nfold
    # the repaired mesh answers searches like any other: the tree is rebuilt from the new coordinates
    q = synthetic.random_sphere_points(2000, seed=seed)
    st, t, vid, w = m.query_triangles(q)
    ost, ot, ovid, ow = O.Octree(om).barycentric_weights(q)
    assert st == ost == 0 and np.array_equal(t, ot) and np.array_equal(w, ow)
    # and a second call finds nothing left to do when the first one converged
    if passes < 1000:
        assert m.unfold() == (0, 0)


def test_unfold_heavily_folded_patch(ctx):
    """a whole neighbourhood collapsed onto one point: many passes, same trajectory as the oracle"""
    xyz, tri = M.make_mesh_from_icosa(4)
    d = np.linalg.norm(xyz - xyz[100], axis=1)
    bad = xyz.copy()
    sel = d < 12.0
    bad[sel] = xyz[100] + 0.05 * (xyz[sel] - xyz[100])[::-1]
    bad *= 100.0 / np.linalg.norm(bad, axis=1, keepdims=True)
    m = M.Mesh(ctx, bad, tri)
    om = O.Mesh(bad, tri)
    res, ores = m.unfold(), O.unfold(om)
    assert res == ores and res[1] > 5
    assert np.array_equal(m.get_coords(), om.xyz)
