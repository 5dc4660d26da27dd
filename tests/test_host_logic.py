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
