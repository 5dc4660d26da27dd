"""The oracle against everything the reference (which ships no tests or golden vectors, SURVEY.md
section 4) lets us pin: the structural statistics SURVEY.md section 8 recorded by running the
reference's own code, plus literal-vs-hashed icosphere generation."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("order,V,T", [(0, 12, 20), (1, 42, 80), (2, 162, 320), (3, 642, 1280), (4, 2562, 5120), (5, 10242, 20480), (6, 40962, 81920)])
def test_icosphere_sizes(order, V, T):
    # Mesh::get_resolution, R/mesh.cpp:810-830; docs/guide.md:41
    assert O.icosphere_counts(order) == (V, T)
    xyz, tri = O.icosphere(order)
    assert xyz.shape == (V, 3) and tri.shape == (T, 3)
    assert np.allclose(np.linalg.norm(xyz, axis=1), 100.0, rtol=0, atol=1e-12)
    assert tri.min() == 0 and tri.max() == V - 1


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_edge_hash_equals_reference_tolerance_search(order):
    a = O.icosphere(order, literal=True)
    b = O.icosphere(order, literal=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_icosphere_numbering_is_hierarchical():
    # the first vertices of a finer icosphere are the coarser one's (up to re-normalisation rounding)
    x4, _ = O.icosphere(4)
    x6, _ = O.icosphere(6)
    assert np.allclose(x6[: len(x4)], x4, rtol=0, atol=1e-12)


def test_octree_statistics_ico6():
    # SURVEY.md section 8 [probe]: 14 281 nodes, 12 496 leaves, depth 6, 176 096 references, <= 49 per leaf
    m = O.Mesh(*O.icosphere(6))
    s = O.Octree(m).stats()
    assert s == dict(nodes=14281, leaves=12496, depth=6, refs=176096, max_leaf=49)


def test_triangle_tests_per_query():
    # 33.9 distance_to_triangle calls per query on average (92.17 M / 2.716 M)
    xyz, tri = O.icosphere(5)
    t = O.Octree(O.Mesh(xyz, tri))
    R = O.rotation_matrix([0, 0, 1], [0.05, 0.03, 1])
    ids, ntests = t.closest_triangle(xyz @ R.T, count_tests=True)
    assert (ids >= 0).all()
    assert 32.0 < ntests / len(xyz) < 36.0


def test_patch_sizes_and_label_count_ico6_ico4():
    # P = 49 / 65.4 / 69 (167 458 patch points), L = 19, barycentre set 19 or 3
    x6, t6 = O.icosphere(6)
    x4, t4 = O.icosphere(4)
    target = O.Mesh(x6, t6)
    cp = O.Mesh(x4, t4)
    maxsep, mvd = O.cp_spacings(cp)
    sg = O.Mesh(*O.icosphere(6))
    _, samples, bary = O.label_sampling_grid(sg, 0.5 * mvd)
    assert len(samples) == 19 and len(bary) == 19
    _, _, bary_int = O.label_sampling_grid(sg, 0.5 * mvd, abs_is_int=True)
    assert len(bary_int) == 3
    c = O.Cost("univariate")
    c.set_meshes(target, O.Octree(target), O.Mesh(x6, t6), cp)
    f = np.sin(x6[:, 0] / 20.0)
    c.set_features(f[None], f[None])
    c.set_spacings(maxsep, mvd)
    c.get_source_data()
    ptr, idx = c.patches()
    sz = np.diff(ptr)
    assert (sz.min(), sz.max(), sz.sum()) == (49, 69, 167458)
    assert abs(sz.mean() - 65.4) < 0.05
    aw = c.absolute_weights()
    assert np.allclose(aw, 1.0, rtol=0, atol=1e-14)


def test_known_answers_geometry():
    # closed-form checks of the restated primitives
    a, b, c = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])
    p = O.project_point([2.0, 2.0, 2.0], a, b, c)
    assert np.allclose(p, [1 / 3] * 3, atol=1e-15)
    assert O.point_in_triangle(p, a, b, c)
    assert not O.point_in_triangle([2.0, -1.0, 0.0], a, b, c)
    assert abs(O.dist_to_point([0.5, 0.5, 0.0], a, b, c) - 0.0) < 1e-15  # on an edge
    R = O.rotation_matrix([1.0, 0, 0], [0, 1.0, 0])
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-15)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)
    assert np.array_equal(O.rotation_matrix([0, 0, 2.0], [0, 0, 5.0]), np.eye(3))
    # identical triangles have zero strain energy; a uniform 2x dilation has J = 4, R = 1
    tri = np.array([[100.0, 0, 0], [99.0, 5.0, 0], [99.0, 0, 5.0]])
    assert abs(O.triangular_strain(tri, tri, 0.1, 10.0, 2.0)) < 1e-12
    W = O.triangular_strain(tri, 2 * tri, 0.4, 1.6, 2.0)
    assert abs(W - 0.5 * 1.6 * (16 + 1 / 16 - 2)) < 1e-9
    # correlation of a vector with itself / its negation
    v = np.linspace(0, 1, 17) ** 2
    w = np.ones_like(v)
    assert abs(O.sim_for_min(2, v, v, w) - 0.0) < 1e-15
    assert abs(O.sim_for_min(2, v, -v, w) - 1.0) < 1e-15
    assert O.sim_for_min(2, v, np.zeros_like(v), w) == 0.5  # zero variance -> corr 0
