"""Nearest-triangle search and resampling on the GPU against the oracle.  Index results (triangle,
vertex ids, patch membership, CSR structure) must be bit-exact; barycentric weights too, because the
kernels use the reference's FP64 operation order (no FMA contraction, no transcendental on this path)."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import synthetic
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def queries(xyz, n, seed):
    """mix of: rotated mesh vertices (land on/near vertices and edges), random points, exact vertices, edge midpoints"""
    rng = np.random.default_rng(seed)
    R = synthetic.rotation(rng.normal(size=3), 2.5)
    a = (xyz @ R.T)[rng.permutation(len(xyz))[: n // 2]]
    b = synthetic.random_sphere_points(n // 4, seed + 1)
    c = xyz[rng.permutation(len(xyz))[: n // 8]]
    i, j = rng.integers(0, len(xyz), n // 8), rng.integers(0, len(xyz), n // 8)
    d = xyz[i] + xyz[j]
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 1e-6, d / np.maximum(nrm, 1e-6) * 100.0, xyz[i])
    return np.concatenate([a, b, c, d])


@pytest.mark.parametrize("order", [2, 4, 5])
def test_octree_layout_matches_oracle(ctx, order):
    xyz, tri = M.make_mesh_from_icosa(order)
    m = M.Mesh(ctx, xyz, tri)
    assert m.octree_stats() == O.Octree(O.Mesh(xyz, tri)).stats()


def test_octree_stats_ico6_pin(ctx):
    m = M.Mesh(ctx, *M.make_mesh_from_icosa(6))
    assert m.octree_stats() == dict(nodes=14281, leaves=12496, depth=6, refs=176096, max_leaf=49)


@pytest.mark.parametrize("order,n", [(1, 500), (3, 4000), (5, 20000)])
@pytest.mark.parametrize("mode", [M.WEIGHTS_PROJECTED, M.WEIGHTS_RAW])
def test_query_bit_exact_regular_sphere(ctx, order, n, mode):
    xyz, tri = M.make_mesh_from_icosa(order)
    q = queries(xyz, n, seed=order)
    st, t, vid, w = M.Mesh(ctx, xyz, tri).query_triangles(q, mode=mode)
    ost, ot, ovid, ow = O.Octree(O.Mesh(xyz, tri)).barycentric_weights(q, raw=(mode == M.WEIGHTS_RAW))
    assert st == 0 and ost == 0
    assert np.array_equal(t, ot)
    assert np.array_equal(vid, ovid)
    assert np.array_equal(w, ow)  # bit-exact


def test_query_bit_exact_warped_sphere(ctx):
    xyz, tri = M.make_mesh_from_icosa(5)
    wxyz = synthetic.known_warp(xyz, seed=11, rot_deg=4.0, amp=1.5)
    q = queries(xyz, 20000, seed=9)
    st, t, vid, w = M.Mesh(ctx, wxyz, tri).query_triangles(q)
    ost, ot, ovid, ow = O.Octree(O.Mesh(wxyz, tri)).barycentric_weights(q)
    assert st == 0 and ost == 0
    assert np.array_equal(t, ot) and np.array_equal(vid, ovid) and np.array_equal(w, ow)


def test_query_into_pinned_arrays_of_the_caller(ctx):
    """msm_query_triangles with its four arrays in pinned blocks of the context (msm_host_alloc): read and written by the copy engine where they lie
    (no staging memcpy) -- the same bits as through the staging block, also when only some of the arrays are pinned, and the resampled matrix of
    msm_metric_resample written into a pinned array of the caller"""
    xyz, tri = M.make_mesh_from_icosa(5)
    wxyz = synthetic.known_warp(xyz, seed=11, rot_deg=4.0, amp=1.5)
    mesh = M.Mesh(ctx, wxyz, tri)
    q = queries(xyz, 20000, seed=9)
    st, t, vid, w = mesh.query_triangles(q)
    N = len(q)
    q_soa = ctx.host_array((3, N))
    q_soa[:] = q.T
    o_t, o_v, o_w = ctx.host_array((N,), np.int32), ctx.host_array((3, N), np.int32), ctx.host_array((3, N))
    o_t[:], o_v[:], o_w[:] = -5, -5, -5.0
    assert mesh.query_triangles_soa(q_soa, o_t, o_v, o_w) == 0
    assert np.array_equal(o_t, t) and np.array_equal(o_v.T, vid) and np.array_equal(o_w.T, w)
    # mixed: pageable queries and weights, pinned ids
    p_w = np.full((3, N), -5.0)
    o_t[:], o_v[:] = -5, -5
    assert mesh.query_triangles_soa(np.ascontiguousarray(q.T), o_t, o_v, p_w) == 0
    assert np.array_equal(o_t, t) and np.array_equal(o_v.T, vid) and np.array_equal(p_w.T, w)
    # metric_resample: pinned data in, pinned result out
    coarse = M.Mesh(ctx, *M.make_mesh_from_icosa(4))
    data = synthetic.features(xyz, 3, 4)
    want = M.metric_resample(mesh, data, coarse)
    pin_in, pin_out = ctx.host_array(data.shape), ctx.host_array((3, coarse.V))
    pin_in[:] = data
    pin_out[:] = -7.0
    got = M.metric_resample(mesh, pin_in, coarse, out=pin_out)
    assert got is pin_out and np.array_equal(got, want)
    for arr in (q_soa, o_t, o_v, o_w, pin_in, pin_out):
        ctx.release_host_array(arr)


def test_query_folded_mesh_uses_reference_tie_breaks(ctx):
    # strong high-frequency warp: folds and slivers -> several triangles pass the inside test, fallbacks fire
    xyz, tri = M.make_mesh_from_icosa(4)
    rng = np.random.default_rng(5)
    wxyz = xyz + rng.normal(scale=1.5, size=xyz.shape)
    wxyz = wxyz / np.linalg.norm(wxyz, axis=1, keepdims=True) * 100.0
    q = queries(xyz, 8000, seed=2)
    st, t, vid, w = M.Mesh(ctx, wxyz, tri).query_triangles(q, check_status=False)
    ost, ot, ovid, ow = O.Octree(O.Mesh(wxyz, tri)).barycentric_weights(q)
    assert np.array_equal(t, ot)
    ok = ot >= 0
    assert np.array_equal(vid[ok], ovid[ok]) and np.array_equal(w[ok], ow[ok])
    assert (st == 0) == (ost == 0)


def test_query_outside_root_box_reports_reference_error(ctx):
    xyz, tri = M.make_mesh_from_icosa(2)
    m = M.Mesh(ctx, xyz, tri)
    q = np.array([[0.0, 0.0, 100.0], [0.0, 150.0, 0.0]])
    with pytest.raises(M.MsmError) as e:
        m.query_triangles(q)
    assert e.value.code == -3 and "bounding box" in str(e.value)
    st, t, _, _ = m.query_triangles(q, check_status=False)
    assert t[0] >= 0 and t[1] == -3
    # the context stays usable
    st, t, _, _ = m.query_triangles(q[:1])
    assert st == 0


def test_empty_and_ragged_inputs(ctx):
    xyz, tri = M.make_mesh_from_icosa(2)
    m = M.Mesh(ctx, xyz, tri)
    st, t, vid, w = m.query_triangles(np.zeros((0, 3)))
    assert st == 0 and len(t) == 0
    for n in (1, 63, 64, 65, 257):
        q = synthetic.random_sphere_points(n, seed=n)
        _, t, _, _ = m.query_triangles(q)
        assert np.array_equal(t, O.Octree(O.Mesh(xyz, tri)).closest_triangle(q))


def test_closest_vertex(ctx):
    xyz, tri = M.make_mesh_from_icosa(4)
    q = queries(xyz, 6000, seed=4)
    got = M.Mesh(ctx, xyz, tri).get_closest_vertex_ID(q)
    assert np.array_equal(got, O.Octree(O.Mesh(xyz, tri)).closest_vertex(q))


def test_update_coords_rebuilds_search_structure(ctx):
    xyz, tri = M.make_mesh_from_icosa(4)
    m = M.Mesh(ctx, xyz, tri)
    q = queries(xyz, 3000, seed=8)
    m.query_triangles(q)
    wxyz = synthetic.known_warp(xyz, seed=3, rot_deg=10.0, amp=2.0)
    m.set_coords(wxyz)
    _, t, vid, w = m.query_triangles(q)
    _, ot, ovid, ow = O.Octree(O.Mesh(wxyz, tri)).barycentric_weights(q)
    assert np.array_equal(t, ot) and np.array_equal(w, ow)


@pytest.mark.parametrize("oin,onew", [(4, 3), (3, 4), (5, 4)])
def test_adaptive_barycentric_weights_bit_exact(ctx, oin, onew):
    xin, tin = M.make_mesh_from_icosa(oin)
    xnew, tnew = M.make_mesh_from_icosa(onew)
    xin = synthetic.known_warp(xin, seed=21, rot_deg=5.0, amp=1.0)
    rp, col, val = M.get_adaptive_barycentric_weights(M.Mesh(ctx, xin, tin), M.Mesh(ctx, xnew, tnew))
    orp, ocol, oval = O.adaptive_barycentric_weights(O.Mesh(xin, tin), O.Mesh(xnew, tnew))
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(val, oval)
    # rows are convex combinations
    sums = np.add.reduceat(val, rp[:-1])
    assert np.allclose(sums, 1.0, atol=1e-14)


@pytest.mark.parametrize("oin,onew", [(6, 2), (1, 5), (6, 6), (0, 4)])
def test_adaptive_weights_long_lists_bit_exact(ctx, oin, onew):
    """The device list surgery (resample_kernels.hip) across very different resolutions: hundreds of entries per
    transposed reverse list (fine -> coarse, sorted by the workgroup rank sort) or per correction column (coarse -> fine),
    and two warped ico6 meshes (the gMSM set-up case).  Same CSR and same bits as the reference's serial std::map surgery."""
    xin, tin = M.make_mesh_from_icosa(oin)
    xnew, tnew = M.make_mesh_from_icosa(onew)
    xin = synthetic.known_warp(xin, seed=33, rot_deg=4.0, amp=1.5)
    min_, mnew = M.Mesh(ctx, xin, tin), M.Mesh(ctx, xnew, tnew)
    rp, col, val = M.get_adaptive_barycentric_weights(min_, mnew)
    orp, ocol, oval = O.adaptive_barycentric_weights(O.Mesh(xin, tin), O.Mesh(xnew, tnew))
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(val, oval)
    data = synthetic.features(xin, 2)
    assert np.array_equal(M.metric_resample(min_, data, mnew), O.metric_resample(O.Mesh(xin, tin), data, O.Mesh(xnew, tnew)))


def test_adaptive_weights_with_exclusion(ctx):
    xin, tin = M.make_mesh_from_icosa(4)
    xnew, tnew = M.make_mesh_from_icosa(3)
    excl = (xin[:, 2] > -20).astype(float)
    rp, col, val = M.get_adaptive_barycentric_weights(M.Mesh(ctx, xin, tin), M.Mesh(ctx, xnew, tnew), excl)
    orp, ocol, oval = O.adaptive_barycentric_weights(O.Mesh(xin, tin), O.Mesh(xnew, tnew), excl)
    # rows next to the mask edge divide 0 by a zero scatter-sum in the reference too (NaN): compare NaN-aware
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(val, oval, equal_nan=True)
    assert (np.diff(rp) == 0).any()  # excluded rows are empty


def test_metric_resample_warp_and_nn(ctx):
    xin, tin = M.make_mesh_from_icosa(5)
    xnew, tnew = M.make_mesh_from_icosa(4)
    data = synthetic.features(xin, 3)
    min_, mnew = M.Mesh(ctx, xin, tin), M.Mesh(ctx, xnew, tnew)
    oin, onew = O.Mesh(xin, tin), O.Mesh(xnew, tnew)
    assert np.array_equal(M.metric_resample(min_, data, mnew), O.metric_resample(oin, data, onew))
    # sphere_project_warp: carry a fine sphere through a coarse deformation
    to = synthetic.known_warp(xnew, seed=5, rot_deg=6.0, amp=2.0)
    assert np.array_equal(M.sphere_project_warp(xin, mnew, to), O.sphere_project_warp(xin, onew, to))
    # the same for the coordinates a mesh handle holds, in place on the device (msm_mesh_sphere_project_warp): identical bits, host copy follows,
    # and the moved mesh can be searched and warped again
    from newmsm_amd import api

    moving = M.Mesh(ctx, xin, tin)
    api.sphere_project_warp_mesh(moving, mnew, to)
    once = O.sphere_project_warp(xin, onew, to)
    assert np.array_equal(moving.get_coords(), once)
    api.sphere_project_warp_mesh(moving, mnew, to)
    assert np.array_equal(moving.get_coords(), O.sphere_project_warp(once, onew, to))
    om = O.Mesh(O.sphere_project_warp(once, onew, to), tin)
    assert np.array_equal(M.metric_resample(moving, data, mnew), O.metric_resample(om, data, onew))  # its tree follows the new coordinates
    with pytest.raises(M.MsmError):
        api.sphere_project_warp_mesh(M.Mesh(ctx, xin * 1.2, tin), mnew, to)  # points outside the octree's root: the reference throws, so does this
    q = queries(xin, 3000, seed=6)
    assert np.array_equal(M.nearest_neighbour_interpolation(min_, data, q), O.nearest_neighbour(oin, data, q))


def test_warp_roundtrip_property(ctx):
    # size-independent property at full size: warping through the identity deformation leaves an ico6 sphere
    # where it is (to rounding), and through a rigid rotation rotates it
    x6, t6 = M.make_mesh_from_icosa(6)
    x4, t4 = M.make_mesh_from_icosa(4)
    m4 = M.Mesh(ctx, x4, t4)
    same = M.sphere_project_warp(x6, m4, x4)
    assert np.max(np.abs(same - x6)) < 1e-10
    R = synthetic.rotation([0.3, -1.0, 0.5], 7.0)
    rot = M.sphere_project_warp(x6, m4, x4 @ R.T)
    assert np.max(np.abs(rot - x6 @ R.T)) < 0.05  # piecewise-linear interpolation of a rotation, then re-projection
    assert np.allclose(np.linalg.norm(rot, axis=1), 100.0, atol=1e-10)


def test_repeated_coordinate_updates_grow_and_shrink_the_tree(ctx):
    # the octree's node / leaf-entry counts change with the geometry; every device buffer must follow
    xyz, tri = M.make_mesh_from_icosa(3)
    m = M.Mesh(ctx, xyz, tri)
    q = queries(xyz, 2000, seed=12)
    rng = np.random.default_rng(9)
    sizes = set()
    for k, scale in enumerate([0.0, 3.0, 0.2, 6.0, 0.0]):
        w = xyz + rng.normal(scale=scale, size=xyz.shape)
        w = w / np.linalg.norm(w, axis=1, keepdims=True) * 100.0
        m.set_coords(w)
        sizes.add(m.octree_stats()["nodes"])
        _, t, vid, wt = m.query_triangles(q, check_status=False)
        _, ot, ovid, ow = O.Octree(O.Mesh(w, tri)).barycentric_weights(q)
        assert np.array_equal(t, ot)
        ok = ot >= 0
        assert np.array_equal(wt[ok], ow[ok])
    assert len(sizes) > 1


@pytest.mark.parametrize("order,sigma", [(4, 4.0), (5, 2.0)])
def test_smooth_data(ctx, order, sigma):
    # smooth_data, R/resampler.cpp:168-230: exact neighbourhoods (the membership test sees the reference's bits), weights
    # through asin/exp (device libm vs glibc): rtol 1e-12
    xyz, tri = M.make_mesh_from_icosa(order)
    rng = np.random.default_rng(5)
    data = np.stack([synthetic.smooth_feature(xyz, 0), rng.normal(size=len(xyz)), synthetic.smooth_feature(xyz, 2) ** 2])
    m = M.Mesh(ctx, xyz, tri)
    om = O.Mesh(xyz, tri)
    got = M.smooth_data(m, data, m, sigma)
    want = O.smooth_data(om, data, om, sigma)
    assert got.shape == want.shape == (3, len(xyz))
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14), np.max(np.abs(got - want))
    assert np.std(got[1]) < 0.5 * np.std(data[1])  # it does smooth
    # with an exclusion mask: excluded centres stay 0, excluded neighbours do not contribute, the mask is smoothed too
    excl = (rng.uniform(size=len(xyz)) > 0.2).astype(float)
    got, gmask = M.smooth_data(m, data, m, sigma, excl)
    want, wmask = O.smooth_data(om, data, om, sigma, excl)
    assert np.array_equal(got == 0.0, want == 0.0)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14) and np.allclose(gmask, wmask, rtol=1e-12, atol=1e-14)
    assert np.all(got[:, excl == 0] == 0.0)


def test_smooth_data_large_neighbourhoods(ctx):
    # sigma so large that every vertex has more neighbours than one LDS pass holds (flush path), D > 64 features
    xyz, tri = M.make_mesh_from_icosa(4)
    rng = np.random.default_rng(6)
    data = rng.normal(size=(70, len(xyz)))
    m = M.Mesh(ctx, xyz, tri)
    om = O.Mesh(xyz, tri)
    assert np.allclose(M.smooth_data(m, data, m, 30.0), O.smooth_data(om, data, om, 30.0), rtol=1e-11, atol=1e-13)


def test_exclusion_masks_in_metric_resample_and_nearest_neighbour(ctx):
    """featurespace::initialise with a cut (M/featurespace.cpp:62-74): create_exclusion, metric_resample and
    nearest_neighbour_interpolation with EXCL -- masked weights, masked sums, and the mask carried to the new mesh"""
    xyz, tri = M.make_mesh_from_icosa(5)
    lo_xyz, lo_tri = M.make_mesh_from_icosa(4)
    data = synthetic.features(xyz, 3, 11)
    thr = np.quantile(data[0], 0.2)
    excl = M.create_exclusion(data, thr, 1e9)
    assert np.array_equal(excl, O.create_exclusion(data, thr, 1e9)) and 0.1 < excl.mean() < 0.9
    src, dst = M.Mesh(ctx, xyz, tri), M.Mesh(ctx, lo_xyz, lo_tri)
    osrc, odst = O.Mesh(xyz, tri), O.Mesh(lo_xyz, lo_tri)
    got, gmask = M.metric_resample(src, data, dst, excl=excl)
    want, wmask = O.metric_resample_excl(osrc, data, odst, excl)
    assert np.array_equal(got, want, equal_nan=True) and np.array_equal(gmask, wmask, equal_nan=True)
    assert np.isfinite(got).mean() > 0.5 and (gmask == 0).any() and (gmask > 0).any()
    q = synthetic.random_sphere_points(3000, seed=5)
    got, gmask = M.nearest_neighbour_interpolation(src, data, q, excl=excl)
    want, wmask = O.nearest_neighbour_excl(osrc, data, q, excl)
    assert np.array_equal(got, want) and np.array_equal(gmask, wmask) and (gmask == 0).any()
    assert np.array_equal(M.nearest_neighbour_interpolation(src, data, q), O.nearest_neighbour(osrc, data, q))


@pytest.mark.parametrize("case", ["ico6", "ico5_warped", "ico6_jittered", "ico5_radial", "ico4_small"])
def test_gpu_built_octree_has_the_reference_leaves(ctx, case):
    """The level-by-level GPU build (octree_kernels.hip) against the host build (the reference's insertion order restated,
    octree.cpp): same node / leaf / reference counts, same depth, and the same leaves -- box and ordered triangle list of
    every leaf (signature).  ico4 is below the size at which the GPU build is chosen: there both paths are the host's."""
    from newmsm_amd import synthetic

    order = int(case[3])
    xyz, tri = M.make_mesh_from_icosa(order)
    if "warped" in case:
        xyz = synthetic.known_warp(xyz, seed=3, rot_deg=5.0, amp=1.5)
    if "jittered" in case:  # folds and slivers: boxes overlap many cells, deeper splits
        rng = np.random.default_rng(5)
        xyz = xyz + rng.normal(scale=0.4, size=xyz.shape)
        xyz = xyz / np.linalg.norm(xyz, axis=1, keepdims=True) * 100.0
    if "radial" in case:
        xyz = xyz * (1.0 + 0.01 * synthetic.smooth_feature(xyz, 1, 11))[:, None]
    stats_host, sig_host = M.octree_signature(xyz, tri)
    m = M.Mesh(ctx, xyz, tri)
    stats_dev, sig_dev = m.octree_signature()
    assert stats_dev == stats_host, (stats_dev, stats_host)
    assert sig_dev == sig_host
    if case == "ico6":
        assert (stats_dev["nodes"], stats_dev["leaves"], stats_dev["depth"], stats_dev["refs"]) == (14281, 12496, 6, 176096)
    # new coordinates: the tree is rebuilt from the device copy
    xyz2 = synthetic.known_warp(xyz, seed=8, rot_deg=1.0, amp=0.5)
    m.set_coords(xyz2)
    s2, g2 = m.octree_signature()
    assert (s2, g2) == M.octree_signature(xyz2, tri)


def test_forest_build_gives_every_tree_of_the_single_builds(ctx):
    """gpu_build_forest (the gMSM set-up: a subject's data mesh rotated to every label, all trees built by one set of launches) against
    one GPU build per coordinate set: identical leaves (boxes and ordered triangle lists) for every tree, at two sizes."""
    for order, B in ((4, 5), (6, 3)):
        xyz, tri = M.make_mesh_from_icosa(order)
        sets = [xyz] + [synthetic.known_warp(xyz, seed=70 + b, rot_deg=2.0 * b, amp=0.4 * b) for b in range(1, B)]
        got = ctx.forest_signatures(sets, tri)
        want = []
        for x in sets:
            m = M.Mesh(ctx, x, tri)
            want.append(m.octree_signature()[1])
        assert got == want


@pytest.mark.parametrize("order,n", [(4, 6000), (6, 60000)])
@pytest.mark.parametrize("mode", [M.WEIGHTS_PROJECTED, M.WEIGHTS_RAW])
def test_query_through_the_direction_table(ctx, order, n, mode):
    """A target with a direction table (msm_mesh_prepare_search; a cost function's target, a group's template) answers plain queries through it from 4 096
    queries on (kernels.hip: k_query_rays, what the table cannot vouch for through k_query_open): the reference's triangles and weights bit for bit --
    random points, points on and a hair off edges and vertices, points off the radius (outside the table's shell: the complete search)."""
    xyz, tri = M.make_mesh_from_icosa(order)
    rng = np.random.default_rng(order)
    q = queries(xyz, n, seed=order + 20)
    edge = 0.5 * (xyz[tri[:2000, 0]] + xyz[tri[:2000, 1]])
    edge = edge * (100.0 / np.linalg.norm(edge, axis=1, keepdims=True))
    q[:2000] = edge + rng.normal(scale=1e-9, size=edge.shape)      # on the edges, to rounding
    q[2000:3000] = xyz[:1000] + rng.normal(scale=1e-7, size=(1000, 3))  # at the vertices
    q[3000:3200] *= 1.01                                            # off the radius
    mesh = M.Mesh(ctx, xyz, tri)
    plain = mesh.query_triangles(q, mode=mode)
    mesh.prepare_search(wait=True)
    table = mesh.query_triangles(q, mode=mode)
    for a, b in zip(plain, table):
        assert np.array_equal(a, b)
    ost, ot, ovid, ow = O.Octree(O.Mesh(xyz, tri)).barycentric_weights(q, raw=(mode == M.WEIGHTS_RAW))
    assert table[0] == 0 and np.array_equal(table[1], ot) and np.array_equal(table[2], ovid) and np.array_equal(table[3], ow)
