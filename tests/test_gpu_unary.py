"""Unary label-cost evaluation on the GPU against the oracle.

Tolerance: patch membership, AbsoluteWeights and triangle choices are exact; the cost values go through
acos/sincos (device libm vs glibc, <= 1-2 ulp) and a wavefront-parallel summation instead of the
reference's serial one, so costs are compared to rtol 1e-10 / atol 1e-12 (north_star's bar is 1e-4 rad
on final coordinates)."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem, synthetic
from oracle import oracle as O
from tests.helpers import oracle_cost

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-10, 1e-12


def run_pair(ctx, inp, kind, **kw):
    cf, keep = problem.build_cost(ctx, inp, kind=kind, **kw)
    cf.get_source_data()
    oc = oracle_cost(inp, kind, **kw)
    oc.get_source_data()
    return cf, oc, keep


@pytest.mark.parametrize("data_order,cp_order", [(4, 2), (5, 3)])
def test_source_data_exact(ctx, data_order, cp_order):
    inp = problem.pairwise_inputs(data_order, cp_order, D=1)
    cf, oc, _ = run_pair(ctx, inp, "univariate")
    ptr, idx = cf.patches()
    optr, oidx = oc.patches()
    assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx)
    assert np.array_equal(cf.absolute_weights(), oc.absolute_weights())


def test_source_data_exact_ties_regular_grids(ctx):
    # un-warped grids: every control point's farthest neighbour sits exactly on the range threshold
    inp = problem.pairwise_inputs(5, 3, D=1, warp_amp=0.0, warp_rot=0.0)
    cf, oc, _ = run_pair(ctx, inp, "univariate")
    ptr, idx = cf.patches()
    optr, oidx = oc.patches()
    assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx)


@pytest.mark.parametrize("data_order,cp_order,sim", [(4, 2, 2), (5, 3, 2), (5, 3, 1)])
def test_unary_table_univariate(ctx, data_order, cp_order, sim):
    inp = problem.pairwise_inputs(data_order, cp_order, D=1)
    cf, oc, _ = run_pair(ctx, inp, "univariate", simmeasure=sim)
    U = cf.computeUnaryCosts()
    Uo = oc.unary_table()
    assert U.shape == Uo.shape == (len(inp["labels"]), len(inp["cp_xyz"]))
    assert np.isfinite(U).all()
    assert np.allclose(U, Uo, rtol=RTOL, atol=ATOL), np.max(np.abs(U - Uo))
    # on-demand evaluation (Fusion's per-label sweep) returns the same numbers
    nodes = np.arange(0, cf.N, 7, dtype=np.int32)
    labels = (nodes * 5) % cf.L
    assert np.array_equal(cf.computeUnaryCost(nodes, labels), U[labels, nodes])


def test_unary_table_into_mapped_host_memory(ctx):
    """computeUnaryCosts(out=Context.host_array(...)): the table is written by a copy kernel straight into mapped pinned host
    memory (no staging copy); repeated calls, a table after a coordinate change, and the default path give the same numbers."""
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.get_source_data()
    plain = cf.computeUnaryCosts().copy()
    out = ctx.host_array((cf.L, cf.N))
    out[:] = -1.0
    got = cf.computeUnaryCosts(out=out)
    assert got is out and np.array_equal(out, plain)
    out[:] = -2.0
    cf.reset_source(keep["source"])  # invalidates the table: recomputed, same inputs
    cf.get_source_data()
    assert np.array_equal(cf.computeUnaryCosts(out=out), plain)
    odd = ctx.host_array((cf.L * cf.N + 1,))[1:].reshape(cf.L, cf.N)  # 8-byte aligned only: the staged route
    assert np.array_equal(cf.computeUnaryCosts(out=odd), plain)


def test_unary_table_with_cfweight_and_samples_set(ctx):
    inp = problem.pairwise_inputs(5, 3, D=1, rescale=False)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.2, 1.0, size=(1, len(inp["source_xyz"])))
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.set_dataaffintyweighting(w)
    cf.get_source_data()
    oc = oracle_cost(inp, "univariate")
    oc.set_cfweight(w)
    oc.get_source_data()
    assert np.array_equal(cf.absolute_weights(), oc.absolute_weights())
    assert np.allclose(cf.computeUnaryCosts(), oc.unary_table(), rtol=RTOL, atol=ATOL)


def test_full_size_properties_ico6(ctx):
    # BASELINE config 2 size (ico6 data / ico4 control grid): checked through size-independent properties
    inp = problem.pairwise_inputs(6, 4, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.get_source_data()
    ptr, idx = cf.patches()
    assert len(ptr) == 2563 and 40 <= np.diff(ptr).min() and np.diff(ptr).max() <= 90
    U = cf.computeUnaryCosts()
    assert U.shape == (19, 2562) and np.isfinite(U).all()
    assert (U >= -1e-12).all() and (U <= 1.0 + 1e-9).all()  # AbsW * (1 - (1 + r)/2), r in [-1, 1]
    assert np.array_equal(U, cf.computeUnaryCosts())          # deterministic
    # spot-check 40 (node,label) evaluations against the oracle at full size
    oc = oracle_cost(inp, "univariate")
    oc.get_source_data()
    rng = np.random.default_rng(1)
    for n, l in zip(rng.integers(0, 2562, 40), rng.integers(0, 19, 40)):
        assert abs(U[l, n] - oc.unary(n, l)) <= ATOL + RTOL * abs(U[l, n])
    assert cf.counters()["samples"] == 2 * 19 * int(ptr[-1])


@pytest.mark.parametrize("noise,warp", [(0.0, 2.0), (0.6, 0.0), (1.5, 0.0)])
def test_unary_table_irregular_and_folded_targets(ctx, noise, warp):
    # warp: smooth, fold-free but irregular target -> the nearest-centroid shortcut is legal but often misses;
    # noise: jittered target with slivers / folds -> the shortcut must be disabled and the reference's tie-breaks decide
    inp = problem.pairwise_inputs(5, 3, D=1, target_noise=noise, target_warp=warp)
    cf, oc, _ = run_pair(ctx, inp, "univariate")
    U = cf.computeUnaryCosts()
    Uo = oc.unary_table()
    assert np.isfinite(Uo).all()
    assert np.allclose(U, Uo, rtol=RTOL, atol=ATOL), np.max(np.abs(U - Uo))


def test_ray_table_and_general_kernel_agree(ctx, monkeypatch):
    # the same simple-surface target through both sampling kernels: identical triangles and weights, one reduction kernel
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, oc, keep = run_pair(ctx, inp, "univariate")
    U_ray = cf.computeUnaryCosts()
    monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")  # read when the target's search structures are built
    cf2, keep2 = problem.build_cost(ctx, inp, kind="univariate")
    cf2.get_source_data()
    U_gen = cf2.computeUnaryCosts()
    monkeypatch.delenv("MSMHIP_DISABLE_RAYTABLE")
    Uo = oc.unary_table()
    assert np.array_equal(U_ray, U_gen)  # one common reduction: the table does not depend on the search path
    assert np.allclose(U_ray, Uo, rtol=RTOL, atol=ATOL) and np.allclose(U_gen, Uo, rtol=RTOL, atol=ATOL)


def test_target_moved_on_the_device_gets_a_new_direction_table(ctx):
    # a target whose coordinates are rewritten on the device after its direction table was built (msm_mesh_sphere_project_warp, in place):
    # the next cost function over the same handle must search the new surface -- tree, masks and table all follow the coordinates
    from newmsm_amd import api

    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.get_source_data()
    U_before = cf.computeUnaryCosts()  # builds the table of the unwarped target
    cpx, cpt = M.make_mesh_from_icosa(3)
    to = synthetic.known_warp(cpx, seed=21, rot_deg=4.0, amp=1.5)
    target = keep["target"]
    api.sphere_project_warp_mesh(target, M.Mesh(ctx, cpx, cpt), to)
    moved = O.sphere_project_warp(inp["target_xyz"], O.Mesh(cpx, cpt), to)
    assert np.array_equal(target.get_coords(), moved)
    cf2 = api.DiscreteCostFunction(ctx, kind="univariate", simmeasure=2, rmode=3)
    keep["source"].set_coords(inp["source_orig_xyz"])
    keep["cpgrid"].set_coords(inp["cp_orig_xyz"])
    cf2.set_meshes(target, keep["source"], keep["cpgrid"])
    keep["source"].set_coords(inp["source_xyz"])
    keep["cpgrid"].set_coords(inp["cp_xyz"])
    cf2.reset_source(keep["source"])
    cf2.reset_CPgrid(keep["cpgrid"])
    cf2.set_featurespace(inp["src_feat"])
    cf2.set_spacings(inp["maxsep"], inp["mvdmax"])
    cf2.set_labels(inp["labels"], inp["rot"])
    cf2.setTriplets(inp["triplets"])
    cf2.get_source_data()
    U_after = cf2.computeUnaryCosts()
    inp2 = dict(inp, target_xyz=moved)
    oc = oracle_cost(inp2, "univariate")
    oc.get_source_data()
    Uo = oc.unary_table()
    assert np.allclose(U_after, Uo, rtol=RTOL, atol=ATOL), np.max(np.abs(U_after - Uo))
    assert not np.allclose(U_after, U_before, rtol=1e-6, atol=1e-9)


def test_ray_table_multivariate_weights_are_bit_exact_with_general_kernel(ctx, monkeypatch):
    # D > 1: the sampling kernels store (triangle, raw weights) per sample and one common kernel reduces them, so the
    # two search paths must give bit-identical tables
    inp = problem.pairwise_inputs(5, 3, D=3)
    cf, keep = problem.build_cost(ctx, inp, kind="multivariate")
    cf.get_source_data()
    U_ray = cf.computeUnaryCosts()
    monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")
    cf2, keep2 = problem.build_cost(ctx, inp, kind="multivariate")
    cf2.get_source_data()
    U_gen = cf2.computeUnaryCosts()
    monkeypatch.delenv("MSMHIP_DISABLE_RAYTABLE")
    assert np.array_equal(U_ray, U_gen)


@pytest.mark.parametrize("scale", [1.0 + 5e-5, 1.003, 0.99])
def test_sources_off_the_sphere(ctx, scale):
    # the ray table only vouches for queries within 1e-4 of radius 100; anything else must take the complete search
    # and still give the reference's answer (the octree descends with the point itself, not its direction)
    inp = problem.pairwise_inputs(4, 2, D=1)
    inp["source_xyz"] = inp["source_xyz"] * scale
    cf, oc, _ = run_pair(ctx, inp, "univariate")
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    both = np.isfinite(Uo)
    assert np.array_equal(np.isfinite(U), both)
    assert np.allclose(U[both], Uo[both], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("radial", [1e-4, 3e-3])
def test_non_spherical_star_shaped_targets(ctx, radial):
    # the target's vertices leave the sphere radially: still one triangle per ray (the ray table is built), but the
    # triangles' boxes no longer sit where the radius-100 queries descend -> leaf membership decides, as in the reference
    inp = problem.pairwise_inputs(5, 3, D=1, target_radial=radial)
    cf, oc, _ = run_pair(ctx, inp, "univariate")
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    both = np.isfinite(Uo)
    assert np.array_equal(np.isfinite(U), both) and both.mean() > 0.9
    assert np.allclose(U[both], Uo[both], rtol=RTOL, atol=ATOL), np.nanmax(np.abs(U - Uo))


def test_randomised_configurations(ctx):
    # a sweep over seeds, resolutions, warps and target shapes (regular / smoothly warped / radially perturbed / jittered)
    rng = np.random.default_rng(2024)
    for k in range(10):
        data_order = int(rng.choice([3, 4, 5]))
        cp_order = int(rng.integers(1, data_order - 1)) if data_order > 2 else 1
        shape = k % 4
        kw = dict(seed=int(rng.integers(1, 10**6)), warp_amp=float(rng.uniform(0.0, 1.2)), warp_rot=float(rng.uniform(0.0, 4.0)),
                  labeldist=float(rng.uniform(0.3, 0.7)), rescale=bool(rng.integers(0, 2)))
        if shape == 1:
            kw["target_warp"] = float(rng.uniform(0.5, 3.0))
        elif shape == 2:
            kw["target_radial"] = float(10 ** rng.uniform(-5, -2.5))
        elif shape == 3:
            kw["target_noise"] = float(rng.uniform(0.1, 1.0))
        sim = int(rng.choice([1, 2, 4]))
        inp = problem.pairwise_inputs(data_order, cp_order, D=1, **kw)
        cf, oc, _ = run_pair(ctx, inp, "univariate", simmeasure=sim)
        U, Uo = cf.computeUnaryCosts(), oc.unary_table()
        both = np.isfinite(Uo)
        assert np.array_equal(np.isfinite(U), both), (k, kw)
        assert np.allclose(U[both], Uo[both], rtol=RTOL, atol=ATOL), (k, kw, np.nanmax(np.abs(U - Uo)))


def test_larger_than_baseline_ico7(ctx):
    """one resolution above BASELINE (ico7 data: 163 842 vertices / 327 680 triangles, ico5 control grid: 10 242 control points,
    12.7 M point samples per table): searches bit-exact on a sample of queries, table spot-checked against the oracle"""
    inp = problem.pairwise_inputs(7, 5, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.get_source_data()
    ptr, idx = cf.patches()
    assert len(ptr) == 10243 and np.diff(ptr).min() >= 40
    U = cf.computeUnaryCosts()
    assert U.shape == (len(inp["labels"]), 10242) and np.isfinite(U).all()
    assert (U >= -1e-12).all() and (U <= 1.0 + 1e-9).all()
    assert cf.counters()["samples"] == len(inp["labels"]) * int(ptr[-1])
    oc = oracle_cost(inp, "univariate")
    oc.get_source_data()
    optr, oidx = oc.patches()
    assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx)
    rng = np.random.default_rng(2)
    for n, l in zip(rng.integers(0, 10242, 30), rng.integers(0, len(inp["labels"]), 30)):
        assert abs(U[l, n] - oc.unary(n, l)) <= ATOL + RTOL * abs(U[l, n])
    q = synthetic.random_sphere_points(20000, seed=4)
    st, t, vid, w = keep["target"].query_triangles(q)
    ost, ot, ovid, ow = O.Octree(O.Mesh(inp["target_xyz"], inp["target_tri"])).barycentric_weights(q)
    assert st == ost == 0 and np.array_equal(t, ot) and np.array_equal(vid, ovid) and np.array_equal(w, ow)


@pytest.mark.parametrize("kind,D", [("univariate", 1), ("multivariate", 3)])
def test_background_ray_table(ctx, monkeypatch, kind, D):
    """default mode: the first tables are served by the complete search while the direction table is built on a host thread;
    once it is there the table kernels take over, and the tables are bit-identical across the switch"""
    import time

    monkeypatch.setenv("MSMHIP_RAYTABLE", "async")
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind)
    cf.get_source_data()
    first = cf.computeUnaryCosts()
    t0 = time.time()
    while not keep["target"].prepare_search(wait=False):
        assert time.time() - t0 < 30.0
        time.sleep(0.005)
    later = cf.computeUnaryCosts()
    assert np.array_equal(first, later)
    # new coordinates: the table of the old tree is dropped, a new one is built; waiting for it is allowed too
    keep["target"].set_coords(inp["target_xyz"] * (1.0 + 1e-13))
    cf.computeUnaryCosts()
    assert keep["target"].prepare_search(wait=True)
    monkeypatch.setenv("MSMHIP_RAYTABLE", "off")
    cf2, keep2 = problem.build_cost(ctx, inp, kind=kind)
    cf2.get_source_data()
    assert np.array_equal(cf2.computeUnaryCosts(), first) and keep2["target"].prepare_search(wait=True)


def test_background_ray_table_lifetime(ctx, monkeypatch):
    """meshes that go away, or move, while their direction table is still being built"""
    monkeypatch.setenv("MSMHIP_RAYTABLE", "async")
    inp = problem.pairwise_inputs(5, 3, D=1)
    ref = None
    for rep in range(4):
        cf, keep = problem.build_cost(ctx, inp, kind="univariate")
        cf.get_source_data()
        U = cf.computeUnaryCosts()          # starts the build
        ref = U if ref is None else ref
        assert np.array_equal(U, ref)
        if rep % 2:
            keep["target"].set_coords(inp["target_xyz"])  # a new tree while the old table is in flight
            assert np.array_equal(cf.computeUnaryCosts(), ref)
        cf.close()
        for m in keep.values():
            m.close()                        # joins the build


@pytest.mark.parametrize("range_", [0.0, 0.12, 0.3])
@pytest.mark.parametrize("kind,D,sim", [("univariate", 1, 2), ("univariate", 1, 1), ("multivariate", 3, 2), ("patchwise", 3, 1), ("univariate", 1, 4)])
def test_empty_and_single_point_patches(ctx, kind, D, sim, range_):
    """range factors so small that a control point sees no source vertex (0), only the one it sits on (0.12), or a handful
    (0.3): the reference then divides by an empty patch (SSD: sqrt(0) / 0 = NaN) or correlates a single point (variance 0 ->
    r = 0); same values, NaN for NaN"""
    inp = problem.pairwise_inputs(4, 2, D=D)
    cf, oc, _ = run_pair(ctx, inp, kind, simmeasure=sim, range_=range_)
    ptr, idx = cf.patches()
    optr, oidx = oc.patches()
    assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx)
    sizes = np.diff(ptr)
    assert sizes.max() == (0 if range_ == 0.0 else 1) or (range_ == 0.3 and 1 < sizes.max() < 12)
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert np.array_equal(np.isnan(U), np.isnan(Uo))
    ok = ~np.isnan(Uo)
    assert np.allclose(U[ok], Uo[ok], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("kind,D", [("univariate", 1), ("multivariate", 4), ("patchwise", 4)])
@pytest.mark.parametrize("sg_order,labeldist,lo,hi", [(5, 0.5, 60, 120), (6, 0.5, 250, 400), (4, 0.05, 1, 1)])
def test_many_labels_and_a_single_label(ctx, kind, D, sg_order, labeldist, lo, hi):
    """label sets far from the usual 19: a finer sampling grid (≈ 80 and ≈ 300 labels: several passes of the per-label groups,
    more label ranges per control point) and a radius so small that only the centre is left"""
    inp = problem.pairwise_inputs(4, 2, D=D, sg_order=sg_order, labeldist=labeldist, rescale=False)
    assert lo <= len(inp["labels"]) <= hi, len(inp["labels"])
    cf, oc, _ = run_pair(ctx, inp, kind)
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert U.shape == Uo.shape == (len(inp["labels"]), 162)
    assert np.allclose(U, Uo, rtol=1e-9, atol=1e-11), np.abs(U - Uo).max()


def test_direction_table_is_shared_by_content(ctx):
    """a second target with the same coordinates and triangles takes the first one's direction table (kept by content in the process,
    compared in full on a hit): same costs bit for bit and no second build; a target that differs in one coordinate builds its own"""
    import time

    inp = problem.pairwise_inputs(5, 3, D=1, seed=77)
    tables, times = [], []
    for k in range(5):  # A (built), A, A, A (copies), B (built)
        txyz = inp["target_xyz"].copy()
        txyz[17] = txyz[17] * (1.0 + (3e-9 if k < 4 else 1e-9))  # off the shell by ~1e-7 mm: meshes no other test has built; the last differs
        cf, keep = problem.build_cost(ctx, dict(inp, target_xyz=txyz), kind="univariate")
        cf.get_source_data()
        t0 = time.perf_counter()
        keep["target"].prepare_search(wait=True)
        times.append(time.perf_counter() - t0)
        tables.append(cf.computeUnaryCosts().copy())
        cf.close()
    assert all(np.array_equal(tables[0], t) for t in tables[1:4])
    assert min(times[1:4]) < 0.5 * min(times[0], times[4]), times  # the copy is cheaper than either build (the best of three copies: no flake on a busy box)
    assert np.allclose(tables[4], tables[0], rtol=0, atol=1e-6) and not np.array_equal(tables[4], tables[0])
