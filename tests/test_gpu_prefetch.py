"""msm_cost_triplet_octets_prefetch: the next label step of Fusion::optimize queued while the host solves the current one (I/Fusion/Fusion.h:181-221: the
binary solve between two steps leaves the GPU idle, and most steps of a converging level change no label).  A hint: whatever happens to it, the costs a
call returns are those of the synchronous call."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem, registration, synthetic

pytestmark = pytest.mark.gpu
HCP = dict(rmode=3, lambda_=0.01, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 16), ("univariate", 1)])
def test_prefetched_step_equals_the_synchronous_one(ctx, kind, D):
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind, **HCP)
    cf.get_source_data()
    rng = np.random.default_rng(2)
    lab = rng.integers(0, cf.L, cf.N).astype(np.int32)
    A, B = ctx.host_array((cf.T, 8)), ctx.host_array((cf.T, 8))
    want4, want7 = np.array(cf.tripletOctets(lab, 4)), np.array(cf.tripletOctets(lab, 7))
    assert cf.prefetch_stats() == (0, 0)
    # taken: the same labeling, label and array
    A[:] = -1.0
    cf.prefetchTripletOctets(lab, 4, A)
    got = cf.tripletOctets(lab, 4, A)
    assert got is A and np.array_equal(A, want4) and cf.prefetch_stats() == (1, 0)
    # dropped: another label is asked for
    cf.prefetchTripletOctets(lab, 4, A)
    assert np.array_equal(cf.tripletOctets(lab, 7, B), want7) and cf.prefetch_stats() == (1, 1)
    # dropped: the labeling changed in between
    lab2 = lab.copy()
    lab2[5] = (lab2[5] + 1) % cf.L
    cf.prefetchTripletOctets(lab, 4, A)
    want = np.array(cf.tripletOctets(lab2, 4))           # (another array: dropped as well)
    assert cf.prefetch_stats() == (1, 2)
    cf.prefetchTripletOctets(lab, 4, A)
    assert np.array_equal(cf.tripletOctets(lab2, 4, A), want) and cf.prefetch_stats() == (1, 3)
    # dropped by any other entry point: a total cost, new labels
    cf.prefetchTripletOctets(lab, 4, A)
    tot = cf.evaluateTotalCostSum(lab)[0]
    assert np.isfinite(tot) and cf.prefetch_stats() == (1, 4)
    cf.prefetchTripletOctets(lab, 4, A)
    cf.set_labels(inp["labels"] * 1.0, inp["rot"])
    assert cf.prefetch_stats() == (1, 5)
    assert np.array_equal(cf.tripletOctets(lab, 4, A), want4)
    # ignored: the array is not pinned memory of the context
    cf.prefetchTripletOctets(lab, 7, np.zeros((cf.T, 8)))
    assert cf.prefetch_stats() == (1, 5)
    assert np.array_equal(cf.tripletOctets(lab, 7, B), want7) and cf.prefetch_stats() == (1, 5)
    # a chain of steps as the loop makes them: prefetch the next while "solving", take it when nothing changed
    # (the comparison values first: another cost function's call on the same context resolves a queued step -- it shares the stream, the status word and
    # the flags with it --, which would turn every hit below into a drop)
    wants = [cf_plain(ctx, inp, kind, lab, step) for step in range(6)]
    for step in range(6):
        out, nxt = (A, B) if step % 2 == 0 else (B, A)
        got = cf.tripletOctets(lab, step, out)
        cf.prefetchTripletOctets(lab, step + 1, nxt)
        assert np.array_equal(got, wants[step])
    assert cf.prefetch_stats()[0] == 1 + 5
    cf.close()


_plain = {}


def cf_plain(ctx, inp, kind, lab, label):
    """the same step from a cost function that never prefetches"""
    key = (id(inp), kind)
    if key not in _plain:
        c, keep = problem.build_cost(ctx, inp, kind=kind, **HCP)
        c.get_source_data()
        _plain.clear()
        _plain[key] = (c, keep)
    return np.array(_plain[key][0].tripletOctets(lab, label))


def test_speculating_level_equals_the_plain_one(ctx):
    """the level loop with and without the hint: identical labelings, energies and spheres; most steps are taken from a prefetch"""
    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, 8, 33)
    src = synthetic.features(synthetic.known_warp(xyz, seed=35, rot_deg=4.0, amp=2.5), 8, 33)
    kw = dict(cp_order=2, iters=3, seed=5, kind="ho_multivariate", rescale_labels=True, cost_params=dict(lambda_=0.01, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0), optimiser="fusion")
    t = {}
    a = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, timings=t, **kw)
    b = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, speculate=False, **kw)
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3])) and a[2] == b[2] and np.array_equal(a[0], b[0])
    assert "fusion_prefetch" in t
