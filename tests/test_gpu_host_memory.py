"""Host memory the GPU touches (round 5).  The rule: copy engines and kernels only ever see pinned blocks the library allocated
itself (staging blocks of csrc/stager.cpp, msm_host_alloc) or whole pages the caller registered -- never a caller's or a local's pageable pages, which the HIP
runtime would page-lock (whole pages, device address = host address) behind the caller's back.  These tests exercise the lifetimes behind that rule:
staging blocks that must grow and rotate while several streams of a group set-up copy through them, pinned result arrays released with a label step
still queued into them, and registrations that are refused."""
import ctypes as C

import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem

pytestmark = pytest.mark.gpu


def test_staging_blocks_grow_and_rotate_under_a_group_setup(built, monkeypatch):
    """Staging blocks of 64 KB at first (MSMHIP_STAGE_MIN_KB): every upload of the set-up -- features, rotation matrices (72 V bytes per subject), search
    structures, patch row offsets -- outgrows them or finds them busy while the two set-up pipelines, their batch streams and the lanes copy through the
    contexts' blocks at once; subjects on data meshes of different sizes make the requests grow from subject to subject.  A block is never moved or freed
    while a copy may still read it (the round-4 set-up freed a context's one staging block on growth); the results equal the oracle's."""
    import test_gpu_group as T

    monkeypatch.setenv("MSMHIP_STAGE_MIN_KB", "64")
    ctx = M.Context(0)
    try:
        g, og, keep = T.build(ctx, S=5, D=3, subject_orders=[3, 4, 3, 5, 4])
        stats = ctx.staging_stats()
        assert stats["allocated"] >= 3 and stats["blocks"] == stats["allocated"], stats  # grew several times; nothing was given back
        rng = np.random.default_rng(11)
        for s, v, l in zip(rng.integers(0, 5, 40), rng.integers(0, 162, 40), rng.integers(0, g.L, 40)):
            ids, data = g.patch(s, v, l)
            oids, odata = og.patch(s, v, l)
            assert np.array_equal(ids, oids)
            assert np.allclose(data, odata, rtol=1e-10, atol=1e-11)
        p = rng.integers(0, g.P, 200).astype(np.int32)
        la, lb = rng.integers(0, g.L, 200).astype(np.int32), rng.integers(0, g.L, 200).astype(np.int32)
        got, want = g.computePairwiseCost(p, la, lb), np.array([og.pairwise(*q) for q in zip(p, la, lb)])
        both = np.isnan(want) & np.isnan(got)
        assert np.allclose(got[~both], want[~both], rtol=T.RTOL, atol=T.ATOL)
        # a second set-up reuses the blocks it has (fenced by events), it does not grow again
        before = ctx.staging_stats()["allocated"]
        g.setupCostFunction()
        assert ctx.staging_stats()["allocated"] <= before + 1
        g.close()
        del keep
    finally:
        ctx.close()


HCP = dict(rmode=3, lambda_=0.01, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("univariate", 1)])
def test_pinned_result_array_released_with_a_label_step_queued(ctx, kind, D):
    """A label step queued ahead (msm_cost_triplet_octets_prefetch) writes its costs straight into the caller's pinned array.  Releasing that array
    (msm_host_free), or closing the cost function, with the step still queued must wait for it -- the block is unmapped from the device's address space when it
    goes -- and leave the context usable: the next synchronous step equals the one of a cost function that never prefetched."""
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind=kind, **HCP)
    cf.get_source_data()
    rng = np.random.default_rng(3)
    lab = rng.integers(0, cf.L, cf.N).astype(np.int32)
    want = np.array(cf.tripletOctets(lab, 5))
    for _ in range(3):
        A = ctx.host_array((cf.T, 8))
        cf.prefetchTripletOctets(lab, 5, A)
        ctx.release_host_array(A)  # the queued kernel has finished with the block when it is unmapped
        del A
    B = ctx.host_array((cf.T, 8))
    assert np.array_equal(cf.tripletOctets(lab, 5, B), want)
    cf.prefetchTripletOctets(lab, 6, B)
    cf.close()  # a step still queued into B
    ctx.release_host_array(B)
    cf2, keep2 = problem.build_cost(ctx, inp, kind=kind, **HCP)
    cf2.get_source_data()
    assert np.array_equal(np.array(cf2.tripletOctets(lab, 5)), want)
    cf2.close()


def test_host_register_takes_whole_pages_only(ctx):
    """msm_host_register page-locks the caller's memory where it lies.  Page-locking works on whole pages and the device address of such a block is its host
    address: a range that shares its first or last page with other heap data would share that page's GPU mapping with whatever else gets page-locked there.
    Refused, with the remedy in the message; a page-aligned mapping is taken."""
    import mmap

    a = np.zeros(1 << 18)  # somewhere in the heap: neither end on a page boundary as a rule
    addr = a.ctypes.data + (8 if a.ctypes.data % 4096 == 0 else 0)
    with pytest.raises(M.MsmError) as e:
        ctx.register_host(addr, a.nbytes - 8)
    assert "whole pages" in str(e.value)
    mm = mmap.mmap(-1, 1 << 20)
    base = C.addressof(C.c_char.from_buffer(mm))
    assert base % 4096 == 0
    with pytest.raises(M.MsmError):
        ctx.register_host(base, (1 << 20) - 512)
    ctx.register_host(base, 1 << 20)
    ctx.unregister_host(base)


def test_queued_step_is_resolved_before_another_call_uses_the_context(ctx):
    """A queued label step shares the stream, the status word and the mapped flags with every other call on its context: another cost function's
    unary table, or a mesh update, between the hint and the matching call resolves it first -- the step is evaluated again, its costs are those of the
    synchronous call."""
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="ho_univariate", **HCP)
    cf.get_source_data()
    other, keep2 = problem.build_cost(ctx, problem.pairwise_inputs(4, 2, D=1), kind="univariate")
    other.get_source_data()
    rng = np.random.default_rng(4)
    lab = rng.integers(0, cf.L, cf.N).astype(np.int32)
    A = ctx.host_array((cf.T, 8))
    want = np.array(cf.tripletOctets(lab, 3))
    U0 = other.computeUnaryCosts()
    cf.prefetchTripletOctets(lab, 3, A)
    U1 = other.computeUnaryCosts()  # another cost function on the same context: the queued step is waited for and discarded first
    assert np.array_equal(U0, U1) and cf.prefetch_stats() == (0, 1)
    assert np.array_equal(cf.tripletOctets(lab, 3, A), want) and cf.prefetch_stats() == (0, 1)
    cf.prefetchTripletOctets(lab, 3, A)
    keep2["target"].set_coords(keep2["target"].get_coords())  # any mesh update on the context: a step queued before it is not taken afterwards
    assert np.array_equal(cf.tripletOctets(lab, 3, A), want) and cf.prefetch_stats() == (0, 2)
    cf.prefetchTripletOctets(lab, 3, A)
    assert np.array_equal(cf.tripletOctets(lab, 3, A), want) and cf.prefetch_stats() == (1, 2)
    cf.close()
    other.close()
    ctx.release_host_array(A)
