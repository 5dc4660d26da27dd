"""DICE / genDICE similarity (sparsesimkernel::DICE, genDICE; M/similarities.cpp:201-253) in every cost class.

The measures count elements above the idx-th smallest value of each vector: integer work on bit-exact sampled
values, so the data terms agree with the oracle exactly; the triplet costs add a strain term that goes through
pow() (rtol 1e-9 as in test_gpu_cost_kinds.py)."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem
from tests.helpers import oracle_cost

pytestmark = pytest.mark.gpu


def pair(ctx, inp, kind, **kw):
    cf, keep = problem.build_cost(ctx, inp, kind=kind, **kw)
    cf.get_source_data()
    oc = oracle_cost(inp, kind, **kw)
    oc.get_source_data()
    return cf, oc, keep


@pytest.mark.parametrize("sim", [4, 5])
@pytest.mark.parametrize("percentile", [0.75, 0.31])
def test_univariate_dice_table_is_exact(ctx, sim, percentile):
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, oc, _ = pair(ctx, inp, "univariate", simmeasure=sim, percentile=percentile)
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert np.isfinite(U).all() and np.array_equal(U, Uo)
    assert len(np.unique(np.round(U / np.maximum(cf.absolute_weights()[None, :], 1e-300), 12))) > 5  # not a constant table


@pytest.mark.parametrize("sim", [4, 5])
def test_univariate_dice_general_kernel(ctx, sim, monkeypatch):
    # folded target: the three-phase kernel with its fused reduction
    inp = problem.pairwise_inputs(5, 3, D=1, target_noise=0.6)
    cf, oc, _ = pair(ctx, inp, "univariate", simmeasure=sim)
    assert np.array_equal(cf.computeUnaryCosts(), oc.unary_table())


@pytest.mark.parametrize("kind", ["multivariate", "patchwise"])
@pytest.mark.parametrize("sim", [4, 5])
def test_feature_kinds_dice(ctx, kind, sim):
    inp = problem.pairwise_inputs(4, 2, D=6)
    cf, oc, _ = pair(ctx, inp, kind, simmeasure=sim, percentile=0.6)
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert np.isfinite(U).all()
    assert np.allclose(U, Uo, rtol=1e-13, atol=0), np.max(np.abs(U - Uo))  # means over points / channels: last-bit sums


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 5)])
def test_triclique_dice(ctx, kind, D):
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, oc, _ = pair(ctx, inp, kind, simmeasure=4, rmode=3, lambda_=0.1)
    rng = np.random.default_rng(3)
    t = rng.integers(0, cf.T, 800).astype(np.int32)
    la, lb, lc = (rng.integers(0, cf.L, 800).astype(np.int32) for _ in range(3))
    got = cf.computeTripletCost(t, la, lb, lc)
    want = np.array([oc.triplet(*q) for q in zip(t, la, lb, lc)])
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=1e-9, atol=1e-11), np.max(np.abs(got - want))


def test_percentile_is_validated(ctx):
    with pytest.raises(M.MsmError, match="Percentile"):
        M.DiscreteCostFunction(ctx, kind="univariate", simmeasure=4, percentile=1.0)
    with pytest.raises(M.MsmError, match="Unknown similarity"):
        M.DiscreteCostFunction(ctx, kind="univariate", simmeasure=3)
