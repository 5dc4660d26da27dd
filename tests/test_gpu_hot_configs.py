"""The kernels the BASELINE configurations actually run, at the sizes they run them, against the oracle.

config 3  HCP MSMAll (config/HCP_multimodal_alignment/MSMAllStrainFinalconf1to1_1to3_2: --triclique, HOCR, regoption 3,
          shearmod 0.4, bulkmod 1.6, k_exponent 2, regexp 2): HOMultivariate triplet_likelihood
          (M/DiscreteCostFunction.cpp:565-618) called 8 x T times per label step by Fusion (I/Fusion/Fusion.h:181-196).
          That is the fused fusion-move kernel k_ho_move<., 2> (move_kernels.hip), with the tail kernel on demand.
config 4  NeuroImage2017 sMSM_STR (same options, one feature): HOUnivariate, k_ho_move<., 0>.
coarse    control grids whose bins exceed 128 points keep the three-kernel path k_ho_octets_sample / _fix / _reduce.
patchwise the PatchwiseMultivariate class with 16 and 48 feature rows (k_unary_reduce_pw8<4> / <8>).
gMSM      8 subjects at ico5 data / ico3 control grid, including a whole label step of Fusion.

Each fusion move is checked on the direction-table path and with MSMHIP_DISABLE_RAYTABLE=1 (the complete search)."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import problem
from tests.helpers import oracle_cost

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-11
HCP = dict(rmode=3, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)  # --shearmod --bulkmod --k_exponent --regexp of the two configs


def close(got, want):
    return np.allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True)


def move_labelings(cf, seed):
    """two labelings a fusion sweep meets: all control points still on the centre label, and a mixed one"""
    rng = np.random.default_rng(seed)
    return [(np.zeros(cf.N, dtype=np.int32), int(rng.integers(1, cf.L))), (rng.integers(0, cf.L, cf.N).astype(np.int32), int(rng.integers(0, cf.L)))]


def check_moves(cf, oc, triplets, seed, full):
    """tripletOctets vs Fusion.h:188-195 replayed on the oracle: every triplet (full) or >= 200 of them, all 8 combinations"""
    for labeling, label in move_labelings(cf, seed):
        E = cf.tripletOctets(labeling, label)
        assert E.shape == (cf.T, 8) and np.isfinite(E).all()
        if full:
            want = oc.triplet_octets(labeling, label, threads=8)
            assert close(E, want), np.abs(E - want).max()
            folded = want >= 1e6 * oc.params.lambda_
            assert np.array_equal(E >= 1e6 * oc.params.lambda_, folded)
        else:
            rng = np.random.default_rng(seed + 1)
            for t in rng.choice(cf.T, 240, replace=False):
                ids = triplets[t]
                for k in range(8):
                    lab = [label if k >> (2 - j) & 1 else int(labeling[ids[j]]) for j in range(3)]
                    w = oc.triplet(int(t), *lab)
                    assert abs(E[t, k] - w) <= ATOL + RTOL * abs(w), (t, k, E[t, k], w)


def ho_pair(ctx, inp, kind, lam):
    cf, keep = problem.build_cost(ctx, inp, kind=kind, lambda_=lam, **HCP)
    cf.get_source_data()
    oc = oracle_cost(inp, kind, lambda_=lam, **HCP)
    oc.get_source_data()
    return cf, oc, keep


@pytest.mark.parametrize("search", ["raytable", "complete"])
def test_config3_msmall_fusion_move_d32_ico5_full(ctx, monkeypatch, search):
    if search == "complete":
        monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")  # read when the target's search structures are built
    inp = problem.pairwise_inputs(5, 3, D=32)
    cf, oc, _ = ho_pair(ctx, inp, "ho_multivariate", 0.0075)  # --lambda of the ico3 level
    assert np.array_equal(cf.patches()[1], oc.patches()[1])
    check_moves(cf, oc, inp["triplets"], seed=31, full=True)


@pytest.mark.parametrize("search", ["raytable", "complete"])
def test_config3_msmall_fusion_move_d32_ico6(ctx, monkeypatch, search):
    """BASELINE config 3 at its last level: ico6 data, ico4 control grid, 32 features, 40 960 evaluations per move"""
    if search == "complete":
        monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")
    inp = problem.pairwise_inputs(6, 4, D=32)
    cf, oc, _ = ho_pair(ctx, inp, "ho_multivariate", 0.01)
    assert cf.T == 5120 and cf.L == 19
    check_moves(cf, oc, inp["triplets"], seed=32, full=False)
    # the whole move once, against the oracle's OpenMP replay of Fusion's loop
    labeling, label = move_labelings(cf, 33)[1]
    E, want = cf.tripletOctets(labeling, label), oc.triplet_octets(labeling, label, threads=8)
    assert close(E, want), np.abs(E - want).max()
    # evaluateTotalCostSum of the HO class: unary part 0, triplet part = column 000 of a move summed in triplet order
    tot, parts = cf.evaluateTotalCostSum(labeling)
    assert parts[0] == 0.0 and abs(parts[2] - want[:, 0].sum()) <= 1e-9 * abs(parts[2])


@pytest.mark.parametrize("search", ["raytable", "complete"])
def test_config4_smsm_str_fusion_move_univariate_ico6(ctx, monkeypatch, search):
    """BASELINE config 4 (NeuroImage2017 sMSM_STR) at its last level: HOUnivariate, ico6 / ico4"""
    if search == "complete":
        monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")
    inp = problem.pairwise_inputs(6, 4, D=1)
    cf, oc, _ = ho_pair(ctx, inp, "ho_univariate", 0.025)
    check_moves(cf, oc, inp["triplets"], seed=41, full=True)


def test_config4_levels_ico4_and_ico5(ctx):
    # the first two levels of the same configuration (CPgrid 2,3 / datagrid 4,5): bins of ~8 points as well
    for data_order, cp_order in [(4, 2), (5, 3)]:
        inp = problem.pairwise_inputs(data_order, cp_order, D=1)
        cf, oc, _ = ho_pair(ctx, inp, "ho_univariate", 0.025)
        check_moves(cf, oc, inp["triplets"], seed=42 + cp_order, full=True)


def test_fusion_move_warped_target_and_ssd(ctx):
    # an irregular (still simple) target and the SSD measure through the same three kernels, D = 16
    inp = problem.pairwise_inputs(5, 3, D=16, target_warp=0.8)
    cf, keep = problem.build_cost(ctx, inp, kind="ho_multivariate", simmeasure=1, lambda_=0.01, **HCP)
    cf.get_source_data()
    oc = oracle_cost(inp, "ho_multivariate", simmeasure=1, lambda_=0.01, **HCP)
    oc.get_source_data()
    check_moves(cf, oc, inp["triplets"], seed=51, full=True)


@pytest.mark.parametrize("D", [16, 48])
@pytest.mark.parametrize("sim", [2, 1])
def test_patchwise_wide_features(ctx, D, sim):
    """PatchwiseMultivariate (M/DiscreteCostFunction.cpp:652-692) with 16 and 48 rows: k_unary_reduce_pw8<4> / <8>"""
    inp = problem.pairwise_inputs(5, 3, D=D)
    cf, keep = problem.build_cost(ctx, inp, kind="patchwise", simmeasure=sim)
    cf.get_source_data()
    oc = oracle_cost(inp, "patchwise", simmeasure=sim)
    oc.get_source_data()
    U, Uo = cf.computeUnaryCosts(), oc.unary_table(threads=8)
    assert np.isfinite(U).all() and close(U, Uo), np.abs(U - Uo).max()


def test_patchwise_d32_ico6_spot(ctx):
    inp = problem.pairwise_inputs(6, 4, D=32)
    cf, keep = problem.build_cost(ctx, inp, kind="patchwise")
    cf.get_source_data()
    oc = oracle_cost(inp, "patchwise")
    oc.get_source_data()
    U = cf.computeUnaryCosts()
    rng = np.random.default_rng(6)
    for n, l in zip(rng.integers(0, cf.N, 40), rng.integers(0, cf.L, 40)):
        w = oc.unary(int(n), int(l))
        assert abs(U[l, n] - w) <= ATOL + RTOL * abs(w)


def test_multivariate_d32_full_table_ico5(ctx):
    """the non-HO multivariate class with 32 rows (k_unary_reduce_mv8), whole table.  (HCP MSMAll itself runs --triclique:
    its unary cost is 0 and the similarity goes through the triplets, see the config 3 tests above.)"""
    inp = problem.pairwise_inputs(5, 3, D=32)
    cf, keep = problem.build_cost(ctx, inp, kind="multivariate")
    cf.get_source_data()
    oc = oracle_cost(inp, "multivariate")
    oc.get_source_data()
    U, Uo = cf.computeUnaryCosts(), oc.unary_table(threads=8)
    assert close(U, Uo), np.abs(U - Uo).max()


@pytest.mark.parametrize("path", ["raytable", "complete"])
def test_univariate_reads_the_first_feature_row(ctx, monkeypatch, path):
    """A univariate cost on multi-row features uses row 1 only (get_source_data :343-347 reads get_input_val(1, i),
    get_target_data :371 get_ref_val(1, n)): the table equals that of the one-row problem, on both search paths."""
    if path == "complete":
        monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")
    inp = problem.pairwise_inputs(4, 2, D=2)
    cf, keep = problem.build_cost(ctx, inp, kind="univariate")
    cf.get_source_data()
    oc = oracle_cost(inp, "univariate")
    oc.get_source_data()
    U, Uo = cf.computeUnaryCosts(), oc.unary_table()
    assert np.allclose(U, Uo, rtol=1e-10, atol=1e-12), np.abs(U - Uo).max()
    one = dict(inp, ref_feat=inp["ref_feat"][:1], src_feat=inp["src_feat"][:1], D=1)
    cf1, keep1 = problem.build_cost(ctx, one, kind="univariate")
    cf1.get_source_data()
    assert np.array_equal(cf1.computeUnaryCosts(), U)
    # the triclique univariate class as well
    ch, keeph = problem.build_cost(ctx, inp, kind="ho_univariate", lambda_=0.025, **HCP)
    ch.get_source_data()
    oh = oracle_cost(inp, "ho_univariate", lambda_=0.025, **HCP)
    oh.get_source_data()
    check_moves(ch, oh, inp["triplets"], seed=71, full=True)


@pytest.mark.parametrize("sim", [4, 5, 1])
def test_fusion_move_univariate_dice_and_ssd(ctx, sim):
    """the triclique univariate class with DICE / genDICE (rank thresholds over a bin's values: serial in the reduction of the
    fused kernel) and SSD, whole label steps against the oracle's replay"""
    inp = problem.pairwise_inputs(5, 3, D=1)
    cf, keep = problem.build_cost(ctx, inp, kind="ho_univariate", simmeasure=sim, lambda_=0.05, **HCP)
    cf.get_source_data()
    oc = oracle_cost(inp, "ho_univariate", simmeasure=sim, lambda_=0.05, **HCP)
    oc.get_source_data()
    check_moves(cf, oc, inp["triplets"], seed=81 + sim, full=True)


@pytest.mark.parametrize("cp_order,kind,D", [(1, "ho_univariate", 1), (0, "ho_univariate", 1), (1, "ho_multivariate", 16)])
def test_fusion_move_coarse_control_grid(ctx, cp_order, kind, D):
    """first levels of a multiresolution run with a fine data grid: 128 / 512 source vertices per control triangle -- beyond what a
    workgroup of the fused kernel holds, so the three-kernel path (sample / fix up / reduce, clique_kernels.hip) evaluates the move"""
    inp = problem.pairwise_inputs(5, cp_order, D=D, warp_amp=0.3, warp_rot=1.0)
    cf, oc, _ = ho_pair(ctx, inp, kind, 0.05)
    assert np.diff(cf.patches()[0]).max() > 128
    check_moves(cf, oc, inp["triplets"], seed=91 + cp_order, full=True)


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 32)])
@pytest.mark.parametrize("search", ["raytable", "complete"])
def test_total_cost_of_the_triclique_classes(ctx, monkeypatch, kind, D, search):
    """evaluateTotalCostSum (M/DiscreteCostFunction.cpp:55-77) of the HO classes: unary part 0, triplet part = computeTripletCost of every
    control triangle at the labeling -- on the direction-table path the fused move kernel with ONE combination per triangle, otherwise the
    general on-demand kernel; both against the oracle, and against column 000 of a whole move"""
    if search == "complete":
        monkeypatch.setenv("MSMHIP_DISABLE_RAYTABLE", "1")
    inp = problem.pairwise_inputs(5, 3, D=D, labeldist=1.2)  # long label moves: some proposals fold (1e7 x lambda)
    cf, oc, _ = ho_pair(ctx, inp, kind, 0.0075)
    oc.set_pairs(np.zeros((0, 2), dtype=np.int32))
    rng = np.random.default_rng(71)
    for labeling in (np.zeros(cf.N, dtype=np.int32), rng.integers(0, cf.L, cf.N).astype(np.int32)):
        tot, parts = cf.evaluateTotalCostSum(labeling)
        otot, oparts = oc.total(labeling)
        assert parts[0] == 0.0 and parts[1] == 0.0
        assert abs(parts[2] - oparts[2]) <= 1e-9 * abs(oparts[2]) and abs(tot - otot) <= 1e-9 * abs(otot), (parts, oparts)
        E = cf.tripletOctets(labeling, 3)
        assert abs(parts[2] - E[:, 0].sum()) <= 1e-9 * abs(parts[2])
    assert (E[:, 0] >= 1e6 * 0.0075).any()  # the random labeling folds some control triangles
