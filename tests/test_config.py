"""newmsm_amd/config.py: the reference's configuration grammar (M/mesh_registration.cpp:459-784) -> level schedules.  The tests write their own
configuration text with the keys of the shipped HCP / NeuroImage2017 / basic presets (the reference tree is not read)."""
import numpy as np
import pytest

from newmsm_amd import config, registration


def f32(v):
    return float(np.float32(v))


def test_hcp_msmall_config_gives_the_msmall_schedule():
    text = """
# HCP MSMAll, final stage
--simval=2,2,2
--sigma_in=0,0,0
--sigma_ref=0,0,0
--lambda=0.00001,0.0075,0.01

--it=10,15,15
--opt=DISCRETE,DISCRETE,DISCRETE
--CPgrid=2,3,4
--SGgrid=4,5,6
--datagrid=4,5,6
--regoption=3
--regexp=2
--dopt=HOCR
--VN
--rescaleL
--triclique
--k_exponent=2
--bulkmod=1.6
--shearmod=0.4"""
    cfg = config.parse_config(text)
    assert cfg["levels"] == 3 and cfg["dopt"] == "HOCR" and cfg["regoption"] == 3 and cfg["VN"] and cfg["rescaleL"] and cfg["triclique"] and not cfg["patchwise"]
    levels, run_kw, skipped = config.levels_from_config(cfg, D=32)
    assert skipped == [] and run_kw == dict(varnorm=True)
    assert levels == registration.hcp_msmall_levels()
    for k, lv in enumerate(levels):
        assert (lv["data_order"], lv["cp_order"], lv["sg_order"], lv["iters"]) == (4 + k, 2 + k, 4 + k, (10, 15, 15)[k])
        assert lv["kind"] == "ho_multivariate" and lv["optimiser"] == "fusion" and lv["rmode"] == 3 and lv["simmeasure"] == 2 and lv["rescale_labels"]
        assert lv["sigma_in"] == 0.0 and lv["sigma_ref"] == 0.0
        p = lv["cost_params"]
        # the reference's float options: the value reaches the cost function through a float
        assert p["lambda_"] == f32((0.00001, 0.0075, 0.01)[k]) and p["mu"] == f32(0.4) and p["kappa"] == f32(1.6) and p["k_exp"] == 2.0 and p["rexp"] == 2.0
        assert p["range_"] == 1.0
    assert levels[1]["cost_params"]["lambda_"] != 0.0075  # 0.007499999832361937
    # one feature row: the univariate triclique class
    assert config.levels_from_config(cfg, D=1)[0][0]["kind"] == "ho_univariate"
    assert registration.hcp_msmall_levels((2, 3, 4))[2]["iters"] == 4


def test_defaults_and_the_affine_level():
    text = """
--sigma_in=6,6,4,2
--sigma_ref=6,6,4,2
--lambda=0,0.1,0.2,0.3
--it=50,5,10,10
--opt=AFFINE,DISCRETE,DISCRETE,DISCRETE
--CPgrid=0,2,3,4
--SGgrid=0,4,5,6
--datagrid=5,5,5,6
--regoption=1
"""
    cfg = config.parse_config(text)
    assert cfg["simval"] == [2, 2, 2, 2] and cfg["dopt"] == "FastPD" and cfg["regoption"] == 1 and not cfg["VN"]
    assert cfg["anatgrid"] == [2, 4, 5, 6] and cfg["mciters"] == [100000] * 4
    levels, run_kw, skipped = config.levels_from_config(cfg, D=1)
    assert skipped == [(0, "AFFINE")] and run_kw == dict(varnorm=False) and len(levels) == 3
    assert [lv["optimiser"] for lv in levels] == ["fastpd"] * 3 and [lv["rmode"] for lv in levels] == [1] * 3 and levels[0]["kind"] == "univariate"
    assert [lv["data_order"] for lv in levels] == [5, 5, 6] and [lv["sigma_in"] for lv in levels] == [6.0, 4.0, 2.0]
    assert levels[2]["cost_params"]["lambda_"] == f32(0.3) and not levels[0]["rescale_labels"]
    # --dopt=FastPD (the default) forces regoption 1 whatever the file says (:684); sigma_ref defaults to sigma_in; CPgrid counts up from 2; SG = CP + 2
    cfg = config.parse_config("--opt=DISCRETE,DISCRETE\n--lambda=0.1,0.2\n--regoption=3\n--sigma_in=3,1\n")
    assert cfg["regoption"] == 1 and cfg["sigma_ref"] == [3.0, 1.0] and cfg["CPgrid"] == [2, 3] and cfg["SGgrid"] == [4, 5] and cfg["datagrid"] == [5, 5]
    assert cfg["it"] == [3, 3]
    # the commented-out regoption of sMSM_PAIR_longitudinal_alignment: FastPD, pairwise
    lv, kw, _ = config.preset_levels("sMSM_PAIR", 1)
    assert kw == dict(varnorm=True) and lv[0]["optimiser"] == "fastpd" and lv[0]["rmode"] == 1 and lv[0]["rescale_labels"] and lv[0]["cost_params"]["lambda_"] == f32(0.4)
    # MCMC options, NMI replaced by correlation, DICE percentile
    cfg = config.parse_config("--opt=DISCRETE\n--lambda=0.5\n--dopt=MCMC\n--regoption=3\n--mciters=2000\n--mcparam=0.6\n--simval=3\n--patchwise\n")
    lv = config.levels_from_config(cfg, D=4)[0][0]
    assert lv["optimiser"] == "mcmc" and lv["mciters"] == 2000 and lv["mcparam"] == f32(0.6) and lv["simmeasure"] == 2 and lv["kind"] == "patchwise"
    cfg = config.parse_config("--opt=DISCRETE\n--lambda=0.5\n--dopt=HOCR\n--regoption=3\n--simval=4\n--percentile=0.6\n")
    assert config.levels_from_config(cfg, D=1)[0][0]["cost_params"]["percentile"] == f32(0.6)


def test_no_config_is_the_sulc_configuration():
    empty = config.parse_config("# nothing but a comment\n")  # an empty FILE is not "no config" (the reference tests the file name): zero levels
    assert empty["opt"] == [] and empty["levels"] == 0 and config.levels_from_config(empty, D=1)[0] == []
    cfg = config.parse_config(None)
    assert cfg["opt"] == ["RIGID", "DISCRETE", "DISCRETE", "DISCRETE"] and cfg["it"] == [50, 3, 3, 3] and cfg["datagrid"] == [4, 4, 5, 6]
    assert cfg["regoption"] == 1 and cfg["dopt"] == "FastPD"
    levels, _, skipped = config.levels_from_config(cfg, D=1)
    assert skipped == [(0, "RIGID")] and [lv["cp_order"] for lv in levels] == [2, 3, 4] and [lv["sg_order"] for lv in levels] == [4, 5, 6]


@pytest.mark.parametrize("text,message", [
    ("--opt=DISCRETE,DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=3\n", "inconsistent: --lambda"),
    ("--opt=DISCRETE\n--lambda=0.1\n--it=3,3\n--dopt=HOCR\n--regoption=3\n", "inconsistent: --it"),
    ("--opt=DISCRETE\n--lambda=0.1\n--triclique\n--patchwise\n--dopt=HOCR\n--regoption=3\n", "patchwise and triclique"),
    ("--opt=DISCRETE\n--lambda=0.1\n--percentile=1.0\n", "Percentile must be between 0 and 1."),
    ("--opt=DISCRETE\n--lambda=0.1\n--cutthr=0.5\n", "cut threshold"),
    ("--opt=DISCRETE\n--lambda=0.1\n--nosuchoption=1\n", "unrecognised option"),
    ("--opt=DISCRETE\n--lambda\n", "requires an argument"),
    ("--opt=DISCRETE\n--lambda=0.1\n--VN=1\n", "takes no argument"),
])
def test_the_reference_error_messages(text, message):
    with pytest.raises(config.ConfigError, match=message):
        config.parse_config(text)


def test_what_the_path_does_not_cover_is_reported():
    cfg = config.parse_config("--opt=DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=5\n")
    with pytest.raises(config.ConfigError, match="anatomical meshes"):
        config.levels_from_config(cfg, D=1)
    cfg = config.parse_config("--opt=DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=3\n--IN\n")
    with pytest.raises(config.ConfigError, match="histogram"):
        config.levels_from_config(cfg, D=1)
    cfg = config.parse_config("--opt=DISCRETE\n--lambda=0.1\n--dopt=Simplex\n--regoption=3\n")
    with pytest.raises(config.ConfigError, match="Unrecognized optimiser"):
        config.levels_from_config(cfg, D=1)


def test_every_shipped_preset_parses():
    for name in config.PRESETS:
        levels, run_kw, skipped = config.preset_levels(name, 1, anat=name.startswith("aMSM"))
        assert len(levels) == 3 and len(skipped) == (1 if name.startswith("standard") else 0)
    # the aMSM preset (config/NeuroImage2017_configs/aMSM_STR_longitudinal_alignment): --regoption=5 with --anatgrid=4,5,6, refused without the surfaces
    levels, _, _ = config.preset_levels("aMSM_STR", 1, anat=True)
    assert [lv["rmode"] for lv in levels] == [5, 5, 5] and [lv["anat_order"] for lv in levels] == [4, 5, 6] and levels[0]["kind"] == "ho_univariate"
    with pytest.raises(config.ConfigError, match="requires anatomical meshes"):
        config.preset_levels("aMSM_STR", 1)
    with pytest.raises(config.ConfigError, match="regoption 4 has been removed"):
        config.levels_from_config(config.parse_config("--opt=DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=4\n"), 1, anat=True)
