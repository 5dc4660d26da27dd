"""The reference's configuration-file grammar (`newmsm --conf=<file>`) -> the level schedule of newmsm_amd.registration.run_multiresolution.

    Mesh_registration::parse_reg_options       M/mesh_registration.cpp:459-784   keys, types, defaults, consistency checks
    Mesh_registration::fix_parameters_for_level  M/mesh_registration.cpp:786-817  what a level hands to the model / cost function
    Utilities::OptionParser::parse_config_file  (FSL utils; behaviour as used by the shipped configs: one `--key=value` or `--flag` per
                                                line, `#` comments, blank lines -- config/NeuroImage2017_configs/sMSM_PAIR_longitudinal_alignment:13)

Host logic only (no GPU, no library call).  Values whose option type is `float` in the reference (--lambda, --sigma_in, --sigma_ref, --cutthr,
--shearmod, --bulkmod, --k_exponent, --regexp, --cprange, --stepsize, --gradsampling, --mcparam, --percentile) are rounded to float32 before
they become doubles, as `Utilities::Option<float>` / `std::vector<float>` do there: --lambda=0.0075 reaches the cost function as
0.007499999832361937, --shearmod=0.4 as 0.4000000059604645.

Out of scope here, reported instead of silently dropped: AFFINE / RIGID levels (`--opt=AFFINE,...`: the affine stage is not part of the path;
`levels_from_config` lists them in `skipped`), --IN / --INc (FSL's histogram matching is not in the reference tree), --excl.  --regoption=5 (aMSM)
needs the anatomical surfaces, which come from the command line (--inanat / --refanat): `levels_from_config(cfg, D, anat=True)` says the caller has them.
"""
import numpy as np

EPSILON = 1e-8  # R/point.h:31


class ConfigError(ValueError):
    """parse_reg_options' MeshregException / X_OptionError"""


def _f32(v):
    return float(np.float32(float(v)))


_INT_LIST = ("simval", "it", "datagrid", "CPgrid", "SGgrid", "anatgrid", "mciters")
_FLOAT_LIST = ("sigma_in", "sigma_ref", "lambda", "cutthr")
_STR_LIST = ("opt",)
_INT = ("regoption", "numthreads")
_FLOAT = ("shearmod", "bulkmod", "k_exponent", "regexp", "cprange", "stepsize", "gradsampling", "mcparam", "percentile")
_STR = ("dopt",)
_FLAG = ("triclique", "patchwise", "fixnan", "rescaleL", "IN", "INc", "VN", "excl")
# defaults of the scalar options, M/mesh_registration.cpp:505-563 (float ones as the float literal they are there)
_DEFAULT = dict(regoption=1, dopt="FastPD", shearmod=_f32(0.4), bulkmod=_f32(1.6), k_exponent=2.0, regexp=2.0, cprange=1.0, stepsize=_f32(0.01),
                gradsampling=0.5, mcparam=_f32(0.8), percentile=0.75, numthreads=1, cutthr=[0.0, _f32(0.0001)])


def parse_config(text):
    """`text`: the contents of a configuration file, or None when no --conf was given.  Returns the options as parse_reg_options holds them
    after its defaults and checks (per-level lists have one entry per resolution level); raises ConfigError -- the consistency checks with
    the reference's messages, the grammar errors (which FSL's option parser reports in its own words) with the line number.
    The reference branches on the file NAME being empty (M/mesh_registration.cpp:627): only `None` selects the built-in sulc schedule; a
    file that is empty or holds comments only goes through the other branch and yields zero levels.  --regoption is passed through as
    given (2 is documented as an alias of 3 in the help text but nothing in parse_reg_options maps it; the cost function treats 2 and 3
    alike, M/DiscreteCostFunction.cpp:158-160)."""
    raw = {}
    no_config = text is None
    for lineno, line in enumerate((text or "").splitlines(), 1):
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        if not line.startswith("--"):
            raise ConfigError("line %d: expected --key=value or --flag, got %r" % (lineno, line))
        key, eq, value = line[2:].partition("=")
        key, value = key.strip(), value.strip()
        known = _INT_LIST + _FLOAT_LIST + _STR_LIST + _INT + _FLOAT + _STR + _FLAG
        if key not in known:
            raise ConfigError("line %d: unrecognised option --%s" % (lineno, key))
        if key in _FLAG:
            if eq:
                raise ConfigError("line %d: --%s takes no argument" % (lineno, key))
            raw[key] = True
            continue
        if not eq or value == "":
            raise ConfigError("line %d: --%s requires an argument" % (lineno, key))
        try:
            if key in _INT_LIST:
                raw[key] = [int(v) for v in value.split(",")]
            elif key in _FLOAT_LIST:
                raw[key] = [_f32(v) for v in value.split(",")]
            elif key in _STR_LIST:
                raw[key] = [v.strip() for v in value.split(",")]
            elif key in _INT:
                raw[key] = int(value)
            elif key in _FLOAT:
                raw[key] = _f32(value)
            else:
                raw[key] = value
        except ValueError:
            raise ConfigError("line %d: cannot read the value of --%s: %r" % (lineno, key, value))
    cfg = dict(_DEFAULT)
    cfg.update({k: False for k in _FLAG})
    cfg.update(raw)
    if no_config:  # no config: the sulc configuration of September 2014 (M/mesh_registration.cpp:629-642)
        cfg.update(opt=["RIGID", "DISCRETE", "DISCRETE", "DISCRETE"], **{"lambda": [0.0, _f32(0.1), _f32(0.2), _f32(0.3)]}, simval=[1, 2, 2, 2],
                   sigma_in=[2.0, 2.0, 3.0, 2.0], sigma_ref=[2.0, 2.0, 1.5, 1.0], it=[50, 3, 3, 3], CPgrid=[0, 2, 3, 4], anatgrid=[0, 4, 5, 6],
                   datagrid=[4, 4, 5, 6], SGgrid=[0, 4, 5, 6])
    else:
        cost = cfg.setdefault("opt", [])
        n = len(cost)
        cfg.setdefault("lambda", [])
        cfg["simval"] = [2 if v == 3 else v for v in cfg.get("simval", [2] * n)]  # NMI (3) was removed: Pearson's correlation instead (:648-653)
        cfg.setdefault("it", [3] * n)
        cfg.setdefault("sigma_in", [2.0] * n)
        cfg.setdefault("sigma_ref", list(cfg["sigma_in"]))
        cfg.setdefault("datagrid", [5] * n)
        cfg.setdefault("CPgrid", [2 + i for i in range(n)])
        cfg.setdefault("anatgrid", [g + 2 for g in cfg["CPgrid"][:n]] + [2] * max(0, n - len(cfg["CPgrid"])))
        cfg.setdefault("SGgrid", [g + 2 for g in cfg["CPgrid"][:n]] + [0] * max(0, n - len(cfg["CPgrid"])))
    n = len(cfg["opt"])
    cfg.setdefault("mciters", [100000] * n)
    if cfg["dopt"] == "FastPD":
        cfg["regoption"] = 1  # :684
    if cfg["regoption"] > 1 and cfg["dopt"] == "FastPD":  # unreachable after the line above, kept in the reference's order (:759-760)
        raise ConfigError("MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ")
    if len(cfg["cutthr"]) != 2:
        raise ConfigError("MeshREG ERROR:: the cut threshold does not contain a limit for upper and lower threshold (too few inputs)")
    for key, flag in (("simval", "--simval"), ("it", "--it"), ("sigma_in", "--sigma_in"), ("sigma_ref", "--sigma_ref"), ("lambda", "--lambda"),
                      ("datagrid", "--datagrid"), ("CPgrid", "--CPgrid"), ("SGgrid", "--SGres")):
        if len(cfg[key]) != n:
            raise ConfigError("MeshREG ERROR:: config file parameter list lengths are inconsistent: " + flag)
    if cfg["patchwise"] and cfg["triclique"]:
        raise ConfigError("Cannot use patchwise and triclique options together. Choose one.")
    if cfg["percentile"] < 0.0 + EPSILON or cfg["percentile"] > 1.0 - EPSILON:
        raise ConfigError("Percentile must be between 0 and 1.")
    cfg["levels"] = n
    return cfg


def levels_from_config(cfg, D, anat=False, groupwise=False):
    """The DISCRETE levels of `cfg` (parse_config's result) for data with D feature rows, as keyword sets of run_multiresolution, plus what
    applies to the whole run: returns (levels, run_kw, skipped) -- run_multiresolution(ops, ..., levels, **run_kw).  skipped: the (index,
    method) of levels that are not DISCRETE (the affine stage is outside the path).  fix_parameters_for_level + NonLinearSRegDiscreteModel::
    set_parameters / initialize_cost_function (M/mesh_registration.cpp:786-817, M/DiscreteModel.cpp:26-60).  groupwise: the levels of a --groupwise
    run (group_registration.run_group_multiresolution), whose model has its own regulariser (strain triplets per subject, M/DiscreteGroupCostFunction.cpp:
    26-52) whatever --regoption says: the checks on --regoption do not apply (Group_Mesh_registration::initialize_level has none)."""
    if cfg["IN"] or cfg["INc"]:
        raise ConfigError("--IN / --INc (histogram matching through FSL's MISCMATHS::Histogram, M/reg_tools.cpp:745-802) is not available")
    if cfg["excl"]:
        raise ConfigError("--excl (exclusion masks from the cut thresholds) is not wired into run_multiresolution")
    if groupwise:
        pass
    elif cfg["regoption"] == 4:  # M/mesh_registration.cpp:101-102
        raise ConfigError("--regoption 4 has been removed from newMSM. Use --regoption 3 for spherical mesh regularisation or --regoption 5 for anatomical mesh "
                          "regularisation.")
    elif cfg["regoption"] == 5 and not anat:  # :103-104: the anatomical meshes come from the command line (--inanat / --refanat), `anat` says they are there
        raise ConfigError("--regoption 5 requires anatomical meshes. Use --regoption 3 for spherical mesh regularisation or provide anatomical meshes.")
    multivariate = D > 1
    if multivariate:  # initialize_cost_function, M/DiscreteModel.cpp:44-58
        kind = "patchwise" if cfg["patchwise"] else ("ho_multivariate" if cfg["triclique"] else "multivariate")
    else:
        kind = "ho_univariate" if cfg["triclique"] else "univariate"
    optimiser = {"HOCR": "fusion", "MCMC": "mcmc", "FastPD": "fastpd"}.get(cfg["dopt"])
    if optimiser is None:
        raise ConfigError("Unrecognized optimiser")  # M/mesh_registration.cpp:202
    rmode = cfg["regoption"]
    if optimiser != "fastpd" and rmode == 1 and not groupwise:
        raise ConfigError("--regoption=1 (pairwise regulariser) is driven by FastPD only in the reference; Fusion / MCMC read triplets")
    levels, skipped = [], []
    for i, method in enumerate(cfg["opt"]):
        if method != "DISCRETE":
            skipped.append((i, method))
            continue
        params = dict(lambda_=cfg["lambda"][i], mu=cfg["shearmod"], kappa=cfg["bulkmod"], k_exp=cfg["k_exponent"], rexp=cfg["regexp"], range_=cfg["cprange"])
        if cfg["simval"][i] in (4, 5):
            params["percentile"] = cfg["percentile"]
        levels.append(dict(data_order=cfg["datagrid"][i], cp_order=cfg["CPgrid"][i], sg_order=cfg["SGgrid"][i], sigma_in=cfg["sigma_in"][i],
                           sigma_ref=cfg["sigma_ref"][i], iters=cfg["it"][i], mciters=cfg["mciters"][i], mcparam=cfg["mcparam"], kind=kind,
                           simmeasure=cfg["simval"][i], rmode=rmode, rescale_labels=cfg["rescaleL"], optimiser=optimiser, cost_params=params,
                           anat_order=cfg["anatgrid"][i] if i < len(cfg["anatgrid"]) else cfg["CPgrid"][i] + 2))
    return levels, dict(varnorm=cfg["VN"]), skipped


# The shipped configurations the BASELINE configs name, as text (the files themselves live in the reference tree, which is not available at run
# time): same keys and values as config/HCP_multimodal_alignment/MSMAllStrainFinalconf1to1_1to3_2, config/NeuroImage2017_configs/
# sMSM_STR_longitudinal_alignment, config/NeuroImage2017_configs/aMSM_STR_longitudinal_alignment (--regoption=5: needs the anatomical surfaces),
# config/NeuroImage2017_configs/sMSM_PAIR_longitudinal_alignment (whose --regoption line is commented out: FastPD,
# regoption 1) and config/basic_configs/config_standard_MSM_strain / config_standard_MSMpair.
PRESETS = {
    "HCP_MSMAll": """
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
--shearmod=0.4
""",
    "sMSM_STR": """
--simval=2,2,2
--sigma_in=6,4,2
--sigma_ref=6,4,2
--lambda=0.025,0.025,0.025
--it=40,40,40
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
--shearmod=0.4
""",
    "sMSM_PAIR": """
--simval=2,2,2
--sigma_in=6,4,2
--sigma_ref=6,4,2
--lambda=0.4,0.4,0.4
--it=40,40,40
--opt=DISCRETE,DISCRETE,DISCRETE
--CPgrid=2,3,4
--SGgrid=4,5,6
--datagrid=4,5,6
--rescaleL
--VN

#--regoption=1
""",
    "aMSM_STR": """
--simval=2,2,2
--sigma_in=6,4,2
--sigma_ref=6,4,2
--lambda=0.025,0.025,0.025
--it=40,40,40
--opt=DISCRETE,DISCRETE,DISCRETE
--CPgrid=2,3,4
--SGgrid=4,5,6
--datagrid=4,5,6
--anatgrid=4,5,6
--regoption=5
--regexp=2
--dopt=HOCR
--VN
--rescaleL
--triclique
--k_exponent=2
--bulkmod=1.6
--shearmod=0.4
""",
    "standard_MSM_strain": """
--simval=2,2,2,2
--sigma_in=2,4,2,1
--sigma_ref=2,4,2,1
--lambda=0,0.2,0.2,0.2
--it=50,20,25,25
--opt=AFFINE,DISCRETE,DISCRETE,DISCRETE
--CPgrid=0,2,3,4
--SGgrid=0,4,5,6
--datagrid=5,5,5,6
--regoption=3
--regexp=2
--dopt=HOCR
--VN
--k_exponent=2
--bulkmod=1.6
--shearmod=0.4
--rescaleL
""",
    "standard_MSMpair": """
--sigma_in=6,6,4,2
--sigma_ref=6,6,4,2
--lambda=0,0.1,0.2,0.3
--it=50,5,10,10
--opt=AFFINE,DISCRETE,DISCRETE,DISCRETE
--CPgrid=0,2,3,4
--SGgrid=0,4,5,6
--datagrid=5,5,5,6
--regoption=1
""",
}


def preset_levels(name, D, iterations=None, anat=False):
    """(levels, run_kw, skipped) of a shipped configuration; iterations (optional): overrides --it of the DISCRETE levels, in order; anat: the caller
    has the anatomical surfaces a --regoption=5 preset needs"""
    levels, run_kw, skipped = levels_from_config(parse_config(PRESETS[name]), D, anat=anat)
    if iterations is not None:
        for lv, it in zip(levels, iterations):
            lv["iters"] = it
    return levels, run_kw, skipped
