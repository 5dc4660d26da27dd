"""include/msmhip_config.hpp (the reference's configuration grammar for the C++ host side) against newmsm_amd/config.py on the shipped presets,
hand-written variants and malformed files.  Host logic: no GPU needed (the program links libmsmhip.so for msmhip.hpp's symbols only)."""
import json
import os
import subprocess

import pytest

from newmsm_amd import api, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "config_levels.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "config_levels")
LIBDIR = os.path.join(ROOT, "newmsm_amd")


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g

    g.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", LIBDIR, "-lmsmhip",
                           "-Wl,-rpath," + LIBDIR])
    return EXE


def run(exe, tmp_path, text, D, anat=False):
    path = tmp_path / "conf"
    path.write_text(text)
    out = subprocess.run([exe, str(path), str(D)] + (["anat"] if anat else []), capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


EXTRA = ["--opt=DISCRETE,DISCRETE\n--lambda=0.1,0.2\n--regoption=3\n--sigma_in=3,1\n",
         "--opt=DISCRETE\n--lambda=0.5\n--dopt=MCMC\n--regoption=3\n--mciters=2000\n--mcparam=0.6\n--simval=3\n--patchwise\n",
         "--opt=DISCRETE\n--lambda=0.5\n--dopt=HOCR\n--regoption=3\n--simval=4\n--percentile=0.6\n--cprange=1.5  # wider patches\n", ""]


@pytest.mark.parametrize("text", list(config.PRESETS.values()) + EXTRA)
@pytest.mark.parametrize("D", [1, 32])
def test_cpp_parser_gives_the_python_schedule(exe, tmp_path, text, D):
    anat = "--regoption=5" in text   # the aMSM preset: the caller says it has the anatomical surfaces
    got = run(exe, tmp_path, text, D, anat)
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(text), D, anat=anat)
    assert got["varnorm"] == run_kw["varnorm"] and [tuple(s) for s in got["skipped"]] == skipped and len(got["levels"]) == len(levels)
    for g, w in zip(got["levels"], levels):
        flat = dict(w, **w["cost_params"])
        flat["kind"] = api.KINDS[flat["kind"]]
        flat.setdefault("percentile", g["percentile"])  # only handed over with the DICE measures on the Python side
        del flat["cost_params"]
        assert g == flat


def test_cpp_parser_without_a_config_file(exe):
    """no --conf: the built-in sulc schedule (M/mesh_registration.cpp:627-642); an empty file is something else (zero levels, EXTRA above)"""
    out = subprocess.run([exe, "NONE", "1"], capture_output=True, text=True, timeout=60)
    got = json.loads(out.stdout.strip().splitlines()[-1])
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(None), 1)
    assert [tuple(s) for s in got["skipped"]] == skipped == [(0, "RIGID")]
    assert [(g["data_order"], g["cp_order"], g["sg_order"], g["iters"]) for g in got["levels"]] == [(w["data_order"], w["cp_order"], w["sg_order"], w["iters"]) for w in levels]


@pytest.mark.parametrize("text", ["--opt=DISCRETE,DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=3\n", "--opt=DISCRETE\n--lambda=0.1\n--triclique\n--patchwise\n--dopt=HOCR\n--regoption=3\n",
                                  "--opt=DISCRETE\n--lambda=0.1\n--percentile=1.0\n", "--opt=DISCRETE\n--lambda=0.1\n--nosuchoption=1\n", "--opt=DISCRETE\n--lambda\n",
                                  "--opt=DISCRETE\n--lambda=0.1\n--VN=1\n", "--opt=DISCRETE\n--lambda=0.1\n--dopt=HOCR\n--regoption=5\n",
                                  "--opt=DISCRETE\n--lambda=0.1\n--dopt=Simplex\n--regoption=3\n", "--opt=DISCRETE\n--lambda=abc\n"])
def test_cpp_parser_reports_the_same_errors(exe, tmp_path, text):
    got = run(exe, tmp_path, text, 1)
    with pytest.raises(config.ConfigError) as e:
        config.levels_from_config(config.parse_config(text), 1)
    assert "error" in got
    # the reference's messages word for word; the grammar errors of the two parsers differ only in quoting
    assert got["error"].replace("'", "").split(":")[-1].strip()[:30] == str(e.value).replace("'", "").replace('"', "").split(":")[-1].strip()[:30]
