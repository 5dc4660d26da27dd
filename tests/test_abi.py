"""The C-ABI library loads without a GPU and exports exactly what include/msmhip.h declares."""
import os
import re

import pytest

import newmsm_amd as M
from newmsm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "msmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built):
    L = M.lib()
    names = declared_symbols()
    assert len(names) >= 50
    for n in names:
        assert hasattr(L, n), "libmsmhip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert L.msm_abi_version() == 11


def test_no_cpu_fallback(built):
    if M.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(M.MsmError) as e:
        M.Context(0)
    assert e.value.code == -5


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "newmsm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "msm_oracle" not in src and "orc_" not in src and "from oracle" not in src and "import oracle" not in src, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):  # measurement scripts and compiled host programs of the product: no oracle either
        for f in files:
            if not f.endswith((".py", ".sh", ".cpp", ".hpp")):
                continue
            src = open(os.path.join(dirpath, f), errors="replace").read()
            assert "from oracle" not in src and "import oracle" not in src and "tests.helpers" not in src and "libmsm_oracle" not in src and "orc_" not in src, f


def test_argument_validation(built):
    with pytest.raises(M.MsmError):
        M.icosphere_counts(-1)
    with pytest.raises(M.MsmError):
        M.icosphere_counts(99)
