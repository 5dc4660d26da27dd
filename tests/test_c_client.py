"""include/msmhip.h is consumed by a plain C99 program (tests/cpp/abi_demo.c) linked against libmsmhip.so:
the boundary works without Python and without HIP headers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "abi_demo.c")
EXE = os.path.join(ROOT, "tests", "cpp", "abi_demo")
LIBDIR = os.path.join(ROOT, "newmsm_amd")


def build_demo():
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", LIBDIR, "-lmsmhip", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_c_client_host_entry_points(built):
    build_demo()
    out = subprocess.run([EXE, "host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "ok" in out.stdout


@pytest.mark.gpu
def test_c_client_gpu(built):
    build_demo()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
