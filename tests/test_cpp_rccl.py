"""include/msmhip_rccl.hpp (the gMSM collectives for a C++ host, RCCL) driven by a compiled program: sharded set-up, a gathered label step
and the template all-reduce through a one-rank communicator ≡ the unsharded ABI calls.  CPU part: the header compiles and links."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "rccl_group.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "rccl_group")
LIBDIR = os.path.join(ROOT, "newmsm_amd")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", LIBDIR, "-lmsmhip",
                           "-lrccl", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc and the RCCL headers")
def test_rccl_header_compiles_and_links():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_rccl_collectives_match_the_unsharded_calls():
    build()
    out = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["mismatches"] == 0 and res["template_max_err"] < 1e-9 and res["n_subjects"] == 3 and res["pairs"] > 0 and res["triplets"] > 0
