import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The direction table of a cost target is built in the background by default (api.cpp: ensure_rays) and the complete search
# serves the first tables, so a test that evaluates one table would never reach the table kernels: build it on first use
# here.  tests/test_gpu_unary.py::test_background_ray_table covers the default mode.
os.environ.setdefault("MSMHIP_RAYTABLE", "sync")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build libmsmhip.so and the oracle once per session (no-ops when already built)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import newmsm_amd as M

    if M.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the product has no CPU fallback)")
    c = M.Context(0)
    yield c
    c.close()
