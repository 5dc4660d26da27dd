"""msm-mi355x: MI355X-native (gfx950) implementation of newMSM's data-parallel hot path.

The product is libmsmhip.so (hand-written HIP behind the C ABI of include/msmhip.h); this package is a
thin ctypes mirror of the reference's interfaces used by tests and bench.py.  Importing the package
does not load the library; the first call does, and fails loudly if it has not been built.
"""
from ._lib import LIB_PATH, MsmError, lib  # noqa: F401
from .api import *  # noqa: F401,F403
