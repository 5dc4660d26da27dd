"""tools/group_patch_digest.py [S] [data_order cp_order [label_order_offset]] -- sha256 over the exported set-up products (resampled maps, patch row offsets, patch index lists) of S subjects (default: ico6 / ico4):
two builds or two settings of the library (MSMHIP_RANGE_GRID=off, MSMHIP_RANGE_CLUSTER=off, ...) must print the same digest."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
do, co = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 4)
ctx = M.Context(0)
off = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g, keep = problem.build_group(ctx, S, do, co, D=2, label_order_offset=off)
g.setupCostFunction()
h = hashlib.sha256()
n = 0
for s in range(S):
    F, pptr, pidx = g.export_subject(s)
    for a in (F, pptr, pidx):
        h.update(np.ascontiguousarray(a).tobytes())
    n += len(pidx)
print("digest %s  (%d subjects, %d labels, %d patch entries)" % (h.hexdigest()[:32], S, g.L, n))
