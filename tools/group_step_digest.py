"""tools/group_step_digest.py [S] [data_order cp_order] -- sha256 over the 4 P pair and 8 T triplet costs of a few label steps (first and second visits, labelings that change
between steps) of a synthetic group: two builds of the library (MSM_LIB_PATH) or two settings must print the same digest when a change claims bit-identical costs."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import newmsm_amd as M
from newmsm_amd import problem

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
do, co = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 4)
ctx = M.Context(0)
g, keep = problem.build_group(ctx, S, do, co, D=2)
g.setupCostFunction()
rng = np.random.default_rng(11)
lab = rng.integers(0, g.L, g.num_nodes).astype(np.int32)
h = hashlib.sha256()
for step, label in enumerate([3, 7, 3, 0, 7, 12]):
    quads, octets = g.fusionMove(lab, label)
    h.update(np.ascontiguousarray(quads).tobytes())
    h.update(np.ascontiguousarray(octets).tobytes())
    lab = np.where(rng.random(g.num_nodes) < 0.10, rng.integers(0, g.L, g.num_nodes), lab).astype(np.int32)
print("digest %s  (%d subjects, %d pairs, %d triplets, 6 label steps)" % (h.hexdigest()[:32], S, g.P, g.T))
