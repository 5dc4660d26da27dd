"""Inputs of tools/cpp/registration_bench for a profiler run: python tools/make_registration_inputs.py DIR PRESET D writes DIR/in.bag (the synthetic
ico6 subject of bench.py: seeds 7 / 9) and DIR/conf (the text of a shipped preset, newmsm_amd/config.py: PRESETS).  Then e.g.
    rocprofv3 --kernel-trace --stats -- tools/cpp/registration_bench DIR/in.bag DIR/out.bag DIR/conf 2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import newmsm_amd as M
from newmsm_amd import config, synthetic
from newmsm_amd.bag import write_bag

d, preset, D = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.makedirs(d, exist_ok=True)
xyz, tri = M.make_mesh_from_icosa(6)
ref = synthetic.features(xyz, D, 7)
src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), D, 7)
write_bag(os.path.join(d, "in.bag"), orders=np.array([6, D], dtype=np.int32), in_data=src, ref_data=ref)
with open(os.path.join(d, "conf"), "w") as f:
    f.write(config.PRESETS[preset])
print("wrote", d)
