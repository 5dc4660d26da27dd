#!/usr/bin/env python3
"""tools/summarise_pmc.py TAG [KERNEL_PREFIX [COMMAND]] -- averages the rocprofv3 counter_collection.csv files of tools/collect_profile.sh per
kernel and writes gpurun_out/TAG_pmc.json (per-launch means, HBM traffic per launch, kernel time from the stats pass)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "msm::k_unary"
command = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
per = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_%s_*" % tag)):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            per[k][c] = sum(v) / len(v)
import os

avg_ns, calls = {}, {}
for r in csv.DictReader(open("gpurun_out/%s_kernel_stats.csv" % tag)):
    avg_ns[r["Name"].split("(")[0]] = float(r["AverageNs"])
    calls[r["Name"].split("(")[0]] = int(r["Calls"])
steps = int(os.environ.get("MSM_PROFILE_STEPS", "0"))  # label steps the profiled command ran (tools/collect_group_profile.sh): launches per step
import hashlib


def source_stamp():
    """sha256 over the kernel sources the profiled library was built from (newmsm_amd/csrc/*.hip, *.hpp): bench.py reports counters of a profile only
    while the sources still hash to this (a committed profile must not be priced against a different kernel binary)"""
    h = hashlib.sha256()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "newmsm_amd", "csrc")
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()


out = {
    "command": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- " + command,
    "kernel_source_sha256": source_stamp(),
    "traffic_formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are reported in KiB; gfx950 FETCH_SIZE counts "
                       "half of the bytes of wide reads (MI355X_MICROARCH.md, HBM section)",
    "kernels": {},
}
for k, v in per.items():
    if prefix not in k:
        continue
    e = {"per_launch_mean": v, "kernel_avg_ns_from_kernel_stats": avg_ns.get(k), "calls_in_stats_pass": calls.get(k)}
    if steps and calls.get(k):
        e["launches_per_step"] = calls[k] / steps
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["hbm_traffic_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    out["kernels"][k] = e
json.dump(out, open("gpurun_out/%s_pmc.json" % tag, "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, "avg_ns", e["kernel_avg_ns_from_kernel_stats"], "hbm_bytes", e.get("hbm_traffic_bytes_per_launch"))
