"""tools/setup_profile_summary.py TRACE.csv [SETUPS] -- a rocprofv3 --kernel-trace run of tools/time_group_setup_only.py: per kernel the calls, average and total time per
set-up, and for the forest's level kernels of the last set-up the median / mean / maximum per level (what the critical path of a subject is made of)."""
import collections
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
setups = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def short(n):
    return n.replace("msm::(anonymous namespace)::", "").replace("msm::", "").replace("void ", "").split("(")[0]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, _ in ev:
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
print("kernel time per set-up: %.1f ms" % (sum(v[1] for v in tot.values()) / setups / 1e3))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:32]:
    print("  %-36s %6d calls  %8.1f us  %6.1f ms per set-up" % (n[:36], c // setups, t / c, t / setups / 1e3))
t_end = ev[-1][1]
streams = collections.Counter(e[3] for e in ev if "k_oct_decide" in e[2])
for sid in streams:
    se = [e for e in ev if e[0] > t_end - 120e6 and e[3] == sid]
    lev = collections.defaultdict(lambda: collections.defaultdict(list))
    level = -1
    for s, e, n, _ in se:
        if n.startswith("k_oct_init"):
            level = -1
        if n.startswith("k_oct_decide"):
            level += 1
        if n.startswith("k_oct_") and level >= 0:
            lev[level][n.split("<")[0][6:]].append((e - s) / 1e3)
    for level in sorted(lev):
        print("stream %s level %d: " % (sid, level) + " | ".join("%s %.0f / %.0f / %.0f" % (k, np.median(v), np.mean(v), max(v)) for k, v in lev[level].items()))
