"""tools/stream_busy.py TRACE.csv [WINDOW_MS] -- per stream of a rocprofv3 --kernel-trace run: the time inside kernels and the idle gaps of the last WINDOW_MS (default 115:
one gMSM set-up at S = 64), the largest gap classes by the kernels either side."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 115.0


def short(n):
    return n.replace("msm::(anonymous namespace)::", "").replace("msm::", "").replace("void ", "").split("(")[0]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
t0 = ev[-1][1] - win * 1e6
for sid in sorted(set(e[3] for e in ev)):
    se = [e for e in ev if e[0] > t0 and e[3] == sid]
    if len(se) < 2:
        continue
    busy = sum(e[1] - e[0] for e in se)
    gaps = [((se[i + 1][0] - se[i][1]) / 1e3, se[i][2], se[i + 1][2]) for i in range(len(se) - 1)]
    big = [g for g in gaps if g[0] > 20]
    print("stream %s: %d kernels, %.1f ms inside kernels of %.1f; %d gaps > 20 us = %.1f ms" % (sid, len(se), busy / 1e6, (se[-1][1] - se[0][0]) / 1e6, len(big), sum(g[0] for g in big) / 1e3))
    c = collections.Counter()
    for g in big:
        c[(g[1], g[2])] += g[0]
    for k, v in c.most_common(6):
        print("      %5.1f ms  after %s before %s" % (v / 1e3, k[0], k[1]))
