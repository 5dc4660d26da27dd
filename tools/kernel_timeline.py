"""tools/kernel_timeline.py TRACE.csv [gap_us] -- the kernels of a rocprofv3 --kernel-trace run in start order, split into bursts at idle gaps
of more than gap_us (default 150): for each burst of the last three its kernels with start offset, duration and the gap before each.  Shows
what a short host-driven phase (a set-up call, a label step) is made of: kernel time against launch gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
bursts, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > gap_us * 1e3:
        bursts.append(cur)
        cur = []
    cur.append(e)
if cur:
    bursts.append(cur)
print("%d kernels in %d bursts" % (len(ev), len(bursts)))
for b in bursts[-3:]:
    t0, busy = b[0][0], sum(e[1] - e[0] for e in b)
    print("--- burst of %d kernels, %.1f us from first start to last end, %.1f us inside kernels" % (len(b), (b[-1][1] - t0) / 1e3, busy / 1e3))
    prev = t0
    for s, e, name in b:
        short = name.replace("msm::(anonymous namespace)::", "").replace("msm::", "").replace("void ", "").split("(")[0]
        print("  +%8.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short[:70]))
        prev = e
