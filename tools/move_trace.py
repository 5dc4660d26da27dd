import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
reasons, a = a[-1, :4], a[:-1]   # the last row: why samples were left open (move_kernels.hip: ray_open_reason)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
names = ["start", "phase0 done", "barrier1", "sampled", "barrier2", "fixup", "finish"]
print("blocks", len(a), "kernel span %.2f us" % ((a[:, :7].max() - t0) / 100.0))
for k in range(7):
    ok = a[:, k] > 0
    rel = (a[ok, k] - t0) / 100.0
    line = "%-12s p50 %.2f  p90 %.2f  max %.2f us" % (names[k], np.median(rel), np.percentile(rel, 90), rel.max())
    if k:
        ok2 = ok & (a[:, k - 1] > 0)
        d = (a[ok2, k] - a[ok2, k - 1]) / 100.0
        line += "   delta p50 %.2f p90 %.2f max %.2f" % (np.median(d), np.percentile(d, 90), d.max())
    print(line)
print("open samples (left to the leaf search): %d in %d of %d workgroups" % (a[:, 7].sum(), (a[:, 7] > 0).sum(), len(a)))
print("  of them: no cell %d, no candidate near its threshold %d, near but outside in FP64 %d, leaf may not list the triangle %d" % tuple(reasons))
st = np.sort((a[:, 0] - t0) / 100.0)
print("block start times: ", " ".join("%.1f" % st[int(q * (len(st) - 1))] for q in (0, .25, .5, .6, .7, .75, .8, .9, 1.0)))
