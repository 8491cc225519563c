#!/usr/bin/env python3
"""Groups a rocprofv3 --pmc counter_collection.csv of tools/probes/placement_variance.py by build round:
mean duration and counter value per launch of the lower and upper sweep kernels.
usage: placement_pmc.py <counter_collection.csv> <launches per round and kernel>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
by = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "sweepw_kernel<4, 0, 0, 0" in k:
        name = "L"
    elif "sweepw_kernel<4, 1, 1, 1" in k:
        name = "U"
    else:
        continue
    d = by.setdefault((name, int(r["Dispatch_Id"])), {"t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
for name in "LU":
    seq = [v for (n, _), v in by.items() if n == name]
    print(name, "launches", len(seq))
    for rnd in range(len(seq) // per):
        grp = seq[rnd * per:(rnd + 1) * per][per // 2:]  # second half: past the warm-up applies
        keys = [k for k in grp[0] if k != "t"]
        line = "  round %2d: %.3f ms" % (rnd, sum(g["t"] for g in grp) / len(grp))
        for k in keys:
            line += "  %s %.4g" % (k, sum(g[k] for g in grp) / len(grp))
        print(line)
