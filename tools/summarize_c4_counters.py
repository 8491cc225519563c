#!/usr/bin/env python3
"""Per-kernel averages of the counter passes of tools/r03_c4_counters.sh (sweepodd kernels only)."""
import collections
import csv
import glob
import os
import sys


def main(root):
    out = collections.defaultdict(dict)
    for v in ("default", "nt1", "probe"):
        for pas in ("fetch", "write", "rdreq", "tcc"):
            for f in glob.glob(os.path.join(root, "%s_%s" % (v, pas), "*", "*_counter_collection.csv")):
                agg = collections.defaultdict(lambda: collections.defaultdict(list))
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "sweepodd_kernel<5, 1," in k or "sweepodd_kernel<5, 0," in k:
                        short = "upper" if "<5, 1," in k else "lower"
                        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for short, cs in agg.items():
                    for c, vals in cs.items():
                        out[(v, short)][c] = sum(vals) / len(vals)
                        out[(v, short)]["launches_" + c] = len(vals)
    for (v, short), cs in sorted(out.items()):
        line = "%-8s %-6s" % (v, short)
        f, w = cs.get("FETCH_SIZE"), cs.get("WRITE_SIZE")
        if f is not None and w is not None:
            line += " FETCH_SIZE %.1f MB  WRITE_SIZE %.1f MB  traffic (2*F + W) %.1f MB" % (f / 1024, w / 1024, (2 * f + w) / 1024)
        for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
            if c in cs:
                line += "  %s %.3e" % (c.replace("_sum", ""), cs[c])
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            line += "  L2 hit rate %.3f" % (cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1))
        print(line)


if __name__ == "__main__":
    main(sys.argv[1])
