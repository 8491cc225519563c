import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import test_gpu_parity as T
from blasted_amd import capi, workloads as W
import oracle as O
import os
golden = os.path.join("/root/repo/tests", "golden")
for name, m in (("poisson16", W.poisson3d(16, 4)), ("2dcyl1", T.matrices(golden)["2dcyl1_bs4_col"]())):
    p = T.make_prec(m)
    res = {}
    for k in ("0", "2"):
        capi.set_tuning("factorsf=" + k)
        p.ilu0_factorize(-1)
        res[k] = p.get_iluvals().reshape(-1, 16)
    print(name, p.level_stats())
    d = np.abs(res["0"] - res["2"]).max(axis=1) / np.abs(res["0"]).max()
    rp = np.asarray(m["browptr"]); di = np.asarray(m["diagind"]); col = np.asarray(m["bcolind"])
    pl = O.ilu_positions(m); pp = np.asarray(pl[0])
    lv, rows, ptr = p.get_levels()
    bad = np.nonzero(d > 1e-13)[0]
    print(name, "bad entries", len(bad), "of", len(d))
    rowof = np.repeat(np.arange(m["nbrows"]), rp[1:] - rp[:-1])
    for j in bad[:25]:
        i = rowof[j]
        print("  entry", j, "row", i, "level", lv[i], "q", j - rp[i], "nl", di[i] - rp[i], "ne", rp[i+1] - rp[i], "pairs", pp[j+1] - pp[j], "rowpairs", pp[rp[i+1]] - pp[rp[i]], "err %.2e" % d[j])
    p.close()
