#!/usr/bin/env python3
"""Distance to the exact factor after 1, 2, 3 asynchronous sweeps -- unfused / fused initialisation, clean / poisoned
factor storage -- for the same random pattern with column- and row-major blocks (is the spread of the row-major cases in
test_fused_initialisation_builds_the_same_factor the kernels' or the asynchronous iteration's?)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle as O  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for bs, nb, off, seed in ((4, 777, 5, 7), (8, 400, 5, 5), (5, 900, 7, 21)):
    for rm in (False, True):
        m = W.random_bsr(nb, bs, avg_offdiag=off, seed=seed, rowmajor=rm)
        exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
        p = capi.Prec(0)
        p.set_matrix(m)
        rng = np.random.default_rng(3)
        other = m["vals"].reshape(m["nnzb"], -1) * rng.uniform(0.5, 2.0, size=(m["nnzb"], 1))
        out = []
        for trial in range(3):
            row = {}
            for k in ("0", "1"):
                capi.set_tuning("factorfuse=" + k)
                for sweeps in (1, 2, 3):
                    p.ilu0_factorize(sweeps, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
                    row[k, sweeps] = rel(p.get_iluvals(), exact)
            capi.set_tuning("factorfuse=1")
            for sweeps in (1, 2, 3):
                p.set_values(np.ascontiguousarray(other.reshape(-1)))
                p.ilu0_factorize(5, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
                p.set_values(m["vals"])
                p.ilu0_factorize(sweeps, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
                row["p", sweeps] = rel(p.get_iluvals(), exact)
            out.append(row)
        p.close()
        print("bs %d %s-major" % (bs, "ROW" if rm else "col"))
        for k, name in (("0", "unfused"), ("1", "fused"), ("p", "fused, poisoned storage")):
            print("   %-26s" % name + "   ".join("%d sweeps: " % s + " / ".join("%.2e" % r[k, s] for r in out) for s in (1, 2, 3)))
