#!/usr/bin/env python3
"""Dependency latency of the one-launch exact pass: a block-tridiagonal matrix has one row per level, so
(time of an exact lower+upper solve) / (2 * rows) is the store -> poll -> compute -> store round trip."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi  # noqa: E402


def tridiag(nb, bs):
    rows = np.repeat(np.arange(nb), 3)
    cols = rows + np.tile([-1, 0, 1], nb)
    keep = (cols >= 0) & (cols < nb)
    rows, cols = rows[keep], cols[keep]
    rp = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp)
    rng = np.random.default_rng(1)
    vals = rng.uniform(-0.1, 0.1, (rows.size, bs, bs))
    dg = np.nonzero(rows == cols)[0]
    vals[dg] += np.eye(bs)[None] * 2.0
    return {"nbrows": nb, "nnzb": int(rows.size), "bs": bs, "rowmajor": False, "browptr": rp.astype(np.int32),
            "bcolind": cols.astype(np.int32), "diagind": dg.astype(np.int32), "vals": vals.reshape(-1)}


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    for bs in (4, 5, 1):
        m = tridiag(nb, bs)
        p = capi.Prec(0)
        p.set_matrix(m)
        p.ilu0_factorize(-1)
        r = np.ones(nb * bs)
        for impl in ("syncfree", "launch"):
            capi.set_tuning("level=" + impl)
            p.ilu0_apply(r, 1, mode=capi.LEVEL)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                p.ilu0_apply(r, 1, mode=capi.LEVEL)
            dt = (time.perf_counter() - t0) / reps
            print("bs=%d %d levels %-9s exact apply %8.2f ms = %6.2f us per level and triangle  %s" % (
                bs, p.level_count(), impl, dt * 1e3, dt / (2 * nb) * 1e6, p.level_stats()), flush=True)
        p.close()


if __name__ == "__main__":
    main()
