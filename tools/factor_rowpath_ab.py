#!/usr/bin/env python3
"""bs=4 / bs=8 in-place factorisation sweep: block-by-block loop (factorN=2, rounds 1-2) against the up-front row path
(factorN=1, round 3) on the 7-point Poisson pattern.  Per-sweep time from the difference of a 13- and a 3-sweep build;
quality = distance of the 3-sweep factor to the exact one.  usage: factor_rowpath_ab.py [bs=8] [n ...] (default 100 128)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
import bench  # noqa: E402


def timed(f, reps):
    f()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def main():
    bs = ([int(a[3:]) for a in sys.argv[1:] if a.startswith("bs=")] or [8])[-1]
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("bs=")] or [100, 128]
    dev = torch.device("cuda", 0)
    for n in sizes:
        ab = bench.algorithmic_bytes(n, bs)
        m = W.poisson3d_device(n, bs, dev, grid="uniform")
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        p.ilu0_factorize(-1)
        exact = torch.from_numpy(p.get_iluvals())
        for spec in ("factor%d=2" % bs, "factor%d=1" % bs):
            capi.set_tuning(spec)
            t3 = timed(lambda: p.ilu0_factorize(3), 5)
            t13 = timed(lambda: p.ilu0_factorize(13), 5)
            per = (t13 - t3) / 10
            p.ilu0_factorize(3)
            f3 = torch.from_numpy(p.get_iluvals())
            # (the stored factor has inverted diagonal blocks in both)
            dist = float((f3 - exact).norm() / exact.norm())
            print("n=%d bs=%d %-10s sweep (with its pre-pass, if any) %.3f ms: %.2f TB/s every-array-once = %.3f of peak, %.2f TB/s touched = %.3f; "
                  "3-sweep build %.2f ms, distance of the 3-sweep factor to the exact one %.3e" % (
                      n, bs, spec, per * 1e3, ab["factor_sweep"] / per / 1e12, ab["factor_sweep"] / per / 8e12,
                      ab["factor_sweep_touched"] / per / 1e12, ab["factor_sweep_touched"] / per / 8e12, t3 * 1e3, dist), flush=True)
        capi.set_tuning("factor%d=1" % bs)
        p.close()
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
