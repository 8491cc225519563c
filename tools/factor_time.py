#!/usr/bin/env python3
"""Time of the in-place factorisation sweep and of a three-sweep build: `odd` = config 4 and four smaller bs = 5 / 7
patterns (kernels_factorodd.hip), `bs4` = config 2 (256^3), 128^3 and an unstructured bs = 4 pattern
(kernels_factor4.hip).  usage: factor_time.py odd|bs4|rowmajor [tuning ...]  (several tuning strings = an A/B in one process)"""
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


VARIANTS = [""]


def run(name, m):
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    for rep in range(2):
        for v in VARIANTS:
            if v:
                capi.set_tuning(v)
            t3 = timed(lambda: p.ilu0_factorize(3), 3)
            t13 = timed(lambda: p.ilu0_factorize(13), 3)
            print("%-30s %-16s sweep %.3f ms   3-sweep build %.3f ms" % (name, v, (t13 - t3) / 10 * 1e3, t3 * 1e3), flush=True)
    p.close()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "odd"
    if len(sys.argv) > 2:
        VARIANTS[:] = sys.argv[2:]   # tuning strings measured side by side on every pattern, twice
    dev = torch.device("cuda", 0)
    if which == "rowmajor":
        # the same matrices with their blocks stored row-major (the reference instantiates bs = 4 RowMajor for its own
        # drivers; PETSc's blocks are column-major)
        for n, bs in ((128, 4), (100, 8), (100, 5)):
            m = W.poisson3d_device(n, bs, dev)
            run("poisson %d^3 bs=%d col-major" % (n, bs), m)
            mr = dict(m)
            mr["vals"] = m["vals"].view(-1, bs, bs).transpose(1, 2).contiguous().view(-1)
            mr["rowmajor"] = True
            run("poisson %d^3 bs=%d ROW-major" % (n, bs), mr)
        return
    if which == "bs4":
        run("poisson 128^3 bs=4", W.poisson3d_device(128, 4, dev))
        run("unstructured 100^3 bs=4", W.unstructured_bsr(100, 4, device=dev))
        run("config 2 (poisson 256^3 bs=4)", W.poisson3d_device(256, 4, dev))
        return
    cfg = bench.CONFIGS[4]
    run("config 4 (unstructured bs=5)", W.unstructured_bsr(cfg["n"], cfg["bs"], device=dev))
    run("unstructured 60^3 bs=5", W.unstructured_bsr(60, 5, device=dev))
    run("poisson 100^3 bs=5", W.poisson3d_device(100, 5, dev))
    run("poisson 80^3 bs=7", W.poisson3d_device(80, 7, dev))
    run("unstructured 60^3 bs=7", W.unstructured_bsr(60, 7, device=dev))


if __name__ == "__main__":
    main()
