#!/usr/bin/env python3
"""Times the level-scheduled (exact) operators beside the asynchronous ones.
usage: python tools/level_perf.py [n=256] [bs=4]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(n, bs, dev, grid="uniform") if n > 0 else W.unstructured_bsr(-n, bs, device=dev)
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(r)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(3)
    p.jacobi_compute()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nl = p.level_count()
    torch.cuda.synchronize()
    print("nbrows %d bs %d: %d levels, schedule built in %.1f ms" % (m["nbrows"], bs, nl, (time.perf_counter() - t0) * 1e3))
    print("ilu factor exact      %8.3f ms" % timed(lambda: p.ilu0_factorize(-1), 2))
    print("ilu factor ASYNC s=3  %8.3f ms" % timed(lambda: p.ilu0_factorize(3), 2))
    print("ilu apply  LEVEL      %8.3f ms" % timed(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z), 5), p.level_stats())
    capi.set_tuning("sfonestep=0")
    print("   sfonestep=0         %8.3f ms" % timed(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z), 5))
    capi.set_tuning("sfonestep=1")
    if bs in (4, 8):
        for spec in ("levelwide=0", "levelstore=0", "level=launch"):
            capi.set_tuning(spec)
            print("   %-14s      %8.3f ms" % (spec, timed(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z), 5)))
        for spec in ("levelwide=1", "levelstore=1", "level=syncfree"):
            capi.set_tuning(spec)
    for s in (1, 3, 10):
        print("ilu apply  ASYNC s=%-2d %8.3f ms" % (s, timed(lambda: p.ilu0_apply(r, s, out=z), 5)))
    print("sgs apply  LEVEL      %8.3f ms" % timed(lambda: p.sgs_apply(r, 1, mode=capi.LEVEL, out=z), 5))
    print("sgs apply  ASYNC s=3  %8.3f ms" % timed(lambda: p.sgs_apply(r, 3, out=z), 5))
    x = torch.zeros_like(r)
    print("sgs relax  LEVEL 1 it %8.3f ms" % timed(lambda: p.sgs_relax(r, x, 1, mode=capi.LEVEL), 5))
    print("sgs relax  ASYNC 1 it %8.3f ms" % timed(lambda: p.sgs_relax(r, x, 1), 5))
    p.close()


if __name__ == "__main__":
    main()
