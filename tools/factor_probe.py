#!/usr/bin/env python3
"""Where the bs=5 factorisation sweep's time goes (config 4, unstructured): the sweep as it is, without the LDS tiles
of the block products (factorprobe=1) and without the operand loads of the pairs (factorprobe=2) -- the probes give
WRONG factors, only their time is of interest.  usage: factor_probe.py"""
import os
import sys
import time

os.environ.setdefault("BLASTED_HIP_PROBES", "1")  # the timing experiments exist in the probes build only (make -C blasted_amd/csrc probes)
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
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[4]
    m = W.unstructured_bsr(cfg["n"], cfg["bs"], device=dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    for spec in ("factorprobe=0", "factorprobe=1", "factorprobe=2", "factorprobe=3", "factorprobe=4", "factorprobe=5", "factorskip=0", "factorodd=0"):
        capi.set_tuning(spec)
        t3 = timed(lambda: p.ilu0_factorize(3), 3)
        t13 = timed(lambda: p.ilu0_factorize(13), 3)
        print("config 4 bs=5 %-14s sweep (with its inversion pre-pass) %.3f ms" % (spec, (t13 - t3) / 10 * 1e3), flush=True)
    capi.set_tuning("factorprobe=0")
    capi.set_tuning("factorskip=1")
    capi.set_tuning("factorodd=1")
    p.close()


if __name__ == "__main__":
    main()
