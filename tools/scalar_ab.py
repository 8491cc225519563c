#!/usr/bin/env python3
"""Scalar (CSR) kernels A/B on the 7-point Poisson pattern: the row sweeps with four lanes per row (scalarlane=0) and
one lane per row (1, 2 = rows per lane in flight), and the in-place factorisation sweep with / without the
precomputed plan (factor1plan).  Per-sweep times come from differences of runs with different sweep counts, so
initialisation passes and launch overheads of the call cancel.  usage: scalar_ab.py [n ...]  (default 128 200 256)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
import bench  # noqa: E402

PEAK = 8.0e12


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
    sizes = [int(a) for a in sys.argv[1:]] or [128, 200, 256]
    dev = torch.device("cuda", 0)
    for n in sizes:
        ab = bench.algorithmic_bytes(n, 1)
        m = W.poisson3d_device(n, 1, dev, grid="uniform")
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        r = W.rhs_vector_device(n ** 3, dev)
        z = torch.empty_like(r)
        for plan in (0, 1):
            capi.set_tuning("factor1plan=%d" % plan)
            t3 = timed(lambda: p.ilu0_factorize(3), 5)
            t13 = timed(lambda: p.ilu0_factorize(13), 5)
            per = (t13 - t3) / 10
            print("n=%d factor1plan=%d  in-place factorisation sweep %.4f ms  %.2f TB/s algorithmic = %.3f of peak (3-sweep build %.3f ms)" % (
                n, plan, per * 1e3, ab["factor_sweep"] / per / 1e12, ab["factor_sweep"] / per / PEAK, t3 * 1e3), flush=True)
        capi.set_tuning("factor1plan=1")
        p.ilu0_factorize(30)
        exact = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
        p.jacobi_compute()
        for lanes, unr in ((0, 0), ("auto", 0), (3, 0)):
            capi.set_tuning("scalarlane=%s" % lanes)
            capi.set_tuning("gunroll=%d" % unr)
            if unr:
                print("(general kernel with 2 / 4 row steps in flight: gunroll=2)")
            for mode, name in ((capi.ASYNC, "async"), (capi.JACOBI_SYNC, "sync")):
                t2 = timed(lambda: p.ilu0_apply(r, 2, mode=mode, out=z), 10)
                t12 = timed(lambda: p.ilu0_apply(r, 12, mode=mode, out=z), 10)
                per = (t12 - t2) / 10
                z3 = p.ilu0_apply(r, 3, mode=mode, out=z)
                dist = float((z3 - exact).norm() / exact.norm())
                print("n=%d scalarlane=%s %-5s ILU apply, one L+U sweep pair %.4f ms  %.2f TB/s = %.3f of peak; 3+3 sweeps: distance to the exact solve %.4f" % (
                    n, lanes, name, per * 1e3, ab["ilu_pair"] / per / 1e12, ab["ilu_pair"] / per / PEAK, dist), flush=True)
            t1 = timed(lambda: p.sgs_apply(r, 1, mode=capi.JACOBI_SYNC, out=z), 10)
            t11 = timed(lambda: p.sgs_apply(r, 11, mode=capi.JACOBI_SYNC, out=z), 10)
            per = (t11 - t1) / 10
            x0 = torch.zeros_like(r)
            tr1 = timed(lambda: p.sgs_relax(r, x0, 1, mode=capi.JACOBI_SYNC), 10)
            tr6 = timed(lambda: p.sgs_relax(r, x0, 6, mode=capi.JACOBI_SYNC), 10)
            print("n=%d scalarlane=%s SGS relaxation (sync), one forward+backward step %.4f ms = %.3f of peak" % (
                n, lanes, (tr6 - tr1) / 5 * 1e3, 2 * ab["sgs_relax_pass"] / ((tr6 - tr1) / 5) / PEAK), flush=True)
            ts = timed(lambda: p.spmv(r, out=z), 20)
            print("n=%d scalarlane=%s SGS sync sweep pair %.4f ms = %.3f of peak; SpMV %.4f ms = %.3f of peak" % (
                n, lanes, per * 1e3, ab["sgs_pair"] / per / PEAK, ts * 1e3, ab["spmv"] / ts / PEAK), flush=True)
        capi.set_tuning("scalarlane=auto")
        capi.set_tuning("gunroll=0")
        p.close()
        del m, r, z
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
