#!/usr/bin/env python3
"""A/B of the generic sweep kernel's row-step unrolling (tuning "gunroll=1" = off) on the sizes that map
one block-row per wave.  usage: python tools/ab_generic.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    cases = [("poisson128 bs5", lambda: W.poisson3d_device(128, 5, dev, grid="uniform")),
             ("poisson128 bs7", lambda: W.poisson3d_device(128, 7, dev, grid="uniform")),
             ("unstructured126 bs5", lambda: W.unstructured_bsr(126, 5, device=dev)),
             ("poisson160 bs3", lambda: W.poisson3d_device(160, 3, dev, grid="uniform"))]
    for name, gen in cases:
        m = gen()
        bs = m["bs"]
        r = W.rhs_vector_device(m["nbrows"] * bs, dev)
        z = torch.zeros_like(r)
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        p.ilu0_factorize(2)
        p.jacobi_compute()
        nnzb, nb = m["nnzb"], m["nbrows"]
        pair_bytes = (nnzb * (8 * bs * bs + 4)) + 4 * nb * 4 + 6 * nb * 8 * bs
        for rep in range(2):
            for spec in ("sweepodd=0", "sweepodd=1"):
                capi.set_tuning(spec)
                t = timed(lambda: p.ilu0_apply(r, 3, out=z))
                ts = timed(lambda: p.sgs_apply(r, 3, out=z))
                print("%-22s %-10s ilu apply s=3 %7.3f ms (%5.0f GB/s)   sgs apply s=3 %7.3f ms" % (
                    name, spec, t, 3 * pair_bytes / t / 1e6, ts), flush=True)
        p.close()
        del m, r, z
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
