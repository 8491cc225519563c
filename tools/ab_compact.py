#!/usr/bin/env python3
"""A/B: asynchronous ILU apply on the factor in place vs on natural-order compact copies of its
triangles (tuning "compact=1").  usage: ab_compact.py [n] [bs]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


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
    zs = {}
    for rep in range(3):
        for spec in ("compact=0", "compact=1"):
            capi.set_tuning(spec)
            p.set_timing(True)
            for _ in range(2):
                p.ilu0_apply(r, 3, out=z)
            p.get_timing(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                p.ilu0_apply(r, 3, out=z)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            t = p.get_timing(reset=True)
            p.set_timing(False)
            zs[spec] = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC).clone()
            print("%s: apply s=3 %.3f ms; lower %.3f ms, upper %.3f ms per sweep" % (
                spec, dt, t["lower_ms"] / t["lower_launches"], t["upper_ms"] / t["upper_launches"]), flush=True)
    d = float((zs["compact=0"] - zs["compact=1"]).abs().max() / zs["compact=0"].abs().max())
    print("synchronous 3-sweep results differ by %.2e (relative)" % d)
    p.close()


if __name__ == "__main__":
    main()
