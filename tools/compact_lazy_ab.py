#!/usr/bin/env python3
"""A build followed by K applications (3+3 asynchronous sweeps each), with the compact triangle copies made at once
(compactafter=0, rounds 1-2), never (compact=0) and by the product's rule (compactafter=-1: once they pay).
usage: compact_lazy_ab.py [bs:n | config4 ...]  (default 4:256 config4 1:256)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
import bench  # noqa: E402


def main():
    cases = sys.argv[1:] or ["4:256", "config4", "1:256"]
    dev = torch.device("cuda", 0)
    for c in cases:
        if c == "config4":
            cfg = bench.CONFIGS[4]
            m = W.unstructured_bsr(cfg["n"], cfg["bs"], device=dev)
            bs = cfg["bs"]
        else:
            bs, n = (int(x) for x in c.split(":"))
            m = W.poisson3d_device(n, bs, dev, grid="uniform")
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        r = W.rhs_vector_device(m["nbrows"] * bs, dev)
        z = torch.empty_like(r)
        for specs, name in ((("compact=1", "compactafter=0"), "copies at once"), (("compact=0",), "never"),
                            (("compact=1", "compactafter=-1"), "product rule")):
            for s in specs:
                capi.set_tuning(s)
            line = "%-8s bs=%d %-15s build + K applications [ms]:" % (c, bs, name)
            for K in (1, 4, 8, 16, 24, 48):
                best = 1e30
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    p.ilu0_factorize(3)
                    for _ in range(K):
                        p.ilu0_apply(r, 3, out=z)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                line += "  K=%d %.2f" % (K, best * 1e3)
            print(line, flush=True)
        capi.set_tuning("compact=1")
        capi.set_tuning("compactafter=-1")
        p.close()
        del m, r, z
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
