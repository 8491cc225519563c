#!/usr/bin/env python3
"""Three-sweep asynchronous ILU(0) build from INIT_F_ORIGINAL with a separate initialisation pass (factorfuse=0) and
with the pass fused into the first sweep (factorfuse=1): wall time per build and the distance of the
resulting factor to the exact one.  usage: factor_fuse_ab.py [bs:n | config4 ...]  (default 4:256 4:128 config4 8:100 5:128; also unstructured:N, random:BS:NROWS)"""
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
    cases = sys.argv[1:] or ["4:256", "4:128", "config4", "8:100", "5:128"]
    dev = torch.device("cuda", 0)
    for c in cases:
        if c == "config4":
            cfg = bench.CONFIGS[4]
            m = W.unstructured_bsr(cfg["n"], cfg["bs"], device=dev)
            bs = cfg["bs"]
        elif c.startswith("unstructured:"):
            bs = 5
            m = W.unstructured_bsr(int(c.split(":")[1]), bs, device=dev)
        elif c.startswith("random:"):
            bs = int(c.split(":")[1])
            mm = W.random_bsr(int(c.split(":")[2]), bs, avg_offdiag=8, seed=3)
            m = {k: (torch.from_numpy(v).to(dev) if hasattr(v, "dtype") else v) for k, v in mm.items()}
        else:
            bs, n = (int(x) for x in c.split(":"))
            m = W.poisson3d_device(n, bs, dev, grid="uniform")
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        p.ilu0_factorize(-1)
        exact = torch.from_numpy(p.get_iluvals()) if m["vals"].numel() < 4e8 else None
        for spec in ("factorfuse=0", "factorfuse=1"):
            capi.set_tuning(spec)
            for sweeps in ((1, 2, 3, 4, 5, 8) if exact is not None else (1, 3)):
                t = timed(lambda: p.ilu0_factorize(sweeps), 4)
                dist = float("nan")
                if exact is not None:
                    p.ilu0_factorize(sweeps)
                    f = torch.from_numpy(p.get_iluvals())
                    dist = float((f - exact).norm() / exact.norm())
                print("%-8s bs=%d %-13s %d-sweep build %.3f ms, distance of the factor to the exact one %.3e" % (
                    c, bs, spec, sweeps, t * 1e3, dist), flush=True)
        capi.set_tuning("factorfuse=1")
        p.close()
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
