#!/usr/bin/env python3
"""How close s asynchronous sweeps get to the exact triangular solves they iterate towards, and what they
cost, with the interleaved in-chunk row order on and off.  usage: async_quality.py [n=256] [bs=4]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    ms = W.poisson3d_device(n, 1, dev, grid="uniform")
    r_, c_ = torch.meshgrid(torch.arange(bs, device=dev), torch.arange(bs, device=dev), indexing="ij")
    Mb = torch.eye(bs, dtype=torch.float64, device=dev) * (1.0 + 0.1 * r_) + 0.03 * (((r_ + 2 * c_) % 3) - 1) * (r_ != c_)
    m = dict(ms)
    m.update(bs=bs, vals=(ms["vals"][:, None] * Mb.t().reshape(-1)[None, :]).reshape(-1), rowmajor=False)
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(r)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
    for spec in (sys.argv[3:] or ["interleave=0", "interleave=1"]):
        capi.set_tuning(spec)
        for s in (1, 3, 10, 30):
            p.ilu0_apply(r, s, out=z)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                p.ilu0_apply(r, s, out=z)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3 * 1e3
            err = float((z - ze).norm() / ze.norm())
            print("%s  %2d+%2d sweeps: %7.2f ms, relative distance to the exact solve %.2e" % (spec, s, s, dt, err), flush=True)
    p.close()


if __name__ == "__main__":
    main()
