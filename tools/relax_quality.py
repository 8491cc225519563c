#!/usr/bin/env python3
"""What k asynchronous SGS relaxation steps (BASELINE config 3's operator) leave of the residual, and what they cost,
per tuning string.  usage: relax_quality.py [n=256] [bs=4] [tuning ...]"""
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
    b = W.rhs_vector_device(m["nbrows"] * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.jacobi_compute()
    bn = float(b.norm())
    for spec in (sys.argv[3:] or ["interleave=0", "interleave=3"]):
        capi.set_tuning(spec)
        for k in (1, 5, 20):
            x = torch.zeros_like(b)
            p.sgs_relax(b, x, k)
            torch.cuda.synchronize()
            res = float((b - p.spmv(x)).norm()) / bn
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p.sgs_relax(b, x, k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            print("%-14s %2d relaxation steps: %8.2f ms, relative residual %.3e" % (spec, k, dt, res), flush=True)
    capi.set_tuning("interleave=0")
    p.close()


if __name__ == "__main__":
    main()
