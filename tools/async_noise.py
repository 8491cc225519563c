#!/usr/bin/env python3
"""How much the asynchronous application differs from itself: two applications of s+s in-place sweeps to the same
right-hand side, relative 2-norm of their difference, next to their distance from the exact solves -- per tuning.
usage: async_noise.py [n=160] [bs=4 | -4 for ROW-major blocks] [tuning ...]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rowmajor = bs < 0
    bs = abs(bs)
    dev = torch.device("cuda", 0)
    ms = W.poisson3d_device(n, 1, dev, grid="uniform")
    r_, c_ = torch.meshgrid(torch.arange(bs, device=dev), torch.arange(bs, device=dev), indexing="ij")
    Mb = torch.eye(bs, dtype=torch.float64, device=dev) * (1.0 + 0.1 * r_) + 0.03 * (((r_ + 2 * c_) % 3) - 1) * (r_ != c_)
    m = dict(ms)
    m.update(bs=bs, vals=(ms["vals"][:, None] * (Mb if rowmajor else Mb.t()).reshape(-1)[None, :]).reshape(-1), rowmajor=rowmajor)
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
    for spec in (sys.argv[3:] or ["interleave=0", "interleave=1", "interleave=2"]):
        capi.set_tuning(spec)
        for s in (1, 3, 5, 10):
            zs = [p.ilu0_apply(r, s).clone() for _ in range(4)]
            noise = max(float((zs[i] - zs[0]).norm() / zs[0].norm()) for i in range(1, 4))
            dist = float((zs[0] - ze).norm() / ze.norm())
            print("%-14s %2d+%2d sweeps: distance to exact %.2e, difference between two applications %.2e (%.1f %% of the distance)" % (
                spec, s, s, dist, noise, 100 * noise / dist), flush=True)
    capi.set_tuning("interleave=0")
    p.close()


if __name__ == "__main__":
    main()
