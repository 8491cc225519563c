#!/usr/bin/env python3
"""The agent-scope experiment of SURVEY section 7 on the asynchronous bs=4 sweeps (GPU box only): for each
kernel variant, time per sweep, distance to the exact triangular solves after s+s sweeps, contraction per
sweep, sweeps / milliseconds to reach 1e-2 and 1e-6, and BiCGStab iterations with 3 and 5 sweeps.
usage: scope_ab.py [n=256] [maxit=400] [variant[+tuning...]] ...
A variant is a BLASTED_HIP_SWEEPW string ("default", "r128,nt1,u1,s1,c1" = agent-scope iterate accesses);
"+interleave=1" etc. appends other tuning strings."""
import math
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
from tools.solve_compare import bicgstab  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    variants = sys.argv[3:] or ["default", "r128,nt1,u1,s1,c1", "default+interleave=1", "r128,nt1,u1,s1,c1+interleave=1"]
    bs = 4
    dev = torch.device("cuda", 0)
    ms = W.poisson3d_device(n, 1, dev, grid="uniform")
    r_, c_ = torch.meshgrid(torch.arange(bs, device=dev), torch.arange(bs, device=dev), indexing="ij")
    Mb = torch.eye(bs, dtype=torch.float64, device=dev) * (1.0 + 0.1 * r_) + 0.03 * (((r_ + 2 * c_) % 3) - 1) * (r_ != c_)
    m = dict(ms)
    m.update(bs=bs, vals=(ms["vals"][:, None] * Mb.t().reshape(-1)[None, :]).reshape(-1), rowmajor=False)
    del ms
    b = W.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(b)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(3)
    ze = p.ilu0_apply(b, 1, mode=capi.LEVEL).clone()
    nz = float(ze.norm())
    A = lambda v: p.spmv(v)
    print("(7-point Poisson %d^3) x (fixed 4x4 block), %d block-rows; factor: 3 asynchronous build sweeps" % (n, m["nbrows"]))
    for var in variants:
        parts = var.split("+")
        capi.set_tuning("interleave=0")
        capi.set_tuning(None if parts[0] == "default" else parts[0])
        for extra in parts[1:]:
            capi.set_tuning(extra)
        # time per sweep (HIP events of the library around the sweep phases)
        for _ in range(2):
            p.ilu0_apply(b, 3, out=z)
        p.set_timing(True)
        p.get_timing(reset=True)
        for _ in range(6):
            p.ilu0_apply(b, 3, out=z)
        t = p.get_timing(reset=True)
        p.set_timing(False)
        lo, up = t["lower_ms"] / t["lower_launches"], t["upper_ms"] / t["upper_launches"]
        d = {}
        for s in (1, 3, 5, 10):
            p.ilu0_apply(b, s, out=z)
            d[s] = float((z - ze).norm()) / nz
        rho = (d[10] / d[3]) ** (1.0 / 7)
        line = "%-34s L %.3f U %.3f ms/sweep | dist 1+1 %.3f 3+3 %.3f 5+5 %.3f 10+10 %.2e | contraction %.3f" % (
            var, lo, up, d[1], d[3], d[5], d[10], rho)
        for tol in (1e-2, 1e-6):
            k = max(3, 3 + math.ceil(math.log(tol / d[3]) / math.log(rho)))
            line += " | to %.0e: %d sweeps %.1f ms" % (tol, k, k * (lo + up))
        print(line, flush=True)
        for s in (3, 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, its, res = bicgstab(A, lambda v: p.ilu0_apply(v, s), b, maxit=maxit)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            state = "converged" if res < 1e-8 else ("DIVERGED" if (res != res or res > 1e6) else "not converged")
            print("    BiCGStab, ilu0 %d+%d asynchronous sweeps: %s after %d iterations, residual %.1e, %.2f s" % (
                s, s, state, its, res, dt), flush=True)
    capi.set_tuning("interleave=0")
    capi.set_tuning(None)
    # the deterministic alternatives on the same factor, for reference
    for name, M in (("3+3 synchronous sweeps", lambda v: p.ilu0_apply(v, 3, mode=capi.JACOBI_SYNC)),
                    ("exact solves (LEVEL)", lambda v: p.ilu0_apply(v, 1, mode=capi.LEVEL))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, its, res = bicgstab(A, M, b, maxit=maxit)
        torch.cuda.synchronize()
        print("    BiCGStab, %s: %d iterations, residual %.1e, %.2f s" % (name, its, res, time.perf_counter() - t0), flush=True)
    p.close()


if __name__ == "__main__":
    main()
