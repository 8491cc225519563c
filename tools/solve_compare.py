#!/usr/bin/env python3
"""End-to-end effect of the preconditioner variants: right-preconditioned BiCGStab and the reference's flexible
GCR (tests/solvers.cpp:247-352; device vectors, torch for the vector algebra, this library for SpMV and the
preconditioner) on the block-inflated 3-D Poisson matrix (Poisson (x) one fixed block) -- iterations and time to
a relative residual of 1e-8.
usage: solve_compare.py [n=160] [bs=4] [solver=bcgs|gcr|both] [restart=30] [gen=poisson|unstructured] [only=substring ...] [tuning strings ...]
(with tuning strings, e.g. interleave=1, only the asynchronous variants are run)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def bicgstab(A, M, b, tol=1e-8, maxit=1200):
    x = torch.zeros_like(b)
    r = b.clone()
    rhat = r.clone()
    rho = alpha = omega = 1.0
    v = torch.zeros_like(b)
    p = torch.zeros_like(b)
    bn = float(b.norm())
    for it in range(1, maxit + 1):
        rho1 = float(torch.dot(rhat, r))
        if rho1 == 0.0:
            return x, it, float(r.norm()) / bn
        beta = (rho1 / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        ph = M(p)
        v = A(ph)
        alpha = rho1 / float(torch.dot(rhat, v))
        s = r - alpha * v
        if float(s.norm()) / bn < tol:
            return x + alpha * ph, it, float(s.norm()) / bn
        sh = M(s)
        t = A(sh)
        omega = float(torch.dot(t, s)) / float(torch.dot(t, t))
        x = x + alpha * ph + omega * sh
        r = s - omega * t
        rho = rho1
        res = float(r.norm()) / bn
        if res < tol:
            return x, it, res
        if not (res == res) or res > 1e6:
            return x, it, res  # diverged
    return x, maxit, float(r.norm()) / bn


def gcr(A, M, b, tol=1e-8, maxit=1200, restart=30):
    """Restarted right-preconditioned GCR, the reference's flexible solver (tests/solvers.cpp:247-352): the
    direction p_k = M(r_k) is kept beside q_k = A p_k, so M may be a different operator at every application."""
    x = torch.zeros_like(b)
    bn = float(b.norm())
    P = torch.empty((restart, b.numel()), dtype=b.dtype, device=b.device)
    Q = torch.empty_like(P)
    qq = torch.empty(restart, dtype=b.dtype, device=b.device)
    step = 0
    rel = 1.0
    while step < maxit:
        res = b - A(x)
        P[0] = M(res)
        Q[0] = A(P[0])
        qq[0] = torch.dot(Q[0], Q[0])
        for k in range(restart):
            alpha = torch.dot(res, Q[k]) / qq[k]
            x += alpha * P[k]
            res -= alpha * Q[k]
            rel = float(res.norm()) / bn
            step += 1
            if rel < tol or k == restart - 1 or step >= maxit or not (rel == rel) or rel > 1e6:
                break
            z = M(res)
            q = A(z)
            beta = -(Q[:k + 1] @ q) / qq[:k + 1]
            P[k + 1] = z + beta @ P[:k + 1]
            Q[k + 1] = q + beta @ Q[:k + 1]
            qq[k + 1] = torch.dot(Q[k + 1], Q[k + 1])
        if rel < tol or not (rel == rel) or rel > 1e6:
            break
    return x, step, rel


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    only = [a[5:] for a in sys.argv[3:] if a.startswith("only=")]
    solver = ([a[7:] for a in sys.argv[3:] if a.startswith("solver=")] or ["bcgs"])[-1]
    restart = int(([a[8:] for a in sys.argv[3:] if a.startswith("restart=")] or ["30"])[-1])
    gen = ([a[4:] for a in sys.argv[3:] if a.startswith("gen=")] or ["poisson"])[-1]
    specs = [a for a in sys.argv[3:] if not a.startswith(("only=", "solver=", "restart=", "gen="))]
    dev = torch.device("cuda", 0)
    # Kronecker product (scalar 7-point Poisson) x (one fixed, slightly non-symmetric bs x bs block with
    # positive spectrum): a well-posed system.  (The slot-dependent inflation of workloads.poisson3d, made to
    # exercise every block entry, gives a strongly indefinite operator -- fine for the fixed-point parity
    # tests, useless for a Krylov comparison.)
    ms = W.poisson3d_device(n, 1, dev, grid="uniform")
    if gen == "unstructured":
        # the pattern of bench config 4's generator (n^3 rows, ~14 neighbours a row, window-shuffled numbering) with
        # the values of a shifted graph Laplacian: -1 off the diagonal, degree x 1.02 on it
        ms = dict(W.unstructured_bsr(n, 1, device=dev))
        rp = ms["browptr"].long()
        v = -torch.ones(int(ms["nnzb"]), dtype=torch.float64, device=dev)
        v[ms["diagind"].long()] = 1.02 * (rp[1:] - rp[:-1] - 1).double()
        ms["vals"] = v
    r_, c_ = torch.meshgrid(torch.arange(bs, device=dev), torch.arange(bs, device=dev), indexing="ij")
    Mb = torch.eye(bs, dtype=torch.float64, device=dev) * (1.0 + 0.1 * r_) + 0.03 * (((r_ + 2 * c_) % 3) - 1) * (r_ != c_)
    vals = (ms["vals"][:, None] * Mb.t().reshape(-1)[None, :]).reshape(-1)   # column-major blocks
    m = dict(ms)
    m.update(bs=bs, vals=vals, rowmajor=False)
    b = W.rhs_vector_device(m["nbrows"] * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    A = lambda v: p.spmv(v)
    print("%s %d^3, bs=%d, %d block-rows; solver %s to 1e-8" % ("3-D Poisson" if gen == "poisson" else "unstructured Laplacian", n, bs, m["nbrows"], solver))
    variants = [
        ("none", None, lambda v: v),
        ("jacobi", lambda: p.jacobi_compute(), lambda v: p.jacobi_apply(v)),
        ("sgs async 3 sweeps", lambda: p.jacobi_compute(), lambda v: p.sgs_apply(v, 3)),
        ("sgs DETERMINISTIC 3 sweeps (host default)", lambda: p.jacobi_compute(), lambda v: p.sgs_apply(v, 3, mode=capi.DETERMINISTIC)),
        ("sgs exact (level_sgs)", lambda: p.jacobi_compute(), lambda v: p.sgs_apply(v, 1, mode=capi.LEVEL)),
        ("ilu0 async 3 build + 1 apply sweeps", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 1)),
        ("ilu0 async 3 build + 3 apply sweeps", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 3)),
        ("ilu0 async 3 build + 5 apply sweeps", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 5)),
        ("ilu0 async 3 build + 10 apply sweeps", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 10)),
        ("ilu0 async 3 build + 3 DETERMINISTIC apply sweeps (host default)", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 3, mode=capi.DETERMINISTIC)),
        ("ilu0 async 3 build + 10 DETERMINISTIC apply sweeps", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 10, mode=capi.DETERMINISTIC)),
        ("sapilu0: async 3 build, exact apply", lambda: p.ilu0_factorize(3), lambda v: p.ilu0_apply(v, 1, mode=capi.LEVEL)),
        ("seqilu0: exact build, exact apply", lambda: p.ilu0_factorize(-1), lambda v: p.ilu0_apply(v, 1, mode=capi.LEVEL)),
    ]
    if only:
        variants = [v for v in variants if any(o in v[0] for o in only)]
    for spec in specs:
        capi.set_tuning(spec)
    if specs:
        print("tuning: " + " ".join(specs))
        variants = [v for v in variants if "async" in v[0] and "sapilu0" not in v[0]]
    solvers = {"bcgs": [("bcgs", bicgstab)], "gcr": [("gcr(%d)" % restart, lambda A_, M_, b_: gcr(A_, M_, b_, restart=restart))]}
    solvers["both"] = solvers["bcgs"] + solvers["gcr"]
    for name, setup, M in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if setup:
            setup()
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        for sname, solve in solvers[solver]:
            napp = [0]

            def Mc(v):
                napp[0] += 1
                return M(v)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            x, its, res = solve(A, Mc, b)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            true = float((b - A(x)).norm() / b.norm())
            print("%-62s %-8s setup %7.1f ms  solve %8.1f ms  iterations %4d  prec applications %4d  residual %.1e (true %.1e)" % (
                name, sname, setup_ms, (t2 - t1) * 1e3, its, napp[0], res, true), flush=True)
    p.close()


if __name__ == "__main__":
    main()
