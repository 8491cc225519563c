#!/usr/bin/env python3
"""Randomised differential test: many small random BSR matrices (size, block size, layout, density, with or
without scaling) through every operator and mode of the C ABI against the CPU oracle.
usage: fuzz_parity.py [cases=150] [seed=1]"""
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    O.set_num_threads(8)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    worst = {}

    def check(name, got, want, tol):
        e = rel(got, want)
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, (name, e, case)

    for it in range(ncases):
        bs = int(rng.choice([1, 2, 3, 4, 5, 7, 8]))
        nb = int(rng.integers(1, 700)) if rng.integers(0, 12) else int(rng.integers(5000, 60000))
        rm = bool(rng.integers(0, 2)) and bs > 1
        dens = int(rng.integers(1, 12))
        sc = bool(rng.integers(0, 2))
        # a random kernel-variant selection, so that every implementation of a pass meets every shape
        tune = {k: int(rng.integers(0, 2)) for k in ("levelstore", "levelwide", "levelperm", "compact",
                                                     "interleave", "sweepodd", "sweepwr", "factorodd",
                                                     "factor4", "factor8", "gunroll")}
        tune["factor1"] = int(rng.integers(0, 2))
        tune["factor8"] = int(rng.choice([0, 1, 1, 2]))
        tune["factorodd"] = int(rng.choice([0, 1, 1, 2]))        # general / staged, batched operands / round-2 kernel            # generic / up-front row path / block-by-block loop
        tune["level"] = str(rng.choice(["syncfree", "syncfree", "launch"]))
        tune["copies"] = str(rng.choice(["one", "both"]))
        tune["xcdsuper"] = int(rng.choice([1, 4, 16, 64]))
        tune["levelserial"] = int(rng.choice([4096, 4096, 8]))   # 8: the in-order fall-back of the level build
        tune["levelfast"] = int(rng.choice([1, 1, 0]))           # the polling launch of the level build
        tune["factorsf"] = str(rng.choice(["0", "1", "1", "2", "3", "p0", "p1"]))  # forms of the exact factorisation
        tune["compactafter"] = int(rng.choice([0, 0, 1, 3]))      # compact triangle copies made with the (N+1)-th application
        tune["factorfuse"] = int(rng.integers(0, 2))             # initialisation pass fused into the first in-place sweep
        tune["factorskip"] = int(rng.integers(0, 2))             # fixed upper blocks left alone by in-place sweeps
        tune["interleave"] = int(rng.choice([0, 0, 1, 2]))       # row order of the in-place triangular sweeps
        tune["latestore"] = int(rng.choice([0, 1, 2, 2, 4]))     # a workgroup's results stored once, steps in flight
        tune["invertrow"] = int(rng.integers(0, 2))              # eight-lanes-per-block inversion (bs 5..8)
        tune["scalarlane"] = str(rng.choice(["auto", "0", "1", "2", "3", "4"]))       # scalar row sweeps: lanes per row / rows per lane
        tune["scalarstage"] = int(rng.integers(0, 2))            # whole-row scalar operators: products staged through LDS
        tune["factor1plan"] = int(rng.integers(0, 2))            # scalar in-place factorisation on the precomputed plan
        for k, v in tune.items():
            capi.set_tuning("%s=%s" % (k, v))
        if tune["sweepodd"]:
            tune["sweepodd_nt"] = str(rng.choice(["nt0", "nt1"]))
            tune["sweepodd_occ"] = str(rng.choice(["occ0", "occ1"]))
            capi.set_tuning("sweepodd=" + tune["sweepodd_nt"])
            capi.set_tuning("sweepodd=" + tune["sweepodd_occ"])
        tune["sweepw"] = str(rng.choice(["generic", "r128,nt1,u2,s1", "r128,nt0,u2,s2", "r128,nt1,u1,s3",
                                         "r256,nt1,u2,s1", "r256,nt0,u1,s2", "r128,nt1,u1,s1", "r128,nt1,u1,s1,c1"]))
        capi.set_tuning(tune["sweepw"])
        case = dict(it=it, bs=bs, nb=nb, rowmajor=rm, avg_offdiag=dens, scaling=sc, tune=tune)
        if rng.integers(0, 4) == 0:
            # a stencil (short rows: what the plan kernels of the exact factorisation take)
            g = int(rng.integers(3, 15))
            m = W.poisson3d(g, bs, rowmajor=rm)
            nb = m["nbrows"]
            case.update(stencil=g, nb=nb)
        else:
            m = W.random_bsr(nb, bs, avg_offdiag=dens, seed=int(rng.integers(1, 1 << 30)), rowmajor=rm)
        n = nb * bs
        r = rng.uniform(-1, 1, n)
        x0 = rng.uniform(-1, 1, n)
        p = capi.Prec(0)
        p.set_matrix(m)
        # integer structures
        pos = O.ilu_positions(m)
        # SpMV
        check("spmv", p.spmv(r), O.spmv(m, r), 1e-12)
        check("gemv3", p.gemv3(0.7, r, -1.3, x0), O.gemv3(m, 0.7, r, -1.3, x0), 1e-12)
        # factorisation: synchronous sweeps and the exact form
        # (INIT_F_ZERO: the reference's own native block cases start from it, tests/CMakeLists.txt:104-111,157-173;
        # its first synchronous sweeps invert zero diagonal blocks -- compare where the oracle's model is finite)
        init = int(rng.choice([capi.INIT_F_ORIGINAL, capi.INIT_F_SGS, capi.INIT_F_ZERO]))
        p.ilu0_factorize(2, init=init, usescale=sc, mode=capi.JACOBI_SYNC)
        want = O.ilu0_factorize(m, pos, 2, mode=O.JACOBI_SYNC, init=init, usescale=sc)
        if np.all(np.isfinite(want["iluvals"])) and np.abs(want["iluvals"]).max() < 1e8:
            check("factor_sync", p.get_iluvals(), want["iluvals"], 1e-10)
        elif init == capi.INIT_F_ZERO:
            wb, gb = want["iluvals"].reshape(-1, bs * bs), p.get_iluvals().reshape(-1, bs * bs)
            fin = np.isfinite(wb).all(axis=1)
            assert np.array_equal(fin, np.isfinite(gb).all(axis=1)), dict(case, what="factor_sync_zero finite blocks")
            if fin.any() and np.abs(wb[fin]).max() < 1e8:
                check("factor_sync_zero", gb[fin], wb[fin], 1e-10)
        p.ilu0_factorize(-1, usescale=sc)
        fe = O.ilu0_factorize(m, pos, 1, mode=O.GS_SERIAL, usescale=sc)
        if not (np.all(np.isfinite(fe["iluvals"])) and np.abs(fe["iluvals"]).max() < 1e8):
            p.close()
            continue
        check("factor_exact", p.get_iluvals(), fe["iluvals"], 1e-9)
        f = p.get_iluvals()
        # in-place asynchronous sweeps run to convergence reach the same factor (with and without the shortcut)
        p.ilu0_factorize(p.level_count() + 3, init=init, usescale=sc, mode=capi.ASYNC)
        check("factor_async", p.get_iluvals(), fe["iluvals"], 1e-8)
        p.ilu0_factorize(-1, usescale=sc)
        scale = p.get_scale() if sc else None
        # apply: synchronous, exact, asynchronous to convergence
        ainit = int(rng.choice([capi.INIT_A_ZERO, capi.INIT_A_JACOBI]))
        check("apply_sync", p.ilu0_apply(r, 3, init=ainit, mode=capi.JACOBI_SYNC),
              O.ilu0_apply(m, f, r, 3, mode=O.JACOBI_SYNC, init=ainit, scale=scale), 1e-10)
        ze = O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL, scale=scale)
        if np.all(np.isfinite(ze)) and np.abs(ze).max() < 1e8:
            check("apply_exact", p.ilu0_apply(r, 1, mode=capi.LEVEL), ze, 1e-9)
            check("apply_async", p.ilu0_apply(r, p.level_count() + 2, mode=capi.ASYNC), ze, 1e-8)
        # Jacobi / SGS / relaxations
        p.jacobi_compute()
        d = p.get_dblocks()
        check("jacobi", p.jacobi_apply(r), O.jacobi_apply(m, d, r), 1e-12)
        check("sgs_sync", p.sgs_apply(r, 2, mode=capi.JACOBI_SYNC), O.sgs_apply(m, d, r, 2, mode=O.JACOBI_SYNC), 1e-10)
        zs, ys = O.sgs_apply(m, d, r, 1, mode=O.GS_SERIAL, return_y=True)
        check("sgs_exact", p.sgs_apply(r, 1, mode=capi.LEVEL), zs, 1e-9)
        if np.all(np.isfinite(zs)) and np.abs(zs).max() < 1e8:
            # the product modes: exact forward half; deterministic = synchronous backward sweeps from it
            check("sgs_determ", p.sgs_apply(r, 2, mode=capi.DETERMINISTIC),
                  O.sgs_apply(m, d, r, 2, mode=O.JACOBI_SYNC, init=O.INIT_A_NONE, y0=ys, z0=np.zeros(n)), 1e-9)
            p.sgs_apply(r, 1, mode=capi.ASYNC)
            check("sgs_async_fwd", p.get_ytemp(), ys, 1e-9)
            check("sgs_async", p.sgs_apply(r, p.level_count() + 2, mode=capi.ASYNC), zs, 1e-8)
        ms = p.memory_stats()
        assert ms["bytes"] > 0 and ms["peak_bytes"] >= ms["bytes"] and (tune["copies"] == "both" or ms["derived_copies"] <= 4), (ms, case)  # (<= 2 triangles each of factor and matrix)
        xr = O.sgs_relax(m, d, r, x0=x0, maxits=2, mode=O.GS_SERIAL)
        if np.all(np.isfinite(xr)) and np.abs(xr).max() < 1e8:
            check("relax_exact", p.sgs_relax(r, x0.copy(), 2, mode=capi.LEVEL), xr, 1e-9)
            check("relax_sync", p.sgs_relax(r, x0.copy(), 2, mode=capi.JACOBI_SYNC),
                  O.sgs_relax(m, d, r, x0=x0, maxits=2, mode=O.JACOBI_SYNC), 1e-10)
            check("gs_exact", p.gs_relax(r, x0.copy(), 2, mode=capi.LEVEL),
                  O.gs_relax(m, d, r, x0=x0, nsweeps=2, mode=O.GS_SERIAL), 1e-9)
            xj = x0.copy()
            p.jacobi_relax(r, xj, 2)
            check("jacobi_relax", xj, O.jacobi_relax(m, d, r, x0=x0, maxits=2)[0], 1e-10)
        assert p.level_stats()["syncfree_aborts"] == 0, case
        p.close()
    for spec in ("level=syncfree", "levelstore=1", "levelwide=1", "levelperm=1", "compact=1", "interleave=0",
                 "sweepodd=1", "sweepodd=nt0", "sweepodd=occ1", "sweepwr=1", "factorodd=1", "factor1=1", "factor4=1",
                 "factor8=1", "gunroll=0", "copies=one", "xcdsuper=auto", "levelserial=4096", "levelfast=1",
                 "factorsf=1", "factorsf=p1", "factorskip=1", "compactafter=0", "factorfuse=1", "latestore=2", "invertrow=1", "scalarlane=auto", "scalarstage=1", "factor1plan=1",
                 "r128,nt1,u1,s1"):
        capi.set_tuning(spec)
    print("%d cases in %.1f s; worst relative differences:" % (ncases, time.time() - t0))
    for k in sorted(worst):
        print("  %-14s %.2e" % (k, worst[k]))
    print("fuzz ok")


if __name__ == "__main__":
    main()
