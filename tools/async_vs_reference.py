#!/usr/bin/env python3
"""What s asynchronous sweeps are worth, reference side and GPU side on the SAME matrices (SURVEY 8d tier P4).

For each matrix: the exact ILU(0) factor (oracle, one serial sweep) and the exact application z* = U^-1 L^-1 r;
then, for s in {1, 2, 3, 5, 10}, the relative 2-norm distance of z to z* after s lower + s upper sweeps of
  * the oracle's ASYNC_OMP -- the reference's threaded loop nest (omp for schedule(dynamic, 256) nowait inside
    one parallel region, src/solverops_ilu0.cpp:99-118) on this box's host cores (median of 3 runs: it is
    nondeterministic),
  * the oracle's JACOBI_SYNC -- the deterministic worst case (a sweep sees nothing of the same sweep),
  * HIP ASYNC, the product's chaotic sweeps (what bench.py measures), default row order and interleave=1,
and the contraction per sweep between s = 3 and s = 10, with the time of a sweep pair on each side.
Same for the factorisation: distance of the off-diagonal blocks to the exact factor after 1, 3, 5 build sweeps.

usage: async_vs_reference.py [out.txt] [sizes=96,128] [golden=tests/golden]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
import oracle as O  # noqa: E402  (the checker: this is a measurement tool, not product code)
from blasted_amd import capi, mtxio, workloads as W  # noqa: E402

SWEEPS = (1, 2, 3, 5, 10)


def dist(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def contraction(errs):
    e3, e10 = errs.get(3), errs.get(10)
    if e3 and e10 and e10 > 1e-14 and e3 > e10:
        return (e10 / e3) ** (1.0 / 7)
    return None


def fmt(errs):
    return "  ".join("%9.2e" % errs[s] for s in SWEEPS)


def one_matrix(name, m, out):
    bs = m["bs"]
    n = m["nbrows"] * bs
    r = W.rhs_vector(n)
    plist = O.ilu_positions(m)
    exact_f = O.ilu0_factorize(m, plist, 1, mode=O.GS_SERIAL)["iluvals"]
    zex = O.ilu0_apply(m, exact_f, r, 1, mode=O.GS_SERIAL)
    threads = O.num_threads()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)
    emit("== %s: %d block-rows, bs=%d, %d blocks, %d dependency levels" % (
        name, m["nbrows"], bs, m["nnzb"], int(W.dependency_levels(m).max()) + 1 if m["nbrows"] <= 300000 else -1))

    # ---- apply: reference side
    ref, sync = {}, {}
    t_ref = None
    for s in SWEEPS:
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            z = O.ilu0_apply(m, exact_f, r, s, mode=O.ASYNC_OMP, init=O.INIT_A_ZERO, chunk=256)
            dt = time.perf_counter() - t0
            runs.append(dist(z, zex))
            if s == 10:
                t_ref = dt / s * 1e3 if t_ref is None else min(t_ref, dt / s * 1e3)
        ref[s] = float(np.median(runs))
        sync[s] = dist(O.ilu0_apply(m, exact_f, r, s, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO), zex)
    O.set_num_threads(1)
    ser1 = dist(O.ilu0_apply(m, exact_f, r, 1, mode=O.ASYNC_OMP, init=O.INIT_A_ZERO, chunk=256), zex)
    O.set_num_threads(threads)

    # ---- apply: GPU side (the exact factor on the device: one exact factorisation)
    p = capi.Prec(0)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    rd = torch.from_numpy(r).cuda()
    zd = torch.empty_like(rd)
    hip = {}
    t_hip = {}
    for spec in ("interleave=0", "interleave=1"):
        capi.set_tuning(spec)
        errs = {}
        for s in SWEEPS:
            p.ilu0_apply(rd, s, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=zd)
            errs[s] = dist(zd.cpu().numpy(), zex)
        p.ilu0_apply(rd, 10, mode=capi.ASYNC, out=zd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            p.ilu0_apply(rd, 10, mode=capi.ASYNC, out=zd)
        torch.cuda.synchronize()
        t_hip[spec] = (time.perf_counter() - t0) / 50 * 1e3
        hip[spec] = errs
    capi.set_tuning("interleave=0")
    zlev = p.ilu0_apply(rd, 1, mode=capi.LEVEL, out=zd).cpu().numpy()
    p.ilu0_apply(rd, 1, mode=capi.LEVEL, out=zd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        p.ilu0_apply(rd, 1, mode=capi.LEVEL, out=zd)
    torch.cuda.synchronize()
    t_exact = (time.perf_counter() - t0) / 5 * 1e3

    emit("ILU(0) application, relative distance to the exact triangular solves after s+s sweeps")
    emit("%-58s %s   contraction/sweep (3->10)   ms per sweep pair" % ("s =", "  ".join("%9d" % s for s in SWEEPS)))
    rows = [("reference loop nest, ASYNC_OMP, %d threads, chunk 256" % threads, ref, t_ref),
            ("synchronous Jacobi sweeps (oracle JACOBI_SYNC)", sync, None),
            ("HIP ASYNC (bench.py's mode), default row order", hip["interleave=0"], t_hip["interleave=0"]),
            ("HIP ASYNC, interleave=1", hip["interleave=1"], t_hip["interleave=1"])]
    for label, errs, t in rows:
        c = contraction(errs)
        emit("%-58s %s   %-27s %s" % (label, fmt(errs), "%.3f" % c if c else "(converged before 10)",
                                      "%.3f" % t if t else "-"))
    emit("reference loop nest at ONE thread, 1+1 sweeps: %.2e (the exact solve); HIP exact (LEVEL) application: "
         "%.2e from the oracle's, %.3f ms" % (ser1, dist(zlev, zex), t_exact))

    # ---- factorisation: off-diagonal blocks against the exact factor
    bs2 = bs * bs
    off = np.ones(m["nnzb"], dtype=bool)
    off[np.asarray(m["diagind"])] = False
    ex_off = exact_f.reshape(-1, bs2)[off]
    fref, fsync, fhip = {}, {}, {}
    for s in (1, 3, 5):
        runs = [dist(O.ilu0_factorize(m, plist, s, mode=O.ASYNC_OMP, chunk=256)["iluvals"].reshape(-1, bs2)[off], ex_off)
                for _ in range(3)]
        fref[s] = float(np.median(runs))
        fsync[s] = dist(O.ilu0_factorize(m, plist, s, mode=O.JACOBI_SYNC)["iluvals"].reshape(-1, bs2)[off], ex_off)
        p.ilu0_factorize(s, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
        fhip[s] = dist(p.get_iluvals().reshape(-1, bs2)[off], ex_off)
    emit("ILU(0) factorisation from the matrix (INIT_F_ORIGINAL), distance of the off-diagonal blocks to the exact factor")
    emit("%-58s %9d  %9d  %9d" % ("build sweeps =", 1, 3, 5))
    for label, e in (("reference loop nest, ASYNC_OMP, %d threads" % threads, fref),
                     ("synchronous Jacobi sweeps", fsync), ("HIP ASYNC", fhip)):
        emit("%-58s %9.2e  %9.2e  %9.2e" % (label, e[1], e[3], e[5]))
    emit("")
    p.close()
    out.write("\n".join(lines) + "\n")
    out.flush()
    return {"ref": ref, "sync": sync, "hip": hip["interleave=0"], "hip_interleave": hip["interleave=1"]}


def main():
    outp = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_async_vs_reference.txt")
    sizes = [int(x) for x in ([a[6:] for a in sys.argv[2:] if a.startswith("sizes=")] or ["96,128"])[-1].split(",")]
    golden = ([a[7:] for a in sys.argv[2:] if a.startswith("golden=")] or [os.path.join(ROOT, "tests", "golden")])[-1]
    O.set_num_threads(O.cpu_budget())
    os.makedirs(os.path.dirname(outp), exist_ok=True)
    with open(outp, "w") as out:
        hdr = ("# tools/async_vs_reference.py on %s; oracle ASYNC_OMP with %d OpenMP threads (the CPUs granted to this "
               "process), chunk 256; r_i = sin(0.37 i) + 1.1; y0 = z0 = 0" % (torch.cuda.get_device_name(0), O.num_threads()))
        print(hdr)
        out.write(hdr + "\n")
        one_matrix("2dcyl1 bs=4 (the reference's CFD fixture)", mtxio.read_mtx_bsr(os.path.join(golden, "2dcyl1.mtx"), 4, False), out)
        for n in sizes:
            one_matrix("Poisson %d^3 bs=4 (bench generator, uniform grid)" % n, W.poisson3d(n + 2, 4, grid="uniform"), out)


if __name__ == "__main__":
    main()
