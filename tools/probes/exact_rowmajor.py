#!/usr/bin/env python3
"""Exact (level-scheduled) ILU application and exact factorisation of the same matrix with column- and row-major
blocks: which of them still go through the general single-launch kernels?  usage: exact_rowmajor.py [n=128]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for bs, nn in ((4, n), (8, int(n * 0.63)), (5, int(n * 0.8))):
    m = W.poisson3d_device(nn, bs, dev)
    mr = dict(m)
    mr["vals"] = m["vals"].view(-1, bs, bs).transpose(1, 2).contiguous().view(-1)
    mr["rowmajor"] = True
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(r)
    for name, mm in (("col-major", m), ("ROW-major", mr)):
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(mm)
        fx = t(lambda: p.ilu0_factorize(-1), 3)
        ap = t(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z))
        p.jacobi_compute()
        sg = t(lambda: p.sgs_apply(r, 1, mode=capi.LEVEL, out=z))
        print("%d^3 bs=%d %s: exact factorisation %.3f ms, exact ILU apply %.3f ms, exact SGS apply %.3f ms" % (nn, bs, name, fx, ap, sg), flush=True)
        p.close()
