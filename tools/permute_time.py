#!/usr/bin/env python3
"""Cost of the copy pass that refreshes the compact / level-ordered copies of the factor after a
factorisation (first apply after factorize minus a steady-state apply), 256^3 bs=4."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W
dev = torch.device("cuda", 0)
m = W.poisson3d_device(256, 4, dev, grid="uniform")
r = W.rhs_vector_device(m["nbrows"] * 4, dev); z = torch.zeros_like(r)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream); p.set_matrix(m)
capi.set_tuning("compact=1")
for rep in range(3):
    p.ilu0_factorize(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); p.ilu0_apply(r, 1, out=z); torch.cuda.synchronize(); t1 = time.perf_counter()
    p.ilu0_apply(r, 1, out=z); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("first apply after factorize %.2f ms, next %.2f ms -> compact copy pass %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t1-t0-(t2-t1))*1e3))
    p.ilu0_factorize(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z); torch.cuda.synchronize(); t1 = time.perf_counter()
    p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("first LEVEL apply after factorize %.2f ms, next %.2f ms -> level copy pass %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t1-t0-(t2-t1))*1e3))
