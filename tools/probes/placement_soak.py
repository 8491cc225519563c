#!/usr/bin/env python3
"""Stress of the class-aware placement path: many operators created, applied and destroyed in ONE process (address
ranges are never reused, physical pieces come and go), thorough and quick searches alternating, results checked against
the first operator's (synchronous sweeps: same bits every time).  usage: placement_soak.py [N=128] [ROUNDS=24]"""
import sys
import time

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
capi.set_tuning("compactafter=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
n = m["nbrows"] * 4
r = W.rhs_vector_device(n, dev)
ref = None
t0 = time.perf_counter()
for k in range(ROUNDS):
    capi.set_tuning("placement=%d" % (2 if k % 2 == 0 else 1))
    p = capi.Prec(0)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    z = torch.zeros_like(r)
    p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC, out=z)
    p.ilu0_apply(r, 1, mode=capi.LEVEL)            # the level ordering takes the copies' storage over
    za = p.ilu0_apply(r, 3, mode=capi.ASYNC)       # ... and hands it back
    assert torch.isfinite(za).all()
    z2 = torch.zeros_like(r)
    p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC, out=z2)
    if ref is None:
        ref = z.clone()
    assert torch.equal(z, ref) and torch.equal(z2, ref), k
    if k % 3 == 0:
        p.ilu0_factorize(3)                        # refactorisation keeps the placed storage
        p.ilu0_apply(r, 3, mode=capi.ASYNC)
    p.close()
    print("round %2d ok, %.1f s, %s" % (k, time.perf_counter() - t0, capi.placement_stats()), flush=True)
print("soak ok")
