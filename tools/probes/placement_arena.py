#!/usr/bin/env python3
"""Does a result vector carved from a LARGE allocation avoid the slow mode that separately allocated
537 MB vectors fall into now and then?  One build of the 256^3 bs=4 problem; 12 rounds of: a fresh
537 MB tensor, then a view into a fresh 2 GiB / 4 GiB arena (allocations of earlier rounds are kept or
freed at random so that the free lists change)."""
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
m = W.poisson3d_device(256, 4, dev, grid="uniform")
n = m["nbrows"] * 4
r = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)
nbytes = n * 8
rng = np.random.default_rng(5)


def upper(z):
    for _ in range(2):
        p.ilu0_apply(r, 3, out=z)
    p.synchronize()
    p.get_timing()
    up = []
    for _ in range(6):
        p.ilu0_apply(r, 3, out=z)
        p.synchronize()
        t = p.get_timing()
        up.append(t["upper_ms"] / t["upper_launches"])
    return np.median(up)


kept = []
for rnd in range(12):
    zs = torch.zeros(n, dtype=torch.float64, device=dev)
    a = upper(zs)
    arena2 = torch.zeros(2 << 30, dtype=torch.uint8, device=dev)
    b = upper(arena2[:nbytes].view(torch.float64))
    arena4 = torch.zeros(4 << 30, dtype=torch.uint8, device=dev)
    c = upper(arena4[:nbytes].view(torch.float64))
    print("round %2d: own 537 MB tensor %.3f | in a 2 GiB arena %.3f | in a 4 GiB arena %.3f" % (rnd, a, b, c), flush=True)
    for t in (zs, arena2, arena4):
        if rng.integers(0, 3) == 0 and len(kept) < 6:
            kept.append(t)
    if kept and rng.integers(0, 2):
        kept.pop(int(rng.integers(0, len(kept))))
    del zs, arena2, arena4
    torch.cuda.empty_cache()
