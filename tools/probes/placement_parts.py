#!/usr/bin/env python3
"""Which part of a build decides whether its sweeps run in the fast or the slow mode
(tools/probes/placement_variance.py)?  Per build: (A) as built, (B) after an idle second, (C) with new rhs /
result vectors, (D) after the library re-made its triangle copies (set_values + factorisation again)."""
import sys
import time

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
grid = 256


def measure(p, r, z, n=8):
    for _ in range(2):
        p.ilu0_apply(r, 3, out=z)
    p.synchronize()
    p.get_timing()
    lo, up = [], []
    for _ in range(n):
        p.ilu0_apply(r, 3, out=z)
        p.synchronize()
        t = p.get_timing()
        lo.append(t["lower_ms"] / t["lower_launches"])
        up.append(t["upper_ms"] / t["upper_launches"])
    return "L %.3f U %.3f" % (np.median(lo), np.median(up))


for rnd in range(8):
    m = W.poisson3d_device(grid, 4, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * 4, dev)
    z = torch.zeros_like(r)
    torch.cuda.synchronize()
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    p.set_timing(True)
    a = measure(p, r, z)
    time.sleep(1.0)
    b = measure(p, r, z)
    keep = [r, z]
    r2 = r.clone()
    z2 = torch.zeros_like(r)
    c = measure(p, r2, z2)
    p.set_values(m["vals"])
    p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    d = measure(p, r2, z2)
    print("round %d | built: %s | idle 1 s: %s | new vectors: %s | refactored: %s" % (rnd, a, b, c, d), flush=True)
    p.close()
    del p, m, r, z, r2, z2, keep
    torch.cuda.empty_cache()
