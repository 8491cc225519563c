#!/usr/bin/env python3
"""Is the process-to-process spread of the sweep times (+-8 %) tied to where the buffers land?  Builds
the 256^3 bs=4 problem several times in ONE process (freeing everything in between, optionally with a
spacer allocation that shifts the addresses) and times the sweeps each time."""
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
VARIANTS = [v for v in sys.argv[2:] if not v.startswith("n=")] or ["r128,nt1,u1,s1"]
NAPPLY = int(([v[2:] for v in sys.argv[2:] if v.startswith("n=")] or ["12"])[0])
spacers = [0, 0, 1 << 20, 3 << 20, 64 << 20, (1 << 30) + (5 << 20), 0, 7 << 20, 300 << 20, 0, (2 << 30) + (11 << 20), 33 << 20]
for rnd, sp in enumerate(spacers):
    spacer = torch.empty(sp, dtype=torch.uint8, device=dev) if sp else None
    m = W.poisson3d_device(grid, 4, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * 4, dev)
    z = torch.zeros_like(r)
    torch.cuda.synchronize()
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    for _ in range(3):
        p.ilu0_apply(r, 3, out=z)
    p.set_timing(True)
    line = "round %2d:" % rnd
    for spec in VARIANTS:
        capi.set_tuning(spec)
        lo, up = [], []
        for _ in range(NAPPLY):
            p.ilu0_apply(r, 3, out=z)
            p.synchronize()
            t = p.get_timing()
            lo.append(t["lower_ms"] / t["lower_launches"])
            up.append(t["upper_ms"] / t["upper_launches"])
        line += "  %s: L %.3f U %.3f" % (spec.split(",")[0], np.median(lo), np.median(up))
    print(line, flush=True)
    p.close()
    del p, m, r, z, spacer
    torch.cuda.empty_cache()
