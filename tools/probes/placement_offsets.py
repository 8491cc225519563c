#!/usr/bin/env python3
"""Upper-sweep time against the placement of the result vector INSIDE one allocation: z is a view at
different byte offsets of one 3 GB arena (same physical backing, shifted addresses)."""
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
for arena_round in range(3):
    arena = torch.zeros(3 << 30, dtype=torch.uint8, device=dev)
    line = []
    for off in [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 100 << 20, 256 << 20, 1 << 30, (1 << 30) + 4096 * 37]:
        z = arena[off:off + nbytes].view(torch.float64)
        for _ in range(2):
            p.ilu0_apply(r, 3, out=z)
        p.synchronize()
        p.get_timing()
        up, lo = [], []
        for _ in range(6):
            p.ilu0_apply(r, 3, out=z)
            p.synchronize()
            t = p.get_timing()
            up.append(t["upper_ms"] / t["upper_launches"])
            lo.append(t["lower_ms"] / t["lower_launches"])
        line.append("%#x: U %.3f (L %.3f)" % (off, np.median(up), np.median(lo)))
    print("arena %d at %#x\n   " % (arena_round, arena.data_ptr()) + "\n   ".join(line), flush=True)
    del arena, z
    torch.cuda.empty_cache()
    spacer = torch.empty((arena_round + 1) * (77 << 20), dtype=torch.uint8, device=dev)
