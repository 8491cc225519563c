#!/usr/bin/env python3
"""Times the lower / upper asynchronous sweep kernels of the headline configuration with one of the
timing-only library variants (tools/probes/build_diag.sh).  usage: diag_sweep.py <0|1|2|3> [grid=256]"""
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

d = int(sys.argv[1])
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if d:
    capi.LIBPATH = capi.LIBPATH.replace("libblasted_hip.so", "libblasted_hip_diag%d.so" % d)
dev = torch.device("cuda:0")
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
for spec in sys.argv[3:] or ["interleave=0"]:
  capi.set_tuning(spec)
  lo, up = [], []
  for _ in range(40):
    p.ilu0_apply(r, 3, out=z)
    p.synchronize()
    t = p.get_timing()
    lo.append(t["lower_ms"] / t["lower_launches"])
    up.append(t["upper_ms"] / t["upper_launches"])
  lo, up = np.sort(lo), np.sort(up)
  print("diag %d %s: lower min %.3f median %.3f ms, upper min %.3f median %.3f ms per sweep"
        % (d, spec, lo[0], lo[len(lo) // 2], up[0], up[len(up) // 2]))
