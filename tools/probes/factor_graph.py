#!/usr/bin/env python3
"""Exact (per-level) factorisation as stream launches against one hipGraph of the same launches
(BLASTED_HIP_FACTOR_GRAPH=1 prints the graph's device time)."""
import sys
import time
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
m = W.poisson3d_device(256, 4, dev, grid="uniform")
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.ilu0_factorize(-1)
    p.synchronize()
    print("exact factorisation call: %.2f ms wall" % ((time.perf_counter() - t0) * 1e3), flush=True)
