#!/usr/bin/env python3
"""Launch-bound sizes: one asynchronous ILU(0) application (fills + 3+3 sweeps = 8 launches) as stream launches
against the same launches replayed as ONE graph (captured here through torch.cuda.CUDAGraph on the operator's own
stream).  usage: apply_graph.py [n=64] [bs=1] [sweeps=3]"""
import sys
import time
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m = W.poisson3d_device(n, bs, dev, grid="uniform")
    p = capi.Prec(0, s.cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(3)
    r = W.rhs_vector_device(n ** 3 * bs, dev)
    z = torch.empty_like(r)
    for _ in range(3):
        p.ilu0_apply(r, sweeps, out=z)
    s.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        p.ilu0_apply(r, sweeps, out=z)
    s.synchronize()
    t_stream = (time.perf_counter() - t0) / reps
    zref = z.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        p.ilu0_apply(r, sweeps, out=z)
    for _ in range(3):
        g.replay()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    s.synchronize()
    t_graph = (time.perf_counter() - t0) / reps
    diff = float((z - zref).norm() / zref.norm())
print("poisson %d^3 bs=%d, %d+%d asynchronous sweeps: %.1f us per application as stream launches, %.1f us as one graph "
      "(relative difference of the results, chaotic sweeps: %.1e)" % (n, bs, sweeps, sweeps, t_stream * 1e6, t_graph * 1e6, diff))
p.close()
