import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
from blasted_amd import capi, workloads as W
dev = torch.device("cuda:0")
m = W.poisson3d_device(256, 4, dev, grid="uniform")
r = W.rhs_vector_device(m["nbrows"] * 4, dev)
z = torch.zeros_like(r)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(3)
for mode, name in ((capi.ASYNC, "async"), (capi.JACOBI_SYNC, "sync"), (capi.ASYNC, "async"), (capi.JACOBI_SYNC, "sync")):
    for _ in range(3):
        p.ilu0_apply(r, 3, mode=mode, out=z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        p.ilu0_apply(r, 3, mode=mode, out=z)
    torch.cuda.synchronize()
    print("%s 3+3 apply: %.3f ms" % (name, (time.perf_counter() - t0) * 100))
