import sys, time, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W
dev = torch.device("cuda", 0)
m = W.poisson3d_device(256, 4, dev, grid="uniform")
r = W.rhs_vector_device(m["nbrows"] * 4, dev); z = torch.zeros_like(r)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream); p.set_matrix(m); p.ilu0_factorize(3)
for _ in range(3): p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
p.set_timing(True); p.get_timing(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
torch.cuda.synchronize(); t1 = time.perf_counter()
t = p.get_timing(reset=True)
print("wall per apply %.3f ms; GPU phases per apply: lower %.3f ms, upper %.3f ms, other %.3f ms" % ((t1-t0)/10*1e3, t["lower_ms"]/10, t["upper_ms"]/10, t["other_ms"]/10))
