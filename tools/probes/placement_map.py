#!/usr/bin/env python3
"""Map of the sweeps' mode over a large arena: the upper triangle copy sits at +CG GiB of one ARENA_GIB allocation,
the result vector z moves through the whole arena in 1 GiB steps (nothing is re-allocated); repeated for several CG.
placement_pairs.py found ONE sharp boundary (z on the copy's side of +64 GiB of a 96 GiB arena: slow, beyond: fast).
usage: placement_map.py [N=256] [ARENA_GIB=200] [CG,CG,...=0,100,150]"""
import ctypes as C
import os
import sys

os.environ["BLASTED_HIP_PROBES"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AG = int(sys.argv[2]) if len(sys.argv) > 2 else 200
CGS = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,100,150").split(",")]
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
M, G = 1 << 20, 1 << 30
n = N ** 3 * 4
nbytes = n * 8
capi.set_tuning("compactafter=0")
free0, total = torch.cuda.mem_get_info()
m = W.poisson3d_device(N, 4, dev, grid="uniform")
r0 = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)
nb, nnzb = m["nbrows"], m["nnzb"]
usize = ((nnzb - nb) // 2 + nb) * 128


def measure(r, z, reps=3):
    for _ in range(2):
        p.ilu0_apply(r, 3, out=z)
    p.synchronize()
    p.get_timing()
    lo, up = [], []
    for _ in range(reps):
        p.ilu0_apply(r, 3, out=z)
        p.synchronize()
        t = p.get_timing()
        lo.append(t["lower_ms"] / t["lower_launches"])
        up.append(t["upper_ms"] / t["upper_launches"])
    return float(np.median(lo)), float(np.median(up))


z0 = torch.zeros(n, dtype=torch.float64, device=dev)
print("device memory: %.1f GiB total, %.1f GiB free at start; baseline (own buffers): lower %.3f upper %.3f" % (
    (total / G, free0 / G) + measure(r0, z0)), flush=True)
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()
print("arena %d GiB at %#x" % (AG, A), flush=True)
for cg in CGS:
    capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", C.c_void_p(A + cg * G)))
    row = []
    for k in range(0, AG):
        off = k * G
        if off + nbytes > cg * G and off < cg * G + usize:
            row.append("  -  ")
            continue
        z = arena[off:off + nbytes].view(torch.float64)
        row.append("%.3f" % measure(r0, z)[1])
    print("## ucopy at +%d GiB: upper sweep ms with z at +0, +1, ... GiB" % cg, flush=True)
    for i in range(0, AG, 16):
        print("  +%3d: " % i + " ".join(row[i:i + 16]), flush=True)
capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", C.c_void_p(0)))
p.close()
