#!/usr/bin/env python3
"""Which buffer's placement decides the fast / slow mode of the 256^3 bs=4 triangular sweeps?  
ONE operator; one buffer at a time is moved -- the caller's z and r as views at byte offsets of an arena, the
operator's ytemp / upper copy / lower copy re-allocated at byte offsets inside a larger block ("allocoff", probes
build) -- and the lower / upper sweep are timed after every move.  The baseline is re-measured between scans.
usage: BLASTED_HIP_PROBES=1 placement_streams.py [N=256]"""
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
dev = torch.device("cuda:0")
capi.set_tuning("compactafter=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
n = m["nbrows"] * 4
nbytes = n * 8
r0 = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)
L = capi.lib()
L.blasted_hip_probe_move.argtypes = [C.c_void_p, C.c_char_p]
L.blasted_hip_probe_addresses.argtypes = [C.c_void_p, C.c_void_p]


def addresses():
    out = (C.c_ulong * 6)()
    capi._check(L.blasted_hip_probe_addresses(p._h, out))
    return dict(zip(("ytemp", "lcopy", "ucopy", "iluvals", "lcol", "ucol"), [int(v) for v in out]))


def measure(r, z, reps=5):
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


def move(what, off):
    capi.set_tuning("allocoff=%d" % off)
    capi._check(L.blasted_hip_probe_move(p._h, what.encode()))
    capi.set_tuning("allocoff=0")


K, M = 1 << 10, 1 << 20
OFFS = [0, 256, 512, 1 * K, 2 * K, 4 * K, 8 * K, 16 * K, 32 * K, 64 * K, 128 * K, 256 * K, 512 * K, 1 * M, 2 * M, 3 * M, 4 * M, 8 * M,
        16 * M, 32 * M, 64 * M, 128 * M, 256 * M, 37 * 4 * K, 1 * M + 64 * K, 2 * M + 4 * K, 0]
z0 = torch.zeros(n, dtype=torch.float64, device=dev)
a = addresses()
print("N=%d  r %#x  z %#x  " % (N, r0.data_ptr(), z0.data_ptr()) + "  ".join("%s %#x" % kv for kv in a.items()), flush=True)
print("baseline: lower %.3f upper %.3f" % measure(r0, z0), flush=True)

arena = torch.zeros(nbytes + 300 * M, dtype=torch.uint8, device=dev)
print("## z = view of one arena (%#x) at byte offsets; r, operator fixed" % arena.data_ptr(), flush=True)
for off in OFFS:
    z = arena[off:off + nbytes].view(torch.float64)
    lo, up = measure(r0, z)
    print("z   +%-10d lower %.3f upper %.3f" % (off, lo, up), flush=True)
print("## r = view of the arena at byte offsets; z, operator fixed", flush=True)
for off in OFFS:
    r = arena[off:off + nbytes].view(torch.float64)
    r.copy_(r0)
    lo, up = measure(r, z0)
    print("r   +%-10d lower %.3f upper %.3f" % (off, lo, up), flush=True)
del arena
torch.cuda.empty_cache()
print("baseline again: lower %.3f upper %.3f" % measure(r0, z0), flush=True)
for what in ("ytemp", "ucopy", "lcopy"):
    print("## %s re-allocated at byte offsets inside a larger block; everything else fixed" % what, flush=True)
    for off in OFFS:
        move(what, off)
        lo, up = measure(r0, z0)
        print("%-5s +%-10d at %#x lower %.3f upper %.3f" % (what, off, addresses()[what], lo, up), flush=True)
    print("baseline again: lower %.3f upper %.3f" % measure(r0, z0), flush=True)
# fresh caller vectors, several of them (the round-2 observation: the mode is a property of z under one operator)
print("## fresh z vectors (own allocations)", flush=True)
zs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(8)]
for i, z in enumerate(zs):
    lo, up = measure(r0, z)
    print("z#%d at %#x lower %.3f upper %.3f" % (i, z.data_ptr(), lo, up), flush=True)
p.close()
