#!/usr/bin/env python3
"""Sweep time against the DISTANCE between the streamed triangle copy and the vector the sweep writes, both inside
one arena (nothing re-allocated): the upper copy sits at +CG GiB, z moves through the arena in 1 GiB steps; then the
lower copy at +CG GiB and ytemp moving.  (placement_slots.py showed slots far from each other are
all alike.)
usage: placement_pairs.py [N=256] [ARENA_GIB=96] [CG=32]"""
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
AG = int(sys.argv[2]) if len(sys.argv) > 2 else 96
CG = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
M, G = 1 << 20, 1 << 30
n = N ** 3 * 4
nbytes = n * 8
capi.set_tuning("compactafter=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
r0 = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)
nb, nnzb = m["nbrows"], m["nnzb"]
usize = ((nnzb - nb) // 2 + nb) * 128
lsize = ((nnzb - nb) // 2) * 128


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


def place(what, ptr):
    capi._check(L.blasted_hip_probe_place(p._h, what.encode(), C.c_void_p(ptr)))


z0 = torch.zeros(n, dtype=torch.float64, device=dev)
print("baseline (own buffers): lower %.3f upper %.3f" % measure(r0, z0), flush=True)
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()
print("arena %d GiB at %#x" % (AG, A), flush=True)
place("ucopy", A + CG * G)
place("lcopy", A + CG * G + 10 * G)
place("ytemp", A + CG * G + 18 * G)
print("## ucopy at +%d GiB (%.1f GiB long), lcopy at +%d, ytemp at +%d; z moves (GiB from the arena's start: upper ms)" % (
    CG, usize / G, CG + 10, CG + 18), flush=True)
for k2 in range(0, 2 * AG - 1):
    off = k2 * 512 * M
    if off + nbytes > CG * G and off < CG * G + 19 * G:
        continue
    z = arena[off:off + nbytes].view(torch.float64)
    lo, up = measure(r0, z)
    print("z +%5.1f GiB upper %.3f (lower %.3f)" % (off / G, up, lo), flush=True)
zfix = arena[:nbytes].view(torch.float64)
print("## ytemp moves (z fixed at +0): lower / upper ms", flush=True)
for k2 in range(1, 2 * AG - 1):
    off = k2 * 512 * M
    if off + nbytes > CG * G and off < CG * G + 17 * G:
        continue
    place("ytemp", A + off)
    lo, up = measure(r0, zfix)
    print("ytemp +%5.1f GiB lower %.3f upper %.3f" % (off / G, lo, up), flush=True)
place("ytemp", 0)
place("ucopy", 0)
place("lcopy", 0)
p.close()
