#!/usr/bin/env python3
"""Which ADDRESS BITS of a buffer's backing decide the sweeps' mode?    placement_streams.py:
byte offsets up to 256 MiB inside one allocation change nothing, a new allocation at the same virtual address does.
Here one large arena is allocated once and each buffer is placed at slots of it, 512 MiB (vectors) or 2 GiB / 512 MiB
(triangle copies) apart, nothing is re-allocated in between: the sweep time per slot shows at which granularity the
mode changes and whether it is a property of the slot or of the pair (z slot, copy slot).
usage: placement_slots.py [N=256] [ARENA_GIB=96]"""
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


def measure(r, z, reps=4):
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
print("baseline (operator's own buffers): lower %.3f upper %.3f" % measure(r0, z0), flush=True)
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()
print("arena %d GiB at %#x" % (AG, A), flush=True)


def zslot(k):
    return arena[k * 512 * M:][:nbytes].view(torch.float64)


print("## z at 512 MiB slots of the arena (operator's own copies)", flush=True)
for k in range(0, min(2 * AG, 96)):
    lo, up = measure(r0, zslot(k), reps=3)
    print("z slot %3d (+%5.1f GiB) lower %.3f upper %.3f" % (k, k / 2, lo, up), flush=True)
print("## ucopy at 2 GiB slots (z = own allocation z0), then at 512 MiB steps inside the first slots", flush=True)
nslots = (AG * G - usize) // (2 * G)
for k in range(nslots):
    place("ucopy", A + k * 2 * G)
    lo, up = measure(r0, z0, reps=3)
    print("ucopy slot %2d (+%3d GiB) lower %.3f upper %.3f" % (k, 2 * k, lo, up), flush=True)
for k in range(0, 12):
    place("ucopy", A + k * 512 * M)
    lo, up = measure(r0, z0, reps=3)
    print("ucopy +%4.1f GiB lower %.3f upper %.3f" % (k / 2, lo, up), flush=True)
# pair structure: a few copy slots x a few z slots (z slots beyond the copy)
print("## pairs: upper sweep ms for ucopy slot (rows) x z slot (columns; 512 MiB slots counted from the arena's end)", flush=True)
zk = [2 * AG - 1 - j for j in range(8)]
print("          " + " ".join("z%-5d" % k for k in zk), flush=True)
for k in range(0, min(nslots, 8)):
    place("ucopy", A + k * 2 * G)
    print("ucopy %2d  " % k + " ".join("%.3f " % measure(r0, zslot(j), reps=3)[1] for j in zk), flush=True)
place("ucopy", 0)
print("## lcopy at 2 GiB slots; then ytemp at 512 MiB slots (lcopy back in the operator's own allocation)", flush=True)
for k in range((AG * G - lsize) // (2 * G)):
    place("lcopy", A + k * 2 * G)
    lo, up = measure(r0, z0, reps=3)
    print("lcopy slot %2d (+%3d GiB) lower %.3f upper %.3f" % (k, 2 * k, lo, up), flush=True)
place("lcopy", 0)
for k in range(0, min(2 * AG, 48)):
    place("ytemp", A + k * 512 * M)
    lo, up = measure(r0, z0, reps=3)
    print("ytemp slot %3d (+%5.1f GiB) lower %.3f upper %.3f" % (k, k / 2, lo, up), flush=True)
place("ytemp", 0)
p.close()
