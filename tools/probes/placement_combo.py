#!/usr/bin/env python3
"""Upper / lower sweep time against the address CLASSES of all three vector-sized streams: one arena, its 2 GiB slots
classified with the read-beside-write probe; the triangle copies in a run of slots of one class, and ytemp, z, r
each in a slot of a chosen class.  Which combinations are fast?
usage: placement_combo.py [N=256] [ARENA_GIB=200]"""
import ctypes as C
import itertools
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
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
L.blasted_hip_probe_rw.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
M, G = 1 << 20, 1 << 30
n = N ** 3 * 4
nbytes = n * 8
capi.set_tuning("compactafter=0")
capi.set_tuning("placement=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
r0 = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()


def rw_ms(rd, rd_bytes, wr, wr_bytes, reps=4):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_rw(C.c_void_p(rd), rd_bytes, C.c_void_p(wr), wr_bytes, reps, C.byref(out)))
    return out.value


slots = list(range(0, AG - 1, 2))
cls, reps_of = {}, []
for k in slots:
    base = A + k * G
    for c, kr in enumerate(reps_of):
        ref = A + kr * G
        if rw_ms(ref, 2 * G, base, 128 * M) > 0.955 * rw_ms(ref, 2 * G, ref + 2 * G - 128 * M, 128 * M):
            cls[k] = c
            break
    else:
        cls[k] = len(reps_of)
        reps_of.append(k)
names = "ABCDEFGH"
print("classes of the 2 GiB slots: " + "".join(names[cls[k]] for k in slots), flush=True)


def run_of(c, need_gib, skip=()):
    """first run of consecutive slots of class c that is need_gib long and does not touch the slots in `skip`"""
    cnt = 0
    for k in slots:
        cnt = cnt + 1 if (cls[k] == c and k not in skip) else 0
        if cnt * 2 >= need_gib:
            return k - 2 * (cnt - 1)
    return None


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


ncls = len(reps_of)
print("lower = the lower sweep (streams the lower copy, reads r, writes ytemp); upper = the upper sweep (streams the upper copy, "
      "reads ytemp, writes z)", flush=True)
for cc in range(min(ncls, 2)):               # the class both triangle copies are in
    ku = run_of(cc, 10)
    if ku is None:
        continue
    kl = run_of(cc, 8, skip=set(range(ku, ku + 10, 2)))
    if kl is None:
        continue
    used = set(range(ku, ku + 10, 2)) | set(range(kl, kl + 8, 2))
    place("ucopy", A + ku * G)
    place("lcopy", A + kl * G)
    for cy, cz, cr in itertools.product(range(ncls), repeat=3):
        ks = []
        taken = set(used)
        for c in (cy, cz, cr):
            k = next((k for k in slots if cls[k] == c and k not in taken), None)
            ks.append(k)
            taken.add(k)
        if None in ks:
            continue
        place("ytemp", A + ks[0] * G)
        z = arena[ks[1] * G:ks[1] * G + nbytes].view(torch.float64)
        r = arena[ks[2] * G:ks[2] * G + nbytes].view(torch.float64)
        r.copy_(r0)
        lo, up = measure(r, z)
        print("copies %s | ytemp %s  z %s  r %s : lower %.3f  upper %.3f" % (names[cc], names[cy], names[cz], names[cr], lo, up), flush=True)
place("ytemp", 0)
place("ucopy", 0)
place("lcopy", 0)
p.close()
