#!/usr/bin/env python3
"""Does a READ stream run faster when its pieces alternate between two address classes (placement_rwprobe.py: three
classes of 96 GiB, presumably the three ranks of the 12-high HBM3E stacks) than when it stays inside one?  One arena;
classes found with the read-beside-write probe; a read stream of 2 x 4 GiB whose 64 KiB ... 256 MiB pieces alternate
between two 4 GiB buffers of the same / of different classes.
usage: placement_interleave.py [ARENA_GIB=200]"""
import ctypes as C
import os
import sys

os.environ["BLASTED_HIP_PROBES"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from blasted_amd import capi  # noqa: E402

AG = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_rw.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
L.blasted_hip_probe_read2.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
M, G = 1 << 20, 1 << 30
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()


def rw_ms(rd, rd_bytes, wr, wr_bytes, reps=5):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_rw(C.c_void_p(rd), rd_bytes, C.c_void_p(wr), wr_bytes, reps, C.byref(out)))
    return out.value


def read2(p0, p1, each, piece, reps=5):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_read2(C.c_void_p(p0), C.c_void_p(p1), each, piece, reps, C.byref(out)))
    return out.value


# classes of the arena's 4 GiB slots: slot k is of the class of the first slot that "sees" it as itself
slots = list(range(0, AG - 3, 4))
cls = {}
reps_of = []
for k in slots:
    base = A + k * G
    self_ms = rw_ms(base, 2 * G, base + 3 * G, 128 * M)
    for c, kr in enumerate(reps_of):
        if rw_ms(A + kr * G, 2 * G, base, 128 * M) > 0.95 * rw_ms(A + kr * G, 2 * G, A + kr * G + 3 * G, 128 * M):
            cls[k] = c
            break
    else:
        cls[k] = len(reps_of)
        reps_of.append(k)
print("classes of the 4 GiB slots: " + " ".join("%d:%s" % (k, "ABCDEFGH"[cls[k]]) for k in slots), flush=True)
byc = {}
for k in slots:
    byc.setdefault(cls[k], []).append(k)
names = sorted(byc)
print("slots per class: " + ", ".join("%s %d" % ("ABCDEFGH"[c], len(byc[c])) for c in names), flush=True)
pairs = []
for c in names:
    if len(byc[c]) >= 2:
        pairs.append(("same class %s" % "ABCDEFGH"[c], byc[c][0], byc[c][1]))
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        pairs.append(("classes %s+%s" % ("ABCDEFGH"[names[i]], "ABCDEFGH"[names[j]]), byc[names[i]][0], byc[names[j]][0]))
for what, k0, k1 in pairs:
    line = []
    for piece in (64 << 10, 1 * M, 2 * M, 16 * M, 64 * M, 256 * M, 4 * G):
        line.append("%s %.0f" % ("%dK" % (piece >> 10) if piece < M else "%dM" % (piece >> 20), read2(A + k0 * G, A + k1 * G, 4 * G, piece)))
    print("%-18s slots +%d / +%d GiB: GB/s by piece size: %s" % (what, k0, k1, "  ".join(line)), flush=True)
