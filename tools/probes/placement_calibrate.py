#!/usr/bin/env python3
"""Calibration of the address-class probe as the product uses it (every launch reads about 4 GiB: a small piece several
times): time ratio (writes in the reference) / (writes inside the piece) for pieces of 256 MiB ... 2 GiB whose class
relation to the reference is KNOWN (arena slots classified with 2 GiB pieces first), ten repetitions each.
usage: placement_calibrate.py [ARENA_GIB=200]"""
import ctypes as C
import os
import sys

os.environ["BLASTED_HIP_PROBES"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi  # noqa: E402

AG = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_rw.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
M, G = 1 << 20, 1 << 30
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()


def rw_ms(rd, rd_bytes, wr, wr_bytes, reps):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_rw(C.c_void_p(rd), rd_bytes, C.c_void_p(wr), wr_bytes, reps, C.byref(out)))
    return out.value


slots = list(range(0, AG - 1, 2))
cls, reps_of = {}, []
for k in slots:
    base = A + k * G
    for c, kr in enumerate(reps_of):
        ref = A + kr * G
        if rw_ms(ref, 2 * G, base, 128 * M, 4) > 0.955 * rw_ms(ref, 2 * G, ref + 2 * G - 128 * M, 128 * M, 4):
            cls[k] = c
            break
    else:
        cls[k] = len(reps_of)
        reps_of.append(k)
print("classes of the 2 GiB slots: " + "".join("ABCDEFGH"[cls[k]] for k in slots), flush=True)
ka = slots[0]
same = [k for k in slots[4:] if cls[k] == cls[ka]][:3]
other = [k for k in slots if cls[k] != cls[ka]][:3]
for size in (256 * M, 512 * M, 1 * G, 2 * G):
    wr = size // 16
    for what, ks in (("same class", same), ("another class", other)):
        ratios = []
        for k in ks:
            piece = A + k * G
            for _ in range(4):
                t_self = rw_ms(piece, size, piece + size - wr, wr, -3)
                t_ref = rw_ms(piece, size, A + ka * G, wr, -3)
                ratios.append(t_ref / t_self)
        r = np.array(ratios)
        print("piece %4d MiB, reference in %-13s: ratio min %.3f median %.3f max %.3f (%d samples; one launch %.3f ms)" % (
            size // M, what, r.min(), np.median(r), r.max(), r.size, t_self), flush=True)
