#!/usr/bin/env python3
"""Does the in-place factorisation sweep (reads the matrix, reads and writes the factor) care about address classes?
One arena, its slots classified; the factor storage placed in a run of slots of each class, the matrix (the caller's)
where it is; three-sweep builds timed.  usage: placement_factor.py [N=256] [ARENA_GIB=160]"""
import ctypes as C
import os
import sys
import time

os.environ["BLASTED_HIP_PROBES"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AG = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
L.blasted_hip_probe_rw.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
M, G = 1 << 20, 1 << 30
capi.set_tuning("placement=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(3)
fbytes = m["nnzb"] * 128
arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()


def rw_ms(rd, rd_bytes, wr, wr_bytes, reps=-3):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_rw(C.c_void_p(rd), rd_bytes, C.c_void_p(wr), wr_bytes, reps, C.byref(out)))
    return out.value


def same_class(piece, ref):
    return rw_ms(piece, 2 * G, ref, 128 * M) > 0.955 * rw_ms(piece, 2 * G, piece + 2 * G - 128 * M, 128 * M)


slots = list(range(0, AG - 1, 2))
cls, reps_of = {}, []
for k in slots:
    for c, kr in enumerate(reps_of):
        if same_class(A + kr * G, A + k * G):
            cls[k] = c
            break
    else:
        cls[k] = len(reps_of)
        reps_of.append(k)
print("classes of the 2 GiB slots: " + "".join("ABCDEFGH"[cls[k]] for k in slots), flush=True)
vals = m["vals"].data_ptr()
vcls = [next((c for c, kr in enumerate(reps_of) if same_class(A + kr * G, vals + off)), -1) for off in (0, 7 * G, 13 * G)]
print("the matrix's value array (15 GB) at its start / middle / end is of class: " + " ".join("ABCDEFGH?"[c] for c in vcls), flush=True)


def timed_build(reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        p.ilu0_factorize(3)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return min(ts)


print("factor storage where the operator allocated it: three-sweep build %.2f ms" % timed_build(), flush=True)
need = int(np.ceil(fbytes / (2 * G)))
for c in range(len(reps_of)):
    run, start = 0, None
    for k in slots:
        run = run + 1 if cls[k] == c else 0
        if run >= need:
            start = k - 2 * (need - 1)
            break
    if start is None:
        print("class %s: no run of %d GiB" % ("ABCDEFGH"[c], 2 * need))
        continue
    capi._check(L.blasted_hip_probe_place(p._h, b"iluvals", C.c_void_p(A + start * G)))
    print("factor storage in class %s (+%d GiB): three-sweep build %.2f ms" % ("ABCDEFGH"[c], start, timed_build()), flush=True)
capi._check(L.blasted_hip_probe_place(p._h, b"iluvals", C.c_void_p(0)))
