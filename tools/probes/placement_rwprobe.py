#!/usr/bin/env python3
"""Can a small kernel tell whether two buffers are in the same "class" (placement_map.py: a sweep is slow when the
vector it WRITES and the triangle copy it STREAMS lie in the same class of address ranges)?  One arena; a read stream
over a 2 GiB piece at +RG GiB beside rewrites of a 512 MiB piece that moves through the arena in 2 GiB steps; next to
it the real upper sweep with the copy at +RG and z at the same positions.
usage: placement_rwprobe.py [N=256] [ARENA_GIB=200] [RG,RG,..=0,100]"""
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
RGS = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,100").split(",")]
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
L.blasted_hip_probe_rw.argtypes = [C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_int, C.POINTER(C.c_double)]
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


def sweep_ms(z, reps=3):
    for _ in range(2):
        p.ilu0_apply(r0, 3, out=z)
    p.synchronize()
    p.get_timing()
    up = []
    for _ in range(reps):
        p.ilu0_apply(r0, 3, out=z)
        p.synchronize()
        t = p.get_timing()
        up.append(t["upper_ms"] / t["upper_launches"])
    return float(np.median(up))


def rw_ms(rd, rd_bytes, wr, wr_bytes, reps=5):
    out = C.c_double(0)
    capi._check(L.blasted_hip_probe_rw(C.c_void_p(rd), rd_bytes, C.c_void_p(wr), wr_bytes, reps, C.byref(out)))
    return out.value


arena = torch.zeros(AG * G, dtype=torch.uint8, device=dev)
A = arena.data_ptr()
print("arena %d GiB at %#x" % (AG, A), flush=True)
for rg in RGS:
    capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", C.c_void_p(A + rg * G)))
    self_ms = rw_ms(A + rg * G, 2 * G, A + rg * G + 2 * G - 128 * M, 128 * M)
    print("## read piece / upper copy at +%d GiB; probe with the rewrites INSIDE the read piece: %.3f ms" % (rg, self_ms), flush=True)
    for k in range(0, AG - 1, 2):
        off = k * G
        if off + nbytes > rg * G and off < rg * G + usize:
            continue
        z = arena[off:off + nbytes].view(torch.float64)
        print("  +%3d GiB: probe 2 GiB / 128 MiB %.3f ms, 512 MiB / 32 MiB %.4f ms | upper sweep %.3f ms" % (
            k, rw_ms(A + rg * G, 2 * G, A + off, 128 * M), rw_ms(A + rg * G, 512 * M, A + off, 32 * M, reps=10), sweep_ms(z)), flush=True)
capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", C.c_void_p(0)))
p.close()
