#!/usr/bin/env python3
"""Does the sweeps' fast / slow mode follow how the driver MAPPED a buffer (size of the contiguous fragments behind
it)?  (placement_streams.py showed that re-allocating the triangle copies at the SAME virtual
address changes the mode, i.e. the physical backing decides.)  A one-lane walk with one dependent load per 2 MiB /
64 KiB pays a page-table walk per load where translations are small; its ns per load is printed beside the sweep
time for: an allocation made FIRST in the process, the operator's triangle copies re-allocated a dozen times, fresh
result vectors, result vectors carved from the early allocation.
usage: placement_frag.py [N=256] [REALLOCS=10]"""
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
REALLOCS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
L = capi.lib()
L.blasted_hip_probe_move.argtypes = [C.c_void_p, C.c_char_p]
L.blasted_hip_probe_addresses.argtypes = [C.c_void_p, C.c_void_p]
L.blasted_hip_probe_page_walk.argtypes = [C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(C.c_double)]
M = 1 << 20
walk_no = [0]


def walk(ptr, nbytes, stride):
    out = C.c_double(0)
    walk_no[0] += 1
    start = (walk_no[0] * 4160) % min(stride, 1 << 16)  # fresh lines every time
    capi._check(L.blasted_hip_probe_page_walk(C.c_void_p(ptr), nbytes, stride, start - start % 8, C.byref(out)))
    return out.value


def walks(ptr, nbytes):
    return "walk ns/load @2M %.0f @64K %.0f @4K %.0f" % (walk(ptr, nbytes, 2 * M), walk(ptr, min(nbytes, 512 * M), 64 << 10),
                                                           walk(ptr, min(nbytes, 32 * M), 4096))


n = N ** 3 * 4
nbytes = n * 8
early = torch.zeros(4 * nbytes + 64 * M, dtype=torch.uint8, device=dev)  # the first device allocation of the process
print("early allocation at %#x: %s" % (early.data_ptr(), walks(early.data_ptr(), early.numel())), flush=True)
capi.set_tuning("compactafter=0")
m = W.poisson3d_device(N, 4, dev, grid="uniform")
r0 = W.rhs_vector_device(n, dev)
p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
p.set_matrix(m)
p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
p.set_timing(True)


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


z0 = torch.zeros(n, dtype=torch.float64, device=dev)
lo, up = measure(r0, z0)
a = addresses()
nb, nnzb = m["nbrows"], m["nnzb"]
usize = ((nnzb - nb) // 2 + nb) * 128
lsize = ((nnzb - nb) // 2) * 128
print("baseline lower %.3f upper %.3f | z %#x %s | ytemp %#x %s" % (lo, up, z0.data_ptr(), walks(z0.data_ptr(), nbytes),
                                                                 a["ytemp"], walks(a["ytemp"], nbytes)), flush=True)
print("   ucopy %#x %s | lcopy %#x %s" % (a["ucopy"], walks(a["ucopy"], usize), a["lcopy"], walks(a["lcopy"], lsize)), flush=True)
for what, size in (("ucopy", usize), ("lcopy", lsize)):
    print("## %s freed and allocated again %d times (same size, everything else fixed)" % (what, REALLOCS), flush=True)
    for k in range(REALLOCS):
        capi._check(L.blasted_hip_probe_move(p._h, what.encode()))
        lo, up = measure(r0, z0)
        a = addresses()
        print("%s #%-2d at %#x lower %.3f upper %.3f | %s" % (what, k, a[what], lo, up, walks(a[what], size)), flush=True)
print("## fresh result vectors (torch allocations made now)", flush=True)
zs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(6)]
for i, z in enumerate(zs):
    lo, up = measure(r0, z)
    print("z#%d at %#x lower %.3f upper %.3f | %s" % (i, z.data_ptr(), lo, up, walks(z.data_ptr(), nbytes)), flush=True)
print("## result vectors carved from the EARLY allocation", flush=True)
for i in range(4):
    z = early[i * nbytes + (i * 2 * M):][:nbytes].view(torch.float64)
    lo, up = measure(r0, z)
    print("early z#%d at %#x lower %.3f upper %.3f | %s" % (i, z.data_ptr(), lo, up, walks(z.data_ptr(), nbytes)), flush=True)
print("## right-hand sides carved from the early allocation (z0 as the result)", flush=True)
for i in range(2):
    r = early[i * nbytes:][:nbytes].view(torch.float64)
    r.copy_(r0)
    lo, up = measure(r, z0)
    print("early r#%d lower %.3f upper %.3f" % (i, lo, up), flush=True)
p.close()
