#!/usr/bin/env python3
"""placement_slots.py: a result vector / triangle copy placed ANYWHERE in one 96 GiB allocation runs the sweeps in
their fast mode.  What about that allocation does it -- its size, the alignment of its address, how it is backed?
(1) plain hipMalloc blocks of 0.5 ... 64 GiB, z at their start and at their end; the one-lane page-walk meter on
each; (2) hand-built ranges (hipMemAddressReserve / hipMemCreate / hipMemMap) with chosen address alignment and
physical chunk size.
usage: placement_arena_size.py [N=256]"""
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
L = capi.lib()
L.blasted_hip_probe_vmm_alloc.argtypes = [C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(C.c_void_p)]
L.blasted_hip_probe_page_walk.argtypes = [C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(C.c_double)]
L.blasted_hip_probe_place.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
L.blasted_hip_buffer_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_ulong, C.c_int]
L.blasted_hip_buffer_free.argtypes = [C.c_void_p]
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
walk_no = [0]


def walk(ptr, nb_, stride):
    out = C.c_double(0)
    walk_no[0] += 1
    start = (walk_no[0] * 4160) % (1 << 16)
    capi._check(L.blasted_hip_probe_page_walk(C.c_void_p(ptr), nb_, stride, start - start % 8, C.byref(out)))
    return out.value


def as_tensor(ptr, nbytes_):
    """a float64 torch view of raw device memory (no ownership)"""
    class Holder:
        pass
    h = Holder()
    h.__cuda_array_interface__ = {"shape": (nbytes_ // 8,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=dev)


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


def align_of(ptr):
    a = 0
    while ptr and not (ptr >> a) & 1:
        a += 1
    return a


z0 = torch.zeros(n, dtype=torch.float64, device=dev)
print("baseline (own buffers, z0 at %#x, address aligned to 2^%d): lower %.3f upper %.3f" % (
    (z0.data_ptr(), align_of(z0.data_ptr())) + measure(r0, z0)), flush=True)
print("## plain hipMalloc blocks: z at the start / at the end of each", flush=True)
for gib in (0.5, 1, 2, 4, 8, 16, 32, 64):
    size = int(gib * G)
    ptr = C.c_void_p(0)
    capi._check(L.blasted_hip_buffer_alloc(C.byref(ptr), size, 0))
    base = ptr.value
    torch.cuda.synchronize()
    a = measure(r0, as_tensor(base, nbytes))
    b = measure(r0, as_tensor(base + size - nbytes, nbytes))
    print("hipMalloc %5.1f GiB at %#x (2^%d): z at start upper %.3f, at end %.3f | walk ns/load @2M %.0f @64K %.0f" % (
        gib, base, align_of(base), a[1], b[1], walk(base, size, 2 * M), walk(base, min(size, 512 * M), 64 << 10)), flush=True)
    capi._check(L.blasted_hip_buffer_free(ptr))
print("## hand-built ranges of 1 GiB: address alignment x physical chunk size; z at the start", flush=True)
for va_align in (2 * M, 64 * M, 1 * G, 2 * G):
    for chunk in (2 * M, 64 * M, 512 * M, 1 * G):
        ptr = C.c_void_p(0)
        rc = L.blasted_hip_probe_vmm_alloc(1 * G, va_align, chunk, C.byref(ptr))
        if rc != 0:
            print("vmm align %d MiB chunk %d MiB: failed: %s" % (va_align // M, chunk // M, L.blasted_hip_last_error().decode()), flush=True)
            continue
        a = measure(r0, as_tensor(ptr.value, nbytes))
        print("vmm 1 GiB at %#x (2^%d), asked alignment %4d MiB, chunks of %4d MiB: upper %.3f lower %.3f | walk @2M %.0f" % (
            ptr.value, align_of(ptr.value), va_align // M, chunk // M, a[1], a[0], walk(ptr.value, 1 * G, 2 * M)), flush=True)
print("## the upper copy in hand-built ranges (z = z0)", flush=True)
usz = (usize + 2 * M - 1) // (2 * M) * (2 * M)
for va_align, chunk in ((2 * M, 2 * M), (2 * M, 1 * G), (1 * G, 1 * G), (2 * G, 2 * G)):
    ptr = C.c_void_p(0)
    size = (usz + chunk - 1) // chunk * chunk
    rc = L.blasted_hip_probe_vmm_alloc(size, va_align, chunk, C.byref(ptr))
    if rc != 0:
        print("vmm ucopy align %d chunk %d failed: %s" % (va_align // M, chunk // M, L.blasted_hip_last_error().decode()), flush=True)
        continue
    capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", ptr))
    a = measure(r0, z0)
    print("ucopy in vmm range at %#x, alignment %4d MiB, chunks of %4d MiB: upper %.3f" % (ptr.value, va_align // M, chunk // M, a[1]), flush=True)
capi._check(L.blasted_hip_probe_place(p._h, b"ucopy", C.c_void_p(0)))
p.close()
