#!/usr/bin/env python3
"""How long do device allocations take (hipMalloc / hipFree, hipMemCreate + hipMemMap + hipMemSetAccess) by size?"""
import ctypes as C
import os
import sys
import time

os.environ["BLASTED_HIP_PROBES"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from blasted_amd import capi  # noqa: E402

torch.zeros(1, device="cuda")
L = capi.lib()
L.blasted_hip_probe_vmm_alloc.argtypes = [C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(C.c_void_p)]
L.blasted_hip_buffer_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_ulong, C.c_int]
L.blasted_hip_buffer_free.argtypes = [C.c_void_p]
G = 1 << 30
for gib in (0.5, 2, 8, 16, 32, 64):
    size = int(gib * G)
    ptr = C.c_void_p(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi._check(L.blasted_hip_buffer_alloc(C.byref(ptr), size, 0))
    t1 = time.perf_counter()
    capi._check(L.blasted_hip_buffer_free(ptr))
    t2 = time.perf_counter()
    print("hipMalloc %5.1f GiB: %.2f ms, hipFree %.2f ms" % (gib, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
for gib, chunk in ((2, 2), (8, 2), (8, 8), (16, 16), (32, 32), (64, 64)):
    ptr = C.c_void_p(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.blasted_hip_probe_vmm_alloc(gib * G, 2 << 20, chunk * G, C.byref(ptr))
    t1 = time.perf_counter()
    print("hipMemCreate/Map/SetAccess + memset %3d GiB in pieces of %2d GiB: rc %d, %.2f ms (never freed)" % (gib, chunk, rc, (t1 - t0) * 1e3), flush=True)
