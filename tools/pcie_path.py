#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-vector path (what the PCSHELL boundary pays: r H2D, z D2H from pageable
host memory) beside the device-resident path, at BASELINE's headline size.  usage: pcie_path.py [n] [bs]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(n, bs, dev, grid="uniform")
    nv = m["nbrows"] * bs
    rd = W.rhs_vector_device(nv, dev)
    zd = torch.zeros_like(rd)
    rh = rd.cpu().numpy().copy()
    zh = np.zeros_like(rh)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(3)
    s = 3

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    td = timed(lambda: p.ilu0_apply(rd, s, out=zd))
    th = timed(lambda: p.ilu0_apply(rh, s, out=zh))
    # same operator through both paths (synchronous sweeps are deterministic; the async ones are not)
    zs_h = p.ilu0_apply(rh, s, mode=capi.JACOBI_SYNC)
    zs_d = p.ilu0_apply(rd, s, mode=capi.JACOBI_SYNC)
    assert np.array_equal(zs_h, zs_d.cpu().numpy())
    # the same with the two host vectors page-locked in place (blasted_hip_host_register: the caller's choice)
    capi.host_register(rh)
    capi.host_register(zh)
    tp = timed(lambda: p.ilu0_apply(rh, s, out=zh))
    capi.host_unregister(rh)
    capi.host_unregister(zh)
    # set_values from host memory: a 4 GB slice of the value array, pageable and registered
    nslice = min(m["vals"].numel(), 1 << 29)
    vh = np.ones(nslice)
    vd = torch.empty(nslice, dtype=torch.float64, device=dev)

    import ctypes as C
    capi.lib().blasted_hip_buffer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong]

    def upload():
        capi._check(capi.lib().blasted_hip_buffer_upload(vd.data_ptr(), vh.ctypes.data, vh.nbytes))
    tu = timed(upload, reps=3)
    capi.host_register(vh)
    tur = timed(upload, reps=3)
    capi.host_unregister(vh)
    mb = 2 * nv * 8 / 1e6
    print("host-vector apply with r and z page-locked (blasted_hip_host_register): %.2f ms = %.1f sweep pairs/s, "
          "transfers %.2f ms = %.1f GB/s; value upload of %.1f GB: pageable %.1f GB/s, page-locked %.1f GB/s" % (
              tp * 1e3, s / tp, (tp - td) * 1e3, mb / 1e3 / (tp - td), vh.nbytes / 1e9, vh.nbytes / 1e9 / tu,
              vh.nbytes / 1e9 / tur))
    print("n=%d bs=%d napplysweeps=%d: device-resident apply %.2f ms = %.1f sweep pairs/s; host-vector apply "
          "%.2f ms = %.1f sweep pairs/s (PCIe-inclusive; %.0f MB moved, transfers %.2f ms = %.1f GB/s)" % (
              n, bs, s, td * 1e3, s / td, th * 1e3, s / th, mb, (th - td) * 1e3, mb / 1e3 / (th - td)))
    p.close()


if __name__ == "__main__":
    main()
