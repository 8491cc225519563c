#!/usr/bin/env python3
"""Diagonal-block inversion (Jacobi compute) and a factorisation sweep with the two inversion kernels of 5 <= bs <= 8:
time, and the inverse against numpy.  usage: invert_ab.py [config=5]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
import bench  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg = bench.CONFIGS[k]
    dev = torch.device("cuda", 0)
    m = W.unstructured_bsr(cfg["n"], cfg["bs"], device=dev) if cfg["gen"] == "unstructured" else \
        W.poisson3d_device(cfg["n"], cfg["bs"], dev, grid=cfg["grid"])
    bs = cfg["bs"]
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    dg = m["diagind"].long()
    blocks = m["vals"].reshape(-1, bs, bs)[dg].transpose(1, 2)   # column-major storage -> [r][c]
    ref = torch.linalg.inv(blocks)
    for spec in ("invertrow=0", "invertrow=1"):
        capi.set_tuning(spec)
        p.jacobi_compute()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            p.jacobi_compute()
        torch.cuda.synchronize()
        tj = (time.perf_counter() - t0) / 10 * 1e3
        d = torch.from_numpy(p.get_dblocks()).reshape(-1, bs, bs).transpose(1, 2).to(dev)
        err = float((d - ref).abs().max() / ref.abs().max())
        p.ilu0_factorize(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            p.ilu0_factorize(3)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 3 * 1e3
        print("config %d bs=%d %-12s jacobi_compute %.3f ms, max rel err of the inverses vs torch %.2e, 3-sweep factorisation %.2f ms" % (
            k, bs, spec, tj, err, tf), flush=True)
    capi.set_tuning("invertrow=1")
    p.close()


if __name__ == "__main__":
    main()
