#!/usr/bin/env python3
"""Beyond the headline size: 3-D Poisson n^3 (default 320: 32.8 M block-rows, 29 GB of factor, element
offsets above 2^31) -- SpMV against torch, asynchronous and exact ILU(0) factor + apply consistency,
throughput.  usage: bigsize_check.py [n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tests")
from blasted_amd import capi, workloads as W  # noqa: E402
from test_gpu_fullsize import torch_part_matvec, relmax  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(n, 4, dev, grid="uniform")
    nb = m["nbrows"]
    print("n=%d: %d block-rows, %d blocks, %.1f GB of values, %d value entries (2^31 = %d)" % (
        n, nb, m["nnzb"], m["nnzb"] * 128 / 1e9, m["nnzb"] * 16, 2 ** 31), flush=True)
    r = W.rhs_vector_device(nb * 4, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    y = p.spmv(r)
    assert relmax(y, torch_part_matvec(m, m["vals"], r, "all")) < 1e-13
    info = p.ilu0_factorize(3, compute_info=True)
    assert np.isfinite(info).all() and info[0] < 1e-3 * info[1]
    z = torch.zeros_like(r)
    for _ in range(2):
        p.ilu0_apply(r, 3, out=z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        p.ilu0_apply(r, 3, out=z)
    torch.cuda.synchronize()
    ta = (time.perf_counter() - t0) / 5
    nnzl = 3 * n ** 3 - 3 * n ** 2
    pair = (2 * nnzl + nb) * 128 + 2 * nnzl * 4 + 4 * nb * 4 + 6 * nb * 32
    print("async apply 3+3 sweeps: %.2f ms = %.0f sweep pairs/s = %.2f TB/s algorithmic" % (
        ta * 1e3, 3 / ta, 3 * pair / ta / 1e12), flush=True)
    info = p.ilu0_factorize(-1, compute_info=True)
    assert info[0] < 2e-15 * info[1]
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    F = torch.from_numpy(p.get_iluvals()).to(dev)
    yt = torch.from_numpy(p.get_ytemp()).to(dev)
    assert relmax(yt, r - torch_part_matvec(m, F, yt, "lower")) < 1e-12
    dinv = F.view(-1, 4, 4)[m["diagind"].to(torch.int64)].transpose(1, 2)
    t = yt - torch_part_matvec(m, F, ze, "upper")
    assert relmax(ze, torch.einsum("brc,bc->br", dinv, t.view(nb, 4)).reshape(-1)) < 1e-12
    t0 = time.perf_counter()
    for _ in range(5):
        p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
    torch.cuda.synchronize()
    print("exact apply: %.2f ms, %s" % ((time.perf_counter() - t0) / 5 * 1e3, p.level_stats()))
    zl = p.ilu0_apply(r, 80, mode=capi.ASYNC)
    print("async 80 sweeps vs exact: %.2e" % relmax(zl, ze))
    print("bigsize ok")


if __name__ == "__main__":
    main()
