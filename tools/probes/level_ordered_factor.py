#!/usr/bin/env python3
"""How much would the exact (level-scheduled) factorisation gain from level-ordered storage?  Measured without
writing the kernel: the matrix is symmetrically permuted into its own level order ON THE DEVICE (P A P^T, rows
sorted by dependency level, columns renumbered and re-sorted), so that for the existing exact factorisation
natural order IS level order -- every level's rows, their A blocks, their factor blocks and the rows they
gather from are contiguous.  Same arithmetic, same kernels, only the storage order differs.
usage: level_ordered_factor.py [n=256] [bs=4]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def permute_symmetric_device(m, rows):
    """P A P^T on the device: new block-row k is old block-row rows[k]."""
    dev = m["vals"].device
    nb, bs = m["nbrows"], m["bs"]
    bs2 = bs * bs
    rows = rows.to(torch.int64)
    inv = torch.empty(nb, dtype=torch.int64, device=dev)
    inv[rows] = torch.arange(nb, dtype=torch.int64, device=dev)
    rp = m["browptr"].to(torch.int64)
    cnt = (rp[1:] - rp[:-1])[rows]
    nrp = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    nrp[1:] = torch.cumsum(cnt, 0)
    nnzb = int(nrp[-1])
    newrow = torch.repeat_interleave(torch.arange(nb, dtype=torch.int64, device=dev), cnt)
    src = rp[rows][newrow] + (torch.arange(nnzb, dtype=torch.int64, device=dev) - nrp[newrow])
    newcol = inv[m["bcolind"].to(torch.int64)[src]]
    order = torch.argsort(newrow * nb + newcol)
    src, newrow, newcol = src[order], newrow[order], newcol[order]
    vals = torch.empty(nnzb * bs2, dtype=torch.float64, device=dev)
    v2 = m["vals"].view(-1, bs2)
    chunk = 1 << 22
    for s0 in range(0, nnzb, chunk):
        vals[s0 * bs2:min(nnzb, s0 + chunk) * bs2] = v2[src[s0:s0 + chunk]].reshape(-1)
    diagind = torch.nonzero(newrow == newcol).flatten()
    assert diagind.numel() == nb
    return {"nbrows": nb, "nnzb": nnzb, "bs": bs, "rowmajor": False, "browptr": nrp.to(torch.int32),
            "bcolind": newcol.to(torch.int32), "diagind": diagind.to(torch.int32), "vals": vals}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(n, bs, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    lv, rows, ptr = p.get_levels()
    t_nat = timed(lambda: p.ilu0_factorize(-1))
    t_nat_apply = timed(lambda: p.ilu0_apply(r, 1, mode=capi.LEVEL))
    t_async = timed(lambda: p.ilu0_factorize(3))
    z_nat = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
    p.ilu0_factorize(-1)
    z_nat = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
    p.close()
    rows_t = torch.from_numpy(rows).to(dev)
    mp = permute_symmetric_device(m, rows_t)
    del m
    torch.cuda.empty_cache()
    q = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    q.set_matrix(mp)
    rp_ = r.view(-1, bs)[rows_t.to(torch.int64)].reshape(-1).contiguous()
    t_lvl = timed(lambda: q.ilu0_factorize(-1))
    t_lvl_apply = timed(lambda: q.ilu0_apply(rp_, 1, mode=capi.LEVEL))
    t_lvl_async = timed(lambda: q.ilu0_factorize(3))
    q.ilu0_factorize(-1)
    z_lvl = q.ilu0_apply(rp_, 1, mode=capi.LEVEL)
    back = torch.empty_like(z_lvl)
    back.view(-1, bs)[rows_t.to(torch.int64)] = z_lvl.view(-1, bs)
    err = float((back - z_nat).abs().max() / z_nat.abs().max())
    print("%d^3 bs=%d, %d levels: exact factorisation %.2f ms in natural order, %.2f ms on the level-ordered matrix; "
          "exact apply %.2f / %.2f ms; three asynchronous factor sweeps %.2f / %.2f ms; results agree to %.1e" % (
              n, bs, len(ptr) - 1, t_nat, t_lvl, t_nat_apply, t_lvl_apply, t_async, t_lvl_async, err))
    q.close()


if __name__ == "__main__":
    main()
