#!/usr/bin/env python3
"""Exact (level-scheduled) ILU(0) factorisation: one launch per level ("factorsf=0") against one launch with
overlapping levels -- the general kernel ("factorsf=3") and, where it applies (bs = 4 stencils), the matrix-core
kernel that prepares a row before it waits ("factorsf=2") -- on bench.py's configs.  usage: exact_factor_time.py [config | poisson:N:BS ...]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    # an argument is a bench config number, or "poisson:N:BS"
    for arg in sys.argv[1:] or ["2"]:
        if arg.startswith("poisson:"):
            _, n_, bs_ = arg.split(":")
            c, cfg = 0, dict(bench.CONFIGS[2], n=int(n_), bs=int(bs_), workload="poisson3d_%s_bs%s" % (n_, bs_))
        else:
            c, cfg = int(arg), bench.CONFIGS[int(arg)]
        n, bs = cfg["n"], cfg["bs"]
        m = W.unstructured_bsr(n, bs, device=dev) if cfg["gen"] == "unstructured" else W.poisson3d_device(n, bs, dev, grid=cfg["grid"])
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        res = {}
        for mode in ("0", "3", "2"):
            capi.set_tuning("factorsf=" + mode)
            p.ilu0_factorize(-1)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                p.ilu0_factorize(-1)
            torch.cuda.synchronize()
            res[mode] = ((time.perf_counter() - t) / 3 * 1e3, p.get_iluvals() if m["vals"].numel() < (1 << 28) else None)
        st = p.level_stats()
        same = None
        if res["0"][1] is not None:
            import numpy as np
            same = float(np.abs(res["0"][1] - res["2"][1]).max() / np.abs(res["0"][1]).max())
        print("config %d (%s): %d levels; exact factorisation %.2f ms per-level launches, %.2f ms as one launch "
              "(general kernel), %.2f ms as one launch (factorsf=2); aborts %d; max rel difference %s" % (
                  c, cfg["workload"], st["levels"], res["0"][0], res["3"][0], res["2"][0], st["syncfree_aborts"], same),
              flush=True)
        capi.set_tuning("factorsf=1")
        p.close()
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
