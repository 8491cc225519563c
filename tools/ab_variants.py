#!/usr/bin/env python3
"""Interleaved A/B timing of sweep-kernel variants in ONE process on ONE device (GPU box only).
usage: tools/ab_variants.py [--n 256] [--rounds 5] [--op ilu_apply|sgs_apply|sgs_relax|spmv] v1 v2 ..."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blasted_amd import capi, workloads  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--op", default="ilu_apply")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    m = workloads.poisson3d_device(a.n, 4, dev, grid="uniform")
    r = workloads.rhs_vector_device(m["nbrows"] * 4, dev)
    z = torch.zeros_like(r)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    ab = bench.algorithmic_bytes(a.n, 4)
    if a.op == "factor":
        step = lambda: p.ilu0_factorize(3)
        lb, ub = ab["factor_sweep"], 0
    elif a.op == "ilu_apply":
        p.ilu0_factorize(3)
        step = lambda: p.ilu0_apply(r, 3, out=z)
        lb, ub = ab["lower_sweep"], ab["upper_sweep"]
    else:
        p.jacobi_compute()
        if a.op == "sgs_apply":
            step = lambda: p.sgs_apply(r, 3, out=z)
            lb = ub = ab["sgs_pair"] / 2
        elif a.op == "sgs_relax":
            step = lambda: p.sgs_relax(r, z, 3)
            lb = ub = ab["sgs_relax_pass"]
        else:
            step = lambda: p.spmv(r, out=z)
            lb = ub = ab["spmv"]
    res = {v: {"L": [], "U": []} for v in a.variants}
    p.set_timing(True)
    for rd in range(a.rounds + 1):
        for v in a.variants:
            capi.set_tuning(None if v == "default" else v)
            p.get_timing(reset=True)
            for _ in range(a.steps):
                step()
            t = p.get_timing(reset=True)
            if rd == 0:
                continue  # warm-up round
            res[v]["L"].append(t["lower_ms"] / max(t["lower_launches"], 1))
            res[v]["U"].append(t["upper_ms"] / max(t["upper_launches"], 1))
    for v in a.variants:
        L, U = res[v]["L"], res[v]["U"]
        lm, um = statistics.median(L), statistics.median(U)
        print("%-22s L med %.3f min %.3f ms (%.0f GB/s) | U med %.3f min %.3f ms (%.0f GB/s) | pair %.3f ms" % (
            v, lm, min(L), lb / lm / 1e6, um, min(U), (ub / um / 1e6) if um > 0 else 0, lm + um))


if __name__ == "__main__":
    main()
