#!/usr/bin/env python3
"""Interleaved A/B timing of tuning strings on one of bench.py's configs, in ONE process on ONE device (GPU box
only): process-to-process placement differences (up to 10 %) drop out.  A variant is a ';'-separated list of
blasted_hip_set_tuning strings, applied in full before each of its turns (so give every variant a value for
every knob that any variant changes).
usage: tools/ab_config.py [--config 4] [--rounds 5] [--steps 4] [--op ilu_apply|factor|sgs_relax|spmv] v1 v2 ..."""
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
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--bs", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--op", default=None)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    cfg = dict(bench.CONFIGS[a.config])
    if a.n:
        cfg["n"] = a.n
    if a.bs:
        cfg["bs"] = a.bs
    op = a.op or cfg["op"]
    n, bs, s = cfg["n"], cfg["bs"], cfg["sweeps"]
    dev = torch.device("cuda", 0)
    if cfg["gen"] == "unstructured":
        m = workloads.unstructured_bsr(n, bs, device=dev)
    else:
        m = workloads.poisson3d_device(n, bs, dev, grid=cfg["grid"])
    r = workloads.rhs_vector_device(m["nbrows"] * bs, dev)
    z = torch.zeros_like(r)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    nb, nnzb, nnzl, nnzu = bench.matrix_counts(m)
    npairs = nnzl
    if op in ("ilu_apply", "factor"):
        p.ilu0_factorize(cfg["build"])
        npairs = p.ilu0_positions_size()
    else:
        p.jacobi_compute()
    ab = bench.pattern_bytes(nb, nnzb, nnzl, nnzu, npairs, bs)
    step = {"ilu_apply": lambda: p.ilu0_apply(r, s, out=z), "factor": lambda: p.ilu0_factorize(cfg["build"]),
            "sgs_apply": lambda: p.sgs_apply(r, s, out=z), "sgs_relax": lambda: p.sgs_relax(r, z, s),
            "spmv": lambda: p.spmv(r, out=z)}[op]
    lb, ub = {"ilu_apply": (ab["lower_sweep"], ab["upper_sweep"]), "factor": (ab["factor_sweep"], 0),
              "sgs_apply": (ab["sgs_pair"] - ab["sgs_bwd"], ab["sgs_bwd"]),
              "sgs_relax": (ab["sgs_relax_pass"], ab["sgs_relax_pass"]), "spmv": (ab["spmv"], 0)}[op]
    res = {v: {"L": [], "U": []} for v in a.variants}
    p.set_timing(True)
    for rd in range(a.rounds + 1):
        for v in a.variants:
            for spec in v.split(";"):
                capi.set_tuning(None if spec == "default" else spec)
            p.get_timing(reset=True)
            for _ in range(a.steps):
                step()
            t = p.get_timing(reset=True)
            if rd == 0:
                continue  # warm-up round
            res[v]["L"].append(t["lower_ms"] / max(t["lower_launches"], 1))
            res[v]["U"].append(t["upper_ms"] / max(t["upper_launches"], 1))
    print("config %d (%s), op %s: %d block-rows, bs %d" % (a.config, cfg["workload"], op, nb, bs))
    for v in a.variants:
        L, U = res[v]["L"], res[v]["U"]
        lm, um = statistics.median(L), statistics.median(U)
        print("%-44s L med %.4f min %.4f ms (%.0f GB/s) | U med %.4f min %.4f ms (%.0f GB/s) | sum %.4f ms" % (
            v, lm, min(L), lb / lm / 1e6, um, min(U), (ub / um / 1e6) if um > 0 else 0, lm + um), flush=True)
    p.close()


if __name__ == "__main__":
    main()
