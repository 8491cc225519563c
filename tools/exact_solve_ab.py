#!/usr/bin/env python3
"""A/B of the exact (level-scheduled) ILU application in one process: tuning strings given on the command line are
applied in turn, interleaved over `rounds` repetitions; per variant the lower / upper pass times from the library's
HIP events, the whole apply, and whether the result is the same bits as the first variant's.
usage: exact_solve_ab.py [n=256] [bs=4] [variant ...]   (default variants: levelpersist=0 levelpersist=1)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    variants = sys.argv[3:] or ["levelpersist=0", "levelpersist=1"]
    dev = torch.device("cuda", 0)
    gen = "poisson"
    if n < 0:
        n, gen = -n, "unstructured"
    m = W.unstructured_bsr(n, bs, device=dev) if gen == "unstructured" else W.poisson3d_device(n, bs, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    z = torch.empty_like(r)
    ref = None
    stats = {v: {"lower": [], "upper": [], "total": []} for v in variants}
    same = {}
    for rnd in range(4):
        for v in variants:
            for spec in v.split("+"):
                capi.set_tuning(spec)
            p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
            torch.cuda.synchronize()
            if ref is None:
                ref = z.clone()
            same[v] = bool(torch.equal(z, ref))
            p.set_timing(True)
            p.get_timing(reset=True)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps * 1e3
            tm = p.get_timing(reset=True)
            p.set_timing(False)
            stats[v]["lower"].append(tm["lower_ms"] / max(tm["lower_launches"], 1))
            stats[v]["upper"].append(tm["upper_ms"] / max(tm["upper_launches"], 1))
            stats[v]["total"].append(dt)
    st = p.level_stats()
    print("%s %d^3 bs=%d: %d rows, %d levels, single-launch passes %d, aborts %d" % (
        gen, n, bs, m["nbrows"], st["levels"], st["syncfree_passes"], st["syncfree_aborts"]))
    for v in variants:
        s = stats[v]
        print("%-40s lower %.3f ms  upper %.3f ms  apply %.3f ms (min %.3f)  same bits as first: %s" % (
            v, min(s["lower"]), min(s["upper"]), sorted(s["total"])[len(s["total"]) // 2], min(s["total"]), same[v]), flush=True)
    p.close()


if __name__ == "__main__":
    main()
