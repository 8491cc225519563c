#!/usr/bin/env python3
"""Per-block-size throughput of the row-sweep kernels (GPU box only): Poisson n^3 pattern inflated to
bs, algorithmic GB/s of lower sweep / upper sweep / SpMV / factor sweep."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blasted_amd import capi, workloads  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--bs", type=int, nargs="+", default=[1, 2, 3, 4, 5, 7, 8])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--lib", default=None, help="another build of libblasted_hip.so (A/B of two commits)")
    a = ap.parse_args()
    if a.lib:
        capi.LIBPATH = os.path.abspath(a.lib)
    dev = torch.device("cuda", 0)
    for bs in a.bs:
        n = a.n
        m = workloads.poisson3d_device(n, bs, dev, grid="uniform")
        r = workloads.rhs_vector_device(m["nbrows"] * bs, dev)
        z = torch.zeros_like(r)
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        ab = bench.algorithmic_bytes(n, bs)
        p.set_timing(True)
        p.ilu0_factorize(2)
        p.get_timing(reset=True)
        p.ilu0_factorize(3)
        t = p.get_timing(reset=True)
        fac = t["lower_ms"] / max(t["lower_launches"], 1)
        for _ in range(2):
            p.ilu0_apply(r, 3, out=z)
        p.get_timing(reset=True)
        for _ in range(a.steps):
            p.ilu0_apply(r, 3, out=z)
        t = p.get_timing(reset=True)
        lo = t["lower_ms"] / t["lower_launches"]
        up = t["upper_ms"] / t["upper_launches"]
        for _ in range(2):
            p.spmv(r, out=z)
        p.get_timing(reset=True)
        for _ in range(a.steps):
            p.spmv(r, out=z)
        t = p.get_timing(reset=True)
        sp = t["lower_ms"] / t["lower_launches"]
        print("bs=%d n=%d rows=%d | lower %.3f ms %.0f GB/s | upper %.3f ms %.0f GB/s | spmv %.3f ms %.0f GB/s | factor %.3f ms %.0f GB/s" % (
            bs, n, ab["nbrows"], lo, ab["lower_sweep"] / lo / 1e6, up, ab["upper_sweep"] / up / 1e6,
            sp, ab["spmv"] / sp / 1e6, fac, ab["factor_sweep"] / fac / 1e6), flush=True)
        p.close()
        del m, r, z
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
