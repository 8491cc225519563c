#!/usr/bin/env python3
"""Soak test of the single-launch (dependency-polling) exact passes: many repetitions at several sizes
and block sizes; every repetition must reproduce the first result BIT FOR BIT (the exact pass is
deterministic: each row is one fixed expression of final inputs) and no pass may have given up waiting.
The single-launch exact FACTORISATIONS (kernels_factor.hip / kernels_factor4.hip) are soaked the same way, with
the exact apply of the fresh factor as the checksum.
usage: soak_exact.py [reps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    cases = [(256, 4, reps), (96, 4, 4 * reps), (100, 8, 2 * reps), (-126, 5, reps), (128, 7, reps),
             (160, 1, reps), (64, 3, 2 * reps), (33, 4, 8 * reps)]
    for n, bs, k in cases:
        m = W.poisson3d_device(n, bs, dev, grid="uniform") if n > 0 else W.unstructured_bsr(-n, bs, device=dev)
        r = W.rhs_vector_device(m["nbrows"] * bs, dev)
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
        p.set_matrix(m)
        p.ilu0_factorize(2)
        p.jacobi_compute()
        z0 = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
        s0 = p.sgs_apply(r, 1, mode=capi.LEVEL).clone()
        z = torch.empty_like(r)
        t0 = time.perf_counter()
        bad = 0
        for i in range(k):
            p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
            if not torch.equal(z, z0):
                bad += 1
            if i % 4 == 0:
                p.sgs_apply(r, 1, mode=capi.LEVEL, out=z)
                if not torch.equal(z, s0):
                    bad += 1
        torch.cuda.synchronize()
        st = p.level_stats()
        print("n=%d bs=%d rows=%d: %d exact applies in %.1f s, mismatches %d, %s" % (
            n, bs, m["nbrows"], k, time.perf_counter() - t0, bad, st), flush=True)
        assert bad == 0 and st["syncfree_aborts"] == 0
        # exact factorisations: the exact apply of each fresh factor must be the same bits
        p.ilu0_factorize(-1)
        f0 = p.ilu0_apply(r, 1, mode=capi.LEVEL).clone()
        kf = max(1, k // 4)
        t0 = time.perf_counter()
        for i in range(kf):
            p.ilu0_factorize(-1)
            p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
            if not torch.equal(z, f0):
                bad += 1
        torch.cuda.synchronize()
        st2 = p.level_stats()
        print("n=%d bs=%d: %d exact factorisations in %.1f s, mismatches %d, single-launch passes %d, %s" % (
            n, bs, kf, time.perf_counter() - t0, bad, st2["syncfree_passes"] - st["syncfree_passes"], st2), flush=True)
        assert bad == 0 and st2["syncfree_aborts"] == 0
        p.close()
        del m, r, z, z0, s0
        torch.cuda.empty_cache()
    print("soak ok")


if __name__ == "__main__":
    main()
