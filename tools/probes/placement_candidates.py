#!/usr/bin/env python3
"""Is the fast / slow mode of the upper sweep a stable property of the vector it iterates on?  One operator, K
result vectors allocated side by side; the upper sweep is timed on each, twice, in two orders.  If a vector keeps
its mode, a library-owned iterate could be picked among a few candidates at set-up.
usage: placement_candidates.py [K=10]"""
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blasted_amd import capi, workloads as W  # noqa: E402

dev = torch.device("cuda:0")


def measure(p, r, z, n=6):
    for _ in range(2):
        p.ilu0_apply(r, 3, out=z)
    p.synchronize()
    p.get_timing()
    lo, up = [], []
    for _ in range(n):
        p.ilu0_apply(r, 3, out=z)
        p.synchronize()
        t = p.get_timing()
        lo.append(t["lower_ms"] / t["lower_launches"])
        up.append(t["upper_ms"] / t["upper_launches"])
    return float(np.median(lo)), float(np.median(up))


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    m = W.poisson3d_device(256, 4, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * 4, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    p.set_timing(True)
    zs = [torch.zeros_like(r) for _ in range(K)]
    first = [measure(p, r, z) for z in zs]
    second = [measure(p, r, zs[i]) for i in reversed(range(K))][::-1]
    for i in range(K):
        print("candidate %2d at 0x%x: upper %.3f / %.3f ms (lower sweep, same ytemp: %.3f / %.3f)" % (
            i, zs[i].data_ptr(), first[i][1], second[i][1], first[i][0], second[i][0]), flush=True)
    up = np.array([f[1] for f in first])
    print("upper: min %.3f median %.3f max %.3f; candidates within 2 %% of the best: %d of %d; largest change of a "
          "candidate between the two passes %.1f %%" % (up.min(), np.median(up), up.max(), int((up < 1.02 * up.min()).sum()), K,
                                                     100 * max(abs(a[1] - b[1]) / a[1] for a, b in zip(first, second))))
    # the same question for the right-hand side the lower sweep streams
    rs = [r.clone() for _ in range(4)]
    for i, rr in enumerate(rs):
        lo, upv = measure(p, rr, zs[0])
        print("rhs copy %d: lower %.3f ms, upper %.3f ms" % (i, lo, upv), flush=True)
    p.close()
    # the same vectors under a NEW operator (new ytemp, new triangle copies): do they keep their mode?
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    p.set_timing(True)
    third = [measure(p, r, z) for z in zs]
    for i in range(K):
        print("candidate %2d under a second operator: upper %.3f ms (was %.3f), lower %.3f" % (
            i, third[i][1], first[i][1], third[i][0]), flush=True)
    p.close()


if __name__ == "__main__":
    main()
