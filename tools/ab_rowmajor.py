#!/usr/bin/env python3
"""A/B of the tuned row-major bs=4/8 sweep kernel (tuning "sweepwr=0/1") beside the column-major one on the
same matrix with transposed blocks; `ab_rowmajor.py odd`: bs = 5 / 7 / 3 (kernels_sweepodd.hip: "sweepodd=0" is the
general kernel the row-major blocks of these sizes went through before round 4)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from blasted_amd import capi, workloads as W
dev = torch.device("cuda", 0)
capi.set_tuning("compactafter=0")   # the compact copies with the first application: no copy pass inside a timed loop
ODD = len(sys.argv) > 1 and sys.argv[1] == "odd"
for bs in ((5, 7, 3) if ODD else (4, 8)):
    n = {4: 160, 8: 100, 5: 110, 7: 90, 3: 140}[bs]
    m = W.poisson3d_device(n, bs, dev, grid="uniform")
    # row-major copy: transpose every block
    v = m["vals"].view(-1, bs, bs).transpose(1, 2).contiguous().view(-1)
    mr = dict(m); mr["vals"] = v; mr["rowmajor"] = True
    r = W.rhs_vector_device(m["nbrows"] * bs, dev); z = torch.zeros_like(r)
    for name, mm in (("colmajor", m), ("rowmajor", mr)):
        p = capi.Prec(0, torch.cuda.current_stream().cuda_stream); p.set_matrix(mm); p.ilu0_factorize(2); p.jacobi_compute()
        p.ilu0_apply(r, 3, out=z); p.sgs_apply(r, 3, out=z); torch.cuda.synchronize()
        for spec in (("sweepodd=0", "sweepodd=1") if ODD else ("sweepwr=0", "sweepwr=1")):
            capi.set_tuning(spec)
            def t(fn, reps=10):
                fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps): fn()
                torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
            print("bs=%d %s %s: ilu apply s=3 %.3f ms, sgs apply s=3 %.3f ms, spmv %.3f ms, relax 1 it %.3f ms" % (
                bs, name, spec, t(lambda: p.ilu0_apply(r, 3, out=z)), t(lambda: p.sgs_apply(r, 3, out=z)),
                t(lambda: p.spmv(r, out=z)), t(lambda: p.sgs_relax(r, z, 1))), flush=True)
        p.close()
