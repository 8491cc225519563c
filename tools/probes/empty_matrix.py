import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
from blasted_amd import capi
for bs in (1, 4, 5):
    m = dict(nbrows=0, nnzb=0, bs=bs, rowmajor=False, browptr=np.zeros(1, np.int32), bcolind=np.zeros(0, np.int32),
             diagind=np.zeros(0, np.int32), vals=np.zeros(0))
    p = capi.Prec(0)
    try:
        p.set_matrix(m)
        r = np.zeros(0)
        p.ilu0_factorize(3)
        p.ilu0_factorize(-1)
        z = p.ilu0_apply(r, 3)
        z = p.ilu0_apply(r, 1, mode=capi.LEVEL)
        p.jacobi_compute()
        p.sgs_apply(r, 2)
        p.sgs_relax(r, np.zeros(0), 2)
        p.spmv(r)
        print("bs", bs, "empty matrix ok", z.shape)
    except capi.BlastedHipError as e:
        print("bs", bs, "rejected:", e)
    p.close()
