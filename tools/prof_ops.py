#!/usr/bin/env python3
"""Runs factor / apply / spmv a few times for one block size (to be wrapped by rocprofv3 --stats)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blasted_amd import capi, workloads
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--bs", type=int, default=8)
ap.add_argument("--unstructured", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = workloads.unstructured_bsr(a.n, a.bs, device=dev) if a.unstructured else workloads.poisson3d_device(a.n, a.bs, dev, grid="uniform")
r = workloads.rhs_vector_device(m["nbrows"] * a.bs, dev)
z = torch.zeros_like(r)
p = capi.Prec(0)
p.set_matrix(m)
for _ in range(3):
    p.ilu0_factorize(3)
for _ in range(5):
    p.ilu0_apply(r, 3, out=z)
    p.spmv(r, out=z)
p.jacobi_compute()
for _ in range(3):
    p.sgs_relax(r, z, 2)
torch.cuda.synchronize()
print("rows", m["nbrows"], "nnzb", m["nnzb"])
