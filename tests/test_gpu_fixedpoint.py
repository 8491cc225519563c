"""GPU mirrors of the reference's "ExactFixedPoint" cases and of the block INIT_F_NONE warm start (VERDICT r03,
"What's missing" 2 and 3).

Reference: tests/solverops/CMakeLists.txt:42-84 -- AsyncILU-ExactFixedPoint-{CSR,BSR4,Scaled-BSR4}-2dcyl and
AsyncILUTriangular-ExactFixedPoint-{CSR,BSR4}-2dcyl: `-initialization exact -max_sweeps 5 -tolerance 1e-16`; the drivers
(tests/solverops/async_ilu_convergence.cpp:357-375, async_triangular_factors_convergence.cpp:121-141) start the
ASYNCHRONOUS sweeps from the exact result and require that they do not move it.  The reference compares with the result
of its own serial sweep -- the same arithmetic, so its tolerance is 1e-16; here the exact result comes from the CPU
oracle (different rounding than the GPU kernels: FMA contraction, matrix-core accumulation order), so the mirror is
two-fold: the sweeps stay within a few ulps of the oracle's exact result, and they are BIT-stationary in the GPU's own
arithmetic (more sweeps change nothing)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from blasted_amd import capi, mtxio, workloads as W

pytestmark = pytest.mark.gpu


def G(golden, name):
    return os.path.join(golden, name)


def upload_factor(p, values):
    """host array -> the operator's factor storage in HBM (blasted_hip_iluvals_device + blasted_hip_buffer_upload)"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    L = capi.lib()
    L.blasted_hip_buffer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong]
    p.synchronize()
    capi._check(L.blasted_hip_buffer_upload(C.c_void_p(p.iluvals_device_ptr()), v.ctypes.data, v.nbytes))


def split(m, f):
    """(strictly lower + strictly upper entries, diagonal blocks) of a factor array"""
    bs2 = m["bs"] ** 2
    blocks = f.reshape(-1, bs2)
    return np.delete(blocks, m["diagind"], axis=0), blocks[m["diagind"]]


def cases(golden):
    return {
        "2dcyl1_csr": lambda: mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 1),
        "2dcyl1_bsr4": lambda: mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 4, False),
        "poisson12_bs5": lambda: W.poisson3d(12, 5),
        "poisson9_bs8": lambda: W.poisson3d(9, 8),
    }


@pytest.mark.parametrize("case,usescale", [("2dcyl1_csr", False), ("2dcyl1_bsr4", False), ("2dcyl1_bsr4", True),
                                           ("poisson12_bs5", False), ("poisson9_bs8", False)])
def test_async_factor_sweeps_hold_the_exact_fixed_point(golden, case, usescale):
    """AsyncILU-ExactFixedPoint-{CSR, BSR4, Scaled-BSR4} (+ our bs 5 and 8): the exact factor -- diagonal blocks as the
    sweeps iterate on them, not yet inverted -- is uploaded, 5 asynchronous sweeps run from it (INIT_F_NONE)."""
    m = cases(golden)[case]()
    bs = m["bs"]
    exact_iter = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale, invert_diag=False)["iluvals"]
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"]
    p = capi.Prec(0)
    p.set_matrix(m)
    p.ilu0_factorize(0, init=capi.INIT_F_ORIGINAL, usescale=usescale, mode=capi.ASYNC)   # allocates the factor storage
    got = {}
    nlev = int(W.dependency_levels(m).max()) + 1
    for sweeps in (5, nlev + 2, nlev + 7):
        upload_factor(p, exact_iter)
        p.ilu0_factorize(sweeps, init=capi.INIT_F_NONE, usescale=usescale, mode=capi.ASYNC)
        got[sweeps] = p.get_iluvals()
    assert np.all(np.isfinite(got[5]))
    off, diag = split(m, got[5])
    eoff, ediag = split(m, exact)
    # strictly lower and upper entries: stored as computed -- within a few ulps of the oracle's exact ones
    assert np.abs(off - eoff).max() <= 4e-15 * np.abs(eoff).max()
    if bs == 1:
        assert np.abs(diag - ediag).max() <= 4e-15 * np.abs(ediag).max()
    else:
        # diagonal blocks are stored inverted: rounding differences of the two inversions scale with the blocks'
        # condition numbers, so compare through the product with the exact (un-inverted) block
        d = diag.reshape(-1, bs, bs)
        e = split(m, exact_iter)[1].reshape(-1, bs, bs)
        if not m.get("rowmajor"):
            d, e = d.transpose(0, 2, 1), e.transpose(0, 2, 1)
        assert np.abs(d @ e - np.eye(bs)).max() <= 1e-12
    # ... and in the GPU's OWN arithmetic the fixed point is stationary to the last bit (the oracle's differs from it in
    # the last ulps, which takes as many sweeps as the pattern has dependency levels to settle): five more sweeps change
    # nothing
    assert np.array_equal(got[nlev + 7], got[nlev + 2])
    assert np.abs(got[nlev + 2] - got[5]).max() <= 4e-15 * np.abs(got[5]).max()
    p.close()


@pytest.mark.parametrize("case", ["2dcyl1_csr", "2dcyl1_bsr4", "poisson12_bs5", "poisson9_bs8"])
def test_async_triangular_sweeps_hold_the_exact_fixed_point(golden, case):
    """AsyncILUTriangular-ExactFixedPoint-{CSR, BSR4} (+ bs 5, 8): the lower and upper sweeps start from the exact
    y = L^-1 r and z = U^-1 y (test hook `applynone=1`: blasted_hip_ilu0_apply takes INIT_A_NONE and keeps both
    iterates) and 5 asynchronous sweeps leave them where they are."""
    import torch
    m = cases(golden)[case]()
    n = m["nbrows"] * m["bs"]
    p = capi.Prec(0)
    p.set_matrix(m)
    p.ilu0_factorize(-1)
    f = p.get_iluvals()
    r = W.rhs_vector(n)
    ze, ye = O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL, return_y=True)
    rd = torch.from_numpy(r).cuda()
    z = p.ilu0_apply(rd, 1, mode=capi.LEVEL)          # the GPU's exact solves leave y in the operator, z here
    assert np.abs(z.cpu().numpy() - ze).max() <= 1e-12 * np.abs(ze).max()
    capi.set_tuning("applynone=1")
    try:
        res = {}
        nlev = int(W.dependency_levels(m).max()) + 1
        for sweeps in (5, nlev + 2, nlev + 7):
            z = p.ilu0_apply(rd, 1, mode=capi.LEVEL)
            p.ilu0_apply(rd, sweeps, init=capi.INIT_A_NONE, mode=capi.ASYNC, out=z)
            res[sweeps] = (p.get_ytemp(), z.cpu().numpy())
    finally:
        capi.set_tuning("applynone=0")
    y5, z5 = res[5]
    assert np.abs(y5 - ye).max() <= 1e-14 * np.abs(ye).max()
    assert np.abs(z5 - ze).max() <= 1e-13 * np.abs(ze).max()
    # stationary to the last bit in the sweep kernels' own arithmetic once every dependency level has been passed
    assert np.array_equal(res[nlev + 7][0], res[nlev + 2][0]) and np.array_equal(res[nlev + 7][1], res[nlev + 2][1])
    # without the hook INIT_A_NONE is the reference's error (src/solverops_ilu0.cpp:125-126)
    with pytest.raises(capi.BlastedHipError):
        p.ilu0_apply(rd, 1, init=capi.INIT_A_NONE, mode=capi.ASYNC)
    p.close()


@pytest.mark.parametrize("case", ["poisson16_bs4", "poisson12_bs5", "poisson9_bs8", "2dcyl1_bsr4"])
def test_block_warm_start_init_none_starts_from_the_stored_factor(golden, case):
    """SURVEY Q2, decided: a second factorisation with INIT_F_NONE starts from the factor storage AS IT IS -- for block
    matrices with the diagonal blocks INVERTED, the state the first factorisation leaves (the reference does the same:
    include/async_initialization_decl.hpp:22-34, src/async_blockilu_factor.cpp:47-149 has no case for it and :143-146
    inverts in place).  Synchronous sweeps equal the oracle's from that very array; they are NOT the continuation of
    the first build; and the iteration still reaches the exact factor (every entry depends on earlier ones only)."""
    m = dict(cases(golden), poisson16_bs4=lambda: W.poisson3d(16, 4))[case]()
    p = capi.Prec(0)
    p.set_matrix(m)
    p.ilu0_factorize(2, init=capi.INIT_F_ORIGINAL, mode=capi.JACOBI_SYNC)
    stored = p.get_iluvals()                      # diagonal blocks inverted
    p.ilu0_factorize(3, init=capi.INIT_F_NONE, mode=capi.JACOBI_SYNC)
    got = p.get_iluvals()
    want = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC, init=O.INIT_F_NONE, iluvals=stored)["iluvals"]
    # (sweeps that start from inverted diagonal blocks pass through badly scaled iterates -- 2dcyl1: entries up to 237 --
    # and magnify the last-bit differences between the two implementations: 1.2e-11 seen, where ordinary sweeps meet 1e-12)
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    continued = O.ilu0_factorize(m, None, 5, mode=O.JACOBI_SYNC, init=O.INIT_F_ORIGINAL)["iluvals"]
    assert np.abs(got - continued).max() > 1e-6 * np.abs(continued).max()
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    nlev = int(W.dependency_levels(m).max()) + 1
    p.ilu0_factorize(nlev + 2, init=capi.INIT_F_NONE, mode=capi.ASYNC)
    assert np.abs(p.get_iluvals() - exact).max() <= 1e-10 * np.abs(exact).max()
    p.close()
