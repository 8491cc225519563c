"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical
inputs.  Tiers:
  P0  integer structures bit-exact;
  P2  HIP JACOBI_SYNC == oracle JACOBI_SYNC at the same sweep counts, rel <= 1e-12;
  P3  HIP ASYNC run to convergence == oracle exact (serial) result, rel <= 1e-10
      (the north-star's "FP within 1e-10 rel for the Poisson-FD case");
plus the reference's known-answer fixtures run on the GPU.
"""
import os

import numpy as np
import pytest

import oracle as O
from blasted_amd import capi, mtxio, workloads as W
from krylov import bicgstab

pytestmark = pytest.mark.gpu

TOL_SYNC = 1e-12
TOL_EXACT = 1e-10
DBL_EPS = np.finfo(np.float64).eps


def G(golden, name):
    return os.path.join(golden, name)


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def mtxio_vec(golden, name):
    return mtxio.read_mtx_dense(G(golden, name))


def make_prec(m):
    p = capi.Prec(0)
    p.set_matrix(m)
    return p


@pytest.fixture(params=["compact", "inplace"])
def factor_storage(request):
    """The asynchronous ILU sweeps read compact copies of the factor's triangles (default) or the factor
    in place: the tests of the apply run with both."""
    capi.set_tuning("compact=" + ("1" if request.param == "compact" else "0"))
    yield request.param
    capi.set_tuning("compact=1")


def matrices(golden):
    return {
        "2dcyl1_bs4_col": lambda: mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 4, False),
        "2dcyl1_bs4_row": lambda: mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 4, True),
        "2dcyl1_csr": lambda: mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 1),
        "msc_csr": lambda: mtxio.read_mtx_bsr(G(golden, "msc00726.mtx"), 1),
        "poisson16_csr": lambda: W.poisson3d(16, 1),
        "poisson16_bs4": lambda: W.poisson3d(16, 4),
        "poisson11_bs4_row": lambda: W.poisson3d(11, 4, rowmajor=True),
        "poisson12_bs5": lambda: W.poisson3d(12, 5),
        "poisson9_bs8": lambda: W.poisson3d(9, 8),
        "poisson8_bs3": lambda: W.poisson3d(8, 3),
        "poisson8_bs7_row": lambda: W.poisson3d(8, 7, rowmajor=True),
        "poisson8_bs2": lambda: W.poisson3d(8, 2),
        "random_bs5": lambda: W.random_bsr(1500, 5, avg_offdiag=8, seed=12345),
        "random_bs4": lambda: W.random_bsr(777, 4, avg_offdiag=5, seed=7),
        "random_csr": lambda: W.random_bsr(1001, 1, avg_offdiag=6, seed=11),
        # row-major blocks at the sizes whose factorisation kernels have a row-major form of their own (round 4)
        "poisson10_bs5_row": lambda: W.poisson3d(10, 5, rowmajor=True),
        "poisson8_bs8_row": lambda: W.poisson3d(8, 8, rowmajor=True),
        "random_bs5_row": lambda: W.random_bsr(900, 5, avg_offdiag=7, seed=21, rowmajor=True),
        "random_bs4_row": lambda: W.random_bsr(777, 4, avg_offdiag=5, seed=7, rowmajor=True),
        "random_bs8_row": lambda: W.random_bsr(400, 8, avg_offdiag=5, seed=5, rowmajor=True),
    }


ALL = ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "2dcyl1_csr", "msc_csr", "poisson16_csr", "poisson16_bs4",
       "poisson11_bs4_row", "poisson12_bs5", "poisson9_bs8", "poisson8_bs3", "poisson8_bs7_row",
       "poisson8_bs2", "random_bs5", "random_bs4", "random_csr", "poisson10_bs5_row", "poisson8_bs8_row",
       "random_bs5_row", "random_bs4_row", "random_bs8_row"]


# ---------------------------------------------------------------------------- P0 integer structures

@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "msc_csr", "poisson16_csr", "random_bs5", "random_csr"])
def test_ilu_positions_bit_exact(golden, case):
    m = matrices(golden)[case]()
    p = make_prec(m)
    got = p.ilu0_positions()
    want = O.ilu_positions(m)
    for g, w in zip(got, want):
        assert g.dtype == np.int32 and np.array_equal(g, w)
    p.close()


def test_invalid_patterns_are_rejected():
    m = W.poisson3d(6, 4)
    bad = dict(m)
    bc = m["bcolind"].copy()
    bc[[0, 1]] = bc[[1, 0]]  # unsorted columns in row 0
    bad["bcolind"] = bc
    p = capi.Prec(0)
    with pytest.raises(capi.BlastedHipError) as ei:
        p.set_matrix(bad)
    assert ei.value.code == capi.EINVAL
    bad = dict(m)
    dg = m["diagind"].copy()
    dg[3] += 1
    bad["diagind"] = dg
    p2 = capi.Prec(0)
    with pytest.raises(capi.BlastedHipError):
        p2.set_matrix(bad)
    bad = dict(m)
    bad["bs"] = 6
    p3 = capi.Prec(0)
    with pytest.raises(capi.BlastedHipError) as ei:
        p3.set_matrix(bad)
    assert ei.value.code == capi.ENOTIMPL


@pytest.mark.parametrize("bs", [1, 4, 5])
def test_empty_subdomain_is_a_no_op(bs):
    """A rank whose subdomain has no rows (nbrows = 0): every operator succeeds and returns empty vectors."""
    m = dict(nbrows=0, nnzb=0, bs=bs, rowmajor=False, browptr=np.zeros(1, np.int32),
             bcolind=np.zeros(0, np.int32), diagind=np.zeros(0, np.int32), vals=np.zeros(0))
    p = capi.Prec(0)
    p.set_matrix(m)
    r = np.zeros(0)
    p.ilu0_factorize(3)
    p.ilu0_factorize(-1)
    assert p.ilu0_apply(r, 3).shape == (0,)
    assert p.ilu0_apply(r, 1, mode=capi.LEVEL).shape == (0,)
    assert p.ilu0_apply(r, 2, mode=capi.JACOBI_SYNC).shape == (0,)
    p.jacobi_compute()
    assert p.jacobi_apply(r).shape == (0,)
    assert p.sgs_apply(r, 2).shape == (0,)
    assert p.sgs_apply(r, 1, mode=capi.LEVEL).shape == (0,)
    p.sgs_relax(r, np.zeros(0), 2)
    p.sgs_relax(r, np.zeros(0), 1, mode=capi.LEVEL)
    p.gs_relax(r, np.zeros(0), 2)
    assert p.spmv(r).shape == (0,)
    assert p.level_count() == 0
    p.close()
    # the same through the raw ABI with the null pointers an empty std::vector hands over
    import ctypes as C
    h = C.c_void_p(0)
    L = capi.lib()
    assert L.blasted_hip_create(C.byref(h), 0, None, 1) == 0
    rp = np.zeros(1, np.int32)
    assert L.blasted_hip_set_pattern(h, 0, 0, bs, 0, rp.ctypes.data_as(C.c_void_p), None, None, capi.HOST) == 0
    assert L.blasted_hip_set_values(h, None, capi.HOST) == 0
    assert L.blasted_hip_ilu0_factorize(h, 3, capi.INIT_F_ORIGINAL, 0, capi.ASYNC, None) == 0
    L.blasted_hip_destroy(h)


def test_memory_stats_and_one_derived_copy_per_triangle():
    """An operator that is applied both ways keeps each triangle of the factor in ONE derived ordering (the one
    asked for last) unless copies=both; blasted_hip_memory_stats follows every allocation."""
    m = W.poisson3d(14, 4)
    n = m["nbrows"] * 4
    r = W.rhs_vector(n)
    p = make_prec(m)
    base = p.memory_stats()
    assert base["derived_copies"] == 0 and base["bytes"] >= m["vals"].nbytes   # pattern + value mirror
    p.ilu0_factorize(3)
    fbytes = m["vals"].nbytes
    after_f = p.memory_stats()
    assert after_f["bytes"] >= base["bytes"] + fbytes and after_f["derived_copies"] == 0
    za = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)          # natural-order triangles
    s1 = p.memory_stats()
    assert s1["derived_copies"] == 2 and s1["bytes"] >= after_f["bytes"] + fbytes
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL)                # level-ordered triangles replace them
    s2 = p.memory_stats()
    assert s2["derived_copies"] == 2
    assert np.array_equal(za, p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC))   # and back: same result
    assert p.memory_stats()["derived_copies"] == 2
    capi.set_tuning("copies=both")
    try:
        p.ilu0_apply(r, 1, mode=capi.LEVEL)
        assert p.memory_stats()["derived_copies"] == 4
        assert np.array_equal(ze, p.ilu0_apply(r, 1, mode=capi.LEVEL))
    finally:
        capi.set_tuning("copies=one")
    assert p.memory_stats()["peak_bytes"] >= p.memory_stats()["bytes"]
    # the product-mode SGS application holds the level-ordered lower and the natural-order upper triangle
    q = make_prec(m)
    q.jacobi_compute()
    q.sgs_apply(r, 2, mode=capi.DETERMINISTIC)
    assert q.memory_stats()["derived_copies"] == 2
    q.close()
    p.close()


def test_host_register_is_explicit_and_checked():
    """blasted_hip_host_register page-locks a caller-owned range for the host-vector entry points; results are
    the same with and without it, double registration and unknown addresses are refused."""
    m = W.poisson3d(40, 4)          # 2.0 MB vectors
    n = m["nbrows"] * 4
    r = W.rhs_vector(n)
    z = np.zeros(n)
    p = make_prec(m)
    p.ilu0_factorize(-1)
    want = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    capi.host_register(r)
    capi.host_register(z)
    assert p.memory_stats()["pinned_host_bytes"] >= r.nbytes + z.nbytes
    with pytest.raises(capi.BlastedHipError) as ei:
        capi.host_register(r)
    assert ei.value.code == capi.ESTATE
    got = p.ilu0_apply(r, 1, mode=capi.LEVEL, out=z)
    assert np.array_equal(got, want)
    capi.host_unregister(r)
    capi.host_unregister(z)
    with pytest.raises(capi.BlastedHipError):
        capi.host_unregister(z)
    assert p.memory_stats()["pinned_host_bytes"] == 0
    assert np.array_equal(p.ilu0_apply(r, 1, mode=capi.LEVEL), want)
    p.close()


@pytest.mark.parametrize("bs", [1, 4, 5, 8])
def test_xcd_super_chunk_size_does_not_change_results(bs):
    """the run-time super-chunk size of the XCD-aware chunk numbering only reorders workgroups: synchronous
    sweeps are bit-identical for every value, and bad values are rejected"""
    m = W.poisson3d(20, bs)
    r = W.rhs_vector(m["nbrows"] * bs)
    p = make_prec(m)
    p.ilu0_factorize(2, mode=capi.JACOBI_SYNC)
    ref = None
    try:
        for x in (16, 1, 4, 64, 4096):
            capi.set_tuning("xcdsuper=%d" % x)
            z = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
            y = p.spmv(r)
            if ref is None:
                ref = (z, y)
            assert np.array_equal(z, ref[0]) and np.array_equal(y, ref[1])
        with pytest.raises(capi.BlastedHipError):
            capi.set_tuning("xcdsuper=12")
    finally:
        capi.set_tuning("xcdsuper=auto")
    p.close()


def test_measure_read_stream_reports_a_plausible_rate():
    """The measurement aid behind bench.py's roofline.measured_read_stream_gbps: a 256 MB buffer streams at
    well above 1 TB/s and below the 8 TB/s peak; misaligned and null buffers are rejected."""
    import ctypes as C
    import torch
    x = torch.zeros(1 << 25, dtype=torch.float64, device="cuda")
    g = capi.measure_read_stream(x, reps=5)
    assert 1000.0 < g < 8000.0, g
    out = C.c_double(0.0)
    rc = capi.lib().blasted_hip_measure_read_stream(C.c_void_p(x.data_ptr() + 8), C.c_ulong(1 << 20), 1, C.byref(out))
    assert rc == capi.EINVAL
    rc = capi.lib().blasted_hip_measure_read_stream(C.c_void_p(0), C.c_ulong(1 << 20), 1, C.byref(out))
    assert rc == capi.EINVAL


# ---------------------------------------------------------------------------- SpMV (K11)

@pytest.mark.parametrize("bs,rowmajor", [(1, False), (7, True), (7, False)])
def test_spmv_dk01r_known_answer(golden, bs, rowmajor):
    m = mtxio.read_mtx_bsr(G(golden, "DK01R.mtx"), bs, rowmajor)
    x = mtxio.read_mtx_dense(G(golden, "DK01R_x.mtx"))
    b = mtxio.read_mtx_dense(G(golden, "DK01R_b.mtx"))
    p = make_prec(m)
    y = p.spmv(x)
    assert np.all(np.abs(y - b) < 10 * DBL_EPS)  # tests/mat_ops/testbsrmatrix.cpp:46-48
    assert rel(y, O.spmv(m, x)) < 1e-13 or np.abs(y - O.spmv(m, x)).max() < 1e-14
    p.close()


@pytest.mark.parametrize("rowmajor", [False, True])
def test_spmv_small_block3_known_answer(golden, rowmajor):
    m = mtxio.read_mtx_bsr(G(golden, "small_block3_matrix.mtx"), 3, rowmajor)
    x = mtxio.read_mtx_dense(G(golden, "small_block3_matrix_x.mtx"))
    b = mtxio.read_mtx_dense(G(golden, "small_block3_matrix_b.mtx"))
    p = make_prec(m)
    assert np.all(np.abs(p.spmv(x) - b) < 10 * DBL_EPS)
    p.close()


@pytest.mark.parametrize("case", ALL)
def test_spmv_gemv3_vs_oracle(golden, case):
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    x = W.rhs_vector(n)
    y = np.cos(0.11 * np.arange(n))
    p = make_prec(m)
    assert rel(p.spmv(x), O.spmv(m, x)) < 1e-13
    assert rel(p.gemv3(-1.5, x, 0.25, y), O.gemv3(m, -1.5, x, 0.25, y)) < 1e-13
    p.close()


def test_spmv_2dcyl1_residual(golden):
    m = mtxio.read_mtx_bsr(G(golden, "2dcyl1.mtx"), 4)
    x = mtxio.read_mtx_dense(G(golden, "2dcyl1_x.mtx"))
    b = mtxio.read_mtx_dense(G(golden, "2dcyl1_b.mtx"))
    p = make_prec(m)
    assert np.linalg.norm(p.spmv(x) - b) < 1e-12
    p.close()


# ---------------------------------------------------------------------------- ILU(0) factorisation

@pytest.mark.parametrize("case", ALL)
@pytest.mark.parametrize("init", [capi.INIT_F_ORIGINAL, capi.INIT_F_SGS])
@pytest.mark.parametrize("usescale", [False, True])
def test_ilu_factor_sync_sweeps_match_oracle(golden, case, init, usescale):
    m = matrices(golden)[case]()
    if usescale and case.startswith("random"):
        pytest.skip("random test matrices may have negative diagonal entries (sqrt)")
    p = make_prec(m)
    for sweeps in (1, 3):
        p.ilu0_factorize(sweeps, init=init, usescale=usescale, mode=capi.JACOBI_SYNC)
        want = O.ilu0_factorize(m, None, sweeps, mode=O.JACOBI_SYNC, init=init, usescale=usescale)
        assert rel(p.get_iluvals(), want["iluvals"]) < TOL_SYNC
        if usescale:
            assert rel(p.get_scale(), want["scale"]) < 1e-15
    p.close()


@pytest.mark.parametrize("case", ALL)
def test_ilu_factor_async_converges_to_exact(golden, case):
    m = matrices(golden)[case]()
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, init=O.INIT_F_ORIGINAL, compute_info=True)
    p = make_prec(m)
    nsw = 90
    info = p.ilu0_factorize(nsw, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC, compute_info=True)
    assert rel(p.get_iluvals(), exact["iluvals"]) < TOL_EXACT
    # -blasted_compute_preconditioner_info, tests/testutils.cpp:297-308
    assert info[1] > 0 and info[0] / info[1] < 1e-12
    assert rel(info[1:2], exact["precinfo"][1:2]) < 1e-12          # initial remainder: same input
    assert np.abs(info[2:] - exact["precinfo"][2:]).max() < 1e-8   # diagonal dominance of the factors
    p.close()


ZERO_INIT_CASES = ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "poisson16_bs4", "poisson11_bs4_row", "poisson12_bs5",
                   "poisson9_bs8", "poisson8_bs3", "poisson8_bs7_row", "poisson8_bs2", "random_bs5", "random_bs4",
                   "poisson10_bs5_row", "poisson8_bs8_row", "random_bs4_row", "random_bs8_row"]


@pytest.mark.parametrize("case", ZERO_INIT_CASES)
@pytest.mark.parametrize("factorskip", ["1", "0"])
@pytest.mark.parametrize("usescale", [False, True])
def test_block_zero_init_factorisation_converges_to_exact(golden, case, factorskip, usescale):
    """INIT_F_ZERO for the BLOCK factorisation (src/async_blockilu_factor.cpp:65-69), the initial guess of the
    reference's own native cases (tests/CMakeLists.txt:104-111, 157-173: BSR4ILU0{Row,Col}major,
    ThreadedBSR4ILU0Colmajor, all `--fact_init_type init_zero`).  From a zero factor the first in-place sweeps
    invert diagonal blocks that are still zero: the lower blocks that depend on them are NaN / inf until the
    rows they read are final, and then they are recomputed from the matrix block -- nothing non-finite may
    survive (neither through the adjugate inverse, nor the MFMA accumulation, nor the fixed-upper-block
    shortcut, which may only skip a block once a sweep has stored it).  After max(90, 2 x levels + 2) sweeps the
    factor is the exact one to 1e-10, with and without the shortcut, and the two agree bit for bit on the blocks
    the shortcut leaves alone."""
    if usescale and case.startswith("random"):
        pytest.skip("random test matrices may have negative diagonal entries (sqrt)")
    m = matrices(golden)[case]()
    bs2 = m["bs"] ** 2
    nlev = int(W.dependency_levels(m).max()) + 1
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, init=O.INIT_F_ZERO, usescale=usescale, compute_info=True)
    # the serial sweep does not depend on the initial guess (every entry it reads it has already written)
    assert np.array_equal(exact["iluvals"], O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"])
    p = make_prec(m)
    capi.set_tuning("factorskip=" + factorskip)
    try:
        info = p.ilu0_factorize(max(90, 2 * nlev + 2), init=capi.INIT_F_ZERO, usescale=usescale, mode=capi.ASYNC, compute_info=True)
        got = p.get_iluvals()
        assert np.all(np.isfinite(got))
        assert rel(got, exact["iluvals"]) < TOL_EXACT
        # PrecInfo: the initial remainder of a zero factor is ||A|| (scaled), the final one vanishes
        assert info[1] > 0 and info[0] / info[1] < 1e-12
        assert rel(info[1:2], exact["precinfo"][1:2]) < 1e-12
        # a few sweeps only (the reference's threaded case builds with 10): finite wherever the oracle's
        # synchronous model of the same sweeps is, i.e. the non-finite entries of early sweeps do not spread
        p.ilu0_factorize(3, init=capi.INIT_F_ZERO, usescale=usescale, mode=capi.JACOBI_SYNC)
        g3 = p.get_iluvals().reshape(-1, bs2)
        w3 = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC, init=O.INIT_F_ZERO, usescale=usescale)["iluvals"].reshape(-1, bs2)
        # (diagonal blocks are stored inverted at the end: a block with one bad entry is bad as a whole)
        fin_w = np.isfinite(w3).all(axis=1)
        fin_g = np.isfinite(g3).all(axis=1)
        assert np.array_equal(fin_w, fin_g)
        assert fin_w.any() and rel(g3[fin_w], w3[fin_w]) < TOL_SYNC
    finally:
        capi.set_tuning("factorskip=1")
    p.close()


@pytest.mark.parametrize("case", ["poisson16_bs4", "2dcyl1_bs4_col", "poisson9_bs8", "poisson12_bs5", "poisson16_csr"])
@pytest.mark.parametrize("init", [capi.INIT_F_ORIGINAL, capi.INIT_F_SGS, capi.INIT_F_ZERO])
@pytest.mark.parametrize("usescale", [False, True])
def test_factor_sweeps_leave_fixed_upper_blocks_alone(golden, case, init, usescale):
    """In-place factorisation sweeps after the first neither read nor write an upper block without position pairs
    (tuning "factorskip"): its value is the (scaled) matrix block from the first sweep on.  Those blocks are the
    same bits with and without the shortcut, whatever the initial factor, and the factor converges to the exact
    one either way."""
    m = matrices(golden)[case]()
    bs2 = m["bs"] ** 2
    posptr = np.asarray(O.ilu_positions(m)[0])
    rp = np.asarray(m["browptr"])
    rowof = np.repeat(np.arange(m["nbrows"]), rp[1:] - rp[:-1])
    fixed = (np.asarray(m["bcolind"]) > rowof) & (posptr[1:] == posptr[:-1])
    assert fixed.any()
    p = make_prec(m)
    res = {}
    try:
        for k in ("1", "0"):
            capi.set_tuning("factorskip=" + k)
            p.ilu0_factorize(3, init=init, usescale=usescale)
            res[k] = p.get_iluvals().reshape(-1, bs2)
            # (from a ZERO factor three in-place sweeps may still hold NaN in lower blocks whose diagonal block
            # was inverted while zero -- so would the reference's threaded sweeps; the fixed upper blocks are the
            # matrix blocks from the first sweep on all the same)
            assert init == capi.INIT_F_ZERO or np.all(np.isfinite(res[k]))
            assert np.all(np.isfinite(res[k][fixed]))
        assert np.array_equal(res["1"][fixed], res["0"][fixed])
        a = np.asarray(m["vals"]).reshape(-1, bs2)
        if not usescale:
            assert np.array_equal(res["1"][fixed], a[fixed])
        capi.set_tuning("factorskip=1")
        p.ilu0_factorize(90, init=init, usescale=usescale)
        exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"]
        got = p.get_iluvals().reshape(-1, bs2)
        # (the factor is stored with inverted diagonal blocks after a factorisation: compare what the sweeps leave)
        off = np.ones(len(got), dtype=bool)
        off[np.asarray(m["diagind"])] = False
        assert rel(got[off], exact.reshape(-1, bs2)[off]) < TOL_EXACT
    finally:
        capi.set_tuning("factorskip=1")
    p.close()


def test_scalar_zero_init_falls_through(golden):
    """async_ilu_factor.cpp:48-54: for bs==1 INIT_F_ZERO behaves as INIT_F_ORIGINAL."""
    m = matrices(golden)["msc_csr"]()
    p = make_prec(m)
    p.ilu0_factorize(0, init=capi.INIT_F_ZERO)
    assert np.array_equal(p.get_iluvals(), m["vals"])
    p.close()


@pytest.mark.parametrize("case", ["msc_csr", "poisson16_csr", "2dcyl1_csr", "random_csr", "dense_rows_csr"])
@pytest.mark.parametrize("usescale", [False, True])
def test_scalar_factor_kernels_agree(golden, case, usescale):
    """The chunk-staged scalar factorisation kernel (kernels_factor1.hip) against the general one-lane-per-row
    kernel and the oracle: synchronous sweeps are the same arithmetic in the same order; in place both reach
    the exact factor.  dense_rows_csr has ~40 entries per row: most of a chunk lies beyond the staged range."""
    if case == "dense_rows_csr":
        m = W.random_bsr(600, 1, avg_offdiag=40, seed=3)
    else:
        m = matrices(golden)[case]()
    p = make_prec(m)
    res = {}
    for k in ("1", "0"):
        capi.set_tuning("factor1=" + k)
        try:
            p.ilu0_factorize(3, usescale=usescale, mode=capi.JACOBI_SYNC)
            res[k] = p.get_iluvals()
            p.ilu0_factorize(60, usescale=usescale, mode=capi.ASYNC)
            res[k + "x"] = p.get_iluvals()
        finally:
            capi.set_tuning("factor1=1")
    want = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC, usescale=usescale)["iluvals"]
    assert rel(res["1"], want) < TOL_SYNC and rel(res["1"], res["0"]) < 1e-14
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"]
    assert rel(res["1x"], exact) < TOL_EXACT and rel(res["0x"], exact) < TOL_EXACT
    p.close()


@pytest.mark.parametrize("shape", ["short_rows", "poisson5", "dense_rows"])
@pytest.mark.parametrize("init", [capi.INIT_F_ORIGINAL, capi.INIT_F_ZERO])
def test_scalar_plan_factor_kernel_is_the_staged_one_bit_for_bit(shape, init):
    """factor1p_kernel (precomputed plan, 16-bit LDS positions, parallel division of pair-free lower entries) against
    factor1_kernel on matrices of at most 128 rows = ONE workgroup, where an in-place sweep is deterministic: every
    sweep must give the same bits, non-finite early INIT_F_ZERO values included.  dense_rows has ~40 entries per
    row: 2 400 entries against a staging capacity of 1 024, so most of the chunk takes the from-memory path, whose
    results go to memory while other lanes read them (chaotic in both kernels): there the check is the fixed point."""
    m = {"short_rows": lambda: W.random_bsr(100, 1, avg_offdiag=6, seed=2), "poisson5": lambda: W.poisson3d(5, 1),
         "dense_rows": lambda: W.random_bsr(100, 1, avg_offdiag=40, seed=3)}[shape]()
    assert m["nbrows"] <= 128
    p = make_prec(m)
    try:
        for sweeps in (1, 2, 3, 7):
            res = {}
            for k in ("0", "1"):
                capi.set_tuning("factor1plan=" + k)
                p.ilu0_factorize(sweeps, init=init, mode=capi.ASYNC)
                res[k] = p.get_iluvals()
            if shape != "dense_rows":
                assert np.array_equal(res["0"], res["1"], equal_nan=True), sweeps
        exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
        for k in ("0", "1"):
            capi.set_tuning("factor1plan=" + k)
            p.ilu0_factorize(120, init=init, mode=capi.ASYNC)
            assert rel(p.get_iluvals(), exact) < TOL_EXACT, k
    finally:
        capi.set_tuning("factor1plan=1")
        p.close()


@pytest.mark.parametrize("case", ["poisson16_csr", "random_csr", "2dcyl1_csr"])
def test_scalar_plan_factor_kernel_reaches_the_exact_factor(golden, case):
    """Many chunks: the two in-place scalar kernels reach the same exact factor (P3), from both initialisations."""
    m = matrices(golden)[case]()
    p = make_prec(m)
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    try:
        for k in ("0", "1"):
            capi.set_tuning("factor1plan=" + k)
            for init, sweeps in ((capi.INIT_F_ORIGINAL, 60), (capi.INIT_F_ZERO, 120)):
                p.ilu0_factorize(sweeps, init=init, mode=capi.ASYNC)
                assert rel(p.get_iluvals(), exact) < TOL_EXACT, (k, init)
    finally:
        capi.set_tuning("factor1plan=1")
        p.close()


@pytest.mark.parametrize("bs", [8])
@pytest.mark.parametrize("shape", ["poisson9", "poisson5", "random_short", "random_long"])
@pytest.mark.parametrize("usescale", [False, True])
def test_block_in_place_factor_row_path_reaches_the_exact_factor(bs, shape, usescale):
    """The up-front row path of factor8_kernel (in place: a row's matrix blocks, the inverses for its lower blocks and
    the u_kj of its pairs requested together, finished lower blocks handed on in registers) against the
    block-by-block loop (factor8=2) and the general kernel (factor8=0): all reach the exact factor from both
    initialisations.  random_short mixes rows that fit the path's window (<= 3 lower blocks, <= 4 pairs) with rows
    that do not; random_long has none that fit.  (The same path at bs=4 was measured slower and removed.)"""
    m = {"poisson9": lambda: W.poisson3d(9, bs), "poisson5": lambda: W.poisson3d(5, bs),
         "random_short": lambda: W.random_bsr(400, bs, avg_offdiag=3, seed=21),
         "random_long": lambda: W.random_bsr(150, bs, avg_offdiag=14, seed=22)}[shape]()
    if usescale and shape.startswith("random"):
        pytest.skip("random test matrices may have negative diagonal entries (sqrt)")
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"]
    p = make_prec(m)
    try:
        for k in ("1", "2", "0"):
            capi.set_tuning("factor%d=%s" % (bs, k))
            for init, sweeps in ((capi.INIT_F_ORIGINAL, 60), (capi.INIT_F_ZERO, 120)):
                p.ilu0_factorize(sweeps, init=init, usescale=usescale, mode=capi.ASYNC)
                assert rel(p.get_iluvals(), exact) < TOL_EXACT, (k, init)
        # one sweep from the same start: the three kernels do the same arithmetic on iterates that differ only by
        # which neighbours' updates a row happened to see
        res = {}
        for k in ("1", "2"):
            capi.set_tuning("factor%d=%s" % (bs, k))
            p.ilu0_factorize(1, init=capi.INIT_F_ORIGINAL, usescale=usescale, mode=capi.ASYNC)
            res[k] = p.get_iluvals()
        assert np.all(np.isfinite(res["1"])) and rel(res["1"], res["2"]) < 0.2
    finally:
        capi.set_tuning("factor%d=1" % bs)
        p.close()


@pytest.mark.parametrize("shape", ["poisson12_bs5", "poisson8_bs7", "random_bs5", "random_bs7", "random_long_bs5"])
def test_odd_block_factor_kernels_agree(shape):
    """factorodd_kernel (16 / 32 lanes per block-row, products through LDS tiles) against the general kernel
    (factorodd=0): synchronous sweeps match the oracle's, in place both reach the exact factor from both
    initialisations.  random_long_bs5 has ~30 blocks a row and many pairs per block."""
    m = {"poisson12_bs5": lambda: W.poisson3d(12, 5), "poisson8_bs7": lambda: W.poisson3d(8, 7),
         "random_bs5": lambda: W.random_bsr(1500, 5, avg_offdiag=8, seed=12345),
         "random_bs7": lambda: W.random_bsr(500, 7, avg_offdiag=6, seed=5),
         "random_long_bs5": lambda: W.random_bsr(300, 5, avg_offdiag=30, seed=9)}[shape]()
    nlev = int(W.dependency_levels(m).max()) + 1
    p = make_prec(m)
    res = {}
    try:
        for k in ("1", "0"):
            capi.set_tuning("factorodd=" + k)
            p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)
            res[k] = p.get_iluvals()
            p.ilu0_factorize(90, mode=capi.ASYNC)
            res[k + "x"] = p.get_iluvals()
            # (from a zero factor the finite values advance at least one dependency level per sweep)
            p.ilu0_factorize(max(90, 2 * nlev + 2), init=capi.INIT_F_ZERO, mode=capi.ASYNC)
            res[k + "z"] = p.get_iluvals()
    finally:
        capi.set_tuning("factorodd=1")
        p.close()
    want = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC)["iluvals"]
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    assert rel(res["1"], want) < TOL_SYNC and rel(res["0"], want) < TOL_SYNC
    for k in ("1x", "0x", "1z", "0z"):
        assert rel(res[k], exact) < TOL_EXACT, k


@pytest.mark.parametrize("case", ["poisson16_bs4", "2dcyl1_bs4_col", "2dcyl1_bs4_row", "poisson12_bs5", "poisson9_bs8",
                                  "poisson8_bs3", "poisson8_bs7_row", "poisson8_bs2", "random_bs5", "random_bs4",
                                  "poisson16_csr", "random_csr", "2dcyl1_csr", "msc_csr", "poisson10_bs5_row",
                                  "poisson8_bs8_row", "random_bs5_row", "random_bs4_row", "random_bs8_row"])
def test_fused_initialisation_builds_the_same_factor(golden, case):
    """Asynchronous builds from INIT_F_ORIGINAL fuse the initialisation pass into the first sweep (in = the matrix,
    out = the factor, a row's own lower blocks read back fresh): after ONE sweep the pair-less upper blocks hold
    their final value a_ij bit for bit and everything is finite; per sweep the build converges like the one with a
    separate pass (within a factor 10 of its distance to the exact factor after 1, 2 and 3 sweeps -- the two differ in
    what a row sees of its neighbours' first sweep: 1.0 ... 3.7 on the column-major cases, up to 7.1 on the random
    row-major ones, same figures on clean and on poisoned storage: tools/probes/fused_init_rowmajor.py); both reach it."""
    m = matrices(golden)[case]()
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    pos = O.ilu_positions(m)
    bs2 = m["bs"] ** 2
    rows = np.repeat(np.arange(m["nbrows"]), np.diff(m["browptr"]))
    fixed = (m["bcolind"] > rows) & (np.diff(pos[0]) == 0)
    fmask = np.repeat(fixed, bs2)
    p = make_prec(m)
    dist = {}
    try:
        for k in ("0", "1"):
            capi.set_tuning("factorfuse=" + k)
            for sweeps in (1, 2, 3):
                p.ilu0_factorize(sweeps, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
                f = p.get_iluvals()
                assert np.all(np.isfinite(f))
                if sweeps == 1:
                    assert np.array_equal(f[fmask], m["vals"][fmask])
                dist[k, sweeps] = rel(f, exact)
            p.ilu0_factorize(90, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
            assert rel(p.get_iluvals(), exact) < TOL_EXACT
        # the fused first sweep reads a row's own lower blocks back from the FACTOR storage right after other
        # lanes of the wave stored them.  A stale read would return what the storage held before -- so poison it: the
        # factor of a DIFFERENT matrix on the same pattern (values scaled row by row), then the fused build of the
        # original one must come out as it did on clean storage.
        rng = np.random.default_rng(3)
        other = m["vals"].reshape(m["nnzb"], -1) * rng.uniform(0.5, 2.0, size=(m["nnzb"], 1))
        for sweeps in (1, 2):
            p.set_values(np.ascontiguousarray(other.reshape(-1)))
            p.ilu0_factorize(5, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
            p.set_values(m["vals"])
            p.ilu0_factorize(sweeps, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
            f = p.get_iluvals()
            assert np.all(np.isfinite(f))
            assert rel(f, exact) < 10 * dist["0", sweeps] + 1e-13, (sweeps, rel(f, exact), dist)
            # ... and stays where the same build on clean storage got (a stale block of the other factor is an O(1) error)
            assert rel(f, exact) <= 4 * dist["1", sweeps] + 1e-13, (sweeps, rel(f, exact), dist)
    finally:
        capi.set_tuning("factorfuse=1")
        p.close()
    for sweeps in (1, 2, 3):
        assert dist["1", sweeps] < 10 * dist["0", sweeps] + 1e-13, (sweeps, dist)


def test_compact_copies_are_made_lazily(golden):
    """The compact triangle copies the asynchronous sweeps read are made when the (compactafter+1)-th sweep application
    since the last factorisation comes along, not before: a caller that refactorises every few applications never
    pays the copy pass.  Before and after, synchronous sweeps give the same bits; a new factorisation starts the
    count again and invalidates the copies."""
    m = matrices(golden)["poisson16_bs4"]()
    r = W.rhs_vector(m["nbrows"] * 4)
    p = make_prec(m)
    capi.set_tuning("compactafter=3")
    try:
        p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)  # (synchronous: the same factor both times)
        seen = []
        for k in range(5):
            seen.append((p.memory_stats()["derived_copies"], p.ilu0_apply(r, 2, mode=capi.JACOBI_SYNC)))
        seen.append((p.memory_stats()["derived_copies"], None))
        assert [c for c, _ in seen] == [0, 0, 0, 0, 2, 2]
        for _, z in seen[1:5]:
            assert np.array_equal(z, seen[0][1])
        p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)  # (synchronous: the same factor both times)
        z = p.ilu0_apply(r, 2, mode=capi.JACOBI_SYNC)
        assert np.array_equal(z, seen[0][1])
        p.jacobi_compute()
        c0 = p.memory_stats()["derived_copies"]
        for k in range(4):
            p.sgs_apply(r, 2, mode=capi.JACOBI_SYNC)
        assert p.memory_stats()["derived_copies"] >= c0
    finally:
        capi.set_tuning("compactafter=" + os.environ.get("BLASTED_HIP_COMPACT_AFTER", "-1"))  # what it was (conftest.py)
        p.close()


@pytest.mark.parametrize("case,threshold", [("poisson16_csr", 4), ("poisson16_bs4", 16), ("poisson12_bs5", 8),
                                            ("poisson9_bs8", 8), ("random_bs5", 8)])
def test_product_default_makes_the_copies_inside_one_solve(golden, case, threshold):
    """the SHIPPED policy (compactafter = -1: by block size -- 4 applications for scalar rows, 16 at bs = 4,
    8 otherwise) is crossed inside one sequence of applications, as a solve does: the sweeps read the factor in place,
    then switch kernels and data layout once the threshold-th application has come -- same bits before and after
    (synchronous sweeps), the oracle's result throughout, and the asynchronous sweeps keep converging to the exact solves
    across the switch."""
    m = matrices(golden)[case]()
    r = W.rhs_vector(m["nbrows"] * m["bs"])
    p = make_prec(m)
    capi.set_tuning("compactafter=-1")
    try:
        p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)
        gf = p.get_iluvals()
        want = O.ilu0_apply(m, gf, r, 2, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO)
        copies, first = [], None
        for k in range(threshold + 3):
            copies.append(p.memory_stats()["derived_copies"])
            z = p.ilu0_apply(r, 2, mode=capi.JACOBI_SYNC)
            assert rel(z, want) < TOL_SYNC
            first = z if first is None else first
            assert np.array_equal(z, first)
        assert copies[:threshold + 1] == [0] * (threshold + 1) and copies[threshold + 1:] == [2, 2]
        # a new factorisation starts the count again; the asynchronous sweeps across the threshold
        p.ilu0_factorize(-1)
        exact = O.ilu0_apply(m, p.get_iluvals(), r, 1, mode=O.GS_SERIAL)
        nlev = int(W.dependency_levels(m).max()) + 1
        for k in range(threshold + 2):
            z = p.ilu0_apply(r, nlev + 2, mode=capi.ASYNC)
            assert rel(z, exact) < TOL_EXACT, k
        assert p.memory_stats()["derived_copies"] == 2
    finally:
        capi.set_tuning("compactafter=" + os.environ.get("BLASTED_HIP_COMPACT_AFTER", "-1"))
        p.close()


def test_warm_start_init_none(golden):
    m = matrices(golden)["poisson16_csr"]()
    p = make_prec(m)
    p.ilu0_factorize(2, init=capi.INIT_F_ORIGINAL, mode=capi.JACOBI_SYNC)
    p.ilu0_factorize(3, init=capi.INIT_F_NONE, mode=capi.JACOBI_SYNC)
    want = O.ilu0_factorize(m, None, 5, mode=O.JACOBI_SYNC, init=O.INIT_F_ORIGINAL)["iluvals"]
    assert rel(p.get_iluvals(), want) < TOL_SYNC
    p.close()


# ---------------------------------------------------------------------------- ILU(0) apply (the metric's kernel)

def exact_factor(m, usescale=False):
    return O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, init=O.INIT_F_ORIGINAL, usescale=usescale)


def load_exact_factor(p, m, usescale=False):
    """Factor on the GPU until converged; the factor is then checked and used for apply tests."""
    p.ilu0_factorize(90, init=capi.INIT_F_ORIGINAL, usescale=usescale, mode=capi.ASYNC)


@pytest.mark.parametrize("case", ALL)
@pytest.mark.parametrize("init", [capi.INIT_A_ZERO, capi.INIT_A_JACOBI])
def test_ilu_apply_sync_sweeps_match_oracle(golden, case, init, factor_storage):
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    f = exact_factor(m)
    p = make_prec(m)
    load_exact_factor(p, m)
    assert rel(p.get_iluvals(), f["iluvals"]) < TOL_EXACT
    gf = p.get_iluvals()  # the oracle applies the GPU's own factor: isolates the apply kernels
    for sweeps in (0, 1, 2, 5):
        z = p.ilu0_apply(r, sweeps, init=init, mode=capi.JACOBI_SYNC)
        want, wy = O.ilu0_apply(m, gf, r, sweeps, mode=O.JACOBI_SYNC, init=init, return_y=True)
        assert rel(z, want) < TOL_SYNC
        assert rel(p.get_ytemp(), wy) < TOL_SYNC or np.abs(wy).max() == 0
    p.close()


@pytest.mark.parametrize("case", ALL)
@pytest.mark.parametrize("usescale", [False, True])
def test_ilu_apply_async_converges_to_exact(golden, case, usescale, factor_storage):
    m = matrices(golden)[case]()
    if usescale and case.startswith("random"):
        pytest.skip("random test matrices may have negative diagonal entries (sqrt)")
    n = m["nbrows"] * m["bs"]
    f = exact_factor(m, usescale)
    p = make_prec(m)
    load_exact_factor(p, m, usescale)
    for r in (np.full(n, 1.1), W.rhs_vector(n)):  # r = 1.1: async_triangular_factors_convergence.cpp:62
        exact = O.ilu0_apply(m, f["iluvals"], r, 1, mode=O.GS_SERIAL, scale=f["scale"])
        nsw = 60
        for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
            z = p.ilu0_apply(r, nsw, init=init, mode=capi.ASYNC)
            assert rel(z, exact) < TOL_EXACT
    p.close()


def test_ilu_apply_device_pointers_and_errors(golden):
    import torch
    m = matrices(golden)["poisson16_bs4"]()
    n = m["nbrows"] * 4
    r = W.rhs_vector(n)
    p = make_prec(m)
    with pytest.raises(capi.BlastedHipError) as ei:  # apply before factorize
        p.ilu0_apply(r, 1)
    assert ei.value.code == capi.ESTATE
    load_exact_factor(p, m)
    with pytest.raises(capi.BlastedHipError) as ei:  # src/solverops_ilu0.cpp:125-126
        p.ilu0_apply(r, 1, init=capi.INIT_A_NONE)
    assert ei.value.code == capi.EINVAL and "Invalid init type" in str(ei.value)
    zh = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
    rd = torch.from_numpy(r).cuda()
    zd = p.ilu0_apply(rd, 3, mode=capi.JACOBI_SYNC)
    torch.cuda.synchronize()
    assert np.array_equal(zd.cpu().numpy(), zh)
    p.close()


# ---------------------------------------------------------------------------- Jacobi / SGS / relaxation

@pytest.mark.parametrize("case", ALL)
def test_jacobi_and_sgs_match_oracle(golden, case):
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    d = O.jacobi_compute(m)
    p = make_prec(m)
    p.jacobi_compute()
    assert rel(p.get_dblocks(), d) < 1e-11
    gd = p.get_dblocks()
    assert rel(p.jacobi_apply(r), O.jacobi_apply(m, gd, r)) < 1e-13
    for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
        for sweeps in (0, 1, 3):
            z = p.sgs_apply(r, sweeps, init=init, mode=capi.JACOBI_SYNC)
            want = O.sgs_apply(m, gd, r, sweeps, mode=O.JACOBI_SYNC, init=init)
            assert rel(z, want) < TOL_SYNC or np.abs(want).max() == 0
    # relaxation from x = 0 and from a non-zero guess
    for x0 in (np.zeros(n), 0.3 * np.sin(np.arange(n))):
        for its in (1, 5):
            x = p.sgs_relax(r, x0.copy(), its, mode=capi.JACOBI_SYNC)
            want = O.sgs_relax(m, gd, r, x0=x0, maxits=its, mode=O.JACOBI_SYNC)
            assert rel(x, want) < 1e-11
    p.close()


@pytest.mark.parametrize("case", ["poisson16_csr", "random_csr", "poisson5_csr"])
@pytest.mark.parametrize("lanes", ["auto", 0, 1, 2, 3, 4])
def test_scalar_lane_per_row_sweeps_match_oracle(golden, case, lanes):
    """kernels_sweep1.hip (one lane per scalar row, 1 or 2 rows per lane in flight) and the general four-lanes-per-row
    kernel (scalarlane=0) on every operator of the scalar path, synchronous sweeps against the oracle (P2), and the
    in-place sweeps against the exact solve (P3).  random_csr has rows of 1..14 entries (empty lower / upper parts,
    the remainder loop behind the four / eight straight-line entries) and 1001 rows (a ragged last chunk)."""
    m = W.poisson3d(5, 1) if case == "poisson5_csr" else matrices(golden)[case]()
    n = m["nbrows"]
    r = W.rhs_vector(n)
    x1 = 0.3 * np.sin(np.arange(n))
    capi.set_tuning("scalarlane=%s" % lanes)
    p = make_prec(m)
    try:
        assert rel(p.spmv(x1), O.spmv(m, x1)) < 1e-13
        assert rel(p.gemv3(-1.5, x1, 0.25, r), O.gemv3(m, -1.5, x1, 0.25, r)) < 1e-13
        load_exact_factor(p, m)
        gf = p.get_iluvals()
        for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
            for sweeps in (1, 3):
                z = p.ilu0_apply(r, sweeps, init=init, mode=capi.JACOBI_SYNC)
                assert rel(z, O.ilu0_apply(m, gf, r, sweeps, mode=O.JACOBI_SYNC, init=init)) < TOL_SYNC
        exact = O.ilu0_apply(m, gf, r, 1, mode=O.GS_SERIAL)
        assert rel(p.ilu0_apply(r, 60, mode=capi.ASYNC), exact) < TOL_EXACT
        assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), exact) < TOL_EXACT
        p.jacobi_compute()
        gd = p.get_dblocks()
        assert rel(p.jacobi_apply(r), O.jacobi_apply(m, gd, r)) < 1e-13
        for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
            z = p.sgs_apply(r, 3, init=init, mode=capi.JACOBI_SYNC)
            assert rel(z, O.sgs_apply(m, gd, r, 3, mode=O.JACOBI_SYNC, init=init)) < TOL_SYNC
        assert rel(p.sgs_apply(r, 80, mode=capi.ASYNC), O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL, init=O.INIT_A_ZERO)) < TOL_EXACT
        x = p.sgs_relax(r, x1.copy(), 4, mode=capi.JACOBI_SYNC)
        assert rel(x, O.sgs_relax(m, gd, r, x0=x1, maxits=4, mode=O.JACOBI_SYNC)) < 1e-11
        x = p.gs_relax(r, x1.copy(), 3, mode=capi.JACOBI_SYNC)
        assert rel(x, O.gs_relax(m, gd, r, x0=x1, nsweeps=3, mode=O.JACOBI_SYNC)) < 1e-11
    finally:
        capi.set_tuning("scalarlane=auto")
        p.close()


@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_csr", "poisson16_bs4", "poisson12_bs5", "random_bs4"])
def test_sgs_async_converges_to_exact(golden, case):
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    exact = O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL, init=O.INIT_A_ZERO)
    nsw = 60 if case.startswith("poisson") else 80
    z = p.sgs_apply(r, nsw, init=capi.INIT_A_ZERO, mode=capi.ASYNC)
    assert rel(z, exact) < TOL_EXACT
    # INIT_A_NONE: z is the initial guess of the backward sweeps
    z2 = p.sgs_apply(r, 2, init=capi.INIT_A_NONE, mode=capi.JACOBI_SYNC, out=z.copy())
    assert rel(z2, exact) < TOL_EXACT
    p.close()


@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "2dcyl1_csr", "poisson16_bs4", "poisson12_bs5",
                                  "poisson9_bs8", "random_bs4", "random_bs5", "random_csr"])
def test_sgs_apply_forward_half_is_exact_at_low_sweep_counts(golden, case):
    """Q3 (src/solverops_sgs.cpp:62-66, kernels_sgs.hpp:127): the reference's forward loop is outside the
    parallel region, so ytemp = (D+L)^-1 r is the exact serial solve at every sweep and thread count; only
    the backward sweeps are asynchronous.  The product (ASYNC) mode must give that ytemp at the reference's
    own low sweep counts, and a z no farther from the exact backward solve than synchronous Jacobi backward
    sweeps from the same y (the serial Gauss-Seidel sweep is the exact solve, distance 0)."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    ze, ye = O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL, init=O.INIT_A_ZERO, return_y=True)
    for nsw in (1, 3):
        for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
            z = p.sgs_apply(r, nsw, init=init, mode=capi.ASYNC)
            assert rel(p.get_ytemp(), ye) < 1e-12
            # the oracle's threaded mode keeps the forward half serial as well
            zo, yo = O.sgs_apply(m, gd, r, nsw, mode=O.ASYNC_OMP, init=init, return_y=True)
            assert rel(yo, ye) < 1e-13
            # exact forward half + nsw synchronous Jacobi backward sweeps (ye is a fixed point of the forward sweep)
            z0 = np.zeros(n) if init == capi.INIT_A_ZERO else ye
            zj = O.sgs_apply(m, gd, r, nsw, mode=O.JACOBI_SYNC, init=O.INIT_A_NONE, y0=ye, z0=z0)
            dj, dg = np.linalg.norm(zj - ze), np.linalg.norm(z - ze)
            assert dg <= 1.05 * dj + 1e-12 * np.linalg.norm(ze)
    # the all-asynchronous form of round 1 stays reachable as a tuning variant and differs at one sweep
    capi.set_tuning("sgsfwd=async")
    try:
        p.sgs_apply(r, 1, init=capi.INIT_A_ZERO, mode=capi.ASYNC)
        assert rel(p.get_ytemp(), ye) > 1e-6
    finally:
        capi.set_tuning("sgsfwd=exact")
    p.close()


@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_csr", "poisson12_bs5", "poisson9_bs8", "random_bs4", "random_csr"])
def test_deterministic_mode(golden, case):
    """BLASTED_HIP_DETERMINISTIC, the host layer's default: a fixed operator.  ILU apply = the synchronous
    sweeps of JACOBI_SYNC; SGS apply = exact forward half + synchronous backward sweeps; bit-identical from
    call to call."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)
    f = p.get_iluvals()
    for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
        for nsw in (1, 3):
            z = p.ilu0_apply(r, nsw, init=init, mode=capi.DETERMINISTIC)
            assert np.array_equal(z, p.ilu0_apply(r, nsw, init=init, mode=capi.JACOBI_SYNC))
            assert rel(z, O.ilu0_apply(m, f, r, nsw, mode=O.JACOBI_SYNC, init=init)) < TOL_SYNC
    p.jacobi_compute()
    gd = p.get_dblocks()
    ze, ye = O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL, return_y=True)
    for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
        for nsw in (1, 2, 3):
            z = p.sgs_apply(r, nsw, init=init, mode=capi.DETERMINISTIC)
            assert rel(p.get_ytemp(), ye) < 1e-12
            z0 = np.zeros(n) if init == capi.INIT_A_ZERO else ye
            want = O.sgs_apply(m, gd, r, nsw, mode=O.JACOBI_SYNC, init=O.INIT_A_NONE, y0=ye, z0=z0)
            assert rel(z, want) < 1e-12
            assert np.array_equal(z, p.sgs_apply(r, nsw, init=init, mode=capi.DETERMINISTIC))
    # INIT_A_NONE: z is the initial guess of the backward sweeps
    z0 = 0.5 * np.cos(np.arange(n))
    z = p.sgs_apply(r, 2, init=capi.INIT_A_NONE, mode=capi.DETERMINISTIC, out=z0.copy())
    assert rel(z, O.sgs_apply(m, gd, r, 2, mode=O.JACOBI_SYNC, init=O.INIT_A_NONE, y0=ye, z0=z0)) < 1e-12
    p.close()


def test_sgs_relax_async_reduces_residual(golden):
    """config 3: async block-SGS relaxation, 5 sweeps; asynchronous result lies between the
    synchronous-Jacobi and the serial Gauss-Seidel iterates in residual."""
    m = W.poisson3d(18, 4)
    n = m["nbrows"] * 4
    b = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    x = p.sgs_relax(b, np.zeros(n), 5, mode=capi.ASYNC)
    res = lambda v: np.linalg.norm(b - O.spmv(m, v)) / np.linalg.norm(b)
    xj = O.sgs_relax(m, gd, b, maxits=5, mode=O.JACOBI_SYNC)
    xs = O.sgs_relax(m, gd, b, maxits=5, mode=O.GS_SERIAL)
    assert res(x) < 1.0
    assert res(x) <= res(xj) * 1.05 and res(x) >= res(xs) * 0.5
    p.close()


# ---------------------------------------------------------------------------- solve-level known answers (G7)

@pytest.mark.parametrize("mat,bs,rowmajor,prec", [
    ("msc00726", 1, False, "sgs"), ("msc00726", 1, False, "ilu0"),
    ("2dcyl1", 1, False, "ilu0"), ("2dcyl1", 4, True, "sgs"), ("2dcyl1", 4, True, "ilu0"),
    ("2dcyl1", 4, False, "jacobi"), ("2dcyl1", 4, False, "sgs"), ("2dcyl1", 4, False, "ilu0"),
])
def test_solve_known_answer_on_gpu(golden, mat, bs, rowmajor, prec):
    """tests/CMakeLists.txt:34-173 with the preconditioner and the SpMV on the GPU (ThreadedBSR4ILU0
    style: 10 build sweeps, 15 apply sweeps, :165-173)."""
    m = mtxio.read_mtx_bsr(G(golden, mat + ".mtx"), bs, rowmajor)
    b = mtxio.read_mtx_dense(G(golden, mat + "_b.mtx"))
    xk = mtxio.read_mtx_dense(G(golden, mat + "_x.mtx"))
    p = make_prec(m)
    if prec == "jacobi":
        p.jacobi_compute()
        P = lambda v: p.jacobi_apply(v)
    elif prec == "sgs":
        p.jacobi_compute()
        P = lambda v: p.sgs_apply(v, 15, init=capi.INIT_A_ZERO)
    else:
        p.ilu0_factorize(10 if bs > 1 else 60, init=capi.INIT_F_ORIGINAL)
        P = lambda v: p.ilu0_apply(v, 15, init=capi.INIT_A_ZERO)
    x, its, relres = bicgstab(lambda v: p.spmv(v), P, b, 1e-10, 400)
    assert relres < 1e-10
    x, its, relres = bicgstab(lambda v: p.spmv(v), P, b, 1e-14, 800)
    floor = 2e-9 if mat == "msc00726" else 0.0
    assert np.linalg.norm(x - xk) < max(1e-8, floor)
    p.close()


# ---------------------------------------------------------------------------- edge cases

def test_tiny_and_ragged_matrices(factor_storage):
    # 1 block-row; rows with empty lower / upper parts; row count not a multiple of the rows per workgroup
    for nb, bs, rm in ((1, 4, 0), (2, 5, 0), (3, 1, 0), (17, 4, 0), (65, 8, 0), (5, 3, 0), (129, 7, 0), (33, 3, 0),
                       (131, 5, 0), (9, 2, 0), (150, 4, 1), (67, 8, 1), (40, 5, 1)):
        m = W.random_bsr(nb, bs, avg_offdiag=2 if not rm else 9, seed=nb, rowmajor=bool(rm))
        n = nb * bs
        r = W.rhs_vector(n)
        p = make_prec(m)
        assert rel(p.spmv(r), O.spmv(m, r)) < 1e-13
        p.ilu0_factorize(nb + 2, mode=capi.ASYNC)
        f = O.ilu0_factorize(m, None, 1)["iluvals"]
        assert rel(p.get_iluvals(), f) < TOL_EXACT
        z = p.ilu0_apply(r, nb + 2)
        assert rel(z, O.ilu0_apply(m, f, r, 1)) < TOL_EXACT
        assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), O.ilu0_apply(m, f, r, 1)) < TOL_EXACT
        p.ilu0_factorize(-1)
        assert rel(p.get_iluvals(), f) < TOL_EXACT
        p.jacobi_compute()
        assert rel(p.sgs_apply(r, 2, mode=capi.JACOBI_SYNC),
                   O.sgs_apply(m, p.get_dblocks(), r, 2, mode=O.JACOBI_SYNC)) < TOL_SYNC
        p.close()


def test_medium_poisson_64_against_oracle():
    """64^3 bs=4 (262144 block-rows): HIP sync sweeps == oracle sync sweeps at the bench's sweep count."""
    m = W.poisson3d(66, 4, grid="uniform")
    n = m["nbrows"] * 4
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)
    want = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC)["iluvals"]
    gf = p.get_iluvals()
    assert rel(gf, want) < TOL_SYNC
    z = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
    assert rel(z, O.ilu0_apply(m, gf, r, 3, mode=O.JACOBI_SYNC)) < TOL_SYNC
    # async: 3 sweeps lie between synchronous Jacobi and exact in error
    exact = O.ilu0_apply(m, gf, r, 1, mode=O.GS_SERIAL)
    za = p.ilu0_apply(r, 3, mode=capi.ASYNC)
    assert np.abs(za - exact).max() <= np.abs(z - exact).max() * 1.0001
    p.close()


@pytest.mark.parametrize("interleave", ["0", "1"])
def test_async_sweeps_are_never_worse_than_synchronous_ones_64(interleave):
    """SURVEY 8(d) tier P4 as a gate: on Poisson 64^3 bs=4 (the bench generator) the distance of the HIP ASYNC
    application to the exact triangular solves after s+s sweeps, s in {1, 3, 5}, is at most that of the oracle's
    synchronous Jacobi sweeps at the same count (a chaotic in-place sweep sees at least what the previous sweep
    left), in 2-norm and for the intermediate y as well; and it decreases with s.  Both row orders of the sweep."""
    m = W.poisson3d(66, 4, grid="uniform")
    r = W.rhs_vector(m["nbrows"] * 4)
    fe = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    ze = O.ilu0_apply(m, fe, r, 1, mode=O.GS_SERIAL)
    p = make_prec(m)
    p.ilu0_factorize(-1)
    capi.set_tuning("interleave=" + interleave)
    try:
        last = np.inf
        for s in (1, 3, 5):
            zs = O.ilu0_apply(m, fe, r, s, mode=O.JACOBI_SYNC)
            es = np.linalg.norm(zs - ze) / np.linalg.norm(ze)
            worst = 0.0
            for _ in range(3):   # chaotic: not the same result twice
                za = p.ilu0_apply(r, s, mode=capi.ASYNC)
                worst = max(worst, np.linalg.norm(za - ze) / np.linalg.norm(ze))
            assert worst <= es * (1 + 1e-9), (s, worst, es)
            assert worst < last
            last = worst
    finally:
        capi.set_tuning("interleave=0")  # the default
    p.close()


@pytest.mark.parametrize("case", ["poisson16_bs4", "2dcyl1_bs4_col", "random_bs4"])
def test_in_place_sweep_variants_reach_the_exact_solves(golden, case):
    """Every form of the in-place bs = 4 triangular sweep -- results stored step by step (rounds 1-2) or once per
    workgroup with 1, 2 (default) or 4 row steps of a wave in flight; natural row order, interleaved with the wave's
    registers, interleaved through memory -- is the same fixed-point iteration: run past the dependency depth it gives
    the exact triangular solves (<= 1e-10), and synchronous sweeps are bit-identical whatever the setting."""
    m = matrices(golden)[case]()
    r = W.rhs_vector(m["nbrows"] * m["bs"])
    p = make_prec(m)
    p.ilu0_factorize(-1)
    f = p.get_iluvals()
    ze = O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)
    nsw = int(W.dependency_levels(m).max()) + 3
    p.jacobi_compute()
    zsgs = O.sgs_apply(m, O.jacobi_compute(m), r, 1, mode=O.GS_SERIAL)
    sync_ref = None
    try:
        for spec in ("latestore=0", "latestore=1", "latestore=2", "latestore=4", "interleave=1", "interleave=2"):
            capi.set_tuning(spec)
            assert rel(p.ilu0_apply(r, nsw, mode=capi.ASYNC), ze) < TOL_EXACT, spec
            assert rel(p.sgs_apply(r, nsw, mode=capi.ASYNC), zsgs) < TOL_EXACT, spec
            zs = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
            sync_ref = zs if sync_ref is None else sync_ref
            assert np.array_equal(zs, sync_ref), spec
            if spec.startswith("interleave"):
                capi.set_tuning("interleave=0")
    finally:
        capi.set_tuning("interleave=0")
        capi.set_tuning("latestore=2")
    p.close()


def test_interleaved_sweep_forwards_the_finished_row_in_registers():
    """The interleaved row order with the wave's registers (kernels_sweepw.hip, IW), pinned deterministically: on 32
    block-rows ONE wave owns the whole sweep -- lane group g takes rows 4g .. 4g+3 in its four steps -- so an in-place
    sweep is a fixed sequence: a row sees the NEW value of the neighbour its own lanes finished in the step before
    (rows 4g+1, 4g+2, 4g+3 in an ascending sweep) and the old one otherwise (rows 4g: their predecessor belongs to the
    next lane group's last step).  Block-tridiagonal bs = 4 matrix, SGS application with asynchronous forward and
    backward sweeps (one each, from zero): y and z against that sequence evaluated in numpy."""
    rng = np.random.default_rng(5)
    nb, bs = 32, 4
    rp = [0]
    ci = []
    for i in range(nb):
        ci += [j for j in (i - 1, i, i + 1) if 0 <= j < nb]
        rp.append(len(ci))
    ci = np.array(ci, dtype=np.int32)
    rp = np.array(rp, dtype=np.int32)
    rowof = np.repeat(np.arange(nb), rp[1:] - rp[:-1])
    blocks = rng.uniform(-0.5, 0.5, (ci.size, bs, bs))          # [block][r][c]
    blocks[ci == rowof] += 3.0 * np.eye(bs)
    dg = np.array([int(np.where((rowof == i) & (ci == i))[0][0]) for i in range(nb)], dtype=np.int32)
    m = dict(nbrows=nb, nnzb=int(ci.size), bs=bs, rowmajor=False, browptr=rp, bcolind=ci, diagind=dg,
             vals=np.ascontiguousarray(np.transpose(blocks, (0, 2, 1))).reshape(-1))   # column-major blocks
    r = W.rhs_vector(nb * bs).reshape(nb, bs)
    Dinv = [np.linalg.inv(blocks[dg[i]]) for i in range(nb)]
    blk = {(int(rowof[k]), int(ci[k])): blocks[k] for k in range(ci.size)}
    # forward sweep, ascending, from y = 0: step s takes rows 4g + s
    y = np.zeros((nb, bs))
    for step in range(4):
        new = {}
        for g in range(8):
            i = 4 * g + step
            acc = blk[(i, i - 1)] @ y[i - 1] if i > 0 else np.zeros(bs)   # y[i-1]: new iff step > 0 (same lane group)
            new[i] = Dinv[i] @ (r[i] - acc)
        for i, v in new.items():
            y[i] = v
    # backward sweep, descending, from z = 0: step s takes rows 31 - 4g - s
    z = np.zeros((nb, bs))
    for step in range(4):
        new = {}
        for g in range(8):
            i = nb - 1 - (4 * g + step)
            acc = blk[(i, i + 1)] @ z[i + 1] if i + 1 < nb else np.zeros(bs)
            new[i] = y[i] - Dinv[i] @ acc
        for i, v in new.items():
            z[i] = v
    p = make_prec(m)
    p.jacobi_compute()
    capi.set_tuning("sgsfwd=async")
    capi.set_tuning("interleave=1")
    try:
        got_z = p.sgs_apply(r.reshape(-1), 1, init=capi.INIT_A_ZERO, mode=capi.ASYNC)
        got_y = p.get_ytemp()
        # the natural order is NOT this sequence (rows 0..7 of a step are Jacobi among themselves, four waves race)
        capi.set_tuning("interleave=0")
        nat_y = None
        p.sgs_apply(r.reshape(-1), 1, init=capi.INIT_A_ZERO, mode=capi.ASYNC)
        nat_y = p.get_ytemp()
    finally:
        capi.set_tuning("sgsfwd=exact")
        capi.set_tuning("interleave=0")
    assert rel(got_y, y.reshape(-1)) < 1e-13
    assert rel(got_z, z.reshape(-1)) < 1e-13
    assert rel(nat_y, y.reshape(-1)) > 1e-6
    p.close()


def test_config0_scalar_poisson_64_chebyshev():
    """BASELINE config 0: tests/poisson3d-fd 64^3 scalar CSR on the reference's default Chebyshev grid
    (input/poisson.control), async ILU(0) with 3 build and 3 apply sweeps."""
    m = W.poisson3d(66, 1, grid="chebyshev")
    assert m["nbrows"] == 64 ** 3
    r = W.rhs_vector(m["nbrows"])
    p = make_prec(m)
    p.ilu0_factorize(3, mode=capi.JACOBI_SYNC)
    gf = p.get_iluvals()
    assert rel(gf, O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC)["iluvals"]) < TOL_SYNC
    z = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
    assert rel(z, O.ilu0_apply(m, gf, r, 3, mode=O.JACOBI_SYNC)) < TOL_SYNC
    # the asynchronous forms: finite, and with enough sweeps equal to the serial (exact) result
    p.ilu0_factorize(3, mode=capi.ASYNC)
    assert np.isfinite(p.get_iluvals()).all()
    nlev = 3 * 64 - 2
    p.ilu0_factorize(nlev + 2, mode=capi.ASYNC)
    fe = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    assert rel(p.get_iluvals(), fe) < TOL_EXACT
    ze = O.ilu0_apply(m, fe, r, 1, mode=O.GS_SERIAL)
    assert rel(p.ilu0_apply(r, nlev + 2, mode=capi.ASYNC), ze) < TOL_EXACT
    assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), ze) < TOL_EXACT
    p.close()


# ---------------------------------------------------------------------------- chaotic relaxation (gs)

@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_csr", "poisson12_bs5", "poisson9_bs8", "random_bs4"])
def test_gs_relaxation_matches_oracle(golden, case):
    """The `gs` type, src/relaxation_chaotic.cpp:21-70: forward passes only, x is guess and result."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    b = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    for x0 in (np.zeros(n), 0.2 * np.cos(np.arange(n))):
        x = p.gs_relax(b, x0.copy(), 3, mode=capi.JACOBI_SYNC)
        assert rel(x, O.gs_relax(m, gd, b, x0=x0, nsweeps=3, mode=O.JACOBI_SYNC)) < 1e-11
    # async forward sweeps converge to one serial forward Gauss-Seidel pass repeated: compare fixed points
    xs = O.gs_relax(m, gd, b, nsweeps=3000, mode=O.GS_SERIAL)
    xa = p.gs_relax(b, np.zeros(n), 6000, mode=capi.ASYNC)
    if np.all(np.isfinite(xs)) and np.abs(xs).max() < 1e6:
        assert rel(xa, xs) < 1e-8
    p.close()


# ---------------------------------------------------------------------------- sequential variants

@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "msc_csr", "poisson16_bs4", "poisson12_bs5", "random_bs4"])
def test_sequential_variants_are_exact(golden, case):
    """seqilu0 / sfilu0 / sapilu0 (threadedfactor / threadedapply = false, or -blasted_async_sweeps -1):
    a negative sweep count sweeps until stationary = one in-order serial pass of the reference."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    exact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, init=O.INIT_F_ORIGINAL)["iluvals"]
    p = make_prec(m)
    p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL)
    gf = p.get_iluvals()
    assert rel(gf, exact) < TOL_EXACT
    # bitwise fixed point: more sweeps do not change a single bit
    p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL)
    assert np.array_equal(p.get_iluvals(), gf)
    z = p.ilu0_apply(r, -1, init=capi.INIT_A_ZERO)
    assert rel(z, O.ilu0_apply(m, gf, r, 1, mode=O.GS_SERIAL)) < TOL_EXACT
    z2 = p.ilu0_apply(r, -1, init=capi.INIT_A_JACOBI)
    assert np.array_equal(z, z2)  # the stationary point does not depend on the initial guess
    p.jacobi_compute()
    zs = p.sgs_apply(r, -1, init=capi.INIT_A_ZERO)
    assert rel(zs, O.sgs_apply(m, p.get_dblocks(), r, 1, mode=O.GS_SERIAL)) < TOL_EXACT
    p.close()


@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "msc_csr", "poisson12_bs5", "poisson9_bs8", "random_bs4"])
def test_exact_factorisation_ignores_the_stored_factor(golden, case):
    """The exact in-order factorisation skips the initialisation pass (every entry is written before it is
    read): with the factor storage poisoned by NaNs beforehand the result is bit-identical, whatever the
    initialisation type, and equal to the pass that did run the initialisation (compute_info=True)."""
    import ctypes as C
    m = matrices(golden)[case]()
    p = make_prec(m)
    p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL)
    f0 = p.get_iluvals()
    assert np.all(np.isfinite(f0))
    poison = np.full(f0.size, np.nan)
    for init in (capi.INIT_F_ORIGINAL, capi.INIT_F_SGS, capi.INIT_F_ZERO):
        rc = capi.lib().blasted_hip_buffer_upload(C.c_void_p(p.iluvals_device_ptr()), poison.ctypes.data_as(C.c_void_p),
                                                  C.c_ulong(poison.nbytes))
        assert rc == 0
        assert np.all(np.isnan(p.get_iluvals()))
        p.ilu0_factorize(-1, init=init)
        assert np.array_equal(p.get_iluvals(), f0)
    info = p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL, compute_info=True)
    # (with the remainder asked for the factorisation runs level by level on the un-inverted factor; without,
    # as one launch -- at bs = 4 of the matrix-core kernel, whose products are summed in the same order)
    assert np.array_equal(p.get_iluvals(), f0) and np.all(np.isfinite(info))
    p.close()


# ---------------------------------------------------------------------------- Jacobi relaxation

@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "msc_csr", "poisson12_bs5", "poisson9_bs8", "random_bs4"])
def test_jacobi_relaxation_matches_oracle(golden, case):
    """BJacobiSRPreconditioner::apply_relax, src/solverops_jacobi.cpp:66-119: synchronous steps, with and
    without the convergence test on the step difference (same stopping step as the oracle)."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    b = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    x0 = 0.1 * np.sin(np.arange(n))
    x = x0.copy()
    assert p.jacobi_relax(b, x, 4) == 4
    want, steps = O.jacobi_relax(m, gd, b, x0=x0, maxits=4)
    assert steps == 4
    if np.all(np.isfinite(want)) and np.abs(want).max() < 1e8:
        assert rel(x, want) < 1e-11
    # convergence test: relative tolerance reached (or divergence detected) at the same step
    for rtol, dtol in ((0.3, 1e300), (0.0, 1.5)):
        x = np.zeros(n)
        got_steps = p.jacobi_relax(b, x, 200, check_tol=True, rtol=rtol, atol=0.0, dtol=dtol)
        want, steps = O.jacobi_relax(m, gd, b, maxits=200, ctol=True, rtol=rtol, atol=0.0, dtol=dtol)
        assert got_steps == steps
        if np.all(np.isfinite(want)) and np.abs(want).max() < 1e8:
            assert rel(x, want) < 1e-9
    p.close()


# ---------------------------------------------------------------------------- iteration counts (compare_its)

@pytest.mark.parametrize("prec", ["sgs_1_8", "ilu0_4_8", "ilu0_4_8_scaled"])
def test_async_preconditioning_iteration_counts(golden, prec):
    """The reference's `compare_its` tests (tests/CMakeLists.txt:374-401, input/asyncpreconditioning.perc):
    Richardson on 2dcyl1 (bs=4) to rtol 1e-5 within 200 iterations, preconditioned by asynchronous SGS with
    (1,8) sweeps / ILU(0) with (4,8) sweeps, needs the iteration count of the exact SGS / ILU(0) to 1 %."""
    from krylov import richardson
    m = matrices(golden)["2dcyl1_bs4_col"]()
    n = m["nbrows"] * 4
    b = mtxio_vec(golden, "2dcyl1_b.mtx")
    p = make_prec(m)
    A = lambda v: p.spmv(v)
    if prec.startswith("sgs"):
        p.jacobi_compute()
        M_async = lambda v: p.sgs_apply(v, 8, init=capi.INIT_A_ZERO, mode=capi.ASYNC)
        M_exact = lambda v: p.sgs_apply(v, 1, mode=capi.LEVEL)
        x, its_async, rel_a = richardson(A, M_async, b, 1e-5, 200)
        x, its_exact, rel_e = richardson(A, M_exact, b, 1e-5, 200)
    else:
        sc = prec.endswith("scaled")
        p.ilu0_factorize(4, init=capi.INIT_F_ORIGINAL, usescale=sc, mode=capi.ASYNC)
        x, its_async, rel_a = richardson(A, lambda v: p.ilu0_apply(v, 8, mode=capi.ASYNC), b, 1e-5, 200)
        p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL, usescale=sc)
        x, its_exact, rel_e = richardson(A, lambda v: p.ilu0_apply(v, 1, mode=capi.LEVEL), b, 1e-5, 200)
    assert rel_a < 1e-5 and rel_e < 1e-5 and its_exact < 200
    assert abs(its_async - its_exact) <= max(1, round(0.01 * its_exact)), (its_async, its_exact)
    p.close()


def test_async_relaxation_convergence_and_bound(golden):
    """The reference's relaxation tests on 2dcyl1 bs=4 (tests/CMakeLists.txt:357-372, input/
    asyncrelaxation.perc): outer Richardson to rtol 1e-5 within 200 iterations whose inner solve is 10
    relaxation iterations from a zero guess.  `convergence`: the asynchronous `gs` relaxation gets there;
    `upper_bound_its`: asynchronous SGS relaxation needs fewer outer iterations than block-Jacobi."""
    from krylov import richardson
    m = matrices(golden)["2dcyl1_bs4_col"]()
    n = m["nbrows"] * 4
    b = mtxio_vec(golden, "2dcyl1_b.mtx")
    p = make_prec(m)
    p.jacobi_compute()
    A = lambda v: p.spmv(v)

    def inner(kind):
        def M(v):
            x = np.zeros(n)
            if kind == "jacobi":
                p.jacobi_relax(v, x, 10)
            elif kind == "gs":
                p.gs_relax(v, x, 10, mode=capi.ASYNC)
            else:
                p.sgs_relax(v, x, 10, mode=capi.ASYNC)
            return x
        return M
    res = {k: richardson(A, inner(k), b, 1e-5, 200) for k in ("jacobi", "gs", "sgs")}
    assert res["gs"][2] < 1e-5 and res["gs"][1] < 200          # convergence
    assert res["sgs"][2] < 1e-5 and res["sgs"][1] < res["jacobi"][1]   # upper_bound_its
    p.close()


# ---------------------------------------------------------------------------- small problems: no fills

@pytest.mark.parametrize("case", ["poisson16_csr", "msc_csr", "2dcyl1_csr", "random_csr", "poisson16_bs4", "poisson12_bs5", "2dcyl1_bs4_col"])
@pytest.mark.parametrize("usescale", [False, True])
def test_small_application_without_fills_converges_to_exact(golden, case, usescale):
    """VERDICT r03 item 6 (the part that was kept): a cache-resident problem's asynchronous ILU(0) application runs as
    2 s launches instead of 2 s + 2 -- the first sweep of each triangle reads an operator-owned vector of zeros instead of
    a freshly zeroed iterate -- and reaches the oracle's exact solves like the form with the fills (`smallapply=0`), from
    both initial guesses, with and without scaling, for device and host vectors.  Scalar matrices (`smallapply=2`, the
    default) save one more launch: the last lower sweep stores z1 = D^-1 y beside y, the upper sweeps start from it."""
    import torch
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.ilu0_factorize(-1, usescale=usescale)
    f = p.get_iluvals()
    scale = p.get_scale() if usescale else None
    exact, ey = O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL, scale=scale, return_y=True)
    nlev = int(W.dependency_levels(m).max()) + 1
    s = nlev + 2
    rd = torch.from_numpy(r).cuda()
    try:
        for small in ("2", "1", "0"):
            capi.set_tuning("smallapply=" + small)
            for init in (capi.INIT_A_ZERO, capi.INIT_A_JACOBI):
                z = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")   # whatever the caller's z held
                p.set_timing(True)
                p.get_timing()
                p.ilu0_apply(rd, s, init=init, mode=capi.ASYNC, out=z)
                t = p.get_timing()
                p.set_timing(False)
                assert rel(z.cpu().numpy(), exact) < TOL_EXACT, (small, init)
                assert rel(p.get_ytemp(), ey) < TOL_EXACT
                launches = t["lower_launches"] + t["upper_launches"] + t["other_launches"]
                fills = (0 if small != "0" else 1 + (init == capi.INIT_A_ZERO)) + (1 if usescale else 0)  # (+ z := S z)
                # (unscaled: the first lower sweep from zero is y1 = r, so the second one reads r -- one launch less;
                #  scalar, from zero: the first upper sweep rides on the last lower one -- one less again)
                fused = small == "2" and m["bs"] == 1 and init == capi.INIT_A_ZERO
                assert launches == 2 * s + fills - (1 if small != "0" and not usescale else 0) - (1 if fused else 0)
            zh = p.ilu0_apply(r, s, mode=capi.ASYNC)            # host vectors
            assert rel(zh, exact) < TOL_EXACT
    finally:
        capi.set_tuning("smallapply=2")
        p.close()


@pytest.mark.parametrize("case", ["poisson16_csr", "2dcyl1_csr", "random_csr"])
@pytest.mark.parametrize("usescale", [False, True])
def test_small_scalar_application_with_the_fused_first_upper_sweep(golden, case, usescale):
    """The fused form against the unfused one where both are deterministic: ONE sweep per triangle of a small
    application reads the zeros vector, i.e. y = S r, z = S D^-1 y whatever the order of the rows (every product of the
    sweeps is with a zero) -- bit for bit the same with `smallapply=2` (one launch) and `=1` (two), and equal to the
    oracle's synchronous sweep.  (`=0`, in place on freshly zeroed iterates, sees rows the sweep has already written.)"""
    import torch
    m = matrices(golden)[case]()
    n = m["nbrows"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.ilu0_factorize(-1, usescale=usescale)
    f = p.get_iluvals()
    scale = p.get_scale() if usescale else None
    want = O.ilu0_apply(m, f, r, 1, mode=O.JACOBI_SYNC, scale=scale)
    rd = torch.from_numpy(r).cuda()
    got = {}
    try:
        for small in ("2", "1"):
            capi.set_tuning("smallapply=" + small)
            z = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
            p.set_timing(True)
            p.get_timing()
            p.ilu0_apply(rd, 1, init=capi.INIT_A_ZERO, mode=capi.ASYNC, out=z)
            t = p.get_timing()
            p.set_timing(False)
            got[small] = (z.cpu().numpy(), p.get_ytemp(), t["lower_launches"] + t["upper_launches"])
    finally:
        capi.set_tuning("smallapply=2")
        p.close()
    assert got["2"][2] == 1 and got["1"][2] == 2
    assert np.array_equal(got["2"][1], got["1"][1])      # y
    assert np.array_equal(got["2"][0], got["1"][0])      # z
    assert rel(got["2"][0], want) < 1e-14
