"""GPU parity tests of the level-scheduled (exact in-order) sweeps: the reference's `level_sgs` and
`async_level_ilu0` types (src/solverops_levels_sgs.cpp, src/solverops_levels_ilu0.cpp) and the
sequential apply variants.

 * P0: the device level schedule equals the in-order host definition bit for bit, and the reference's
   own computeLevels (src/levelschedule.cpp:13-72), run on the matrix renumbered level by level, returns
   exactly the device's level boundaries.
 * LEVEL-mode sweeps on the ORIGINAL ordering == the oracle's serial pass (rel <= 1e-12) and == the
   reference's level-scheduled operators on the renumbered matrix, mapped back.
"""
import numpy as np
import pytest

import oracle as O
from blasted_amd import capi, workloads as W
from test_gpu_parity import matrices, make_prec, rel

pytestmark = pytest.mark.gpu

TOL = 1e-12
CASES = ["2dcyl1_bs4_col", "2dcyl1_bs4_row", "2dcyl1_csr", "msc_csr", "poisson16_csr", "poisson16_bs4",
         "poisson12_bs5", "poisson9_bs8", "poisson8_bs3", "poisson8_bs7_row", "poisson8_bs2", "random_bs5",
         "random_bs4", "random_csr"]


@pytest.fixture(params=["syncfree", "syncfree_natural", "syncfree_general", "syncfree_inplace", "launch"], autouse=True)
def level_impl(request):
    """Every test runs with each implementation of an exact pass: one launch that polls its dependencies
    -- the streaming kernel on level-ordered copies of the factor with a level-ordered iterate (default, bs
    4/8 column-major) or with natural-order vectors, the general kernel on those copies, the general kernel
    on the factor in place -- and one launch per level."""
    capi.set_tuning("level=" + ("launch" if request.param == "launch" else "syncfree"))
    capi.set_tuning("levelstore=" + ("0" if request.param == "syncfree_inplace" else "1"))
    capi.set_tuning("levelwide=" + ("0" if request.param == "syncfree_general" else "1"))
    capi.set_tuning("levelperm=" + ("0" if request.param == "syncfree_natural" else "1"))
    yield request.param
    capi.set_tuning("level=syncfree")
    capi.set_tuning("levelstore=1")
    capi.set_tuning("levelwide=1")
    capi.set_tuning("levelperm=1")


def check_stats(p, impl):
    st = p.level_stats()
    assert st["syncfree_aborts"] == 0
    assert (st["syncfree_passes"] > 0) == (impl != "launch")


def one_sided(m):
    """Drops the strictly-upper block of some coupled pairs: a structurally NON-symmetric pattern."""
    nb, bs2 = m["nbrows"], m["bs"] ** 2
    rp, ci = m["browptr"], m["bcolind"]
    rows = np.repeat(np.arange(nb), rp[1:] - rp[:-1])
    keep = ~((ci > rows) & ((rows + ci) % 3 == 0))
    nrp = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(nrp, rows[keep] + 1, 1)
    nrp = np.cumsum(nrp).astype(np.int32)
    nci = ci[keep]
    nrows = rows[keep]
    out = dict(m)
    out.update(nnzb=int(keep.sum()), browptr=nrp, bcolind=nci.astype(np.int32),
               diagind=np.nonzero(nrows == nci)[0].astype(np.int32),
               vals=np.ascontiguousarray(m["vals"].reshape(-1, bs2)[keep].reshape(-1)))
    return out


@pytest.mark.parametrize("case", CASES)
def test_level_schedule_bit_exact(golden, case):
    m = matrices(golden)[case]()
    p = make_prec(m)
    lv, rows, ptr = p.get_levels()
    ref = W.dependency_levels(m)
    assert np.array_equal(lv, ref)
    # rows sorted by (level, row), boundaries consistent
    order = np.lexsort((np.arange(m["nbrows"]), ref))
    assert np.array_equal(rows, order.astype(np.int32))
    assert ptr[0] == 0 and ptr[-1] == m["nbrows"] and np.all(np.diff(ptr) > 0)
    assert np.array_equal(np.searchsorted(ref[order], np.arange(ptr.size - 1)), ptr[:-1])
    # the reference's computeLevels on the level-ordered matrix finds the same boundaries
    mp = W.permute_symmetric(m, rows)
    assert np.array_equal(O.compute_levels(mp), ptr)
    p.close()


@pytest.mark.parametrize("shape", ["chain", "one_sided_chain", "poisson"])
def test_level_schedule_deep_graph_takes_the_in_order_pass(shape):
    """A dependency graph about as deep as it is long (a banded, one-dimensional ordering: levels = rows) would
    cost one parallel relaxation pass per level, O(nbrows * nnz) in all; after `levelserial` passes the build
    finishes with ONE in-order pass instead.  Same levels, and the exact solves on it are the serial ones."""
    if shape == "poisson":
        m = W.poisson3d(12, 4)   # 28 levels: with the threshold at 8 the serial pass has to finish a 3-D pattern too
    else:
        n = 3000
        rows = np.repeat(np.arange(n), 3)
        cols = (rows.reshape(n, 3) + np.array([-1, 0, 1])).reshape(-1)
        keep = (cols >= 0) & (cols < n)
        if shape == "one_sided_chain":   # only the upper couplings are stored: the lower ones are implied
            keep &= cols >= rows
        rows, cols = rows[keep], cols[keep]
        rp = np.zeros(n + 1, np.int32)
        np.add.at(rp, rows + 1, 1)
        rp = np.cumsum(rp).astype(np.int32)
        rng = np.random.default_rng(5)
        vals = rng.uniform(-0.3, 0.3, (rows.size, 4))
        vals[rows == cols] = np.array([2.0, 0.1, -0.1, 2.5])
        m = dict(nbrows=n, nnzb=int(rows.size), bs=2, rowmajor=False, browptr=rp, bcolind=cols.astype(np.int32),
                 diagind=np.nonzero(rows == cols)[0].astype(np.int32), vals=np.ascontiguousarray(vals.reshape(-1)))
    ref = W.dependency_levels(m)
    # the build as it runs by default: one dependency-polling launch over the stored lower entries plus one checking
    # pass where the pattern is structurally symmetric; the relaxation passes from there on where it is not
    q = make_prec(m)
    lvq, _, _ = q.get_levels()
    assert np.array_equal(lvq, ref)
    if shape != "one_sided_chain":
        assert q.level_stats()["build_passes"] == 2
    q.close()
    capi.set_tuning("levelserial=8")
    capi.set_tuning("levelfast=0")
    try:
        p = make_prec(m)
        lv, rows_by_level, ptr = p.get_levels()
        st = p.level_stats()
    finally:
        capi.set_tuning("levelserial=4096")
        capi.set_tuning("levelfast=1")
    assert np.array_equal(lv, ref)
    assert st["build_passes"] < 0    # negative: settled by the serial pass after that many parallel ones
    assert st["levels"] == ref.max() + 1
    nvec = m["nbrows"] * m["bs"]
    r = W.rhs_vector(nvec)
    p.ilu0_factorize(-1)
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    assert rel(p.get_iluvals(), f) < TOL
    assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)) < TOL
    p.close()


@pytest.mark.parametrize("case", CASES + ["poisson9_bs8", "random_bs5", "poisson8_bs7_row"])
@pytest.mark.parametrize("usescale", [False, True])
def test_exact_factorisation_single_launch_equals_per_level(golden, case, usescale):
    """The exact factorisation as ONE launch whose rows wait for the rows they depend on (factorsf) is the same
    arithmetic in the same order as one launch per level: bit-identical factors, for every block size and layout
    (factorsf=3 forces the general kernel everywhere; the default (1) and factorsf=2 take the matrix-core kernel
    at bs = 4 where it applies: its products are summed in the order of the general kernels)."""
    if usescale and case.startswith("random"):
        pytest.skip("random test matrices may have negative diagonal entries (sqrt)")
    m = matrices(golden)[case]()
    p = make_prec(m)
    res = {}
    try:
        for k in ("0", "3", "2", "1"):
            capi.set_tuning("factorsf=" + k)
            p.ilu0_factorize(-1, usescale=usescale)
            res[k] = p.get_iluvals()
    finally:
        capi.set_tuning("factorsf=1")
    # (with symmetric scaling too: every form scales an entry as (a s_i) s_j, the reference's two roundings)
    assert np.array_equal(res["0"], res["3"])
    assert np.array_equal(res["0"], res["2"]) and np.array_equal(res["0"], res["1"])
    assert p.level_stats()["syncfree_aborts"] == 0
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=usescale)["iluvals"]
    assert rel(res["2"], f) < TOL
    p.close()


@pytest.mark.parametrize("n,scaled,rowmajor", [(16, False, False), (11, True, False), (24, False, False),
                                               (12, False, True), (9, True, True)])
def test_exact_factorisation_bs4_matrix_core_single_launch(n, scaled, rowmajor):
    """bs = 4 stencil rows: the single-launch exact factorisation is the matrix-core kernel that stages a row's
    operands before it waits (sffactor4_kernel).  It has run (a single-launch pass is counted: at bs = 4 the
    default takes no other single-launch kernel), nobody gave up waiting, and the factor is the serial one."""
    m = W.poisson3d(n, 4, rowmajor=rowmajor)
    p = make_prec(m)
    before = p.level_stats()["syncfree_passes"]
    p.ilu0_factorize(-1, usescale=scaled)
    st = p.level_stats()
    impl_is_launch = st["syncfree_passes"] == before
    capi.set_tuning("factorsf=0")
    try:
        g = p.get_iluvals()
        p.ilu0_factorize(-1, usescale=scaled)
        assert np.array_equal(g, p.get_iluvals())   # the per-level form: same bits
    finally:
        capi.set_tuning("factorsf=1")
    assert st["syncfree_aborts"] == 0
    if not impl_is_launch:   # ("level=launch" keeps one launch per level)
        assert st["syncfree_passes"] == before + 1
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=scaled)["iluvals"]
    assert rel(g, f) < TOL
    p.close()


@pytest.mark.parametrize("n,scaled", [(16, False), (13, True), (40, False)])
def test_exact_factorisation_scalar_single_launch(n, scaled):
    """Scalar stencil rows: the single-launch exact factorisation is the lane-per-row kernel on row plans
    (sffactor1_kernel).  It has run, nobody gave up waiting, the factor is the per-level one bit for bit and the
    serial one within rounding."""
    m = W.poisson3d(n, 1)
    p = make_prec(m)
    before = p.level_stats()["syncfree_passes"]
    p.ilu0_factorize(-1, usescale=scaled)
    st = p.level_stats()
    g = p.get_iluvals()
    capi.set_tuning("factorsf=0")
    try:
        p.ilu0_factorize(-1, usescale=scaled)
        assert np.array_equal(g, p.get_iluvals())
    finally:
        capi.set_tuning("factorsf=1")
    assert st["syncfree_aborts"] == 0
    if st["syncfree_passes"] != before:   # ("level=launch" keeps one launch per level)
        assert st["syncfree_passes"] == before + 1
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL, usescale=scaled)["iluvals"]
    assert rel(g, f) < TOL
    p.close()


def test_exact_factorisation_narrow_levels_keep_the_general_kernels():
    """A one-dimensional ordering (levels = rows): padding every level to a workgroup would multiply the plan array,
    so the plan kernels decline and the factorisation still is the serial one (scalar and bs = 4)."""
    n = 2500
    for bs in (1, 4):
        rows = np.repeat(np.arange(n), 3)
        cols = (rows.reshape(n, 3) + np.array([-1, 0, 1])).reshape(-1)
        keep = (cols >= 0) & (cols < n)
        rows, cols = rows[keep], cols[keep]
        rp = np.zeros(n + 1, np.int32)
        np.add.at(rp, rows + 1, 1)
        rp = np.cumsum(rp).astype(np.int32)
        rng = np.random.default_rng(17)
        vals = rng.uniform(-0.2, 0.2, (rows.size, bs * bs))
        vals[rows == cols] = (2.0 * np.eye(bs) + 0.05).reshape(-1)
        m = dict(nbrows=n, nnzb=int(rows.size), bs=bs, rowmajor=False, browptr=rp, bcolind=cols.astype(np.int32),
                 diagind=np.nonzero(rows == cols)[0].astype(np.int32), vals=np.ascontiguousarray(vals.reshape(-1)))
        p = make_prec(m)
        p.level_count()
        before = p.memory_stats()["bytes"]
        p.ilu0_factorize(-1)
        assert p.level_stats()["levels"] == n
        # no plan array: 64 bytes per row of a level padded to 16 (bs = 4) or 256 (scalar) rows would be n * 1 KB / 16 KB
        assert p.memory_stats()["bytes"] - before < n * 600 + (1 << 20)
        f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
        assert rel(p.get_iluvals(), f) < TOL
        p.close()


@pytest.mark.parametrize("case", ["poisson16_bs4", "poisson16_csr", "poisson9_bs8", "2dcyl1_bs4_row", "random_bs5"])
def test_exact_factorisation_falls_back_when_a_wave_gives_up(golden, case):
    """The single-launch factorisations bound their spins; a wave that runs out raises a flag and the host redoes
    the factorisation with one launch per level, whatever the attempt left behind.  Forced here (tuning
    "factorsf=a1": the attempt leaves the fill pattern all over the diagonal + upper part and reports failure):
    the factor is the usual one bit for bit and the give-up is counted."""
    m = matrices(golden)[case]()
    p = make_prec(m)
    p.ilu0_factorize(-1)
    f0 = p.get_iluvals()
    before = p.level_stats()
    capi.set_tuning("factorsf=a1")
    try:
        p.ilu0_factorize(-1)
        f1 = p.get_iluvals()
        st = p.level_stats()
    finally:
        capi.set_tuning("factorsf=a0")
    assert np.all(np.isfinite(f1)) and np.array_equal(f0, f1)
    if st["syncfree_passes"] != before["syncfree_passes"]:   # ("level=launch" never tries the single launch)
        assert st["syncfree_aborts"] == before["syncfree_aborts"] + 1
        # an operator whose single-launch passes keep giving up stops trying (a give-up costs the whole spin budget):
        # after the third one the per-level launches are used straight away, for the solves too
        capi.set_tuning("factorsf=a1")
        try:
            for _ in range(4):
                p.ilu0_factorize(-1)
            st2 = p.level_stats()
            assert np.array_equal(f0, p.get_iluvals())
        finally:
            capi.set_tuning("factorsf=a0")
        assert st2["syncfree_aborts"] == before["syncfree_aborts"] + 3
        p.ilu0_factorize(-1)
        r = W.rhs_vector(m["nbrows"] * m["bs"])
        z = p.ilu0_apply(r, 1, mode=capi.LEVEL)
        st3 = p.level_stats()
        assert st3["syncfree_passes"] == st2["syncfree_passes"] and st3["syncfree_aborts"] == st2["syncfree_aborts"]
        assert rel(z, O.ilu0_apply(m, O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"], r, 1, mode=O.GS_SERIAL)) < TOL
    p.close()


def test_level_schedule_falls_back_when_the_polling_launch_gives_up():
    """The same for the level-schedule build (tuning "levelfast=2": the polling launch's result is thrown away as if
    it had given up): the relaxation passes find the same levels from scratch."""
    m = W.poisson3d(14, 4)
    ref = W.dependency_levels(m)
    capi.set_tuning("levelfast=2")
    try:
        p = make_prec(m)
        lv, _, _ = p.get_levels()
        st = p.level_stats()
    finally:
        capi.set_tuning("levelfast=1")
    assert np.array_equal(lv, ref) and st["build_passes"] > 2
    p.close()


def test_level_schedule_nonsymmetric_pattern(golden):
    m = one_sided(W.poisson3d(10, 4))
    p = make_prec(m)
    lv, rows, ptr = p.get_levels()
    assert np.array_equal(lv, W.dependency_levels(m))
    with pytest.raises(ValueError):  # the reference gives up on such a pattern
        O.compute_levels(W.permute_symmetric(m, rows))
    # exact passes all the same
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL)
    gf = p.get_iluvals()
    assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), O.ilu0_apply(m, gf, r, 1, mode=O.GS_SERIAL)) < TOL
    p.jacobi_compute()
    gd = p.get_dblocks()
    x = p.sgs_relax(r, np.zeros(n), 2, mode=capi.LEVEL)
    assert rel(x, O.sgs_relax(m, gd, r, maxits=2, mode=O.GS_SERIAL)) < TOL
    p.close()


@pytest.mark.parametrize("scaling", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_level_ilu0_apply(golden, case, scaling, level_impl):
    """async_level_ilu0: asynchronous factorisation, exact (level-scheduled) triangular solves."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    # two synchronous sweeps leave msc00726's factor non-finite: use the exact factor there
    nb = -1 if case == "msc_csr" else 2
    p.ilu0_factorize(nb, init=capi.INIT_F_ORIGINAL, usescale=scaling, mode=capi.JACOBI_SYNC)
    gf = p.get_iluvals()
    sc = p.get_scale() if scaling else None
    z = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    want = O.ilu0_apply(m, gf, r, 1, mode=O.GS_SERIAL, scale=sc)
    assert rel(z, want) < TOL
    # more sweeps or the sequential symbol change nothing: one pass is already exact
    assert np.array_equal(p.ilu0_apply(r, 3, mode=capi.LEVEL), z)
    assert np.array_equal(p.ilu0_apply(r, -1), z)
    check_stats(p, level_impl)
    # the reference's level-scheduled operator on the level-ordered matrix, mapped back
    lv, rows, ptr = p.get_levels()
    try:
        levels = O.compute_levels(W.permute_symmetric(m, rows))
    except ValueError:
        levels = None
    if levels is not None:
        mp = W.permute_symmetric(m, rows)
        bs = m["bs"]
        pp = make_prec(mp)
        pp.ilu0_factorize(nb, init=capi.INIT_F_ORIGINAL, usescale=scaling, mode=capi.JACOBI_SYNC)
        perm = (rows[:, None].astype(np.int64) * bs + np.arange(bs)[None, :]).reshape(-1)
        zp = O.level_ilu0_apply(mp, pp.get_iluvals(), levels, r[perm], scale=pp.get_scale() if scaling else None)
        assert rel(zp, z[perm]) < 1e-11
        pp.close()
    p.close()


@pytest.mark.parametrize("case", CASES)
def test_level_sgs(golden, case, level_impl):
    """level_sgs: exact symmetric Gauss-Seidel application and relaxation."""
    m = matrices(golden)[case]()
    n = m["nbrows"] * m["bs"]
    bs = m["bs"]
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.jacobi_compute()
    gd = p.get_dblocks()
    z = p.sgs_apply(r, 1, mode=capi.LEVEL)
    assert rel(z, O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL)) < TOL
    assert np.array_equal(p.sgs_apply(r, -1), z)
    x0 = 0.2 * np.cos(np.arange(n))
    x = p.sgs_relax(r, x0.copy(), 3, mode=capi.LEVEL)
    want = O.sgs_relax(m, gd, r, x0=x0, maxits=3, mode=O.GS_SERIAL)
    if np.all(np.isfinite(want)) and np.abs(want).max() < 1e6:
        assert rel(x, want) < 1e-11
    xg = p.gs_relax(r, x0.copy(), 3, mode=capi.LEVEL)
    wantg = O.gs_relax(m, gd, r, x0=x0, nsweeps=3, mode=O.GS_SERIAL)
    if np.all(np.isfinite(wantg)) and np.abs(wantg).max() < 1e6:
        assert rel(xg, wantg) < 1e-11
    check_stats(p, level_impl)
    lv, rows, ptr = p.get_levels()
    mp = W.permute_symmetric(m, rows)
    perm = (rows[:, None].astype(np.int64) * bs + np.arange(bs)[None, :]).reshape(-1)
    levels = O.compute_levels(mp)
    gdp = O.jacobi_compute(mp)
    assert rel(O.level_sgs_apply(mp, gdp, levels, r[perm]), z[perm]) < 1e-11
    xr = O.level_sgs_relax(mp, gdp, levels, r[perm], x0=x0[perm], maxits=3)
    if np.all(np.isfinite(want)) and np.abs(want).max() < 1e6:
        assert rel(xr, x[perm]) < 1e-10
    p.close()


def test_level_mode_rejected_by_factorize(golden):
    p = make_prec(W.poisson3d(6, 4))
    with pytest.raises(capi.BlastedHipError):
        p.ilu0_factorize(1, mode=capi.LEVEL)
    p.close()


def test_exact_pass_is_bit_reproducible(level_impl):
    """An exact pass is deterministic -- each row is one fixed expression of final inputs -- whatever the
    order in which waves happen to run: repeated applies agree bit for bit, and none gives up waiting
    (tools/soak_exact.py is the long version of this test)."""
    m = W.poisson3d(34, 4, grid="uniform")
    n = m["nbrows"] * 4
    r = W.rhs_vector(n)
    p = make_prec(m)
    p.ilu0_factorize(2)
    p.jacobi_compute()
    z0 = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    s0 = p.sgs_apply(r, 1, mode=capi.LEVEL)
    for _ in range(40):
        assert np.array_equal(p.ilu0_apply(r, 1, mode=capi.LEVEL), z0)
        assert np.array_equal(p.sgs_apply(r, 1, mode=capi.LEVEL), s0)
    check_stats(p, level_impl)
    p.close()


@pytest.mark.parametrize("case", ["2dcyl1_bs4_col", "msc_csr"])
def test_no_dependence_inside_a_level(golden, case):
    """The reference's own level-schedule test (tests/mat_ops/testlevelschedule.cpp:18-48, run on 2dcyl1
    bs=4 and msc00726 bs=1, tests/mat_ops/CMakeLists.txt:126-132): no row of a level has a stored block
    in the column of another row of the same level."""
    m = matrices(golden)[case]()
    p = make_prec(m)
    lv, rows, ptr = p.get_levels()
    rp, ci = m["browptr"], m["bcolind"]
    rowof = np.repeat(np.arange(m["nbrows"]), np.diff(rp))
    off = rowof != ci
    assert not np.any(lv[rowof[off]] == lv[ci[off]])
    # and the same check, literally, on the reference's consecutive-row levels of the level-ordered matrix
    mp = W.permute_symmetric(m, rows)
    levels = O.compute_levels(mp)
    lvl_of = np.repeat(np.arange(levels.size - 1), np.diff(levels))
    rowofp = np.repeat(np.arange(mp["nbrows"]), np.diff(mp["browptr"]))
    offp = rowofp != mp["bcolind"]
    assert not np.any(lvl_of[rowofp[offp]] == lvl_of[mp["bcolind"][offp]])
    p.close()


def arrowhead(nb, bs, rowmajor=False):
    """Block arrowhead + tridiagonal pattern: row 0 is coupled to every row (a row far longer than the
    register-held block passes of the single-launch kernels), row i to i-1 and i+1 (nb levels)."""
    rows, cols = [], []
    for i in range(nb):
        cs = {i, 0}
        if i > 0:
            cs.add(i - 1)
        if i + 1 < nb:
            cs.add(i + 1)
        if i == 0:
            cs = set(range(nb))
        for c in sorted(cs):
            rows.append(i)
            cols.append(c)
    rows, cols = np.array(rows), np.array(cols)
    rp = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp)
    rng = np.random.default_rng(5)
    vals = rng.uniform(-1.0, 1.0, (rows.size, bs, bs)) * (0.2 / bs)
    vals[rows == 0] *= 4.0 / nb
    vals[cols == 0] *= 4.0 / nb
    dg = np.nonzero(rows == cols)[0]
    vals[dg] = np.eye(bs)[None] * 2.0 + rng.uniform(-0.1, 0.1, (nb, bs, bs))
    if rowmajor:
        vals = vals.transpose(0, 2, 1)
    return {"nbrows": nb, "nnzb": int(rows.size), "bs": bs, "rowmajor": bool(rowmajor),
            "browptr": rp.astype(np.int32), "bcolind": cols.astype(np.int32), "diagind": dg.astype(np.int32),
            "vals": np.ascontiguousarray(vals.reshape(-1))}


@pytest.mark.parametrize("bs,rowmajor", [(4, False), (5, False), (8, False), (1, False), (4, True)])
def test_exact_passes_with_a_very_long_row(bs, rowmajor, level_impl):
    """Rows longer than what the single-launch kernels hold in registers: their remainder loops (and, for
    the general kernel, the fallback to one launch per level) give the serial result all the same."""
    m = arrowhead(300, bs, rowmajor)
    n = m["nbrows"] * bs
    r = W.rhs_vector(n)
    p = make_prec(m)
    assert p.level_count() == 300
    p.ilu0_factorize(-1)
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    assert rel(p.get_iluvals(), f) < 1e-11
    assert rel(p.ilu0_apply(r, 1, mode=capi.LEVEL), O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)) < 1e-11
    p.jacobi_compute()
    gd = p.get_dblocks()
    assert rel(p.sgs_apply(r, 1, mode=capi.LEVEL), O.sgs_apply(m, gd, r, 1, mode=O.GS_SERIAL)) < 1e-11
    x = p.sgs_relax(r, np.zeros(n), 2, mode=capi.LEVEL)
    assert rel(x, O.sgs_relax(m, gd, r, maxits=2, mode=O.GS_SERIAL)) < 1e-11
    # and the asynchronous kernels' remainder loops
    z = p.ilu0_apply(r, 310, mode=capi.ASYNC)
    assert rel(z, O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)) < 1e-10
    assert rel(p.spmv(r), O.spmv(m, r)) < 1e-13
    assert p.level_stats()["syncfree_aborts"] == 0
    p.close()
