"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties
against an independent torch (eager, gather + bmm + index_add) evaluation of the same sweeps:
  config 2: 3-D Poisson 256^3, bs=4, async block-ILU(0) factor + apply
  config 3: same matrix, async block-SGS relaxation
  config 4: unstructured bs=5, ~2 M block-rows
  config 5: bs=8, 10^6 block-rows
The oracle cannot run these sizes in seconds; the oracle comparisons of the same kernels at small and
medium sizes are in test_gpu_parity.py.
"""
import numpy as np
import pytest

from blasted_amd import capi, workloads as W

pytestmark = pytest.mark.gpu


def torch_part_matvec(m, vals, x, part, dblocks_mode=None):
    """sum_{j in part(i)} A_ij x_j for every block-row, with plain torch ops (independent checker)."""
    import torch
    bs, nb, nnzb = m["bs"], m["nbrows"], m["nnzb"]
    dev = vals.device
    bp = m["browptr"].to(torch.int64)
    dg = m["diagind"].to(torch.int64)
    counts = bp[1:] - bp[:-1]
    out = torch.zeros(nb, bs, dtype=torch.float64, device=dev)
    xv = x.view(nb, bs)
    chunk = 1 << 22
    rows_all = torch.repeat_interleave(torch.arange(nb, device=dev), counts)
    for p0 in range(0, nnzb, chunk):
        p1 = min(nnzb, p0 + chunk)
        rows = rows_all[p0:p1]
        pos = torch.arange(p0, p1, device=dev)
        d = dg[rows]
        if part == "lower":
            sel = pos < d
        elif part == "upper":
            sel = pos > d
        elif part == "offdiag":
            sel = pos != d
        elif part == "diag":
            sel = pos == d
        else:
            sel = torch.ones_like(pos, dtype=torch.bool)
        blocks = vals[p0 * bs * bs:p1 * bs * bs].view(-1, bs, bs)  # column-major: [c][r]
        cols = m["bcolind"][p0:p1].to(torch.int64)
        xg = xv[cols]                                             # [nblk, bs] (index c)
        contrib = torch.einsum("bcr,bc->br", blocks, xg)
        out.index_add_(0, rows[sel], contrib[sel])
        del rows, pos, d, sel, blocks, cols, xg, contrib
    return out.reshape(-1)


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def poisson256():
    import torch
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(256, 4, dev, grid="uniform")
    r = W.rhs_vector_device(m["nbrows"] * 4, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    yield m, r, p
    p.close()


def test_config2_sizes_and_positions(poisson256):
    m, r, p = poisson256
    assert m["nbrows"] == 16777216 and m["nnzb"] == 117047296
    import ctypes as C
    capi._check(capi.lib().blasted_hip_ilu0_positions(p._h))
    npairs = C.c_long(0)
    capi._check(capi.lib().blasted_hip_ilu0_positions_size(p._h, C.byref(npairs)))
    assert npairs.value == 3 * 256 ** 3 - 3 * 256 ** 2  # 50 135 040: only diagonal entries have pairs


def test_config2_spmv_matches_torch(poisson256):
    import torch
    m, r, p = poisson256
    y = p.spmv(r)
    want = torch_part_matvec(m, m["vals"], r, "all")
    assert relmax(y, want) < 1e-13
    z = p.gemv3(-2.0, r, 0.5, y)
    assert relmax(z, -2.0 * want + 0.5 * y) < 1e-13


def test_config2_ilu_apply_sync_sweeps_match_torch(poisson256):
    """With the un-factored matrix as 'factor' (0 build sweeps, INIT_F_ORIGINAL: iluvals = A with
    inverted diagonal blocks) two synchronous sweeps per triangle are re-derived with torch."""
    import torch
    m, r, p = poisson256
    p.ilu0_factorize(0, init=capi.INIT_F_ORIGINAL)
    z = p.ilu0_apply(r, 2, init=capi.INIT_A_ZERO, mode=capi.JACOBI_SYNC)
    A = m["vals"]
    # lower: y1 = r - L*0 = r ; y2 = r - L y1
    y2 = r - torch_part_matvec(m, A, r, "lower")
    # upper with D = inverse of A's diagonal blocks: z1 = D y2 ; z2 = D (y2 - U z1)
    nb = m["nbrows"]
    dblk = A.view(-1, 4, 4)[m["diagind"].to(torch.int64)]          # [nb][c][r]
    dinv = torch.linalg.inv(dblk.transpose(1, 2))                   # math layout [r][c]
    z1 = torch.einsum("brc,bc->br", dinv, y2.view(nb, 4)).reshape(-1)
    t = y2 - torch_part_matvec(m, A, z1, "upper")
    z2 = torch.einsum("brc,bc->br", dinv, t.view(nb, 4)).reshape(-1)
    assert relmax(z, z2) < 1e-12
    # linearity of the synchronous apply
    r2 = torch.cos(0.11 * torch.arange(r.numel(), dtype=torch.float64, device=r.device))
    za = p.ilu0_apply(r2, 2, mode=capi.JACOBI_SYNC)
    zc = p.ilu0_apply(2.0 * r - 3.0 * r2, 2, mode=capi.JACOBI_SYNC)
    assert relmax(zc, 2.0 * z - 3.0 * za) < 1e-12


def test_config2_async_factor_and_apply_properties(poisson256):
    """3 async build sweeps + async apply at full size: finite, remainder far below the initial one, and
    the async apply lies between the synchronous iterate and the converged solve."""
    import torch
    m, r, p = poisson256
    info = p.ilu0_factorize(3, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC, compute_info=True)
    assert np.isfinite(info).all() and info[0] < 1e-3 * info[1]
    zs = p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC)
    za = p.ilu0_apply(r, 3, mode=capi.ASYNC)
    zlong = p.ilu0_apply(r, 40, mode=capi.ASYNC)
    assert torch.isfinite(za).all()
    es, ea = float((zs - zlong).abs().max()), float((za - zlong).abs().max())
    assert ea <= es * 1.0001


def test_config2_level_scheduled_apply_is_exact(poisson256):
    """LEVEL mode at full size: 766 wavefront levels (i+j+k), and the result satisfies both triangular
    systems -- checked with the torch evaluation of L y and U z."""
    import torch
    m, r, p = poisson256
    assert p.level_count() == 3 * 256 - 2
    lv, rows, ptr = p.get_levels()
    i = np.arange(m["nbrows"])
    assert np.array_equal(lv, (i % 256 + (i // 256) % 256 + i // 65536).astype(np.int32))
    p.ilu0_factorize(3, init=capi.INIT_F_ORIGINAL, mode=capi.ASYNC)
    z = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    nb = m["nbrows"]
    F = torch.from_numpy(p.get_iluvals()).to(r.device)   # factor, diagonal blocks inverted
    y = torch.from_numpy(p.get_ytemp()).to(r.device)
    assert relmax(y, r - torch_part_matvec(m, F, y, "lower")) < 1e-12
    dinv = F.view(-1, 4, 4)[m["diagind"].to(torch.int64)].transpose(1, 2)   # math layout [r][c]
    t = y - torch_part_matvec(m, F, z, "upper")
    assert relmax(z, torch.einsum("brc,bc->br", dinv, t.view(nb, 4)).reshape(-1)) < 1e-12
    del F
    # the async iteration converges to it
    zlong = p.ilu0_apply(r, 60, mode=capi.ASYNC)
    z3 = p.ilu0_apply(r, 3, mode=capi.ASYNC)
    assert float((zlong - z).abs().max()) < float((z3 - z).abs().max())


def test_config2_exact_factorisation_has_no_remainder(poisson256):
    """seqilu0's factorisation at full size (one launch per dependency level): A - LU vanishes on the
    pattern, the reference's own criterion (tests/solverops/async_ilu_convergence.cpp:462-490,574-575)."""
    m, r, p = poisson256
    info = p.ilu0_factorize(-1, init=capi.INIT_F_ORIGINAL, compute_info=True)
    assert np.isfinite(info).all()
    assert info[0] < 2e-15 * info[1]
    # three asynchronous sweeps leave a remainder many orders above that
    info3 = p.ilu0_factorize(3, init=capi.INIT_F_ORIGINAL, compute_info=True)
    assert info3[0] > 1e3 * info[0]


def test_config3_sgs_relaxation_matches_torch(poisson256):
    import torch
    m, r, p = poisson256
    p.jacobi_compute()
    nb = m["nbrows"]
    A = m["vals"]
    dblk = A.view(-1, 4, 4)[m["diagind"].to(torch.int64)]
    dinv = torch.linalg.inv(dblk.transpose(1, 2))
    x = torch.zeros_like(r)
    p.sgs_relax(r, x, 1, mode=capi.JACOBI_SYNC)  # ascending pass then descending pass, synchronous
    x1 = torch.einsum("brc,bc->br", dinv, r.view(nb, 4)).reshape(-1)
    t = r - torch_part_matvec(m, A, x1, "offdiag")
    x2 = torch.einsum("brc,bc->br", dinv, t.view(nb, 4)).reshape(-1)
    assert relmax(x, x2) < 1e-12
    # config 3: async relaxation, 5 steps: the residual keeps decreasing (slowly: the right-hand side has
    # a smooth component that SGS on a 256^3 Poisson grid damps slowly)
    x = torch.zeros_like(r)
    p.sgs_relax(r, x, 1, mode=capi.ASYNC)
    res1 = float((r - p.spmv(x)).norm() / r.norm())
    p.sgs_relax(r, x, 4, mode=capi.ASYNC)
    res5 = float((r - p.spmv(x)).norm() / r.norm())
    assert res5 < res1 < 1.0


@pytest.mark.parametrize("cfg", ["config4_unstructured_bs5", "config5_poisson100_bs8"])
def test_config4_config5(cfg):
    import torch
    dev = torch.device("cuda", 0)
    if cfg.startswith("config4"):
        m = W.unstructured_bsr(126, 5, device=dev)      # 2 000 376 block-rows
        assert 1.9e6 < m["nbrows"] < 2.1e6
    else:
        m = W.poisson3d_device(100, 8, dev, grid="uniform")
        assert m["nbrows"] == 1000000
    bs, nb = m["bs"], m["nbrows"]
    r = W.rhs_vector_device(nb * bs, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    A = m["vals"]
    assert relmax(p.spmv(r), torch_part_matvec(m, A, r, "all")) < 1e-13
    p.ilu0_factorize(0, init=capi.INIT_F_ORIGINAL)
    z = p.ilu0_apply(r, 2, init=capi.INIT_A_ZERO, mode=capi.JACOBI_SYNC)
    y2 = r - torch_part_matvec(m, A, r, "lower")
    dblk = A.view(-1, bs, bs)[m["diagind"].to(torch.int64)]
    dinv = torch.linalg.inv(dblk.transpose(1, 2))
    z1 = torch.einsum("brc,bc->br", dinv, y2.view(nb, bs)).reshape(-1)
    t = y2 - torch_part_matvec(m, A, z1, "upper")
    z2 = torch.einsum("brc,bc->br", dinv, t.view(nb, bs)).reshape(-1)
    assert relmax(z, z2) < 1e-11
    # async factor + apply: converges to a fixed point (one more sweep changes nothing)
    nsw = 40 if cfg.startswith("config4") else 300
    info = p.ilu0_factorize(nsw, mode=capi.ASYNC, compute_info=True)
    assert np.isfinite(info).all() and info[0] < 1e-10 * info[1]
    za = p.ilu0_apply(r, nsw, mode=capi.ASYNC)
    zb = p.ilu0_apply(r, nsw + 1, mode=capi.ASYNC)
    assert relmax(za, zb) < 1e-12
    # the exact forms reach the same fixed points in one pass: level-scheduled factorisation and solves
    fa = torch.from_numpy(p.get_iluvals()).to(dev)
    info = p.ilu0_factorize(-1, compute_info=True)
    assert info[0] < 1e-14 * info[1]
    fe = torch.from_numpy(p.get_iluvals()).to(dev)
    assert relmax(fa, fe) < 1e-9
    del fa
    ze = p.ilu0_apply(r, 1, mode=capi.LEVEL)
    assert relmax(za, ze) < 1e-9
    y = torch.from_numpy(p.get_ytemp()).to(dev)
    assert relmax(y, r - torch_part_matvec(m, fe, y, "lower")) < 1e-12
    assert p.level_stats()["syncfree_aborts"] == 0
    p.close()


def test_scalar_many_rows_takes_the_wide_chunks():
    """Scalar matrices of a million rows and more run the row sweeps with 256 rows per workgroup (sweep_geo.hpp);
    the tuning "gunroll=1" keeps the 128-row form.  Synchronous sweeps are deterministic: every operator gives the
    same bits in both, and the SpMV is the torch one."""
    import torch
    dev = torch.device("cuda", 0)
    m = W.poisson3d_device(104, 1, dev, grid="uniform")     # 1 124 864 rows
    assert m["nbrows"] >= 1 << 20
    n = m["nbrows"]
    r = W.rhs_vector_device(n, dev)
    p = capi.Prec(0, torch.cuda.current_stream().cuda_stream)
    p.set_matrix(m)
    p.ilu0_factorize(2, mode=capi.JACOBI_SYNC)
    p.jacobi_compute()
    res = {}
    try:
        for k in ("0", "1"):
            capi.set_tuning("gunroll=" + k)
            x0 = torch.zeros_like(r)
            res[k] = [p.spmv(r).clone(), p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC).clone(),
                      p.sgs_apply(r, 2, mode=capi.JACOBI_SYNC).clone(), p.jacobi_apply(r).clone(),
                      p.sgs_relax(r, x0, 2, mode=capi.JACOBI_SYNC).clone()]
    finally:
        capi.set_tuning("gunroll=0")
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)
    assert relmax(res["0"][0], torch_part_matvec(m, m["vals"], r, "all")) < 1e-13
    p.close()


def test_class_aware_placement_changes_where_not_what():
    """Round 4: the compact triangle copies of an operator large enough for it (copies of 64 MiB and more:
    here Poisson 128^3 bs=4, 768 / 1018 MiB) are built from 1 GiB pieces that are checked, with a read-beside-write probe,
    against the vectors the sweeps read and write.  Whatever the search does -- off, quick, thorough -- the synchronous
    sweeps give the SAME BITS; the placement counters move; `placement_check` answers; and the operator's memory
    accounting still adds up when it is destroyed and another one is made."""
    import torch
    dev = torch.device("cuda:0")
    m = W.poisson3d_device(128, 4, dev, grid="uniform")
    n = m["nbrows"] * 4
    r = W.rhs_vector_device(n, dev)
    results = {}
    import os
    capi.set_tuning("compactafter=0")   # the copies with the first application, whatever the suite runs under
    capi.set_tuning("placeafter=0")     # ... and the quick search with them (the default waits for the 256th application)
    try:
        for mode in ("0", "1", "2"):
            capi.set_tuning("placement=" + mode)
            before = capi.placement_stats()
            p = capi.Prec(0)
            p.set_matrix(m)
            p.ilu0_factorize(-1)
            z = torch.zeros_like(r)
            p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC, out=z)
            results[mode] = z.clone()
            za = p.ilu0_apply(r, 3, mode=capi.ASYNC).clone()     # asynchronous sweeps on the same copies
            assert torch.isfinite(za).all()
            after = capi.placement_stats()
            where = p.placement_check(r, z)
            assert where["lower_pieces"] == 1 and where["upper_pieces"] == 1      # 768 MiB and 1018 MiB: one piece each
            if mode == "0":
                assert after["buffers"] == before["buffers"]
            else:
                assert after["buffers"] >= before["buffers"] + 2 and after["probes"] > before["probes"]
                # the musts: whatever was checked is not in the class of the vector the sweep writes
                if after["unchecked"] == before["unchecked"]:
                    assert where["upper_in_z_class"] == 0 and where["lower_in_ytemp_class"] == 0
            st = p.memory_stats()
            assert st["bytes"] > 0 and st["derived_copies"] == 2
            p.close()
    finally:
        capi.set_tuning("placement=1")
        capi.set_tuning("compactafter=" + os.environ.get("BLASTED_HIP_COMPACT_AFTER", "-1"))
        capi.set_tuning("placeafter=" + os.environ.get("BLASTED_HIP_PLACE_AFTER", "256"))
    assert torch.equal(results["0"], results["1"]) and torch.equal(results["0"], results["2"])


def test_quick_placement_waits_until_the_operator_has_lived_long_enough():
    """The default's two stages: plain compact copies when they pay (here: at once), placed ones -- made beside the plain
    ones, which are then freed -- once the operator has been applied `placeafter` times in its life, refactorisations
    included.  Synchronous sweeps give the same bits before and after; the accounting returns to one copy's worth."""
    import os
    import torch
    dev = torch.device("cuda:0")
    m = W.poisson3d_device(128, 4, dev, grid="uniform")
    n = m["nbrows"] * 4
    r = W.rhs_vector_device(n, dev)
    capi.set_tuning("compactafter=0")
    capi.set_tuning("placeafter=6")
    capi.set_tuning("placement=1")
    try:
        p = capi.Prec(0)
        p.set_matrix(m)
        p.ilu0_factorize(-1)
        z = torch.zeros_like(r)
        before = capi.placement_stats()
        p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC, out=z)
        first = z.clone()
        assert capi.placement_stats()["buffers"] == before["buffers"]            # plain copies, no search
        bytes_plain = p.memory_stats()["bytes"]
        p.ilu0_apply(r, 3, mode=capi.ASYNC)
        p.ilu0_factorize(-1)                                                     # a refactorisation keeps the count
        for _ in range(3):
            p.ilu0_apply(r, 3, mode=capi.ASYNC)
        p.ilu0_apply(r, 1, mode=capi.LEVEL)        # (the level ordering takes the plain copies' storage over: not counted)
        assert capi.placement_stats()["buffers"] == before["buffers"]            # applications 1 .. 5
        p.ilu0_apply(r, 3, mode=capi.ASYNC)                                      # the 6th: natural order again, still plain
        assert capi.placement_stats()["buffers"] == before["buffers"]
        bytes_plain = p.memory_stats()["bytes"]                                  # (with the level schedule and its iterates)
        for _ in range(2):
            p.ilu0_apply(r, 3, mode=capi.ASYNC)                                  # ... the 7th places
        after = capi.placement_stats()
        # the 7th application looked at the plain copies (a triangle that meets its must stays where it is) and placed the
        # others beside them
        assert after["probes"] > before["probes"]
        st = p.memory_stats()
        assert st["derived_copies"] == 2 and abs(st["bytes"] - bytes_plain) < 64 << 20   # replaced plain copies are gone
        p.ilu0_apply(r, 3, mode=capi.JACOBI_SYNC, out=z)
        assert torch.equal(z, first)
        z2 = torch.zeros_like(r)
        probes = capi.placement_stats()["probes"]
        p.ilu0_apply(r, 3, mode=capi.ASYNC, out=z2)
        assert capi.placement_stats()["probes"] == probes                        # ... once
        where = p.placement_check(r, z)
        assert where["lower_pieces"] == 1 and where["upper_pieces"] == 1
        if after["unchecked"] == before["unchecked"]:
            assert where["lower_in_ytemp_class"] == 0
        p.close()
    finally:
        capi.set_tuning("compactafter=" + os.environ.get("BLASTED_HIP_COMPACT_AFTER", "-1"))
        capi.set_tuning("placeafter=" + os.environ.get("BLASTED_HIP_PLACE_AFTER", "256"))


def test_thorough_placement_keeps_its_own_matrix_copy_for_relaxation():
    """Round 4, thorough placement only (`placement=2`): after a few asynchronous relaxation passes the operator
    streams its OWN copy of the matrix values, kept out of the address class of the vector the passes write (the
    borrowed values lie where the caller put them).  The copy changes where the pass reads, not what: the residual
    history is the default's within the spread of asynchronous passes, the memory accounting shows the copy, and new
    values (`set_values`) are followed."""
    import torch
    dev = torch.device("cuda:0")
    m = W.poisson3d_device(128, 4, dev, grid="uniform")      # 1.9 GB of values
    n = m["nbrows"] * 4
    r = W.rhs_vector_device(n, dev)
    res = {}
    try:
        for mode in ("0", "2"):
            capi.set_tuning("placement=" + mode)
            p = capi.Prec(0)
            p.set_matrix(m)
            p.jacobi_compute()
            x = torch.zeros_like(r)
            hist = []
            for _ in range(6):
                p.sgs_relax(r, x, 1, mode=capi.ASYNC)        # two passes per call
                hist.append(float((r - p.spmv(x)).norm() / r.norm()))
            res[mode] = hist
            st = p.memory_stats()
            assert st["derived_copies"] == (2 if mode == "2" else 0)
            if mode == "2":
                # new values: twice the matrix -- the relaxation must follow them (x <- the solution of 2 A x = r)
                m2 = dict(m)
                m2["vals"] = m["vals"] * 2.0
                p.set_values(m2["vals"])
                p.jacobi_compute()
                x2 = torch.zeros_like(r)
                for _ in range(6):
                    p.sgs_relax(r, x2, 1, mode=capi.ASYNC)
                assert float((2.0 * x2 - x).norm() / x.norm()) < 0.05
            p.close()
    finally:
        capi.set_tuning("placement=1")
    assert all(b < a for a, b in zip(res["2"], res["2"][1:]))
    assert all(abs(a - b) < 0.02 * a for a, b in zip(res["0"], res["2"]))
