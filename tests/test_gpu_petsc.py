"""GPU tests of the PCSHELL surface (SURVEY 8 rows b / f1): blasted_amd/host/src/blasted_petsc.cpp is built
against the test-only mini-PETSc of tests/petsc_stub and EXECUTED by tests/cpp/petsc_driver.cpp the way a
PETSc application runs it (the reference's tests/runpetsc.c flow): options database -> setup_blasted_stack
-> KSPSetUp / set-up on blocks -> compute_preconditioner_blasted -> apply_local_blasted / relax_local_blasted
-> a recompute on changed values -> computeTotalTimes -> cleanup_blasted, on the reference's 2dcyl1 matrix
(PETSc binary, block size from its .info file) as SeqBAIJ and SeqAIJ, with host and with HIP-resident vectors.
Results are compared with the CPU oracle at the same settings (reference: src/blasted_petsc.cpp:403-575,
578-721)."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from blasted_amd import mtxio, workloads as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "build", "petsc_driver")
G = os.path.join(ROOT, "tests", "golden")
PMAT = os.path.join(G, "2dcyl1.pmat")

ASYNC_OPTS = ["-blasted_async_fact_init_type", "init_original", "-blasted_async_apply_init_type", "init_zero",
              "-blasted_thread_chunk_size", "128", "-blasted_use_symmetric_scaling", "0"]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run(tmp_path, petsc_opts, mat_type="baij", vec_type="seq", pc=("bjacobi", "shell"), env=None, extra=(),
        expect_rc=0):
    out = str(tmp_path / "o")
    tree = ["-pc_type", pc[0]] + (["-sub_pc_type", pc[1]] if pc[1] else [])
    cmd = [DRIVER, "--mat_file", PMAT, "--mat_type", mat_type, "--vec_type", vec_type, "--out", out] + list(extra) + \
          ["--"] + tree + list(petsc_opts)
    e = dict(os.environ)
    for k in ("BLASTED_HIP_SYNC_SWEEPS", "BLASTED_HIP_EXACT_APPLY", "BLASTED_HIP_SWEEP_MODE"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e)
    # (the driver's report is line-buffered: if the process ever dies, what it printed shows how far it got)
    assert r.returncode == expect_rc, r.stdout[-3000:] + r.stderr[-2000:]
    rep = {}
    for line in r.stdout.splitlines():
        if " = " in line:
            k, v = line.split(" = ", 1)
            rep[k.strip()] = v.strip()
    vecs = {}
    for name in ("z", "x", "z2"):
        f = out + "_%s.bin" % name
        if os.path.exists(f):
            vecs[name] = np.fromfile(f, np.float64)
    return rep, vecs, r


def matrix(mat_type):
    return mtxio.read_petsc_bsr(PMAT, None if mat_type == "baij" else 1)


def check_common(rep, vecs, bs, homogeneous=True):
    assert rep["done"] == "1" and rep["blasted_contexts"] == "1"
    assert rep["outstanding_accesses"] == "0"        # every Get... of the glue has its Restore...
    assert int(rep["block_size"]) == bs and int(rep["node_bs"]) == bs
    assert float(rep["factor_walltime"]) > 0 and float(rep["apply_walltime"]) > 0
    # values * 2 in place, KSPSetOperators, set-up again: the operator of 2A applied to r is half that of A
    # (not for a few factorisation sweeps from A itself as the initial guess: that iteration is not
    # homogeneous in A -- those cases compare the second application with the oracle on 2A)
    if homogeneous:
        assert rel(vecs["z2"], 0.5 * vecs["z"]) < 1e-12


def doubled(m):
    m2 = dict(m)
    m2["vals"] = 2.0 * m["vals"]
    return m2


SYNC = {"BLASTED_HIP_SYNC_SWEEPS": "1"}


@pytest.mark.parametrize("mat_type,vec_type", [("baij", "seq"), ("aij", "seq"), ("baij", "hip"), ("aij", "hip")])
def test_pcshell_ilu0_matches_oracle(tmp_path, mat_type, vec_type):
    """-blasted_pc_type ilu0 -blasted_async_sweeps 3,3 with deterministic (synchronous) sweeps: factor and
    apply equal the oracle's synchronous sweeps at the same counts; HIP vectors take the device branch."""
    m = matrix(mat_type)
    bs = m["bs"]
    rep, vecs, _ = run(tmp_path, ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "3,3"] + ASYNC_OPTS,
                       mat_type, vec_type, env=SYNC)
    check_common(rep, vecs, bs, homogeneous=False)
    assert rep["node_prectype"] == "ilu0" and rep["node_sweeps"] == "3,3" and rep["richardson_callback"] == "0"
    r = W.rhs_vector(m["nbrows"] * bs)
    for mm, key in ((m, "z"), (doubled(m), "z2")):  # first set-up, and the set-up after the values changed
        f = O.ilu0_factorize(mm, None, 3, mode=O.JACOBI_SYNC, init=O.INIT_F_ORIGINAL)["iluvals"]
        want = O.ilu0_apply(mm, f, r, 3, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO)
        assert rel(vecs[key], want) < 1e-12
    if vec_type == "hip":
        assert int(rep["hip_vector_accesses"]) >= 4   # r and z of two applies went through VecHIPGetArray...
    else:
        assert rep["hip_vector_accesses"] == "0"


def test_pcshell_pinned_value_array(tmp_path):
    """-blasted_pin_host_arrays (not a reference option): the Mat's value array is page-locked for the operator's
    lifetime and released with it (cleanup_blasted); results are the same bits as without, through the set-up on
    changed values too."""
    opts = ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "3,3"] + ASYNC_OPTS
    _, plain, _ = run(tmp_path, opts, env=SYNC)
    rep, pinned, _ = run(tmp_path, opts + ["-blasted_pin_host_arrays", "1"], env=SYNC)
    check_common(rep, pinned, 4, homogeneous=False)
    assert np.array_equal(plain["z"], pinned["z"]) and np.array_equal(plain["z2"], pinned["z2"])


def test_pcshell_sweep_mode_follows_the_ksp_tree(tmp_path):
    """-blasted_sweep_mode (not a reference option) chooses how ilu0 / sgs apply their sweeps.  Without it the glue
    follows the KSP tree: under a Krylov method that assumes a fixed preconditioner (PETSc's default
    gmres, bcgs, cg) the sweeps are the deterministic (synchronous) ones -- the oracle's, to 1e-10 -- and
    setup_blasted_stack says so; under a flexible method (fgmres, gcr, richardson) they are the reference's chaotic
    sweeps.  A mistyped mode is a PETSc error code, not an exception through the C caller."""
    m = matrix("baij")
    r = W.rhs_vector(m["nbrows"] * 4)
    fexact = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    zsync = O.ilu0_apply(m, fexact, r, 3, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO)
    zexact = O.ilu0_apply(m, fexact, r, 1, mode=O.GS_SERIAL)
    # (the factorisation sweeps stay asynchronous in every mode: build far past the fixed point, compare the apply)
    opts = ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "40,3"] + ASYNC_OPTS
    notice = "assumes a fixed preconditioner: the ilu0 sweeps are applied in the deterministic"
    _, vecs, res = run(tmp_path, opts)                      # the stub's default -ksp_type is PETSc's: gmres
    assert "-ksp_type gmres " + notice in res.stdout
    assert rel(vecs["z"], zsync) < 1e-10
    zdet = vecs["z"]
    _, vecs, res = run(tmp_path, opts + ["-ksp_type", "bcgs"])
    assert "-ksp_type bcgs " + notice in res.stdout and rel(vecs["z"], zsync) < 1e-10
    for flexible in ("fgmres", "gcr", "richardson"):
        _, vecs, res = run(tmp_path, opts + ["-ksp_type", flexible])
        assert "fixed preconditioner" not in res.stdout
        # three chaotic sweeps: between the synchronous sweeps and the exact solves, and not the synchronous bits
        assert np.linalg.norm(vecs["z"] - zexact) <= 1.05 * np.linalg.norm(zsync - zexact)
        assert not np.array_equal(vecs["z"], zdet)
    _, vecs, res = run(tmp_path, opts + ["-blasted_sweep_mode", "async"])   # an explicit choice is kept, quietly
    assert "fixed preconditioner" not in res.stdout
    assert np.linalg.norm(vecs["z"] - zexact) <= 1.05 * np.linalg.norm(zsync - zexact)
    _, vecs, res = run(tmp_path, opts, env={"BLASTED_HIP_SWEEP_MODE": "async"})   # so is a process-wide one
    assert "fixed preconditioner" not in res.stdout
    _, _, res = run(tmp_path, ["-blasted_pc_type", "seqilu0", "-blasted_async_sweeps", "1,1"] + ASYNC_OPTS)
    assert "fixed preconditioner" not in res.stdout   # exact passes: a fixed operator already
    _, vecs, res = run(tmp_path, opts + ["-blasted_sweep_mode", "deterministic", "-ksp_type", "fgmres"])
    assert "fixed preconditioner" not in res.stdout and rel(vecs["z"], zsync) < 1e-10
    _, vecs, _ = run(tmp_path, opts + ["-blasted_sweep_mode", "exact"])
    assert rel(vecs["z"], zexact) < 1e-10
    _, _, res = run(tmp_path, opts + ["-blasted_sweep_mode", "bogus"], expect_rc=10 + 62)  # PETSC_ERR_ARG_WRONG
    assert "sweep mode must be" in res.stderr and "terminate" not in res.stderr


def write_petsc_mat(path, m):
    """scalar CSR dict -> PETSc binary Mat (big-endian: classid, M, N, nnz, row lengths, columns, values)"""
    assert m["bs"] == 1
    rp = np.asarray(m["browptr"], dtype=np.int64)
    with open(path, "wb") as f:
        np.array([1211216, m["nbrows"], m["nbrows"], m["nnzb"]], dtype=">i4").tofile(f)
        (rp[1:] - rp[:-1]).astype(">i4").tofile(f)
        np.asarray(m["bcolind"]).astype(">i4").tofile(f)
        np.asarray(m["vals"]).astype(">f8").tofile(f)


def write_petsc_vec(path, v):
    with open(path, "wb") as f:
        np.array([1211214, v.size], dtype=">i4").tofile(f)
        v.astype(">f8").tofile(f)


def test_pcshell_default_options_converge_under_bcgs_on_a_mid_size_grid(tmp_path):
    """an out-of-the-box user (-pc_type shell under PETSc's non-flexible default Krylov methods, no
    -blasted_sweep_mode) must get a converging solve.  Poisson 48^3 (110 592 rows), ilu0 with the reference's 3+3
    sweeps, the driver's right-preconditioned BiCGStab: with default options the glue picks the deterministic sweeps
    and the solve reaches 1e-8; the iteration count is the one the explicit deterministic mode gives."""
    m = W.poisson3d(50, 1, grid="uniform")
    n = m["nbrows"]
    xs = np.sin(0.01 * np.arange(n)) + 1.0
    import scipy.sparse as sp
    A = sp.csr_matrix((m["vals"], m["bcolind"], m["browptr"]), shape=(n, n))
    mat, bf, xf = str(tmp_path / "p48.pmat"), str(tmp_path / "p48_b.pvec"), str(tmp_path / "p48_x.pvec")
    write_petsc_mat(mat, m)
    write_petsc_vec(bf, A @ xs)
    write_petsc_vec(xf, xs)

    def solve(extra, env=None):
        out = str(tmp_path / "s")
        cmd = [DRIVER, "--mat_file", mat, "--mat_type", "aij", "--out", out, "--b_file", bf, "--x_file", xf,
               "--solver_tol", "1e-8", "--max_iter", "300", "--", "-pc_type", "bjacobi", "-sub_pc_type", "shell",
               "-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "3,3"] + ASYNC_OPTS + extra
        e = {k: v for k, v in os.environ.items() if not k.startswith("BLASTED_HIP_S") and k != "BLASTED_HIP_EXACT_APPLY"}
        e.update(env or {})
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        rep = dict(line.split(" = ", 1) for line in r.stdout.splitlines() if " = " in line)
        return int(rep["solve_iterations"]), float(rep["solve_relres"]), float(rep["solve_error_l2"]), r.stdout

    for ksp in ([], ["-ksp_type", "bcgs"]):
        its, relres, err, text = solve(ksp)
        assert "applied in the deterministic" in text
        assert relres < 1e-8 and its < 300, (its, relres)
        assert err < 1e-5 * np.linalg.norm(xs)
    its_det, relres_det, _, _ = solve(["-blasted_sweep_mode", "deterministic"])
    assert (its_det, relres_det) == (its, relres)   # the same fixed operator: the same iteration, bit for bit


def test_pcshell_ilu0_scaled_with_info(tmp_path):
    """symmetric scaling and -blasted_compute_preconditioner_info: one PrecInfo per compute(), remainder
    below the initial remainder, values as the oracle's."""
    m = matrix("aij")
    opts = ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "4,3", "-blasted_async_fact_init_type", "init_original",
            "-blasted_async_apply_init_type", "init_jacobi", "-blasted_thread_chunk_size", "64",
            "-blasted_use_symmetric_scaling", "1", "-blasted_compute_preconditioner_info", "1"]
    rep, vecs, _ = run(tmp_path, opts, "aij", env=SYNC)
    check_common(rep, vecs, 1, homogeneous=False)
    assert rep["precinfo_entries"] == "2"
    r = W.rhs_vector(m["nbrows"])
    fo = O.ilu0_factorize(m, None, 4, mode=O.JACOBI_SYNC, init=O.INIT_F_ORIGINAL, usescale=True, compute_info=True)
    want = O.ilu0_apply(m, fo["iluvals"], r, 3, mode=O.JACOBI_SYNC, init=O.INIT_A_JACOBI, scale=fo["scale"])
    assert rel(vecs["z"], want) < 1e-12
    rem, rem0 = float(rep["precinfo_0_factor_remainder"]), float(rep["precinfo_0_factor_init_rem"])
    assert 0 < rem < rem0
    assert abs(rem - fo["precinfo"][0]) <= 1e-9 * fo["precinfo"][0]
    assert abs(rem0 - fo["precinfo"][1]) <= 1e-9 * fo["precinfo"][1]


@pytest.mark.parametrize("mat_type,vec_type", [("baij", "seq"), ("baij", "hip"), ("aij", "seq")])
def test_pcshell_sgs_apply_and_richardson(tmp_path, mat_type, vec_type):
    m = matrix(mat_type)
    bs = m["bs"]
    rep, vecs, _ = run(tmp_path, ["-blasted_pc_type", "sgs", "-blasted_async_sweeps", "1,3"] + ASYNC_OPTS,
                       mat_type, vec_type, env=SYNC, extra=["--relax_its", "4"])
    check_common(rep, vecs, bs)
    assert rep["richardson_callback"] == "1" and rep["richardson_its"] == "4"
    assert rep["richardson_reason"] == "4"  # PCRICHARDSON_CONVERGED_ITS
    r = W.rhs_vector(m["nbrows"] * bs)
    d = O.jacobi_compute(m)
    assert rel(vecs["z"], O.sgs_apply(m, d, r, 3, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO)) < 1e-12
    # guesszero = TRUE: x starts from 0 whatever the vector held
    assert rel(vecs["x"], O.sgs_relax(m, d, r, maxits=4, mode=O.JACOBI_SYNC)) < 1e-11


@pytest.mark.parametrize("sweep_mode", [None, "async", "deterministic"])
def test_pcshell_sgs_product_modes_forward_half_exact(tmp_path, sweep_mode):
    """the reference's low sweep count in the product modes.  Deterministic: exact forward half, then synchronous
    backward sweeps -- equal to the oracle's composition of the two.  Async (the default: the reference's chaotic
    sweeps): z is no farther from the exact SGS application than that (Q3)."""
    m = matrix("baij")
    env = {"BLASTED_HIP_SWEEP_MODE": sweep_mode} if sweep_mode else None
    fixed = sweep_mode == "deterministic"
    rep, vecs, _ = run(tmp_path, ["-blasted_pc_type", "sgs", "-blasted_async_sweeps", "1,2"] + ASYNC_OPTS, env=env)
    check_common(rep, vecs, 4, homogeneous=fixed)  # (chaotic sweeps are not reproducible to the last bit)
    r = W.rhs_vector(m["nbrows"] * 4)
    d = O.jacobi_compute(m)
    ze, ye = O.sgs_apply(m, d, r, 1, mode=O.GS_SERIAL, return_y=True)
    zj = O.sgs_apply(m, d, r, 2, mode=O.JACOBI_SYNC, init=O.INIT_A_NONE, y0=ye, z0=np.zeros_like(r))
    if fixed:
        assert rel(vecs["z"], zj) < 1e-12
    else:
        assert np.linalg.norm(vecs["z"] - ze) <= 1.05 * np.linalg.norm(zj - ze) + 1e-12 * np.linalg.norm(ze)


def test_pcshell_deterministic_mode_is_a_fixed_operator(tmp_path):
    """two runs with synchronous sweeps give bit-identical applications (what a non-flexible Krylov method needs),
    and BLASTED_HIP_SWEEP_MODE is validated"""
    opts = ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "3,3"] + ASYNC_OPTS
    _, v1, _ = run(tmp_path, opts, env=SYNC)   # synchronous factorisation as well: everything deterministic
    _, v2, _ = run(tmp_path, opts, env=SYNC)
    assert np.array_equal(v1["z"], v2["z"])
    _, _, r = run(tmp_path, opts, env={"BLASTED_HIP_SWEEP_MODE": "bogus"}, expect_rc=3)
    assert "sweep mode must be" in r.stderr


@pytest.mark.parametrize("opts,kind", [
    (["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "-1,-1"], "seqilu0"),
    (["-blasted_pc_type", "seqilu0", "-blasted_async_sweeps", "1,1"], "seqilu0"),
    (["-blasted_pc_type", "sapilu0", "-blasted_async_sweeps", "40,1"], "seqilu0"),
    (["-blasted_pc_type", "async_level_ilu0", "-blasted_async_sweeps", "40,1"], "seqilu0"),
    (["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "40,40"], "seqilu0"),
    (["-blasted_pc_type", "level_sgs"], "sgs"),
    (["-blasted_pc_type", "jacobi"], "jacobi"),
])
def test_pcshell_exact_types(tmp_path, opts, kind):
    """sequential / level-scheduled types, and asynchronous sweeps run to convergence, against the oracle's
    serial result (the reference at OMP_NUM_THREADS=1)."""
    m = matrix("baij")
    needs = [] if opts[1] in ("level_sgs", "jacobi") else ASYNC_OPTS
    rep, vecs, _ = run(tmp_path, opts + needs)
    check_common(rep, vecs, 4)
    r = W.rhs_vector(m["nbrows"] * 4)
    if kind == "seqilu0":
        f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
        want = O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)
        tol = 1e-10
    elif kind == "sgs":
        want = O.sgs_apply(m, O.jacobi_compute(m), r, 1, mode=O.GS_SERIAL)
        tol = 1e-12
        assert rep["richardson_callback"] == "1"
        assert rel(vecs["x"], O.sgs_relax(m, O.jacobi_compute(m), r, maxits=3, mode=O.GS_SERIAL)) < 1e-11
    else:
        d = O.jacobi_compute(m)
        want = O.jacobi_apply(m, d, r)
        tol = 1e-12
        assert rel(vecs["x"], O.jacobi_relax(m, d, r, maxits=3)[0]) < 1e-11
    assert rel(vecs["z"], want) < tol


@pytest.mark.parametrize("pc", [("asm", "shell"), ("ksp", "shell"), ("shell", None)])
def test_pcshell_tree_walk(tmp_path, pc):
    """setup_blasted_stack finds the PCSHELL under asm / ksp containers and at the top level
    (src/blasted_petsc.cpp:578-661)."""
    m = matrix("baij")
    rep, vecs, _ = run(tmp_path, ["-blasted_pc_type", "seqilu0", "-blasted_async_sweeps", "1,1"] + ASYNC_OPTS, pc=pc)
    check_common(rep, vecs, 4)
    r = W.rhs_vector(m["nbrows"] * 4)
    f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
    assert rel(vecs["z"], O.ilu0_apply(m, f, r, 1, mode=O.GS_SERIAL)) < 1e-10


@pytest.mark.parametrize("opts,vec_type", [
    (["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "10,15"], "seq"),     # ThreadedBSR4ILU0Colmajor's counts
    (["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "10,15"], "hip"),
    (["-blasted_pc_type", "sgs", "-blasted_async_sweeps", "1,15"], "seq"),
    (["-blasted_pc_type", "sapilu0", "-blasted_async_sweeps", "12,1"], "hip"),
])
def test_pcshell_solve_known_answer(tmp_path, opts, vec_type):
    """BiCGStab with PCApply as the preconditioner reaches the reference's shipped solution of 2dcyl1
    (tests/CMakeLists.txt:159-173 through the PETSc route, tests/CMakeLists.txt:222-269)."""
    rep, vecs, _ = run(tmp_path, opts + ASYNC_OPTS, vec_type=vec_type,
                       extra=["--b_file", os.path.join(G, "2dcyl1_b.pmat"), "--x_file", os.path.join(G, "2dcyl1_x.pmat"),
                              "--solver_tol", "1e-12", "--max_iter", "400"])
    check_common(rep, vecs, 4, homogeneous=opts[1] == "sgs")
    assert float(rep["solve_relres"]) < 1e-12 and float(rep["solve_error_l2"]) < 1e-8


def test_pcshell_rejects_bad_options(tmp_path):
    _, _, r = run(tmp_path, ["-blasted_pc_type", "bogus", "-blasted_async_sweeps", "1,1"] + ASYNC_OPTS, expect_rc=3)
    assert "Preconditioner type not available" in r.stderr
    # a PC tree without a shell: nothing to install, and the driver's first PCApply has no operator
    out = subprocess.run([DRIVER, "--mat_file", PMAT, "--", "-pc_type", "bjacobi", "-sub_pc_type", "none",
                          "-blasted_pc_type", "jacobi"], capture_output=True, text=True, timeout=120)
    assert "blasted_contexts = 0" in out.stdout and out.returncode != 0


def test_two_ranks_share_one_gpu(tmp_path):
    """VERDICT r03 item 8 (reference: one subdomain per rank, src/blasted_petsc.cpp:604-606; its own tests run
    `mpirun -n 3/4`, tests/CMakeLists.txt:213-220): two PCSHELL driver processes at the same time, launched as an MPI
    launcher would (OMPI_COMM_WORLD_LOCAL_RANK = 0 / 1).  On a one-GPU box both ranks map to device 0 (local rank modulo
    the device count): each builds its own operator there -- independent contexts, each holding its own bytes of HBM --
    and both match the oracle while the other one is alive."""
    m = matrix("baij")
    r = W.rhs_vector(m["nbrows"] * 4)
    opts = ["-blasted_pc_type", "ilu0", "-blasted_async_sweeps", "3,3"] + ASYNC_OPTS
    f = O.ilu0_factorize(m, None, 3, mode=O.JACOBI_SYNC, init=O.INIT_F_ORIGINAL)["iluvals"]
    want = O.ilu0_apply(m, f, r, 3, mode=O.JACOBI_SYNC, init=O.INIT_A_ZERO)
    procs = []
    for rank in (0, 1):
        out = str(tmp_path / ("rank%d" % rank))
        cmd = [DRIVER, "--mat_file", PMAT, "--mat_type", "baij", "--vec_type", "seq", "--out", out, "--hold_s", "3",
               "--", "-pc_type", "bjacobi", "-sub_pc_type", "shell"] + opts
        e = {k: v for k, v in os.environ.items() if k not in ("BLASTED_HIP_DEVICE", "BLASTED_HIP_SWEEP_MODE", "BLASTED_HIP_EXACT_APPLY")}
        e.update(SYNC, OMPI_COMM_WORLD_LOCAL_RANK=str(rank), OMPI_COMM_WORLD_RANK=str(rank), OMPI_COMM_WORLD_SIZE="2")
        procs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)))
    import torch
    ndev = torch.cuda.device_count()
    for rank, (out, pr) in enumerate(procs):
        so, se = pr.communicate(timeout=300)
        assert pr.returncode == 0, so[-3000:] + se[-2000:]
        rep = dict(line.split(" = ", 1) for line in so.splitlines() if " = " in line)
        assert rep["done"] == "1" and rep["blasted_contexts"] == "1"
        assert int(rep["hip_device"]) == rank % ndev          # one GPU: both on device 0
        assert int(rep["operator_device_bytes"]) > m["nnzb"] * 16 * 8    # its own factor (and more) in HBM
        assert rel(np.fromfile(out + "_z.bin", np.float64), want) < 1e-12
