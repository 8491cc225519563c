"""GPU tests of the host C++ layer through the native driver tests/cpp/testsolve (the counterpart of
the reference's tests/testsolve.cpp): the reference's own solve-level cases, tests/CMakeLists.txt:34-173,
with SRFactory-created operators whose compute/apply and the Krylov solver's SpMV run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "build", "testsolve")
G = os.path.join(ROOT, "tests", "golden")

# (name, matrix, extra args, solver_tol, test_tol, max_iter)
CASES = [
    ("SPDCSRJacobi", "msc00726", ["--preconditioner_type", "jacobi", "--mat_type", "csr"], 1e-10, 1e-9, 200),
    ("SPDCSRSGS", "msc00726", ["--preconditioner_type", "sgs", "--mat_type", "csr", "--apply_sweeps", "20"], 1e-10, 1e-10, 200),
    ("SPDCSRILU0", "msc00726", ["--preconditioner_type", "ilu0", "--mat_type", "csr", "--build_sweeps", "60", "--apply_sweeps", "30"], 1e-10, 1e-10, 200),
    ("SPDCSRSeqILU0", "msc00726", ["--preconditioner_type", "seqilu0", "--mat_type", "csr"], 1e-10, 1e-10, 200),
    ("CSRILU0", "2dcyl1", ["--preconditioner_type", "ilu0", "--mat_type", "csr", "--build_sweeps", "30", "--apply_sweeps", "30"], 1e-10, 1e-8, 200),
    ("BSR4JacobiRowmajor", "2dcyl1", ["--preconditioner_type", "jacobi", "--mat_type", "bsr", "--storage_order", "rowmajor"], 1e-10, 1e-8, 200),
    ("BSR4SGSRowmajor", "2dcyl1", ["--preconditioner_type", "sgs", "--mat_type", "bsr", "--storage_order", "rowmajor", "--apply_sweeps", "15"], 1e-10, 1e-8, 200),
    ("BSR4ILU0Rowmajor", "2dcyl1", ["--preconditioner_type", "ilu0", "--mat_type", "bsr", "--storage_order", "rowmajor", "--build_sweeps", "10", "--apply_sweeps", "15"], 1e-10, 1e-8, 200),
    ("BSR4BiCGStabNoneColmajor", "2dcyl1", ["--preconditioner_type", "none", "--mat_type", "bsr"], 1e-12, 1e-8, 1000),
    ("BSR4JacobiColmajor", "2dcyl1", ["--mat_type", "bsr"], 1e-10, 1e-8, 200),
    ("BSR4SGSColmajor", "2dcyl1", ["--preconditioner_type", "sgs", "--mat_type", "bsr", "--apply_sweeps", "15"], 1e-10, 1e-8, 200),
    ("ThreadedBSR4ILU0Colmajor", "2dcyl1", ["--preconditioner_type", "ilu0", "--mat_type", "bsr", "--build_sweeps", "10", "--apply_sweeps", "15"], 1e-10, 1e-8, 200),
    ("BSR4SeqILU0Colmajor", "2dcyl1", ["--preconditioner_type", "seqilu0", "--mat_type", "bsr"], 1e-10, 1e-8, 200),
    ("BSR4SapILU0Colmajor", "2dcyl1", ["--preconditioner_type", "sapilu0", "--mat_type", "bsr", "--build_sweeps", "12"], 1e-10, 1e-8, 200),
    ("BSR4RichardsonGS", "2dcyl1", ["--solver_type", "richardson", "--preconditioner_type", "gs", "--mat_type", "bsr", "--apply_sweeps", "10"], 1e-8, 1e-5, 2000),
    ("BSR4LevelSGSColmajor", "2dcyl1", ["--preconditioner_type", "level_sgs", "--mat_type", "bsr"], 1e-10, 1e-8, 200),
    ("CSRLevelSGS", "msc00726", ["--preconditioner_type", "level_sgs", "--mat_type", "csr"], 1e-10, 1e-10, 200),
    ("BSR4AsyncLevelILU0Colmajor", "2dcyl1", ["--preconditioner_type", "async_level_ilu0", "--mat_type", "bsr", "--build_sweeps", "10"], 1e-10, 1e-8, 200),
    ("BSR4AsyncLevelILU0Rowmajor", "2dcyl1", ["--preconditioner_type", "async_level_ilu0", "--mat_type", "bsr", "--storage_order", "rowmajor", "--build_sweeps", "10"], 1e-10, 1e-8, 200),
    ("CSRAsyncLevelILU0", "2dcyl1", ["--preconditioner_type", "async_level_ilu0", "--mat_type", "csr", "--build_sweeps", "30"], 1e-10, 1e-8, 200),
    ("BSR4RichardsonLevelSGS", "2dcyl1", ["--solver_type", "richardson", "--preconditioner_type", "level_sgs", "--mat_type", "bsr"], 1e-8, 1e-5, 2000),
    ("BSR4RichardsonSGS", "2dcyl1", ["--solver_type", "richardson", "--preconditioner_type", "sgs", "--mat_type", "bsr", "--apply_sweeps", "15"], 1e-8, 1e-5, 2000),
]


# the reference's cases that name `--fact_init_type init_zero` on a BLOCK ILU(0) (tests/CMakeLists.txt:104-111,
# 157-173) and the flexible-solver case (:122-129) -- run with exactly that initial guess
ZERO_INIT = {"BSR4ILU0Rowmajor", "ThreadedBSR4ILU0Colmajor", "BSR4SapILU0Colmajor", "BSR4AsyncLevelILU0Colmajor",
             "BSR4AsyncLevelILU0Rowmajor", "SPDCSRILU0", "CSRILU0", "CSRAsyncLevelILU0", "BSR4SeqILU0Colmajor",
             "BSR4SGSColmajor", "BSR4SGSRowmajor", "BSR4JacobiRowmajor", "BSR4JacobiColmajor", "SPDCSRSGS"}


def run_case(mat, extra, tol, testtol, maxiter, fact_init, env=None):
    args = [DRIVER, "--fact_init_type", fact_init, "--apply_init_type", "init_zero",
            "--mat_file", os.path.join(G, mat + ".mtx"), "--b_file", os.path.join(G, mat + "_b.mtx"),
            "--x_file", os.path.join(G, mat + "_x.mtx"), "--solver_tol", repr(tol),
            "--test_tol", repr(testtol), "--max_iter", str(maxiter)] + extra
    return subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)


@pytest.mark.parametrize("name,mat,extra,tol,testtol,maxiter", CASES, ids=[c[0] for c in CASES])
def test_native_solve(name, mat, extra, tol, testtol, maxiter):
    """Two runs per case.  (1) The reference's OWN solver_tol and max_iter (tests/CMakeLists.txt): the solver must
    converge within the reference's iteration budget.  The error against the shipped solution is residual-limited
    at that tolerance -- relres 1e-10 on msc00726 (cond 4e5) leaves ~1e-6, on 2dcyl1 (cond 8e3) a small multiple
    of test_tol that passes or fails on the last residual drop -- and the reference's own check is an assert()
    (tests/testsolve.cpp:115) that its Release build compiles out (CMakeLists.txt:327, -DNDEBUG): run (1) bounds the
    error by what the residual allows, (2) repeats the case with the residual tolerance four digits tighter and
    holds the error to the reference's test_tol (msc00726's shipped x only satisfies ||A x - b|| = 1.5e-6, which
    floors its error at ~2e-9: tests/test_oracle_pins.py::test_solve_known_answer)."""
    # (2e-9 with the oracle's serial preconditioners; the chaotic sweeps of the default mode end a converged solve
    # anywhere within a few 1e-10 of that: 2.4e-9 seen)
    floor = 4e-9 if mat == "msc00726" else 0.0
    fact_init = "init_zero" if name in ZERO_INIT else "init_original"
    loose = 1e-5 if mat == "msc00726" else 4 * testtol
    r = run_case(mat, extra, tol, loose, maxiter, fact_init)
    assert r.returncode == 0, r.stdout + r.stderr
    tight = tol if "Richardson" in name else tol * 1e-4
    r = run_case(mat, extra, tight, max(testtol, floor), 2 * maxiter, fact_init)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gcr_none_colmajor():
    """BSR4GCRNoneColmajor, tests/CMakeLists.txt:122-129, at the reference's parameters."""
    r = run_case("2dcyl1", ["--solver_type", "gcr", "--preconditioner_type", "none", "--mat_type", "bsr",
                            "--solver_restart", "200"], 1e-12, 1e-8, 1500, "init_zero")
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("prec,sweeps", [("ilu0", ["--build_sweeps", "10", "--apply_sweeps", "3"]),
                                         ("sgs", ["--apply_sweeps", "3"]),
                                         ("ilu0", ["--build_sweeps", "10", "--apply_sweeps", "1"])])
def test_gcr_with_asynchronous_sweeps(prec, sweeps):
    """The reference's flexible solver (GCR, tests/solvers.cpp:247-352) with the reference's chaotic in-place
    sweeps as the preconditioner (BLASTED_HIP_SWEEP_MODE=async) at FEW sweeps -- an operator that changes from
    one application to the next, which is what GCR is there for.  From a zero initial factor, as the
    reference's threaded case."""
    r = run_case("2dcyl1", ["--solver_type", "gcr", "--preconditioner_type", prec, "--mat_type", "bsr",
                            "--solver_restart", "30"] + sweeps, 1e-12, 1e-8, 600, "init_zero",
                 env=dict(os.environ, BLASTED_HIP_SWEEP_MODE="async"))
    assert r.returncode == 0, r.stdout + r.stderr


def test_exact_apply_switch():
    """BLASTED_HIP_EXACT_APPLY=1: the `ilu0` type with ONE asynchronous apply sweep -- far too few for this
    matrix -- converges like the exact variants, because its application runs as exact passes."""
    base = [DRIVER, "--fact_init_type", "init_original", "--apply_init_type", "init_zero",
            "--mat_file", os.path.join(G, "2dcyl1.mtx"), "--b_file", os.path.join(G, "2dcyl1_b.mtx"),
            "--x_file", os.path.join(G, "2dcyl1_x.mtx"), "--solver_tol", "1e-12", "--test_tol", "1e-8",
            "--max_iter", "60", "--preconditioner_type", "ilu0", "--mat_type", "bsr", "--build_sweeps", "30",
            "--apply_sweeps", "1"]
    env = dict(os.environ)
    for k in ("BLASTED_HIP_EXACT_APPLY", "BLASTED_HIP_SWEEP_MODE", "BLASTED_HIP_SYNC_SWEEPS"):
        env.pop(k, None)
    r0 = subprocess.run(base, capture_output=True, text=True, timeout=300, env=env)
    assert r0.returncode != 0   # one asynchronous sweep (the default mode) does not get there in 60 iterations
    ra = subprocess.run(base, capture_output=True, text=True, timeout=300, env=dict(env, BLASTED_HIP_SWEEP_MODE="deterministic"))
    assert ra.returncode != 0   # nor does one synchronous sweep
    for switch in ({"BLASTED_HIP_EXACT_APPLY": "1"}, {"BLASTED_HIP_SWEEP_MODE": "exact"}):
        r1 = subprocess.run(base, capture_output=True, text=True, timeout=300, env=dict(env, **switch))
        assert r1.returncode == 0, r1.stdout + r1.stderr


@pytest.mark.parametrize("sweep_mode", ["async", "deterministic"])
def test_native_solve_in_both_product_modes(sweep_mode):
    """the reference's threaded case (ThreadedBSR4ILU0Colmajor, 10 build / 15 apply sweeps) with the reference's
    chaotic sweeps (the default) and with deterministic (synchronous) ones"""
    args = [DRIVER, "--fact_init_type", "init_original", "--apply_init_type", "init_zero",
            "--mat_file", os.path.join(G, "2dcyl1.mtx"), "--b_file", os.path.join(G, "2dcyl1_b.mtx"),
            "--x_file", os.path.join(G, "2dcyl1_x.mtx"), "--solver_tol", "1e-12", "--test_tol", "1e-8",
            "--max_iter", "400", "--preconditioner_type", "ilu0", "--mat_type", "bsr", "--build_sweeps", "10",
            "--apply_sweeps", "15"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, BLASTED_HIP_SWEEP_MODE=sweep_mode))
    assert r.returncode == 0, r.stdout + r.stderr


def test_factory_rejects_out_of_scope_types():
    args = [DRIVER, "--preconditioner_type", "cscbgs", "--mat_type", "csr",
            "--mat_file", os.path.join(G, "2dcyl1.mtx"), "--b_file", os.path.join(G, "2dcyl1_b.mtx")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "outside the MI355X backend" in r.stderr
    args[2] = "bogus"
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "Preconditioner type not available" in r.stderr


DEVVEC = os.path.join(ROOT, "tests", "cpp", "build", "devvec")


@pytest.mark.parametrize("mat,mat_type,bs,prec,extra", [
    ("2dcyl1", "bsr", 4, "seqilu0", []),
    ("2dcyl1", "bsr", 4, "level_sgs", []),
    ("2dcyl1", "bsr", 4, "jacobi", []),
    ("msc00726", "csr", 1, "seqilu0", []),
    ("small_block3_matrix", "bsr", 3, "jacobi", []),
])
def test_device_vector_round_trip(tmp_path, mat, mat_type, bs, prec, extra):
    """SURVEY 8 row a15 (include/device_container.hpp:19-20 becomes the HIP buffer holder): a C++ program fills
    device_vector<double>s, to_device(), runs SRPreconditioner::apply_device and SRMatrixView::apply_device on
    device_data(), to_host() -- and the results equal the CPU oracle's (exact operator types: <= 1e-12) and, bit
    for bit, the host-vector members of the same operators.  Copy / move / resize of the mirror as documented."""
    import numpy as np
    import oracle as O
    from blasted_amd import mtxio, workloads as W
    out = str(tmp_path / "dv")
    env = dict(os.environ)
    for k in ("BLASTED_HIP_EXACT_APPLY", "BLASTED_HIP_SWEEP_MODE", "BLASTED_HIP_SYNC_SWEEPS"):
        env.pop(k, None)
    r = subprocess.run([DEVVEC, "--mat_file", os.path.join(G, mat + ".mtx"), "--mat_type", mat_type, "--block_size", str(bs),
                        "--preconditioner_type", prec, "--out", out] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = dict(line.split(" = ", 1) for line in r.stdout.splitlines() if " = " in line)
    assert rep["done"] == "1"
    for key in ("mirror_after_upload", "host_untouched_before_download", "copy_has_no_mirror", "copy_equals_host",
                "move_keeps_mirror", "moved_from_has_none", "second_upload_seen", "resized_upload_seen", "released"):
        assert rep[key] == "1", (key, r.stdout)
    assert rep["mirror_before_upload"] == "0"
    assert float(rep["max_abs_diff_host_vs_device_apply"]) == 0.0 and float(rep["max_abs_diff_host_vs_device_spmv"]) == 0.0
    m = mtxio.read_mtx_bsr(os.path.join(G, mat + ".mtx"), bs)
    n = m["nbrows"] * bs
    assert int(rep["n"]) == n
    rhs = W.rhs_vector(n)
    z, y = np.fromfile(out + "_z.bin"), np.fromfile(out + "_y.bin")

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    assert rel(y, O.spmv(m, rhs)) < 1e-13
    if prec == "seqilu0":
        f = O.ilu0_factorize(m, None, 1, mode=O.GS_SERIAL)["iluvals"]
        want = O.ilu0_apply(m, f, rhs, 1, mode=O.GS_SERIAL)
    elif prec == "jacobi":
        want = O.jacobi_apply(m, O.jacobi_compute(m), rhs)
    else:
        want = O.sgs_apply(m, O.jacobi_compute(m), rhs, 1, mode=O.GS_SERIAL)
    assert rel(z, want) < 1e-12
