"""ctypes front-end of liboracle (oracle/blasted_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Every function takes/returns numpy arrays; a matrix is the dict produced by
blasted_amd.workloads / blasted_amd.mtxio:
    {"nbrows", "nnzb", "bs", "rowmajor", "browptr", "bcolind", "diagind", "vals"}
"""
import ctypes as C
import os
import subprocess

import numpy as np

GS_SERIAL, JACOBI_SYNC, ASYNC_OMP = 0, 1, 2
INIT_F_ZERO, INIT_F_ORIGINAL, INIT_F_SGS, INIT_F_NONE = 0, 1, 2, 3
INIT_A_ZERO, INIT_A_JACOBI, INIT_A_NONE = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libblasted_oracle.so")
_lib = None


class _Bsr(C.Structure):
    _fields_ = [("nbrows", C.c_int), ("nnzb", C.c_int), ("bs", C.c_int), ("rowmajor", C.c_int),
                ("browptr", C.c_void_p), ("bcolind", C.c_void_p), ("diagind", C.c_void_p),
                ("vals", C.c_void_p)]


def build(force=False):
    """Compile the C restatement (gcc); building the checker is not using it."""
    src = os.path.join(_HERE, "blasted_oracle.c")
    if force or not os.path.exists(_LIBPATH) or os.path.getmtime(_LIBPATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIBPATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            build()
        _lib = C.CDLL(_LIBPATH)
        _lib.orc_ilu_positions_count.restype = C.c_long
        _lib.orc_ilu0_nonlinear_res.restype = C.c_double
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def cpu_budget():
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota (cpu.max) -- a
    container that sees 256 hardware threads may be granted 16 CPUs' worth of time."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


class _Mat:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, m, vals=None):
        self.browptr = _i32(m["browptr"])
        self.bcolind = _i32(m["bcolind"])
        self.diagind = _i32(m["diagind"])
        self.vals = _f64(m["vals"] if vals is None else vals)
        self.bs = int(m["bs"])
        self.nbrows = int(m["nbrows"])
        self.n = self.nbrows * self.bs
        self.nvals = int(self.browptr[-1]) * self.bs * self.bs
        assert self.vals.size == self.nvals
        self.c = _Bsr(self.nbrows, int(m["nnzb"]), self.bs, int(bool(m.get("rowmajor", False))),
                      self.browptr.ctypes.data, self.bcolind.ctypes.data, self.diagind.ctypes.data,
                      self.vals.ctypes.data)

    @property
    def ref(self):
        return C.byref(self.c)


def ilu_positions(m):
    """-> (posptr[nnzb+1], lowerp[P], upperp[P]) int32; src/ilu_pattern.cpp:32-163."""
    M = _Mat(m)
    posptr = np.zeros(int(M.browptr[-1]) + 1, dtype=np.int32)
    total = lib().orc_ilu_positions_count(M.ref, _ptr(posptr))
    lowerp = np.zeros(total, dtype=np.int32)
    upperp = np.zeros(total, dtype=np.int32)
    lib().orc_ilu_positions_fill(M.ref, _ptr(posptr), _ptr(lowerp), _ptr(upperp))
    return posptr, lowerp, upperp


def scaling_vector(m):
    M = _Mat(m)
    s = np.zeros(M.n)
    lib().orc_scaling_vector(M.ref, _ptr(s))
    return s


def block_inverse(a, bs, rowmajor=False):
    a = _f64(a).reshape(-1)
    out = np.zeros(bs * bs)
    rc = lib().orc_block_inverse(bs, int(rowmajor), _ptr(a), _ptr(out))
    return out, rc


def ilu0_factorize(m, plist=None, nbuildsweeps=1, chunk=256, mode=GS_SERIAL, init=INIT_F_ORIGINAL,
                   usescale=False, iluvals=None, compute_info=False, invert_diag=True):
    """-> dict(iluvals, scale, precinfo).  iluvals (optional) is the warm start for INIT_F_NONE.
    invert_diag=False leaves the diagonal blocks as the sweeps iterate on them (the reference's driver inverts them
    at the end of a block factorisation, src/async_blockilu_factor.cpp:143-146; its fixed-point tests do not)."""
    M = _Mat(m)
    if plist is None:
        plist = ilu_positions(m)
    posptr, lowerp, upperp = (_i32(p) for p in plist)
    ilu = np.zeros(M.nvals) if iluvals is None else _f64(iluvals).copy()
    scale = np.zeros(M.n) if usescale else None
    info = np.zeros(6) if compute_info else None
    rc = lib().orc_ilu0_factorize(M.ref, _ptr(posptr), _ptr(lowerp), _ptr(upperp), int(nbuildsweeps),
                                  int(chunk), int(mode), int(init) | (0 if invert_diag else 256), _ptr(ilu), _ptr(scale),
                                  _ptr(info))
    if rc != 0:
        raise ValueError("orc_ilu0_factorize: invalid argument")
    return {"iluvals": ilu, "scale": scale, "precinfo": info}


def ilu0_apply(m, iluvals, r, napplysweeps=1, chunk=256, mode=GS_SERIAL, init=INIT_A_ZERO, scale=None,
               return_y=False):
    M = _Mat(m)
    ilu = _f64(iluvals)
    r = _f64(r)
    z = np.zeros(M.n)
    y = np.zeros(M.n)
    sc = _f64(scale) if scale is not None else None
    rc = lib().orc_ilu0_apply(M.ref, _ptr(ilu), _ptr(sc), _ptr(y), int(napplysweeps), int(chunk),
                              int(mode), int(init), _ptr(r), _ptr(z))
    if rc != 0:
        raise RuntimeError("scalar_ilu0_apply: Invalid init type!")
    return (z, y) if return_y else z


def jacobi_compute(m):
    M = _Mat(m)
    d = np.zeros(M.nbrows * M.bs * M.bs)
    lib().orc_jacobi_compute(M.ref, _ptr(d))
    return d


def jacobi_apply(m, dblocks, r):
    M = _Mat(m)
    d, r = _f64(dblocks), _f64(r)
    z = np.zeros(M.n)
    lib().orc_jacobi_apply(M.ref, _ptr(d), _ptr(r), _ptr(z))
    return z


def jacobi_relax(m, dblocks, b, x0=None, maxits=1, ctol=False, rtol=0.0, atol=0.0, dtol=1e300):
    """-> (x, steps taken); src/solverops_jacobi.cpp:66-119."""
    M = _Mat(m)
    d, b = _f64(dblocks), _f64(b)
    x = np.zeros(M.n) if x0 is None else _f64(x0).copy()
    lib().orc_jacobi_relax.restype = C.c_int
    lib().orc_jacobi_relax.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_double, C.c_void_p, C.c_void_p]
    steps = lib().orc_jacobi_relax(M.ref, _ptr(d), int(maxits), int(bool(ctol)), float(rtol), float(atol),
                                   float(dtol), _ptr(b), _ptr(x))
    return x, int(steps)


def sgs_apply(m, dblocks, r, napplysweeps=1, chunk=256, mode=GS_SERIAL, init=INIT_A_ZERO, z0=None,
              y0=None, return_y=False):
    M = _Mat(m)
    d, r = _f64(dblocks), _f64(r)
    z = np.zeros(M.n) if z0 is None else _f64(z0).copy()
    y = np.zeros(M.n) if y0 is None else _f64(y0).copy()
    lib().orc_sgs_apply(M.ref, _ptr(d), _ptr(y), int(napplysweeps), int(chunk), int(mode), int(init),
                        _ptr(r), _ptr(z))
    return (z, y) if return_y else z


def sgs_relax(m, dblocks, b, x0=None, maxits=1, chunk=256, mode=GS_SERIAL):
    M = _Mat(m)
    d, b = _f64(dblocks), _f64(b)
    x = np.zeros(M.n) if x0 is None else _f64(x0).copy()
    lib().orc_sgs_relax(M.ref, _ptr(d), int(maxits), int(chunk), int(mode), _ptr(b), _ptr(x))
    return x


def gs_relax(m, dblocks, b, x0=None, nsweeps=1, chunk=256, mode=GS_SERIAL):
    M = _Mat(m)
    d, b = _f64(dblocks), _f64(b)
    x = np.zeros(M.n) if x0 is None else _f64(x0).copy()
    lib().orc_gs_relax(M.ref, _ptr(d), int(nsweeps), int(chunk), int(mode), _ptr(b), _ptr(x))
    return x


def compute_levels(m):
    """computeLevels, src/levelschedule.cpp:13-72 -> int32 boundaries [nlevels+1]; ValueError where the
    reference throws "Faulty dependency list!"."""
    M = _Mat(m)
    lv = np.zeros(M.nbrows + 2, dtype=np.int32)
    nl = lib().orc_compute_levels(M.ref, _ptr(lv))
    if nl < 0:
        raise ValueError("Faulty dependency list!")
    return lv[:nl + 1].copy()


def level_ilu0_apply(m, iluvals, levels, r, scale=None):
    M = _Mat(m)
    ilu, r, lv = _f64(iluvals), _f64(r), _i32(levels)
    z = np.zeros(M.n)
    y = np.zeros(M.n)
    sc = _f64(scale) if scale is not None else None
    lib().orc_level_ilu0_apply(M.ref, _ptr(ilu), _ptr(sc), _ptr(y), _ptr(lv), int(lv.size - 1), _ptr(r), _ptr(z))
    return z


def level_sgs_apply(m, dblocks, levels, r):
    M = _Mat(m)
    d, r, lv = _f64(dblocks), _f64(r), _i32(levels)
    z = np.zeros(M.n)
    y = np.zeros(M.n)
    lib().orc_level_sgs_apply(M.ref, _ptr(d), _ptr(y), _ptr(lv), int(lv.size - 1), _ptr(r), _ptr(z))
    return z


def level_sgs_relax(m, dblocks, levels, b, x0=None, maxits=1):
    M = _Mat(m)
    d, b, lv = _f64(dblocks), _f64(b), _i32(levels)
    x = np.zeros(M.n) if x0 is None else _f64(x0).copy()
    lib().orc_level_sgs_relax(M.ref, _ptr(d), _ptr(lv), int(lv.size - 1), int(maxits), _ptr(b), _ptr(x))
    return x


def spmv(m, x):
    M = _Mat(m)
    x = _f64(x)
    y = np.zeros(M.n)
    lib().orc_spmv(M.ref, _ptr(x), _ptr(y))
    return y


def gemv3(m, a, x, b, y):
    M = _Mat(m)
    x, y = _f64(x), _f64(y)
    z = np.zeros(M.n)
    lib().orc_gemv3(M.ref, C.c_double(a), _ptr(x), C.c_double(b), _ptr(y), _ptr(z))
    return z


def ilu0_nonlinear_res(m, plist, iluvals, scale=None):
    M = _Mat(m)
    posptr, lowerp, upperp = (_i32(p) for p in plist)
    ilu = _f64(iluvals)
    sc = _f64(scale) if scale is not None else None
    return float(lib().orc_ilu0_nonlinear_res(M.ref, _ptr(posptr), _ptr(lowerp), _ptr(upperp), _ptr(sc),
                                              _ptr(ilu)))


def diag_dominance(m, factor_vals):
    M = _Mat(m, vals=factor_vals)
    out = np.zeros(4)
    lib().orc_diag_dominance(M.ref, _ptr(out))
    return out


# --------------------------------------------------------------------- timing helper (cpu_baseline)

def time_op(op, m, r, sweeps, chunk, repeats, iluvals=None, dblocks=None, plist=None):
    """Times `repeats` calls of one operator in the reference's threaded form (ASYNC_OMP); returns the
    minimum seconds per call.  op: ilu_apply | sgs_apply | sgs_relax | spmv | factor."""
    import time
    M = _Mat(m)
    r = _f64(r)
    z = np.zeros(M.n)
    y = np.zeros(M.n)
    L = lib()
    if op == "ilu_apply":
        ilu = _f64(iluvals)
        call = lambda: L.orc_ilu0_apply(M.ref, _ptr(ilu), _ptr(None), _ptr(y), int(sweeps), int(chunk),
                                        ASYNC_OMP, INIT_A_ZERO, _ptr(r), _ptr(z))
    elif op == "sgs_apply":
        d = _f64(dblocks)
        call = lambda: L.orc_sgs_apply(M.ref, _ptr(d), _ptr(y), int(sweeps), int(chunk), ASYNC_OMP,
                                       INIT_A_ZERO, _ptr(r), _ptr(z))
    elif op == "sgs_relax":
        d = _f64(dblocks)
        call = lambda: L.orc_sgs_relax(M.ref, _ptr(d), int(sweeps), int(chunk), ASYNC_OMP, _ptr(r), _ptr(z))
    elif op == "spmv":
        call = lambda: L.orc_spmv(M.ref, _ptr(r), _ptr(z))
    elif op == "factor":
        posptr, lowerp, upperp = (_i32(p) for p in plist)
        ilu = np.zeros(M.nvals)
        call = lambda: L.orc_ilu0_factorize(M.ref, _ptr(posptr), _ptr(lowerp), _ptr(upperp), int(sweeps),
                                            int(chunk), ASYNC_OMP, INIT_F_ORIGINAL, _ptr(ilu), _ptr(None),
                                            _ptr(None))
    else:
        raise ValueError(op)
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        call()
        best = min(best, time.perf_counter() - t0)
    return best


def time_ilu0_apply(m, iluvals, r, napplysweeps, chunk, repeats):
    """Times `repeats` threaded apply calls (reference loop nest); returns seconds per call (min)."""
    import time
    M = _Mat(m)
    ilu, r = _f64(iluvals), _f64(r)
    z = np.zeros(M.n)
    y = np.zeros(M.n)
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        lib().orc_ilu0_apply(M.ref, _ptr(ilu), _ptr(None), _ptr(y), int(napplysweeps), int(chunk),
                             ASYNC_OMP, INIT_A_ZERO, _ptr(r), _ptr(z))
        best = min(best, time.perf_counter() - t0)
    return best
