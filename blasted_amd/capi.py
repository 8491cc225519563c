"""ctypes binding of include/blasted_hip.h (plumbing for tests and bench.py; the product boundary is
the C ABI itself).  There is no fallback: if the shared library or a GPU is missing, calls raise."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "lib", "libblasted_hip.so")
# measurement tools only (tools/probes/): BLASTED_HIP_PROBES=1 loads the -DBHIP_PROBES build of the same sources
if os.environ.get("BLASTED_HIP_PROBES") == "1":
    LIBPATH = os.path.join(HERE, "lib", "libblasted_hip_probes.so")

OK, EINVAL, ENODEV, ERUNTIME, ESTATE, ENOTIMPL = 0, 1, 2, 3, 4, 5
COLMAJOR, ROWMAJOR = 0, 1
HOST, DEVICE = 0, 1
ASYNC, JACOBI_SYNC, LEVEL, DETERMINISTIC = 0, 1, 2, 3
INIT_F_ZERO, INIT_F_ORIGINAL, INIT_F_SGS, INIT_F_NONE = 0, 1, 2, 3
INIT_A_ZERO, INIT_A_JACOBI, INIT_A_NONE = 0, 1, 2

# every symbol include/blasted_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "blasted_hip_last_error", "blasted_hip_device_count", "blasted_hip_create", "blasted_hip_destroy",
    "blasted_hip_synchronize", "blasted_hip_set_pattern", "blasted_hip_set_values",
    "blasted_hip_ilu0_positions", "blasted_hip_ilu0_positions_size", "blasted_hip_ilu0_get_positions",
    "blasted_hip_ilu0_factorize", "blasted_hip_ilu0_apply", "blasted_hip_jacobi_compute",
    "blasted_hip_jacobi_apply", "blasted_hip_sgs_apply", "blasted_hip_sgs_relax", "blasted_hip_spmv",
    "blasted_hip_gemv3", "blasted_hip_get_iluvals", "blasted_hip_get_dblocks", "blasted_hip_get_scale",
    "blasted_hip_get_ytemp", "blasted_hip_iluvals_device", "blasted_hip_set_timing",
    "blasted_hip_get_timing", "blasted_hip_buffer_alloc", "blasted_hip_buffer_free",
    "blasted_hip_buffer_upload", "blasted_hip_buffer_download", "blasted_hip_measure_read_stream",
    "blasted_hip_set_tuning",
    "blasted_hip_gs_relax", "blasted_hip_level_schedule", "blasted_hip_level_count",
    "blasted_hip_get_levels", "blasted_hip_level_stats", "blasted_hip_jacobi_relax",
    "blasted_hip_device_synchronize", "blasted_hip_memory_stats", "blasted_hip_host_register",
    "blasted_hip_host_unregister", "blasted_hip_placement_stats", "blasted_hip_placement_check",
]

_lib = None


class BlastedHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("blasted_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise ImportError("libblasted_hip.so is not built (run __graft_entry__.build()): " + LIBPATH)
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64 (soname
        # libamdhip64.so.7, requested by torch as "libamdhip64.so").  If it is loaded first, our library's
        # DT_NEEDED libamdhip64.so.7 resolves to that same copy; loaded the other way round the process
        # ends up with two runtimes and the second one sees no GPU.  So load torch's first when present.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIBPATH)
        _lib.blasted_hip_last_error.restype = C.c_char_p
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        _lib.blasted_hip_create.argtypes = [C.POINTER(vp), ci, vp, ci]
        _lib.blasted_hip_destroy.argtypes = [vp]
        _lib.blasted_hip_synchronize.argtypes = [vp]
        _lib.blasted_hip_set_pattern.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, ci]
        _lib.blasted_hip_set_values.argtypes = [vp, vp, ci]
        _lib.blasted_hip_ilu0_positions.argtypes = [vp]
        _lib.blasted_hip_ilu0_positions_size.argtypes = [vp, C.POINTER(C.c_long)]
        _lib.blasted_hip_ilu0_get_positions.argtypes = [vp, vp, vp, vp]
        _lib.blasted_hip_ilu0_factorize.argtypes = [vp, ci, ci, ci, ci, vp]
        _lib.blasted_hip_ilu0_apply.argtypes = [vp, vp, vp, ci, ci, ci, ci]
        _lib.blasted_hip_jacobi_compute.argtypes = [vp]
        _lib.blasted_hip_jacobi_apply.argtypes = [vp, vp, vp, ci]
        _lib.blasted_hip_jacobi_relax.argtypes = [vp, vp, vp, ci, ci, cd, cd, cd, C.POINTER(ci), ci]
        _lib.blasted_hip_sgs_apply.argtypes = [vp, vp, vp, ci, ci, ci, ci]
        _lib.blasted_hip_sgs_relax.argtypes = [vp, vp, vp, ci, ci, ci]
        _lib.blasted_hip_gs_relax.argtypes = [vp, vp, vp, ci, ci, ci]
        _lib.blasted_hip_level_schedule.argtypes = [vp]
        _lib.blasted_hip_level_count.argtypes = [vp, C.POINTER(ci)]
        _lib.blasted_hip_get_levels.argtypes = [vp, vp, vp, vp]
        _lib.blasted_hip_level_stats.argtypes = [vp, vp]
        _lib.blasted_hip_memory_stats.argtypes = [vp, vp]
        _lib.blasted_hip_placement_check.argtypes = [vp, vp, vp, vp]
        _lib.blasted_hip_host_register.argtypes = [vp, C.c_ulong]
        _lib.blasted_hip_host_unregister.argtypes = [vp]
        _lib.blasted_hip_spmv.argtypes = [vp, vp, vp, ci]
        _lib.blasted_hip_gemv3.argtypes = [vp, cd, vp, cd, vp, vp, ci]
        for nm in ("iluvals", "dblocks", "scale", "ytemp"):
            getattr(_lib, "blasted_hip_get_" + nm).argtypes = [vp, vp]
        _lib.blasted_hip_iluvals_device.argtypes = [vp, C.POINTER(vp)]
        _lib.blasted_hip_set_timing.argtypes = [vp, ci]
        _lib.blasted_hip_get_timing.argtypes = [vp, vp, ci]
        # measurements: BLASTED_HIP_TUNING="spec;spec;..." applies blasted_hip_set_tuning strings at load time
        for spec in filter(None, os.environ.get("BLASTED_HIP_TUNING", "").split(";")):
            if _lib.blasted_hip_set_tuning(spec.encode()) != OK:
                raise ValueError("BLASTED_HIP_TUNING: bad tuning string %r" % spec)
    return _lib


def _check(rc):
    if rc != OK:
        raise BlastedHipError(rc, lib().blasted_hip_last_error().decode())


def _ptr(a):
    """numpy array -> host pointer; torch tensor -> its data pointer; int -> as is; None -> NULL."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())


def _loc(a):
    if isinstance(a, np.ndarray):
        return HOST
    if hasattr(a, "is_cuda"):
        return DEVICE if a.is_cuda else HOST
    raise TypeError("expected a numpy array or a torch tensor")


def measure_read_stream(tensor, reps=10):
    """GB/s of a read-only stream over a CUDA tensor's storage (the practical HBM ceiling of this device)."""
    out = C.c_double(0.0)
    nbytes = tensor.numel() * tensor.element_size()
    _check(lib().blasted_hip_measure_read_stream(C.c_void_p(tensor.data_ptr()), C.c_ulong(nbytes), int(reps),
                                                 C.byref(out)))
    return out.value


def set_tuning(spec):
    """Process-wide kernel-variant selection (measurements only)."""
    _check(lib().blasted_hip_set_tuning(None if spec is None else spec.encode()))


def host_register(a):
    """Page-locks a numpy array in place (the caller keeps it alive until host_unregister)."""
    _check(lib().blasted_hip_host_register(a.ctypes.data, a.nbytes))


def host_unregister(a):
    _check(lib().blasted_hip_host_unregister(a.ctypes.data))


def placement_stats():
    """process-wide counters of the class-aware placement (include/blasted_hip.h)"""
    out = (C.c_long * 5)()
    _check(lib().blasted_hip_placement_stats(out))
    return dict(zip(("buffers", "pieces", "turned_down", "unchecked", "probes"), [int(v) for v in out]))


def device_count():
    return int(lib().blasted_hip_device_count())


class Prec:
    """One blasted_hip_prec object.  Vectors may be float64 numpy arrays (host path) or float64 CUDA
    torch tensors (device path, stream-ordered)."""

    def __init__(self, device=0, stream=0, own_stream=False):
        """stream: a hipStream_t handle as an int (0 = the null stream, which is torch's default
        stream); own_stream=True lets the object create a private stream instead."""
        self._h = C.c_void_p(0)
        _check(lib().blasted_hip_create(C.byref(self._h), int(device), C.c_void_p(int(stream or 0)),
                                        int(bool(own_stream))))
        self._keep = []
        self.n = 0

    def close(self):
        if self._h:
            lib().blasted_hip_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- matrix
    def set_matrix(self, m):
        """m: dict with nbrows, nnzb, bs, rowmajor, browptr, bcolind, diagind, vals (all numpy or all
        CUDA torch tensors)."""
        bp, bc, dg = m["browptr"], m["bcolind"], m["diagind"]
        if isinstance(bp, np.ndarray):
            bp, bc, dg = (np.ascontiguousarray(v, dtype=np.int32) for v in (bp, bc, dg))
        self._keep = [bp, bc, dg]
        _check(lib().blasted_hip_set_pattern(self._h, int(m["nbrows"]), int(m["nnzb"]), int(m["bs"]),
                                             ROWMAJOR if m.get("rowmajor") else COLMAJOR,
                                             _ptr(bp), _ptr(bc), _ptr(dg), _loc(bp)))
        self.n = int(m["nbrows"]) * int(m["bs"])
        self.nvals = int(m["nnzb"]) * int(m["bs"]) ** 2
        self.nnzb = int(m["nnzb"])
        self.bs = int(m["bs"])
        self.nbrows = int(m["nbrows"])
        self.set_values(m["vals"])

    def set_values(self, vals):
        if isinstance(vals, np.ndarray):
            vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._vals = vals
        _check(lib().blasted_hip_set_values(self._h, _ptr(vals), _loc(vals)))

    def synchronize(self):
        _check(lib().blasted_hip_synchronize(self._h))

    # -- ILU(0)
    def ilu0_positions(self):
        _check(lib().blasted_hip_ilu0_positions(self._h))
        n = C.c_long(0)
        _check(lib().blasted_hip_ilu0_positions_size(self._h, C.byref(n)))
        posptr = np.zeros(self.nnzb + 1, np.int32)
        lowerp = np.zeros(n.value, np.int32)
        upperp = np.zeros(n.value, np.int32)
        _check(lib().blasted_hip_ilu0_get_positions(self._h, _ptr(posptr), _ptr(lowerp), _ptr(upperp)))
        return posptr, lowerp, upperp

    def ilu0_positions_size(self):
        """Number of (lower, upper) position pairs (after ilu0_positions / ilu0_factorize)."""
        n = C.c_long(0)
        _check(lib().blasted_hip_ilu0_positions_size(self._h, C.byref(n)))
        return int(n.value)

    def ilu0_factorize(self, nbuildsweeps, init=INIT_F_ORIGINAL, usescale=False, mode=ASYNC,
                       compute_info=False):
        info = np.zeros(6) if compute_info else None
        _check(lib().blasted_hip_ilu0_factorize(self._h, int(nbuildsweeps), int(init), int(bool(usescale)),
                                                int(mode), _ptr(info)))
        return info

    def _vec_out(self, like):
        if isinstance(like, np.ndarray):
            return np.zeros(self.n)
        import torch
        return torch.zeros(self.n, dtype=torch.float64, device=like.device)

    @staticmethod
    def _prep(v):
        if isinstance(v, np.ndarray):
            return np.ascontiguousarray(v, dtype=np.float64)
        return v

    def ilu0_apply(self, r, napplysweeps, init=INIT_A_ZERO, mode=ASYNC, out=None):
        r = self._prep(r)
        z = self._vec_out(r) if out is None else out
        _check(lib().blasted_hip_ilu0_apply(self._h, _ptr(r), _ptr(z), int(napplysweeps), int(init),
                                            int(mode), _loc(r)))
        return z

    # -- Jacobi / SGS
    def jacobi_compute(self):
        _check(lib().blasted_hip_jacobi_compute(self._h))

    def jacobi_apply(self, r, out=None):
        r = self._prep(r)
        z = self._vec_out(r) if out is None else out
        _check(lib().blasted_hip_jacobi_apply(self._h, _ptr(r), _ptr(z), _loc(r)))
        return z

    def jacobi_relax(self, b, x, maxits, check_tol=False, rtol=0.0, atol=0.0, dtol=1e300):
        """Synchronous Jacobi relaxation steps; x is updated in place; returns the number of steps taken."""
        b = self._prep(b)
        n = C.c_int(0)
        _check(lib().blasted_hip_jacobi_relax(self._h, _ptr(b), _ptr(x), int(maxits), int(bool(check_tol)),
                                              float(rtol), float(atol), float(dtol), C.byref(n), _loc(b)))
        return n.value

    def sgs_apply(self, r, napplysweeps, init=INIT_A_ZERO, mode=ASYNC, out=None):
        r = self._prep(r)
        z = self._vec_out(r) if out is None else out
        _check(lib().blasted_hip_sgs_apply(self._h, _ptr(r), _ptr(z), int(napplysweeps), int(init),
                                           int(mode), _loc(r)))
        return z

    def sgs_relax(self, b, x, maxits, mode=ASYNC):
        """x is updated in place and returned."""
        b = self._prep(b)
        _check(lib().blasted_hip_sgs_relax(self._h, _ptr(b), _ptr(x), int(maxits), int(mode), _loc(b)))
        return x

    def gs_relax(self, b, x, nsweeps, mode=ASYNC):
        """Forward (ascending) relaxation sweeps; x is updated in place and returned."""
        b = self._prep(b)
        _check(lib().blasted_hip_gs_relax(self._h, _ptr(b), _ptr(x), int(nsweeps), int(mode), _loc(b)))
        return x

    # -- level schedule
    def level_count(self):
        n = C.c_int(0)
        _check(lib().blasted_hip_level_count(self._h, C.byref(n)))
        return n.value

    def level_stats(self):
        """-> dict(levels, build_passes, syncfree_passes, syncfree_aborts)"""
        out = (C.c_long * 4)()
        _check(lib().blasted_hip_level_stats(self._h, C.cast(out, C.c_void_p)))
        return dict(zip(("levels", "build_passes", "syncfree_passes", "syncfree_aborts"), [int(v) for v in out]))

    def memory_stats(self):
        out = (C.c_long * 4)()
        _check(lib().blasted_hip_memory_stats(self._h, out))
        return {"bytes": out[0], "peak_bytes": out[1], "derived_copies": out[2], "pinned_host_bytes": out[3]}

    def get_levels(self):
        """-> (level_of_row[nbrows], rows_by_level[nbrows], level_ptr[nlevels+1]) int32 numpy arrays."""
        nl = self.level_count()
        lv = np.zeros(self.nbrows, dtype=np.int32)
        rows = np.zeros(self.nbrows, dtype=np.int32)
        ptr = np.zeros(nl + 1, dtype=np.int32)
        _check(lib().blasted_hip_get_levels(self._h, lv.ctypes.data, rows.ctypes.data, ptr.ctypes.data))
        return lv, rows, ptr

    def placement_check(self, r, z):
        """address classes of the ILU application's buffers against device vectors r, z (include/blasted_hip.h)"""
        out = (C.c_long * 8)()
        _check(lib().blasted_hip_placement_check(self._h, _ptr(r), _ptr(z), out))
        keys = ("lower_pieces", "lower_in_ytemp_class", "lower_in_r_class", "upper_pieces", "upper_in_z_class",
                "upper_in_ytemp_class", "ytemp_in_r_class", "ytemp_in_z_class")
        return dict(zip(keys, [int(v) for v in out]))

    # -- SpMV
    def spmv(self, x, out=None):
        x = self._prep(x)
        y = self._vec_out(x) if out is None else out
        _check(lib().blasted_hip_spmv(self._h, _ptr(x), _ptr(y), _loc(x)))
        return y

    def gemv3(self, a, x, b, y, out=None):
        x, y = self._prep(x), self._prep(y)
        z = self._vec_out(x) if out is None else out
        _check(lib().blasted_hip_gemv3(self._h, float(a), _ptr(x), float(b), _ptr(y), _ptr(z), _loc(x)))
        return z

    # -- state read-back
    def get_iluvals(self):
        out = np.zeros(self.nvals)
        _check(lib().blasted_hip_get_iluvals(self._h, _ptr(out)))
        return out

    def get_dblocks(self):
        out = np.zeros(self.nbrows * self.bs * self.bs)
        _check(lib().blasted_hip_get_dblocks(self._h, _ptr(out)))
        return out

    def get_scale(self):
        out = np.zeros(self.n)
        _check(lib().blasted_hip_get_scale(self._h, _ptr(out)))
        return out

    def get_ytemp(self):
        out = np.zeros(self.n)
        _check(lib().blasted_hip_get_ytemp(self._h, _ptr(out)))
        return out

    def iluvals_device_ptr(self):
        p = C.c_void_p(0)
        _check(lib().blasted_hip_iluvals_device(self._h, C.byref(p)))
        return p.value

    # -- timing
    def set_timing(self, enable):
        _check(lib().blasted_hip_set_timing(self._h, int(bool(enable))))

    def get_timing(self, reset=True):
        out = np.zeros(6)
        _check(lib().blasted_hip_get_timing(self._h, _ptr(out), int(bool(reset))))
        return {"lower_ms": out[0], "lower_launches": out[1], "upper_ms": out[2],
                "upper_launches": out[3], "other_ms": out[4], "other_launches": out[5]}
