"""Matrix-Market fixtures -> sparse-row storage (host side, numpy).

The reference loads its fixtures with COOMatrix::readMatrixMarket + convertToBSR
(src/coomatrix.cpp:186-403).  Unlike the reference (blocks ordered by first appearance),
block columns are SORTED here, because the ILU position lists and the "lower = before diagind" rule
assume ascending columns (src/ilu_pattern.cpp:51,67).
"""
import numpy as np


def read_mtx_coo(path):
    """'matrix coordinate real general' -> (nrows, ncols, rows, cols, vals), 0-based."""
    with open(path) as f:
        header = f.readline().split()
        if len(header) < 5 or header[2] != "coordinate" or header[4] != "general":
            raise ValueError("can only read general coordinate matrices: " + path)
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nr, nc, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2)
    if data.shape[0] != nnz:
        raise ValueError("nnz mismatch in " + path)
    return nr, nc, data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1, data[:, 2].copy()


def read_mtx_dense(path):
    """'matrix array real general' n x 1 -> vector (readDenseMatrixMarket, src/coomatrix.cpp)."""
    with open(path) as f:
        header = f.readline().split()
        if len(header) < 3 or header[2] != "array":
            raise ValueError("not a dense matrix-market file: " + path)
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nr, nc = (int(t) for t in line.split())
        v = np.loadtxt(f, dtype=np.float64).reshape(-1)
    if v.size != nr * nc:
        raise ValueError("size mismatch in " + path)
    return v


def coo_to_bsr(nrows, rows, cols, vals, bs, rowmajor=False):
    """COO -> BSR dict with sorted block columns; every block-row must hold its diagonal block."""
    if nrows % bs != 0:
        raise ValueError("matrix size not divisible by block size")
    nbrows = nrows // bs
    br, bc = rows // bs, cols // bs
    key = br * nbrows + bc
    ukeys, inv = np.unique(key, return_inverse=True)
    nnzb = ukeys.size
    bcolind = (ukeys % nbrows).astype(np.int32)
    brow_of = (ukeys // nbrows).astype(np.int64)
    browptr = np.zeros(nbrows + 1, dtype=np.int32)
    np.add.at(browptr, brow_of + 1, 1)
    browptr = np.cumsum(browptr).astype(np.int32)
    r, c = rows % bs, cols % bs
    inblk = (r * bs + c) if rowmajor else (c * bs + r)
    v = np.zeros(nnzb * bs * bs)
    np.add.at(v, inv * (bs * bs) + inblk, vals)
    isdiag = np.nonzero(brow_of == bcolind)[0]
    if isdiag.size != nbrows:
        raise ValueError("some block-row has no diagonal block")
    diagind = isdiag.astype(np.int32)
    return {"nbrows": nbrows, "nnzb": nnzb, "bs": bs, "rowmajor": bool(rowmajor),
            "browptr": browptr, "bcolind": bcolind, "diagind": diagind, "vals": v}


def read_mtx_bsr(path, bs, rowmajor=False):
    nr, nc, rows, cols, vals = read_mtx_coo(path)
    if nr != nc:
        raise ValueError("square matrix required")
    return coo_to_bsr(nr, rows, cols, vals, bs, rowmajor)


def convert_layout(m, rowmajor):
    """Same matrix with the other in-block layout."""
    if bool(m["rowmajor"]) == bool(rowmajor):
        return dict(m)
    bs = m["bs"]
    v = m["vals"].reshape(-1, bs, bs).transpose(0, 2, 1).reshape(-1).copy()
    out = dict(m)
    out["vals"] = v
    out["rowmajor"] = bool(rowmajor)
    return out


def bsr_to_scipy(m):
    import scipy.sparse as sp
    bs = m["bs"]
    blocks = m["vals"].reshape(-1, bs, bs)
    if not m["rowmajor"]:
        blocks = blocks.transpose(0, 2, 1)
    n = m["nbrows"] * bs
    return sp.bsr_matrix((blocks, m["bcolind"], m["browptr"]), shape=(n, n)).tocsr()
