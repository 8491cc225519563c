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


# ----------------------------------------------------------------------------- PETSc binary files
# What the reference's PETSc drivers load (MatLoad / VecLoad in tests/testutils.cpp and the
# tests/input/**/*.pmat fixtures): big-endian; Mat = {1211216, rows, cols, nnz, rowlengths[rows],
# colidx[nnz], values[nnz]}, Vec = {1211214, n, values[n]}.

PETSC_MAT_CLASSID, PETSC_VEC_CLASSID = 1211216, 1211214


def read_petsc_mat_coo(path):
    """PETSc binary AIJ matrix -> (nrows, ncols, rows, cols, vals), 0-based."""
    raw = np.fromfile(path, dtype=np.uint8)
    head = raw[:16].view(">i4")
    if int(head[0]) != PETSC_MAT_CLASSID:
        raise ValueError("not a PETSc binary matrix: " + path)
    nr, nc, nnz = int(head[1]), int(head[2]), int(head[3])
    if nnz < 0:
        raise ValueError("dense PETSc binary matrices are not supported: " + path)
    o = 16
    lens = raw[o:o + 4 * nr].view(">i4").astype(np.int64)
    o += 4 * nr
    cols = raw[o:o + 4 * nnz].view(">i4").astype(np.int64)
    o += 4 * nnz
    vals = raw[o:o + 8 * nnz].view(">f8").astype(np.float64)
    if int(lens.sum()) != nnz or o + 8 * nnz != raw.size:
        raise ValueError("inconsistent PETSc binary matrix: " + path)
    rows = np.repeat(np.arange(nr, dtype=np.int64), lens)
    return nr, nc, rows, cols, vals


def read_petsc_vec(path):
    raw = np.fromfile(path, dtype=np.uint8)
    head = raw[:8].view(">i4")
    if int(head[0]) != PETSC_VEC_CLASSID:
        raise ValueError("not a PETSc binary vector: " + path)
    n = int(head[1])
    if raw.size != 8 + 8 * n:
        raise ValueError("inconsistent PETSc binary vector: " + path)
    return raw[8:].view(">f8").astype(np.float64)


def read_petsc_bsr(path, bs=None, rowmajor=False):
    """PETSc binary matrix -> BSR dict.  bs = None reads the block size from the `.info` file next to it
    (`-matload_block_size N`, what MatLoad honours), 1 if there is none."""
    if bs is None:
        bs = 1
        try:
            for tok in open(path + ".info").read().split("\n"):
                t = tok.split()
                if len(t) == 2 and t[0] == "-matload_block_size":
                    bs = int(t[1])
        except OSError:
            pass
    nr, nc, rows, cols, vals = read_petsc_mat_coo(path)
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
