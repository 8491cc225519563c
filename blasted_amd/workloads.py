"""Deterministic synthetic inputs of the BASELINE.json configs (host: numpy; device: torch).

* scalar 3-D Poisson: restates computeLHS of the reference's test problem
  (tests/poisson3d-fd/poisson3d_fd.cpp:83-149, grid tests/poisson3d-fd/cartmesh.cpp:145-176):
  npdim points per axis including the boundary, unknowns = (npdim-2)^3 interior points, row
  r = i + n (j + n k), columns in the order k-1, j-1, i-1, diag, i+1, j+1, k+1 (ascending).
* block inflation to bs in {4,5,8,...}: the reference has no block Poisson generator; this is the
  definition of SURVEY.md section 8(d): block(i,j) = a_ij * M_t, t = stencil slot 0..6,
  M_t(r,c) = (1 + 0.1 r) delta_rc + eps_t(r,c),
  eps_diag(r,c) = 0.05*((7r+3c) mod 5 - 2)/6, eps_off(r,c) = 0.02*((r+2c+t) mod 3 - 1)   (r != c).
* vectors: r_i = sin(0.37 i) + 1.1.
"""
import numpy as np

PI = 3.141592653589793238  # tests/poisson3d-fd/cartmesh.hpp:20
NSLOT = 7
DIAG_SLOT = 3


def grid_coords(npdim, grid="chebyshev", a=-1.0, b=1.0):
    i = np.arange(npdim, dtype=np.float64)
    if grid == "chebyshev":
        theta = PI / (npdim - 1)
        return (b + a) * 0.5 + (b - a) * 0.5 * np.cos(PI - i * theta)
    if grid == "uniform":
        return a + (b - a) * i / (npdim - 1)
    raise ValueError("grid must be 'chebyshev' or 'uniform'")


def _axis_coeffs(X):
    n = X.size - 2
    I = np.arange(1, n + 1)
    lo = -1.0 / ((X[I] - X[I - 1]) * 0.5 * (X[I + 1] - X[I - 1]))
    hi = -1.0 / ((X[I + 1] - X[I]) * 0.5 * (X[I + 1] - X[I - 1]))
    dg = 2.0 / (X[I + 1] - X[I - 1]) * (1.0 / (X[I + 1] - X[I]) + 1.0 / (X[I] - X[I - 1]))
    return lo, hi, dg


def slot_matrices(bs, rowmajor=False):
    """The 7 inflation matrices M_t, each flattened in block storage order -> [7, bs*bs]."""
    M = np.zeros((NSLOT, bs, bs))
    r = np.arange(bs)[:, None]
    c = np.arange(bs)[None, :]
    for t in range(NSLOT):
        if t == DIAG_SLOT:
            eps = 0.05 * (((7 * r + 3 * c) % 5) - 2) / 6.0
        else:
            eps = 0.02 * (((r + 2 * c + t) % 3) - 1)
        Mt = np.where(r == c, 1.0 + 0.1 * r, eps)
        M[t] = Mt
    if bs == 1:
        M[:] = 1.0
    if not rowmajor:
        M = M.transpose(0, 2, 1)
    return np.ascontiguousarray(M.reshape(NSLOT, bs * bs))


def poisson3d(npdim, bs=1, grid="chebyshev", rowmajor=False):
    """-> matrix dict (+ "slot": stencil slot of every stored block)."""
    n = npdim - 2
    if n < 1:
        raise ValueError("need at least 3 points per axis")
    X = grid_coords(npdim, grid)
    lo, hi, dg = _axis_coeffs(X)
    nb = n * n * n
    idx = np.arange(nb, dtype=np.int64)
    i = idx % n
    j = (idx // n) % n
    k = idx // (n * n)
    valid = np.stack([k > 0, j > 0, i > 0, np.ones(nb, bool), i < n - 1, j < n - 1, k < n - 1], axis=1)
    offs = np.array([-n * n, -n, -1, 0, 1, n, n * n], dtype=np.int64)
    cols = idx[:, None] + offs[None, :]
    diag = dg[i] + dg[j]
    diag = diag + dg[k]
    coef = np.stack([lo[k], lo[j], lo[i], diag, hi[i], hi[j], hi[k]], axis=1)
    slots = np.broadcast_to(np.arange(NSLOT, dtype=np.int8)[None, :], (nb, NSLOT))
    bcolind = cols[valid].astype(np.int32)
    a = coef[valid]
    slot = slots[valid]
    browptr = np.zeros(nb + 1, dtype=np.int64)
    browptr[1:] = np.cumsum(valid.sum(axis=1))
    diagind = (browptr[:-1] + valid[:, :3].sum(axis=1)).astype(np.int32)
    M = slot_matrices(bs, rowmajor)
    vals = (a[:, None] * M[slot]).reshape(-1)
    return {"nbrows": nb, "nnzb": int(bcolind.size), "bs": bs, "rowmajor": bool(rowmajor),
            "browptr": browptr.astype(np.int32), "bcolind": bcolind, "diagind": diagind,
            "vals": np.ascontiguousarray(vals), "slot": slot, "grid": grid, "npdim": npdim}


def rhs_vector(n):
    return np.sin(0.37 * np.arange(n, dtype=np.float64)) + 1.1


def poisson_counts(n, bs):
    """Sizes of the N^3 7-point pattern: nbrows, nnzb, nnzL(=nnzU), plist pairs."""
    nb = n ** 3
    nnzb = 7 * n ** 3 - 6 * n ** 2
    nnzl = 3 * n ** 3 - 3 * n ** 2
    return {"nbrows": nb, "nnzb": nnzb, "nnzl": nnzl, "pairs": nnzl}


def random_bsr(nbrows, bs, avg_offdiag=4, seed=12345, rowmajor=False, diag_weight=None):
    """Seeded unstructured test matrix: symmetric pattern, sorted columns, ragged rows (some rows
    have an empty lower or upper part), block-diagonally dominant values."""
    rng = np.random.default_rng(seed)
    nedges = max(1, nbrows * avg_offdiag // 2)
    a = rng.integers(0, nbrows, nedges)
    # mostly near-diagonal neighbours plus a few far ones (bandwidth like a renumbered mesh)
    span = np.where(rng.random(nedges) < 0.8, rng.integers(1, 12, nedges),
                    rng.integers(1, max(nbrows, 2), nedges))
    b = (a + span) % nbrows
    keep = a != b
    a, b = a[keep], b[keep]
    rows = np.concatenate([a, b, np.arange(nbrows)])
    cols = np.concatenate([b, a, np.arange(nbrows)])
    key = np.unique(rows.astype(np.int64) * nbrows + cols)
    br = (key // nbrows).astype(np.int64)
    bc = (key % nbrows).astype(np.int32)
    browptr = np.zeros(nbrows + 1, dtype=np.int64)
    np.add.at(browptr, br + 1, 1)
    browptr = np.cumsum(browptr)
    nnzb = key.size
    diagind = np.nonzero(br == bc)[0].astype(np.int32)
    deg = (browptr[1:] - browptr[:-1] - 1).astype(np.float64)
    vals = rng.uniform(-1.0, 1.0, (nnzb, bs, bs))
    scale_off = 0.6 / (np.maximum(deg, 1.0)[br] * bs)
    vals *= scale_off[:, None, None]
    w = (1.0 if diag_weight is None else diag_weight)
    dblk = rng.uniform(-0.1, 0.1, (nbrows, bs, bs)) + np.eye(bs)[None] * (w + 0.1 * np.arange(bs))[None, :, None]
    vals[diagind] = dblk
    return {"nbrows": nbrows, "nnzb": int(nnzb), "bs": bs, "rowmajor": bool(rowmajor),
            "browptr": browptr.astype(np.int32), "bcolind": bc, "diagind": diagind,
            "vals": np.ascontiguousarray(vals.reshape(-1))}


def permute_symmetric(m, rows):
    """P A P^T of a host matrix dict: new block-row k is old block-row rows[k]; block columns are
    renumbered with the inverse permutation and re-sorted inside each row (what a level / colouring
    reordering does before the reference's computeLevels is used, src/levelschedule.cpp:13-72)."""
    rows = np.asarray(rows, dtype=np.int64)
    nb, bs = int(m["nbrows"]), int(m["bs"])
    inv = np.empty(nb, dtype=np.int64)
    inv[rows] = np.arange(nb)
    rp = np.asarray(m["browptr"], dtype=np.int64)
    cnt = (rp[1:] - rp[:-1])[rows]
    nrp = np.zeros(nb + 1, dtype=np.int64)
    nrp[1:] = np.cumsum(cnt)
    src = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in rows]) if nb else np.zeros(0, dtype=np.int64)
    newrow = np.repeat(np.arange(nb), cnt)
    newcol = inv[np.asarray(m["bcolind"], dtype=np.int64)[src]]
    order = np.lexsort((newcol, newrow))
    src, newrow, newcol = src[order], newrow[order], newcol[order]
    vals = np.asarray(m["vals"]).reshape(-1, bs * bs)[src]
    diagind = np.nonzero(newrow == newcol)[0].astype(np.int32)
    return {"nbrows": nb, "nnzb": int(m["nnzb"]), "bs": bs, "rowmajor": bool(m.get("rowmajor", False)),
            "browptr": nrp.astype(np.int32), "bcolind": newcol.astype(np.int32), "diagind": diagind,
            "vals": np.ascontiguousarray(vals.reshape(-1))}


def dependency_levels(m):
    """Longest-path depth of every block-row in the dependency DAG of the (symmetrised) pattern:
    level(i) = 1 + max level(j) over j < i with A_ij or A_ji stored.  Host restatement (one in-order pass)
    of what blasted_hip_level_schedule computes on the device."""
    nb = int(m["nbrows"])
    rp, ci = np.asarray(m["browptr"]), np.asarray(m["bcolind"])
    level = np.zeros(nb, dtype=np.int32)
    for i in range(nb):
        cols = ci[rp[i]:rp[i + 1]]
        lo = cols[cols < i]
        if lo.size:
            level[i] = max(level[i], level[lo].max() + 1)
        hi = cols[cols > i]
        if hi.size:
            level[hi] = np.maximum(level[hi], level[i] + 1)
    return level


# ----------------------------------------------------------------------------- device generation

def poisson3d_device(n, bs, device, grid="uniform", chunk_rows=1 << 21):
    """The same matrix as poisson3d(n+2, bs, grid) built directly in HBM with torch (the 256^3 bs=4
    value array is 15 GB; it never exists on the host).  Returns torch tensors:
    browptr,bcolind,diagind (int32) and vals (float64, column-major blocks)."""
    import torch
    npdim = n + 2
    X = grid_coords(npdim, grid)
    lo, hi, dg = (torch.from_numpy(v).to(device) for v in _axis_coeffs(X))
    M = torch.from_numpy(slot_matrices(bs, False)).to(device)
    nb = n ** 3
    nnzb = 7 * n ** 3 - 6 * n ** 2
    browptr = torch.empty(nb + 1, dtype=torch.int32, device=device)
    diagind = torch.empty(nb, dtype=torch.int32, device=device)
    bcolind = torch.empty(nnzb, dtype=torch.int32, device=device)
    vals = torch.empty(nnzb * bs * bs, dtype=torch.float64, device=device)
    offs = torch.tensor([-n * n, -n, -1, 0, 1, n, n * n], dtype=torch.int64, device=device)
    slots = torch.arange(NSLOT, device=device)
    pos = 0
    browptr[0] = 0
    for r0 in range(0, nb, chunk_rows):
        r1 = min(nb, r0 + chunk_rows)
        idx = torch.arange(r0, r1, dtype=torch.int64, device=device)
        i = idx % n
        j = (idx // n) % n
        k = idx // (n * n)
        valid = torch.stack([k > 0, j > 0, i > 0, torch.ones_like(i, dtype=torch.bool), i < n - 1,
                             j < n - 1, k < n - 1], dim=1)
        cnt = valid.sum(dim=1)
        ends = torch.cumsum(cnt, 0) + pos
        browptr[r0 + 1:r1 + 1] = ends.to(torch.int32)
        diagind[r0:r1] = (ends - cnt + valid[:, :3].sum(dim=1)).to(torch.int32)
        cols = (idx[:, None] + offs[None, :])[valid]
        diag = dg[i] + dg[j]
        diag = diag + dg[k]
        coef = torch.stack([lo[k], lo[j], lo[i], diag, hi[i], hi[j], hi[k]], dim=1)[valid]
        slot = slots[None, :].expand(r1 - r0, NSLOT)[valid]
        m = cols.numel()
        bcolind[pos:pos + m] = cols.to(torch.int32)
        vals[pos * bs * bs:(pos + m) * bs * bs] = (coef[:, None] * M[slot]).reshape(-1)
        pos += m
        del idx, i, j, k, valid, cnt, ends, cols, diag, coef, slot
    assert pos == nnzb
    return {"nbrows": nb, "nnzb": nnzb, "bs": bs, "rowmajor": False, "browptr": browptr,
            "bcolind": bcolind, "diagind": diagind, "vals": vals, "grid": grid, "npdim": npdim}


def rhs_vector_device(n, device):
    import torch
    return torch.sin(0.37 * torch.arange(n, dtype=torch.float64, device=device)) + 1.1


# ----------------------------------------------------------------------------- config 4: unstructured

_MASK63 = (1 << 63) - 1


def _mix64(x):
    """Counter-based pseudo-random bits (splitmix64-style finaliser on torch int64, wrap-around
    arithmetic), identical on CPU and GPU: value = f(index), no generator state."""
    import torch
    x = (x ^ (x >> 30).bitwise_and(0x3FFFFFFFF)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    x = (x ^ (x >> 27).bitwise_and(0x1FFFFFFFFF)) * (-7723592293110705685)  # 0x94D049BB133111EB
    x = x ^ (x >> 31).bitwise_and(0x1FFFFFFFF)
    return x


def _uniform(idx, seed, lo, hi):
    """Deterministic U(lo,hi) of an int64 index tensor."""
    bits = _mix64(idx * 2 + seed * 7919 + 12345).bitwise_and((1 << 53) - 1)
    return lo + (hi - lo) * (bits.to(__import__("torch").float64) / float(1 << 53))


def unstructured_bsr(nside, bs=5, device="cpu", window=4096, seed=12345):
    """BASELINE config 4 stand-in (the reference ships no large unstructured matrix):
    a 15-point grid graph on nside^3 nodes (7-point star + the 8 body diagonals, <= 14 neighbours) whose
    node numbering is shuffled inside windows of `window` consecutive nodes (mesh-like locality, no band
    structure), symmetric pattern, sorted columns, counter-based pseudo-random block values:
    off-diagonal blocks U(-1,1) * 0.8/(5*deg), diagonal block (1 + 0.1 r) delta_rc + U(-0.02,0.02),
    so every scalar row is strictly diagonally dominant.  Returns torch tensors (column-major blocks)."""
    import torch
    dev = torch.device(device)
    n = nside
    nb = n ** 3
    idx = torch.arange(nb, dtype=torch.int64, device=dev)
    i, j, k = idx % n, (idx // n) % n, idx // (n * n)
    # node renumbering: stable sort by (window id, random key)
    key = (idx // window) * (1 << 40) + _mix64(idx + seed).bitwise_and((1 << 40) - 1)
    order = torch.argsort(key)           # order[new] = old
    perm = torch.empty_like(order)
    perm[order] = idx                    # perm[old] = new
    offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    offs += [(a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)]
    rows, cols = [], []
    for (a, b, c) in offs:
        ii, jj, kk = i + a, j + b, k + c
        ok = (ii >= 0) & (ii < n) & (jj >= 0) & (jj < n) & (kk >= 0) & (kk < n)
        old_col = (ii + n * (jj + n * kk))[ok]
        rows.append(perm[idx[ok]])
        cols.append(perm[old_col])
    rows = torch.cat(rows)
    cols = torch.cat(cols)
    keyrc = torch.sort(rows * nb + cols).values
    rows, cols = keyrc // nb, keyrc % nb
    nnzb = rows.numel()
    counts = torch.bincount(rows, minlength=nb)
    browptr = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    browptr[1:] = torch.cumsum(counts, 0)
    pos = torch.arange(nnzb, dtype=torch.int64, device=dev)
    diagind = pos[rows == cols]
    assert diagind.numel() == nb
    deg = (counts - 1).clamp(min=1).to(torch.float64)
    bs2 = bs * bs
    vals = torch.empty(nnzb * bs2, dtype=torch.float64, device=dev)
    chunk = 1 << 22
    e = torch.arange(bs2, dtype=torch.int64, device=dev)
    r_of_e, c_of_e = e % bs, e // bs     # column-major
    for p0 in range(0, nnzb, chunk):
        p1 = min(nnzb, p0 + chunk)
        blk = pos[p0:p1]
        u = _uniform(blk[:, None] * bs2 + e[None, :], seed, -1.0, 1.0)
        isd = (rows[p0:p1] == cols[p0:p1])[:, None]
        off = u * (0.8 / (5.0 * deg[rows[p0:p1]]))[:, None]
        dg = torch.where((r_of_e == c_of_e)[None, :], (1.0 + 0.1 * r_of_e.to(torch.float64))[None, :] + 0 * u,
                         0.02 * u)
        vals[p0 * bs2:p1 * bs2] = torch.where(isd, dg, off).reshape(-1)
    return {"nbrows": nb, "nnzb": int(nnzb), "bs": bs, "rowmajor": False, "browptr": browptr.to(torch.int32),
            "bcolind": cols.to(torch.int32), "diagind": diagind.to(torch.int32), "vals": vals}


def to_numpy(m):
    """Host copy of a matrix dict made of torch tensors."""
    out = dict(m)
    for k in ("browptr", "bcolind", "diagind", "vals"):
        out[k] = m[k].detach().cpu().numpy()
    return out
