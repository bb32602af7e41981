"""Deterministic synthetic inputs shared by tests, bench.py and the fixture
generator.  Everything is lower-triangle COORDINATE with 1-based indices, like GALAHAD's SMT_type."""
import numpy as np


def kat_indefinite():
    """5x5 indefinite known-answer system of src/sls/slst.f90:29-40 (x = 1..5)."""
    row = np.array([1, 2, 2, 3, 3, 4, 5], dtype=np.int32)
    col = np.array([1, 1, 5, 2, 3, 3, 5], dtype=np.int32)
    val = np.array([2.0, 3.0, 6.0, 4.0, 1.0, 5.0, 1.0])
    rhs = np.array([8.0, 45.0, 31.0, 15.0, 17.0])
    return 5, row, col, val, rhs, np.arange(1.0, 6.0)


def kat_definite():
    """5x5 positive-definite known-answer system of src/sls/slst.f90:41-51 (x = 1..5)."""
    row = np.array([1, 2, 3, 3, 4, 5, 5], dtype=np.int32)
    col = np.array([1, 2, 2, 3, 4, 1, 5], dtype=np.int32)
    val = np.array([6.0, 7.0, 2.0, 3.0, 4.0, 1.0, 5.0])
    rhs = np.array([11.0, 20.0, 13.0, 16.0, 26.0])
    return 5, row, col, val, rhs, np.arange(1.0, 6.0)


def banded_spd(n, semibw, seed=20240101):
    """cfg2: off-diagonals U(-0.5,0.5), diagonal 2*semibw+1 (strictly diagonally dominant => SPD),
    row-major within the band; x*_i = 1 + (i mod 7)."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    i = np.arange(n, dtype=np.int64)
    for d in range(semibw, 0, -1):
        r = i[d:]
        rows.append(r)
        cols.append(r - d)
        vals.append(rng.uniform(-0.5, 0.5, len(r)))
    rows.append(i)
    cols.append(i)
    vals.append(np.full(n, 2.0 * semibw + 1.0))
    row = np.concatenate(rows)
    col = np.concatenate(cols)
    val = np.concatenate(vals)
    o = np.lexsort((col, row))
    row, col, val = row[o], col[o], val[o]
    xstar = 1.0 + (np.arange(n) % 7)
    rhs = sym_matvec(n, row, col, val, xstar)
    return n, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def grid2d(nx, ny, shift=0.0):
    """5-point Laplacian on an nx x ny grid minus shift*I (cfg4: shift=1 makes it indefinite)."""
    idx = np.arange(nx * ny).reshape(ny, nx)
    row = [idx.ravel(), idx[:, 1:].ravel(), idx[1:, :].ravel()]
    col = [idx.ravel(), idx[:, :-1].ravel(), idx[:-1, :].ravel()]
    val = [np.full(nx * ny, 4.0 - shift), np.full(idx[:, 1:].size, -1.0), np.full(idx[1:, :].size, -1.0)]
    row, col, val = np.concatenate(row), np.concatenate(col), np.concatenate(val)
    n = nx * ny
    xstar = np.ones(n)
    rhs = sym_matvec(n, row, col, val, xstar)
    return n, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def grid3d(nx, ny, nz):
    """7-point Laplacian + I on an nx x ny x nz grid (SPD)."""
    idx = np.arange(nx * ny * nz).reshape(nz, ny, nx)
    row = [idx.ravel(), idx[:, :, 1:].ravel(), idx[:, 1:, :].ravel(), idx[1:, :, :].ravel()]
    col = [idx.ravel(), idx[:, :, :-1].ravel(), idx[:, :-1, :].ravel(), idx[:-1, :, :].ravel()]
    val = [np.full(idx.size, 7.0)] + [np.full(r.size, -1.0) for r in row[1:]]
    row, col, val = np.concatenate(row), np.concatenate(col), np.concatenate(val)
    n = idx.size
    xstar = np.ones(n)
    rhs = sym_matvec(n, row, col, val, xstar)
    return n, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def kkt_qpband(n, m, seed=20240102):
    """cfg3: K = [H A^T; A 0], H = tridiag(2,-1) + diag(sigma), sigma ~ logU(1e-4,1e4);
    A row i = e_i + e_{m+i} (QPBAND pattern, examples/QPBAND.SIF:14-60); expected inertia (n, m, 0)."""
    rng = np.random.default_rng(seed)
    sigma = 10.0 ** rng.uniform(-4, 4, n)
    i = np.arange(n, dtype=np.int64)
    row = [i, i[1:], n + np.arange(m), n + np.arange(m)]
    col = [i, i[:-1], np.arange(m), m + np.arange(m)]
    val = [2.0 + sigma, np.full(n - 1, -1.0), np.ones(m), np.ones(m)]
    row, col, val = np.concatenate(row), np.concatenate(col), np.concatenate(val)
    N = n + m
    xstar = np.ones(N)
    rhs = sym_matvec(N, row, col, val, xstar)
    return N, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def random_sparse(n, avg_deg, seed, spd=True):
    rng = np.random.default_rng(seed)
    nz = int(n * avg_deg / 2)
    r = rng.integers(0, n, nz)
    c = rng.integers(0, n, nz)
    keep = r != c
    r, c = np.maximum(r[keep], c[keep]), np.minimum(r[keep], c[keep])
    v = rng.uniform(-1, 1, len(r))
    absrow = np.zeros(n)
    np.add.at(absrow, r, np.abs(v))
    np.add.at(absrow, c, np.abs(v))
    diag = absrow + 1.0 if spd else rng.uniform(-1, 1, n) * (absrow + 1.0)
    row = np.concatenate([r, np.arange(n)])
    col = np.concatenate([c, np.arange(n)])
    val = np.concatenate([v, diag])
    xstar = rng.uniform(-1, 1, n)
    rhs = sym_matvec(n, row, col, val, xstar)
    return n, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def sym_matvec(n, row0, col0, val, x):
    """y = A x for lower-triangle COO with 0-based indices (duplicates summed)."""
    y = np.zeros(n)
    np.add.at(y, row0, val * x[col0])
    off = row0 != col0
    np.add.at(y, col0[off], val[off] * x[row0[off]])
    return y


def scaled_residual(n, row, col, val, x, b):
    """||b - A x||_inf / (||A||_inf ||x||_inf + ||b||_inf), 1-based inputs."""
    r0, c0 = np.asarray(row) - 1, np.asarray(col) - 1
    res = b - sym_matvec(n, r0, c0, val, x)
    absrow = np.zeros(n)
    np.add.at(absrow, r0, np.abs(val))
    off = r0 != c0
    np.add.at(absrow, c0[off], np.abs(val[off]))
    return np.abs(res).max() / (absrow.max() * np.abs(x).max() + np.abs(b).max())


def grid3d_27pt_perturbed(nx, ny, nz, seed=20240105, frac=0.10):
    """cfg5: 27-point-style stencil on an nx x ny x nz grid with a fraction `frac` of its
    off-diagonal couplings removed and as many longer-range ones added (offsets from the distance-2 shell of the grid:
    a uniformly random pattern of this size has no usable ordering -- the survey says so -- so "long range" means
    beyond the stencil, not across the domain), made SPD by diagonal dominance; lower triangle, 1-based."""
    rng = np.random.default_rng(seed)
    idx = np.arange(nx * ny * nz, dtype=np.int64).reshape(nz, ny, nx)

    def pairs(dz, dy, dx):
        a = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)]
        b = idx[max(0, dz):nz - max(0, -dz), max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)]
        return a.ravel(), b.ravel()          # b = a shifted by the offset

    rows, cols, vals = [], [], []
    near = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) > (0, 0, 0)]
    removed = 0
    for off in near:
        a, b = pairs(*off)
        keep = rng.random(len(a)) >= frac
        removed += int((~keep).sum())
        rows.append(np.maximum(a, b)[keep])
        cols.append(np.minimum(a, b)[keep])
        vals.append(np.full(int(keep.sum()), -1.0))
    far = [(dz, dy, dx) for dz in range(-2, 3) for dy in range(-2, 3) for dx in range(-2, 3)
           if max(abs(dz), abs(dy), abs(dx)) == 2 and (dz, dy, dx) > (0, 0, 0)]
    per = removed // len(far) + 1
    for off in far:
        a, b = pairs(*off)
        pick = rng.choice(len(a), size=min(per, len(a)), replace=False)
        rows.append(np.maximum(a, b)[pick])
        cols.append(np.minimum(a, b)[pick])
        vals.append(np.full(len(pick), -0.5))
    row, col, val = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    n = idx.size
    absrow = np.zeros(n)
    np.add.at(absrow, row, np.abs(val))
    np.add.at(absrow, col, np.abs(val))
    row = np.concatenate([row, np.arange(n)])
    col = np.concatenate([col, np.arange(n)])
    val = np.concatenate([val, absrow + 1.0])
    xstar = np.ones(n)
    rhs = sym_matvec(n, row, col, val, xstar)
    return n, (row + 1).astype(np.int32), (col + 1).astype(np.int32), val, rhs, xstar


def qpband(N):
    """examples/QPBAND.SIF / QPBAND.qplib (BASELINE.json configs[0]): min 1/2 x'Hx + g'x, H = tridiag(2, -1),
    g_i = -i/N, x_i + x_{M+i} >= 1 for i = 1..M = N/2, 0 <= x <= 2.  Returns (n, m, H, A, g, c_l, c_u, x_l, x_u) with
    1-based COO triples, H lower triangle."""
    n, m = N, N // 2
    i = np.arange(1, n + 1, dtype=np.int32)
    hr = np.r_[i, i[1:]]
    hc = np.r_[i, i[:-1]]
    hv = np.r_[np.full(n, 2.0), np.full(n - 1, -1.0)]
    k = np.arange(1, m + 1, dtype=np.int32)
    ar = np.r_[k, k]
    ac = np.r_[k, k + m]
    av = np.ones(2 * m)
    g = -np.arange(1, n + 1) / float(N)
    return (n, m, (hr, hc, hv), (ar, ac, av), g, np.ones(m), np.full(m, 1e20), np.zeros(n), np.full(n, 2.0))
