"""Synthetic equality-constrained QP workloads (BASELINE.json `configs`).

The reference publishes no data for its hot path, only sizes; these generators are this build's
specification of them.  Every random number comes from a counter-based generator
``u(seed, i, k) = splitmix64(seed ^ i*GOLDEN ^ k*C2) / 2**64`` so any language can reproduce the same bits.

User model (what FletcherPenaltyNLP wraps, model-Fletcherpenaltynlp.jl:105-188):
    f(x) = 1/2 x' diag(q) x + d' x,     c(x) = A x - b = 0
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_C2 = np.uint64(0xD1B54A32D192ED03)


def splitmix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, i, k) -> np.ndarray:
    """u(seed, i, k) in [0, 1); i and k broadcast."""
    i = np.asarray(i, dtype=np.uint64)
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) ^ (i * _GOLDEN) ^ (k * _C2)
    return (splitmix64(z) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class EqQP:
    """Equality-constrained QP with a CSR Jacobian (0-based int32 indices, fp64 values)."""

    name: str
    n: int
    m: int
    rowptr: np.ndarray
    colind: np.ndarray
    vals: np.ndarray
    qdiag: np.ndarray
    d: np.ndarray
    b: np.ndarray
    x: np.ndarray  # evaluation point (infeasible)
    xhat: np.ndarray  # a feasible point, b = A xhat

    @property
    def nnz(self) -> int:
        return int(self.vals.size)

    def scipy_csr(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.vals, self.colind, self.rowptr), shape=(self.m, self.n))

    def point(self, t: int) -> np.ndarray:
        """t-th distinct evaluation point (a line-search never evaluates the same x twice,
        which is what defeats the hash(x) memo of model-Fletcherpenaltynlp.jl:235-237)."""
        idx = np.arange(self.n)
        return self.xhat + 0.1 * (2.0 * uniform01(977 + t, idx, 5) - 1.0)


def _finish(name, n, m, rowptr, colind, vals, seed) -> EqQP:
    import scipy.sparse as sp

    idx = np.arange(n)
    qdiag = 1.0 + 9.0 * uniform01(seed, idx, 1)
    d = 2.0 * uniform01(seed, idx, 2) - 1.0
    xhat = 2.0 * uniform01(seed, idx, 3) - 1.0
    A = sp.csr_matrix((vals, colind, rowptr), shape=(m, n))
    b = A @ xhat
    x = xhat + 0.1 * (2.0 * uniform01(seed, idx, 4) - 1.0)
    return EqQP(name, n, m, rowptr.astype(np.int32), colind.astype(np.int32), vals, qdiag, d, b, x, xhat)


def _stratified_rows(m, n, per_row, start, width, seed, diag_col=None, diag_boost=0.0):
    """Each of the m rows gets `per_row` sorted, distinct columns: one per stratum of
    [start[i], start[i] + width).  If diag_col is given, the stratum containing diag_col[i] is
    placed exactly there and its value gets +diag_boost."""
    rows = np.repeat(np.arange(m, dtype=np.int64), per_row)
    ks = np.tile(np.arange(per_row, dtype=np.int64), m)
    lo = (ks * width) // per_row
    hi = ((ks + 1) * width) // per_row
    eid = rows * per_row + ks
    off = lo + np.floor(uniform01(seed, eid, 11) * (hi - lo)).astype(np.int64)
    vals = 2.0 * uniform01(seed, eid, 12) - 1.0
    cols = np.repeat(start, per_row) + off
    if diag_col is not None:
        rel = np.repeat(diag_col - start, per_row)
        hit = (rel >= lo) & (rel < hi)
        cols = np.where(hit, np.repeat(diag_col, per_row), cols)
        vals = np.where(hit, vals + diag_boost, vals)
    assert cols.min() >= 0 and cols.max() < n
    rowptr = np.arange(m + 1, dtype=np.int64) * per_row
    return rowptr, cols, vals


def random_eqqp(n=100_000, m=10_000, per_row=100, seed=1234) -> EqQP:
    """configs[1]: random sparse eq-QP; every row has `per_row` nonzeros spread over all n columns."""
    start = np.zeros(m, dtype=np.int64)
    rowptr, cols, vals = _stratified_rows(m, n, per_row, start, n, seed)
    return _finish(f"random-eqqp-n{n}-m{m}", n, m, rowptr, cols, vals, seed)


def pde_control_like(n=1_000_000, m=100_000, per_row=100, window=8192, seed=1234) -> EqQP:
    """configs[4] / the headline: row i has `per_row` nonzeros in a column window of width `window`
    centred at floor(i n / m) (clamped), and A[i, floor(i n / m)] carries an extra +4."""
    window = min(window, n)
    center = (np.arange(m, dtype=np.int64) * n) // m
    start = np.clip(center - window // 2, 0, n - window)
    rowptr, cols, vals = _stratified_rows(m, n, per_row, start, window, seed, diag_col=center, diag_boost=4.0)
    return _finish(f"pde-control-like-n{n}-m{m}", n, m, rowptr, cols, vals, seed)


def _hashed_rows(m, n, per_row, start, width, seed, diag_col, diag_boost):
    """SURVEY.md 8(d) as written: row i has `per_row` DISTINCT columns start[i] + off, the offsets hashed over the whole
    window, off = floor(u(seed, entry, 11 + 16 a) * width) with a = 0 and a re-draw (a + 1, a + 2, ...) of every entry that
    collides with an EARLIER entry of its row; entry 0 of a row is its centre column diag_col[i] (value + diag_boost), so a row
    has exactly per_row entries.  Columns sorted within a row; values 2u - 1 by entry."""
    per = per_row
    eid = np.arange(m, dtype=np.int64)[:, None] * per + np.arange(per, dtype=np.int64)[None, :]
    off = np.floor(uniform01(seed, eid, 11) * width).astype(np.int64)
    off[:, 0] = diag_col - start
    attempt = np.zeros((m, per), dtype=np.int64)
    for _ in range(64):
        order = np.argsort(off, axis=1, kind="stable")          # equal offsets keep their entry order
        so = np.take_along_axis(off, order, axis=1)
        dup_sorted = np.zeros((m, per), dtype=bool)
        dup_sorted[:, 1:] = so[:, 1:] == so[:, :-1]              # every entry but the earliest of a run of equal offsets
        if not dup_sorted.any():
            break
        dup = np.zeros((m, per), dtype=bool)
        np.put_along_axis(dup, order, dup_sorted, axis=1)
        attempt[dup] += 1
        off[dup] = np.floor(uniform01(seed, eid[dup], 11 + 16 * attempt[dup]) * width).astype(np.int64)
    else:
        raise RuntimeError("hashed offsets: collisions left after 64 re-draws")
    vals = 2.0 * uniform01(seed, eid, 12) - 1.0
    vals[:, 0] += diag_boost
    cols = start[:, None] + off
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    assert cols.min() >= 0 and cols.max() < n and np.all(cols[:, 1:] > cols[:, :-1])
    return np.arange(m + 1, dtype=np.int64) * per, cols.ravel(), vals.ravel()


def pde_control_hashed(n=1_000_000, m=100_000, per_row=100, window=8192, seed=1234) -> EqQP:
    """The headline shape with SURVEY.md 8(d)'s LITERAL column rule: row i has `per_row` nonzeros at distinct HASHED offsets
    of a column window of width `window` centred at floor(i n / m) (clamped), re-drawn on collision, values 2u - 1, and
    A[i, floor(i n / m)] carries an extra +4.  `pde_control_like` (the bench headline since round 1) draws one column per
    equal slice of the window instead -- a more regular gather pattern; this variant shows what the layouts do without that
    regularity (bench.py --workload "pde-control-hashed ...", profiles/r04_configs.md)."""
    window = min(window, n)
    center = (np.arange(m, dtype=np.int64) * n) // m
    start = np.clip(center - window // 2, 0, n - window)
    rowptr, cols, vals = _hashed_rows(m, n, per_row, start, window, seed, center, 4.0)
    return _finish(f"pde-control-hashed-n{n}-m{m}", n, m, rowptr, cols, vals, seed)


def aug2dc_like(N=100, seed=1234) -> EqQP:
    """configs[3] stand-in ("AUG2DC-like", restated from the published description; CUTEst/SIF is not
    available offline, so this is NOT SIF-verified): variables = edges of an N x N grid graph with a
    boundary ring (n = 2N(N+1)), constraints = nodes (m = N^2), A = signed node-edge incidence."""
    m = N * N
    node = lambda i, j: i * N + j
    rows, cols, vals = [], [], []
    e = 0
    # horizontal edges: (i, j-1) -> (i, j) for j = 0..N  (j = 0 and j = N touch one node only)
    for i in range(N):
        for j in range(N + 1):
            if j > 0:
                rows.append(node(i, j - 1)); cols.append(e); vals.append(-1.0)
            if j < N:
                rows.append(node(i, j)); cols.append(e); vals.append(1.0)
            e += 1
    for j in range(N):
        for i in range(N + 1):
            if i > 0:
                rows.append(node(i - 1, j)); cols.append(e); vals.append(-1.0)
            if i < N:
                rows.append(node(i, j)); cols.append(e); vals.append(1.0)
            e += 1
    n = e
    import scipy.sparse as sp

    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    return _finish(f"aug2dc-like-N{N}", n, m, A.indptr.astype(np.int64), A.indices.astype(np.int64),
                   A.data.astype(np.float64), seed)


def dense_block(n=4096, m=2048, seed=1234) -> EqQP:
    """configs[2]: dense Jacobian A[i, j] = (2u - 1)/sqrt(n), stored as CSR with every entry present."""
    eid = np.arange(m * n, dtype=np.int64)
    vals = (2.0 * uniform01(seed, eid, 12) - 1.0) / np.sqrt(n)
    rowptr = np.arange(m + 1, dtype=np.int64) * n
    cols = np.tile(np.arange(n, dtype=np.int64), m)
    return _finish(f"dense-block-n{n}-m{m}", n, m, rowptr, cols, vals, seed)
