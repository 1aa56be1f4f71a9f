"""The synthetic workload generators (fps_amd.problems): structure, determinism and conditioning of what
bench.py and the parity tests run on -- checked on the CPU at reduced sizes (the generators are size-independent recipes)."""
import numpy as np
import pytest

import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.distributed import halo_plan, row_partition


@pytest.mark.parametrize("gen", [problems.pde_control_like, problems.pde_control_hashed])
def test_pde_control_generators_follow_the_survey_recipe(gen):
    """Row i: `per_row` nonzeros at DISTINCT, sorted columns inside the window of width `window` centred at floor(i n / m)
    (clamped), values 2u - 1, the centre column carrying + 4; reproducible bit for bit; full row rank with
    singular values away from 0; and banded enough for the halo layout of the row-sharded path."""
    n, m, per, win = 24000, 2400, 24, 512
    qp = gen(n=n, m=m, per_row=per, window=win, seed=7)
    again = gen(n=n, m=m, per_row=per, window=win, seed=7)
    assert np.array_equal(qp.colind, again.colind) and np.array_equal(qp.vals, again.vals) and np.array_equal(qp.b, again.b)
    other = gen(n=n, m=m, per_row=per, window=win, seed=8)
    assert not np.array_equal(qp.colind, other.colind)
    assert qp.nnz == m * per and np.all(np.diff(qp.rowptr) == per)
    cols = qp.colind.reshape(m, per).astype(np.int64)
    assert np.all(np.diff(cols, axis=1) > 0)                       # distinct and sorted within a row
    center = (np.arange(m, dtype=np.int64) * n) // m
    start = np.clip(center - win // 2, 0, n - win)
    assert np.all(cols >= start[:, None]) and np.all(cols < (start + win)[:, None])
    vals = qp.vals.reshape(m, per)
    hit = cols == center[:, None]
    assert np.all(hit.sum(axis=1) == 1)                            # the centre column is an entry of every row ...
    assert np.all(vals[hit] > 3.0) and np.all(np.abs(vals[~hit]) <= 1.0)   # ... and the only one boosted by + 4
    A = qp.scipy_csr()
    assert np.allclose(A @ qp.xhat, qp.b)
    s = np.linalg.svd(A[:300].toarray(), compute_uv=False)        # (a block of rows: cheap, and rows couple only locally)
    assert s.min() > 1.0
    bounds = row_partition(qp.rowptr, 3)
    plan = halo_plan(qp.rowptr, qp.colind, n, bounds)
    assert plan is not None and plan.max_exchange_doubles() <= 2 * 2 * win


def test_hashed_offsets_are_not_stratified():
    """What distinguishes the two headline generators: the stratified one puts exactly one column in every slice of the
    window, the literal (hashed) one does not -- its gaps between neighbouring columns are far more irregular."""
    a = problems.pde_control_like(n=40000, m=2000, per_row=40, window=2048, seed=3)
    b = problems.pde_control_hashed(n=40000, m=2000, per_row=40, window=2048, seed=3)
    ga = np.diff(a.colind.reshape(2000, 40).astype(np.int64), axis=1)
    gb = np.diff(b.colind.reshape(2000, 40).astype(np.int64), axis=1)
    assert ga.max() <= 2 * (2048 // 40) + 2          # one column per stratum of ~51: a gap spans at most two strata
    assert gb.max() > 4 * (2048 // 40)               # hashed: some gaps are several strata wide ...
    assert (gb == 1).sum() > (ga == 1).sum()         # ... and some columns are immediate neighbours
