"""Row sharding of the constraint Jacobian across the GPUs of one node.

Rank r owns the constraints (rows of A) [bounds[r], bounds[r+1]): its handle is created with the GLOBAL n and the
LOCAL m, m-vectors (c, b, ys, q1, q2, w) are the rank's slices, n-vectors (x, g, gs, gx, p1, p2) are replicated.
Per Krylov iteration the ranks exchange ONE all-reduce of the partial A'u products (n x k doubles, k = 2 when the
two recurrences run fused) and ONE 4-double all-reduce of the m-vector norm partials; everything else is local.
"""
from __future__ import annotations

from dataclasses import replace

import numpy as np


def row_partition(rowptr: np.ndarray, nranks: int) -> np.ndarray:
    """Row boundaries (nranks + 1) balancing the NONZEROS per rank (the products are bandwidth-bound)."""
    m = rowptr.size - 1
    nnz = int(rowptr[-1])
    targets = (np.arange(1, nranks) * nnz) // nranks
    cuts = np.searchsorted(rowptr, targets, side="left")
    bounds = np.concatenate([[0], np.clip(cuts, 0, m), [m]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def shard_qp(qp, r0: int, r1: int):
    """The rows [r0, r1) of an EqQP as an EqQP with local m (global column indices, replicated q, d, x)."""
    rp = qp.rowptr.astype(np.int64)
    s, e = int(rp[r0]), int(rp[r1])
    return replace(qp, name=f"{qp.name}[rows {r0}:{r1}]", m=r1 - r0,
                   rowptr=(rp[r0:r1 + 1] - s).astype(np.int32), colind=qp.colind[s:e].copy(),
                   vals=qp.vals[s:e].copy(), b=qp.b[r0:r1].copy())
