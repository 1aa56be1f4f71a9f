"""Row sharding of the constraint Jacobian across the GPUs of one node.

Rank r owns the constraints (rows of A) [bounds[r], bounds[r+1]); m-vectors (c, b, ys, q1, q2, w) are the rank's
slices.  Two layouts of the n-vectors (x, g, gs, gx, p1, p2):

* replicated (`shard_qp`): the handle is created with the GLOBAL n; per Krylov iteration the ranks all-reduce the
  partial A'u products (n x k doubles, k = 2 when the two recurrences run fused) and one 4-double scalar payload.
* HALO mode (`halo_plan` + `shard_qp_halo`; banded Jacobians such as the headline generator's 8192-column band): the
  rows of rank r only touch a column window [w_lo(r), w_hi(r)); windows tile [0, n) and overlap between NEIGHBOURS
  only.  Every rank holds just its window of each n-vector; per iteration it exchanges the partial products of its two
  overlap regions with its neighbours (<= 2 x 8192 x k doubles instead of n x k) and all-reduces 4 doubles per
  reduction.  On an overlap both ranks compute a + b (resp. b + a) and so hold bitwise identical values; sums over
  n-vectors run over the OWNED prefix [w_lo(r), w_lo(r+1)) of each window.

Everything here is host-side planning / slicing / re-assembly; the per-iteration exchanges happen inside libfpsq on the
solver's stream (RCCL send/recv + all-reduce, csrc/fpsq.hip `comm_reduce_long`).
"""
from __future__ import annotations

from dataclasses import dataclass, replace

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


@dataclass
class HaloPlan:
    """Column windows of a row partition.  w_lo / w_hi: (P,) global window bounds; windows tile [0, n), overlap only
    between neighbours; rank r OWNS [w_lo[r], w_lo[r+1]) (the last rank to n)."""

    n: int
    bounds: np.ndarray
    w_lo: np.ndarray
    w_hi: np.ndarray

    @property
    def nranks(self) -> int:
        return int(self.w_lo.size)

    def overlaps(self, r: int):
        """(overlap_left, overlap_right) of rank r: what fpsq_comm_set_halo takes."""
        left = int(self.w_hi[r - 1] - self.w_lo[r]) if r > 0 else 0
        right = int(self.w_hi[r] - self.w_lo[r + 1]) if r + 1 < self.nranks else 0
        return left, right

    def window(self, r: int) -> slice:
        return slice(int(self.w_lo[r]), int(self.w_hi[r]))

    def owned(self, r: int) -> slice:
        """Global indices rank r owns (a prefix of its window)."""
        hi = int(self.w_lo[r + 1]) if r + 1 < self.nranks else self.n
        return slice(int(self.w_lo[r]), hi)

    def owned_local(self, r: int) -> slice:
        o = self.owned(r)
        return slice(0, o.stop - o.start)

    def max_exchange_doubles(self, nrhs: int = 2) -> int:
        """Largest per-iteration neighbour message (doubles) of any rank, both directions summed."""
        return max(sum(self.overlaps(r)) for r in range(self.nranks)) * nrhs

    def assemble(self, windows) -> np.ndarray:
        """Global n-vector from the ranks' window vectors (each entry from its owner)."""
        out = np.empty(self.n)
        for r, w in enumerate(windows):
            out[self.owned(r)] = np.asarray(w)[self.owned_local(r)]
        return out


def halo_plan(rowptr, colind, n: int, bounds) -> HaloPlan | None:
    """Column windows for the row partition `bounds`, or None when the halo layout does not apply (a rank's rows reach
    beyond its neighbours' windows, i.e. the Jacobian is not banded at this partition): callers then fall back to
    replicated n-vectors + all-reduce."""
    bounds = np.asarray(bounds, dtype=np.int64)
    P = bounds.size - 1
    rp = np.asarray(rowptr, dtype=np.int64)
    lo, hi = np.empty(P, dtype=np.int64), np.empty(P, dtype=np.int64)
    for r in range(P):
        s, e = int(rp[bounds[r]]), int(rp[bounds[r + 1]])
        if e == s:
            return None  # an empty rank: nothing to anchor its window to
        cols = colind[s:e]
        lo[r], hi[r] = int(cols.min()), int(cols.max()) + 1
    if np.any(np.diff(lo) < 0) or np.any(np.diff(hi) < 0):
        return None
    # Window boundaries on multiples of 8 columns (round 5): a rank's overlap regions and its owned part then start and end on
    # 128-byte lines of its interleaved [n][2] fp64 vectors -- what lets the exchange and the finish of the overlap rows ride
    # inside the one-launch iteration (csrc fuse_halo_wg: every line of the long pair has ONE owner).  A window only grows.
    lo = (lo // 8) * 8
    hi = np.minimum(((hi + 7) // 8) * 8, n)
    lo[0], hi[-1] = 0, n  # the windows must tile [0, n): columns no row touches still carry x, g, gx entries
    for r in range(P - 1):
        if lo[r + 1] > hi[r]:
            hi[r] = lo[r + 1]  # close gaps
    for r in range(P - 2):
        if lo[r + 2] < hi[r]:
            return None  # three windows share columns: neighbour-only exchange impossible
    for r in range(P):  # an overlap must not swallow a whole window (owned prefix non-empty, regions disjoint)
        left = hi[r - 1] - lo[r] if r > 0 else 0
        right = hi[r] - lo[r + 1] if r + 1 < P else 0
        if left + right > hi[r] - lo[r] or (r + 1 < P and lo[r + 1] <= lo[r]):
            return None
    return HaloPlan(int(n), bounds, lo, hi)


def shard_qp_halo(qp, plan: HaloPlan, r: int):
    """Rank r's block of an EqQP in halo layout: rows [bounds[r], bounds[r+1]), column indices relative to its window,
    q, d, x, xhat sliced to the window (n = window length)."""
    r0, r1 = int(plan.bounds[r]), int(plan.bounds[r + 1])
    loc = shard_qp(qp, r0, r1)
    w = plan.window(r)
    return replace(loc, name=f"{qp.name}[rows {r0}:{r1}, cols {w.start}:{w.stop}]", n=w.stop - w.start,
                   colind=(loc.colind.astype(np.int64) - w.start).astype(np.int32), qdiag=qp.qdiag[w].copy(),
                   d=qp.d[w].copy(), x=qp.x[w].copy(), xhat=qp.xhat[w].copy())


# ---- host-side collectives over torch.distributed (gloo on CPU, nccl = RCCL on the GPUs): re-assembly of results and
# ---- the reference implementation of the halo protocol that the tests drive

def halo_exchange_add(vec: np.ndarray, plan: HaloPlan, rank: int, group=None) -> np.ndarray:
    """The halo step of the sharded A' product on HOST arrays: `vec` (window x k) holds this rank's partial products;
    returns vec with the neighbours' partials added on the two overlap regions -- the exchange libfpsq performs on the
    device with ncclSend/ncclRecv (csrc/fpsq.hip RcclComm::halo_exchange + k_halo_add)."""
    import torch
    import torch.distributed as dist

    left, right = plan.overlaps(rank)
    v = np.array(vec, dtype=np.float64, copy=True)
    ops, bufs = [], {}
    if left:
        send = torch.from_numpy(np.ascontiguousarray(v[:left]))
        bufs["l"] = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, rank - 1, group), dist.P2POp(dist.irecv, bufs["l"], rank - 1, group)]
    if right:
        send = torch.from_numpy(np.ascontiguousarray(v[v.shape[0] - right:]))
        bufs["r"] = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, rank + 1, group), dist.P2POp(dist.irecv, bufs["r"], rank + 1, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if left:
        v[:left] += bufs["l"].numpy()
    if right:
        v[v.shape[0] - right:] += bufs["r"].numpy()
    return v


def allreduce_sum(values, group=None) -> np.ndarray:
    """Scalar payload all-reduce (the <= 4 doubles per reduction of the Krylov loop)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.atleast_1d(np.asarray(values, dtype=np.float64)).copy())
    dist.all_reduce(t, group=group)
    return t.numpy()


def gather_global(window_vec: np.ndarray, plan: HaloPlan, rank: int, group=None) -> np.ndarray:
    """Global n-vector on every rank from the ranks' windows (each entry taken from its owner)."""
    import torch.distributed as dist

    parts = [None] * plan.nranks
    dist.all_gather_object(parts, np.asarray(window_vec)[plan.owned_local(rank)], group=group)
    return np.concatenate(parts)
