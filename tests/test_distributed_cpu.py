"""CPU, world_size = 2, gloo: the row-sharding scheme of fps_amd.distributed (what bench.py --gpus N runs over RCCL).

Each rank holds a row block of A.  Test-side numpy Golub-Kahan/LSQR with exactly the communication pattern of the
HIP path -- ONE vector all-reduce of the partial A'v products and ONE scalar all-reduce of the sharded-vector norms
per iteration -- must reproduce the unsharded C oracle's iterates."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sharded_lsqr(local, n, b, lam, iters):
    """LSQR on B = A' (n x m): u in R^n replicated, v, w, x in R^m sharded by rows of A."""
    import scipy.sparse as sp

    A = sp.csr_matrix((local.vals, local.colind, local.rowptr), shape=(local.m, n))

    def allsum(t):
        t = torch.from_numpy(np.atleast_1d(np.asarray(t, dtype=np.float64)).copy())
        dist.all_reduce(t)
        return t.numpy()

    x = np.zeros(local.m)
    beta = np.linalg.norm(b)
    u = b / beta
    v = A @ u  # B'u, local rows
    alpha = np.sqrt(allsum(v @ v)[0])
    v /= alpha
    w = v.copy()
    phibar, rhobar = beta, alpha
    for _ in range(iters):
        # beta u = B v - alpha u :   A'v = sum_r A_r' v_r  -> vector all-reduce
        u = allsum(A.T @ v) - alpha * u
        beta = np.linalg.norm(u)  # replicated vector: no communication
        u /= beta
        # alpha v = B'u - beta v : local rows, the norm is a scalar all-reduce
        v = A @ u - beta * v
        alpha = np.sqrt(allsum(v @ v)[0])
        v /= alpha
        rhobar1 = np.hypot(rhobar, lam)
        c1 = rhobar / rhobar1
        phibar = c1 * phibar
        rho = np.hypot(rhobar1, beta)
        c, s = rhobar1 / rho, beta / rho
        theta, rhobar = s * alpha, -c * alpha
        phi, phibar = c * phibar, s * phibar
        x += phi / rho * w
        w = v - theta / rho * w
    return x


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems
    from fps_amd.distributed import row_partition, shard_qp

    qp = problems.pde_control_like(n=3000, m=300, per_row=16, window=256, seed=11)
    bounds = row_partition(qp.rowptr, world)
    local = shard_qp(qp, int(bounds[rank]), int(bounds[rank + 1]))
    g = qp.qdiag * qp.x + qp.d
    xs = {k: _sharded_lsqr(local, qp.n, g, 0.1, k) for k in (1, 4, 9)}
    # replicated quantities agree bitwise across ranks; gather the shards of x on rank 0
    for k, xl in xs.items():
        parts = [None] * world
        dist.all_gather_object(parts, xl)
        if rank == 0:
            out[k] = np.concatenate(parts)
    dist.destroy_process_group()


def test_row_partition_balances_nonzeros():
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems
    from fps_amd.distributed import row_partition, shard_qp

    qp = problems.aug2dc_like(N=12)
    for P in (1, 2, 3, 8):
        b = row_partition(qp.rowptr, P)
        assert b[0] == 0 and b[-1] == qp.m and len(b) == P + 1 and np.all(np.diff(b) >= 0)
        nz = np.diff(qp.rowptr.astype(np.int64)[b])
        assert nz.sum() == qp.nnz and nz.max() - nz.min() <= 2 * np.diff(qp.rowptr).max()
        blocks = [shard_qp(qp, b[r], b[r + 1]) for r in range(P)]
        assert sum(s.m for s in blocks) == qp.m and sum(s.nnz for s in blocks) == qp.nnz
        import scipy.sparse as sp

        A = sp.vstack([sp.csr_matrix((s.vals, s.colind, s.rowptr), shape=(s.m, qp.n)) for s in blocks])
        assert abs(A - qp.scipy_csr()).max() == 0


@pytest.mark.timeout(300)
def test_sharded_lsqr_world2_gloo_matches_oracle(oracle):
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems

    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    qp = problems.pde_control_like(n=3000, m=300, per_row=16, window=256, seed=11)
    g = qp.qdiag * qp.x + qp.d
    for k in (1, 4, 9):
        ref, st = oracle.lsqr(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, g, lam=0.1, itmax=k, transposed=True,
                              axtol=0, btol=0, etol=0, conlim=0)
        assert st.niter == k
        np.testing.assert_allclose(out[k], ref, rtol=0, atol=1e-12 * np.linalg.norm(ref))
