"""CPU, world_size = 2 and 3, gloo: the row-sharding schemes of fps_amd.distributed (what bench.py --gpus N runs over RCCL).

Each rank holds a row block of A.  A numpy Golub-Kahan/LSQR with exactly the communication pattern of the HIP path must
reproduce the unsharded C oracle's iterates, in both layouts:
  replicated -- ONE vector all-reduce of the partial A'v products and ONE scalar all-reduce per iteration;
  halo       -- the product's own planning and collectives (fps_amd.distributed: halo_plan, shard_qp_halo,
                halo_exchange_add over torch.distributed P2P, allreduce_sum, gather_global): neighbour exchange of the
                overlap regions of the column windows + scalar all-reduces, sums over owned prefixes only."""
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


def _halo_lsqr(local, plan, rank, b_win, lam, iters):
    """LSQR on B = A' with n-vectors as COLUMN WINDOWS: the device loop of csrc/fpsq.hip in halo mode, on the host, with
    the collectives of fps_amd.distributed."""
    import scipy.sparse as sp
    from fps_amd.distributed import allreduce_sum, halo_exchange_add

    nw = local.n
    A = sp.csr_matrix((local.vals, local.colind, local.rowptr), shape=(local.m, nw))
    own = plan.owned_local(rank)

    def nrm_long(u):  # sum over the owned prefix, then the scalar all-reduce
        return np.sqrt(allreduce_sum(u[own] @ u[own])[0])

    x = np.zeros(local.m)
    beta = nrm_long(b_win)
    u = b_win / beta
    v = A @ u
    alpha = np.sqrt(allreduce_sum(v @ v)[0])
    v /= alpha
    w = v.copy()
    phibar, rhobar = beta, alpha
    for _ in range(iters):
        u = halo_exchange_add((A.T @ v)[:, None], plan, rank)[:, 0] - alpha * u
        beta = nrm_long(u)
        u /= beta
        v = A @ u - beta * v
        alpha = np.sqrt(allreduce_sum(v @ v)[0])
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
    return x, u


def _halo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems
    from fps_amd.distributed import gather_global, halo_plan, row_partition, shard_qp_halo

    qp = problems.pde_control_like(n=3000, m=300, per_row=16, window=256, seed=11)
    bounds = row_partition(qp.rowptr, world)
    plan = halo_plan(qp.rowptr, qp.colind, qp.n, bounds)
    assert plan is not None
    local = shard_qp_halo(qp, plan, rank)
    g = (qp.qdiag * qp.x + qp.d)[plan.window(rank)]
    for k in (1, 4, 9):
        xl, ul = _halo_lsqr(local, plan, rank, g, 0.1, k)
        parts = [None] * world
        dist.all_gather_object(parts, xl)
        ug = gather_global(ul, plan, rank)  # the long Golub-Kahan vector, re-assembled from the owned prefixes
        # overlaps are bitwise identical on the two ranks that share them
        left, right = plan.overlaps(rank)
        edges = [None] * world
        dist.all_gather_object(edges, (ul[:left].copy(), ul[ul.size - right:].copy()))
        if rank > 0:
            assert np.array_equal(edges[rank - 1][1], ul[:left])
        if rank == 0:
            out[k] = (np.concatenate(parts), ug)
    dist.destroy_process_group()


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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_halo_sharded_lsqr_gloo_matches_oracle(oracle, world):
    """The halo layout through the product's planning and collectives (fps_amd.distributed) on `world` CPU ranks."""
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems

    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_halo_worker, args=(world, port, out), nprocs=world, join=True)
    qp = problems.pde_control_like(n=3000, m=300, per_row=16, window=256, seed=11)
    g = qp.qdiag * qp.x + qp.d
    for k in (1, 4, 9):
        ref, st = oracle.lsqr(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, g, lam=0.1, itmax=k, transposed=True,
                              axtol=0, btol=0, etol=0, conlim=0)
        assert st.niter == k
        x, u = out[k]
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-12 * np.linalg.norm(ref))
        assert u.shape == (qp.n,) and abs(np.linalg.norm(u) - 1.0) < 1e-12  # unit Golub-Kahan vector, assembled


def test_halo_plan_properties_and_fallback():
    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd import problems
    from fps_amd.distributed import halo_plan, row_partition, shard_qp_halo

    qp = problems.pde_control_like(n=20000, m=2000, per_row=20, window=512, seed=5)
    A = qp.scipy_csr()
    for P in (1, 2, 3, 8):
        b = row_partition(qp.rowptr, P)
        plan = halo_plan(qp.rowptr, qp.colind, qp.n, b)
        assert plan is not None and plan.w_lo[0] == 0 and plan.w_hi[-1] == qp.n
        assert plan.max_exchange_doubles(2) <= 2 * 2 * 512
        cover = np.zeros(qp.n, dtype=int)
        for r in range(P):
            cover[plan.owned(r)] += 1
            loc = shard_qp_halo(qp, plan, r)
            assert loc.n == plan.w_hi[r] - plan.w_lo[r] and loc.colind.min() >= 0 and loc.colind.max() < loc.n
            # the block is exactly the rows of A restricted to the window
            import scipy.sparse as sp
            B = sp.csr_matrix((loc.vals, loc.colind, loc.rowptr), shape=(loc.m, loc.n))
            assert abs(B - A[b[r]:b[r + 1], plan.window(r)]).max() == 0
            left, right = plan.overlaps(r)
            assert left + right <= loc.n and (r > 0 or left == 0) and (r < P - 1 or right == 0)
        assert np.all(cover == 1)  # every column has exactly one owner
        x = np.arange(qp.n, dtype=float)
        assert np.array_equal(plan.assemble([x[plan.window(r)] for r in range(P)]), x)
    # a Jacobian whose rows spread over all columns has no halo layout: callers fall back to the all-reduce
    rq = problems.random_eqqp(n=5000, m=500, per_row=10)
    assert halo_plan(rq.rowptr, rq.colind, rq.n, row_partition(rq.rowptr, 4)) is None
