"""One rank of the multi-process peer-to-peer test (tests/test_gpu_p2p_ipc.py): a halo-sharded DeviceEqQP on device 0 whose
set-up collectives go through the loopback stand-in for librccl (FPSQ_RCCL_LIB, test infrastructure) and whose Krylov-loop
exchanges go -- route "p2p" -- through buffers the ranks map from each other with hipIpcOpenMemHandle.

    python tests/p2p_worker.py RANK NRANKS DIR ROUTE DELTA

Writes DIR/out_RANK.npz: window vectors, iteration counts, phi, the route the handle reports."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, nranks, d, route, delta = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], float(sys.argv[5])
    import torch  # noqa: F401  (first: one HIP runtime per process)

    import fps_amd  # noqa: F401
    from fps_amd import problems
    from fps_amd.device_qp import DeviceEqQP, rccl_unique_id
    from fps_amd.distributed import halo_plan, row_partition, shard_qp_halo

    idf = os.path.join(d, "id.bin")
    if rank == 0:
        with open(idf + ".tmp", "wb") as f:
            f.write(rccl_unique_id())
        os.rename(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 120:
            sys.exit("p2p_worker: no unique id after 120 s")
        time.sleep(0.01)
    ident = open(idf, "rb").read()

    qp = problems.pde_control_like(n=24000, m=2400, per_row=24, window=512, seed=29)
    bounds = row_partition(qp.rowptr, nranks)
    plan = halo_plan(qp.rowptr, qp.colind, qp.n, bounds)
    loc = shard_qp_halo(qp, plan, rank)
    dev = DeviceEqQP(loc, sigma=1e3, rho=1.0, delta=delta, comm=("rccl", nranks, rank, ident), halo=plan.overlaps(rank),
                     comm_route=route)
    out = {}
    rng = np.random.default_rng(3)
    v = rng.standard_normal(qp.n)
    w = plan.window(rank)
    fs, its = [], []
    for k in range(3):  # a first call (no run-ahead history), then two with the speculative tail armed
        x = qp.point(1 + k)[w]
        gx, ys, gs = np.empty(loc.n), np.empty(loc.m), np.empty(loc.n)
        f, rc = dev.objgrad(np.ascontiguousarray(x), gx=gx, ys=ys, gs=gs)
        fs.append([f, rc])
        its.append([dev.stats[0].niter, dev.stats[1].niter])
        out[f"gx{k}"], out[f"ys{k}"], out[f"gs{k}"] = gx, ys, gs
        if os.environ.get("FPSQ_TEST_P2P_DESERT") == "1" and rank == 1:  # (the bounded-wait test: this rank walks away)
            dev.close()
            return
    for ha in (2, 1):
        hv = np.empty(loc.n)
        rc = dev.hprod(np.ascontiguousarray(v[w]), hv, ha)
        out[f"hv{ha}"] = hv
        its.append([dev.stats[0].niter, dev.stats[1].niter])
        fs.append([0.0, rc])
    A = qp.scipy_csr()
    g = qp.qdiag * qp.x + qp.d
    c = A @ qp.x - qp.b
    o = [np.empty(loc.n), np.empty(loc.m), np.empty(loc.n), np.empty(loc.m)]
    rc = dev.solve_two_mixed(np.ascontiguousarray(g[w]), np.ascontiguousarray(c[bounds[rank]:bounds[rank + 1]]), *o)
    its.append([dev.stats[0].niter, dev.stats[1].niter])
    fs.append([0.0, rc])
    out["p1"], out["q1"], out["p2"], out["q2"] = o
    out["fs"], out["its"] = np.array(fs), np.array(its)
    i = dev.info()
    out["route"] = np.array([i["comm_route"], i["comm_in_launch_sums"], i["p2p_timeouts"], i["last_fused_launches"]])
    dev.close()
    np.savez(os.path.join(d, f"out_{rank}.npz"), **out)


if __name__ == "__main__":
    main()
