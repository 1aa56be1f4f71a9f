#!/usr/bin/env python
"""bench.py -- penalty grad(phi) evaluations per second on the BASELINE.json headline workload.

One "step" = one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a FRESH x (a memo miss, model-Fletcherpenaltynlp.jl:236)
on the synthetic PDE-control-like equality QP  n = 1e6, m = 1e5, nnz(A) = 1e7, fp64:
user-model f, g, c  +  solve_two_mixed (the two KKT solves)  +  the ys/gs epilogue  +  Hsv  +  rho A'c.
Inputs (A, q, d, b and all evaluation points) are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant kernel = the CSR-stream
SpMV/SpMM, algorithmic bytes / HIP-event time) and `cpu_baseline` (the single-threaded C restatement of the
reference's iterative path, oracle/fps_oracle.c, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # first: one HIP runtime per process (see fps_amd/_lib.py)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import fps_amd  # noqa: E402,F401
from fps_amd import problems  # noqa: E402
from fps_amd.device_qp import DeviceEqQP  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # same guide: measured copy ceiling

WORKLOADS = {
    # name: (generator, kwargs)
    "pde-control-like n=1e6 m=1e5 nnz=1e7": (problems.pde_control_like, dict(n=1_000_000, m=100_000)),
    "random-eqqp n=1e5 m=1e4 nnz=1e6": (problems.random_eqqp, dict(n=100_000, m=10_000)),
}


def product_bytes(n, m, nnz, nrhs):
    """Algorithmic HBM bytes of ONE product-kernel launch (DESIGN.md "Roofline accounting"): matrix stream
    (8 B value + 4 B column per nonzero + row pointers) + gathered input vector + fused axpby read + output write.
    nrhs = 1 or 2 right-hand sides (the 2-RHS launch streams the matrix once for both)."""
    a = 12 * nnz + 4 * (m + 1) + 8 * nrhs * (n + 2 * m)   # y(m) = ca * A x(n) + cb * y(m)
    at = 12 * nnz + 4 * (n + 1) + 8 * nrhs * (m + 2 * n)  # y(n) = ca * A'x(m) + cb * y(n)
    return a, at


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pde-control-like n=1e6 m=1e5 nnz=1e7", choices=list(WORKLOADS))
    ap.add_argument("--op", default="objgrad", choices=["objgrad", "hprod-solves", "hprod"],
                    help="objgrad = the headline metric; hprod-solves = solve_two_least_squares (the two solves of every "
                         "hprod!, solve_linear_system.jl:79-105) on distinct right-hand-side pairs; hprod = the whole "
                         "device-resident hprod! Val(2) (model-Fletcherpenaltynlp.jl:521-570): SURVEY 8(f) rank 1")
    ap.add_argument("--delta", type=float, default=0.0, help="regularisation; 0 = first outer iteration (algo.jl:46)")
    ap.add_argument("--fuse", type=int, default=1)
    ap.add_argument("--lookahead", type=int, default=0, help="override fpsq_options.lookahead (0 = library default)")
    ap.add_argument("--cpu-evals", type=int, default=6, help="evaluations timed for cpu_baseline (0 = skip)")
    ap.add_argument("--parallel", default="auto", choices=["auto", "shard", "replicas"],
                    help="N > 1: 'shard' = rows of A sharded over the ranks, RCCL all-reduce per Krylov iteration "
                         "(fixed total work: strong scaling); 'replicas' = every rank evaluates its own points; "
                         "'auto' = shard only when the local product outweighs the all-reduce (DESIGN.md, Multi-GPU)")
    ap.add_argument("--force-shard", action="store_true", default=False,
                    help="rehearsal on one GPU: run the sharded code path (RCCL communicator of size 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FPSQ_BENCH_REHEARSE=1: rehearsal of the N > 1 control flow on a one-GPU box (every rank on device 0, gloo for the
    # barrier / max-over-ranks; only the replicas mode, which has no data-path collective)
    rehearse = os.environ.get("FPSQ_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    gen, kw = WORKLOADS[args.workload]
    qp = gen(**kw)
    n, m, nnz = qp.n, qp.m, qp.nnz
    sigma, rho = 1e3, 1.0  # parameters.jl:71,75 (first outer iteration)
    # Row sharding exchanges an n x 2 fp64 all-reduce per Krylov iteration (ring: 2 (P-1)/P * 16 n bytes over one
    # ~50 GB/s xGMI link direction) against a local product of 12 nnz / P bytes at ~5 TB/s: it pays only when
    # nnz / n >~ 130 P.  The headline (nnz / n = 10) is far below that at every P, so 'auto' runs replicas.
    shard_pays = (12.0 * nnz / max(world, 1)) / 5e12 > (2.0 * 16.0 * n) / 50e9
    want_shard = args.parallel == "shard" or (args.parallel == "auto" and shard_pays)
    sharded = (world > 1 or args.force_shard) and want_shard
    if sharded:
        from fps_amd.device_qp import rccl_unique_id
        from fps_amd.distributed import row_partition, shard_qp

        bounds = row_partition(qp.rowptr, world)
        ident = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(ident, src=0)
        local = shard_qp(qp, int(bounds[rank]), int(bounds[rank + 1]))
        model = DeviceEqQP(local, sigma=sigma, rho=rho, delta=args.delta, device=local_rank,
                           fuse_two_rhs=args.fuse, comm=("rccl", world, rank, bytes(ident.cpu().numpy())))
    else:
        extra = {"lookahead": args.lookahead} if args.lookahead > 0 else {}
        model = DeviceEqQP(qp, sigma=sigma, rho=rho, delta=args.delta, device=local_rank, fuse_two_rhs=args.fuse, **extra)

    # distinct evaluation points, resident in HBM (sharded: the same replicated x on every rank; replicas: each
    # rank evaluates its own sequence)
    K, W = args.steps, args.warmup
    xs = torch.empty((K + W, n), dtype=torch.float64, device=dev)
    for t in range(K + W):
        xs[t].copy_(torch.from_numpy(qp.point(1 + t + (0 if sharded else rank) * (K + W))))
    gx = torch.empty(n, dtype=torch.float64, device=dev)
    hp = args.op in ("hprod-solves", "hprod")
    hfull = args.op == "hprod"
    if hp:  # the points double as right-hand sides: (xs[t], xs[t] reversed) are the two n-vectors of step t
        if sharded:
            raise SystemExit("--op hprod-solves: single GPU or replicas only")
        xr = torch.flip(xs, dims=[1]).contiguous()
        hp_out = [torch.empty(k, dtype=torch.float64, device=dev) for k in (n, m, n, m)]
    torch.cuda.synchronize()

    def step(t):
        if hfull:
            return None, model.hprod(xs[t], hp_out[0])
        if hp:
            return None, model.solve_two_least_squares(xs[t], xr[t], *hp_out)
        return model.objgrad(xs[t], gx=gx)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    its = []
    for t in range(W):
        step(t)
    barrier()
    t0 = time.perf_counter()
    soft = 0
    for t in range(W, W + K):
        _, rc = step(t)
        soft |= rc
        its.append((model.stats[0].niter, model.stats[1].niter))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    evals = K if sharded else K * world  # sharded: all ranks work on the same K evaluations
    value = evals / elapsed

    # ---- roofline of the dominant kernel: a second pass over the same K points with per-launch HIP events
    model.set_profiling(True)
    pa, pat, t_ms, tot_ms = np.zeros(2), np.zeros(2), 0.0, 0.0
    for t in range(W, W + K):
        step(t)
        info = model.info()
        pa += info["last_prod_a"]
        pat += info["last_prod_at"]
        t_ms += info["last_spmv_ms"]
        tot_ms += info["last_solve_ms"]
    model.set_profiling(False)
    # Algorithmic bytes of the PRODUCTIVE product launches only: with J = max(LSQR, CRAIG iterations)
    # an evaluation runs J two-RHS A' products, J + 1 two-RHS A products (the +1 is LSQR's start-up B'u), c = Ax - b
    # with one right-hand side (A), and p1 = g - A'q1 with rho A'c (A': one raw two-RHS product on a single GPU).  Launches enqueued past convergence (host
    # run-ahead) exit at once: they count in the time, not in the bytes.  Unfused runs: 2 single-RHS products per
    # iteration of each solver.
    m_loc, nnz_loc = (local.m, local.nnz) if sharded else (m, nnz)
    a1, at1 = product_bytes(n, m_loc, nnz_loc, 1)
    a2, at2 = product_bytes(n, m_loc, nnz_loc, 2)
    nbytes = 0.0
    productive = 0
    # Vector updates riding in the product launches (single GPU, fused run): LSQR's x/w update of the previous
    # iteration in the A' launch (read v, w, x; write w, x: 5 m-passes), CRAIG's in the A launch (read v, x; write x:
    # 3 n-passes, +2 for w2 when delta != 0; read u, w, y; write w, y: 5 m-passes).
    upd_at = 0 if sharded else 8 * 5 * m
    upd_a = 0 if sharded else 8 * ((3 if args.delta == 0.0 else 5) * n + 5 * m)
    for il, ic in its:
        if hp:  # two LSQR recurrences: J two-RHS A' products (both x/w updates riding), J + 1 two-RHS A products,
            J = max(il, ic)  # then p_k = rhs_k - A'q_k with one right-hand side each
            nbytes += J * at2 + (J + 1) * a2 + 2 * at1 + (max(il - 1, 0) + max(ic - 1, 0)) * upd_at
            productive += 2 * J + 3
            if hfull and rho > 0.0:  # rho A'(A v): one more single-RHS product of each kind
                nbytes += a1 + at1
                productive += 2
        elif args.fuse:
            J = max(il, ic)
            if sharded:  # c = Ax - b, then p1 = g - A'q1 and rho A'c as two single-RHS products
                nbytes += J * at2 + (J + 1) * a2 + a1 + 2 * at1
                productive += 2 * J + 4
            else:        # single GPU: c = Ax - b rides in the CRAIG lane of LSQR's start-up A product, and the last
                         # two share one raw two-RHS product A'[q1, c] (no yin read)
                nbytes += J * at2 + (J + 1) * a2 + (at2 - 8 * 2 * n)
                productive += 2 * J + 2
            nbytes += max(il - 1, 0) * upd_at + ic * upd_a
        else:
            nbytes += (il + ic) * (a1 + at1) + a1 + a1 + 2 * at1
            productive += 2 * (il + ic) + 4
    launches = int(pa.sum() + pat.sum())
    achieved = nbytes / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "fpsq::k_spmv_rgcs (A) + fpsq::k_spmv (A'): SpMV/SpMM with fused axpby + norm partials + riding vector updates",
                "productive_launches_per_eval": productive / K, "launches_per_eval": launches / K,
                "avg_launch_us": round(1e3 * t_ms / max(launches, 1), 2),
                "avg_productive_launch_us": round(1e3 * t_ms / max(productive, 1), 2),
                "algorithmic_bytes_per_productive_launch": round(nbytes / max(productive, 1)),
                "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
                "spmv_share_of_eval_time": round(t_ms / tot_ms, 3) if tot_ms > 0 else None}

    # HBM traffic per productive launch from the committed PMC passes (profiles/: bench.py cannot run rocprofv3 on itself)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if pmc["workload"] == args.workload and args.fuse and not sharded and world == 1 and not hp:
            roofline["traffic"] = pmc["traffic_bytes_per_productive_launch"]
            roofline["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
    except (OSError, KeyError, ValueError):
        pass

    out = {
        "metric": "penalty hprod! (Val(2)) evals/sec" if hfull
        else "solve_two_least_squares calls/sec (the two KKT solves of one hprod!)" if hp
        else "penalty grad-phi evals/sec", "value": round(value, 3), "unit": "calls/s" if hp else "evals/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 4),
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "m": m, "nnz": nnz, "sigma": sigma, "rho": rho,
                   "delta": args.delta, "fuse_two_rhs": args.fuse,
                   "krylov": ("LSQR+LSQR" if hp else "LSQR+CRAIG") + ", atol=rtol=sqrt(eps) (reference defaults)",
                   "iters_lsqr_craig_median": [int(np.median([i[0] for i in its])), int(np.median([i[1] for i in its]))],
                   "all_solved": soft == 0,
                   "parallelism": "single GPU" if world == 1 else
                   (f"rows of A sharded over {world} GPUs, RCCL all-reduce of the partial A'u products (n x 2 fp64) and "
                    "of the m-vector norm partials every Krylov iteration" if sharded else
                    f"{world} independent replicas (each rank evaluates its own points), no data-path collective; row "
                    "sharding (--parallel shard) is implemented but communication-bound at nnz/n = 10")},
        "roofline": roofline,
    }

    # ---- CPU baseline: the C restatement of the reference's iterative path, one thread, bounded sample
    if rank == 0 and world == 1 and args.cpu_evals > 0:
        from oracle import oracle

        oracle.build()
        t0 = time.perf_counter()
        done = 0
        for t in range(args.cpu_evals):
            if hp:  # (hprod: the solves are all of its CPU cost but two products and three vector passes)
                r1 = qp.point(1 + W + t)
                oracle.solve_two_least_squares(m, n, qp.rowptr, qp.colind, qp.vals, args.delta, r1,
                                               np.ascontiguousarray(r1[::-1]) if not hfull else qp.qdiag * r1)
            else:
                oracle.qp_objgrad(qp, qp.point(1 + W + t), sigma, rho, args.delta)
            done += 1
            if time.perf_counter() - t0 > 30.0:
                break
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(done / dt, 4), "unit": "calls/s" if hp else "evals/s", "cores": 1,
                               "kind": "port",
                               "sample": f"{done} evaluations of the same workload (same points as the first timed "
                                         "steps), oracle/fps_oracle.c, gcc -O3 -march=native, single thread "
                                         "(the reference is single-threaded Julia; Julia is not installed)"}
        # the same restatement with OpenMP on every host core (SURVEY.md 8d asks for both); a second number, not `value`
        try:
            # (a one-GPU box exposes every host core in the affinity mask but grants a 16-core share)
            nthreads = oracle.lib(True).fpo_omp_threads(min(len(os.sched_getaffinity(0)), 16))
            t0 = time.perf_counter()
            done2 = 0
            for t in range(args.cpu_evals):
                if hp:
                    r1 = qp.point(1 + W + t)
                    oracle.solve_two_least_squares(m, n, qp.rowptr, qp.colind, qp.vals, args.delta, r1,
                                                   np.ascontiguousarray(r1[::-1]), threaded=True)
                else:
                    oracle.qp_objgrad(qp, qp.point(1 + W + t), sigma, rho, args.delta, threaded=True)
                done2 += 1
                if time.perf_counter() - t0 > 15.0:
                    break
            out["cpu_baseline"]["all_cores"] = {"value": round(done2 / (time.perf_counter() - t0), 4),
                                                "cores": int(nthreads),
                                                "kind": "port, OpenMP (oracle/libfps_oracle_omp.so)"}
        except (OSError, AttributeError) as e:  # no libgomp on this host
            out["cpu_baseline"]["all_cores"] = {"value": None, "error": str(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
