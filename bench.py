#!/usr/bin/env python
"""bench.py -- penalty grad(phi) evaluations per second on the BASELINE.json headline workload.

One "step" = one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a FRESH x (a memo miss, model-Fletcherpenaltynlp.jl:236)
on the synthetic PDE-control-like equality QP  n = 1e6, m = 1e5, nnz(A) = 1e7, fp64:
user-model f, g, c  +  solve_two_mixed (the two KKT solves)  +  the ys/gs epilogue  +  Hsv  +  rho A'c.
Inputs (A, q, d, b and all evaluation points) are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: bench.py starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything touches the GPU) and relays rank 0's
line; under the driver's own torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  The K-step timed pass
(barrier + synchronize on both sides, max over ranks) is repeated R times (>= 1 s in total); `ms_per_step` / `value`
are the MEDIAN pass, min / max are reported next to them.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant kernel = the SpMV/SpMM product
kernels, algorithmic bytes / per-launch HIP-event time, measured live in a second pass) and `cpu_baseline` (the
single-threaded C restatement of the reference's iterative path, oracle/fps_oracle.c, on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0    # same guide: measured copy ceiling
F64_MFMA_PEAK_TF = 78.6  # dense fp64 matrix peak (vendor figure)

# WORKLOADS[0] = the default = the headline: SURVEY 8(d)'s LITERAL generator (distinct hashed offsets in the 8192-column window;
# round 5 -- rounds 1-4 timed the stratified variant "pde-control-like", which stays available by name)
WORKLOADS = ["pde-control-hashed n=1e6 m=1e5 nnz=1e7", "random-eqqp n=1e5 m=1e4 nnz=1e6", "aug2dc-like N=100",
             "dense-block n=4096 m=2048",
             # the headline shape with one offset drawn per stratum of the window instead of hashed distinct offsets
             "pde-control-like n=1e6 m=1e5 nnz=1e7"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed K-step passes (0 = as many as make >= 1 s of timed work, 3..50); median reported")
    ap.add_argument("--workload", default=WORKLOADS[0], choices=WORKLOADS)
    ap.add_argument("--op", default="objgrad", choices=["objgrad", "hprod-solves", "hprod", "extras"],
                    help="objgrad = the headline metric; hprod-solves = solve_two_least_squares (the two solves of every "
                         "hprod!, solve_linear_system.jl:79-105) on distinct right-hand-side pairs; hprod = the whole "
                         "device-resident hprod! Val(2) (model-Fletcherpenaltynlp.jl:521-570): SURVEY 8(f) rank 1; extras = "
                         "solve_two_extras (LSQR + MINRES on AA' + tau I, solve_linear_system.jl:45-77: the extra solves "
                         "of hprod! Val(1)) -- timing only, no roofline accounting")
    ap.add_argument("--hessian-approx", type=int, default=2, choices=[1, 2],
                    help="--op hprod: Val(2) (model-Fletcherpenaltynlp.jl:521-570) or Val(1) (:572-634: + the solve_two_extras lanes)")
    ap.add_argument("--delta", type=float, default=None, help="regularisation; default 0 = first outer iteration "
                    "(algo.jl:46); 1e-3 for the dense-block workload")
    ap.add_argument("--alternate-delta", action="store_true", default=False,
                    help="side mode (information, not the headline): consecutive evaluations alternate between delta = 0 and "
                         "delta = sqrt(eps), i.e. between two iteration-count regimes, so that the run-ahead's expectation "
                         "(the count of the previous evaluation) is WRONG at every call: what a mispredicted speculative "
                         "epilogue costs")
    ap.add_argument("--fuse", type=int, default=1)
    ap.add_argument("--kkt-method", default="lsqr-craig", choices=["lsqr-craig", "minres-k"],
                    help="lsqr-craig: the reference's iterative path (default, the headline).  minres-k: MINRES on "
                         "K = [I A'; A -delta I] itself (fpsq_options.kkt_method = FPSQ_KKT_MINRES_K; named by "
                         "BASELINE.json's north_star / configs[1], not a path of the reference) -- with --op hprod-solves")
    ap.add_argument("--lookahead", type=int, default=0, help="override fpsq_options.lookahead (0 = library default)")
    ap.add_argument("--cpu-evals", type=int, default=6, help="evaluations timed for cpu_baseline (0 = skip)")
    ap.add_argument("--pointers", default="device", choices=["device", "host", "host+jac", "device+jac"],
                    help="device: x / gx resident in HBM (the `value` of the contract).  host: x and gx are host arrays "
                         "(PCIe inside the timed region: what a Julia caller holding host vectors gets).  host+jac: "
                         "additionally hands over new Jacobian values (fpsq_set_jacobian_values, 8 nnz bytes) before "
                         "every evaluation, like a host-resident NONLINEAR model (solve_linear_system.jl:223-228).  "
                         "device+jac: everything resident in HBM AND new Jacobian values handed over (a device pointer: one "
                         "gather launch, stream-ordered) before every evaluation -- a device-resident nonlinear model")
    ap.add_argument("--parallel", default="auto", choices=["auto", "shard", "shard-allreduce", "replicas"],
                    help="N > 1: 'shard' = rows of A sharded over the ranks in HALO layout (column-window n-vectors, "
                         "neighbour exchange + 4-double all-reduces per Krylov iteration; falls back to replicated "
                         "n-vectors + n x 2 all-reduce when the Jacobian is not banded); 'shard-allreduce' forces that "
                         "fallback; 'replicas' = every rank evaluates its own points, no collective; 'auto' = halo "
                         "sharding when it applies, else replicas")
    ap.add_argument("--comm-route", default="auto", choices=["auto", "p2p", "rccl"],
                    help="halo-sharded runs: how the per-iteration exchanges travel (include/fpsq.h fpsq_comm_set_route).  auto = "
                         "peer to peer (records written into the peers' hipIpc-mapped buffers, sequence flags, no collective "
                         "call in the loop) when every rank can export and map, else RCCL; config.parallelism names what ran")
    ap.add_argument("--fell-back-from", default=None,
                    help="(set by bench.py itself) this process is the FRESH CHILD a rank started after the named route timed out in "
                         "the warm-up of the sharded phase; recorded as config.fell_back_from")
    ap.add_argument("--force-shard", action="store_true", default=False,
                    help="rehearsal on one GPU: run the sharded code path (RCCL communicator of size 1)")
    ap.add_argument("--no-roofline-pass", action="store_true", default=False,
                    help="skip the second (per-launch HIP event) pass -- for runs under rocprofv3")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no rendezvous environment: start the ranks as a fresh child (never exec from a
    process that may touch the GPU) and relay its output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def product_bytes(n, m, nnz, nrhs):
    """Algorithmic HBM bytes of ONE product-kernel launch (DESIGN.md "Roofline accounting"): matrix stream
    (8 B value + 4 B column per nonzero + row pointers) + gathered input vector + fused axpby read + output write.
    nrhs = 1 or 2 right-hand sides (the 2-RHS launch streams the matrix once for both)."""
    a = 12 * nnz + 4 * (m + 1) + 8 * nrhs * (n + 2 * m)   # y(m) = ca * A x(n) + cb * y(m)
    at = 12 * nnz + 4 * (n + 1) + 8 * nrhs * (m + 2 * n)  # y(n) = ca * A'x(m) + cb * y(n)
    return a, at


def make_workload(name):
    from fps_amd import problems

    if name.startswith("pde-control-like"):
        return problems.pde_control_like(n=1_000_000, m=100_000)
    if name.startswith("pde-control-hashed"):
        return problems.pde_control_hashed(n=1_000_000, m=100_000)
    if name.startswith("random-eqqp"):
        return problems.random_eqqp(n=100_000, m=10_000)
    if name.startswith("aug2dc-like"):
        return problems.aug2dc_like(N=100)
    return problems.dense_block(n=4096, m=2048)


class Timer:
    """K-step passes bracketed by barrier + torch.cuda.synchronize(); max over ranks; median over R repeats."""

    def __init__(self, world, rehearse, dev):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.world, self.rehearse, self.dev = torch, dist, world, rehearse, dev

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def maxreduce(self, v):
        if self.world > 1:
            t = self.torch.tensor([v], dtype=self.torch.float64, device="cpu" if self.rehearse else self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return v

    def run(self, step, W, K, repeats, collect=None, warmed=False):
        for t in range(0 if warmed else W):
            step(t)
        times = []
        R = repeats if repeats > 0 else 1
        r = 0
        while r < R:
            self.barrier()
            t0 = time.perf_counter()
            for t in range(W, W + K):
                out = step(t)
                if collect is not None and r == 0:
                    collect(out)
            self.barrier()
            times.append(self.maxreduce(time.perf_counter() - t0))
            if r == 0 and repeats <= 0:  # as many passes as make >= 1 s of timed work (same count on every rank)
                R = int(min(50, max(3, np.ceil(1.0 / max(times[0], 1e-6)))))
            r += 1
        return np.array(times)


def bench_dense(args, rank, world, local_rank, timer, dev):
    """BASELINE configs[2]: dense-block Jacobian through the direct back-end (fp64 MFMA SYRK + blocked Cholesky + two
    right-hand sides), timed AT THE SEAM: one step = one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a fresh x with
    qds = HIPDirectQDSolver -- the user model's f, g, c (a model whose Jacobian lives in HBM, TorchEqQPModel), `jac_coord!`
    handed over as a device pointer (fpsq_dense_set_jacobian_coo), M = AA' + delta I, factorisation, solve_two_mixed (the
    reference's LDLt path, solve_linear_system.jl:206-252), the two hprod! and rho J'c of the gradient (model:372-401)."""
    import torch
    from fps_amd import nlpmodels
    from fps_amd.penalty_nlp import FletcherPenaltyNLP
    from fps_amd.qdsolver import HIPDirectQDSolver

    qp = make_workload(args.workload)
    n, m = qp.n, qp.m
    delta = 1e-3 if args.delta is None else args.delta
    model = nlpmodels.TorchEqQPModel(qp, device=local_rank)
    qds = HIPDirectQDSolver(model, 0.0, device=local_rank)
    fp = FletcherPenaltyNLP(model, sigma=1e3, rho=1.0, delta=delta, hessian_approx=2, qds=qds)
    K, W = args.steps, args.warmup
    xs = [qp.point(1 + t + rank * (K + W)) for t in range(K + W)]
    tsum = np.zeros(3)

    def step(t):
        fp.objgrad(xs[t])
        assert qds.factorized
        i = qds.info()
        return np.array([i["last_syrk_ms"], i["last_chol_ms"], i["last_solve_ms"]])

    def collect(v):
        tsum.__iadd__(v)

    torch.cuda.synchronize()
    times = timer.run(step, W, K, args.repeats, collect)
    med = float(np.median(times))
    # algorithmic flops of one step: SYRK m^2 n (lower triangle of the Gram matrix, 2 flops per multiply-add pair
    # counted on half the entries), Cholesky m^3 / 3, two triangular solves x two right-hand sides 4 m^2,
    # A g, A' q1, A' q2: 6 m n
    flops = 1.0 * m * m * n + m ** 3 / 3.0 + 4.0 * m * m + 6.0 * m * n
    dev_ms = tsum / K
    tf = flops / (dev_ms.sum() * 1e-3) / 1e12
    out = {"metric": "penalty grad-phi evals/sec", "value": round(K * world / med, 3), "unit": "evals/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * med / K, 4),
           "ms_per_step_min": round(1e3 * times.min() / K, 4), "ms_per_step_max": round(1e3 * times.max() / K, 4),
           "repeats": int(times.size), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": args.workload, "n": n, "m": m, "delta": delta,
                      "path": "direct back-end behind the QDSolver seam: one step = FletcherPenaltyNLP.objgrad (user model "
                              "with its Jacobian in HBM; jac_coord! values taken in place) -> M = AA' + delta I "
                              "(v_mfma_f64_16x16x4_f64), blocked Cholesky, solve_two_mixed with 2 RHS, gradient assembly",
                      "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(tf / F64_MFMA_PEAK_TF, 4), "traffic": None,
                        "kernel": "fpsq dense SYRK + blocked Cholesky + triangular solves (device time of the three "
                                  "phases, HIP events on the solver's stream)",
                        "flops_per_step": flops,
                        "device_ms": {"syrk": round(dev_ms[0], 4), "cholesky": round(dev_ms[1], 4),
                                      "solve": round(dev_ms[2], 4)},
                        "syrk_tflops": round(1.0 * m * m * n / (dev_ms[0] * 1e-3) / 1e12, 2),
                        "cholesky_tflops": round(m ** 3 / 3.0 / (dev_ms[1] * 1e-3) / 1e12, 2)}}
    if rank == 0 and world == 1 and args.cpu_evals > 0:
        import scipy.linalg as sla

        Ad = qp.scipy_csr().toarray()
        t0 = time.perf_counter()
        done = 0
        for t in range(args.cpu_evals):
            M = Ad @ Ad.T + delta * np.eye(m)
            cf = sla.cho_factor(M, lower=True)
            x = qp.point(1 + t)
            g, c = qp.qdiag * x + qp.d, Ad @ x - qp.b
            q1 = sla.cho_solve(cf, Ad @ g)
            q2 = -sla.cho_solve(cf, c)
            _ = (g - Ad.T @ q1, -Ad.T @ q2)
            done += 1
            if time.perf_counter() - t0 > 20.0:
                break
        dt = time.perf_counter() - t0
        try:
            from threadpoolctl import threadpool_info
            nth = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:  # noqa: BLE001
            nth = len(os.sched_getaffinity(0))
        out["cpu_baseline"] = {"value": round(done / dt, 4), "unit": "evals/s", "cores": int(nth), "kind": "port",
                               "sample": f"{done} steps of the same dense normal-equations path with numpy/scipy "
                                         "(LAPACK dpotrf/dpotrs, BLAS dgemm) on the host: a stand-in for the "
                                         "reference's LDLFactorizations.jl path, which cannot run here (no Julia)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    qds.close()


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(self_launch(args))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 "(or without a torch.distributed environment, bench.py then starts its own ranks)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch  # first: one HIP runtime per process (see fps_amd/_lib.py)
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import fps_amd  # noqa: F401
    from fps_amd.device_qp import DeviceEqQP

    # FPSQ_BENCH_REHEARSE=1: rehearsal of the N > 1 control flow on a one-GPU box (every rank on device 0, gloo for the
    # barrier / max-over-ranks; only the replicas mode: RCCL refuses two ranks on one device)
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
    timer = Timer(world, rehearse, dev)

    # ---- N > 1: a stalled or failed multi-rank phase must never be scored.  `fail` (watchdog timer or exception path)
    # restores stdout if it was redirected, writes the reason -- and the replicas measurement if there is one, labelled
    # as NOT the requested run -- to STDERR, and ends every rank with a non-zero exit code.  Nothing is printed on stdout.
    state = {"saved_fd": None, "replicas": None}

    def fail(reason):
        try:
            if state["saved_fd"] is not None:
                os.dup2(state["saved_fd"], 1)
            sys.stderr.write("bench.py: FAILED on rank %d: %s\n" % (rank, reason))
            if rank == 0 and state["replicas"] is not None:
                sys.stderr.write("bench.py: FALLBACK information only (independent replicas, NOT the requested sharded "
                                 "run): " + json.dumps(state["replicas"]) + "\n")
            sys.stderr.flush()
        finally:
            os._exit(3)

    watchdog = None
    if world > 1:
        import threading

        wd_s = float(os.environ.get("FPSQ_BENCH_WATCHDOG", "300"))
        watchdog = threading.Timer(wd_s, fail, args=(f"the {world}-rank run did not finish within {wd_s:.0f} s (watchdog)",))
        watchdog.daemon = True
        watchdog.start()

    if args.workload.startswith("dense-block"):
        try:
            bench_dense(args, rank, world, local_rank, timer, dev)
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            fail(f"dense-block run failed: {e}")
        if watchdog is not None:
            watchdog.cancel()
        if world > 1:
            dist.destroy_process_group()
        return

    qp = make_workload(args.workload)
    n, m, nnz = qp.n, qp.m, qp.nnz
    sigma, rho = 1e3, 1.0  # parameters.jl:71,75 (first outer iteration)
    delta = 0.0 if args.delta is None else args.delta
    extra = {"lookahead": args.lookahead} if args.lookahead > 0 else {}
    mk = args.kkt_method == "minres-k"
    if mk:
        if args.op != "hprod-solves":
            raise SystemExit("--kkt-method minres-k: use --op hprod-solves (the fused QP entries keep the reference's method)")
        extra["kkt_method"] = 1

    # ---- how the N ranks share the work
    plan = None
    layout = "single"
    if world > 1 or args.force_shard:
        from fps_amd.distributed import halo_plan, row_partition, shard_qp, shard_qp_halo

        bounds = row_partition(qp.rowptr, world)
        if args.parallel in ("auto", "shard"):
            plan = halo_plan(qp.rowptr, qp.colind, n, bounds)
        if plan is not None:
            layout = "halo"
        elif args.parallel in ("shard", "shard-allreduce") or args.force_shard:
            layout = "allreduce"
        else:
            layout = "replicas"  # auto without a banded Jacobian, or asked for
        if args.parallel == "replicas" and not args.force_shard:
            layout = "replicas"
    sharded = layout in ("halo", "allreduce")
    # (FPSQ_BENCH_REHEARSE with a sharded layout: RCCL refuses two ranks on one device, which exercises the fallback
    # to the replicas line below -- the sharded numerics themselves are covered by the in-process loopback tests)
    if sharded and args.pointers != "device":
        sys.exit("bench.py: --pointers host: single GPU or replicas only")

    def build_model(kind):
        if kind in ("halo", "allreduce"):
            from fps_amd.device_qp import rccl_unique_id

            ident = torch.zeros(128, dtype=torch.uint8, device="cpu" if rehearse else dev)  # (gloo in a rehearsal)
            if rank == 0:
                ident.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(ident, src=0)
            comm = ("rccl", world, rank, bytes(ident.cpu().numpy()))
            if kind == "halo":
                loc = shard_qp_halo(qp, plan, rank)
                return loc, DeviceEqQP(loc, sigma=sigma, rho=rho, delta=delta, device=local_rank,
                                       fuse_two_rhs=args.fuse, comm=comm, halo=plan.overlaps(rank),
                                       comm_route=args.comm_route, **extra)
            loc = shard_qp(qp, int(bounds[rank]), int(bounds[rank + 1]))
            return loc, DeviceEqQP(loc, sigma=sigma, rho=rho, delta=delta, device=local_rank, fuse_two_rhs=args.fuse,
                                   comm=comm, **extra)
        return qp, DeviceEqQP(qp, sigma=sigma, rho=rho, delta=delta, device=local_rank, fuse_two_rhs=args.fuse, **extra)

    # ---- N > 1, sharded: FIRST the same ranks as independent replicas (no data-path collective) -- a second number for
    # the record (`replicas_alternative`), never the `value` of a sharded run
    replicas_alt = None
    if sharded and world > 1 and args.op == "objgrad":
        _, rep = build_model("replicas")
        K0, W0 = args.steps, args.warmup
        xs0 = torch.empty((K0 + W0, n), dtype=torch.float64, device=dev)
        for t in range(K0 + W0):
            xs0[t].copy_(torch.from_numpy(qp.point(1 + t + rank * (K0 + W0))))
        gx0 = torch.empty(n, dtype=torch.float64, device=dev)
        its0 = []
        t0s = timer.run(lambda t: rep.objgrad(xs0[t], gx=gx0), W0, K0, 3,
                        lambda out: its0.append((rep.stats[0].niter, rep.stats[1].niter)))
        med0 = float(np.median(t0s))
        replicas_alt = {"value": round(K0 * world / med0, 3), "unit": "evals/s", "scaling": "weak",
                        "ms_per_step": round(1e3 * med0 / K0, 4),
                        "note": "same ranks, each evaluating its own points with the whole Jacobian, no collective"}
        replicas_alt["iters_lsqr_craig_median"] = [int(np.median([i[0] for i in its0])), int(np.median([i[1] for i in its0]))]
        state["replicas"] = replicas_alt
        rep.close()
        del xs0, gx0

    # RCCL prints a version banner on stdout when a communicator is created: keep stdout for the ONE JSON line
    if sharded:
        sys.stdout.flush()
        state["saved_fd"] = os.dup(1)
        os.dup2(2, 1)
    try:
        local, model = build_model(layout)
    except Exception as e:  # noqa: BLE001  (RCCL unavailable / communicator cannot be formed)
        if world == 1:
            raise
        fail(f"sharded set-up failed: {e}")
    finally:
        if sharded:
            sys.stdout.flush()
            os.dup2(state["saved_fd"], 1)
            os.close(state["saved_fd"])
            state["saved_fd"] = None
    n_loc = local.n  # window length in halo layout, n otherwise

    # distinct evaluation points, resident in HBM (sharded: every rank holds its window of the SAME points; replicas:
    # each rank evaluates its own sequence)
    K, W = args.steps, args.warmup
    host_ptr = args.pointers in ("host", "host+jac")
    win = plan.window(rank) if layout == "halo" else slice(0, n)

    def points(off, length=None):
        arr = np.empty((K + W, n_loc if length is None else length))
        for t in range(K + W):
            arr[t] = qp.point(1 + t + off)[win if length is None else slice(0, length)]
        return arr

    xs_h = points(0 if sharded else rank * (K + W))
    xs = xs_h if host_ptr else torch.from_numpy(xs_h).to(dev)
    gx = np.empty(n_loc) if host_ptr else torch.empty(n_loc, dtype=torch.float64, device=dev)
    hp = args.op in ("hprod-solves", "hprod", "extras")
    hfull = args.op == "hprod"
    extras = args.op == "extras"
    if hp:  # the points double as right-hand sides: (xs[t], xs[t] reversed) are the two n-vectors of step t
        if sharded or host_ptr:
            raise SystemExit("--op hprod-solves / hprod: device pointers, single GPU or replicas only")
        xr = torch.flip(xs, dims=[1]).contiguous()
        hp_out = [torch.empty(k, dtype=torch.float64, device=dev) for k in (n, m, n, m)]
        if extras:  # rhs2 of step t: the first m entries of the reversed point
            xm = xr[:, :m].contiguous()
    jac_vals = np.ascontiguousarray(qp.vals) if args.pointers == "host+jac" else None
    if args.pointers == "device+jac":
        jac_vals = torch.from_numpy(np.ascontiguousarray(qp.vals)).to(dev)
    torch.cuda.synchronize()

    def make_step(mdl, pts, out):
        if not host_ptr:
            pts = pts.unbind(0)  # (the row handles once, outside the timed region: the points themselves are resident already)

        def step(t):
            if hfull:
                return None, mdl.hprod(pts[t], hp_out[0], args.hessian_approx)
            if extras:
                mdl._order(pts[t], xm[t], hp_out[1], hp_out[3])
                return None, mdl._check(mdl._lib.fpsq_solve_two_extras(mdl._h, pts[t].data_ptr(), xm[t].data_ptr(),
                                                                      hp_out[1].data_ptr(), hp_out[3].data_ptr(),
                                                                      mdl.stats))
            if hp:
                return None, mdl.solve_two_least_squares(pts[t], xr[t], *hp_out)
            if jac_vals is not None:
                mdl.set_jacobian_values(jac_vals)
            if args.alternate_delta:
                mdl.set_delta(0.0 if t % 2 == 0 else float(np.sqrt(np.finfo(float).eps)))
            return mdl.objgrad(pts[t], gx=out)
        return step

    step = make_step(model, xs, gx)
    its, soft = [], [0]

    def collect(out):
        soft[0] |= out[1]
        its.append((model.stats[0].niter, model.stats[1].niter))

    # ---- first contact with an N-GPU node: the peer-to-peer route may set up and THEN time out (a link that maps but does not
    # deliver, a peer that is too slow for the bounds).  The warm-up of the sharded phase is where that shows: every rank takes
    # the verdict of ALL ranks (max over the ranks of "my warm-up failed") and, if the route was not RCCL already, starts a FRESH
    # CHILD of itself -- a new process, never an exec from one that touched the GPU -- with --comm-route rccl on the next port,
    # relays its output and exits with its code.  The child's line says so (config.fell_back_from); only a child that fails too
    # ends the run with a non-zero code and an empty stdout.
    warm_err = None
    if sharded and world > 1:
        try:
            for t in range(W):
                step(t)
        except Exception as e:  # noqa: BLE001
            warm_err = e
        try:
            flag = torch.tensor([1.0 if warm_err is not None else 0.0], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            any_failed = bool(flag.item() > 0)
        except Exception as e:  # noqa: BLE001  (the process group itself is gone)
            fail(f"warm-up of the sharded phase failed ({warm_err}) and the ranks could not agree on a fallback: {e}")
        if any_failed:
            route_now = DeviceEqQP.ROUTE_NAMES.get(model.info()["comm_route"], "?")
            if args.comm_route == "rccl" or args.fell_back_from is not None or route_now == "rccl":
                fail(f"warm-up of the sharded phase failed on the {route_now} route: {warm_err}")
            sys.stderr.write(f"bench.py: rank {rank}: the {route_now} route failed in the warm-up ({warm_err}); every rank starts a "
                             "fresh child with --comm-route rccl\n")
            sys.stderr.flush()
            try:
                model.close()
            except Exception:  # noqa: BLE001
                pass
            if watchdog is not None:
                watchdog.cancel()  # (the child has its own)
            argv = [a for a in sys.argv[1:]]
            if "--comm-route" in argv:
                i = argv.index("--comm-route")
                del argv[i:i + 2]
            # (a rendezvous of its own: the children's rank 0 hosts the store on the next port -- torchrun's agent store stays
            # with the parents)
            env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 1),
                       TORCHELASTIC_USE_AGENT_STORE="False")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), *argv, "--comm-route", "rccl", "--fell-back-from", route_now],
                               env=env)
            os._exit(r.returncode)
    try:
        times = timer.run(step, W, K, args.repeats, collect, warmed=sharded and world > 1)
    except Exception as e:  # noqa: BLE001  (a collective failed: every rank must end, non-zero)
        if world == 1:
            raise
        fail(f"timed run failed: {e}")
    med = float(np.median(times))
    evals = K if sharded else K * world  # sharded: all ranks work on the same K evaluations
    value = evals / med

    # ---- roofline of the dominant kernel: a second pass over the same K points with per-launch HIP events
    roofline = None
    if not args.no_roofline_pass and not extras and not (hfull and args.hessian_approx == 1) and not args.alternate_delta:  # (timing only for those)
        model.set_profiling(True)
        pa, pat, t_ms, tot_ms, nfused, nmulti_saved = np.zeros(2), np.zeros(2), 0.0, 0.0, 0, 0
        for t in range(W, W + K):
            step(t)
            info = model.info()
            pa += info["last_prod_a"]
            pat += info["last_prod_at"]
            nfused += info.get("last_fused_launches", 0)
            nmulti_saved += info.get("last_multi_iterations", 0) - info.get("last_multi_launches", 0)
            t_ms += info["last_spmv_ms"]
            tot_ms += info["last_solve_ms"]
        model.set_profiling(False)
        # Algorithmic bytes of the PRODUCTIVE product launches only: with J = max(LSQR, CRAIG
        # iterations) an evaluation runs J two-RHS A' products, J + 1 two-RHS A products (the +1 is LSQR's start-up B'u),
        # c = Ax - b with one right-hand side (A), and p1 = g - A'q1 with rho A'c (A': one raw two-RHS product on a single
        # GPU).  Launches enqueued past convergence (host run-ahead) exit at once: they count in the time, not in the
        # bytes.  Unfused runs: 2 single-RHS products per iteration of each solver.
        m_loc, nnz_loc = local.m, local.nnz
        a1, at1 = product_bytes(n_loc, m_loc, nnz_loc, 1)
        a2, at2 = product_bytes(n_loc, m_loc, nnz_loc, 2)
        nbytes, productive = 0.0, 0
        # Vector updates riding in the product launches (single GPU, fused run): LSQR's x/w update of the previous
        # iteration in the A' launch (read v, w, x; write w, x: 5 m-passes), CRAIG's in the A launch (read v, x; write x:
        # 3 n-passes, +2 for w2 when delta != 0; read u, w, y; write w, y: 5 m-passes).
        # (the halo layout runs the single-GPU launch pattern on the rank's block: riding updates, fast start, paired
        # epilogue product; the replicated-n layout keeps separate update launches and single-RHS epilogue products)
        separate = layout == "allreduce"
        upd_at = 0 if separate else 8 * 5 * m_loc
        upd_a = 0 if separate else 8 * ((3 if delta == 0.0 else 5) * n_loc + 5 * m_loc)
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
                if separate:  # c = Ax - b, then p1 = g - A'q1 and rho A'c as two single-RHS products
                    nbytes += J * at2 + (J + 1) * a2 + a1 + 2 * at1
                    productive += 2 * J + 4
                else:        # single GPU: c = Ax - b rides in the CRAIG lane of LSQR's start-up A product, and the
                             # last two share one raw two-RHS product A'[q1, c] (no yin read)
                    nbytes += J * at2 + (J + 1) * a2 + (at2 - 8 * 2 * n_loc)
                    productive += 2 * J + 2
                nbytes += max(il - 1, 0) * upd_at + ic * upd_a
            else:
                nbytes += (il + ic) * (a1 + at1) + a1 + a1 + 2 * at1
                productive += 2 * (il + ic) + 4
        # One launch per joint iteration (k_iter_fused): such a launch is counted as an A' product AND as an A product above; as a
        # LAUNCH it is one, and it is productive when its iteration is (the run-ahead enqueues whole iterations)
        launches = int(pa.sum() + pat.sum()) - nfused
        if nfused:
            productive -= sum(min(max(il, ic), nfused // max(len(its), 1)) for il, ic in its)
        # several iterations per launch (k_iter_multi): K iterations are ONE launch -- and one productive launch
        launches -= nmulti_saved
        productive -= nmulti_saved
        achieved = nbytes / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": ("fpsq::k_iter_fused (the A' and the A product of a joint iteration in one grid: row groups start as the "
                               "A' blocks they gather from publish themselves; both scalar steps by leader workgroups; vector updates "
                               "riding) + k_spmv_rgcs / k_spmv for the start-up and epilogue products" if nfused else
                               "fpsq::k_spmv_rgcs (A) + fpsq::k_spmv_atl / k_spmv (A'): SpMV/SpMM with fused axpby + norm "
                               "partials + riding vector updates + the previous product's scalar steps (leader workgroups)"),
                    "fused_iteration_launches_per_eval": nfused / K,
                    "iterations_sharing_a_launch_per_eval": (nmulti_saved / K) if nmulti_saved else 0,
                    "productive_launches_per_eval": productive / K, "launches_per_eval": launches / K,
                    "avg_launch_us": round(1e3 * t_ms / max(launches, 1), 2),
                    "avg_productive_launch_us": round(1e3 * t_ms / max(productive, 1), 2),
                    "algorithmic_bytes_per_productive_launch": round(nbytes / max(productive, 1)),
                    "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
                    "spmv_share_of_eval_time": round(t_ms / tot_ms, 3) if tot_ms > 0 else None,
                    "whole_eval_frac_of_peak": round(nbytes / K / (med / K) / 1e9 / HBM_PEAK_GBS, 4) if not sharded else None}
        # HBM traffic per productive launch: PMC counters cannot be read from inside the run (rocprofv3 writes them when
        # the process ends); the number below is from the COMMITTED profile of this same command, labelled as such -- and
        # only while the kernel sources are the ones the profile was taken on (tools/pmc_traffic.py stores their hash):
        # a stale profile is refused, not quoted
        import hashlib

        hh = hashlib.sha256()
        cs = os.path.join(ROOT, "fletcherpenaltysolver.jl_amd", "csrc")
        for fn in sorted(os.listdir(cs)):
            if fn.endswith(".hip") or fn.endswith(".hip.h"):
                hh.update(fn.encode())
                hh.update(open(os.path.join(cs, fn), "rb").read())
        sha = hh.hexdigest()[:16]
        if args.fuse and not sharded and world == 1 and not hp and args.pointers == "device":
            roofline["traffic_source"] = "no committed PMC profile of this workload (profiles/rNN_pmc_traffic.json)"
            for prof in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))
                    if pmc["workload"] != args.workload:
                        continue
                    if pmc.get("kernel_sources_sha16") != sha:
                        roofline["traffic_source"] = (f"REFUSED: profiles/{prof} was taken on other kernel sources (sha16 "
                                                      f"{pmc.get('kernel_sources_sha16')}, this tree {sha}): re-take it with tools/r5_profiles.sh")
                        break
                    roofline["traffic"] = pmc["traffic_bytes_per_productive_launch"]
                    roofline["traffic_source"] = (f"committed profile profiles/{prof} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                  f"this command on kernel sources sha16 {sha}, git {pmc.get('git_head_when_post_processed')}; "
                                                  "not measured in this run)")
                    break
                except (OSError, KeyError, ValueError):
                    continue

    final = model.info()
    route = final["comm_route"] if sharded else 0
    insum = bool(final.get("comm_in_launch_sums", 0)) if sharded else False
    how = ("PEER-TO-PEER route, sums over the ranks formed INSIDE the launches that need them (the leader workgroups write their "
           "rank's local sums into the peers' hipIpc-mapped receive areas and add the ranks' rows up in rank order; no gather "
           "kernel, no collective call inside the loop)" if route == 2 and insum
           else "PEER-TO-PEER route: records written straight into the peers' hipIpc-mapped buffers + sequence flags, one "
           "one-workgroup kernel per exchange, no collective call inside the loop" if route == 2
           else "RCCL route (a communicator of ONE rank: nothing to exchange)" if insum
           else "RCCL route: grouped ncclSend/ncclRecv + ncclAllGather on the solver's stream")
    par = {"single": "single GPU",
           "halo": f"rows of A sharded over {world} GPUs, HALO layout: each rank holds its column window of the n-vectors and "
                   f"runs the single-GPU launch pattern on its block; {how}; per Krylov iteration one exchange of the A'u "
                   f"overlap regions with the neighbours (<= {plan.max_exchange_doubles() if plan else 0} doubles per rank) + "
                   "one all-gather of the norm partials per product (summed in rank order by the leaders of the next product)",
           "allreduce": f"rows of A sharded over {world} GPUs, replicated n-vectors: RCCL all-reduce of the partial A'u "
                        "products (n x 2 fp64) and of the m-vector norm partials every Krylov iteration",
           "replicas": f"{world} independent replicas (each rank evaluates its own points), no data-path collective"}[layout]
    out = {
        "metric": f"penalty hprod! (Val({args.hessian_approx})) evals/sec" if hfull
        else "solve_two_extras calls/sec (LSQR + MINRES lanes, the extra solves of hprod! Val(1))" if extras
        else "solve_two_least_squares calls/sec (the two KKT solves of one hprod!)" if hp
        else "penalty grad-phi evals/sec", "value": round(value, 3), "unit": "calls/s" if hp else "evals/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * med / K, 4),
        "ms_per_step_min": round(1e3 * times.min() / K, 4), "ms_per_step_max": round(1e3 * times.max() / K, 4),
        "repeats": int(times.size),
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "m": m, "nnz": nnz, "sigma": sigma, "rho": rho,
                   "delta": "alternating 0 / sqrt(eps) (every evaluation mispredicts the run-ahead)" if args.alternate_delta else delta,
                   "fuse_two_rhs": args.fuse, "pointers": args.pointers,
                   "krylov": ("MINRES on K = [I A'; A -delta I], two systems in lock-step" if mk else "LSQR+MINRES" if extras
                              else "LSQR+LSQR" if hp else "LSQR+CRAIG") + ", atol=rtol=sqrt(eps) (reference defaults)",
                   "iters_lsqr_craig_median": [int(np.median([i[0] for i in its])), int(np.median([i[1] for i in its]))],
                   **({"iters_lsqr_craig_seen": sorted(set(its))} if args.alternate_delta else {}),
                   "all_solved": soft[0] == 0, "parallelism": par,
                   # the Krylov loop of the LAST evaluation: kernel launches per joint iteration (products, exchanges, stand-alone
                   # steps); and the handle's cumulative "something waited too long" counters (fpsq_info): all 0 on a healthy run
                   "loop_launches_per_iteration": round(final["last_loop_launches"] / max(final["last_loop_iterations"], 1), 3),
                   "fuse_fallbacks": final["fuse_fallbacks"], "wait_timeouts": final["wait_timeouts"],
                   "p2p_timeouts": final["p2p_timeouts"],
                   **({"comm_route": DeviceEqQP.ROUTE_NAMES.get(route, str(route)), "comm_in_launch_sums": insum}
                      if sharded else {}),
                   **({"fell_back_from": args.fell_back_from} if args.fell_back_from else {})},
        "roofline": roofline,
    }

    if replicas_alt is not None:
        out["replicas_alternative"] = replicas_alt

    # ---- CPU baseline: the C restatement of the reference's iterative path, one thread, bounded sample
    if rank == 0 and world == 1 and args.cpu_evals > 0:
        from oracle import oracle

        oracle.build()
        t0 = time.perf_counter()
        done = 0
        for t in range(args.cpu_evals):
            if extras:
                r1 = qp.point(1 + W + t)
                oracle.solve_two_extras(m, n, qp.rowptr, qp.colind, qp.vals, delta, r1, np.ascontiguousarray(r1[::-1][:m]))
            elif mk:
                r1 = qp.point(1 + W + t)
                oracle.minres_kkt(m, n, qp.rowptr, qp.colind, qp.vals, delta, bp=r1)
                oracle.minres_kkt(m, n, qp.rowptr, qp.colind, qp.vals, delta, bp=np.ascontiguousarray(r1[::-1]))
            elif hp:  # (hprod: the solves are all of its CPU cost but two products and three vector passes)
                r1 = qp.point(1 + W + t)
                oracle.solve_two_least_squares(m, n, qp.rowptr, qp.colind, qp.vals, delta, r1,
                                               np.ascontiguousarray(r1[::-1]) if not hfull else qp.qdiag * r1)
            else:
                oracle.qp_objgrad(qp, qp.point(1 + W + t), sigma, rho, delta)
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
                if extras or mk:
                    break
                if hp:
                    r1 = qp.point(1 + W + t)
                    oracle.solve_two_least_squares(m, n, qp.rowptr, qp.colind, qp.vals, delta, r1,
                                                   np.ascontiguousarray(r1[::-1]), threaded=True)
                else:
                    oracle.qp_objgrad(qp, qp.point(1 + W + t), sigma, rho, delta, threaded=True)
                done2 += 1
                if time.perf_counter() - t0 > 15.0:
                    break
            if done2 == 0:
                raise AttributeError("no OpenMP leg for this op")
            out["cpu_baseline"]["all_cores"] = {"value": round(done2 / (time.perf_counter() - t0), 4),
                                                "cores": int(nthreads),
                                                "kind": "port, OpenMP (oracle/libfps_oracle_omp.so)"}
        except (OSError, AttributeError) as e:  # no libgomp on this host
            out["cpu_baseline"]["all_cores"] = {"value": None, "error": str(e)}
    if world > 1:  # every rank reached the end of its measurements: only then is the line printed
        try:
            timer.barrier()
        except Exception as e:  # noqa: BLE001
            fail(f"final barrier failed: {e}")
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
