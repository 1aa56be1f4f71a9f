"""bench.py's contract on a real GPU: ONE JSON line with the keys the driver reads, and -- for N > 1 -- a failed or stalled
multi-rank phase ends with a NON-ZERO exit code and nothing on stdout (never a replicas number under rc 0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--workload", "random-eqqp n=1e5 m=1e4 nnz=1e6", "--steps", "3", "--warmup", "1", "--repeats", "1"]


def _run(args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          env=e, timeout=timeout, cwd=ROOT)


def test_bench_prints_one_contract_line():
    r = _run(SMALL + ["--cpu-evals", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] <= 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]


def test_force_shard_world1_runs_the_sharded_path_through_rccl():
    """The halo-sharded code path with a real RCCL communicator of size 1: same iteration counts as the single-GPU run."""
    head = ["--steps", "3", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0"]
    b = _run(head + ["--force-shard"])
    b2 = _run(head + ["--force-shard", "--comm-route", "rccl"])
    c = _run(head)
    assert b.returncode == 0 and b2.returncode == 0 and c.returncode == 0, (b.stderr[-800:], b2.stderr[-800:], c.stderr[-800:])
    db, db2, dc = (json.loads(r.stdout.strip().splitlines()[-1]) for r in (b, b2, c))
    assert db["scaling"] == "strong" and "HALO" in db["config"]["parallelism"]
    assert db["config"]["comm_route"] == "p2p-ipc"  # (auto: the peer-to-peer route; with one rank there is nothing to map)
    assert db2["config"]["comm_route"] == "rccl"
    for d in (db, db2):
        assert d["config"]["iters_lsqr_craig_median"] == dc["config"]["iters_lsqr_craig_median"]
        # a communicator of one: nothing to exchange -- the sharded handle runs the single-GPU launch pattern, ONE launch per
        # joint iteration (plus the stand-alone step in front of the gated epilogue), and nothing waited too long
        assert d["config"]["comm_in_launch_sums"] is True
        assert d["config"]["loop_launches_per_iteration"] <= 1.2, d["config"]
        assert d["config"]["fuse_fallbacks"] == d["config"]["wait_timeouts"] == d["config"]["p2p_timeouts"] == 0
    assert dc["config"]["loop_launches_per_iteration"] <= 1.2 and dc["config"]["fuse_fallbacks"] == 0
    assert len([ln for ln in b.stdout.splitlines() if ln.strip()]) == 1  # the RCCL banner stays off stdout


def test_failed_multi_rank_phase_exits_nonzero_with_empty_stdout():
    """Two ranks rehearsed on ONE device: RCCL cannot form the sharded communicator (or stalls: the watchdog fires).  Either
    way the run must end non-zero, print no JSON line on stdout, and say why on stderr."""
    r = _run(["--gpus", "2", "--parallel", "shard", "--steps", "2", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0"],
             env={"FPSQ_BENCH_REHEARSE": "1", "FPSQ_BENCH_WATCHDOG": "150"}, timeout=500)
    assert r.returncode != 0
    # (the gloo backend of the rehearsal prints a connection banner on stdout; what must be absent is a result line)
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines()), r.stdout
    assert "FAILED" in r.stderr and "bench.py" in r.stderr


def _shim():
    """The multi-process loopback stand-in for librccl (tests/shim/loopback_rccl.cpp), built on demand."""
    so = os.path.join(ROOT, "tests", "shim", "libloopback_rccl.so")
    src = os.path.join(ROOT, "tests", "shim", "loopback_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-Wno-unused-result", "-o", so, src])
    return so


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_rank_sharded_bench_runs_through_the_loopback_collectives(ranks):
    """bench.py --gpus N, sharded (halo layout), executed FOR REAL as N processes -- torch.distributed.run, unique-id
    broadcast, fpsq_comm_init, the per-iteration all-gathers and grouped send / recv halo exchanges of the C++ RcclComm --
    on this box's one GPU, with the collectives carried by the loopback stand-in for librccl (RCCL refuses two ranks on one
    device).  Timing means nothing here; the run must end with rc 0, ONE JSON line saying strong scaling over a halo layout,
    every evaluation solved, and the iteration counts of the single-GPU run of the same points."""
    env = {"FPSQ_BENCH_REHEARSE": "1", "FPSQ_RCCL_LIB": _shim(), "FPSQ_BENCH_WATCHDOG": "400", "FPSQ_SHIM_TIMEOUT": "120"}
    r = _run(["--gpus", str(ranks), "--parallel", "shard", "--steps", "2", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0",
              "--no-roofline-pass"], env=env, timeout=900)
    if r.returncode != 0:
        # pytest shortens the assertion's text: what the ranks said goes into the REPORT ITSELF (captured output of a failed
        # test is printed in full) -- a side file under gpurun_out/ does not come back from the driver's run
        print(f"==== bench.py --gpus {ranks}: rc {r.returncode} ==== stdout ====\n{r.stdout}\n==== stderr ====\n{r.stderr}")
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_ranks{ranks}_failed.txt"), "w") as fh:
            fh.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and "HALO" in d["config"]["parallelism"]
    # the default route of a sharded run: peer to peer between the processes (hipIpc-mapped buffers); the stand-in library only
    # carries the set-up collectives (unique id, partial counts, the exchange of the IPC handles)
    assert d["config"]["comm_route"] == "p2p-ipc" and "PEER-TO-PEER" in d["config"]["parallelism"]
    assert d["config"]["all_solved"] is True
    one = _run(["--steps", "2", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0", "--no-roofline-pass"])
    assert one.returncode == 0, one.stderr[-1500:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert d["config"]["iters_lsqr_craig_median"] == d1["config"]["iters_lsqr_craig_median"]
    assert "replicas_alternative" in d and d["replicas_alternative"]["scaling"] == "weak"


def test_a_route_that_times_out_in_the_warm_up_falls_back_to_rccl_in_a_fresh_child():
    """First contact with a node: the peer-to-peer route sets up (handles exported, mapped, unanimous) and then its first waits
    expire (here: FPSQ_P2P_POLLS=1 -- a wait gives up after one look).  Every rank sees the verdict of all ranks after the
    warm-up, starts a FRESH child of itself with --comm-route rccl (never an exec from a process that touched the GPU) and
    exits with its code: rc 0, ONE JSON line, labelled config.comm_route = rccl and config.fell_back_from = p2p-ipc."""
    env = {"FPSQ_BENCH_REHEARSE": "1", "FPSQ_RCCL_LIB": _shim(), "FPSQ_BENCH_WATCHDOG": "400", "FPSQ_SHIM_TIMEOUT": "120",
           "FPSQ_P2P_POLLS": "1"}
    r = _run(["--gpus", "2", "--parallel", "shard", "--steps", "2", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0",
              "--no-roofline-pass", "--workload", "pde-control-like n=1e6 m=1e5 nnz=1e7"], env=env, timeout=900)
    if r.returncode != 0:
        print(f"==== rc {r.returncode} ==== stdout ====\n{r.stdout}\n==== stderr ====\n{r.stderr}")
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["comm_route"] == "rccl" and d["config"]["fell_back_from"] == "p2p-ipc"
    assert d["config"]["all_solved"] is True and d["config"]["p2p_timeouts"] == 0
    assert "fresh child with --comm-route rccl" in r.stderr
