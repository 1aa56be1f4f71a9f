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
    a = _run(SMALL + ["--cpu-evals", "0"])
    b = _run(SMALL + ["--cpu-evals", "0", "--force-shard", "--workload", "pde-control-like n=1e6 m=1e5 nnz=1e7"])
    c = _run(["--steps", "3", "--warmup", "1", "--repeats", "1", "--cpu-evals", "0"])
    assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, (a.stderr[-800:], b.stderr[-800:], c.stderr[-800:])
    db, dc = json.loads(b.stdout.strip().splitlines()[-1]), json.loads(c.stdout.strip().splitlines()[-1])
    assert db["scaling"] == "strong" and "HALO" in db["config"]["parallelism"]
    assert db["config"]["iters_lsqr_craig_median"] == dc["config"]["iters_lsqr_craig_median"]
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
