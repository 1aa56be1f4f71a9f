"""The peer-to-peer exchange route BETWEEN PROCESSES (include/fpsq.h fpsq_comm_set_route, csrc/fpsq.hip IpcComm): 2 and 3
ranks, one process each, all on this box's one GPU.  hipIpc handles open between processes that share a device exactly as
between the GPUs of a node, so everything but the link is exercised: export / all-gather / import of the handles at the first
solve, the unanimous decision, k_p2p_gather / k_p2p_halo writing into the peers' mapped buffers, sequence flags, bounded
waits, the phi gather.  The set-up collectives go through the loopback stand-in for librccl (tests/shim: RCCL refuses two
ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
from fps_amd.distributed import halo_plan, row_partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SE = float(np.sqrt(np.finfo(float).eps))


def _shim():
    so = os.path.join(ROOT, "tests", "shim", "libloopback_rccl.so")
    src = os.path.join(ROOT, "tests", "shim", "loopback_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-Wno-unused-result", "-o", so, src])
    return so


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _run_ranks(nranks, route, delta, tmp, extra_env=None, timeout=300):
    env = dict(os.environ, FPSQ_RCCL_LIB=_shim(), FPSQ_SHIM_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), str(nranks), str(tmp),
                               route, repr(delta)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(nranks)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:  # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs


def _reference(delta):
    qp = problems.pde_control_like(n=24000, m=2400, per_row=24, window=512, seed=29)
    ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
    out, its, fs = {}, [], []
    v = np.random.default_rng(3).standard_normal(qp.n)
    for k in range(3):
        gx, ys, gs = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n)
        f, rc = ref.objgrad(qp.point(1 + k), gx=gx, ys=ys, gs=gs)
        fs.append(f)
        its.append([ref.stats[0].niter, ref.stats[1].niter])
        out[f"gx{k}"], out[f"ys{k}"], out[f"gs{k}"] = gx, ys, gs
    for ha in (2, 1):
        hv = np.empty(qp.n)
        assert ref.hprod(v, hv, ha) == 0
        out[f"hv{ha}"] = hv
        its.append([ref.stats[0].niter, ref.stats[1].niter])
    A = qp.scipy_csr()
    o = [np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)]
    ref.solve_two_mixed(qp.qdiag * qp.x + qp.d, A @ qp.x - qp.b, *o)
    its.append([ref.stats[0].niter, ref.stats[1].niter])
    out["p1"], out["q1"], out["p2"], out["q2"] = o
    ref.close()
    return qp, out, np.array(its), fs


@pytest.mark.parametrize("nranks,route", [(2, "p2p"), (3, "p2p"), (2, "auto"), (2, "rccl"), (3, "p2p-two-launch-halo"),
                                          (2, "p2p-in-launch"), (3, "p2p-in-launch"), (3, "p2p-in-launch-late-rank"),
                                          (2, "p2p-in-launch-fused"), (3, "p2p-in-launch-fused"),
                                          (2, "p2p-in-launch-fused-hiccup"), (3, "p2p-in-launch-hiccup")])
@pytest.mark.parametrize("delta", [0.0, SE])
def test_ranks_in_separate_processes_exchange_peer_to_peer(tmp_path, nranks, route, delta):
    """Same iteration counts as the single-GPU handle, vectors to 1e-9 (only the order of the reductions differs), phi
    BITWISE the same on every rank (the four sums are gathered and added in rank order), overlaps bitwise identical on the
    two ranks sharing them; the handles report the route they run on (p2p and auto: peer to peer; rccl: the collectives).
    On the peer-to-peer route the halo exchange and the finish of the overlap rows are ONE launch (k_p2p_halo_finish: the
    finish workgroups wait for the neighbours' records themselves); "p2p-two-launch-halo" (FPSQ_HALO_FUSE=0) keeps the
    exchange kernel and the finish kernel apart.  "p2p-in-launch" (round 5): the sums over the ranks are formed inside the
    launches that need them (fpsq_krylov.hip.h xch_sum) -- what ranks with a device of their own do by default; between
    processes SHARING this box's GPU it has to be forced (FPSQ_LX=2; the grids of this problem are resident all at once);
    "-late-rank": rank 1 holds every one of its pushes back by ~100 us (FPSQ_DEBUG_XCH_DELAY): nothing may depend on when a
    row arrives.  "-hiccup": rank 1 holds the pushes of every 128th exchange back by 150 ms (FPSQ_DEBUG_XCH_LONG_DELAY_MS) -- a host
    that was descheduled in the middle of a solve: the leaders of the other ranks wait for it, and so must every workgroup that
    waits for THEM (RideArgs::more; with one GPU's bound of tens of milliseconds those gave up -- the test below)."""
    qp, want, its_ref, fs_ref = _reference(delta)
    extra = None
    fused = False
    in_launch = route.startswith("p2p-in-launch")
    if route == "p2p-two-launch-halo":
        route, extra = "p2p", {"FPSQ_HALO_FUSE": "0"}
    elif in_launch:
        extra = {"FPSQ_LX": "2"}
        if route.endswith("late-rank"):
            extra["FPSQ_DEBUG_XCH_DELAY"] = "2"
        if route.endswith("-hiccup"):
            extra.update(FPSQ_DEBUG_XCH_DELAY="2", FPSQ_DEBUG_XCH_LONG_DELAY_MS="150")
            route = route[:-len("-hiccup")]
        if route.endswith("fused"):   # ONE launch per joint iteration, the halo exchange and finish inside (k_iter_fused<.., HALO>)
            extra["FPSQ_FUSE_ITER"] = "2"
        fused = route.endswith("fused")
        route = "p2p"
    rcs, outs = _run_ranks(nranks, route, delta, tmp_path, extra_env=extra)
    assert all(rc == 0 for rc in rcs), [o[1][-1500:] for o in outs]
    res = [np.load(os.path.join(tmp_path, f"out_{r}.npz")) for r in range(nranks)]
    bounds = row_partition(qp.rowptr, nranks)
    plan = halo_plan(qp.rowptr, qp.colind, qp.n, bounds)
    for r in range(nranks):
        assert int(res[r]["route"][0]) == (1 if route == "rccl" else 2)
        assert int(res[r]["route"][1]) == (1 if in_launch else 0)   # fpsq_info.comm_in_launch_sums
        assert int(res[r]["route"][2]) == 0                          # p2p_timeouts
        assert (int(res[r]["route"][3]) > 0) == fused                # last_fused_launches of the last call
        assert np.array_equal(res[r]["its"], its_ref), (r, res[r]["its"], its_ref)
        assert np.array_equal(res[r]["fs"], res[0]["fs"])  # phi and the return codes: replicated, bitwise
        assert np.all(res[r]["fs"][:, 1] == 0)
        for k in range(3):
            assert abs(res[r]["fs"][k, 0] - fs_ref[k]) <= 1e-9 * abs(fs_ref[k])
        if r + 1 < nranks:
            t = plan.overlaps(r)[1]
            for key in ("gx0", "gs2", "hv2", "p1", "p2"):
                assert t > 0 and np.array_equal(res[r][key][-t:], res[r + 1][key][:t]), key
    for key in ("gx0", "gx1", "gx2", "gs0", "gs1", "gs2", "hv2", "hv1", "p1", "p2"):
        assert _rel(plan.assemble([res[r][key] for r in range(nranks)]), want[key]) < 1e-9, key
    for key in ("ys0", "ys1", "ys2", "q1", "q2"):
        assert _rel(np.concatenate([res[r][key] for r in range(nranks)]), want[key]) < 1e-9, key


def test_a_missing_peer_ends_in_an_error_not_a_hang(tmp_path):
    """Rank 1 of 2 leaves after the set-up (FPSQ_TEST_P2P_DESERT): rank 0's exchange kernel polls a bounded number of times,
    raises the communicator's failure word and the call returns FPSQ_ERR_TIMEOUT -- within the test's time limit."""
    rcs, outs = _run_ranks(2, "p2p", 0.0, tmp_path, extra_env={"FPSQ_TEST_P2P_DESERT": "1", "FPSQ_P2P_POLLS": "300000"}, timeout=400)
    assert rcs[1] == 0 and rcs[0] != 0
    assert "bounded wait" in outs[0][1] or "did not arrive" in outs[0][1], outs[0][1][-1500:]


def test_a_rank_late_by_more_than_one_gpus_bound_fails_without_the_longer_waits(tmp_path):
    """The control of "-hiccup" above: with the followers' waits cut back to the bound of one GPU (FPSQ_DEBUG_WAIT_MORE=0: 2^15 looks,
    tens of milliseconds) the workgroups behind leaders that wait 150 ms for rank 1's push give up; the call ends in
    FPSQ_ERR_TIMEOUT -- on that rank from its own expired wait (which a rank of several does not answer with a repeat of its own),
    on the late rank, which then waits for a peer that has stopped, from the exchange's bound -- within the test's time limit."""
    rcs, outs = _run_ranks(2, "p2p", 0.0, tmp_path, timeout=400,
                           extra_env={"FPSQ_LX": "2", "FPSQ_FUSE_ITER": "2", "FPSQ_DEBUG_XCH_DELAY": "2", "FPSQ_DEBUG_XCH_LONG_DELAY_MS": "150",
                                      "FPSQ_DEBUG_WAIT_MORE": "0", "FPSQ_P2P_POLLS": "2000000"})
    assert rcs[0] != 0 and rcs[1] != 0, rcs
    assert "bounded wait" in outs[0][1], outs[0][1][-1500:]
    assert "bounded wait" in outs[1][1] or "did not arrive" in outs[1][1], outs[1][1][-1500:]


def test_three_ranks_soak_overlap_rows_stay_bitwise():
    """tools/lx_soak_mp.py, 1500 evaluations on three ranks (three processes) at n = 100000, sums over the ranks inside the launches,
    one launch per joint iteration with the halo exchange and finish inside: phi and the iteration counts bitwise the same on every
    rank, the overlap rows of grad(phi) bitwise the same on the two ranks sharing them, no expired wait.  This is the run that found
    round 5's stale-line bug (an A' workgroup prefetched the old values of overlap rows it does not finish; the line stayed in its
    CU's L1 and was served to the update workgroups that read the finished rows later in the same launch: one or two evaluations in
    a thousand came back with a few 128-byte lines of grad(phi) wrong on ONE of the two ranks, everything else in order)."""
    env = dict(os.environ, FPSQ_P2P_POLLS="3000000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lx_soak_mp.py"), "1500", "3", "0", "100000"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-1500:])


_NINE = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch  # noqa: F401
import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP, rccl_unique_id
from fps_amd.qdsolver import FpsqError
qp = problems.pde_control_like(n=24000, m=2400, per_row=24, window=512, seed=29)
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
g0 = np.empty(qp.n)
f0, rc0 = ref.objgrad(qp.x, gx=g0)
its0 = (ref.stats[0].niter, ref.stats[1].niter)
ref.close()
# rank 0 of NINE (the other eight are phantoms of the stand-in library: their records are zeros), holding every row
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, comm=("rccl", 9, 0, rccl_unique_id()), halo=(0, 0), comm_route="auto")
g = np.empty(qp.n)
f, rc = dev.objgrad(qp.x, gx=g)
i = dev.info()
assert i["comm_route"] == 1, i                      # FPSQ_ROUTE_RCCL: the peer tables hold 8 ranks
assert rc == rc0 and (dev.stats[0].niter, dev.stats[1].niter) == its0
assert abs(f - f0) <= 1e-9 * abs(f0) and np.max(np.abs(g - g0)) <= 1e-9 * np.max(np.abs(g0))
assert i["p2p_timeouts"] == 0 and i["wait_timeouts"] == 0
dev.close()
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, comm=("rccl", 9, 0, rccl_unique_id()), halo=(0, 0), comm_route="p2p")
try:
    dev.objgrad(qp.x, gx=g)
except FpsqError as e:
    assert "more than 8 ranks" in str(e), str(e)
    print("NINE-OK")
else:
    raise SystemExit("an explicit peer-to-peer request with 9 ranks must fail")
'''


def test_more_ranks_than_a_node_stay_on_the_collectives():
    """The peer tables of the peer-to-peer route hold the GPUs of ONE node (kMaxP2PRanks = 8).  A communicator of 9 ranks (two
    nodes, or 16 logical ranks) must stay on RCCL under FPSQ_ROUTE_AUTO -- decided before any table is indexed -- and fail
    an explicit FPSQ_ROUTE_P2P request with FPSQ_ERR_COMM (advisor, round 4: it indexed the tables out of bounds).  One process
    holds rank 0 of 9; the other ranks are phantoms of the stand-in library (FPSQ_SHIM_PHANTOM: zero records)."""
    env = dict(os.environ, FPSQ_RCCL_LIB=_shim(), FPSQ_SHIM_PHANTOM="1", FPSQ_SHIM_TIMEOUT="60")
    r = subprocess.run([sys.executable, "-c", _NINE, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NINE-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
