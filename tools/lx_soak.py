"""Soak of the round-5 sharded path on ONE GPU: P in-process shards (one host thread each), peer-to-peer route with the sums over the
ranks formed INSIDE the launches (FPSQ_LX=2) and one launch per joint iteration with the halo exchange and finish inside
(FPSQ_FUSE_ITER=2), against a single-GPU handle on the same points.

    python tools/lx_soak.py [evaluations=2000] [shards=3] [delta=0]

Per evaluation: phi and the return code BITWISE equal on every shard, the overlap rows of grad(phi) bitwise equal on the two shards
sharing them, the iteration counts equal on every shard and within one of the single-GPU handle's, grad(phi) / ys to 1e-9 of it (1e-6 when one iteration apart); every 10th evaluation an hprod! Val(2)
the same way.  Counts mismatches and the handles' cumulative wait counters (fpsq_info); exit code 1 on any.
(The grids of this problem are resident all at once: shards of ONE device could otherwise starve each other's leaders -- which is
why a communicator only switches the in-launch sums on by itself when every rank has a device of its own.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 3
delta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
os.environ["FPSQ_LX"] = "2"
os.environ["FPSQ_FUSE_ITER"] = "2"
# in-process shards are streams of ONE process: with more streams than hardware queues (4 by default) two shards' launches can
# queue behind each other, and a launch that waits for a peer's sum then waits for a launch that cannot start (the bounded wait
# ends it with an error, as it should).  One process per GPU -- the product's layout -- has no such sharing.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402,F401
import fps_amd  # noqa: E402,F401
from fps_amd import problems  # noqa: E402
from fps_amd.device_qp import DeviceEqQP, LocalGroup  # noqa: E402
from fps_amd.distributed import halo_plan, row_partition, shard_qp_halo  # noqa: E402

qp = problems.pde_control_like(n=200_000, m=20_000, per_row=40, window=2048, seed=41)
bounds = row_partition(qp.rowptr, P)
plan = halo_plan(qp.rowptr, qp.colind, qp.n, bounds)
assert plan is not None
locs = [shard_qp_halo(qp, plan, r) for r in range(P)]
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
group = LocalGroup(P, p2p=True)
shards = [DeviceEqQP(locs[r], sigma=1e3, rho=1.0, delta=delta, comm=("local", group.ptr, r), halo=plan.overlaps(r)) for r in range(P)]
rng = np.random.default_rng(7)
bad = {"phi": 0, "overlap": 0, "iters": 0, "value": 0, "rc": 0}
fused = 0
off_by_one = 0  # another order of summation may stop one iteration apart at the tolerance: counted, not a mismatch
t0 = time.time()
g_ref, ys_ref = np.empty(qp.n), np.empty(qp.m)
for e in range(E):
    x = qp.xhat + (0.3 * 0.97 ** (e % 60)) * rng.standard_normal(qp.n)
    f_ref, rc_ref = ref.objgrad(x, gx=g_ref, ys=ys_ref)
    it_ref = (ref.stats[0].niter, ref.stats[1].niter)
    gx = [np.empty(l.n) for l in locs]
    ys = [np.empty(l.m) for l in locs]
    try:
        res = group.run([lambda r=r: shards[r].objgrad(np.ascontiguousarray(x[plan.window(r)]), gx=gx[r], ys=ys[r]) for r in range(P)])
    except Exception as exc:
        print(f"evaluation {e}: {exc}; counters {[(i['fuse_fallbacks'], i['wait_timeouts'], i['p2p_timeouts']) for i in (s.info() for s in shards)]}")
        sys.exit(1)
    if any(res[r] != res[0] for r in range(P)):
        bad["phi"] += 1
    if res[0][1] != rc_ref:
        bad["rc"] += 1
    its = [(s.stats[0].niter, s.stats[1].niter) for s in shards]
    if any(i != its[0] for i in its) or max(abs(its[0][0] - it_ref[0]), abs(its[0][1] - it_ref[1])) > 1:
        bad["iters"] += 1
    off_by_one += its[0] != it_ref
    for r in range(P - 1):
        t = plan.overlaps(r)[1]
        if t > 0 and not np.array_equal(gx[r][-t:], gx[r + 1][:t]):
            bad["overlap"] += 1
    ga = plan.assemble(gx)
    tol = 1e-9 if its[0] == it_ref else 1e-6  # one iteration apart: both within the solves' own tolerance of the exact value
    if (np.max(np.abs(ga - g_ref)) > tol * np.max(np.abs(g_ref)) or abs(res[0][0] - f_ref) > tol * abs(f_ref)
            or np.max(np.abs(np.concatenate(ys) - ys_ref)) > tol * np.max(np.abs(ys_ref))):
        bad["value"] += 1
    fused += shards[0].info()["last_fused_launches"]
    if e % 10 == 9:
        v = rng.standard_normal(qp.n)
        hv_ref = np.empty(qp.n)
        ref.hprod(v, hv_ref)
        hv = [np.empty(l.n) for l in locs]
        group.run([lambda r=r: shards[r].hprod(np.ascontiguousarray(v[plan.window(r)]), hv[r]) for r in range(P)])
        if np.max(np.abs(plan.assemble(hv) - hv_ref)) > 1e-9 * np.max(np.abs(hv_ref)):
            bad["value"] += 1
    if e % 500 == 499:
        print(f"{e + 1} evaluations, {time.time() - t0:.0f} s, mismatches {bad}", flush=True)
infos = [s.info() for s in shards]
waits = [(i["fuse_fallbacks"], i["wait_timeouts"], i["p2p_timeouts"]) for i in infos]
print(f"{E} evaluations (+ {E // 10} hprod!) on {P} shards, delta = {delta}: in-launch sums = {[i['comm_in_launch_sums'] for i in infos]}, "
      f"fused launches of shard 0: {fused}, evaluations one iteration apart from the single-GPU handle: {off_by_one}, mismatches {bad}, (fuse_fallbacks, wait_timeouts, p2p_timeouts) per shard {waits}, {time.time() - t0:.0f} s")
for s in shards:
    s.close()
group.close()
ref.close()
sys.exit(1 if any(bad.values()) or any(any(w) for w in waits) else 0)
