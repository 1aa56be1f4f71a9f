"""Developer aid: halo-sharded handles (in-process shards, in-launch sums), one launch per iteration against three: where do they differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import fps_amd  # noqa
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP, LocalGroup
from fps_amd.distributed import halo_plan, row_partition, shard_qp_halo

nshards = int(sys.argv[1]) if len(sys.argv) > 1 else 2
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
qp = problems.pde_control_like(n=24000, m=2400, per_row=24, window=512, seed=31)
bounds = row_partition(qp.rowptr, nshards)
plan = halo_plan(qp.rowptr, qp.colind, qp.n, bounds)
locs = [shard_qp_halo(qp, plan, r) for r in range(nshards)]
os.environ["FPSQ_LX"] = "2"
os.environ["FPSQ_AT_ROW_ALIGN"] = "8"
got = {}
for mode in ("0", "2"):
    os.environ["FPSQ_FUSE_ITER"] = mode
    group = LocalGroup(nshards, p2p=True)
    shards = [DeviceEqQP(locs[r], sigma=1e3, rho=1.0, delta=delta, comm=("local", group.ptr, r), halo=plan.overlaps(r)) for r in range(nshards)]
    rec = []
    for k in range(3):
        xk = qp.point(1 + k)
        gx = [np.empty(l.n) for l in locs]; ys = [np.empty(l.m) for l in locs]; gs = [np.empty(l.n) for l in locs]
        res = group.run([lambda r=r: shards[r].objgrad(np.ascontiguousarray(xk[plan.window(r)]), gx=gx[r], ys=ys[r], gs=gs[r]) for r in range(nshards)])
        rec.append((res, gx, ys, gs, [(s.stats[0].niter, s.stats[1].niter, s.stats[0].rnorm, s.stats[1].rnorm, s.stats[0].arnorm) for s in shards],
                    [s.info()["last_fused_launches"] for s in shards]))
    got[mode] = rec
    for s in shards: s.close()
    group.close()
for k in range(3):
    a, b = got["0"][k], got["2"][k]
    print("eval", k, "phi", a[0], b[0], "fused launches", b[5])
    print("   stats", a[4], b[4])
    for r in range(nshards):
        for nm, i in (("gx", 1), ("ys", 2), ("gs", 3)):
            d = np.abs(a[i][r] - b[i][r])
            print(f"   shard {r} {nm}: max diff {d.max():.3e} at {int(d.argmax())} of {d.size}; nonzero diffs {int((d > 0).sum())}")
