"""Developer aid: several iterations per launch against one -- where do the outputs differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import fps_amd  # noqa
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

kmax = sys.argv[1] if len(sys.argv) > 1 else "2"
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (60000, 6000)
qp = problems.pde_control_like(n=n, m=m, per_row=20 if n < 200000 else 100, window=1024 if n < 200000 else 8192, seed=19)
os.environ["FPSQ_FUSE_ITER"] = "2"
got = {}
for mode in ("1", kmax):
    os.environ["FPSQ_MULTI_ITER"] = mode
    dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
    rec = []
    for k in range(3):
        x = qp.point(1 + k)
        gx, ys, gs = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n)
        f, rc = dev.objgrad(x, gx=gx, ys=ys, gs=gs)
        i = dev.info()
        rec.append((f, rc, gx, ys, gs, (dev.stats[0].niter, dev.stats[1].niter, dev.stats[0].rnorm, dev.stats[1].rnorm),
                    (i["last_multi_launches"], i["last_multi_iterations"], i["last_fused_launches"], i["wait_timeouts"])))
    got[mode] = rec
    dev.close()
for k in range(3):
    a, b = got["1"][k], got[kmax][k]
    print("eval", k, "phi", a[0], b[0], "rc", a[1], b[1], "stats", a[5], b[5], "multi", b[6])
    for nm, i in (("gx", 2), ("ys", 3), ("gs", 4)):
        d = np.abs(a[i] - b[i])
        nz = np.nonzero(d)[0]
        print(f"   {nm}: max diff {d.max():.3e}; nonzero diffs {nz.size} of {d.size}; first {nz[:6]} last {nz[-6:]}")
