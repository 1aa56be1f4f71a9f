"""Developer aid: runs cut at k iterations, several iterations per launch against one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import fps_amd  # noqa
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

kmax = sys.argv[1] if len(sys.argv) > 1 else "2"
qp = problems.pde_control_like(n=60000, m=6000, per_row=20, window=1024, seed=19)
os.environ["FPSQ_FUSE_ITER"] = "2"
A = qp.scipy_csr()
for cut in (4, 7, 10, 13):
    got = {}
    for mode in ("1", kmax):
        os.environ["FPSQ_MULTI_ITER"] = mode
        dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, ls_itmax=cut, ln_itmax=cut)
        rec = []
        for k in range(2):
            x = qp.point(1 + k)
            o = [np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)]
            rc = dev.solve_two_mixed(qp.qdiag * x + qp.d, A @ x - qp.b, *o)
            i = dev.info()
            rec.append((rc, o, (dev.stats[0].niter, dev.stats[1].niter, dev.stats[0].rnorm, dev.stats[1].rnorm, dev.stats[0].arnorm),
                        (i["last_multi_launches"], i["last_multi_iterations"], i["last_fused_launches"])))
        got[mode] = rec
        dev.close()
    a, b = got["1"][1], got[kmax][1]
    print("cut", cut, "stats", a[2], b[2], "multi", b[3])
    for nm, i in (("p1", 0), ("q1", 1), ("p2", 2), ("q2", 3)):
        d = np.abs(a[1][i] - b[1][i])
        nz = np.nonzero(d)[0]
        print(f"   {nm}: max diff {d.max():.3e} (max |ref| {np.abs(a[1][i]).max():.3e}); nonzero diffs {nz.size} of {d.size}; first {nz[:5]}")
