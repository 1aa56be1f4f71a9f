import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import fps_amd  # noqa
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
kmax = sys.argv[1]
qp = problems.pde_control_like(n=60000, m=6000, per_row=20, window=1024, seed=19)
os.environ["FPSQ_FUSE_ITER"] = "2"
A = qp.scipy_csr()
cut = 4
for trial in range(3):
    got = {}
    for mode in ("1", kmax):
        os.environ["FPSQ_MULTI_ITER"] = mode
        dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, ls_itmax=cut, ln_itmax=cut)
        for k in range(2):
            x = qp.point(1 + k)
            o = [np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)]
            dev.solve_two_mixed(qp.qdiag * x + qp.d, A @ x - qp.b, *o)
        got[mode] = o
        dev.close()
    a, b = got["1"][2], got[kmax][2]
    nz = np.nonzero(a != b)[0]
    runs = []
    if nz.size:
        s = p = nz[0]
        for i in nz[1:]:
            if i != p + 1:
                runs.append((int(s), int(p)))
                s = i
            p = i
        runs.append((int(s), int(p)))
    print("trial", trial, "diffs", nz.size, "runs", runs[:12])
    for (s, e) in runs[:3]:
        print("   ref", a[s:s+4], "got", b[s:s+4], "ratio", (b[s:s+4] / a[s:s+4]))
    print("   nonzeros of ref:", int((a != 0).sum()))
