import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import fps_amd  # noqa
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
mode = sys.argv[1]
qp = problems.pde_control_like(n=60000, m=6000, per_row=20, window=1024, seed=19)
os.environ["FPSQ_FUSE_ITER"] = "2"
os.environ["FPSQ_MULTI_ITER"] = mode
A = qp.scipy_csr()
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, ls_itmax=3, ln_itmax=3)
for k in range(2):
    x = qp.point(1 + k)
    o = [np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)]
    print("=== call", k, flush=True)
    rc = dev.solve_two_mixed(qp.qdiag * x + qp.d, A @ x - qp.b, *o)
dev.close()
