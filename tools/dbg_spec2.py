import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
tight = dict(ls_atol=1e-13, ls_rtol=1e-13, ls_axtol=1e-13, ls_btol=1e-13, ls_etol=1e-13, ln_atol=1e-13,
             ln_rtol=1e-13, ln_btol=1e-13, ln_conlim=0.0)
qp = problems.pde_control_like(n=3000, m=300)
os.environ["FPSQ_ADAPTIVE_RUNAHEAD"] = "0"
H = {"la1": DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, lookahead=1, **tight),
     "la8": DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, lookahead=8, **tight),
     "la4": DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **tight)}
os.environ["FPSQ_ADAPTIVE_RUNAHEAD"] = "1"
H["ad"] = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **tight)
rng = np.random.default_rng(0)
A = qp.scipy_csr()
for k in range(12):
    scale = 0.5 ** (k % 20) * (1.0 if k % 3 else 1e-3)
    x = qp.xhat + scale * rng.standard_normal(qp.n)
    g = qp.qdiag * x + qp.d; c = A @ x - qp.b
    res = {}
    for name, h in H.items():
        p1, q1, p2, q2 = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)
        h.solve_two_mixed(g, c, p1, q1, p2, q2)
        gx = np.empty(qp.n); fx, _ = h.objgrad(x, gx=gx)
        res[name] = (q1, q2, p2, gx, (h.stats[0].niter, h.stats[1].niter))
    base = res["la1"]
    line = f"k={k} its {base[4]}"
    for name in ("la8", "la4", "ad"):
        r = res[name]
        line += f" | {name}: dq1 {np.max(np.abs(r[0]-base[0])):.1e} dq2 {np.max(np.abs(r[1]-base[1])):.1e} dp2 {np.max(np.abs(r[2]-base[2])):.1e} dgx {np.max(np.abs(r[3]-base[3])):.1e}"
    print(line, flush=True)
