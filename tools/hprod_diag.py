"""Developer diagnostic: repeated solve_two_least_squares / hprod on fresh handles; reports which outputs differ bitwise."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
qp = problems.pde_control_like(n=3000, m=300, per_row=12, window=256, seed=31)
rng = np.random.default_rng(5)
v = rng.standard_normal(qp.n)
w = qp.qdiag * v
mode = sys.argv[2] if len(sys.argv) > 2 else "lsq"
ref = None
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
    for call in range(3):
        if mode == "lsq":
            o = [np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)]
            dev.solve_two_least_squares(v, w, *o)
        else:
            o = [np.empty(qp.n)]
            dev.hprod(v, o[0], 2)
        st = [(dev.stats[k].niter, dev.stats[k].status) for k in range(2)]
        if ref is None:
            ref = [a.copy() for a in o]
        d = [float(np.max(np.abs(a - b))) for a, b in zip(o, ref)]
        if any(x != 0.0 for x in d):
            bad += 1
            print(rep, call, "DIFF vs first result:", d, st)
    dev.close()
print("done, differing calls:", bad)
