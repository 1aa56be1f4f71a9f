"""Developer check: objgrad with the adaptive run-ahead / speculative epilogue against a handle without it, over a
sequence of points whose Krylov iteration counts vary (bitwise equality expected)."""
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
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **tight)
os.environ["FPSQ_ADAPTIVE_RUNAHEAD"] = "1"
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **tight)
rng = np.random.default_rng(0)
bad = 0
for k in range(60):
    scale = 0.5 ** (k % 20) * (1.0 if k % 3 else 1e-3)
    x = qp.xhat + scale * rng.standard_normal(qp.n)
    for mode in ("host", "dev"):
        g1, g2, y1, y2 = np.empty(qp.n), np.empty(qp.n), np.empty(qp.m), np.empty(qp.m)
        if mode == "dev":
            xt = torch.from_numpy(x).cuda(); t1 = torch.empty(qp.n, dtype=torch.float64, device="cuda"); t2 = torch.empty_like(t1)
            f1, rc1 = ref.objgrad(xt, gx=t1, ys=y1); f2, rc2 = dev.objgrad(xt, gx=t2, ys=y2)
            g1, g2 = t1.cpu().numpy(), t2.cpu().numpy()
        else:
            f1, rc1 = ref.objgrad(x, gx=g1, ys=y1); f2, rc2 = dev.objgrad(x, gx=g2, ys=y2)
        i1 = (ref.stats[0].niter, ref.stats[1].niter); i2 = (dev.stats[0].niter, dev.stats[1].niter)
        ok = f1 == f2 and np.array_equal(g1, g2) and np.array_equal(y1, y2) and i1 == i2 and rc1 == rc2
        if not ok:
            bad += 1
            print(f"k={k} {mode} MISMATCH its {i1} vs {i2} rc {rc1},{rc2} f {f1} {f2} dg {np.max(np.abs(g1-g2)):.3e} dy {np.max(np.abs(y1-y2)):.3e}", flush=True)
        elif k < 8:
            print(f"k={k} {mode} ok its {i1}", flush=True)
print("mismatches:", bad)
