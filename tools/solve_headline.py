"""End-to-end demonstration: fps_solve on the headline equality QP (n = 1e6, m = 1e5, nnz = 1e7), entirely
device-resident (fps_amd.fps_solve.fps_solve_device): every obj/grad! is one fpsq_qp_objgrad, every Hessian product of
the Newton-CG sub-solver one fpsq_qp_hprod.  Prints the time to a first-order point and the KKT residuals, checked on
the host with an independent CSR."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
from fps_amd.fps_solve import fps_solve_device

scale = float(os.environ.get("AB_SCALE", "1"))
tol = float(os.environ.get("SOLVE_TOL", "1e-6"))
kry = float(os.environ.get("KRYLOV_TOL", "1e-11"))
qp = problems.pde_control_like(n=int(1_000_000 * scale), m=int(100_000 * scale))
opts = dict(ls_atol=kry, ls_rtol=kry, ls_axtol=kry, ls_btol=kry, ls_etol=kry, ln_atol=kry, ln_rtol=kry, ln_btol=kry,
            ln_conlim=0.0)
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **opts)
d = torch.device("cuda", 0)
x0 = torch.from_numpy(qp.x).to(d)
for sub in ("trunk", "lbfgs"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = fps_solve_device(dev, x0, subproblem_solver=sub, atol=tol, rtol=tol, verbose=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    x = stats.solution.cpu().numpy()
    lam = stats.multipliers.cpu().numpy()
    A = qp.scipy_csr()
    kkt = np.linalg.norm(qp.qdiag * x + qp.d + A.T @ lam, np.inf)
    feas = np.linalg.norm(A @ x - qp.b, np.inf)
    print(f"{sub}: status {stats.status} in {dt * 1e3:.1f} ms, {stats.iter} outer iterations, {stats.solver_specific}, "
          f"f = {stats.objective:.9e}, |grad L|_inf = {kkt:.2e}, |c|_inf = {feas:.2e}", flush=True)
