"""Developer check: a problem 4x (or AB_SCALE x) the headline size through the device-resident objgrad -- sizes, the
32-bit index limits of the padded layouts and the KKT residuals of what comes back."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

scale = int(os.environ.get("AB_SCALE", "4"))
t0 = time.perf_counter()
qp = problems.pde_control_like(n=1_000_000 * scale, m=100_000 * scale)
print(f"generated n={qp.n} m={qp.m} nnz={qp.nnz} in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
print(f"set-up {time.perf_counter() - t0:.1f} s; info {dev.info()}", flush=True)
d = torch.device("cuda", 0)
xs = [torch.from_numpy(qp.point(1 + k)).to(d) for k in range(6)]
gx = torch.empty(qp.n, dtype=torch.float64, device=d)
ys = torch.empty(qp.m, dtype=torch.float64, device=d)
gs = torch.empty(qp.n, dtype=torch.float64, device=d)
for k in range(2):
    dev.objgrad(xs[k], gx=gx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(2, 6):
    fx, rc = dev.objgrad(xs[k], gx=gx, ys=ys, gs=gs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(f"{1 / dt:.1f} evals/s ({dt * 1e3:.2f} ms), iterations {dev.stats[0].niter},{dev.stats[1].niter}, rc {rc}", flush=True)
A = qp.scipy_csr()
x = xs[5].cpu().numpy()
g = qp.qdiag * x + qp.d
c = A @ x - qp.b
gsn, ysn = gs.cpu().numpy(), ys.cpu().numpy()
print("residual gs + A'ys - g:", np.linalg.norm(gsn + A.T @ ysn - g) / np.linalg.norm(g),
      " A gs - sigma c:", np.linalg.norm(A @ gsn - 1e3 * c) / np.linalg.norm(1e3 * c))
