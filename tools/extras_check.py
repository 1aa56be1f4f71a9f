"""Developer check: solve_two_extras (LSQR + MINRES on A A' + tau I; hprod! Val(1)) at the headline size."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems, _lib
from fps_amd.device_qp import DeviceEqQP

qp = problems.pde_control_like(n=1_000_000, m=100_000)
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
d = torch.device("cuda", 0)
r1 = [torch.from_numpy(qp.point(1 + k)).to(d) for k in range(6)]
r2 = [torch.from_numpy(qp.b * (1.0 + 0.1 * k)).to(d) for k in range(6)]
o1 = torch.empty(qp.m, dtype=torch.float64, device=d)
o2 = torch.empty(qp.m, dtype=torch.float64, device=d)
lib = dev._lib
for k in range(6):
    if k == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    rc = lib.fpsq_solve_two_extras(dev._h, _lib.ptr(r1[k]), _lib.ptr(r2[k]), _lib.ptr(o1), _lib.ptr(o2), dev.stats)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(f"solve_two_extras: {dt * 1e3:.2f} ms per call, LSQR {dev.stats[0].niter} + MINRES {dev.stats[1].niter} iterations, rc {rc}, "
      f"launches {dev.info()['last_kernel_launches']}")
