"""Developer check: other matrix shapes at the headline size through objgrad, tuned layouts against plain CSR
(jac_format = 1): random columns (no locality), denser rows."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

d = torch.device("cuda", 0)
cases = {"pde n=1e6 m=1e5 (headline)": lambda: problems.pde_control_like(n=1_000_000, m=100_000),
         "random n=1e6 m=1e5 (100 random columns per row)": lambda: problems.random_eqqp(n=1_000_000, m=100_000),
         "pde n=2e5 m=1e5 (rows overlap heavily)": lambda: problems.pde_control_like(n=200_000, m=100_000)}
for name, gen in cases.items():
    qp = gen()
    for fmt in (0, 1):
        dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, jac_format=fmt)
        xs = [torch.from_numpy(qp.point(1 + k)).to(d) for k in range(8)]
        gx = torch.empty(qp.n, dtype=torch.float64, device=d)
        for k in range(2):
            dev.objgrad(xs[k], gx=gx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 8):
            _, rc = dev.objgrad(xs[k], gx=gx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        print(f"{name:52s} jac_format={fmt}: {1 / dt:7.1f} evals/s ({dt * 1e3:6.2f} ms), iterations "
              f"{dev.stats[0].niter},{dev.stats[1].niter}, rc {rc}, nnz {qp.nnz}", flush=True)
        dev.close()
