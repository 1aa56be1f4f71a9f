"""Developer helper: the headline evaluation with a pinned iteration count (timing of deliberately wrong what-if builds)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP
qp = problems.pde_control_like(n=1_000_000, m=100_000)
it = int(os.environ.get("AB_ITMAX", "15"))
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, device=0, fuse_two_rhs=1, ls_itmax=it, ln_itmax=it)
d = torch.device("cuda", 0)
xs = [torch.from_numpy(qp.point(1 + t)).to(d) for t in range(8)]
gx = torch.empty(qp.n, dtype=torch.float64, device=d)
import time
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(24):
        dev.objgrad(xs[t % 8], gx=gx)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("evals/s %.1f iters %d %d" % (24 / (t1 - t0), dev.stats[0].niter, dev.stats[1].niter), flush=True)
