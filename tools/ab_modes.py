"""Developer A/B: several handles in ONE process (one per FPSQ_RIDE_STEPS setting or library build is not possible
here -- one library per process), evaluation batches interleaved so box-to-box and run-to-run drift cancels.
usage: python tools/ab_modes.py [rounds] [batch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 20
modes = [int(x) for x in os.environ.get("AB_MODES", "0,1,2").split(",")]
qp = problems.pde_control_like(n=1_000_000, m=100_000)
dev = torch.device("cuda", 0)
models = {}
for md in modes:
    os.environ["FPSQ_RIDE_STEPS"] = str(md)
    models[md] = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, device=0, fuse_two_rhs=1)
xs = torch.empty((batch, qp.n), dtype=torch.float64, device=dev)
for t in range(batch):
    xs[t].copy_(torch.from_numpy(qp.point(1 + t)))
gx = torch.empty(qp.n, dtype=torch.float64, device=dev)
res = {md: [] for md in modes}
for r in range(rounds + 1):
    for md in modes:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(batch):
            models[md].objgrad(xs[t], gx=gx)
        torch.cuda.synchronize()
        if r > 0:
            res[md].append(batch / (time.perf_counter() - t0))
for md in modes:
    v = np.array(res[md])
    print(f"ride={md}: median {np.median(v):7.1f}  min {v.min():7.1f}  max {v.max():7.1f} evals/s  fallbacks {models[md].info().get('ride_fallbacks')}")
