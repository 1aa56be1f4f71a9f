"""Developer soak (GPU box, one run): steps riding with leaders against stand-alone steps, BITWISE, over many evaluations.
usage: python tools/ride_soak.py [evaluations] [n] [m]
Two handles on the same problem (FPSQ_RIDE_LEAD=0 / 1), the same random points; every output and every statistic of
objgrad and hprod (Val(1) on every third call: the MINRES lane) must agree bit for bit.  Prints the number of calls compared."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
qp = problems.pde_control_like(n=n, m=m) if n >= 500_000 else problems.random_eqqp(n=n, m=m)
os.environ["FPSQ_RIDE_LEAD"] = "0"
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
os.environ["FPSQ_RIDE_LEAD"] = "1"
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
rng = np.random.default_rng(7)
t0 = time.time()
bad = 0
for k in range(N):
    scale = 0.5 ** (k % 7) * (1.0 if k % 3 else 1e-2)
    x = qp.xhat + scale * rng.standard_normal(qp.n)
    v = scale * rng.standard_normal(qp.n)
    outs = []
    for mdl in (ref, dev):
        gx, ys, gs, hv = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.n)
        f, rc = mdl.objgrad(x, gx=gx, ys=ys, gs=gs)
        st = [(mdl.stats[i].niter, mdl.stats[i].status, mdl.stats[i].rnorm) for i in range(2)]
        rch = mdl.hprod(v, hv, 1 if k % 3 == 0 else 2)
        sth = [(mdl.stats4[i].niter, mdl.stats4[i].status, mdl.stats4[i].rnorm) for i in range(4 if k % 3 == 0 else 2)]
        outs.append([np.array([f, rc, rch]), gx, ys, gs, hv, np.array(st, dtype=float).ravel(), np.array(sth, dtype=float).ravel()])
    if not all(np.array_equal(a, b) for a, b in zip(*outs)):
        bad += 1
        print("MISMATCH at evaluation", k, flush=True)
    if k % 500 == 499:
        print(f"{k + 1} evaluations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"compared {N} objgrad + {N} hprod calls (n={n}, m={m}): {bad} mismatches")
sys.exit(1 if bad else 0)
