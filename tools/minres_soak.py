"""Developer soak (GPU box, one run): hprod! Val(1) -- the solve_two_extras lanes, a MINRES lane next to an LSQR lane -- with the MINRES
lane's merged launch (k_minres_mid: E1 -> step A -> E2 in one launch, FPSQ_MINRES_MERGE=1, the default) against the three launches
(FPSQ_MINRES_MERGE=0), BITWISE, over many products at changing vectors.   usage: python tools/minres_soak.py [products] [n] [m] [delta]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
delta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.4901161193847656e-08
qp = problems.pde_control_hashed(n=n, m=m)
os.environ["FPSQ_MINRES_MERGE"] = "0"
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
os.environ["FPSQ_MINRES_MERGE"] = "1"
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
rng = np.random.default_rng(11)
t0 = time.time()
bad = 0
counts = {}
for k in range(N):
    scale = 0.5 ** (k % 7) * (1.0 if k % 3 else 1e-2)
    v = scale * rng.standard_normal(qp.n)
    outs = []
    for mdl in (ref, dev):
        hv = np.empty(qp.n)
        rc = mdl.hprod(v, hv, 1)
        st = [(mdl.stats4[i].niter, mdl.stats4[i].status, mdl.stats4[i].rnorm) for i in range(4)]
        outs.append([np.array([rc]), hv, np.array(st, dtype=float).ravel()])
    key = tuple(int(c) for c in outs[0][2][0::3])
    counts[key] = counts.get(key, 0) + 1
    if not all(np.array_equal(a, b) for a, b in zip(*outs)):
        bad += 1
        print("MISMATCH at product", k, flush=True)
    if k % 250 == 249:
        print(f"{k + 1} products, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
i = dev.info()
print(f"compared {N} hprod! Val(1) products (n={n}, m={m}, delta={delta}): {bad} mismatches; iteration counts (4 lanes) seen: "
      f"{sorted(counts.items(), key=lambda kv: -kv[1])[:6]}; (fuse_fallbacks, wait_timeouts) of the merged handle: {(i['fuse_fallbacks'], i['wait_timeouts'])}")
sys.exit(1 if bad or i["fuse_fallbacks"] or i["wait_timeouts"] else 0)
