"""Developer soak (GPU box, one run): one-launch iterations (k_iter_fused) against two-launch iterations, BITWISE, over many
evaluations.   usage: python tools/fuse_soak.py [evaluations] [n] [m] [delta] [stratified|hashed] [rotate]
Two handles on the same problem and the same block partition (FPSQ_FUSE_ITER=0 + FPSQ_AT_ROW_ALIGN=8 / FPSQ_FUSE_ITER=2), the
same random points at changing distances from the solution (so iteration counts move and the run-ahead mispredicts);
every output and statistic of objgrad and of hprod Val(2) must agree bit for bit.  The fused launch hands rows between
workgroups on different XCDs without fences (written-through stores, flags, agent-scope gathers): a stale line anywhere
would show here as a mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
delta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
gen = sys.argv[5] if len(sys.argv) > 5 else "stratified"  # "hashed": SURVEY 8(d)'s literal generator
qp = problems.pde_control_hashed(n=n, m=m) if gen == "hashed" else problems.pde_control_like(n=n, m=m)
os.environ["FPSQ_AT_ROW_ALIGN"] = "8"
os.environ["FPSQ_FUSE_ITER"] = "0"
ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
os.environ["FPSQ_FUSE_ITER"] = "2"
if len(sys.argv) > 6:  # every hand-over across XCDs (see test_one_launch_iterations_with_every_hand_over_across_xcds)
    os.environ["FPSQ_DEBUG_FUSE_ROTATE"] = sys.argv[6]
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
rng = np.random.default_rng(7)
t0 = time.time()
bad = fused = 0
counts = {}
for k in range(N):
    scale = 0.5 ** (k % 7) * (1.0 if k % 3 else 1e-2)
    x = qp.xhat + scale * rng.standard_normal(qp.n)
    v = scale * rng.standard_normal(qp.n)
    outs = []
    for mdl in (ref, dev):
        gx, ys, gs, hv = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.n)
        f, rc = mdl.objgrad(x, gx=gx, ys=ys, gs=gs)
        if mdl is dev:
            fused += mdl.info()["last_fused_launches"]
        st = [(mdl.stats[i].niter, mdl.stats[i].status, mdl.stats[i].rnorm) for i in range(2)]
        rch = mdl.hprod(v, hv, 2)
        if mdl is dev:
            fused += mdl.info()["last_fused_launches"]
        sth = [(mdl.stats4[i].niter, mdl.stats4[i].status, mdl.stats4[i].rnorm) for i in range(2)]
        outs.append([np.array([f, rc, rch]), gx, ys, gs, hv, np.array(st, dtype=float).ravel(), np.array(sth, dtype=float).ravel()])
    key = tuple(int(c) for c in outs[0][5][0::3]) + tuple(int(c) for c in outs[0][6][0::3])
    counts[key] = counts.get(key, 0) + 1
    if not all(np.array_equal(a, b) for a, b in zip(*outs)):
        bad += 1
        print("MISMATCH at evaluation", k, flush=True)
    if k % 250 == 249:
        print(f"{k + 1} evaluations, {bad} mismatches, {fused} fused launches, {time.time() - t0:.0f} s", flush=True)
print(f"compared {N} objgrad + {N} hprod calls (n={n}, m={m}, delta={delta}, {gen} offsets, rotate {os.environ.get('FPSQ_DEBUG_FUSE_ROTATE', '0')}): {bad} mismatches; {fused} fused launches; "
      f"iteration counts (lsqr, craig | hprod lsqr, lsqr) seen: {sorted(counts.items(), key=lambda kv: -kv[1])[:8]}")
sys.exit(1 if bad else 0)
