"""Developer soak (GPU box): TWO handles with one-launch iterations evaluating at the same time from two host threads (two streams),
each compared bitwise with its own single-threaded results.  Both grids are full of workgroups waiting for other workgroups of
their own launch and compete for the same XCDs; a circular wait between the two kernels would show as FPSQ_ERR_TIMEOUT, a
visibility problem as a mismatch.   usage: python tools/fuse_soak_two.py [rounds] [n1] [n2]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
n2 = int(sys.argv[3]) if len(sys.argv) > 3 else 500_000
os.environ["FPSQ_FUSE_ITER"] = "2"
qps = [problems.pde_control_like(n=n1, m=n1 // 10, seed=51), problems.pde_control_like(n=n2, m=n2 // 10, seed=52)]
devs = [DeviceEqQP(q, sigma=1e3, rho=1.0, delta=0.0) for q in qps]
rng = np.random.default_rng(21)
xs = [[q.xhat + 0.3 * 0.5 ** k * rng.standard_normal(q.n) for k in range(4)] for q in qps]


def evaluate(dev, q, x):
    gx, ys, gs = np.empty(q.n), np.empty(q.m), np.empty(q.n)
    f, rc = dev.objgrad(x, gx=gx, ys=ys, gs=gs)
    return [np.array([f, rc, dev.stats[0].niter, dev.stats[1].niter]), gx, ys, gs], dev.info()["last_fused_launches"]


want = [[evaluate(d, q, x)[0] for x in xx] for d, q, xx in zip(devs, qps, xs)]
want = [[evaluate(d, q, x)[0] for x in xx] for d, q, xx in zip(devs, qps, xs)]  # (second pass: the run-ahead has settled)
bad, fused, errs = [0, 0], [0, 0], []


def work(k):
    try:
        for rep in range(R):
            for x, w in zip(xs[k], want[k]):
                got, nf = evaluate(devs[k], qps[k], x)
                fused[k] += nf
                if not all(np.array_equal(a, b) for a, b in zip(got, w)):
                    bad[k] += 1
    except BaseException as e:  # noqa: BLE001
        errs.append(repr(e))


t0 = time.time()
ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(f"two handles (n = {n1}, {n2}) at once, {R} rounds of 4 points each: mismatches {bad}, fused launches {fused}, errors {errs[:2]}, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if (sum(bad) or errs) else 0)
