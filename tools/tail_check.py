"""Developer check (GPU box): the tail of an evaluation as ONE launch (k_spmv<.., GRAD>: the rows of the raw A'[q1, c] product go straight
into grad(phi); FPSQ_FUSE_TAIL=1, the default) against the two launches (raw product, then k_qp_penalty_grad; FPSQ_FUSE_TAIL=0):
every output of objgrad -- phi, gx, gs, ys, statistics -- and of an hprod! behind it BITWISE the same, with and without the proximal
term (eta, xk), delta = 0 and sqrt(eps), on the headline generators and smaller shapes.   usage: python tools/tail_check.py [evaluations=40]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
SE = 1.4901161193847656e-08
for name, qp, delta, eta in (("hashed 1e6", problems.pde_control_hashed(n=1_000_000, m=100_000), 0.0, 0.0),
                             ("hashed 1e6, eta", problems.pde_control_hashed(n=1_000_000, m=100_000), SE, 0.5),
                             ("stratified 2e5", problems.pde_control_like(n=200_000, m=20_000, per_row=40, window=2048, seed=5), SE, 0.0),
                             ("stratified 24000, eta", problems.pde_control_like(n=24_000, m=2_400, per_row=24, window=512, seed=31), 0.0, 2.0),
                             ("random 1e5", problems.random_eqqp(n=100_000, m=10_000), 0.0, 0.0)):
    outs, launches = [], []
    for mode in ("0", "1"):
        os.environ["FPSQ_FUSE_TAIL"] = mode
        dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta, eta=eta)
        rec = []
        rng = np.random.default_rng(3)
        for k in range(N):
            scale = 0.5 ** (k % 7) * (1.0 if k % 3 else 1e-2)
            x = qp.xhat + scale * rng.standard_normal(qp.n)
            xk = qp.xhat + 0.1 * rng.standard_normal(qp.n) if eta > 0 else None
            gx, ys, gs, hv = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.n)
            f, rc = dev.objgrad(x, gx=gx, ys=ys, gs=gs, xk=xk)
            st = [(dev.stats[i].niter, dev.stats[i].rnorm) for i in range(2)]
            nl = dev.info()["last_kernel_launches"]
            if k % 4 == 3:
                dev.hprod(rng.standard_normal(qp.n), hv, 2)
            rec += [np.array([f, rc]), gx, ys, gs, hv if k % 4 == 3 else np.zeros(1), np.array(st).ravel()]
        launches.append(nl)
        outs.append(rec)
        dev.close()
    same = all(np.array_equal(a, b) for a, b in zip(*outs))
    print(f"{name} (delta {delta}, eta {eta}): {N} evaluations, launches of the last evaluation {launches[0]} / {launches[1]}:",
          "BITWISE the same" if same else "DIFFERENT", flush=True)
    bad += not same
sys.exit(1 if bad else 0)
