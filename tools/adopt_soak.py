"""Developer soak (GPU box): a handle that enqueues on the caller's registered stream (device-resident torch tensors, stream-ordered
outputs; include/fpsq.h INPUT READINESS) against a handle fed host arrays (its own stream, synchronous), BITWISE, over many
evaluations -- on torch's default stream and on a side stream, x produced by a kernel of that stream right in front of every call,
gx consumed on it right behind.   usage: python tools/adopt_soak.py [evaluations=1000] [n=1000000] [m=100000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fps_amd  # noqa: F401
from fps_amd import problems
from fps_amd.device_qp import DeviceEqQP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
qp = problems.pde_control_hashed(n=n, m=m)
host = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0)
d0 = torch.device("cuda:0")
side = torch.cuda.Stream(device=d0)
xhat = torch.from_numpy(qp.xhat).to(d0)
rng = np.random.default_rng(5)
bad = 0
t0 = time.time()
gx_d = torch.empty(qp.n, dtype=torch.float64, device=d0)
ys_d = torch.empty(qp.m, dtype=torch.float64, device=d0)
for k in range(N):
    scale = 0.5 ** (k % 7) * (1.0 if k % 3 else 1e-2)
    z = rng.standard_normal(qp.n)
    x_h = qp.xhat + scale * z
    gx_h, ys_h = np.empty(qp.n), np.empty(qp.m)
    f_h, rc_h = host.objgrad(x_h, gx=gx_h, ys=ys_h)
    st_h = [(host.stats[i].niter, host.stats[i].rnorm) for i in range(2)]
    stream = side if (k // 50) % 2 else torch.cuda.current_stream()
    up = torch.from_numpy(x_h).to(d0)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        x_d = up * 2.0 - up                          # (exact: the same bits as x_h, produced by kernels of `stream` in front of the call)
        f_d, rc_d = dev.objgrad(x_d, gx=gx_d, ys=ys_d)
        chk = gx_d.clone()                           # consumed on the same stream, no host synchronisation in between
        stream.synchronize()
    st_d = [(dev.stats[i].niter, dev.stats[i].rnorm) for i in range(2)]
    same = (f_d == f_h and rc_d == rc_h and st_d == st_h and np.array_equal(chk.cpu().numpy(), gx_h) and np.array_equal(ys_d.cpu().numpy(), ys_h)
            and np.array_equal(x_d.cpu().numpy(), x_h))
    if not same:
        bad += 1
        print("MISMATCH at evaluation", k, f_d, f_h, st_d, st_h, flush=True)
    if k % 250 == 249:
        print(f"{k + 1} evaluations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
i = dev.info()
print(f"compared {N} evaluations (n={n}, m={m}) of a handle on the caller's stream (default / side, switching every 50) with a host-fed handle: "
      f"{bad} mismatches; (fuse_fallbacks, wait_timeouts) {(i['fuse_fallbacks'], i['wait_timeouts'])}")
sys.exit(1 if bad or i["fuse_fallbacks"] or i["wait_timeouts"] else 0)
