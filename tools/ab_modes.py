"""Developer A/B in ONE process: one model per library build (AB_LIBS=path[,path...], default the in-tree library)
and per environment setting (AB_ENV="NAME=v1,v2"), evaluation batches interleaved so that box-to-box and run-to-run
drift cancels (repeatability ~0.1 %, against ~5 % between separate bench.py runs).
usage: AB_LIBS=tools/ab/libfpsq_base.so,fletcherpenaltysolver.jl_amd/lib/libfpsq.so python tools/ab_modes.py [rounds] [batch]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import problems, _lib
from fps_amd.device_qp import DeviceEqQP

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 20
libs = [os.path.join(ROOT, p) for p in os.environ.get("AB_LIBS", "fletcherpenaltysolver.jl_amd/lib/libfpsq.so").split(",")]
envname, envvals = None, [None]
if os.environ.get("AB_ENV"):
    envname, vals = os.environ["AB_ENV"].split("=")
    envvals = vals.split(",")
optname, optvals = None, [None]
if os.environ.get("AB_OPT"):  # an fpsq_options field, e.g. AB_OPT="lookahead=2,4,8"
    optname, vals = os.environ["AB_OPT"].split("=")
    optvals = [int(v) for v in vals.split(",")]
workload = os.environ.get("AB_WORKLOAD", "headline")
if workload.startswith("pde:"):  # AB_WORKLOAD=pde:n:m -- the headline generator at another size
    qp = problems.pde_control_like(n=int(workload.split(":")[1]), m=int(workload.split(":")[2]))
else:
    qp = problems.pde_control_like(n=1_000_000, m=100_000) if workload == "headline" else problems.random_eqqp(n=100_000, m=10_000)
dev = torch.device("cuda", 0)
models = {}
import ctypes as _C
_ALL_SYMBOLS = list(_lib.SYMBOLS)
for lp in libs:
    _lib._LIB = None
    _lib.LIB_PATH = lp
    # an OLDER build may lack entry points added since: type what it has, stub the optional setters the Python layer calls
    probe = _C.CDLL(lp)
    _lib.SYMBOLS = [sy for sy in _ALL_SYMBOLS if hasattr(probe, sy[0])]
    lib_ = _lib.load()
    for name in ("fpsq_set_output_ordering",):
        if not any(sy[0] == name for sy in _lib.SYMBOLS):
            setattr(lib_, name, lambda *a: 0)
    for ev in envvals:
        if envname:
            os.environ[envname] = ev
        for ov in optvals:
            extra = {optname: ov} if optname else {}
            if os.environ.get("AB_ITMAX"):  # bound the runs of deliberately wrong experimental kernels
                extra.update(ls_itmax=int(os.environ["AB_ITMAX"]), ln_itmax=int(os.environ["AB_ITMAX"]))
            models[(os.path.basename(lp), f"{ev}/{optname}={ov}#{len(models)}")] = DeviceEqQP(
                qp, sigma=1e3, rho=1.0, delta=0.0, device=0, fuse_two_rhs=1, **extra)
dyn_name, dyn_vals = None, None
if os.environ.get("AB_DYN"):  # ONE handle, the variable re-read by the library at every call (FPSQ_AB_DYNAMIC): cancels the
    dyn_name, vals = os.environ["AB_DYN"].split("=")  # +-3 % spread between handles (buffer placement)
    dyn_vals = vals.split(",")
    os.environ["FPSQ_AB_DYNAMIC"] = "1"
    (k0, m0), = list(models.items())[:1]
    for mdl in list(models.values())[1:]:
        mdl.close()
    m0.close()
    m0 = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, device=0, fuse_two_rhs=1)
    models = {(k0[0], f"{dyn_name}={v}"): m0 for v in dyn_vals}
xs = torch.empty((batch, qp.n), dtype=torch.float64, device=dev)
for t in range(batch):
    xs[t].copy_(torch.from_numpy(qp.point(1 + t)))
gx = torch.empty(qp.n, dtype=torch.float64, device=dev)
res = {k: [] for k in models}
for r in range(rounds + 1):
    for k, mdl in models.items():
        if dyn_name:
            os.environ[dyn_name] = k[1].split("=")[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(batch):
            mdl.objgrad(xs[t], gx=gx)
        torch.cuda.synchronize()
        if r > 0:
            res[k].append(batch / (time.perf_counter() - t0))
for k in models:
    v = np.array(res[k])
    print(f"{k[0]:28s} {envname or ''}={k[1]}: median {np.median(v):7.1f}  min {v.min():7.1f}  max {v.max():7.1f} evals/s  "
          f"iters {models[k].stats[0].niter},{models[k].stats[1].niter}")
