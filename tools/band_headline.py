"""Developer/report tool: the sparse direct (block-banded) back-end at the headline size next to the iterative one.
   python tools/band_headline.py > profiles/r02_band_headline.txt"""
import ctypes as C, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd
from fps_amd import _lib, problems
from fps_amd.device_qp import DeviceEqQP

lib = _lib.load()
qp = problems.pde_control_like(n=1_000_000, m=100_000)
A = qp.scipy_csr()
b = C.c_void_p()
rp, ci = qp.rowptr.astype(np.int32), qp.colind.astype(np.int32)
t0 = time.perf_counter()
assert lib.fpsq_band_create(C.byref(b), qp.n, qp.m, rp.ctypes.data, ci.ctypes.data, 0) == 0, lib.fpsq_band_last_error(None)
t_create = time.perf_counter() - t0
x = qp.point(3)
g = qp.qdiag * x + qp.d
c = A @ x - qp.b
info = C.c_int32()
bi = _lib.BandInfo()
for delta in (0.0, float(np.sqrt(np.finfo(float).eps))):
    for rep in range(2):
        t0 = time.perf_counter()
        rc = lib.fpsq_band_factorize(b, qp.vals.ctypes.data, delta, C.byref(info))
        t_fact = time.perf_counter() - t0
    assert rc == 0, (rc, info.value)
    p1, q1, p2, q2 = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)
    t0 = time.perf_counter()
    assert lib.fpsq_band_solve_two_mixed(b, g.ctypes.data, c.ctypes.data, p1.ctypes.data, q1.ctypes.data, p2.ctypes.data, q2.ctypes.data) == 0
    t_solve = time.perf_counter() - t0
    lib.fpsq_band_get_info(b, C.byref(bi))
    d = bi.as_dict()
    r = [np.linalg.norm(p1 + A.T @ q1 - g) / np.linalg.norm(g), np.linalg.norm(A @ p1 - delta * q1) / np.linalg.norm(g),
         np.linalg.norm(p2 + A.T @ q2) / np.linalg.norm(c), np.linalg.norm(A @ p2 - delta * q2 - c) / np.linalg.norm(c)]
    print(f"delta={delta:.3g}: blocks {d['nblocks']}, half bandwidth {d['bandwidth_blocks']} blocks, factor {d['factor_bytes'] / 1e9:.2f} GB; "
          f"device ms: form M {d['last_form_ms']:.2f}, block Cholesky {d['last_chol_ms']:.2f}, two-RHS solve {d['last_solve_ms']:.2f}; "
          f"host wall: factorize {t_fact * 1e3:.1f} ms (incl. 80 MB of values over PCIe), solve {t_solve * 1e3:.1f} ms; "
          f"KKT residuals {max(r):.2e}")
    dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
    a1, b1, a2, b2 = np.empty(qp.n), np.empty(qp.m), np.empty(qp.n), np.empty(qp.m)
    dev.solve_two_mixed(g, c, a1, b1, a2, b2)
    print(f"   iterative (sqrt(eps) tolerances): iterations {dev.stats[0].niter}/{dev.stats[1].niter}, device {dev.info()['last_solve_ms']:.2f} ms; "
          f"|q1 - q1_direct| / |q1| = {np.linalg.norm(b1 - q1) / np.linalg.norm(q1):.2e}, |q2 - q2_direct| / |q2| = {np.linalg.norm(b2 - q2) / np.linalg.norm(q2):.2e}")
    dev.close()
print(f"symbolic phase (host, band structure + transposed pattern): {t_create * 1e3:.0f} ms")
lib.fpsq_band_destroy(b)
