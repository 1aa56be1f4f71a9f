"""Device-resident penalty evaluation on the synthetic eq-QP model: the benchmark's unit of work.

One `objgrad(x)` = one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a fresh x
(src/model-Fletcherpenaltynlp.jl:403-437) executed entirely on the MI355X through `fpsq_qp_objgrad`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .qdsolver import FpsqError


class DeviceEqQP:
    def __init__(self, qp, sigma=1e3, rho=1.0, delta=0.0, eta=0.0, device=0, **opt_overrides):
        self._lib = _lib.load()
        self.qp, self.sigma, self.rho, self.delta, self.eta = qp, sigma, rho, delta, eta
        opts = _lib.Options()
        self._lib.fpsq_default_options(qp.n, qp.m, C.byref(opts))
        opts.device = device
        for k, v in opt_overrides.items():
            setattr(opts, k, v)
        self.opts = opts
        h = C.c_void_p()
        if self._lib.fpsq_create(C.byref(h), qp.n, qp.m, C.byref(opts)) != 0:
            raise FpsqError(self._lib.fpsq_last_error(None).decode())
        self._h = h
        rp = np.ascontiguousarray(qp.rowptr, dtype=np.int32)
        ci = np.ascontiguousarray(qp.colind, dtype=np.int32)
        self._check(self._lib.fpsq_set_jacobian_structure_csr(h, rp.ctypes.data, ci.ctypes.data))
        self._check(self._lib.fpsq_set_jacobian_values(h, np.ascontiguousarray(qp.vals).ctypes.data))
        self._check(self._lib.fpsq_set_delta(h, float(delta)))
        q = C.c_void_p()
        self._check(self._lib.fpsq_qp_create(h, qp.qdiag.ctypes.data, qp.d.ctypes.data, qp.b.ctypes.data, C.byref(q)))
        self._q = q
        self.stats = (_lib.Stats * 2)()

    def _check(self, rc):
        if rc < 0:
            raise FpsqError(self._lib.fpsq_last_error(self._h).decode())
        return rc

    def set_delta(self, delta):
        self.delta = delta
        self._check(self._lib.fpsq_set_delta(self._h, float(delta)))

    def set_profiling(self, on):
        self._check(self._lib.fpsq_set_profiling(self._h, int(on)))

    def objgrad(self, x, gx=None, ys=None, gs=None, xk=None):
        """x / gx / ys / gs / xk: numpy arrays, torch tensors (host or device) or raw addresses.
        Returns (fx, rc): rc > 0 flags a Krylov solve that stopped unsolved (the reference warns)."""
        fx = C.c_double()
        rc = self._check(self._lib.fpsq_qp_objgrad(self._h, self._q, _lib.ptr(x), self.sigma, self.rho, self.eta,
                                                   _lib.ptr(xk), C.byref(fx), _lib.ptr(gx), _lib.ptr(ys),
                                                   _lib.ptr(gs), self.stats))
        return fx.value, rc

    def info(self):
        i = _lib.Info()
        self._check(self._lib.fpsq_get_info(self._h, C.byref(i)))
        return i.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fpsq_qp_destroy(self._q)
            self._lib.fpsq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
