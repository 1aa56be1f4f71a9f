"""Minimal NLPModels.jl-shaped user models (L0 of SURVEY.md §1) used by the tests and the benchmark.

Method names follow NLPModels.jl (obj, grad, cons, jac_structure, jac_coord, jtprod, hprod); indices returned by
jac_structure are 1-based like Julia's.  These are HOST models: the reference's user model is host code too.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


class _Model:
    def __init__(self, nvar, ncon, nnzj, x0, lcon=None, name="model"):
        self.meta = SimpleNamespace(nvar=nvar, ncon=ncon, nnzj=nnzj, x0=np.asarray(x0, float),
                                    lcon=np.zeros(ncon) if lcon is None else np.asarray(lcon, float), name=name)

    def jtprod(self, x, v):
        rows, cols = self.jac_structure()
        out = np.zeros(self.meta.nvar)
        np.add.at(out, cols - 1, self.jac_coord(x) * np.asarray(v)[rows - 1])
        return out

    def jprod(self, x, v):
        rows, cols = self.jac_structure()
        out = np.zeros(self.meta.ncon)
        np.add.at(out, rows - 1, self.jac_coord(x) * np.asarray(v)[cols - 1])
        return out

    def ghjvprod(self, x, g, v):
        """(g' Hess c_i v)_i; zero unless a model overrides it (linear constraints)."""
        return np.zeros(self.meta.ncon)


class SumSquares(_Model):
    """f = x'x, c = sum(x) - 1   (test/unit-test.jl:18, test/test-2.jl:30)"""

    def __init__(self, n=10):
        super().__init__(n, 1, n, np.zeros(n), name="sumsq")

    def obj(self, x): return float(x @ x)
    def grad(self, x): return 2.0 * x
    def cons(self, x): return np.array([x.sum() - 1.0])
    def jac_structure(self): return np.ones(self.meta.nvar, np.int64), np.arange(1, self.meta.nvar + 1)
    def jac_coord(self, x): return np.ones(self.meta.nvar)
    def hprod(self, x, y, v, obj_weight=1.0): return obj_weight * 2.0 * np.asarray(v)


class RosenbrockCircle(_Model):
    """f = (x1-1)^2 + 100 (x2 - x1^2)^2, c = x1^2 + x2^2 - 1   (test/unit-test.jl:79-85)"""

    def __init__(self):
        super().__init__(2, 1, 2, np.zeros(2), name="rosenbrock-circle")

    def obj(self, x): return float((x[0] - 1) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2)
    def grad(self, x): return np.array([2 * (x[0] - 1) - 400 * x[0] * (x[1] - x[0] ** 2), 200 * (x[1] - x[0] ** 2)])
    def cons(self, x): return np.array([x[0] ** 2 + x[1] ** 2 - 1.0])
    def jac_structure(self): return np.array([1, 1]), np.array([1, 2])
    def jac_coord(self, x): return np.array([2 * x[0], 2 * x[1]])

    def hprod(self, x, y, v, obj_weight=1.0):
        H = np.array([[2 - 400 * (x[1] - x[0] ** 2) + 800 * x[0] ** 2, -400 * x[0]], [-400 * x[0], 200.0]])
        return obj_weight * (H @ v) + y[0] * 2.0 * np.asarray(v)

    def ghjvprod(self, x, g, v):
        return np.array([2.0 * float(np.dot(g, v))])


class HS6(_Model):
    """f = (1 - x1)^2, c = 10 (x2 - x1^2), x0 = (-1.2, 1)   (test/test-2.jl:54-55; BASELINE configs[0])"""

    def __init__(self):
        super().__init__(2, 1, 2, np.array([-1.2, 1.0]), name="HS6")

    def obj(self, x): return float((1 - x[0]) ** 2)
    def grad(self, x): return np.array([-2 * (1 - x[0]), 0.0])
    def cons(self, x): return np.array([10 * (x[1] - x[0] ** 2)])
    def jac_structure(self): return np.array([1, 1]), np.array([1, 2])
    def jac_coord(self, x): return np.array([-20 * x[0], 10.0])

    def hprod(self, x, y, v, obj_weight=1.0):
        return np.array([obj_weight * 2 * v[0] + y[0] * (-20.0) * v[0], 0.0])

    def ghjvprod(self, x, g, v):
        return np.array([-20.0 * g[0] * v[0]])


class HS7(_Model):
    """f = log(1 + x1^2) - x2, c = (1 + x1^2)^2 + x2^2 - 4, x0 = (2, 2); x* = (0, sqrt 3)   (test/test-2.jl:76-83)"""

    def __init__(self):
        super().__init__(2, 1, 2, np.array([2.0, 2.0]), name="HS7")

    def obj(self, x): return float(np.log(1 + x[0] ** 2) - x[1])
    def grad(self, x): return np.array([2 * x[0] / (1 + x[0] ** 2), -1.0])
    def cons(self, x): return np.array([(1 + x[0] ** 2) ** 2 + x[1] ** 2 - 4.0])
    def jac_structure(self): return np.array([1, 1]), np.array([1, 2])
    def jac_coord(self, x): return np.array([4 * x[0] * (1 + x[0] ** 2), 2 * x[1]])

    def _hc(self, x):  # Hessian of c (diagonal)
        return np.array([4 + 12 * x[0] ** 2, 2.0])

    def hprod(self, x, y, v, obj_weight=1.0):
        hf = np.array([(2 - 2 * x[0] ** 2) / (1 + x[0] ** 2) ** 2, 0.0])
        return (obj_weight * hf + y[0] * self._hc(x)) * np.asarray(v)

    def ghjvprod(self, x, g, v):
        return np.array([np.sum(np.asarray(g) * self._hc(x) * np.asarray(v))])


class EqQPModel(_Model):
    """Host view of problems.EqQP: f = 1/2 x'diag(q)x + d'x, c = Ax - b."""

    def __init__(self, qp):
        super().__init__(qp.n, qp.m, qp.nnz, qp.x, name=qp.name)
        self.qp = qp
        self._A = qp.scipy_csr()
        rows = np.repeat(np.arange(qp.m, dtype=np.int64), np.diff(qp.rowptr)) + 1
        self._struct = (rows, qp.colind.astype(np.int64) + 1)

    def obj(self, x): return float(x @ (0.5 * self.qp.qdiag * x + self.qp.d))
    def grad(self, x): return self.qp.qdiag * x + self.qp.d
    def cons(self, x): return self._A @ x - self.qp.b
    def jac_structure(self): return self._struct
    def jac_coord(self, x): return self.qp.vals
    def jtprod(self, x, v): return self._A.T @ v
    def jprod(self, x, v): return self._A @ v
    def hprod(self, x, y, v, obj_weight=1.0): return obj_weight * self.qp.qdiag * np.asarray(v)
