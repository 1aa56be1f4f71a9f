"""Minimal NLPModels.jl-shaped user models (L0 of SURVEY.md §1) used by the tests and the benchmark.

Method names follow NLPModels.jl (obj, grad, cons, jac_structure, jac_coord, jtprod, hprod); indices returned by
jac_structure are 1-based like Julia's.  These are HOST models: the reference's user model is host code too.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


class _Model:
    def __init__(self, nvar, ncon, nnzj, x0, lcon=None, name="model", lin=(), ucon=None, lvar=None, uvar=None):
        lin = np.asarray(sorted(lin), dtype=np.int64)          # 0-based indices of the linear constraints
        nln = np.setdiff1d(np.arange(ncon, dtype=np.int64), lin)
        lcon = np.zeros(ncon) if lcon is None else np.asarray(lcon, float)
        ucon = lcon.copy() if ucon is None else np.asarray(ucon, float)   # default: equalities c(x) = lcon
        self.meta = SimpleNamespace(nvar=nvar, ncon=ncon, nnzj=nnzj, x0=np.asarray(x0, float), lcon=lcon, ucon=ucon,
                                    lvar=np.full(nvar, -np.inf) if lvar is None else np.asarray(lvar, float),
                                    uvar=np.full(nvar, np.inf) if uvar is None else np.asarray(uvar, float),
                                    jfix=np.flatnonzero(lcon == ucon), name=name,
                                    lin=lin, nln=nln, nlin=int(lin.size), nnln=int(nln.size))

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


def has_bounds(nlp):
    """NLPModels.has_bounds: some variable has a finite bound."""
    lv, uv = getattr(nlp.meta, "lvar", None), getattr(nlp.meta, "uvar", None)
    return bool((lv is not None and np.isfinite(lv).any()) or (uv is not None and np.isfinite(uv).any()))


def has_inequalities(nlp):
    """NLPModels.has_inequalities: some constraint is not an equality (lcon < ucon)."""
    uc = getattr(nlp.meta, "ucon", None)
    return bool(uc is not None and (np.asarray(uc) != np.asarray(nlp.meta.lcon)).any())


class SlackModel(_Model):
    """NLPModelsModifiers.SlackModel as fps_solve applies it (src/FletcherPenaltySolver.jl:139-143): every inequality
    lcon_i <= c_i(x) <= ucon_i becomes the equality c_i(x) - s_i = 0 with the bounds on the new variable s_i; equalities
    and the bounds of x are kept.  Variables [x; s], s in the order of the inequality constraints."""

    def __init__(self, nlp):
        self.model = nlp
        bm = nlp.meta
        self._ineq = np.flatnonzero(bm.lcon != bm.ucon)
        ns = int(self._ineq.size)
        rows, cols = nlp.jac_structure()
        self._rows = np.concatenate([np.asarray(rows), self._ineq + 1])
        self._cols = np.concatenate([np.asarray(cols), bm.nvar + 1 + np.arange(ns)])
        lcon = bm.lcon.copy()
        lcon[self._ineq] = 0.0
        super().__init__(bm.nvar + ns, bm.ncon, bm.nnzj + ns, np.concatenate([bm.x0, np.zeros(ns)]), lcon=lcon,
                         name=bm.name + "-slack", lin=bm.lin,
                         lvar=np.concatenate([bm.lvar, bm.lcon[self._ineq]]),
                         uvar=np.concatenate([bm.uvar, bm.ucon[self._ineq]]))
        self._n0 = bm.nvar

    def obj(self, x): return self.model.obj(x[: self._n0])
    def grad(self, x): return np.concatenate([self.model.grad(x[: self._n0]), np.zeros(self.meta.nvar - self._n0)])

    def cons(self, x):
        c = np.array(self.model.cons(x[: self._n0]), float)
        c[self._ineq] -= x[self._n0:]
        return c

    def jac_structure(self): return self._rows, self._cols
    def jac_coord(self, x): return np.concatenate([self.model.jac_coord(x[: self._n0]), -np.ones(self._ineq.size)])

    def hprod(self, x, y, v, obj_weight=1.0):
        out = np.zeros(self.meta.nvar)
        out[: self._n0] = self.model.hprod(x[: self._n0], y, v[: self._n0], obj_weight=obj_weight)
        return out

    def ghjvprod(self, x, g, v): return self.model.ghjvprod(x[: self._n0], g[: self._n0], v[: self._n0])


class NonlinearConstraintsView(_Model):
    """The model seen by the penalty function when `explicit_linear_constraints = true`: same objective, only the
    NONLINEAR constraints (NLPModels' cons_nln!, jac_nln_structure!, jac_nln_coord!, jtprod_nln!, and the `lag_mul`
    scatter of hprod_nln! / ghjvprod_nln!, src/model-Fletcherpenaltynlp.jl:282-350)."""

    def __init__(self, nlp):
        self.base = nlp
        rows, cols = nlp.jac_structure()
        rows = np.asarray(rows, dtype=np.int64) - 1
        keep = np.isin(rows, nlp.meta.nln)
        renum = -np.ones(nlp.meta.ncon, dtype=np.int64)
        renum[nlp.meta.nln] = np.arange(nlp.meta.nnln)
        self._keep = keep
        self._struct = (renum[rows[keep]] + 1, np.asarray(cols, dtype=np.int64)[keep])
        super().__init__(nlp.meta.nvar, nlp.meta.nnln, int(keep.sum()), nlp.meta.x0, lcon=nlp.meta.lcon[nlp.meta.nln],
                         name=nlp.meta.name + " (nonlinear constraints)")

    def obj(self, x): return self.base.obj(x)
    def grad(self, x): return self.base.grad(x)
    def cons(self, x): return np.asarray(self.base.cons(x))[self.base.meta.nln]
    def jac_structure(self): return self._struct
    def jac_coord(self, x): return np.asarray(self.base.jac_coord(x))[self._keep]

    def _lag(self, y):
        full = np.zeros(self.base.meta.ncon)
        full[self.base.meta.nln] = y
        return full

    def hprod(self, x, y, v, obj_weight=1.0): return self.base.hprod(x, self._lag(y), v, obj_weight=obj_weight)
    def ghjvprod(self, x, g, v): return np.asarray(self.base.ghjvprod(x, g, v))[self.base.meta.nln]


class LinearPlusCircle(_Model):
    """f = x'x, c1 = x1 + x2 + x3 - 1 (linear), c2 = x1^2 + x2^2 - 1/2 (nonlinear): a small model with both kinds of
    equality constraints for `explicit_linear_constraints` (the reference's option, src/parameters.jl:84)."""

    def __init__(self, with_linear=True):
        self.with_linear = with_linear
        super().__init__(3, 2 if with_linear else 1, 5 if with_linear else 2, np.array([0.3, 0.8, -0.2]),
                         name="linear+circle", lin=(0,) if with_linear else ())

    def obj(self, x): return float(x @ x)
    def grad(self, x): return 2.0 * np.asarray(x, float)

    def cons(self, x):
        nl = x[0] ** 2 + x[1] ** 2 - 0.5
        return np.array([x.sum() - 1.0, nl]) if self.with_linear else np.array([nl])

    def jac_structure(self):
        if self.with_linear:
            return np.array([1, 1, 1, 2, 2]), np.array([1, 2, 3, 1, 2])
        return np.array([1, 1]), np.array([1, 2])

    def jac_coord(self, x):
        nl = np.array([2 * x[0], 2 * x[1]])
        return np.concatenate([np.ones(3), nl]) if self.with_linear else nl

    def hprod(self, x, y, v, obj_weight=1.0):
        ynl = y[1] if self.with_linear else y[0]
        return obj_weight * 2.0 * np.asarray(v, float) + ynl * np.array([2 * v[0], 2 * v[1], 0.0])

    def ghjvprod(self, x, g, v):
        val = 2 * g[0] * v[0] + 2 * g[1] * v[1]
        return np.array([0.0, val]) if self.with_linear else np.array([val])


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


class TorchEqQPModel(EqQPModel):
    """The same model with its Jacobian RESIDENT ON THE GPU (torch): `jac_coord` returns a device tensor, which the direct
    back-ends take in place (fpsq_dense_set_jacobian_coo / fpsq_band_factorize_coo: no Jacobian value crosses PCIe), and
    c(x), J v, J'v run on the device -- a device-resident user model behind the host-side seam (x, g, c stay host
    vectors of length n / m, as the seam's signatures have them)."""

    def __init__(self, qp, device=0):
        import torch

        super().__init__(qp)
        self._t = torch
        self._dev = torch.device("cuda", device)
        crow = torch.from_numpy(np.ascontiguousarray(qp.rowptr, dtype=np.int64)).to(self._dev)
        col = torch.from_numpy(np.ascontiguousarray(qp.colind, dtype=np.int64)).to(self._dev)
        self._vals = torch.from_numpy(np.ascontiguousarray(qp.vals, dtype=np.float64)).to(self._dev)
        if qp.nnz == qp.m * qp.n:   # a dense block: plain GEMV
            self._Ad = self._vals.view(qp.m, qp.n)
            self._mv = lambda v: self._Ad @ v
            self._tmv = lambda v: self._Ad.T @ v
        else:
            A = torch.sparse_csr_tensor(crow, col, self._vals, size=(qp.m, qp.n))
            At = A.to_sparse_coo().t().to_sparse_csr()
            self._mv = lambda v: (A @ v.unsqueeze(1)).squeeze(1)
            self._tmv = lambda v: (At @ v.unsqueeze(1)).squeeze(1)
        self._b = torch.from_numpy(np.ascontiguousarray(qp.b)).to(self._dev)

    def _up(self, v):
        return self._t.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(self._dev)

    def cons(self, x): return (self._mv(self._up(x)) - self._b).cpu().numpy()
    def jprod(self, x, v): return self._mv(self._up(v)).cpu().numpy()
    def jtprod(self, x, v): return self._tmv(self._up(v)).cpu().numpy()
    def jac_coord(self, x): return self._vals


class ADModel(_Model):
    """`ADNLPModel(f, x0, c, lcon, ucon)` of the reference's tests: a host user model whose derivatives come from
    automatic differentiation (here torch.autograd on the CPU in fp64, where the reference uses ForwardDiff through
    ADNLPModels.jl).  `f` and `c` take a 1-D torch tensor and return a scalar / a 1-D tensor (or a list of scalars).
    Dense Jacobian structure, like ADNLPModel's default."""

    def __init__(self, f, x0, c, lcon, name="admodel", lin=(), ucon=None, lvar=None, uvar=None):
        import torch

        x0 = np.asarray(x0, float)
        lcon = np.asarray(lcon, float)
        n, m = x0.size, lcon.size
        super().__init__(n, m, n * m, x0, lcon=lcon, name=name, lin=lin, ucon=ucon, lvar=lvar, uvar=uvar)
        self._t = torch
        self._f = f
        self._c = lambda x: (lambda v: torch.stack(list(v)) if isinstance(v, (list, tuple)) else v)(c(x))
        self._rows = np.repeat(np.arange(1, m + 1), n)
        self._cols = np.tile(np.arange(1, n + 1), m)

    def _x(self, x, grad=False):
        t = self._t.tensor(np.asarray(x, float), dtype=self._t.float64)
        return t.requires_grad_(True) if grad else t

    def obj(self, x):
        return float(self._f(self._x(x)))

    def grad(self, x):
        t = self._x(x, True)
        v = self._f(t)
        if not getattr(v, "requires_grad", False):  # constant objective
            return np.zeros(self.meta.nvar)
        (g,) = self._t.autograd.grad(v, t, allow_unused=True)
        return np.zeros(self.meta.nvar) if g is None else g.numpy()

    def cons(self, x):
        return self._c(self._x(x)).detach().numpy().astype(float)

    def jac_structure(self):
        return self._rows, self._cols

    def _jac(self, x):
        J = self._t.autograd.functional.jacobian(self._c, self._x(x))
        return J.numpy().reshape(self.meta.ncon, self.meta.nvar)

    def jac_coord(self, x):
        return self._jac(x).ravel()

    def hprod(self, x, y, v, obj_weight=1.0):
        """(obj_weight Hess f + sum_i y_i Hess c_i) v"""
        torch = self._t
        t = self._x(x, True)
        yv = torch.tensor(np.asarray(y, float), dtype=torch.float64)
        lag = obj_weight * self._f(t) + (yv * self._c(t)).sum()
        if not getattr(lag, "requires_grad", False):
            return np.zeros(self.meta.nvar)
        (g,) = torch.autograd.grad(lag, t, create_graph=True, allow_unused=True)
        if g is None or not g.requires_grad:
            return np.zeros(self.meta.nvar)
        (hv,) = torch.autograd.grad(g, t, grad_outputs=torch.tensor(np.asarray(v, float), dtype=torch.float64),
                                    allow_unused=True)
        return np.zeros(self.meta.nvar) if hv is None else hv.numpy()

    def ghjvprod(self, x, g, v):
        """(g' Hess c_i v)_i"""
        out = np.zeros(self.meta.ncon)
        for i in range(self.meta.ncon):
            e = np.zeros(self.meta.ncon)
            e[i] = 1.0
            out[i] = float(np.dot(g, self.hprod(x, e, v, obj_weight=0.0)))
        return out


def reference_test_problems():
    """The equality-constrained problems of the reference's integration tests (test/test-2.jl:1-287,
    test/rank-deficient.jl:22-36), as ADModels: name -> (model, checks)."""
    import torch

    P = {}
    P["rosenbrock_sum"] = ADModel(lambda x: (x[0] - 1.0) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2, [-1.2, 1.0],
                                  lambda x: [x.sum()], [1.0], name="Rosenbrock with sum x = 1")       # test-2.jl:1-9
    P["hs8"] = ADModel(lambda x: 0.0 * x[0] - 1.0, [2.0, 1.0],
                       lambda x: [x[0] ** 2 + x[1] ** 2 - 25, x[0] * x[1] - 9], [0.0, 0.0], name="HS8")  # :99-106
    P["hs9"] = ADModel(lambda x: torch.sin(np.pi * x[0] / 12) * torch.cos(np.pi * x[1] / 16), [0.0, 0.0],
                       lambda x: [4 * x[0] - 3 * x[1]], [0.0], name="HS9")                              # :125-132
    P["hs26"] = ADModel(lambda x: (x[0] - x[1]) ** 2 + (x[1] - x[2]) ** 4, [-2.6, 2.0, 2.0],
                        lambda x: [(1 + x[1] ** 2) * x[0] + x[2] ** 4 - 3], [0.0], name="HS26")         # :151-158
    P["hs27"] = ADModel(lambda x: 0.01 * (x[0] - 1) ** 2 + (x[1] - x[0] ** 2) ** 2, [2.0, 2.0, 2.0],
                        lambda x: [x[0] + x[2] ** 2 + 1.0], [0.0], name="HS27")                         # :177-184
    P["huyer_neumaier"] = ADModel(lambda x: x[0] ** 3 * x[1] ** 3, [0.0, 0.0],
                                  lambda x: [x[0] ** 2 + x[1] ** 2 - 1], [0.0], name="Huyer-Neumaier")  # :208-209
    P["estrin_a1"] = ADModel(lambda x: 0.0 * x[0], [0.0], lambda x: [x[0] ** 3 + x[0] - 2.0], [0.0],
                             name="Estrin et al. A.1")                                                 # :238
    P["flt"] = ADModel(lambda x: (x[1] - 1) ** 2, [1.0, 0.0], lambda x: [x[0] ** 2, x[0] ** 3], [0.0, 0.0],
                       name="FLT")                                                                     # :264-271
    P["hs61"] = ADModel(lambda x: 4 * x[0] ** 2 + 2 * x[1] ** 2 + 2 * x[2] ** 2 - 33 * x[0] + 16 * x[1] - 24 * x[2],
                        [0.0, 0.0, 0.0], lambda x: [3 * x[0] - 2 * x[1] ** 2 - 7, 4 * x[0] - x[2] ** 2 - 11],
                        [0.0, 0.0], name="HS61")                                           # rank-deficient.jl:23-29
    return P


def bounded_test_problems():
    """Problems with bounds and / or inequality constraints (the classes of the reference's test/solvertest.jl:
    bound_constrained_nlp and the :bnd / :ineq / :eqnbnd / :gen families): name -> (model, x*, f*)."""
    P = {}
    P["bnd_eq"] = (ADModel(lambda x: (x[0] - 2.0) ** 2 + (x[1] - 1.0) ** 2, [0.0, 0.0], lambda x: [x[0] + x[1]], [2.0],
                           lvar=[0.0, -np.inf], uvar=[0.5, np.inf], name="active bound + linear equality"),
                   np.array([0.5, 1.5]), 2.5)
    P["inactive_bounds"] = (ADModel(lambda x: (x[0] - 1.0) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2, [-1.2, 1.0],
                                    lambda x: [x.sum()], [1.0], lvar=[-5.0, -5.0], uvar=[5.0, 5.0],
                                    name="Rosenbrock, sum x = 1, inactive bounds"),
                            np.array([0.61879562, 0.38120438]), None)
    P["hs14"] = (ADModel(lambda x: (x[0] - 2.0) ** 2 + (x[1] - 1.0) ** 2, [2.0, 2.0],
                         lambda x: [x[0] - 2 * x[1] + 1.0, -x[0] ** 2 / 4 - x[1] ** 2 + 1.0], [0.0, 0.0],
                         ucon=[0.0, np.inf], name="HS14"),
                 np.array([0.5 * (np.sqrt(7) - 1), 0.25 * (np.sqrt(7) + 1)]), 9 - 2.875 * np.sqrt(7))
    P["hs71"] = (ADModel(lambda x: x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2], [1.0, 5.0, 5.0, 1.0],
                         lambda x: [x[0] * x[1] * x[2] * x[3], (x * x).sum()], [25.0, 40.0], ucon=[np.inf, 40.0],
                         lvar=[1.0] * 4, uvar=[5.0] * 4, name="HS71"),
                 np.array([1.0, 4.7429994, 3.8211503, 1.3794082]), 17.0140173)
    return P
