"""Host-side mirror of `FletcherPenaltyNLP` (src/model-Fletcherpenaltynlp.jl:58-103): the hot part
`_compute_ys_gs!` (:234-252), `obj` (:352-370), `grad!` (:372-401), `objgrad!` (:403-437), and the matrix-free
Hessian products `hprod!` Val(2) (:521-570) and Val(1) (:572-634) -- SURVEY.md §8f rows 1-2 -- which reach the
back-end through `solve_two_least_squares` / `solve_two_extras`, and the dense `hess_coord!` (:439-519, small problems).
"""
from __future__ import annotations

import numpy as np

from .qdsolver import HIPQDSolver, QDSolver


class FletcherPenaltyNLP:
    """FletcherPenaltyNLP(nlp, sigma, rho, delta, hessian_approx; qds = HIPQDSolver(nlp, 0.0))

    Defaults follow the keyword constructor at :194-204 (sigma_0 = 1, rho_0 = delta_0 = 0, Val(2)); the
    reference's default `qds` is the CPU LDLtSolver (:113), here it is the MI355X back-end."""

    def __init__(self, nlp, sigma=1.0, rho=0.0, delta=0.0, hessian_approx=2, x0=None, *, qds: QDSolver | None = None,
                 explicit_linear_constraints=False):
        self.nlp = nlp
        self.sigma, self.rho, self.delta, self.eta = sigma, rho, delta, 0.0
        self.hessian_approx = hessian_approx
        # explicit_linear_constraints (:112-141): only the NONLINEAR constraints are penalised -- `pen` is the model the
        # penalty function and the back-end see -- and the linear ones stay constraints of this model (meta.ncon = nlin,
        # cons / jprod / jtprod below), left to a sub-solver that handles linear constraints
        self.explicit_linear_constraints = bool(explicit_linear_constraints)
        if self.explicit_linear_constraints:
            from .nlpmodels import NonlinearConstraintsView
            self.pen = NonlinearConstraintsView(nlp)
        else:
            self.pen = nlp
        n, m = nlp.meta.nvar, self.pen.meta.ncon
        nlin = nlp.meta.nlin if self.explicit_linear_constraints else 0
        self.meta = type(nlp.meta)(nvar=n, ncon=nlin, x0=np.asarray(nlp.meta.x0 if x0 is None else x0, float),
                                   lcon=nlp.meta.lcon[nlp.meta.lin] if nlin else np.zeros(0),
                                   name=f"Fletcher penalization of {nlp.meta.name}")
        self.qdsolver = qds if qds is not None else HIPQDSolver(
            nlp, 0.0, explicit_linear_constraints=self.explicit_linear_constraints)
        self.shahx = None
        self.fx = float("nan")
        self.cx, self.gx = np.empty(m), np.empty(n)
        self.ys, self.gs = np.empty(m), np.empty(n)
        self.v, self.w = np.empty(n), np.empty(m)
        self.xk = np.zeros(n)
        self.counters = dict(neval_obj=0, neval_grad=0)

    # :260-269
    def cons_norhs(self, x):
        return self.pen.cons(x) - self.pen.meta.lcon  # (cons_nln! and lcon[nln] when the linear ones are explicit)

    # :215-227
    def linear_system2(self, x):
        return self.qdsolver.solve_two_mixed(self, x, self.gx, self.cx)

    # :234-252.  The reference memoises on Julia's hash(x); the key here is the full byte content.
    def _compute_ys_gs(self, x):
        key = hash(np.asarray(x, float).tobytes())
        if key != self.shahx:
            self.shahx = key
            self.fx = self.nlp.obj(x)
            self.gx[:] = self.nlp.grad(x)
            self.cx[:] = self.cons_norhs(x)
            p1, q1, p2, q2 = self.linear_system2(x)
            self.gs[:] = p1 + self.sigma * p2
            self.ys[:] = q1 + self.sigma * q2
            self.v[:] = p2
            self.w[:] = q2
        return self.gs, self.ys, self.v, self.w

    # :352-370
    def obj(self, x):
        x = np.asarray(x, float)
        assert x.size == self.meta.nvar
        self.counters["neval_obj"] += 1
        self._compute_ys_gs(x)
        c = self.cx
        fx = self.fx - c @ self.ys + self.rho / 2 * (c @ c)
        if self.eta > 0.0:
            fx += self.eta / 2 * np.linalg.norm(x - self.xk) ** 2
        return fx

    # :372-401  (grad!)
    def grad_(self, x, gx):
        x = np.asarray(x, float)
        assert x.size == self.meta.nvar and gx.size == self.meta.nvar
        self.counters["neval_grad"] += 1
        gs, ys, v, w = self._compute_ys_gs(x)
        Hsv = self.pen.hprod(x, ys, v, obj_weight=1.0)
        Sstw = self.pen.hprod(x, w, gs, obj_weight=0.0)
        gx[:] = gs - Hsv + self.sigma * v + Sstw
        if self.rho > 0.0:
            gx += self.pen.jtprod(x, self.cx) * self.rho
        if self.eta > 0.0:
            gx += self.eta * (x - self.xk)
        return gx

    def grad(self, x):
        return self.grad_(x, np.empty(self.meta.nvar))

    # :403-437  (objgrad!)
    def objgrad_(self, x, gx):
        x = np.asarray(x, float)
        self.counters["neval_obj"] += 1
        self.grad_(x, gx)
        c = self.cx
        fx = self.fx - c @ self.ys
        if self.rho > 0.0:
            fx += self.rho / 2 * (c @ c)
        if self.eta > 0.0:
            fx += self.eta / 2 * np.linalg.norm(x - self.xk) ** 2
        return fx, gx

    def objgrad(self, x):
        return self.objgrad_(x, np.empty(self.meta.nvar))

    # :521-570 (Val(2)) and :572-634 (Val(1))   (hprod!)
    def hprod_(self, x, v, Hv, obj_weight=1.0):
        x, v = np.asarray(x, float), np.asarray(v, float)
        assert x.size == self.meta.nvar and v.size == self.meta.nvar and Hv.size == self.meta.nvar
        self.counters["neval_hprod"] = self.counters.get("neval_hprod", 0) + 1
        sigma, rho = self.sigma, self.rho
        gs, ys, _, _ = self._compute_ys_gs(x)
        c = self.cx
        Jv = -ys                                                             # :536 / :590
        Hsv = self.pen.hprod(x, Jv, v, obj_weight=1.0)                       # :537 / :591
        p1, _, p2, _ = self.qdsolver.solve_two_least_squares(self, x, v, Hsv)  # :542 / :593
        Ptv = v - p1                                                         # :543 / :594
        HsPtv = self.pen.hprod(x, Jv, Ptv, obj_weight=1.0)                   # :545 / :598
        if self.hessian_approx == 2:
            Hv[:] = p2 - HsPtv + 2 * sigma * Ptv                             # :550
        else:
            Ssv = self.pen.ghjvprod(x, gs, v)                                # :600
            invJtJJv, invJtJSsv = self.qdsolver.solve_two_extras(self, x, v, Ssv)  # :602
            JtinvJtJSsv = self.pen.jtprod(x, invJtJSsv)                      # :606
            Hv[:] = p2 - HsPtv + 2 * sigma * Ptv - JtinvJtJSsv               # :612
            Hv -= self.pen.hprod(x, invJtJJv, gs, obj_weight=0.0)            # :613-614
        if rho > 0.0:
            JtJv = self.pen.jtprod(x, self.pen.jprod(x, v))                  # :557-558 / :621-622
            Hcv = self.pen.hprod(x, c, v, obj_weight=0.0)                    # :560 / :624
            Hv += Hcv + rho * JtJv if self.hessian_approx == 2 else rho * (Hcv + JtJv)  # :562 / :626
        if self.eta > 0.0:
            Hv += self.eta * v
        Hv *= obj_weight
        return Hv

    def hprod(self, x, v, obj_weight=1.0):
        return self.hprod_(x, v, np.empty(self.meta.nvar), obj_weight)

    # :439-519  (hess_structure! / hess_coord!): DENSE lower triangle, column by column -- small problems only.  Like the
    # reference it does not go through the QDSolver: (A A' + tau I)^-1 is formed explicitly (`pinv` there).
    def hess_structure(self):
        n = self.meta.nvar
        ij = [(i + 1, j + 1) for j in range(n) for i in range(j, n)]
        return np.array([a for a, _ in ij]), np.array([b for _, b in ij])

    def hess_coord(self, x, obj_weight=1.0):
        x = np.asarray(x, float)
        n, m = self.meta.nvar, self.pen.meta.ncon
        self.counters["neval_hess"] = self.counters.get("neval_hess", 0) + 1
        gs, ys, _, _ = self._compute_ys_gs(x)
        c = self.cx
        rows, cols = self.pen.jac_structure()
        A = np.zeros((m, n))
        np.add.at(A, (np.asarray(rows) - 1, np.asarray(cols) - 1), np.asarray(self.pen.jac_coord(x), float))
        eye = np.eye(n)
        dense = lambda y, w: np.column_stack([self.pen.hprod(x, y, eye[:, j], obj_weight=w) for j in range(n)])
        Hs = dense(-ys, 1.0)                                                 # :477
        tau = max(self.delta, 1e-14)
        invAtA = np.linalg.pinv(A @ A.T + tau * np.eye(m))                   # :480
        AinvAtA = A.T @ invAtA
        Pt = AinvAtA @ A
        Hx = Hs - Pt @ Hs - Hs @ Pt + 2.0 * self.sigma * Pt                  # :484
        if self.rho > 0.0:
            Hx += dense(c * self.rho, 0.0) + self.rho * (A.T @ A)            # :486-489
        if self.hessian_approx == 1:                                         # :491-498
            Ss = np.zeros((m, n))
            for k in range(m):
                ek = np.zeros(m)
                ek[k] = 1.0
                Ss[k, :] = np.column_stack([self.pen.hprod(x, ek, eye[:, j], obj_weight=0.0) for j in range(n)]).T @ gs
            Hx += -AinvAtA @ Ss - Ss.T @ invAtA @ A
        vals = np.array([Hx[i, j] for j in range(n) for i in range(j, n)]) * obj_weight
        if self.eta > 0.0:
            vals[[k for k, (i, j) in enumerate((i, j) for j in range(n) for i in range(j, n)) if i == j]] += obj_weight * self.eta
        return vals

    # :636-726 -- the linear constraints kept explicit (cons_lin!, jprod_lin!, jtprod_lin! of the wrapped model)
    def _lin_rows(self):
        rows, cols = self.nlp.jac_structure()
        rows = np.asarray(rows, dtype=np.int64) - 1
        keep = np.isin(rows, self.nlp.meta.lin)
        renum = -np.ones(self.nlp.meta.ncon, dtype=np.int64)
        renum[self.nlp.meta.lin] = np.arange(self.nlp.meta.nlin)
        return keep, renum[rows[keep]], np.asarray(cols, dtype=np.int64)[keep] - 1

    def cons(self, x):
        if not self.explicit_linear_constraints:
            return np.zeros(0)
        return np.asarray(self.nlp.cons(np.asarray(x, float)))[self.nlp.meta.lin]

    def jprod(self, x, v):
        out = np.zeros(self.meta.ncon)
        if self.meta.ncon:
            keep, r, c = self._lin_rows()
            np.add.at(out, r, np.asarray(self.nlp.jac_coord(x))[keep] * np.asarray(v, float)[c])
        return out

    def jtprod(self, x, v):
        out = np.zeros(self.meta.nvar)
        if self.meta.ncon:
            keep, r, c = self._lin_rows()
            np.add.at(out, c, np.asarray(self.nlp.jac_coord(x))[keep] * np.asarray(v, float)[r])
        return out
