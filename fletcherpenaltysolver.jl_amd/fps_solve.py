"""`fps_solve` -- the entry point of the reference (src/FletcherPenaltySolver.jl:127-186) on top of the HIP back-ends.

SURVEY.md section 8(f) rank 4: a condensed host mirror of the outer loop of src/algo.jl:26-288 for equality-constrained
problems WITHOUT bounds (the case whose every obj / grad! / hprod! goes through the accelerated KKT solves).  What is
mirrored: the parameter schedule of `AlgoData` (src/parameters.jl:69-94), the sub-problem / outer stopping logic
(`Fletcher_penalty_optimality_check`, src/FletcherPenaltySolver.jl:28-50), `update_parameters!` and
`update_parameters_unbdd!` (src/algo.jl:361-390), the tolerance tightening of a feasible-but-not-optimal iterate
(:192-199), and -- for host models -- the feasibility and random restoration phases (src/algo.jl:200-251, :295-359;
`feasibility_step`, src/feasibility.jl:21-189, with small dense solves where the reference runs lsmr / cg on operators).
What is NOT built (out of this build's scope): bounds / slack models and the third-party sub-solvers (ipopt,
knitro, tron, trunk); the device-resident loop (`fps_solve_device`) has no restoration phases -- an iterate that would
enter them ends with status "infeasible" / "stalled" there.  The unconstrained
sub-problem is solved by one of two small built-in methods: `lbfgs` (objgrad! only) or `trunk` (trust-region
Newton-CG on hprod!, i.e. two more KKT solves per CG iteration: the caller SURVEY ranks next after grad!).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from .nlpmodels import SlackModel, has_bounds, has_inequalities
from .penalty_nlp import FletcherPenaltyNLP
from .qdsolver import qdsolver_correspondence

_SE = float(np.sqrt(np.finfo(float).eps))


@dataclass
class AlgoData:
    """Defaults of src/parameters.jl:69-94 (T = Float64)."""
    sigma_0: float = 1e3
    sigma_max: float = 1.0 / _SE
    sigma_update: float = 2.0
    rho_0: float = 1.0
    rho_max: float = 1.0 / _SE
    rho_update: float = 2.0
    delta_0: float = _SE
    delta_max: float = 1.0 / _SE
    delta_update: float = 10.0
    eta_1: float = 0.0
    eta_update: float = 1.0
    subsolver_max_iter: int = 20000
    subpb_unbounded_threshold: float = 1.0 / _SE
    lagrange_bound: float = 1.0 / _SE
    hessian_approx: int = 2
    # key of qdsolver_correspondence: the reference's default :ldlt (parameters.jl:290).  On the device "ldlt" = the direct
    # back-end ("hip_ldlt") whenever the normal equations have a narrow band (every small model does), the iterative one
    # ("hip" = "iterative") otherwise -- qdsolver.AutoQDSolver; which one ran is in stats.solver_specific["qds_backend"].
    # "hip_direct": the dense direct back-end.
    qds_solver: str = "ldlt"
    subproblem_solver: str = "lbfgs"


@dataclass
class ExecutionStats:
    """The fields of SolverCore.GenericExecutionStats the reference fills (src/algo.jl:253-268)."""
    status: str = "unknown"
    solution: np.ndarray | None = None
    objective: float = float("nan")
    primal_feas: float = float("nan")
    dual_feas: float = float("nan")
    multipliers: np.ndarray | None = None
    iter: int = 0
    elapsed_time: float = 0.0
    solver_specific: dict = field(default_factory=dict)


# ------------------------------------------------------------------ vectors: numpy arrays or torch tensors (host or device)

def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch"


def _copy(a):
    return a.clone() if _is_torch(a) else a.copy()


def _empty_like(a):
    return a.new_empty(a.shape) if _is_torch(a) else np.empty_like(a)


def _zeros_like(a):
    return a.new_zeros(a.shape) if _is_torch(a) else np.zeros_like(a)


def _dot(a, b):
    return float(a @ b)


def _nrm2(a):
    return float(a.norm()) if _is_torch(a) else float(np.linalg.norm(a))


def _nrminf(a):
    return float(a.abs().max()) if _is_torch(a) else float(np.linalg.norm(a, np.inf))


def _same(a, b):
    return bool((a == b).all())


# ------------------------------------------------------------------ sub-problem solvers (unconstrained min of phi)

def _lbfgs(fp, x, atol, rtol, max_iter, unbounded_below, mem=7, deadline=None):
    """Limited-memory BFGS with an Armijo backtracking line search.  Returns (x, status, g)."""
    g = _empty_like(x)
    f, _ = fp.objgrad_(x, g)
    tol = atol + rtol * _nrminf(g)
    S, Y = [], []
    for it in range(max_iter):
        if deadline is not None and time.perf_counter() > deadline:  # the caller's max_time (Stopping.jl hands it to the sub-solver)
            return x, "max_time", g
        if _nrminf(g) <= tol:
            return x, "optimal", g
        if f < -unbounded_below:
            return x, "unbounded", g
        q = _copy(g)
        al = []
        for s, y in zip(reversed(S), reversed(Y)):
            a = _dot(s, q) / _dot(y, s)
            al.append(a)
            q -= a * y
        if S:
            q *= _dot(S[-1], Y[-1]) / _dot(Y[-1], Y[-1])
        for (s, y), a in zip(zip(S, Y), reversed(al)):
            b = _dot(y, q) / _dot(y, s)
            q += (a - b) * s
        d = -q
        slope = _dot(g, d)
        if slope >= 0.0:  # not a descent direction: restart from steepest descent
            S, Y = [], []
            d, slope = -g, -_dot(g, g)
        t = 1.0 if S else min(1.0, 1.0 / max(_nrm2(g), 1e-16))
        gn = _empty_like(x)
        # Armijo up to the rounding of phi itself (close to a solution the decrease g's ~ |g|^2 / |H| drops below
        # eps |phi| long before |g| reaches the tolerance: a plain test then rejects every step, or accepts a null one)
        fnoise = 10.0 * np.finfo(float).eps * max(abs(f), 1.0)
        for _ in range(60):
            xn = x + t * d
            fn, _ = fp.objgrad_(xn, gn)
            if np.isfinite(fn) and fn <= f + 1e-4 * t * slope + fnoise:
                break
            t *= 0.5
        else:
            return x, "stalled", g
        s, y = xn - x, gn - g
        if _nrm2(s) <= np.finfo(float).eps * max(1.0, _nrm2(x)):  # a null step: no way forward from here
            return x, "stalled", g
        if _dot(s, y) > 1e-12 * _nrm2(s) * _nrm2(y):
            S.append(s)
            Y.append(y)
            if len(S) > mem:
                S.pop(0)
                Y.pop(0)
        x, f, g = xn, fn, gn
    return x, "max_iter", g


def _plbfgs(fp, x, atol, rtol, max_iter, unbounded_below, mem=7, deadline=None):
    """Bound-constrained sub-problem  min phi(x), lvar <= x <= uvar  (the role `tron` / `ipopt` play for the reference when
    the model has bounds, src/parameters.jl:84, :323): limited-memory BFGS on the free variables with a projected Armijo
    search along P(x + t d).  `fp.bounds = (lvar, uvar)`.  Stops on the projected gradient x - P(x - g) (optim_check_bounded).
    Returns (x, status, g)."""
    lo, hi = fp.bounds
    proj = lambda z: np.minimum(np.maximum(z, lo), hi)
    x = proj(x)
    g = _empty_like(x)
    f, _ = fp.objgrad_(x, g)
    pg = x - proj(x - g)
    tol = atol + rtol * _nrminf(pg)
    S, Y = [], []
    for it in range(max_iter):
        if deadline is not None and time.perf_counter() > deadline:  # the caller's max_time (Stopping.jl hands it to the sub-solver)
            return x, "max_time", g
        pg = x - proj(x - g)
        if _nrminf(pg) <= tol:
            return x, "optimal", g
        if f < -unbounded_below:
            return x, "unbounded", g
        free = ~(((x <= lo) & (g > 0.0)) | ((x >= hi) & (g < 0.0)))   # not pressed against an active bound
        q = np.where(free, g, 0.0)
        al = []
        for s, y in zip(reversed(S), reversed(Y)):
            a = _dot(s, q) / _dot(y, s)
            al.append(a)
            q -= a * y
        if S:
            q *= _dot(S[-1], Y[-1]) / _dot(Y[-1], Y[-1])
        for (s, y), a in zip(zip(S, Y), reversed(al)):
            b = _dot(y, q) / _dot(y, s)
            q += (a - b) * s
        d = np.where(free, -q, 0.0)
        if _dot(g, d) >= 0.0:  # not a descent direction: projected steepest descent
            S, Y = [], []
            d = -pg
        t = 1.0 if S else min(1.0, 1.0 / max(_nrm2(pg), 1e-16))
        gn = _empty_like(x)
        for _ in range(60):
            xn = proj(x + t * d)
            fn, _ = fp.objgrad_(xn, gn)
            if np.isfinite(fn) and fn <= f + 1e-4 * _dot(g, xn - x):
                break
            t *= 0.5
        else:
            return x, "stalled", g
        s, y = xn - x, gn - g
        if not s.any():
            return x, "stalled", g
        if _dot(s, y) > 1e-12 * _nrm2(s) * _nrm2(y):
            S.append(s)
            Y.append(y)
            if len(S) > mem:
                S.pop(0)
                Y.pop(0)
        x, f, g = xn, fn, gn
    return x, "max_iter", g


def _trunk(fp, x, atol, rtol, max_iter, unbounded_below, deadline=None):
    """Trust-region Newton-CG (Steihaug-Toint) on hprod!.  Returns (x, status, g)."""
    n = x.shape[0]
    g = _empty_like(x)
    f, _ = fp.objgrad_(x, g)
    tol = atol + rtol * _nrminf(g)
    radius = max(1.0, 0.1 * _nrm2(g))
    Hd = _empty_like(x)
    for it in range(max_iter):
        if deadline is not None and time.perf_counter() > deadline:  # the caller's max_time (Stopping.jl hands it to the sub-solver)
            return x, "max_time", g
        gnorm = _nrm2(g)
        if _nrminf(g) <= tol:
            return x, "optimal", g
        if f < -unbounded_below:
            return x, "unbounded", g
        # Steihaug CG on  min g's + s'Hs/2,  ||s|| <= radius
        s, r, d = _zeros_like(x), _copy(g), -g
        cg_tol = min(0.1, np.sqrt(gnorm)) * gnorm
        for _ in range(min(2 * n + 10, 500)):
            fp.hprod_(x, d, Hd)
            dHd = _dot(d, Hd)
            rr = _dot(r, r)
            if dHd <= 1e-14 * _dot(d, d):  # negative curvature: to the boundary
                s = s + _to_boundary(s, d, radius) * d
                break
            a = rr / dHd
            if _nrm2(s + a * d) >= radius:
                s = s + _to_boundary(s, d, radius) * d
                break
            s = s + a * d
            r = r + a * Hd
            if _nrm2(r) <= cg_tol:
                break
            d = -r + (_dot(r, r) / rr) * d
        fp.hprod_(x, s, Hd)
        pred = -(_dot(g, s) + 0.5 * _dot(s, Hd))
        gn = _empty_like(x)
        fn, _ = fp.objgrad_(x + s, gn)
        rho_tr = (f - fn) / pred if pred > 0 else -1.0
        if not np.isfinite(fn):
            rho_tr = -1.0
        if rho_tr >= 1e-4:
            x, f, g = x + s, fn, gn
            if rho_tr > 0.75 and _nrm2(s) > 0.9 * radius:
                radius *= 2.0
        else:
            radius = 0.25 * max(_nrm2(s), 1e-16)
            if radius < 1e-14 * max(1.0, _nrm2(x)):
                return x, "stalled", g
        if rho_tr < 0.25 and rho_tr >= 1e-4:
            radius *= 0.5
    return x, "max_iter", g


def _to_boundary(s, d, radius):
    a, b, c = _dot(d, d), 2.0 * _dot(s, d), _dot(s, s) - radius * radius
    return (-b + np.sqrt(max(b * b - 4.0 * a * c, 0.0))) / (2.0 * a)


_SUBSOLVERS = {"lbfgs": _lbfgs, "trunk": _trunk}


# ------------------------------------------------------------------ outer loop

class _HostPenalty:
    """What the outer loop needs from a penalty model, for the host mirror FletcherPenaltyNLP."""

    def __init__(self, fp, nlp):
        self.fp, self.nlp = fp, nlp
        self.objgrad_, self.hprod_ = fp.objgrad_, fp.hprod_
        self.bounds = (nlp.meta.lvar, nlp.meta.uvar) if has_bounds(nlp) else None

    sigma = property(lambda s: s.fp.sigma, lambda s, v: setattr(s.fp, "sigma", v))
    rho = property(lambda s: s.fp.rho, lambda s, v: setattr(s.fp, "rho", v))
    delta = property(lambda s: s.fp.delta, lambda s, v: setattr(s.fp, "delta", v))
    eta = property(lambda s: s.fp.eta, lambda s, v: setattr(s.fp, "eta", v))

    def set_xk(self, x):
        self.fp.xk = _copy(x)

    def invalidate(self):
        self.fp.shahx = None  # phi changed: the memo of _compute_ys_gs! is stale

    def state(self, x):
        """(f(x), ||c(x)||_2, ys(x), ||c(x)||_inf) -- memoised: no new solve when x was the last point evaluated."""
        self.fp._compute_ys_gs(x)
        return self.fp.fx, _nrm2(self.fp.cx), self.fp.ys, _nrminf(self.fp.cx)

    def primal_inf(self, x):
        c = self.nlp.cons(x)
        ucon = getattr(self.nlp.meta, "ucon", self.nlp.meta.lcon)
        return _nrminf(np.maximum(np.maximum(c - ucon, self.nlp.meta.lcon - c), 0.0))   # FletcherPenaltySolver.jl:41

    def grad_f(self, x):
        return self.nlp.grad(x)

    def info(self):
        return {"counters": dict(self.fp.counters)}

    # -- restoration phases (host models only)
    def random_restoration(self, x, atol, rng):
        """random_restoration! (src/algo.jl:340-359): x += radius * rand(n), radius = min(max(atol, 1/sigma, 1e-3), 1)."""
        radius = min(max(atol, 1.0 / self.sigma, 1e-3), 1.0)
        z = x + radius * rng.random(x.size)
        return z if self.bounds is None else np.minimum(np.maximum(z, self.bounds[0]), self.bounds[1])

    def restoration_feasibility(self, x, feas_tol, atol, rng):
        """restoration_feasibility! (src/algo.jl:295-333): a feasibility step; if it fails, a random perturbation."""
        nlp = self.nlp
        c = nlp.cons(x) - nlp.meta.lcon
        z, ok = feasibility_step(nlp, x, c, feas_tol, feas_tol)
        if ok and self.bounds is not None:   # (the reference's feasibility step ignores the bounds too; keep the iterate inside)
            z = np.minimum(np.maximum(z, self.bounds[0]), self.bounds[1])
        return z if ok else self.random_restoration(x, atol, rng)


class _PlainModel:
    """The user model itself as a sub-problem (ncon = 0): objgrad_ / hprod_ of f, bounds passed through."""

    def __init__(self, nlp):
        self.nlp = nlp
        self.bounds = (nlp.meta.lvar, nlp.meta.uvar) if has_bounds(nlp) else None

    def objgrad_(self, x, g):
        g[:] = self.nlp.grad(x)
        return float(self.nlp.obj(x)), g

    def hprod_(self, x, v, Hv):
        Hv[:] = self.nlp.hprod(x, np.zeros(0), v)
        return Hv


def _tr_step(J, c, radius):
    """TR_lsmr (src/feasibility.jl:208-235): min ||c + J d|| s.t. ||d|| <= radius.  The reference runs lsmr with a
    radius on the Jacobian operator; for the small host models of this mirror: the minimum-norm Gauss-Newton step, cut
    back to the trust-region boundary."""
    d = -np.linalg.lstsq(J, c, rcond=None)[0]
    nd = np.linalg.norm(d)
    if nd > radius:
        d *= radius / nd
    return d


def feasibility_step(nlp, x, cx, rho, ctol, *, eta1=1e-3, eta2=0.66, sigma1=0.25, sigma2=2.0, delta0=1.0,
                     bad_steps_lim=3, expected_decrease=0.95, max_feas_iter=1000):
    """feasibility_step (src/feasibility.jl:21-189; defaults of GNSolver, src/parameters.jl:160-170): trust-region
    Gauss-Newton on min ||c(x) - lcon|| with the aggressive second-order step (Hz + Jz'Jz) d = Jz'cz after
    `bad_steps_lim` poor steps.  Returns (z, success)."""
    m, n = nlp.meta.ncon, nlp.meta.nvar
    rows, cols = nlp.jac_structure()

    def jac(z):
        J = np.zeros((m, n))
        np.add.at(J, (np.asarray(rows) - 1, np.asarray(cols) - 1), nlp.jac_coord(z))
        return J

    z, cz, Jz = np.array(x, float), np.array(cx, float), jac(x)
    normcz = np.linalg.norm(cz)
    radius, it, bad, failed, infeasible = delta0, 0, 0, False, False
    while not (normcz <= rho or it > max_feas_iter or infeasible):
        d = _tr_step(Jz, cz, radius)
        infeasible = np.linalg.norm(d) < ctol * min(normcz, 1.0)                    # :226
        if infeasible:
            failed = True                                                            # :67-69
        else:
            zp = z + d
            czp = nlp.cons(zp) - nlp.meta.lcon
            normczp = np.linalg.norm(czp)
            pred = 0.5 * (normcz ** 2 - np.linalg.norm(Jz @ d + cz) ** 2)            # :75-76
            ared = 0.5 * (normcz ** 2 - normczp ** 2)
            if not (pred > 0) or ared / pred < eta1:
                radius = max(1e-8, radius * sigma1)                                   # :78-80
            else:
                bad = bad + 1 if normczp / normcz > expected_decrease else 0          # :85-89
                if ared / pred > eta2 and np.linalg.norm(d) >= 0.99 * radius:
                    radius *= sigma2
                z, cz, normcz, Jz = zp, czp, normczp, jac(zp)
        if normcz > rho and (bad >= bad_steps_lim or failed):                        # :117-140: aggressive normal step
            H = np.column_stack([nlp.hprod(z, cz, e, obj_weight=0.0) for e in np.eye(n)])
            d = np.linalg.lstsq(H + Jz.T @ Jz, Jz.T @ cz, rcond=None)[0]
            zp = z - d
            czp = nlp.cons(zp) - nlp.meta.lcon
            nczp = np.linalg.norm(czp)
            if nczp < normcz:
                infeasible, failed = False, False
                z, cz, normcz, Jz = zp, czp, nczp, jac(zp)
            elif np.linalg.norm(d) < ctol * min(nczp, 1.0):
                infeasible = True
        it += 1
    return z, bool(normcz <= rho)


def fps_solve(nlp, x0=None, *, atol=_SE, rtol=_SE, max_iter=100, max_time=300.0, verbose=0, qds=None, callback=None,
              **kwargs):
    """stats = fps_solve(nlp, x0; kwargs...)   (src/FletcherPenaltySolver.jl:127-186 -> src/algo.jl:26-288).
    Keyword arguments are the fields of `AlgoData`, the others go to the back-end's constructor (parameters.jl:299: e.g.
    `ldlt_r2`); `qds` overrides the back-end instance.  `callback(nlp, pen, stats)` is
    called before the first and after every outer iteration (algo.jl:109, :283) with the running statistics (solution,
    objective, residuals, multipliers, iter); setting `stats.status = "user"` stops the loop."""
    meta = AlgoData(**{k: v for k, v in kwargs.items() if k in AlgoData.__dataclass_fields__})
    x = np.array(nlp.meta.x0 if x0 is None else x0, float)
    if getattr(nlp.meta, "ncon", 0) == 0:
        # no constraints: the sub-problem solver is called on the model itself (algo.jl:63-75)
        t0 = time.perf_counter()
        plain = _PlainModel(nlp)
        sub = _plbfgs if plain.bounds is not None else _SUBSOLVERS[meta.subproblem_solver]
        xs, sub_status, g = sub(plain, x, atol, rtol, max(meta.subsolver_max_iter, 1000), meta.subpb_unbounded_threshold)
        st = ExecutionStats(solution=xs)
        st.status = {"optimal": "first_order", "unbounded": "unbounded", "max_iter": "max_iter"}.get(sub_status, "stalled")
        st.objective = float(nlp.obj(xs))
        st.primal_feas = 0.0
        st.dual_feas = _nrminf(g if plain.bounds is None else xs - np.minimum(np.maximum(xs - g, plain.bounds[0]), plain.bounds[1]))
        st.iter, st.elapsed_time, st.multipliers = 1, time.perf_counter() - t0, np.zeros(0)
        return st
    orig = nlp
    if has_inequalities(nlp):                                                   # FletcherPenaltySolver.jl:139-143
        ns = int((np.asarray(nlp.meta.lcon) != np.asarray(nlp.meta.ucon)).sum())
        x = np.concatenate([x, np.zeros(ns)])
        nlp = SlackModel(nlp)
    if has_bounds(nlp):
        x = np.minimum(np.maximum(x, nlp.meta.lvar), nlp.meta.uvar)
    # src/parameters.jl:299: the back-end's constructor receives the keyword arguments that are not AlgoData's
    # (LDLtSolver's ldlt_tol / ldlt_r1 / ldlt_r2, IterativeSolver's tolerances)
    qkw = {k: v for k, v in kwargs.items() if k not in AlgoData.__dataclass_fields__}
    qds = qds if qds is not None else qdsolver_correspondence[meta.qds_solver](nlp, 0.0, **qkw)
    fp = FletcherPenaltyNLP(nlp, sigma=meta.sigma_0, rho=meta.rho_0, delta=0.0, hessian_approx=meta.hessian_approx,
                            x0=x, qds=qds)                                                     # algo.jl:45-52
    stats = _outer_loop(_HostPenalty(fp, nlp), x, meta, atol, rtol, max_iter, max_time, verbose, callback)
    # which back-end served the seam (a registry key may route: "ldlt" -> the banded direct or the iterative one)
    stats.solver_specific["qds_solver"] = meta.qds_solver
    stats.solver_specific["qds_backend"] = getattr(qds, "qds_backend", type(qds).__name__)
    if getattr(qds, "ldlt_r2", None) is not None:
        stats.solver_specific["ldlt_r2"] = qds.ldlt_r2
    if nlp is not orig:                                                          # :153-170: back to the user's variables
        stats.solver_specific["slack"] = stats.solution[orig.meta.nvar:].copy()
        stats.solution = stats.solution[: orig.meta.nvar].copy()
    return stats


def _outer_loop(pen, x, meta, atol, rtol, max_iter, max_time, verbose, callback=None):
    """src/algo.jl:26-288 on a penalty model `pen` (host mirror or device-resident); x: numpy array or torch tensor."""
    bounds = getattr(pen, "bounds", None)
    sub = _plbfgs if bounds is not None else _SUBSOLVERS[meta.subproblem_solver]
    t_start = time.perf_counter()
    stats = ExecutionStats(solution=_copy(x))

    def score(x, lam, res):
        """Fletcher_penalty_optimality_check (FletcherPenaltySolver.jl:28-50), no bounds."""
        nxk = max(_nrm2(x), 1.0)
        nlk = max(_nrm2(lam), 1.0) if lam is not None else 1.0
        if bounds is not None:   # x - max(min(x - res, uvar), lvar), not scaled (FletcherPenaltySolver.jl:42-43)
            return pen.primal_inf(x) / nxk, _nrminf(x - np.minimum(np.maximum(x - res, bounds[0]), bounds[1]))
        return pen.primal_inf(x) / nxk, _nrminf(res) / nlk

    p0, d0 = score(x, None, pen.grad_f(x))
    tol = max(atol, rtol * max(p0, d0))        # Stopping.jl's default tol_check(atol, rtol, optimality0)
    sub_atol, sub_rtol = atol, rtol            # meta.atol_sub / rtol_sub are the identity by default (parameters.jl:88-89)
    feas_tol = atol
    stalling = unsuccessful = unbounded = 0
    feasibility_phase = restoration_phase = False   # each restoration is entered at most once (algo.jl:58-59, :201, :225)
    can_restore = hasattr(pen, "restoration_feasibility")
    rng = np.random.default_rng(1234)               # (the reference's tests seed the global generator, runtests.jl:6)
    status = "unknown"
    it = 0
    if max(p0, d0) <= tol:
        status = "first_order"
    if callback is not None:                      # algo.jl:109
        callback(getattr(pen, "nlp", None), pen, stats)
        if stats.status == "user":
            status = "user"
    while status == "unknown":
        it += 1
        x_prev = x
        xs, sub_status, res = sub(pen, _copy(x), sub_atol, sub_rtol, meta.subsolver_max_iter,
                                  meta.subpb_unbounded_threshold, deadline=t_start + max_time)
        fx_user, ncx, ys, ncx_inf = pen.state(xs)
        unb_mult = _nrminf(ys) >= meta.lagrange_bound
        feas = ncx < feas_tol
        if sub_status == "optimal":                      # algo.jl:121-151 (taken whatever the multiplier bound says)
            if _same(xs, x_prev):                       # :122-124 -- only ever incremented here, reset by the other branches
                stalling += 1
            unsuccessful = unbounded = 0
            x = xs
            stats.solution, stats.multipliers = _copy(x), -ys
            stats.objective = fx_user
            stats.primal_feas, stats.dual_feas = score(x, stats.multipliers, res)
            if max(stats.primal_feas, stats.dual_feas) <= tol:
                status = "first_order"
                break
        elif sub_status == "unbounded" or unb_mult:                                             # :152-160
            stalling = unsuccessful = 0
            unbounded += 1
            if ncx_inf < feas_tol:                      # :156-157 (the Inf-norm here, the 2-norm everywhere else)
                status = "unbounded"
                break
        else:                                                                                   # :161-181
            stalling = unbounded = 0
            unsuccessful += 1
        if pen.sigma > meta.sigma_max or pen.rho > meta.rho_max or pen.delta > meta.delta_max:  # :189-190
            status = "stalled" if feas else "infeasible"
            break
        if it >= max_iter:
            status = "max_iter"
            break
        if time.perf_counter() - t_start > max_time:
            status = "max_time"
            break
        # ---- not finished: algo.jl:193-251
        if sub_status == "optimal":
            if feas:                                           # tighten the sub-problem tolerances (:195-199)
                sub_atol = max(sub_atol / 10.0, np.finfo(float).eps)
                sub_rtol = max(sub_rtol / 10.0, np.finfo(float).eps)
                pen.eta = max(meta.eta_1, pen.eta * meta.eta_update)
                pen.set_xk(x)
                pen.invalidate()
            elif can_restore and not feasibility_phase and (stalling >= 3 or sub_atol < np.finfo(float).eps):
                # most likely stuck at an infeasible stationary point, or an undetected unbounded problem (:201-209)
                feasibility_phase = True
                unbounded = 0
                x = pen.restoration_feasibility(x, feas_tol, atol, rng)
                stalling = unsuccessful = 0
                pen.invalidate()
            elif stalling >= 3 or sub_atol < np.finfo(float).eps:  # infeasible stationary point (:210-213)
                status = "infeasible"
                break
            else:
                _update_parameters(pen, meta, feas)
        elif sub_status == "unbounded" or unb_mult:
            if can_restore and not feasibility_phase and unbounded >= 3 and not feas:          # :217-226
                feasibility_phase = True
                unbounded = 0
                x = pen.restoration_feasibility(x, feas_tol, atol, rng)
                stalling = unsuccessful = 0
                pen.invalidate()
            elif can_restore and not restoration_phase and unbounded >= 3:                       # :227-231
                restoration_phase = True
                unbounded = 0
                x = pen.random_restoration(x, atol, rng)
                pen.invalidate()
            elif not can_restore and unbounded >= 3 and not feas:
                status = "infeasible"
                break
            else:
                pen.delta = meta.delta_0 if pen.delta == 0.0 else pen.delta * meta.delta_update     # :380-390
                _update_parameters(pen, meta, feas)
        else:
            if can_restore and not restoration_phase and unsuccessful >= 3 and feas:             # :241-245
                restoration_phase = True
                unsuccessful = 0
                x = pen.random_restoration(x, atol, rng)
                pen.invalidate()
            elif can_restore and not feasibility_phase and unsuccessful >= 3 and not feas:       # :246-255
                feasibility_phase = True
                unsuccessful = 0
                x = pen.restoration_feasibility(x, feas_tol, atol, rng)
                stalling = unsuccessful = 0
                pen.invalidate()
            elif not can_restore and unsuccessful >= 3:
                status = "stalled"
                break
            else:
                _update_parameters(pen, meta, feas)
        if verbose:
            print(f"fps_solve it {it:3d} sub={sub_status:9s} f={fx_user: .6e} |c|={ncx:.2e} sigma={pen.sigma:.1e} "
                  f"rho={pen.rho:.1e} delta={pen.delta:.1e}")
        if callback is not None:                  # algo.jl:283 (end of the main loop; `:user` ends it, :111)
            stats.iter = it
            if stats.solution is None or sub_status != "optimal":
                stats.solution = _copy(x)
            callback(getattr(pen, "nlp", None), pen, stats)
            if stats.status == "user":
                status = "user"
    stats.status = status
    stats.iter = it
    stats.elapsed_time = time.perf_counter() - t_start
    if stats.multipliers is None:
        stats.multipliers = -pen.state(stats.solution)[2]
    if not np.isfinite(stats.primal_feas):
        stats.primal_feas, stats.dual_feas = score(stats.solution, stats.multipliers, pen.grad_f(stats.solution))
    stats.solver_specific = {"sigma": pen.sigma, "rho": pen.rho, "delta": pen.delta, **pen.info()}
    return stats


def _update_parameters(pen, meta, feas):
    """update_parameters! (src/algo.jl:361-375): sigma always, rho when the iterate is infeasible."""
    pen.sigma *= meta.sigma_update
    if not feas:
        pen.rho *= meta.rho_update
    pen.invalidate()


# ------------------------------------------------------------------ the same loop on the device-resident eq-QP model

class _DevicePenalty:
    """The penalty function of a DeviceEqQP (fpsq_qp_objgrad / fpsq_qp_hprod): every vector stays in HBM (torch
    tensors); the host only sees scalars."""

    def __init__(self, dev, torch, hessian_approx=2):
        self.dev, self.torch, self.hessian_approx = dev, torch, int(hessian_approx)
        d = torch.device("cuda", int(dev.opts.device))
        qp = dev.qp
        self.q = torch.from_numpy(np.ascontiguousarray(qp.qdiag)).to(d)
        self.d = torch.from_numpy(np.ascontiguousarray(qp.d)).to(d)
        self.b = torch.from_numpy(np.ascontiguousarray(qp.b)).to(d)
        self.ys = torch.empty(qp.m, dtype=torch.float64, device=d)
        self.c = torch.empty(qp.m, dtype=torch.float64, device=d)
        self.scratch = torch.empty(qp.n, dtype=torch.float64, device=d)
        self._at = None
        self.nobjgrad = self.nhprod = 0

    sigma = property(lambda s: s.dev.sigma, lambda s, v: setattr(s.dev, "sigma", v))
    rho = property(lambda s: s.dev.rho, lambda s, v: setattr(s.dev, "rho", v))
    eta = property(lambda s: s.dev.eta, lambda s, v: setattr(s.dev, "eta", v))
    delta = property(lambda s: s.dev.delta, lambda s, v: s.dev.set_delta(v))

    def objgrad_(self, x, g):
        self.nobjgrad += 1
        fx, _ = self.dev.objgrad(x, gx=g, ys=self.ys, xk=self._xk if self.dev.eta > 0.0 else None)
        self._at = x
        return fx, g

    def hprod_(self, x, v, Hv):
        self.nhprod += 1
        self.dev.hprod(v, Hv, self.hessian_approx)   # Val(2) (model:521-570) or Val(1) (:572-634)
        return Hv

    _xk = None

    def set_xk(self, x):
        self._xk = x.clone()

    def invalidate(self):
        self._at = None

    def _cons(self, x):
        self.c.copy_(self.b)
        self.dev.jac_mul(0, 1.0, x, -1.0, self.c)  # c = A x - b
        return self.c

    def state(self, x):
        if self._at is None or self._at is not x:
            self.objgrad_(x, self.scratch)
        f = float(0.5 * (x @ (self.q * x)) + self.d @ x)
        c = self._cons(x)
        return f, _nrm2(c), self.ys, _nrminf(c)

    def primal_inf(self, x):
        return _nrminf(self._cons(x))

    def grad_f(self, x):
        return self.q * x + self.d

    def info(self):
        return {"objgrad_calls": self.nobjgrad, "hprod_calls": self.nhprod}


def fps_solve_device(dev, x0, *, atol=_SE, rtol=_SE, max_iter=100, max_time=300.0, verbose=0, **kwargs):
    """fps_solve on a device-resident equality QP (`DeviceEqQP`): x0 and every iterate are torch tensors in HBM, each
    obj/grad! is one fpsq_qp_objgrad, each Hessian product of the `trunk` sub-solver one fpsq_qp_hprod.
    Returns ExecutionStats whose `solution` / `multipliers` are device tensors."""
    import torch

    meta = AlgoData(**{k: v for k, v in kwargs.items() if k in AlgoData.__dataclass_fields__})
    dev.sigma, dev.rho, dev.eta = meta.sigma_0, meta.rho_0, 0.0
    dev.set_delta(0.0)
    return _outer_loop(_DevicePenalty(dev, torch, meta.hessian_approx), x0, meta, atol, rtol, max_iter, max_time, verbose)
