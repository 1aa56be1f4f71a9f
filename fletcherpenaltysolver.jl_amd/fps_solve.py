"""`fps_solve` -- the entry point of the reference (src/FletcherPenaltySolver.jl:127-186) on top of the HIP back-ends.

SURVEY.md section 8(f) rank 4: a condensed host mirror of the outer loop of src/algo.jl:26-288 for equality-constrained
problems WITHOUT bounds (the case whose every obj / grad! / hprod! goes through the accelerated KKT solves).  What is
mirrored: the parameter schedule of `AlgoData` (src/parameters.jl:69-94), the sub-problem / outer stopping logic
(`Fletcher_penalty_optimality_check`, src/FletcherPenaltySolver.jl:28-50), `update_parameters!` and
`update_parameters_unbdd!` (src/algo.jl:361-390), the tolerance tightening of a feasible-but-not-optimal iterate
(:192-199).  What is NOT built (out of this build's scope): bounds / slack models, the feasibility and random
restoration phases (src/feasibility.jl, src/algo.jl:200-251) -- an iterate that would enter them ends with status
"infeasible" / "stalled" instead -- and the third-party sub-solvers (ipopt, knitro, tron, trunk).  The unconstrained
sub-problem is solved by one of two small built-in methods: `lbfgs` (objgrad! only) or `trunk` (trust-region
Newton-CG on hprod!, i.e. two more KKT solves per CG iteration: the caller SURVEY ranks next after grad!).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

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
    qds_solver: str = "hip"  # key of qdsolver_correspondence (the reference's default is :ldlt, parameters.jl:290)
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


# ------------------------------------------------------------------ sub-problem solvers (unconstrained min of phi)

def _lbfgs(fp, x, atol, rtol, max_iter, unbounded_below, mem=7):
    """Limited-memory BFGS with an Armijo backtracking line search.  Returns (x, status, g)."""
    n = x.size
    g = np.empty(n)
    f, _ = fp.objgrad_(x, g)
    tol = atol + rtol * np.linalg.norm(g, np.inf)
    S, Y = [], []
    for it in range(max_iter):
        if np.linalg.norm(g, np.inf) <= tol:
            return x, "optimal", g
        if f < -unbounded_below:
            return x, "unbounded", g
        q = g.copy()
        al = []
        for s, y in zip(reversed(S), reversed(Y)):
            a = (s @ q) / (y @ s)
            al.append(a)
            q -= a * y
        if S:
            q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
        for (s, y), a in zip(zip(S, Y), reversed(al)):
            b = (y @ q) / (y @ s)
            q += (a - b) * s
        d = -q
        slope = g @ d
        if slope >= 0.0:  # not a descent direction: restart from steepest descent
            S, Y = [], []
            d, slope = -g, -(g @ g)
        t = 1.0 if S else min(1.0, 1.0 / max(np.linalg.norm(g), 1e-16))
        gn = np.empty(n)
        for _ in range(60):
            xn = x + t * d
            fn, _ = fp.objgrad_(xn, gn)
            if np.isfinite(fn) and fn <= f + 1e-4 * t * slope:
                break
            t *= 0.5
        else:
            return x, "stalled", g
        s, y = xn - x, gn - g
        if s @ y > 1e-12 * np.linalg.norm(s) * np.linalg.norm(y):
            S.append(s)
            Y.append(y)
            if len(S) > mem:
                S.pop(0)
                Y.pop(0)
        x, f, g = xn, fn, gn
    return x, "max_iter", g


def _trunk(fp, x, atol, rtol, max_iter, unbounded_below):
    """Trust-region Newton-CG (Steihaug-Toint) on hprod!.  Returns (x, status, g)."""
    n = x.size
    g = np.empty(n)
    f, _ = fp.objgrad_(x, g)
    tol = atol + rtol * np.linalg.norm(g, np.inf)
    radius = max(1.0, 0.1 * np.linalg.norm(g))
    Hd = np.empty(n)
    for it in range(max_iter):
        gnorm = np.linalg.norm(g)
        if np.linalg.norm(g, np.inf) <= tol:
            return x, "optimal", g
        if f < -unbounded_below:
            return x, "unbounded", g
        # Steihaug CG on  min g's + s'Hs/2,  ||s|| <= radius
        s, r, d = np.zeros(n), g.copy(), -g.copy()
        cg_tol = min(0.1, np.sqrt(gnorm)) * gnorm
        for _ in range(2 * n + 10):
            fp.hprod_(x, d, Hd)
            dHd = d @ Hd
            rr = r @ r
            if dHd <= 1e-14 * (d @ d):  # negative curvature: to the boundary
                s = s + _to_boundary(s, d, radius) * d
                break
            a = rr / dHd
            if np.linalg.norm(s + a * d) >= radius:
                s = s + _to_boundary(s, d, radius) * d
                break
            s = s + a * d
            r = r + a * Hd
            if np.linalg.norm(r) <= cg_tol:
                break
            d = -r + (r @ r) / rr * d
        fp.hprod_(x, s, Hd)
        pred = -(g @ s + 0.5 * (s @ Hd))
        gn = np.empty(n)
        fn, _ = fp.objgrad_(x + s, gn)
        rho_tr = (f - fn) / pred if pred > 0 else -1.0
        if not np.isfinite(fn):
            rho_tr = -1.0
        if rho_tr >= 1e-4:
            x, f, g = x + s, fn, gn
            if rho_tr > 0.75 and np.linalg.norm(s) > 0.9 * radius:
                radius *= 2.0
        else:
            radius = 0.25 * max(np.linalg.norm(s), 1e-16)
            if radius < 1e-14 * max(1.0, np.linalg.norm(x)):
                return x, "stalled", g
        if rho_tr < 0.25 and rho_tr >= 1e-4:
            radius *= 0.5
    return x, "max_iter", g


def _to_boundary(s, d, radius):
    a, b, c = d @ d, 2.0 * (s @ d), s @ s - radius * radius
    return (-b + np.sqrt(max(b * b - 4.0 * a * c, 0.0))) / (2.0 * a)


_SUBSOLVERS = {"lbfgs": _lbfgs, "trunk": _trunk}


# ------------------------------------------------------------------ outer loop

def fps_solve(nlp, x0=None, *, atol=_SE, rtol=_SE, max_iter=100, max_time=300.0, verbose=0, qds=None, **kwargs):
    """stats = fps_solve(nlp, x0; kwargs...)   (src/FletcherPenaltySolver.jl:127-186 -> src/algo.jl:26-288).
    Keyword arguments are the fields of `AlgoData`; `qds` overrides the back-end instance."""
    meta = AlgoData(**{k: v for k, v in kwargs.items() if k in AlgoData.__dataclass_fields__})
    if getattr(nlp.meta, "ncon", 0) == 0:
        raise ValueError("fps_solve: this mirror covers equality-constrained problems (ncon > 0)")
    x = np.array(nlp.meta.x0 if x0 is None else x0, float)
    qds = qds if qds is not None else qdsolver_correspondence[meta.qds_solver](nlp, 0.0)
    fp = FletcherPenaltyNLP(nlp, sigma=meta.sigma_0, rho=meta.rho_0, delta=0.0, hessian_approx=meta.hessian_approx,
                            x0=x, qds=qds)                                                     # algo.jl:45-52
    sub = _SUBSOLVERS[meta.subproblem_solver]
    t_start = time.perf_counter()
    stats = ExecutionStats(solution=x.copy(), multipliers=np.zeros(nlp.meta.ncon))
    lcon = nlp.meta.lcon

    def score(x, lam, res):
        """Fletcher_penalty_optimality_check (FletcherPenaltySolver.jl:28-50), no bounds."""
        nxk = max(np.linalg.norm(x), 1.0)
        nlk = max(np.linalg.norm(lam), 1.0)
        return np.linalg.norm(nlp.cons(x) - lcon, np.inf) / nxk, np.linalg.norm(res, np.inf) / nlk

    g0 = nlp.grad(x)
    p0, d0 = score(x, stats.multipliers, g0)
    tol = max(atol, rtol * max(p0, d0))        # Stopping.jl's default tol_check(atol, rtol, optimality0)
    sub_atol, sub_rtol = atol, rtol            # meta.atol_sub / rtol_sub are the identity by default (parameters.jl:88-89)
    feas_tol = atol
    stalling = unsuccessful = unbounded = 0
    status = "unknown"
    it = 0
    if max(p0, d0) <= tol:
        status = "first_order"
    while status == "unknown":
        it += 1
        x_prev = x
        xs, sub_status, res = sub(fp, x.copy(), sub_atol, sub_rtol, meta.subsolver_max_iter,
                                  meta.subpb_unbounded_threshold)
        fp._compute_ys_gs(xs)  # phi's caches at the returned point (memoised: no new solve when it was the last one)
        unb_mult = np.linalg.norm(fp.ys, np.inf) >= meta.lagrange_bound
        ncx = np.linalg.norm(fp.cx)
        feas = ncx < feas_tol
        if sub_status == "optimal" and not unb_mult:                                            # algo.jl:121-151
            stalling = stalling + 1 if np.array_equal(xs, x_prev) else 0
            unsuccessful = unbounded = 0
            x = xs
            stats.solution, stats.multipliers = x.copy(), -fp.ys.copy()
            stats.objective = fp.fx
            stats.primal_feas, stats.dual_feas = score(x, stats.multipliers, res)
            if max(stats.primal_feas, stats.dual_feas) <= tol:
                status = "first_order"
                break
        elif sub_status == "unbounded" or unb_mult:                                             # :152-160
            stalling = unsuccessful = 0
            unbounded += 1
            if feas:
                status = "unbounded"
                break
        else:                                                                                   # :161-181
            stalling = unbounded = 0
            unsuccessful += 1
        if fp.sigma > meta.sigma_max or fp.rho > meta.rho_max or fp.delta > meta.delta_max:     # :189-190
            status = "stalled" if feas else "infeasible"
            break
        if it >= max_iter:
            status = "max_iter"
            break
        if time.perf_counter() - t_start > max_time:
            status = "max_time"
            break
        # ---- not finished: algo.jl:193-251
        if sub_status == "optimal" and not unb_mult:
            if feas:                                           # tighten the sub-problem tolerances (:195-199)
                sub_atol = max(sub_atol / 10.0, np.finfo(float).eps)
                sub_rtol = max(sub_rtol / 10.0, np.finfo(float).eps)
                fp.eta = max(meta.eta_1, fp.eta * meta.eta_update)
                fp.xk = x.copy()
                fp.shahx = None
            elif stalling >= 3 or sub_atol < np.finfo(float).eps:  # infeasible stationary point (:200-214, no restoration)
                status = "infeasible"
                break
            else:
                _update_parameters(fp, meta, feas)
        elif sub_status == "unbounded" or unb_mult:
            if unbounded >= 3 and not feas:                   # would enter the feasibility phase (:216-224)
                status = "infeasible"
                break
            fp.delta = meta.delta_0 if fp.delta == 0.0 else fp.delta * meta.delta_update         # :380-390
            _update_parameters(fp, meta, feas)
        else:
            if unsuccessful >= 3:                             # would enter a restoration phase (:236-247)
                status = "stalled"
                break
            _update_parameters(fp, meta, feas)
        if verbose:
            print(f"fps_solve it {it:3d} sub={sub_status:9s} f={fp.fx: .6e} |c|={ncx:.2e} sigma={fp.sigma:.1e} "
                  f"rho={fp.rho:.1e} delta={fp.delta:.1e}")
    stats.status = status
    stats.iter = it
    stats.elapsed_time = time.perf_counter() - t_start
    if stats.solution is not None and not np.isfinite(stats.primal_feas):
        stats.primal_feas, stats.dual_feas = score(stats.solution, stats.multipliers, nlp.grad(stats.solution))
    stats.solver_specific = {"sigma": fp.sigma, "rho": fp.rho, "delta": fp.delta, "counters": dict(fp.counters)}
    return stats


def _update_parameters(fp, meta, feas):
    """update_parameters! (src/algo.jl:361-375): sigma always, rho when the iterate is infeasible."""
    fp.sigma *= meta.sigma_update
    if not feas:
        fp.rho *= meta.rho_update
    fp.shahx = None  # phi changed: the memo of _compute_ys_gs! is stale
