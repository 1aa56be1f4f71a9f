"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see the header of fps_oracle.c).

Two independent checkers for the penalty-evaluation linear-solve path:

1. `exact_two_mixed` / `exact_two_least_squares` / `exact_two_extras`: the mathematical oracle.  A direct solve
   of  K [p; q] = rhs,  K = [I A'; A -delta I]  (dense numpy below 3000 unknowns, SuperLU beyond).  This is what the
   reference's LDLt back-end computes (solve_linear_system.jl:206-252) and what its iterative back-end
   approximates (:107-140).
2. ctypes bindings of fps_oracle.c, the iteration-for-iteration C restatement of the iterative back-end.

PARITY STATUS: pinned on the reference's known-answer tests (tests/golden/), unpinned against Krylov.jl's
iteration counts (Julia is not available in this pipeline).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_OMP = None


class Stats(C.Structure):
    _fields_ = [("solved", C.c_int32), ("inconsistent", C.c_int32), ("niter", C.c_int32),
                ("status", C.c_int32), ("rnorm", C.c_double), ("arnorm", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Options(C.Structure):
    _fields_ = [("ls_atol", C.c_double), ("ls_rtol", C.c_double), ("ls_itmax", C.c_int64),
                ("ln_atol", C.c_double), ("ln_rtol", C.c_double), ("ln_btol", C.c_double),
                ("ln_conlim", C.c_double), ("ln_itmax", C.c_int64),
                ("ne_atol", C.c_double), ("ne_rtol", C.c_double), ("ne_etol", C.c_double),
                ("ne_itmax", C.c_int64), ("ne_conlim", C.c_double),
                ("ls_axtol", C.c_double), ("ls_btol", C.c_double), ("ls_etol", C.c_double),
                ("ls_conlim", C.c_double), ("ln_method", C.c_int32), ("pad", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfps_oracle.so")
    so2 = os.path.join(_HERE, "libfps_oracle_omp.so")
    src = os.path.join(_HERE, "fps_oracle.c")
    if (force or not os.path.exists(so) or not os.path.exists(so2)
            or min(os.path.getmtime(so), os.path.getmtime(so2)) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib(threaded: bool = False):
    """threaded=True: the OpenMP build (bench.py's all-cores CPU baseline only; the parity oracle is the serial one)."""
    global _LIB, _LIB_OMP
    if threaded:
        if _LIB_OMP is None:
            build()
            _LIB_OMP = C.CDLL(os.path.join(_HERE, "libfps_oracle_omp.so"))
        return _LIB_OMP
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def set_sum_order(mode: int) -> None:
    """Summation order of the restatement's products (fps_oracle.c fpo_set_sum_order): 0 = left to right (the default),
    1 = long rows in the device's order, 2 = right to left.  Tests only; always set back to 0."""
    lib().fpo_set_sum_order(C.c_int(mode))


def default_options(n, m, **kw) -> Options:
    o = Options()
    lib().fpo_default_options(C.c_int64(n), C.c_int64(m), C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _csr64(rowptr, colind, vals):
    return (np.ascontiguousarray(rowptr, dtype=np.int64), np.ascontiguousarray(colind, dtype=np.int64),
            np.ascontiguousarray(vals, dtype=np.float64))


def spmv(m, n, rowptr, colind, vals, x, transposed=False):
    rp, ci, va = _csr64(rowptr, colind, vals)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(n if transposed else m)
    lib().fpo_spmv(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_int(transposed),
                   _p(x), _p(y))
    return y


def lsqr(m, n, rowptr, colind, vals, b, lam=0.0, atol=0.0, rtol=0.0, itmax=0, transposed=False,
         axtol=None, btol=None, etol=None, conlim=None):
    se = np.sqrt(np.finfo(float).eps)
    axtol = se if axtol is None else axtol
    btol = se if btol is None else btol
    etol = se if etol is None else etol
    conlim = 1 / se if conlim is None else conlim
    rp, ci, va = _csr64(rowptr, colind, vals)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(m if transposed else n)
    st = Stats()
    lib().fpo_lsqr(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_int(transposed),
                   _p(b), C.c_double(lam), C.c_double(atol), C.c_double(rtol), C.c_int64(itmax),
                   C.c_double(axtol), C.c_double(btol), C.c_double(etol), C.c_double(conlim), _p(x), C.byref(st))
    return x, st


def craig(m, n, rowptr, colind, vals, b, delta=0.0, atol=None, rtol=None, btol=None, conlim=None, itmax=0,
          transposed=False):
    se = np.sqrt(np.finfo(float).eps)
    atol = se if atol is None else atol
    rtol = se if rtol is None else rtol
    btol = se if btol is None else btol
    conlim = 1 / se if conlim is None else conlim
    rp, ci, va = _csr64(rowptr, colind, vals)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(m if transposed else n)
    y = np.empty(n if transposed else m)
    st = Stats()
    lib().fpo_craig(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_int(transposed),
                    _p(b), C.c_double(delta), C.c_double(atol), C.c_double(rtol), C.c_double(btol),
                    C.c_double(conlim), C.c_int64(itmax), _p(x), _p(y), C.byref(st))
    return x, y, st


def lnlq(m, n, rowptr, colind, vals, b, delta=0.0, atol=None, rtol=None, itmax=0, transposed=False):
    """C restatement of Krylov.jl lnlq! as the reference's generic solve_least_norm calls it (struct.jl:251-281)."""
    se = np.sqrt(np.finfo(float).eps)
    atol = se if atol is None else atol
    rtol = se if rtol is None else rtol
    rp, ci, va = _csr64(rowptr, colind, vals)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(m if transposed else n)
    y = np.empty(n if transposed else m)
    st = Stats()
    lib().fpo_lnlq(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_int(transposed),
                   _p(b), C.c_double(delta), C.c_double(atol), C.c_double(rtol), C.c_int64(itmax), _p(x), _p(y),
                   C.byref(st))
    return x, y, st


def minres_aat(m, n, rowptr, colind, vals, b, lam=0.0, atol=None, rtol=None, etol=None, conlim=None, itmax=0):
    se = np.sqrt(np.finfo(float).eps)
    atol = se if atol is None else atol
    rtol = se if rtol is None else rtol
    etol = se if etol is None else etol
    conlim = 1 / se if conlim is None else conlim
    rp, ci, va = _csr64(rowptr, colind, vals)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(m)
    st = Stats()
    lib().fpo_minres(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), _p(b),
                     C.c_double(lam), C.c_double(atol), C.c_double(rtol), C.c_double(etol), C.c_double(conlim),
                     C.c_int64(itmax), _p(x), C.byref(st))
    return x, st


def minres_kkt(m, n, rowptr, colind, vals, delta, bp=None, bq=None, atol=None, rtol=None, etol=None, conlim=None,
               itmax=0):
    """MINRES on K = [I A'; A -delta I] (fpo_minres_kkt): K [p; q] = [bp; bq]; returns (p, q, stats).  Not a path of the
    reference -- the checker of the library's kkt_method = FPSQ_KKT_MINRES_K."""
    se = np.sqrt(np.finfo(float).eps)
    atol = se if atol is None else atol
    rtol = se if rtol is None else rtol
    etol = se if etol is None else etol
    conlim = 1 / se if conlim is None else conlim
    rp, ci, va = _csr64(rowptr, colind, vals)
    bp = None if bp is None else np.ascontiguousarray(bp, dtype=np.float64)
    bq = None if bq is None else np.ascontiguousarray(bq, dtype=np.float64)
    x = np.empty(n + m)
    st = Stats()
    fn = lib().fpo_minres_kkt
    fn.restype = C.c_int
    fn(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_double(delta),
       None if bp is None else _p(bp), None if bq is None else _p(bq), C.c_double(atol), C.c_double(rtol),
       C.c_double(etol), C.c_double(conlim), C.c_int64(itmax), _p(x), C.byref(st))
    return x[:n].copy(), x[n:].copy(), st


def _two(fn, m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts, outs, threaded=False):
    rp, ci, va = _csr64(rowptr, colind, vals)
    if threaded:
        lib(True).fpo_omp_prepare(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va))
    rhs1 = np.ascontiguousarray(rhs1, dtype=np.float64)
    rhs2 = np.ascontiguousarray(rhs2, dtype=np.float64)
    opts = opts or default_options(n, m)
    st = (Stats * 2)()
    bufs = [np.empty(k) for k in outs]
    rc = fn(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), C.c_double(delta),
            C.byref(opts), _p(rhs1), _p(rhs2), *[_p(b) for b in bufs], st)
    if threaded:
        lib(True).fpo_omp_release()
    return (*bufs, [st[0], st[1]], rc)


def solve_two_mixed(m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts=None):
    """C restatement of solve_linear_system.jl:107-140 -> (p1, q1, p2, q2, [stats1, stats2], rc)."""
    return _two(lib().fpo_solve_two_mixed, m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts, (n, m, n, m))


def solve_two_least_squares(m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts=None, threaded=False):
    """C restatement of solve_linear_system.jl:79-105."""
    return _two(lib(threaded).fpo_solve_two_least_squares, m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts, (n, m, n, m),
                threaded)


def solve_two_extras(m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts=None):
    """C restatement of solve_linear_system.jl:45-77."""
    return _two(lib().fpo_solve_two_extras, m, n, rowptr, colind, vals, delta, rhs1, rhs2, opts, (m, m))


def qp_objgrad(qp, x, sigma, rho, delta, eta=0.0, xk=None, opts=None, threaded=False):
    """C restatement of objgrad! (model-Fletcherpenaltynlp.jl:403-437) on the eq-QP user model.
    Returns dict(fx, gx, ys, gs, stats, rc)."""
    rp, ci, va = _csr64(qp.rowptr, qp.colind, qp.vals)
    n, m = qp.n, qp.m
    x = np.ascontiguousarray(x, dtype=np.float64)
    xk = np.zeros(n) if xk is None else np.ascontiguousarray(xk, dtype=np.float64)
    opts = opts or default_options(n, m)
    gx, ys, gs = np.empty(n), np.empty(m), np.empty(n)
    fx = C.c_double()
    st = (Stats * 2)()
    if threaded:
        lib(True).fpo_omp_prepare(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va))
    rc = lib(threaded).fpo_qp_objgrad(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va),
                              _p(qp.qdiag), _p(qp.d), _p(qp.b), _p(x), C.c_double(sigma), C.c_double(rho),
                              C.c_double(delta), C.c_double(eta), _p(xk), C.byref(opts), _p(gx), C.byref(fx),
                              _p(ys), _p(gs), st)
    if threaded:
        lib(True).fpo_omp_release()
    return dict(fx=fx.value, gx=gx, ys=ys, gs=gs, stats=[st[0], st[1]], rc=rc)


def qp_hprod(qp, v, sigma, rho, delta, eta=0.0, approx=2, opts=None):
    """C restatement of hprod! (model-Fletcherpenaltynlp.jl:521-570 for approx = 2, :572-634 for approx = 1) on the
    eq-QP user model.  Returns dict(Hv, stats (4: the two LSQR solves, then LSQR + MINRES of solve_two_extras), rc)."""
    rp, ci, va = _csr64(qp.rowptr, qp.colind, qp.vals)
    n, m = qp.n, qp.m
    v = np.ascontiguousarray(v, dtype=np.float64)
    opts = opts or default_options(n, m)
    Hv = np.empty(n)
    st = (Stats * 4)()
    rc = lib().fpo_qp_hprod(C.c_int64(m), C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(va), _p(qp.qdiag),
                            _p(v), C.c_double(sigma), C.c_double(rho), C.c_double(delta), C.c_double(eta),
                            C.c_int(approx), C.byref(opts), _p(Hv), st)
    return dict(Hv=Hv, stats=[st[k] for k in range(4)], rc=rc)


# ----------------------------------------------------------------------------- exact (direct) oracle

def _kkt_solve(A, delta, rhs):
    """Solve [I A'; A -delta I] sol = rhs (rhs: (n+m) x k).  A: scipy CSR or dense ndarray, m x n."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    m, n = A.shape
    if n + m <= 3000:
        Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
        K = np.block([[np.eye(n), Ad.T], [Ad, -delta * np.eye(m)]])
        if delta == 0.0 and np.linalg.matrix_rank(Ad) < m:
            return np.linalg.lstsq(K, rhs, rcond=None)[0]
        return np.linalg.solve(K, rhs)
    As = sp.csr_matrix(A)
    K = sp.bmat([[sp.identity(n), As.T], [As, -delta * sp.identity(m)]], format="csc")
    return spla.splu(K).solve(rhs)


def exact_two_mixed(A, delta, rhs1, rhs2):
    """(p1, q1, p2, q2) with K[p1;q1] = [rhs1;0], K[p2;q2] = [0;rhs2] (solve_linear_system.jl:236-251)."""
    m, n = A.shape
    rhs = np.zeros((n + m, 2))
    rhs[:n, 0] = rhs1
    rhs[n:, 1] = rhs2
    sol = _kkt_solve(A, delta, rhs)
    return sol[:n, 0], sol[n:, 0], sol[:n, 1], sol[n:, 1]


def exact_two_least_squares(A, delta, rhs1, rhs2):
    """K[p1;q1] = [rhs1;0], K[p2;q2] = [rhs2;0] (solve_linear_system.jl:189-203)."""
    m, n = A.shape
    rhs = np.zeros((n + m, 2))
    rhs[:n, 0] = rhs1
    rhs[:n, 1] = rhs2
    sol = _kkt_solve(A, delta, rhs)
    return sol[:n, 0], sol[n:, 0], sol[:n, 1], sol[n:, 1]


def exact_two_extras(A, delta, rhs1, rhs2):
    """(A A' + tau I)^-1 A rhs1 and (A A' + tau I)^-1 rhs2, tau = max(delta, 1e-14) (solve_linear_system.jl:51-72)."""
    import scipy.sparse as sp

    tau = max(delta, 1e-14)
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    M = Ad @ Ad.T + tau * np.eye(Ad.shape[0])
    return np.linalg.solve(M, Ad @ rhs1), np.linalg.solve(M, rhs2)


def exact_qp_hprod(qp, v, sigma, rho, delta, eta=0.0):
    """hprod! Val(2) on the eq-QP model through the exact KKT solve (model-Fletcherpenaltynlp.jl:521-570)."""
    A = qp.scipy_csr()
    p1, _, p2, _ = exact_two_least_squares(A, delta, v, qp.qdiag * v)
    ptv = v - p1
    Hv = p2 - qp.qdiag * ptv + 2.0 * sigma * ptv
    if rho > 0:
        Hv = Hv + rho * (A.T @ (A @ v))
    if eta > 0:
        Hv = Hv + eta * v
    return Hv


def exact_qp_objgrad(qp, x, sigma, rho, delta, eta=0.0, xk=None):
    """Closed-form penalty value/gradient on the eq-QP model through the exact KKT solve."""
    A = qp.scipy_csr()
    g = qp.qdiag * x + qp.d
    f = float(x @ (0.5 * qp.qdiag * x + qp.d))
    c = A @ x - qp.b
    p1, q1, p2, q2 = exact_two_mixed(A, delta, g, c)
    gs = p1 + sigma * p2
    ys = q1 + sigma * q2
    gx = gs - qp.qdiag * p2 + sigma * p2
    fx = f - c @ ys
    if rho > 0:
        gx = gx + rho * (A.T @ c)
        fx += rho / 2 * (c @ c)
    if eta > 0:
        dx = x - (np.zeros_like(x) if xk is None else xk)
        gx = gx + eta * dx
        fx += eta / 2 * (dx @ dx)
    return dict(fx=fx, gx=gx, ys=ys, gs=gs, p1=p1, q1=q1, p2=p2, q2=q2)
