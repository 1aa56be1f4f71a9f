"""Pins the CPU oracle (oracle/) before anything is compared against it.

1. against the reference's own known-answer tests (tests/golden/reference_known_answers.json, from
   /root/reference/test/unit-test.jl) -- both the exact KKT oracle and the C restatement of the iterative back-end;
2. the C restatement's LSQR / CRAIG / MINRES against independent implementations (scipy) and the exact solve.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import fps_amd  # noqa: F401
from fps_amd import problems

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))["cases"]
SE = np.sqrt(np.finfo(float).eps)


def _case_matrix(case):
    A = sp.csr_matrix((case["jac_vals"], (case["jac_rows"], case["jac_cols"])), shape=(case["m"], case["n"]))
    A.sort_indices()
    return A


def _hprod(case, y, v, obj_weight):
    """(obj_weight * Hess f + sum y_i Hess c_i) v for the two golden models (what ADNLPModel's hprod! returns)."""
    x = np.array(case["x"])
    v = np.asarray(v)
    if case["model"] == "sumsq":
        return obj_weight * 2.0 * v
    x1, x2 = x
    Hf = np.array([[2 - 400 * (x2 - x1 ** 2) + 800 * x1 ** 2, -400 * x1], [-400 * x1, 200.0]])
    return obj_weight * (Hf @ v) + y[0] * 2.0 * v


def _penalty_from_solution(case, p1, q1, p2, q2):
    """obj / grad! epilogue, model-Fletcherpenaltynlp.jl:244-248, 364, 382-397."""
    sigma, rho = case["sigma"], case["rho"]
    g, c, f = np.array(case["g"]), np.array(case["c"]), case["f"]
    A = _case_matrix(case)
    gs, ys, v, w = p1 + sigma * p2, q1 + sigma * q2, p2, q2
    obj = f - c @ ys + rho / 2 * (c @ c)
    Hsv = _hprod(case, ys, v, 1.0)
    Sstw = _hprod(case, w, gs, 0.0)
    grad = gs - Hsv + sigma * v + Sstw
    if rho > 0:
        grad = grad + rho * (A.T @ c)
    return dict(obj=obj, fx=f, gx=g, ys=ys, cx=c, grad=grad)


@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
@pytest.mark.parametrize("backend", ["exact", "c_iterative"])
def test_oracle_matches_reference_known_answers(oracle, case, backend):
    A = _case_matrix(case)
    g, c = np.array(case["g"]), np.array(case["c"])
    if backend == "exact":
        p1, q1, p2, q2 = oracle.exact_two_mixed(A, case["delta"], g, c)
        slack = 1.0
    else:
        # tolerances far below the reference's sqrt(eps) defaults so the iterative restatement can be held to the
        # LDLt-level assertions of the reference's tests
        o = oracle.default_options(case["n"], case["m"], ls_atol=1e-15, ls_rtol=1e-15, ln_atol=1e-15, ln_rtol=1e-15,
                                   ln_btol=1e-15, ls_axtol=1e-15, ls_btol=1e-15, ls_etol=1e-15)
        p1, q1, p2, q2, st, rc = oracle.solve_two_mixed(case["m"], case["n"], A.indptr, A.indices, A.data,
                                                          case["delta"], g, c, o)
        assert rc == 0 and st[0].solved and st[1].solved
        slack = 4.0
    got = _penalty_from_solution(case, p1, q1, p2, q2)
    for key, want in case["expect"].items():
        atol = max(case["atol"][key], 1e-15) * slack
        np.testing.assert_allclose(got[key], want, rtol=0, atol=atol, err_msg=f"{case['name']}:{key}")


def _rand_problem(m, n, density, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, random_state=rng, format="csr", data_rvs=lambda k: rng.uniform(-1, 1, k))
    A = A + sp.eye(m, n, format="csr") * 3.0
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A, rng


@pytest.mark.parametrize("damp", [0.0, 0.3])
def test_c_lsqr_matches_scipy_lsqr(oracle, damp):
    A, rng = _rand_problem(40, 120, 0.1, 1)
    b = rng.standard_normal(120)
    # LSQR on the operator A' (120 x 40), as the reference calls it (solve_linear_system.jl:123)
    x, st = oracle.lsqr(40, 120, A.indptr, A.indices, A.data, b, lam=damp, atol=1e-14, rtol=1e-14, transposed=True,
                        axtol=1e-14, btol=1e-14, etol=1e-14, conlim=1e14)
    ref = spla.lsqr(A.T.tocsr(), b, damp=damp, atol=1e-15, btol=1e-15, conlim=1e12, iter_lim=5000)[0]
    assert st.solved
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-10 * np.linalg.norm(ref))
    # normal equations residual: (A A' + damp^2 I) x = A b
    M = (A @ A.T).toarray() + damp ** 2 * np.eye(40)
    np.testing.assert_allclose(M @ x, A @ b, rtol=0, atol=1e-9 * np.linalg.norm(A @ b))


def test_c_lsqr_iterates_track_scipy(oracle):
    """Same Golub-Kahan recurrences => the k-th iterate agrees with scipy's k-th iterate."""
    A, rng = _rand_problem(30, 90, 0.15, 2)
    b = rng.standard_normal(90)
    for k in (1, 3, 7):
        x, st = oracle.lsqr(30, 90, A.indptr, A.indices, A.data, b, lam=0.1, atol=0, rtol=0, itmax=k, transposed=True,
                            axtol=0, btol=0, etol=0, conlim=0)
        ref = spla.lsqr(A.T.tocsr(), b, damp=0.1, atol=0, btol=0, conlim=0, iter_lim=k)[0]
        assert st.niter == k
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-12 * max(1.0, np.linalg.norm(ref)))


@pytest.mark.parametrize("delta", [0.0, 0.25, 1e-8])
def test_c_craig_matches_exact(oracle, delta):
    A, rng = _rand_problem(35, 100, 0.12, 3)
    c = rng.standard_normal(35)
    x, y, st = oracle.craig(35, 100, A.indptr, A.indices, A.data, -c, delta=delta, atol=1e-15, rtol=1e-15, btol=1e-15,
                            conlim=0.0)  # conlim=0 disables the condition-number stop (it fires first for tiny delta)
    assert st.solved and not st.inconsistent
    _, _, p2, q2 = oracle.exact_two_mixed(A, delta, np.zeros(100), c)
    # solve_linear_system.jl:132-133: p2 = -x, q2 = y
    np.testing.assert_allclose(-x, p2, rtol=0, atol=1e-10 * np.linalg.norm(p2))
    np.testing.assert_allclose(y, q2, rtol=0, atol=1e-10 * np.linalg.norm(q2))


@pytest.mark.parametrize("lam", [1e-14, 0.25])
def test_c_minres_matches_exact(oracle, lam):
    A, rng = _rand_problem(30, 80, 0.15, 4)
    b = rng.standard_normal(30)
    x, st = oracle.minres_aat(30, 80, A.indptr, A.indices, A.data, b, lam=lam, atol=1e-14, rtol=1e-14, etol=1e-14)
    assert st.solved
    M = (A @ A.T).toarray() + lam * np.eye(30)
    ref = np.linalg.solve(M, b)
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-9 * np.linalg.norm(ref))


def test_c_two_systems_default_tolerances_vs_exact(oracle):
    """SURVEY.md §7 parity definition: at the reference's default sqrt(eps) tolerances the iterative path is within
    1e-6 (relative, inf-norm) of the direct solve."""
    qp = problems.pde_control_like(n=4000, m=400, per_row=20, window=512, seed=7)
    A = qp.scipy_csr()
    g = qp.qdiag * qp.x + qp.d
    c = A @ qp.x - qp.b
    for delta in (0.0, SE):
        p1, q1, p2, q2, st, rc = oracle.solve_two_mixed(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, delta, g, c)
        assert rc == 0
        e = oracle.exact_two_mixed(A, delta, g, c)
        for got, want in zip((p1, q1, p2, q2), e):
            assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))
        assert 0 < st[0].niter < 200 and 0 < st[1].niter < 200


def test_c_two_least_squares_and_extras_vs_exact(oracle):
    qp = problems.random_eqqp(n=600, m=60, per_row=12, seed=5)
    A = qp.scipy_csr()
    rng = np.random.default_rng(0)
    r1, r2, r3 = rng.standard_normal(qp.n), rng.standard_normal(qp.n), rng.standard_normal(qp.m)
    tight = dict(ls_atol=1e-14, ls_rtol=1e-14, ls_axtol=1e-14, ls_btol=1e-14, ls_etol=1e-14,
                 ne_atol=1e-14, ne_rtol=1e-14, ne_etol=1e-14)
    o = oracle.default_options(qp.n, qp.m, **tight)
    p1, q1, p2, q2, st, rc = oracle.solve_two_least_squares(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, 0.01, r1, r2, o)
    e = oracle.exact_two_least_squares(A, 0.01, r1, r2)
    for got, want in zip((p1, q1, p2, q2), e):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * np.linalg.norm(want))
    a, b, st, rc = oracle.solve_two_extras(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, 0.01, r1, r3, o)
    ea, eb = oracle.exact_two_extras(A, 0.01, r1, r3)
    np.testing.assert_allclose(a, ea, rtol=0, atol=1e-9 * np.linalg.norm(ea))
    np.testing.assert_allclose(b, eb, rtol=0, atol=1e-9 * np.linalg.norm(eb))


def test_c_qp_objgrad_vs_exact(oracle):
    qp = problems.pde_control_like(n=3000, m=300, per_row=16, window=256, seed=11)
    o = oracle.default_options(qp.n, qp.m, ls_atol=1e-13, ls_rtol=1e-13, ls_axtol=1e-13, ls_btol=1e-13, ls_etol=1e-13,
                               ln_atol=1e-13, ln_rtol=1e-13, ln_btol=1e-13, ln_conlim=0.0)
    xk = qp.xhat
    got = oracle.qp_objgrad(qp, qp.x, sigma=1e3, rho=1.0, delta=SE, eta=0.5, xk=xk, opts=o)
    want = oracle.exact_qp_objgrad(qp, qp.x, sigma=1e3, rho=1.0, delta=SE, eta=0.5, xk=xk)
    assert got["rc"] == 0
    np.testing.assert_allclose(got["ys"], want["ys"], rtol=0, atol=1e-8 * np.linalg.norm(want["ys"]))
    np.testing.assert_allclose(got["gx"], want["gx"], rtol=0, atol=1e-8 * np.linalg.norm(want["gx"]))
    assert abs(got["fx"] - want["fx"]) <= 1e-8 * abs(want["fx"])


def test_lsqr_zero_rhs_and_zero_atb_edge_cases(oracle):
    A, rng = _rand_problem(5, 12, 0.3, 9)
    x, st = oracle.lsqr(5, 12, A.indptr, A.indices, A.data, np.zeros(12), transposed=True)
    assert st.solved and st.niter == 0 and np.all(x == 0)
    # b orthogonal to range(A') => A b = 0 => x = 0 is the least-squares solution
    Ad = A.toarray()
    b = rng.standard_normal(12)
    b -= Ad.T @ np.linalg.lstsq(Ad.T, b, rcond=None)[0]
    x, st = oracle.lsqr(5, 12, A.indptr, A.indices, A.data, b, atol=SE, rtol=SE, transposed=True)
    assert st.solved and np.linalg.norm(x) < 1e-12


@pytest.mark.parametrize("delta", [0.0, SE, 0.25])
def test_c_lnlq_least_norm_vs_exact(oracle, delta):
    """LNLQ as the reference's GENERIC solve_least_norm calls it (struct.jl:251-281: M = (1/delta) I without sqd): M
    only preconditions, so for every delta the answer is the minimum-norm solution of A x = b with x = A'y."""
    qp = problems.pde_control_like(n=400, m=60, per_row=20, window=128, seed=77)
    A = qp.scipy_csr()
    Ad = A.toarray()
    b = -(A @ qp.x - qp.b)
    ye = np.linalg.solve(Ad @ Ad.T, b)
    xe = Ad.T @ ye
    x, y, st = oracle.lnlq(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, b, delta=delta)
    assert st.solved and st.niter > 10 and st.status in (3, 9)
    assert np.linalg.norm(x - xe) <= 1e-7 * np.linalg.norm(xe) and np.linalg.norm(y - ye) <= 1e-7 * np.linalg.norm(ye)
    x, y, st = oracle.lnlq(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, b, delta=delta, atol=1e-14, rtol=1e-14)
    assert st.solved
    assert np.linalg.norm(x - xe) <= 1e-13 * np.linalg.norm(xe) and np.linalg.norm(y - ye) <= 1e-13 * np.linalg.norm(ye)
    # zero right-hand side and the iteration cap
    x, y, st = oracle.lnlq(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, 0.0 * b, delta=delta)
    assert st.solved and st.niter == 0 and st.status == 1 and not x.any() and not y.any()
    x, y, st = oracle.lnlq(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, b, delta=delta, itmax=3)
    assert not st.solved and st.status == 7 and st.niter == 4  # lnlq! counts one more than the passes it made
    # through solve_two_mixed with the method selector
    o = oracle.default_options(qp.n, qp.m, ln_method=1)
    g = qp.qdiag * qp.x + qp.d
    p1, q1, p2, q2, stats, rc = oracle.solve_two_mixed(qp.m, qp.n, qp.rowptr, qp.colind, qp.vals, delta, g, -b, o)
    assert rc == 0 and np.linalg.norm(p2 + xe) <= 1e-7 * np.linalg.norm(xe)  # p2 = -x  (linear_system.jl:133)


# ---------------------------------------------------------------- the iterative back-end's known-answer case

_KCASE = os.path.join(os.path.dirname(__file__), "golden", "krylov_case_small_pde.json")
_KGOLD = os.path.join(os.path.dirname(__file__), "golden", "krylov_golden.json")


def _krylov_case():
    case = json.load(open(_KCASE))
    A = sp.csr_matrix((case["vals"], (np.array(case["rows"]) - 1, np.array(case["cols"]) - 1)),
                      shape=(case["m"], case["n"]))
    A.sort_indices()
    return case, A, np.array(case["g"]), np.array(case["c"])


def test_krylov_case_fixture_is_what_its_generator_writes():
    """tests/golden/krylov_case_small_pde.json (the input of make_krylov_golden.jl) is reproducible bit for bit from the
    committed generator, and Krylov iterates well past one step on it (unlike the m = 1 known-answer cases)."""
    case, A, g, c = _krylov_case()
    qp = problems.pde_control_like(n=400, m=60, per_row=20, window=128, seed=77)
    assert (case["n"], case["m"]) == (qp.n, qp.m)
    B = qp.scipy_csr()
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    assert np.array_equal(g, qp.qdiag * qp.x + qp.d) and np.array_equal(c, B @ qp.x - qp.b)


@pytest.mark.parametrize("delta", [0.0, SE, 0.25])
def test_oracle_iterates_on_krylov_case(oracle, delta):
    """The C restatement on the Krylov known-answer case at the reference's default tolerances: > 10 iterations of each
    method, and the answers sit where those tolerances put them relative to the exact KKT solve."""
    case, A, g, c = _krylov_case()
    m, n = A.shape
    p1, q1, p2, q2, st, rc = oracle.solve_two_mixed(m, n, A.indptr, A.indices, A.data, delta, g, c)
    assert st[0].niter > 10 and st[1].niter > 10
    e = oracle.exact_two_mixed(A, delta, g, c)
    assert np.linalg.norm(q1 - e[1]) <= 1e-6 * np.linalg.norm(e[1])
    if st[1].solved:
        assert np.linalg.norm(q2 - e[3]) <= 1e-6 * np.linalg.norm(e[3])


def _status_code(s):
    """Krylov.jl stats.status string -> FPO_ST_* code (substring match)."""
    s = s.lower()
    if s.startswith("x = 0 is a zero-residual") or s.startswith("x is a zero-residual"):
        return 1
    if s.startswith("x = 0 is a minimum least-squares"):
        return 2
    if "zero-residual" in s:
        return 4
    if "forward error" in s:
        return 5
    if "condition number" in s:
        return 6
    if "maximum number of iterations" in s:
        return 7
    if "inconsistent" in s:
        return 8
    if "least-squares solution" in s or "good enough" in s or "solution" in s:
        return 3
    return 0


@pytest.mark.skipif(not os.path.exists(_KGOLD), reason="tests/golden/krylov_golden.json absent: it is written by "
                    "tests/golden/make_krylov_golden.jl, which needs Julia + Krylov.jl 0.10 (not in this pipeline) -- "
                    "iteration-level parity with Krylov.jl stays UNPINNED until someone runs it")
def test_oracle_matches_krylov_jl_golden(oracle):
    """Pins oracle/fps_oracle.c against the real Krylov.jl: niter, solved, status and the solution vectors of lsqr,
    craig (sqd for delta != 0) and minres with the reference's keyword arguments."""
    gold = json.load(open(_KGOLD))
    case, A, g, c = _krylov_case()
    m, n = A.shape
    for run in gold["runs"]:
        delta = run["delta"]
        x, st = oracle.lsqr(m, n, A.indptr, A.indices, A.data, g, lam=np.sqrt(delta), atol=SE, rtol=SE,
                            itmax=5 * (m + n), transposed=True)
        w = run["lsqr"]
        assert (st.niter, bool(st.solved)) == (w["stats"]["niter"], w["stats"]["solved"])
        assert st.status == _status_code(w["stats"]["status"])
        np.testing.assert_allclose(x, w["x"], rtol=0, atol=1e-12 * np.linalg.norm(w["x"]))
        xc, yc, st = oracle.craig(m, n, A.indptr, A.indices, A.data, -c, delta=delta, itmax=5 * (m + n))
        w = run["craig"]
        assert (st.niter, bool(st.solved)) == (w["stats"]["niter"], w["stats"]["solved"])
        assert st.status == _status_code(w["stats"]["status"])
        np.testing.assert_allclose(xc, w["x"], rtol=0, atol=1e-12 * np.linalg.norm(w["x"]))
        np.testing.assert_allclose(yc, w["y"], rtol=0, atol=1e-12 * np.linalg.norm(w["y"]))
        if "lnlq" in run:
            xl, yl, st = oracle.lnlq(m, n, A.indptr, A.indices, A.data, -c, delta=delta, itmax=5 * (m + n))
            w = run["lnlq"]
            assert (st.niter, bool(st.solved)) == (w["stats"]["niter"], w["stats"]["solved"])
            np.testing.assert_allclose(xl, w["x"], rtol=0, atol=1e-12 * np.linalg.norm(w["x"]))
            np.testing.assert_allclose(yl, w["y"], rtol=0, atol=1e-12 * np.linalg.norm(w["y"]))
        xm, st = oracle.minres_aat(m, n, A.indptr, A.indices, A.data, c, lam=max(delta, 1e-14))
        w = run["minres"]
        assert (st.niter, bool(st.solved)) == (w["stats"]["niter"], w["stats"]["solved"])
        np.testing.assert_allclose(xm, w["x"], rtol=0, atol=1e-12 * np.linalg.norm(w["x"]))


def test_minres_on_k_restatement_matches_exact_kkt(oracle):
    """fpo_minres_kkt (the checker of the library's kkt_method = FPSQ_KKT_MINRES_K: the MINRES restatement applied to
    K = [I A'; A -delta I] instead of A A' + lambda I): both saddle-point systems against the exact KKT solve -- 1e-6 at
    the reference's sqrt(eps) tolerances, 1e-11 at tight ones -- and the refactored A A' variant against a direct solve."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from fps_amd import problems

    qp = problems.pde_control_like(n=3000, m=300, per_row=20, window=256, seed=5)
    A = qp.scipy_csr()
    rng = np.random.default_rng(0)
    g, c = rng.standard_normal(3000), rng.standard_normal(300)
    for delta in (0.0, 1e-2):
        ex = oracle.exact_two_mixed(A, delta, g, c)
        p, q, st = oracle.minres_kkt(300, 3000, A.indptr, A.indices, A.data, delta, bp=g)
        assert st.solved == 1 and 20 < st.niter < 100
        assert np.linalg.norm(p - ex[0]) <= 1e-6 * np.linalg.norm(ex[0]) and np.linalg.norm(q - ex[1]) <= 1e-6 * np.linalg.norm(ex[1])
        p, q, st = oracle.minres_kkt(300, 3000, A.indptr, A.indices, A.data, delta, bq=c, atol=1e-13, rtol=1e-13, etol=1e-15)
        assert np.linalg.norm(p - ex[2]) <= 1e-11 * np.linalg.norm(ex[2]) and np.linalg.norm(q - ex[3]) <= 1e-11 * np.linalg.norm(ex[3])
    x, st = oracle.minres_aat(300, 3000, A.indptr, A.indices, A.data, c, lam=1e-3)
    M = (A @ A.T + 1e-3 * sp.identity(300)).tocsc()
    assert st.solved == 1 and np.linalg.norm(x - spla.spsolve(M, c)) <= 1e-6 * np.linalg.norm(x)
    p, q, st = oracle.minres_kkt(300, 3000, A.indptr, A.indices, A.data, 0.0)  # zero right-hand side
    assert st.niter == 0 and st.solved == 1 and not p.any() and not q.any()


# ------------------------------------------------------------------------------- summation order (fpo_set_sum_order)

def _counts(oracle, kind, delta, mode):
    from structures import random_structure

    rng = np.random.default_rng(12)
    A = random_structure(kind, rng)
    m, n = A.shape
    rng.standard_normal(n), rng.standard_normal(m)  # (the same draws as the GPU test: tests/test_gpu_parity._awkward_case)
    g, c = rng.standard_normal(n), rng.standard_normal(m)
    r1, r2 = rng.standard_normal(n), rng.standard_normal(n)
    rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data)
    oracle.set_sum_order(mode)
    try:
        a = oracle.solve_two_mixed(m, n, rp, ci, va, delta, g, c)
        b = oracle.solve_two_least_squares(m, n, rp, ci, va, delta, r1, r2)
    finally:
        oracle.set_sum_order(0)
    return (a[4][0].niter, a[4][1].niter, b[4][0].niter, b[4][1].niter), (*a[:4], *b[:4])


def test_summation_order_variants_compute_the_same_products(oracle):
    """fpo_set_sum_order only re-associates: A x and A' u of every variant agree with scipy to rounding, on a matrix with a
    row and a column long enough to take the device-order branches (> 2048 entries)."""
    rng = np.random.default_rng(5)
    A = sp.random(2600, 3000, density=0.002, random_state=np.random.RandomState(2), format="lil")
    A[7, :] = rng.standard_normal(3000)
    A[:, 11] = rng.standard_normal((2600, 1))
    A = sp.csr_matrix(A)
    A.sort_indices()
    x, u = rng.standard_normal(3000), rng.standard_normal(2600)
    rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data)
    for mode in (0, 1, 2):
        oracle.set_sum_order(mode)
        try:
            y = oracle.spmv(2600, 3000, rp, ci, va, x)
            z = oracle.spmv(2600, 3000, rp, ci, va, u, transposed=True)
        finally:
            oracle.set_sum_order(0)
        assert np.max(np.abs(y - A @ x)) <= 1e-12 * np.max(np.abs(A @ x))
        assert np.max(np.abs(z - A.T @ u)) <= 1e-12 * np.max(np.abs(A.T @ u))


@pytest.mark.parametrize("delta", [SE, 0.25])
def test_iteration_counts_depend_on_the_summation_order_only_for_a_dominant_row_or_column(oracle, delta):
    """The MEASUREMENT behind the tolerance of the GPU test of the two order-sensitive structures: the restatement, run with
    its products summed (0) left to right, (1) long rows in the device's order, (2) right to left -- three mathematically
    identical programs -- stops at the SAME iteration on the four well-conditioned awkward structures (so the GPU test
    demands equal counts there; their VECTORS still move with the order, by 1e-16 ... 1e-6 depending on how many iterations
    ran: the GPU test measures that distance per vector instead of assuming a tolerance), and up to two iterations apart on
    the dense-row / dense-column ones (the third order-sensitive structure, empty columns, keeps its count under these three
    orders and loses one iteration under the device's: profiles/r04_fixed_iteration_probe.txt)."""
    from structures import ORDER_SENSITIVE, WELL_CONDITIONED

    for kind in WELL_CONDITIONED:
        ref, vref = _counts(oracle, kind, delta, 0)
        for mode in (1, 2):
            its, v = _counts(oracle, kind, delta, mode)
            assert its == ref, (kind, mode, its, ref)
            assert max(np.max(np.abs(a - b)) / np.max(np.abs(b)) for a, b in zip(v, vref)) < 1e-4
    moved = 0
    for kind in ORDER_SENSITIVE:
        ref, _ = _counts(oracle, kind, delta, 0)
        for mode in (1, 2):
            its, _ = _counts(oracle, kind, delta, mode)
            assert all(abs(a - b) <= 2 for a, b in zip(its, ref)), (kind, mode, its, ref)
            moved += its != ref
    assert moved >= 2  # (the sensitivity is real: were it gone, the GPU test could demand equal counts there too)
