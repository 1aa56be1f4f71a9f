"""The host mirror of the reference interface on the CPU suite: FletcherPenaltyNLP (obj / grad! / hprod! / hess_coord!),
explicit_linear_constraints and fps_solve, driven through a QDSolver backed by the exact KKT oracle
(tests/oracle_qdsolver.py).  The same cases run through the HIP back-ends in the -m gpu suite."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import fps_amd  # noqa: E402,F401
from fps_amd import nlpmodels  # noqa: E402
from fps_amd.fps_solve import fps_solve  # noqa: E402
from fps_amd.penalty_nlp import FletcherPenaltyNLP  # noqa: E402
from oracle_qdsolver import OracleQDSolver  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))["cases"]


@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
def test_reference_known_answers_through_the_host_mirror(oracle, case):
    """test/unit-test.jl:16-152 through FletcherPenaltyNLP itself (the GPU suite does the same with the HIP back-ends)."""
    nlp = nlpmodels.SumSquares(case["n"]) if case["model"] == "sumsq" else nlpmodels.RosenbrockCircle()
    fp = FletcherPenaltyNLP(nlp, case["sigma"], case["rho"], case["delta"], 2, qds=OracleQDSolver(nlp, 0.0))
    x = np.array(case["x"])
    fx, gx = fp.objgrad(x)
    exp, tol = case["expect"], case["atol"]
    got = dict(obj=fx, grad=gx, ys=fp.ys, fx=fp.fx, gx=fp.gx, cx=fp.cx)
    for key, want in exp.items():  # whatever the reference test asserts for this case
        np.testing.assert_allclose(got[key], want, rtol=0, atol=max(tol[key], 1e-15), err_msg=key)


@pytest.mark.parametrize("ha", [1, 2])
@pytest.mark.parametrize("model", ["hs6", "circle"])
def test_hess_coord_is_the_matrix_of_hprod(oracle, model, ha):
    """hess_coord! (:439-519) and hprod! (:521-634) are the same Hessian approximation: H v == hprod(v).
    Reference quirk kept by the mirror: hprod! Val(2) adds `Hcv + rho JtJv` (:562) where hess_coord! and hprod! Val(1)
    add `rho (Hcv + JtJv)` (:486-489, :626) -- the two only agree for rho in {0, 1}, so Val(2) is checked at rho = 1."""
    nlp = nlpmodels.HS6() if model == "hs6" else nlpmodels.LinearPlusCircle(False)
    fp = FletcherPenaltyNLP(nlp, 10.0, 1.0 if ha == 2 else 0.5, 1e-3, ha, qds=OracleQDSolver(nlp, 0.0))
    n = nlp.meta.nvar
    rng = np.random.default_rng(0)
    x = nlp.meta.x0 + 0.1 * rng.standard_normal(n)
    vals = fp.hess_coord(x)
    rows, cols = fp.hess_structure()
    H = np.zeros((n, n))
    H[rows - 1, cols - 1] = vals
    H = H + np.tril(H, -1).T
    for _ in range(3):
        v = rng.standard_normal(n)
        np.testing.assert_allclose(H @ v, fp.hprod(x, v), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("sub,ha", [("lbfgs", 2), ("trunk", 2), ("trunk", 1)])
@pytest.mark.parametrize("model", ["sumsq", "hs6", "hs7"])
def test_fps_solve_end_to_end_on_the_exact_backend(oracle, model, sub, ha):
    """test/test-2.jl:28-97 acceptance: :first_order with residuals < 1e-6 max(||x0||, 1)."""
    nlp = {"sumsq": lambda: nlpmodels.SumSquares(10), "hs6": nlpmodels.HS6, "hs7": nlpmodels.HS7}[model]()
    stats = fps_solve(nlp, nlp.meta.x0, subproblem_solver=sub, hessian_approx=ha, qds=OracleQDSolver(nlp, 0.0))
    bound = 1e-6 * max(np.linalg.norm(nlp.meta.x0), 1.0)
    assert stats.status == "first_order" and stats.dual_feas < bound and stats.primal_feas < bound
    if model == "sumsq":
        assert np.linalg.norm(10 * stats.solution - 1.0) < 1e-6
    if model == "hs7":
        assert abs(stats.objective + np.sqrt(3.0)) < 1e-6


@pytest.mark.parametrize("ha", [1, 2])
def test_explicit_linear_constraints_on_the_exact_backend(oracle, ha):
    full, nl = nlpmodels.LinearPlusCircle(True), nlpmodels.LinearPlusCircle(False)
    fe = FletcherPenaltyNLP(full, 10.0, 0.5, 1e-3, ha, explicit_linear_constraints=True,
                            qds=OracleQDSolver(full, 0.0, explicit_linear_constraints=True))
    fn = FletcherPenaltyNLP(nl, 10.0, 0.5, 1e-3, ha, qds=OracleQDSolver(nl, 0.0))
    rng = np.random.default_rng(4)
    x, v = rng.standard_normal(3), rng.standard_normal(3)
    f1, g1 = fe.objgrad(x)
    f2, g2 = fn.objgrad(x)
    assert abs(f1 - f2) <= 1e-12 * max(1.0, abs(f2)) and np.allclose(g1, g2, rtol=1e-12, atol=1e-12)
    assert np.allclose(fe.hprod(x, v), fn.hprod(x, v), rtol=1e-10, atol=1e-12)
    assert fe.meta.ncon == 1 and np.allclose(fe.cons(x), [x.sum() - 1.0])


_REF_PROBLEMS = ["rosenbrock_sum", "hs8", "hs9", "hs26", "hs27", "huyer_neumaier", "estrin_a1", "flt", "hs61"]


@pytest.mark.parametrize("sub,ha", [("lbfgs", 2), ("trunk", 2), ("trunk", 1)])
@pytest.mark.parametrize("name", _REF_PROBLEMS)
def test_fps_solve_reference_integration_problems(oracle, name, sub, ha):
    """The remaining equality-constrained problems of test/test-2.jl:1-287 and test/rank-deficient.jl:22-36 (HS61) with
    the reference's acceptance: :first_order, primal and dual residuals < 1e-6 max(||x0||, 1).  Huyer-Neumaier starts at
    an infeasible stationary point and only passes through the feasibility restoration (algo.jl:200-209); HS27 / HS61
    with the Newton-CG sub-solver go through it too."""
    nlp = nlpmodels.reference_test_problems()[name]
    stats = fps_solve(nlp, nlp.meta.x0, subproblem_solver=sub, hessian_approx=ha, qds=OracleQDSolver(nlp, 0.0), max_time=60)
    bound = 1e-6 * max(np.linalg.norm(nlp.meta.x0), 1.0)
    assert stats.status == "first_order", (stats.status, stats.solver_specific)
    assert stats.primal_feas < bound and stats.dual_feas < bound
    if name == "rosenbrock_sum":
        assert np.linalg.norm(stats.solution - np.array([-1.612771347383541, 2.612771347383541])) < 1e-5  # test-2.jl:9
    if name == "estrin_a1":
        assert abs(stats.solution[0] - 1.0) < 1e-6  # the desirable solution, not the spurious minimum at -1.56


def test_admodel_derivatives_match_hand_written_models():
    """ADModel (torch.autograd) against the hand-derived HS6 / HS7 models: the stand-in for ADNLPModels.jl."""
    import torch

    hand = nlpmodels.HS7()
    ad = nlpmodels.ADModel(lambda x: torch.log(1 + x[0] ** 2) - x[1], [2.0, 2.0],
                           lambda x: [(1 + x[0] ** 2) ** 2 + x[1] ** 2 - 4.0], [0.0])
    rng = np.random.default_rng(0)
    for _ in range(3):
        x, v, g, y = rng.standard_normal(2), rng.standard_normal(2), rng.standard_normal(2), rng.standard_normal(1)
        assert abs(ad.obj(x) - hand.obj(x)) < 1e-14 and np.allclose(ad.grad(x), hand.grad(x), atol=1e-13)
        assert np.allclose(ad.cons(x), hand.cons(x), atol=1e-13) and np.allclose(ad.jac_coord(x), hand.jac_coord(x), atol=1e-12)
        assert np.allclose(ad.hprod(x, y, v), hand.hprod(x, y, v), atol=1e-12)
        assert np.allclose(ad.hprod(x, y, v, obj_weight=0.0), hand.hprod(x, y, v, obj_weight=0.0), atol=1e-12)
        assert np.allclose(ad.ghjvprod(x, g, v), hand.ghjvprod(x, g, v), atol=1e-12)


def test_feasibility_step_reaches_the_constraint_manifold():
    """feasibility_step (src/feasibility.jl:21-189) from the infeasible stationary point of the Huyer-Neumaier problem
    (c = x1^2 + x2^2 - 1 at x = 0: the Jacobian vanishes, only the aggressive second-order step can move) and from a
    generic point of HS61."""
    from fps_amd.fps_solve import feasibility_step

    P = nlpmodels.reference_test_problems()
    for name, x in (("hs61", np.array([1.0, 1.0, 1.0])), ("hs8", np.array([2.0, 1.0])), ("hs26", np.array([-2.6, 2.0, 2.0]))):
        nlp = P[name]
        z, ok = feasibility_step(nlp, x, nlp.cons(x) - nlp.meta.lcon, 1e-8, 1e-8)
        assert ok and np.linalg.norm(nlp.cons(z) - nlp.meta.lcon) <= 1e-8
    nlp = P["huyer_neumaier"]
    z, ok = feasibility_step(nlp, nlp.meta.x0, nlp.cons(nlp.meta.x0), 1e-8, 1e-8)
    assert not ok and np.allclose(z, 0.0)  # J = 0 and H = 2 c I = -2 I, J'c = 0: no direction -- the caller perturbs x


@pytest.mark.parametrize("name", ["bnd_eq", "inactive_bounds", "hs14", "hs71"])
def test_fps_solve_with_bounds_and_inequalities(oracle, name):
    """Bounds and inequality constraints (src/FletcherPenaltySolver.jl:139-143: inequalities through SlackModel, the bounds
    handled by the sub-problem solver; optimality on the projected residual, :42-43).  The penalty function of the
    reference knows only the equality multipliers, so with ACTIVE bounds it is not exact: the outer loop then converges at
    the rate sigma is increased (|c| ~ 1 / sigma), which is why these run at 1e-6; with inactive bounds the exactness, and
    the sqrt(eps) tolerance, is kept."""
    model, xstar, fstar = nlpmodels.bounded_test_problems()[name]
    inner = nlpmodels.SlackModel(model) if nlpmodels.has_inequalities(model) else model
    tol = np.sqrt(np.finfo(float).eps) if name == "inactive_bounds" else 1e-6
    st = fps_solve(model, qds=OracleQDSolver(inner, 0.0), atol=tol, rtol=tol, max_iter=200)
    assert st.status == "first_order" and st.solution.size == model.meta.nvar
    np.testing.assert_allclose(st.solution, xstar, rtol=0, atol=2e-5)
    if fstar is not None:
        assert abs(st.objective - fstar) <= 1e-4 * max(1.0, abs(fstar))
    lv, uv = model.meta.lvar, model.meta.uvar
    assert np.all(st.solution >= lv - 1e-12) and np.all(st.solution <= uv + 1e-12)
    c = model.cons(st.solution)
    assert np.all(c >= model.meta.lcon - 1e-5) and np.all(c <= model.meta.ucon + 1e-5)


def test_slack_model_and_bound_only_problems():
    """SlackModel: variables [x; s], c_i(x) - s_i = 0 with the constraint bounds moved to s; derivatives consistent.
    ncon = 0: fps_solve hands the model to the sub-problem solver (projected when it has bounds)."""
    model, _, _ = nlpmodels.bounded_test_problems()["hs71"]
    sm = nlpmodels.SlackModel(model)
    assert (sm.meta.nvar, sm.meta.ncon, sm.meta.nnzj) == (5, 2, 9) and not nlpmodels.has_inequalities(sm)
    assert sm.meta.lvar[4] == 25.0 and np.isinf(sm.meta.uvar[4]) and nlpmodels.has_bounds(sm)
    z = np.array([1.1, 4.0, 3.5, 1.4, 20.0])
    np.testing.assert_allclose(sm.cons(z), model.cons(z[:4]) - np.array([20.0, 0.0]))
    rows, cols = sm.jac_structure()
    J = np.zeros((2, 5))
    np.add.at(J, (rows - 1, cols - 1), sm.jac_coord(z))
    h = 1e-6
    for j in range(5):
        e = np.zeros(5)
        e[j] = h
        np.testing.assert_allclose((sm.cons(z + e) - sm.cons(z - e)) / (2 * h), J[:, j], rtol=0, atol=1e-6)
    box = nlpmodels.ADModel(lambda x: ((x - 2.0) ** 2).sum(), [0.0, 0.0, 0.0], lambda x: x[:0], [],
                            lvar=[-1.0, -1.0, 3.0], uvar=[1.0, 5.0, 4.0], name="box")
    st = fps_solve(box)
    assert st.status == "first_order"
    np.testing.assert_allclose(st.solution, [1.0, 2.0, 3.0], atol=1e-7)


def test_callback_and_restart_like_the_reference(oracle):
    """test/runtests.jl:11-34: a callback that records the iterates and stops the solve with `:user` (sigma_0 = 1,
    rho_0 = 0 on HS26; the reference stops at its 4th outer iteration -- with this mirror's sub-problem solver the problem
    is solved in fewer, so the stop is asked for after the first).  test/restart.jl: the feasibility problem mgh01feas (constant objective,
    -x1 = -1 and 10 (x2 - x1^2) = 0) from its x0 and again from x0 = (10, 10) at 1e-10."""
    nlp = nlpmodels.reference_test_problems()["hs26"]
    seen = []

    def cb(model, solver, stats):
        seen.append(np.array(stats.solution, float))
        if stats.iter == 1:
            stats.status = "user"

    st = fps_solve(nlp, qds=OracleQDSolver(nlp, 0.0), sigma_0=1.0, rho_0=0.0, callback=cb)
    assert st.status == "user" and st.iter == 1 and len(seen) == 2
    np.testing.assert_allclose(seen[0], nlp.meta.x0)
    free = fps_solve(nlp, qds=OracleQDSolver(nlp, 0.0), sigma_0=1.0, rho_0=0.0)
    assert free.status == "first_order" and free.iter >= 1
    mgh = nlpmodels.ADModel(lambda x: 0.0 * x[0], [-1.2, 1.0], lambda x: [-x[0], 10 * (x[1] - x[0] ** 2)], [-1.0, 0.0],
                            name="mgh01feas", lin=(0,))
    st = fps_solve(mgh, qds=OracleQDSolver(mgh, 0.0))
    assert st.status == "first_order"
    np.testing.assert_allclose(st.solution, [1.0, 1.0], atol=1e-6)
    st = fps_solve(mgh, np.array([10.0, 10.0]), qds=OracleQDSolver(mgh, 0.0), atol=1e-10, rtol=1e-10)
    assert st.status == "first_order"
    np.testing.assert_allclose(st.solution, [1.0, 1.0], atol=1e-6)
