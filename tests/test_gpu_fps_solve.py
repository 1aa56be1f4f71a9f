"""fps_solve through the HIP back-ends: the end-to-end cases of the reference's test/test-2.jl
(:28-52 sum of squares, :54-72 HS6, :74-97 HS7) with the reference's acceptance bounds:
status == :first_order, dual_feas and primal_feas < 1e-6 * max(||x0||, 1)."""
import numpy as np
import pytest

import fps_amd  # noqa: F401
from fps_amd import nlpmodels
from fps_amd.fps_solve import fps_solve

pytestmark = pytest.mark.gpu


def _accept(stats, x0):
    bound = 1e-6 * max(np.linalg.norm(x0), 1.0)
    assert stats.status == "first_order", (stats.status, stats.solver_specific)
    assert stats.dual_feas < bound and stats.primal_feas < bound


@pytest.mark.parametrize("qds", ["hip", "hip_direct"])
@pytest.mark.parametrize("sub,ha", [("lbfgs", 2), ("trunk", 2), ("trunk", 1)])
def test_sum_of_squares(qds, sub, ha):
    n = 10
    nlp = nlpmodels.SumSquares(n)
    stats = fps_solve(nlp, nlp.meta.x0, qds_solver=qds, subproblem_solver=sub, hessian_approx=ha)
    _accept(stats, nlp.meta.x0)
    assert np.linalg.norm(n * stats.solution - np.ones(n)) < 1e-6      # test-2.jl:37
    assert abs(stats.multipliers[0] + 2.0 / n) < 1e-6                   # grad f + J' lambda = 0: 2x + lambda = 0


@pytest.mark.parametrize("qds", ["hip", "hip_direct"])
@pytest.mark.parametrize("sub,ha", [("lbfgs", 2), ("trunk", 2), ("trunk", 1)])
def test_hs6(qds, sub, ha):
    nlp = nlpmodels.HS6()
    stats = fps_solve(nlp, nlp.meta.x0, qds_solver=qds, subproblem_solver=sub, hessian_approx=ha)
    _accept(stats, nlp.meta.x0)
    assert np.linalg.norm(stats.solution - np.array([1.0, 1.0])) < 1e-5 and abs(stats.objective) < 1e-9


@pytest.mark.parametrize("sub,ha", [("lbfgs", 2), ("trunk", 2)])
def test_hs7(sub, ha):
    nlp = nlpmodels.HS7()
    stats = fps_solve(nlp, nlp.meta.x0, qds_solver="hip_direct", subproblem_solver=sub, hessian_approx=ha)
    _accept(stats, nlp.meta.x0)
    assert abs(stats.objective + np.sqrt(3.0)) < 1e-6 and abs(stats.solution[0]) < 1e-4


@pytest.mark.parametrize("qds,sub,ha", [("hip_direct", "lbfgs", 2), ("hip_direct", "trunk", 1), ("hip", "lbfgs", 2)])
@pytest.mark.parametrize("name", ["rosenbrock_sum", "hs8", "hs9", "hs26", "hs27", "huyer_neumaier", "estrin_a1", "flt", "hs61"])
def test_reference_integration_problems(name, qds, sub, ha):
    """The other equality-constrained problems of test/test-2.jl:1-287 and HS61 of test/rank-deficient.jl:22-36 (rank-
    deficient Jacobian: the direct back-end goes through its dynamic pivot regularisation, the reference's
    ldlt_tol / ldlt_r2) through both MI355X back-ends, with the reference's acceptance bounds.  The models are ADModels
    (torch.autograd on the host, where the reference uses ADNLPModels.jl).  The reference runs these tests with its
    default (direct) back-end only; the iterative back-end is exercised with the first-order sub-solver.  Everything runs
    with the reference's DEFAULT regularisation ldlt_r2 = -sqrt(eps) (struct.jl:314) -- HS61 included -- except ONE case:
    FLT (Jacobian [2 x1 0; 3 x1^2 0]: rank <= 1 everywhere, 0 at the solution) with Newton-CG on the dense direct back-end is
    given the explicit option ldlt_r2 = "drop".  Why: sqrt(eps) on a pivot of M = A A' in natural order is not the
    reference's rule (-sqrt(eps) on pivots of K in AMD order, which on FLT amounts to the uniform shift M + sqrt(eps) I); it
    makes phi(x0) ~ 1e10 and the Newton-CG sub-solver crawls into max_time (measured with the default this round: 27 612
    gradient evaluations in 120 s, status max_time).  INTEGRATION.md says the same."""
    nlp = nlpmodels.reference_test_problems()[name]
    kw = {"ldlt_r2": "drop"} if (name, qds, sub, ha) == ("flt", "hip_direct", "trunk", 1) else {}
    stats = fps_solve(nlp, nlp.meta.x0, qds_solver=qds, subproblem_solver=sub, hessian_approx=ha, max_time=120, **kw)
    if kw:
        from fps_amd.qdsolver import REG_DROP
        assert stats.solver_specific["ldlt_r2"] == -REG_DROP
    elif qds == "hip_direct":
        assert stats.solver_specific["ldlt_r2"] == -float(np.sqrt(np.finfo(float).eps))   # the reference's default
    _accept(stats, nlp.meta.x0)
    if name == "estrin_a1":
        assert abs(stats.solution[0] - 1.0) < 1e-6


@pytest.mark.parametrize("qds", ["hip_direct", "hip"])
@pytest.mark.parametrize("name", ["bnd_eq", "inactive_bounds", "hs14", "hs71"])
def test_bounds_and_inequalities_through_the_hip_backends(name, qds):
    """Bounds and inequality constraints (SlackModel + the projected sub-problem solver, FletcherPenaltySolver.jl:139-143,
    :42-43) with the two systems solved on the MI355X: same acceptance as the CPU suite's exact-back-end run."""
    model, xstar, fstar = nlpmodels.bounded_test_problems()[name]
    tol = np.sqrt(np.finfo(float).eps) if name == "inactive_bounds" else 1e-6
    st = fps_solve(model, qds_solver=qds, atol=tol, rtol=tol, max_iter=200, max_time=120)
    assert st.status == "first_order" and st.solution.size == model.meta.nvar
    np.testing.assert_allclose(st.solution, xstar, rtol=0, atol=5e-5)
    if fstar is not None:
        assert abs(st.objective - fstar) <= 1e-4 * max(1.0, abs(fstar))


@pytest.mark.parametrize("backend", ["hip", "hip_direct"])
@pytest.mark.parametrize("ha", [1, 2])
def test_explicit_linear_constraints_penalise_only_the_nonlinear_ones(backend, ha):
    """`explicit_linear_constraints = true` (src/model-Fletcherpenaltynlp.jl:112-141): phi, grad phi and the Hessian
    products are those of the model WITHOUT its linear constraint, which stays a constraint of the penalised model."""
    from fps_amd.penalty_nlp import FletcherPenaltyNLP
    from fps_amd.qdsolver import qdsolver_correspondence

    Q = qdsolver_correspondence[backend]
    full, nl = nlpmodels.LinearPlusCircle(True), nlpmodels.LinearPlusCircle(False)
    tight = dict(ls_atol=1e-15, ls_rtol=1e-15, ln_atol=1e-15, ln_rtol=1e-15, ln_btol=1e-15, ln_conlim=0.0,
                 ls_axtol=1e-15, ls_btol=1e-15, ls_etol=1e-15, ne_atol=1e-15, ne_rtol=1e-15, ne_etol=1e-15)
    qe = Q(full, 0.0, explicit_linear_constraints=True, **tight)
    qn = Q(nl, 0.0, **tight)
    fe = FletcherPenaltyNLP(full, 10.0, 0.5, 1e-3, ha, explicit_linear_constraints=True, qds=qe)
    fn = FletcherPenaltyNLP(nl, 10.0, 0.5, 1e-3, ha, qds=qn)
    assert fe.meta.ncon == 1 and fn.meta.ncon == 0
    rng = np.random.default_rng(4)
    for _ in range(3):
        x, v = rng.standard_normal(3), rng.standard_normal(3)
        f1, g1 = fe.objgrad(x)
        f2, g2 = fn.objgrad(x)
        assert abs(f1 - f2) <= 1e-10 * max(1.0, abs(f2)) and np.allclose(g1, g2, rtol=1e-9, atol=1e-10)
        assert np.allclose(fe.hprod(x, v), fn.hprod(x, v), rtol=1e-7, atol=1e-8)
        assert np.allclose(fe.cons(x), [x.sum() - 1.0])
        assert np.allclose(fe.jprod(x, v), [v.sum()]) and np.allclose(fe.jtprod(x, np.array([2.0])), 2.0 * np.ones(3))
    qe.close()
    qn.close()


@pytest.mark.parametrize("sub", ["trunk", "lbfgs"])
def test_fps_solve_device_resident_qp_reaches_the_kkt_point(sub):
    """fps_solve_device: the outer loop on a DeviceEqQP (every iterate a tensor in HBM, obj/grad! = fpsq_qp_objgrad,
    Hessian products = fpsq_qp_hprod) must reach the solution of the equality QP's KKT system."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch

    from fps_amd import problems
    from fps_amd.device_qp import DeviceEqQP
    from fps_amd.fps_solve import fps_solve_device

    qp = problems.pde_control_like(n=3000, m=300, seed=5) if "seed" in problems.pde_control_like.__code__.co_varnames \
        else problems.pde_control_like(n=3000, m=300)
    A = qp.scipy_csr()
    K = sp.bmat([[sp.diags(qp.qdiag), A.T], [A, None]], format="csc")
    sol = spla.spsolve(K, np.concatenate([-qp.d, qp.b]))
    xstar, lam = sol[:qp.n], sol[qp.n:]
    # Krylov tolerances well below the outer tolerance (at the reference's sqrt(eps) defaults grad(phi) carries an error
    # of ~sigma * sqrt(eps) and the sub-problem cannot be driven to 1e-7)
    tight = dict(ls_atol=1e-13, ls_rtol=1e-13, ls_axtol=1e-13, ls_btol=1e-13, ls_etol=1e-13, ln_atol=1e-13,
                 ln_rtol=1e-13, ln_btol=1e-13, ln_conlim=0.0)
    dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, **tight)
    x0 = torch.from_numpy(qp.x).to(torch.device("cuda", 0))
    stats = fps_solve_device(dev, x0, subproblem_solver=sub, atol=1e-7, rtol=1e-7)
    assert stats.status == "first_order", (stats.status, stats.solver_specific)
    x = stats.solution.cpu().numpy()
    assert np.linalg.norm(x - xstar) <= 1e-5 * np.linalg.norm(xstar)
    assert np.linalg.norm(stats.multipliers.cpu().numpy() - lam) <= 1e-4 * max(1.0, np.linalg.norm(lam))
    assert abs(stats.objective - (0.5 * xstar @ (qp.qdiag * xstar) + qp.d @ xstar)) <= 1e-6 * max(1.0, abs(stats.objective))
    dev.close()


@pytest.mark.parametrize("sub,ha", [("trunk", 2), ("lbfgs", 2)])
def test_fps_solve_through_the_banded_direct_backend(sub, ha):
    """fps_solve with qds_solver = "hip_ldlt" (the reference's DEFAULT is its LDLt back-end, parameters.jl:290): the outer loop
    on a host eq-QP model, every pair of systems through fpsq_band_* (sparse block-banded factorisation on the MI355X),
    must reach the solution of the QP's KKT system at the reference's sqrt(eps) tolerances."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from fps_amd import problems

    qp = problems.pde_control_like(n=4000, m=400, per_row=16, window=512, seed=21)
    A = qp.scipy_csr()
    K = sp.bmat([[sp.diags(qp.qdiag), A.T], [A, None]], format="csc")
    sol = spla.spsolve(K, np.concatenate([-qp.d, qp.b]))
    xstar, lam = sol[:qp.n], sol[qp.n:]
    model = nlpmodels.EqQPModel(qp)
    stats = fps_solve(model, qp.x, qds_solver="hip_ldlt", subproblem_solver=sub, hessian_approx=ha, max_time=120)
    assert stats.status == "first_order", (stats.status, stats.solver_specific)
    assert np.linalg.norm(stats.solution - xstar) <= 1e-6 * np.linalg.norm(xstar)
    assert np.linalg.norm(stats.multipliers - lam) <= 1e-5 * max(1.0, np.linalg.norm(lam))


def test_default_backend_is_ldlt_and_goes_direct_on_small_models():
    """fps_solve without `qds_solver`: the reference's default key "ldlt" (parameters.jl:290) with the reference's
    ldlt_r2 = -sqrt(eps) (struct.jl:314) -- the banded direct back-end on every small model (exact: the reference's
    acceptance bounds AND its LDLt-level accuracy), the iterative one where A A' has no narrow band; the statistics say which."""
    from fps_amd import problems
    from fps_amd.qdsolver import HIPBandedDirectQDSolver, HIPQDSolver, qdsolver_correspondence

    nlp = nlpmodels.HS6()
    stats = fps_solve(nlp, nlp.meta.x0)
    _accept(stats, nlp.meta.x0)
    assert np.linalg.norm(stats.solution - np.array([1.0, 1.0])) < 1e-5
    assert stats.solver_specific["qds_solver"] == "ldlt" and stats.solver_specific["qds_backend"] == "hip_ldlt"
    assert stats.solver_specific["ldlt_r2"] == -float(np.sqrt(np.finfo(float).eps))
    q = qdsolver_correspondence["ldlt"](nlp, 0.0)
    assert isinstance(q, HIPBandedDirectQDSolver) and q.qds_backend == "hip_ldlt"
    q.close()
    q = qdsolver_correspondence["auto"](nlpmodels.EqQPModel(problems.aug2dc_like(N=40)), 0.0)   # cfg4's family
    assert isinstance(q, HIPBandedDirectQDSolver) and q.info()["bandwidth_blocks"] <= 4
    q.close()
    q = qdsolver_correspondence["ldlt"](nlpmodels.EqQPModel(problems.random_eqqp(n=20000, m=2000)), 0.0)  # cfg2's family
    assert isinstance(q, HIPQDSolver) and q.qds_backend == "hip"
    q.close()


def test_config2_pattern_through_the_direct_backend_at_reduced_size(oracle):
    """BASELINE configs[1]'s Jacobian (100 random columns per row: A A' is DENSE) through "hip_ldlt", the reference's default
    route: the band is full (every block of M is stored and factored: O(m^3) work -- the cliff `auto` steers around), the
    answers are exact.  m = 800 here (the oracle's dense KKT solve stays cheap); the full-size row is in
    profiles/r03_configs.md: 79 blocks, factorise 20.7 ms + solve 1.8 ms against 0.5 ms per evaluation on the iterative path."""
    from fps_amd import problems
    from fps_amd.penalty_nlp import FletcherPenaltyNLP
    from fps_amd.qdsolver import HIPBandedDirectQDSolver

    qp = problems.random_eqqp(n=2100, m=800)
    model = nlpmodels.EqQPModel(qp)
    qds = HIPBandedDirectQDSolver(model, 0.0)
    i = qds.info()
    assert i["bandwidth_blocks"] == i["nblocks"] - 1 == 6
    se = float(np.sqrt(np.finfo(float).eps))
    for delta in (0.0, se):
        fp = FletcherPenaltyNLP(model, sigma=1e3, rho=1.0, delta=delta, hessian_approx=2, qds=qds)
        x = qp.point(3)
        g, c = model.grad(x), model.cons(x)
        got = qds.solve_two_mixed(fp, x, g, c)
        want = oracle.exact_two_mixed(qp.scipy_csr(), delta, g, c)
        for a, b in zip(got, want):
            assert np.max(np.abs(a - b)) <= 1e-9 * max(np.max(np.abs(b)), 1e-300)
        f, gx = fp.objgrad(x)
        ref = oracle.exact_qp_objgrad(qp, x, 1e3, 1.0, delta)
        assert abs(f - ref["fx"]) <= 1e-9 * abs(ref["fx"]) and np.max(np.abs(gx - ref["gx"])) <= 1e-8 * np.max(np.abs(ref["gx"]))
    qds.close()
