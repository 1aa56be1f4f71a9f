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
