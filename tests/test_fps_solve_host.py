"""Host logic of fps_solve that needs no GPU: the two built-in sub-problem solvers on a plain unconstrained function,
and the parameter schedule (src/parameters.jl:69-94, src/algo.jl:361-390)."""
import numpy as np

import fps_amd  # noqa: F401
from fps_amd import fps_solve as F


class _Rosenbrock:
    """objgrad_/hprod_ with the calling convention of FletcherPenaltyNLP."""

    def objgrad_(self, x, g):
        g[0] = 2 * (x[0] - 1) - 400 * x[0] * (x[1] - x[0] ** 2)
        g[1] = 200 * (x[1] - x[0] ** 2)
        return (x[0] - 1) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2, g

    def hprod_(self, x, v, Hv):
        H = np.array([[2 - 400 * (x[1] - 3 * x[0] ** 2), -400 * x[0]], [-400 * x[0], 200.0]])
        Hv[:] = H @ v
        return Hv


def test_subsolvers_minimise_rosenbrock():
    for name in ("lbfgs", "trunk"):
        x, status, g = F._SUBSOLVERS[name](_Rosenbrock(), np.array([-1.2, 1.0]), 1e-9, 0.0, 5000, 1e8)
        assert status == "optimal", (name, status)
        assert np.linalg.norm(x - 1.0) < 1e-6 and np.linalg.norm(g, np.inf) <= 1e-9


def test_parameter_schedule_defaults_and_updates():
    meta = F.AlgoData()
    se = np.sqrt(np.finfo(float).eps)
    assert (meta.sigma_0, meta.rho_0, meta.sigma_update, meta.rho_update) == (1e3, 1.0, 2.0, 2.0)
    assert meta.delta_0 == se and meta.delta_update == 10.0 and meta.sigma_max == 1 / se

    class _FP:
        sigma, rho, delta, stale = 1e3, 1.0, 0.0, False

        def invalidate(self):
            self.stale = True

    fp = _FP()
    F._update_parameters(fp, meta, feas=True)      # feasible iterate: sigma only (algo.jl:364-367)
    assert (fp.sigma, fp.rho, fp.stale) == (2e3, 1.0, True)
    F._update_parameters(fp, meta, feas=False)
    assert (fp.sigma, fp.rho) == (4e3, 2.0)


def test_nonlinear_constraints_view_is_the_model_without_its_linear_rows():
    """explicit_linear_constraints: the penalty function sees cons_nln! / jac_nln_* / the lag_mul scatter of hprod_nln!
    (src/model-Fletcherpenaltynlp.jl:260-350).  The view of a model with one linear and one nonlinear constraint must
    coincide with the same model written without the linear constraint."""
    from fps_amd import nlpmodels

    full, nl = nlpmodels.LinearPlusCircle(True), nlpmodels.LinearPlusCircle(False)
    assert (full.meta.nlin, full.meta.nnln, list(full.meta.lin), list(full.meta.nln)) == (1, 1, [0], [1])
    view = nlpmodels.NonlinearConstraintsView(full)
    rng = np.random.default_rng(2)
    x, v, g = rng.standard_normal(3), rng.standard_normal(3), rng.standard_normal(3)
    y = rng.standard_normal(1)
    assert view.meta.ncon == 1 and np.allclose(view.cons(x), nl.cons(x))
    for a, b in zip(view.jac_structure(), nl.jac_structure()):
        assert np.array_equal(a, b)
    assert np.allclose(view.jac_coord(x), nl.jac_coord(x))
    assert np.allclose(view.jtprod(x, y), nl.jtprod(x, y)) and np.allclose(view.jprod(x, v), nl.jprod(x, v))
    assert np.allclose(view.hprod(x, y, v), nl.hprod(x, y, v)) and np.allclose(view.ghjvprod(x, g, v), nl.ghjvprod(x, g, v))
