"""Test infrastructure: a QDSolver backed by the EXACT KKT oracle (oracle/oracle.py, dense / SuperLU), so that the host
mirror of the reference interface (FletcherPenaltyNLP, fps_solve, explicit_linear_constraints) is exercised by the CPU
suite too.  Never used by the product: the shipped back-ends are HIPQDSolver / HIPDirectQDSolver."""
import numpy as np
import scipy.sparse as sp

from fps_amd.qdsolver import QDSolver


class OracleQDSolver(QDSolver):
    def __init__(self, nlp, _zero=0.0, *, explicit_linear_constraints=False, **kwargs):
        from oracle import oracle

        if explicit_linear_constraints:
            from fps_amd.nlpmodels import NonlinearConstraintsView
            nlp = NonlinearConstraintsView(nlp)
        self._o = oracle
        self.nvar, self.ncon = int(nlp.meta.nvar), int(nlp.meta.ncon)
        rows, cols = nlp.jac_structure()
        self._rows, self._cols = np.asarray(rows) - 1, np.asarray(cols) - 1
        self._A = None

    def _jac(self, nlp, x):
        self._A = sp.csr_matrix((np.asarray(nlp.pen.jac_coord(x), float), (self._rows, self._cols)),
                                shape=(self.ncon, self.nvar))
        return self._A

    def solve_two_mixed(self, nlp, x, rhs1, rhs2):
        return self._o.exact_two_mixed(self._jac(nlp, x), nlp.delta, rhs1, rhs2)

    def solve_two_least_squares(self, nlp, x, rhs1, rhs2):
        return self._o.exact_two_least_squares(self._A if self._A is not None else self._jac(nlp, x), nlp.delta, rhs1, rhs2)

    def solve_two_extras(self, nlp, x, rhs1, rhs2):
        return self._o.exact_two_extras(self._jac(nlp, x), nlp.delta, rhs1, rhs2)

    def close(self):
        pass
