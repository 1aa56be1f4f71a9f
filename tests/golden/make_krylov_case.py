"""Writes tests/golden/krylov_case_small_pde.json: the INPUTS of the iterative back-end's known-answer case.

A small instance of the headline generator (fps_amd.problems.pde_control_like, n = 400, m = 60, 20 nonzeros per
row in a 128-column window) with the right-hand sides of one `solve_two_mixed` call at the point qp.x
(g = q .* x + d, c = A x - b) and of one `solve_two_extras` call.  The matrix is stored as 1-based COO triplets, so
`tests/golden/make_krylov_golden.jl` (which needs Julia + Krylov.jl 0.10; neither exists in this pipeline) can turn it
into `tests/golden/krylov_golden.json` = niter / status / solution of the REAL lsqr / craig / minres -- the file
tests/test_oracle.py::test_oracle_matches_krylov_jl_golden pins oracle/fps_oracle.c against when it is present.

Run:  python tests/golden/make_krylov_case.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import fps_amd  # noqa: E402,F401
from fps_amd import problems  # noqa: E402

qp = problems.pde_control_like(n=400, m=60, per_row=20, window=128, seed=77)
A = qp.scipy_csr().tocoo()
g = qp.qdiag * qp.x + qp.d
c = qp.scipy_csr() @ qp.x - qp.b
se = float(np.sqrt(np.finfo(float).eps))
out = dict(
    source="fps_amd.problems.pde_control_like(n=400, m=60, per_row=20, window=128, seed=77); g, c at qp.x",
    n=qp.n, m=qp.m, rows=(A.row + 1).tolist(), cols=(A.col + 1).tolist(), vals=A.data.tolist(),
    g=g.tolist(), c=c.tolist(), deltas=[0.0, se, 0.25],
    note="floats are shortest round-trip decimal representations of the fp64 values")
with open(os.path.join(HERE, "krylov_case_small_pde.json"), "w") as f:
    json.dump(out, f)
print("wrote krylov_case_small_pde.json:", qp.n, qp.m, A.nnz)
