"""Developer probe (GPU): iteration counts of the device against the C restatement run in three summation orders
(oracle.set_sum_order: 0 left to right, 1 long rows in the device's order, 2 right to left) on the awkward structures of
tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import fps_amd  # noqa: F401,E402
from oracle import oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402

SE = T.SE
for kind in sys.argv[1:] or ["dense-row", "dense-column", "square-ish"]:
    for delta in (SE, 0.25):
        for fuse in (0, 1):
            rng = np.random.default_rng(12)
            A = T._random_structure(kind, rng)
            m, n = A.shape
            H = T._Handle(A, delta=delta, fuse_two_rhs=fuse)
            x, u = rng.standard_normal(n), rng.standard_normal(m)
            g, c = rng.standard_normal(n), rng.standard_normal(m)
            rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data)
            p1, q1, p2, q2, rc = H.solve_two_mixed(g, c)
            dm = (H.st[0].niter, H.st[1].niter)
            r1, r2 = rng.standard_normal(n), rng.standard_normal(n)
            s1, t1, s2, t2, rc2 = H.solve_two_least_squares(r1, r2)
            dl = (H.st[0].niter, H.st[1].niter)
            line = f"{kind:14s} delta={delta:.2e} fuse={fuse} device mixed={dm} lsq={dl} |"
            for mode in (0, 1, 2):
                oracle.set_sum_order(mode)
                o = oracle.solve_two_mixed(m, n, rp, ci, va, delta, g, c)
                o2 = oracle.solve_two_least_squares(m, n, rp, ci, va, delta, r1, r2)
                e = max(T._rel(a, b) for a, b in zip((p1, q1, p2, q2, s1, t1, s2, t2), (*o[:4], *o2[:4])))
                line += f" mode{mode}: {(o[4][0].niter, o[4][1].niter)} {(o2[4][0].niter, o2[4][1].niter)} err {e:.1e} |"
            oracle.set_sum_order(0)
            print(line, flush=True)
            H.close()
