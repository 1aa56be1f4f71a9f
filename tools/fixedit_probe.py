"""Developer probe (GPU): LSQR of solve_two_mixed cut at k = 1, 2, ... iterations on the device and in the C restatement:
where do the residual estimates and the iterates start to differ?   python tools/fixedit_probe.py KIND DELTA KMAX"""
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

kind = sys.argv[1] if len(sys.argv) > 1 else "empty-columns"
delta = float(sys.argv[2]) if len(sys.argv) > 2 else T.SE
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(12)
A = T._random_structure(kind, rng)
m, n = A.shape
rng.standard_normal(n), rng.standard_normal(m)
g, c = rng.standard_normal(n), rng.standard_normal(m)
rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data)
for k in range(1, kmax + 1):
    H = T._Handle(A, delta=delta, ls_itmax=k, ln_itmax=1)
    p1, q1, p2, q2, rc = H.solve_two_mixed(g, c)
    d = (H.st[0].niter, H.st[0].status, H.st[0].rnorm, H.st[0].arnorm)
    H.close()
    o = oracle.solve_two_mixed(m, n, rp, ci, va, delta, g, c, opts=oracle.default_options(n, m, ls_itmax=k, ln_itmax=1))
    s = o[4][0]
    print(f"k={k:3d} dev it={d[0]} st={d[1]} | orc it={s.niter} st={s.status} | rnorm rel diff {abs(d[2] - s.rnorm) / s.rnorm:.2e} "
          f"arnorm dev {d[3]:.6e} orc {s.arnorm:.6e} rel {abs(d[3] - s.arnorm) / max(s.arnorm, 1e-300):.2e} | q1 rel {T._rel(q1, o[1]):.2e}",
          flush=True)
