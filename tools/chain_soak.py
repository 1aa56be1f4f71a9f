"""Developer soak (GPU box, one run): the chained triangular sweeps (k_trsv_chain) against the step kernels (FPSQ_TRSV_CHAIN=0)
over many solves -- dense (16 block rows), banded with two elimination chains, banded wide.  Every solve of the chained
handle must agree with the step handle's to 1e-12 relative (same sums in the same order per block) and return rc 0.
usage: python tools/chain_soak.py [solves]"""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fps_amd  # noqa: F401
from fps_amd import _lib, problems
from test_gpu_dense import _Dense, _Band

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


cases = []
A = np.random.default_rng(5).uniform(-1, 1, (2048, 3000)) / np.sqrt(3000)
cases.append(("dense nb=16", A, _Dense, "fpsq_dense_solve_two_mixed"))
qp = problems.pde_control_like(n=30000, m=7700, per_row=12, window=600, seed=11)
cases.append(("band two chains nb=61", qp.scipy_csr(), _Band, "fpsq_band_solve_two_mixed"))
qp = problems.pde_control_like(n=6000, m=3000, per_row=12, window=5000, seed=11)
cases.append(("band full nb=24", qp.scipy_csr(), _Band, "fpsq_band_solve_two_mixed"))
bad = 0
for name, A, H, fn in cases:
    m, n = A.shape
    hs = {}
    for chain in ("1", "0"):
        os.environ["FPSQ_TRSV_CHAIN"] = chain
        hs[chain] = H(A)
        assert hs[chain].factorize(0.25)[0] == 0
    rng = np.random.default_rng(1)
    t0 = time.time()
    worst = 0.0
    for k in range(N):
        g, c = rng.standard_normal(n) * 10.0 ** (k % 5 - 2), rng.standard_normal(m)
        a = hs["1"].solve(getattr(hs["1"].lib, fn), g, c)
        b = hs["0"].solve(getattr(hs["0"].lib, fn), g, c)
        w = max(rel(x, y) for x, y in zip(a, b))
        worst = max(worst, w)
        if not w < 1e-12:
            bad += 1
            print("MISMATCH", name, k, w, flush=True)
    print(f"{name}: {N} solves, worst relative difference {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
    for h in hs.values():
        h.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
