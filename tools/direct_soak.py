"""Developer soak (GPU box, one run): the two direct back-ends (fpsq_dense_*, fpsq_band_*: include/fpsq.h) over many random shapes, each handle
refactorised and solved again and again:  KKT residuals of solve_two_mixed / solve_two_least_squares <= 1e-10 (relative), and the SAME
inputs a second time must give the same bits (the chained sweeps hand blocks between workgroups through tickets and self-validating
words; the factorisation's launches run back to back: a race would show as a residual or as a changed bit).

    python tools/direct_soak.py [dense handles=40] [band handles=25] [repeats per handle=12]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fps_amd  # noqa: E402,F401
from fps_amd import _lib, problems  # noqa: E402

ND = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 25
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib = _lib.load()
rng = np.random.default_rng(2025)
bad = {"residual": 0, "bits": 0, "rc": 0}
worst = 0.0


def residuals(A, AT, delta, g, r2, outs, ls):
    """relative KKT residuals of the two systems: [I A'; A -delta I] (p, q) = (g, 0) and (0, c) [mixed] / (g2, 0) [least squares]"""
    p1, q1, p2, q2 = outs
    m = q1.size
    z = np.zeros(m)
    rhs = ((g, z), (r2, z)) if ls else ((g, z), (np.zeros(g.size), r2))
    out = 0.0
    for (p, q), (a, b) in zip(((p1, q1), (p2, q2)), rhs):
        res = np.concatenate([p + AT @ q - a, A @ p - delta * q - b])
        out = max(out, np.linalg.norm(res) / ((np.linalg.norm(a) + np.linalg.norm(b)) * max(1.0, np.linalg.norm(q), np.linalg.norm(p))))
    return out


def run(handle_solve, A, AT, delta, n, m, what):
    global worst
    g, c, g2 = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(n)
    for ls, r2, fn in ((False, c, 0), (True, g2, 1)):
        first = handle_solve(fn, g, r2)
        r = residuals(A, AT, delta, g, r2, first, ls)
        worst = max(worst, r)
        if not r <= 1e-10:
            bad["residual"] += 1
            print(f"RESIDUAL {r:.3e}: {what} delta={delta} ls={ls}", flush=True)
        again = handle_solve(fn, g, r2)
        if not all(np.array_equal(a, b) for a, b in zip(first, again)):
            bad["bits"] += 1
            print(f"BITS differ between two solves of the same systems: {what} delta={delta} ls={ls}", flush=True)


t0 = time.time()
nsolve = 0
for k in range(ND):
    m = int(rng.integers(1, 700))
    n = int(rng.integers(m, 3 * m + 8))
    d = C.c_void_p()
    if lib.fpsq_dense_create(C.byref(d), n, m, 0) != 0:
        bad["rc"] += 1
        continue
    fns = (lib.fpsq_dense_solve_two_mixed, lib.fpsq_dense_solve_two_least_squares)

    def solve(fn, r1, r2):
        outs = [np.empty(n), np.empty(m), np.empty(n), np.empty(m)]
        r1, r2 = np.ascontiguousarray(r1), np.ascontiguousarray(r2)
        if fns[fn](d, r1.ctypes.data, r2.ctypes.data, *[o.ctypes.data for o in outs]) != 0:
            bad["rc"] += 1
        return outs

    for rep in range(REP):
        A = np.ascontiguousarray(rng.uniform(-1, 1, (m, n)) / np.sqrt(n))
        delta = float(rng.choice([0.0, 1.4901161193847656e-08, 0.25]))
        info = C.c_int32()
        if lib.fpsq_dense_set_jacobian(d, A.ctypes.data) != 0 or lib.fpsq_dense_factorize(d, delta, C.byref(info)) != 0 or info.value != 0:
            bad["rc"] += 1
            continue
        run(solve, A, A.T, delta, n, m, f"dense m={m} n={n} rep={rep}")
        nsolve += 4
    lib.fpsq_dense_destroy(d)
    if k % 10 == 9:
        print(f"dense: {k + 1} handles, {time.time() - t0:.0f} s, {bad}, worst residual {worst:.2e}", flush=True)
for k in range(NB):
    m = int(rng.integers(40, 3000))
    n = int(rng.integers(2 * m, 6 * m))
    per_row = int(rng.integers(6, 30))
    window = int(rng.integers(max(per_row * 2, 64), max(per_row * 2 + 1, min(n, 4000))))
    qp = problems.pde_control_like(n=n, m=m, per_row=per_row, window=window, seed=int(rng.integers(1, 10**6)))
    A0 = qp.scipy_csr()
    A0.sort_indices()
    b = C.c_void_p()
    rp, ci = A0.indptr.astype(np.int32), A0.indices.astype(np.int32)
    if lib.fpsq_band_create(C.byref(b), n, m, rp.ctypes.data, ci.ctypes.data, 0) != 0:
        print("band create refused:", lib.fpsq_band_last_error(None).decode()[:200], f"(m={m} n={n} per_row={per_row} window={window})", flush=True)
        continue
    fns = (lib.fpsq_band_solve_two_mixed, lib.fpsq_band_solve_two_least_squares)

    def solve(fn, r1, r2):
        outs = [np.empty(n), np.empty(m), np.empty(n), np.empty(m)]
        r1, r2 = np.ascontiguousarray(r1), np.ascontiguousarray(r2)
        if fns[fn](b, r1.ctypes.data, r2.ctypes.data, *[o.ctypes.data for o in outs]) != 0:
            bad["rc"] += 1
        return outs

    for rep in range(REP):
        A = A0.copy()
        A.data = A0.data * rng.uniform(0.5, 2.0, A0.data.size)
        delta = float(rng.choice([0.0, 1.4901161193847656e-08, 0.25]))
        info = C.c_int32()
        v = np.ascontiguousarray(A.data)
        if lib.fpsq_band_factorize(b, v.ctypes.data, delta, C.byref(info)) != 0 or info.value != 0:
            bad["rc"] += 1
            print(f"band factorize: rc / info {info.value}: m={m} n={n} per_row={per_row} window={window} delta={delta}", flush=True)
            continue
        run(solve, A, A.T.tocsr(), delta, n, m, f"band m={m} n={n} per_row={per_row} window={window} rep={rep}")
        nsolve += 4
    lib.fpsq_band_destroy(b)
    if k % 5 == 4:
        print(f"band: {k + 1} handles, {time.time() - t0:.0f} s, {bad}, worst residual {worst:.2e}", flush=True)
print(f"{ND} dense + {NB} banded handles x {REP} factorisations, {nsolve} two-system solves: {bad}; worst relative KKT residual {worst:.2e}; {time.time() - t0:.0f} s")
sys.exit(1 if any(bad.values()) else 0)
