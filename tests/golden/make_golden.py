"""Writes tests/golden/reference_known_answers.json.

The numbers are the reference's OWN known-answer assertions for the penalty-evaluation path, i.e. the
inputs and expected outputs stated in /root/reference/test/unit-test.jl (cited per case below).  They
are data (inputs + expected outputs), evaluated here from the closed forms the test file asserts; no
reference source is copied and nothing under /root/reference is read at run time.

Run:  python tests/golden/make_golden.py
"""
import json
import math
import os

cases = []

# --- test/unit-test.jl:16-76 (Val(1)) and :154-214 (Val(2)): f = x'x, c = sum(x) - 1, n = 10,
#     FletcherPenaltyNLP(nlp, 0.5, Val(k)) => sigma = .5, rho = delta = 0; default LDLt back-end.
n = 10
sigma = 0.5
ys = lambda x: (2 - sigma) / n * sum(x) + sigma / n          # unit-test.jl:27
Ys = (2 - sigma) / n                                          # unit-test.jl:28 (times ones(n))
xfeas = [1.0 / n] * n                                         # unit-test.jl:40
cases.append(dict(
    name="sumsq_n10_xfeas", cite="test/unit-test.jl:40-48", model="sumsq", n=n, m=1,
    sigma=sigma, rho=0.0, delta=0.0, x=xfeas,
    jac_rows=[0] * n, jac_cols=list(range(n)), jac_vals=[1.0] * n,
    g=[0.2] * n, c=[0.0], f=0.1,
    expect=dict(obj=0.1, fx=0.1, gx=[0.2] * n, ys=[ys(xfeas)], cx=[0.0], grad=[0.0] * n),
    atol=dict(obj=1e-14, fx=1e-14, gx=1e-14, ys=1e-14, cx=1e-14, grad=1e-14)))
xr = [0.0] + [1.0] * 9                                        # unit-test.jl:54
cx = 8.0                                                      # unit-test.jl:56
cases.append(dict(
    name="sumsq_n10_xr", cite="test/unit-test.jl:54-59", model="sumsq", n=n, m=1,
    sigma=sigma, rho=0.0, delta=0.0, x=xr,
    jac_rows=[0] * n, jac_cols=list(range(n)), jac_vals=[1.0] * n,
    g=[2 * v for v in xr], c=[cx], f=9.0,
    expect=dict(obj=9.0 - cx * ys(xr), ys=[ys(xr)], cx=[cx],
                grad=[2 * v - Ys * cx - ys(xr) for v in xr]),     # unit-test.jl:58-59
    atol=dict(obj=1e-13, ys=1e-13, cx=0.0, grad=1e-13)))

# hprod known answer of the same model (unit-test.jl:190-191 at xfeas, atol 1e-13; :201-202 at xr, atol 1e-12):
#   hprod(fpnlp, x, v) = 2 v - 2 ones(n) Ys' v   for ANY v (the reference draws v = rand(n)); fixed v here.
vfix = [((7 * i + 3) % 11) / 11.0 for i in range(n)]
hv = [2 * vi - 2 * Ys * sum(vfix) for vi in vfix]
cases[0]["hprod"] = dict(v=vfix, expect=hv, atol=1e-13, cite="test/unit-test.jl:190-191")
cases[1]["hprod"] = dict(v=vfix, expect=hv, atol=1e-12, cite="test/unit-test.jl:201-202")

# --- test/unit-test.jl:78-152 (and :216-287): Rosenbrock + unit circle,
#     FletcherPenaltyNLP(nlp, 0.5, 0.1, 0.25, Val(k)) => sigma = .5, rho = .1, delta = .25
sigma, rho, delta = 0.5, 0.1, 0.25
x1, x2 = math.sqrt(6) / 3, math.sqrt(3) / 3                   # unit-test.jl:100
D = -(4 * x1 ** 2 + 4 * x2 ** 2 + delta)                      # unit-test.jl:102
c = x1 ** 2 + x2 ** 2 - 1
ys_c = (2 * x1 * (-2 * (x1 - 1) + 400 * x1 * (x2 - x1 ** 2)) - 400 * x2 * (x2 - x1 ** 2) + sigma * c) / D  # :103-107
phi = (math.sqrt(6) - 3) ** 2 / 9 + 100 * (math.sqrt(3) - 2) ** 2 / 9     # unit-test.jl:118
gx = [2 * (math.sqrt(6) / 3 - 1) - 400 * math.sqrt(6) / 3 * (math.sqrt(3) / 3 - 6 / 9),
      200 * (math.sqrt(3) / 3 - 6 / 9)]                                    # unit-test.jl:121-124
cases.append(dict(
    name="rosenbrock_circle_xr", cite="test/unit-test.jl:100-126", model="rosenbrock_circle", n=2, m=1,
    sigma=sigma, rho=rho, delta=delta, x=[x1, x2],
    jac_rows=[0, 0], jac_cols=[0, 1], jac_vals=[2 * x1, 2 * x2],
    g=gx, c=[0.0], f=phi,
    expect=dict(obj=phi, fx=phi, gx=gx, ys=[ys_c], cx=[0.0]),
    atol=dict(obj=1e-14, fx=1e-14, gx=1e-13, ys=1e-14, cx=1e-14)))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
with open(out, "w") as fh:
    json.dump(dict(source="JuliaSmoothOptimizers/FletcherPenaltySolver.jl v0.3.0 test/unit-test.jl", cases=cases),
              fh, indent=1)
print("wrote", out, len(cases), "cases")
