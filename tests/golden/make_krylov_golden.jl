# make_krylov_golden.jl -- pins the ITERATIVE back-end against the real Krylov.jl.   *** UNEXECUTED in this pipeline ***
# (no Julia in the build image or on the GPU box).  Anyone with Julia >= 1.6 can settle what the CPU restatement
# (oracle/fps_oracle.c) leaves "unpinned": iteration counts, stats.status and solutions of lsqr / craig(sqd) / minres
# with the keyword arguments FletcherPenaltySolver.jl v0.3.0 passes
#   src/solve_two_systems_struct.jl:173-181 (lsqr: lambda, atol, rtol, itmax), :216-239 (craig: M = (1/delta) I, sqd = true, or plain),
#   src/solve_linear_system.jl:58-70 (minres on Aop * Aop' with lambda = tau).
#
#   julia --project=@. -e 'using Pkg; Pkg.add([PackageSpec(name="Krylov", version="0.10"), PackageSpec(name="LinearOperators"), PackageSpec(name="JSON")])'
#   julia tests/golden/make_krylov_golden.jl          # reads krylov_case_small_pde.json, writes krylov_golden.json
#
# The output layout is what tests/test_oracle.py::test_oracle_matches_krylov_jl_golden reads.
using JSON, Krylov, LinearOperators, SparseArrays, LinearAlgebra

here = @__DIR__
case = JSON.parsefile(joinpath(here, "krylov_case_small_pde.json"))
n, m = case["n"], case["m"]
A = sparse(Int.(case["rows"]), Int.(case["cols"]), Float64.(case["vals"]), m, n)
g = Float64.(case["g"])
c = Float64.(case["c"])
T = Float64
se = sqrt(eps(T))
itmax = 5 * (m + n)                                   # struct.jl:101-103, :108-110
Aop = LinearOperator(A)

statsdict(st) = Dict("solved" => st.solved, "inconsistent" => getfield_or(st, :inconsistent, false),
                     "niter" => st.niter, "status" => st.status)
getfield_or(st, f, d) = hasproperty(st, f) ? getproperty(st, f) : d

runs = []
for delta in Float64.(case["deltas"])
  # solve_least_square(qds, Aop', g, sqrt(delta))                       solve_linear_system.jl:123
  ls = LsqrWorkspace(n, m, Vector{T})                 # struct.jl:116-120 (operator is Aop': n x m)
  krylov_solve!(ls, Aop', g, λ = sqrt(delta), atol = se, rtol = se, itmax = itmax)
  # solve_least_norm(qds, Aop, -c, delta)                               solve_linear_system.jl:132
  cr = CraigWorkspace(m, n, Vector{T})                # struct.jl:121-125
  if delta != 0
    craig!(cr, Aop, -c, M = 1 / delta * opEye(m), sqd = true, atol = se, rtol = se, btol = se, conlim = 1 / se, itmax = itmax)
  else
    craig!(cr, Aop, -c, atol = se, rtol = se, btol = se, conlim = 1 / se, itmax = itmax)
  end
  # solve_two_extras: lsqr with lambda = sqrt(tau), minres on Aop * Aop' with lambda = tau    solve_linear_system.jl:51-70
  tau = max(delta, 1e-14)
  ls2 = LsqrWorkspace(n, m, Vector{T})
  krylov_solve!(ls2, Aop', g, λ = sqrt(tau), atol = se, rtol = se, itmax = itmax)
  mr = MinresWorkspace(m, m, Vector{T})               # struct.jl:126-130
  krylov_solve!(mr, Aop * Aop', c, λ = tau, atol = se, rtol = se, etol = se, conlim = 1 / se, itmax = 0)
  # LNLQ, the commented alternative of struct.jl:121 through the generic solve_least_norm (:251-281)
  lq = LnlqWorkspace(m, n, Vector{T})
  if delta != 0
    krylov_solve!(lq, Aop, -c, M = 1 / delta * opEye(m), atol = se, rtol = se, itmax = itmax)
  else
    krylov_solve!(lq, Aop, -c, atol = se, rtol = se, itmax = itmax)
  end
  push!(runs, Dict(
    "delta" => delta,
    "lsqr" => Dict("x" => copy(ls.x), "stats" => statsdict(ls.stats)),
    "craig" => Dict("x" => copy(cr.x), "y" => copy(cr.y), "stats" => statsdict(cr.stats)),
    "lsqr_tau" => Dict("x" => copy(ls2.x), "stats" => statsdict(ls2.stats)),
    "minres" => Dict("x" => copy(mr.x), "stats" => statsdict(mr.stats)),
    "lnlq" => Dict("x" => copy(lq.x), "y" => copy(lq.y), "stats" => statsdict(lq.stats)),
  ))
end

open(joinpath(here, "krylov_golden.json"), "w") do io
  JSON.print(io, Dict("generator" => "tests/golden/make_krylov_golden.jl",
                      "krylov_version" => string(pkgversion(Krylov)), "julia" => string(VERSION),
                      "case" => "krylov_case_small_pde.json", "runs" => runs))
end
println("wrote krylov_golden.json")
