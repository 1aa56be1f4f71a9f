# bench/julia_ref.jl -- baseline B4 of BASELINE.md section 3: the REAL reference timed on the headline workload.
#                                                                         *** UNEXECUTED in this pipeline ***
# (Julia is not installed in the build image nor on the GPU box; this script is for any later environment that has
#  Julia >= 1.6 with FletcherPenaltySolver v0.3.0, NLPModels 0.21, Krylov 0.10.)
#
#   julia -e 'using Pkg; Pkg.add(["FletcherPenaltySolver", "NLPModels", "JSON"])'
#   julia bench/julia_ref.jl [n m] [delta]          # default: n = 1_000_000, m = 100_000, delta = 0
#
# What it times: `grad!(FletcherPenaltyNLP(qp, sigma, rho, delta, Val(2); qds = IterativeSolver(qp, 0.0)), x, gx)` at
# DISTINCT points x (each a memo miss of `hash(x)`, src/model-Fletcherpenaltynlp.jl:235-237) -- exactly the unit of work of
# bench.py -- on the synthetic PDE-control-like equality QP of fps_amd/problems.py, regenerated here bit for bit from the
# same counter-based generator  u(seed, i, k) = splitmix64(seed xor i*GOLDEN xor k*C2) >> 11 / 2^53.
# It prints one JSON line with evals/s, the iteration counts of lsqr / craig and the core count (1: the reference is
# single-threaded), to be put next to bench.py's `cpu_baseline`.
using FletcherPenaltySolver, NLPModels, LinearAlgebra, SparseArrays, Printf

const GOLDEN = 0x9E3779B97F4A7C15
const C2 = 0xD1B54A32D192ED03

function splitmix64(z::UInt64)
  z += GOLDEN
  z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
  z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
  return z ⊻ (z >> 31)
end
u01(seed, i, k) = Float64(splitmix64(UInt64(seed) ⊻ (UInt64(i) * GOLDEN) ⊻ (UInt64(k) * C2)) >> 11) / 9007199254740992.0

# problems.pde_control_like: row i has `per_row` nonzeros, one per stratum of a `window`-wide column window centred at
# floor(i n / m) (clamped); the stratum holding the centre column is placed exactly there with +4 on its value.
# i, k and every column index are 0-based inside the generator (as in the Python file); the Julia arrays are 1-based.
function pde_control_like(n, m; per_row = 100, window = 8192, seed = 1234)
  window = min(window, n)
  rows = Vector{Int}(undef, m * per_row); cols = similar(rows); vals = Vector{Float64}(undef, m * per_row)
  for i in 0:(m - 1)
    center = (i * n) ÷ m
    start = clamp(center - window ÷ 2, 0, n - window)
    for k in 0:(per_row - 1)
      lo = (k * window) ÷ per_row; hi = ((k + 1) * window) ÷ per_row
      eid = i * per_row + k
      off = lo + floor(Int, u01(seed, eid, 11) * (hi - lo))
      v = 2.0 * u01(seed, eid, 12) - 1.0
      col = start + off
      rel = center - start
      if lo <= rel < hi
        col = center; v += 4.0
      end
      rows[eid + 1] = i + 1; cols[eid + 1] = col + 1; vals[eid + 1] = v
    end
  end
  A = sparse(rows, cols, vals, m, n)                 # (distinct columns per row by construction: no duplicates summed)
  q = [1.0 + 9.0 * u01(seed, i, 1) for i in 0:(n - 1)]
  d = [2.0 * u01(seed, i, 2) - 1.0 for i in 0:(n - 1)]
  xhat = [2.0 * u01(seed, i, 3) - 1.0 for i in 0:(n - 1)]
  b = A * xhat
  return A, q, d, b, xhat
end
point(xhat, t) = [xhat[i + 1] + 0.1 * (2.0 * u01(977 + t, i, 5) - 1.0) for i in 0:(length(xhat) - 1)]

# minimal NLPModel:  f = 1/2 x' diag(q) x + d'x,  c(x) = A x - b = 0
mutable struct EqQP{T, S} <: AbstractNLPModel{T, S}
  meta::NLPModelMeta{T, S}
  counters::Counters
  A::SparseMatrixCSC{T, Int}
  q::S
  d::S
  b::S
  jrows::Vector{Int}
  jcols::Vector{Int}
end
function EqQP(A, q, d, b, x0)
  m, n = size(A)
  jr, jc, _ = findnz(A)
  meta = NLPModelMeta(n; ncon = m, x0 = x0, lcon = zeros(m), ucon = zeros(m), nnzj = nnz(A), nnzh = n,
                      minimize = true, name = "pde-control-like")
  return EqQP(meta, Counters(), A, q, d, b, jr, jc)
end
NLPModels.obj(p::EqQP, x::AbstractVector) = (increment!(p, :neval_obj); dot(x, 0.5 .* p.q .* x .+ p.d))
NLPModels.grad!(p::EqQP, x::AbstractVector, g::AbstractVector) = (increment!(p, :neval_grad); g .= p.q .* x .+ p.d; g)
NLPModels.cons!(p::EqQP, x::AbstractVector, c::AbstractVector) = (increment!(p, :neval_cons); mul!(c, p.A, x); c .-= p.b; c)
NLPModels.jac_structure!(p::EqQP, r::AbstractVector{<:Integer}, c::AbstractVector{<:Integer}) = (r .= p.jrows; c .= p.jcols; (r, c))
NLPModels.jac_coord!(p::EqQP, x::AbstractVector, v::AbstractVector) = (increment!(p, :neval_jac); v .= nonzeros(p.A); v)
NLPModels.jprod!(p::EqQP, x::AbstractVector, v::AbstractVector, Jv::AbstractVector) = (increment!(p, :neval_jprod); mul!(Jv, p.A, v); Jv)
NLPModels.jtprod!(p::EqQP, x::AbstractVector, v::AbstractVector, Jtv::AbstractVector) = (increment!(p, :neval_jtprod); mul!(Jtv, p.A', v); Jtv)
NLPModels.hess_structure!(p::EqQP, r::AbstractVector{<:Integer}, c::AbstractVector{<:Integer}) = (r .= 1:p.meta.nvar; c .= 1:p.meta.nvar; (r, c))
NLPModels.hess_coord!(p::EqQP, x::AbstractVector, v::AbstractVector; obj_weight = 1.0) = (v .= obj_weight .* p.q; v)
NLPModels.hess_coord!(p::EqQP, x::AbstractVector, y::AbstractVector, v::AbstractVector; obj_weight = 1.0) = (v .= obj_weight .* p.q; v)
NLPModels.hprod!(p::EqQP, x::AbstractVector, v::AbstractVector, Hv::AbstractVector; obj_weight = 1.0) =
  (increment!(p, :neval_hprod); Hv .= obj_weight .* p.q .* v; Hv)
NLPModels.hprod!(p::EqQP, x::AbstractVector, y::AbstractVector, v::AbstractVector, Hv::AbstractVector; obj_weight = 1.0) =
  (increment!(p, :neval_hprod); Hv .= obj_weight .* p.q .* v; Hv)

function main()
  n = length(ARGS) >= 2 ? parse(Int, ARGS[1]) : 1_000_000
  m = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 100_000
  delta = length(ARGS) >= 3 ? parse(Float64, ARGS[3]) : 0.0
  A, q, d, b, xhat = pde_control_like(n, m)
  qp = EqQP(A, q, d, b, point(xhat, 0))
  sigma, rho = 1e3, 1.0                                # src/parameters.jl:71,75
  qds = FletcherPenaltySolver.IterativeSolver(qp, 0.0) # src/solve_two_systems_struct.jl:94 (reference defaults)
  fp = FletcherPenaltyNLP(qp, sigma, rho, delta, Val(2); qds = qds)
  gx = zeros(n)
  warm, K = 3, 10                                      # protocol of SURVEY.md 8(d)
  for t in 1:warm
    grad!(fp, point(xhat, t), gx)
  end
  xs = [point(xhat, warm + t) for t in 1:K]
  times = Float64[]
  its = Tuple{Int, Int}[]
  for x in xs
    t0 = time_ns()
    grad!(fp, x, gx)
    push!(times, (time_ns() - t0) / 1e9)
    push!(its, (qds.solver_struct_least_square.stats.niter, qds.solver_struct_least_norm.stats.niter))
  end
  med = sort(times)[(K + 1) ÷ 2]
  @printf("{\"metric\": \"penalty grad-phi evals/sec\", \"kind\": \"reference\", \"value\": %.4f, \"min_s\": %.4f, \"median_s\": %.4f, \"cores\": 1, \"n\": %d, \"m\": %d, \"nnz\": %d, \"delta\": %.3e, \"iters_lsqr_craig\": %s, \"julia\": \"%s\"}\n",
          1 / med, minimum(times), med, n, m, nnz(A), delta, string(its[end]), string(VERSION))
end

main()
