# HIPQDSolver.jl -- the reference-side binding of libfpsq.so (UNEXECUTED: julia is not available in the build
# pipeline; kept in sync with include/fpsq.h by hand).  Drop this file next to src/solve_two_systems_struct.jl of
# FletcherPenaltySolver.jl, `include` it from src/FletcherPenaltySolver.jl after solve_two_systems_struct.jl, and
#     FletcherPenaltySolver.qdsolver_correspondence[:hip] = HIPQDSolver       # src/parameters.jl:197
# so that `fps_solve(nlp; qds_solver = :hip)` (src/parameters.jl:290,299) builds it.

const libfpsq = get(ENV, "LIBFPSQ", "libfpsq.so")

struct FpsqStats            # fpsq_stats
  solved::Int32
  inconsistent::Int32
  niter::Int32
  status::Int32
  rnorm::Float64
  arnorm::Float64
end

mutable struct FpsqOptions  # fpsq_options (same field order as include/fpsq.h)
  ls_atol::Float64; ls_rtol::Float64; ls_itmax::Int64
  ln_atol::Float64; ln_rtol::Float64; ln_btol::Float64; ln_conlim::Float64; ln_itmax::Int64
  ne_atol::Float64; ne_rtol::Float64; ne_etol::Float64; ne_itmax::Int64; ne_conlim::Float64
  ls_axtol::Float64; ls_btol::Float64; ls_etol::Float64; ls_conlim::Float64
  fuse_two_rhs::Int32; lookahead::Int32; device::Int32; jac_format::Int32
  ln_method::Int32   # 0 = craig! (the default workspace, struct.jl:121), 1 = lnlq! through the generic solve_least_norm
  kkt_method::Int32  # 0 = LSQR + CRAIG (the reference's path), 1 = MINRES on K itself (not a reference path)
  FpsqOptions() = new()
end

"""
    HIPQDSolver(nlp::AbstractNLPModel, ::T; kwargs...) <: QDSolver

MI355X back-end for the systems `[I A'; A -δI]`; same constructor contract as `IterativeSolver`
(src/solve_two_systems_struct.jl:94-131): the `ls_*`, `ln_*`, `ne_*` keywords are honoured, others are swallowed.
`ln_method = 1` plays the role of passing `solver_struct_least_norm = LnlqWorkspace(...)` (struct.jl:121).
"""
mutable struct HIPQDSolver{T, S} <: QDSolver
  handle::Ptr{Cvoid}
  nvar::Int
  ncon::Int
  vals::S            # jac_coord! output buffer (COO order of jac_structure!)
  p1::S; q1::S; p2::S; q2::S
  stats::Vector{FpsqStats}
  δ::T
  explicit_linear_constraints::Bool   # only the nonlinear rows of the Jacobian enter the systems (struct.jl:101-103)
end

function HIPQDSolver(nlp::AbstractNLPModel{T, S}, ::T; explicit_linear_constraints = false, kwargs...) where {T, S}
  T == Float64 || error("HIPQDSolver is fp64 only")
  nvar = nlp.meta.nvar
  ncon = explicit_linear_constraints ? nlp.meta.nnln : nlp.meta.ncon
  nnzj = explicit_linear_constraints ? nlp.meta.nln_nnzj : nlp.meta.nnzj
  opts = FpsqOptions()
  ccall((:fpsq_default_options, libfpsq), Cvoid, (Int64, Int64, Ref{FpsqOptions}), nvar, ncon, opts)
  for (k, v) in kwargs
    hasfield(FpsqOptions, k) && setfield!(opts, k, convert(fieldtype(FpsqOptions, k), v))
  end
  h = Ref{Ptr{Cvoid}}(C_NULL)
  rc = ccall((:fpsq_create, libfpsq), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Ref{FpsqOptions}), h, nvar, ncon, opts)
  rc == 0 || error(unsafe_string(ccall((:fpsq_last_error, libfpsq), Cstring, (Ptr{Cvoid},), C_NULL)))
  rows, cols = explicit_linear_constraints ? jac_nln_structure(nlp) : jac_structure(nlp)   # replaces struct.jl:331-337
  rc = ccall((:fpsq_set_jacobian_structure_coo, libfpsq), Cint,
             (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int32), h[], nnzj, Int64.(rows), Int64.(cols), 1)
  rc == 0 || error(unsafe_string(ccall((:fpsq_last_error, libfpsq), Cstring, (Ptr{Cvoid},), h[])))
  qds = HIPQDSolver{T, S}(h[], nvar, ncon, S(undef, nnzj), S(undef, nvar), S(undef, ncon), S(undef, nvar),
                          S(undef, ncon), Vector{FpsqStats}(undef, 2), T(NaN), explicit_linear_constraints)
  finalizer(q -> ccall((:fpsq_destroy, libfpsq), Cint, (Ptr{Cvoid},), q.handle), qds)
  return qds
end

function _refresh!(qds::HIPQDSolver, nlp, x; values = true)
  if values                                             # replaces linear_system.jl:118-122 / :223-228
    qds.explicit_linear_constraints ? jac_nln_coord!(nlp.nlp, x, qds.vals) : jac_coord!(nlp.nlp, x, qds.vals)
    ccall((:fpsq_set_jacobian_values, libfpsq), Cint, (Ptr{Cvoid}, Ptr{Float64}), qds.handle, qds.vals)
  end
  if qds.δ != nlp.δ
    ccall((:fpsq_set_delta, libfpsq), Cint, (Ptr{Cvoid}, Float64), qds.handle, nlp.δ)
    qds.δ = nlp.δ
  end
end

function solve_two_mixed(nlp::FletcherPenaltyNLP{T, S, A, P, HIPQDSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  qds = nlp.qdsolver
  _refresh!(qds, nlp, x)
  rc = ccall((:fpsq_solve_two_mixed, libfpsq), Cint,
             (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{FpsqStats}),
             qds.handle, rhs1, rhs2, qds.p1, qds.q1, qds.p2, qds.q2, qds.stats)
  rc < 0 && error(unsafe_string(ccall((:fpsq_last_error, libfpsq), Cstring, (Ptr{Cvoid},), qds.handle)))
  (rc & 1) != 0 && @warn "Failed solving 1st linear system lsqr in mixed."
  (rc & 2) != 0 && @warn "Failed solving 2nd linear system craig in mixed."
  return qds.p1, qds.q1, qds.p2, qds.q2
end

function solve_two_least_squares(nlp::FletcherPenaltyNLP{T, S, A, P, HIPQDSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  qds = nlp.qdsolver
  _refresh!(qds, nlp, x; values = false)               # the reference trusts the cached operator, linear_system.jl:85-86
  rc = ccall((:fpsq_solve_two_least_squares, libfpsq), Cint,
             (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{FpsqStats}),
             qds.handle, rhs1, rhs2, qds.p1, qds.q1, qds.p2, qds.q2, qds.stats)
  rc < 0 && error(unsafe_string(ccall((:fpsq_last_error, libfpsq), Cstring, (Ptr{Cvoid},), qds.handle)))
  (rc & 1) != 0 && @warn "Failed solving 1st linear system lsqr."
  (rc & 2) != 0 && @warn "Failed solving 2nd linear system lsqr."
  return qds.p1, qds.q1, qds.p2, qds.q2
end

function solve_two_extras(nlp::FletcherPenaltyNLP{T, S, A, P, HIPQDSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  qds = nlp.qdsolver
  _refresh!(qds, nlp, x; values = false)
  rc = ccall((:fpsq_solve_two_extras, libfpsq), Cint,
             (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{FpsqStats}),
             qds.handle, rhs1, rhs2, qds.q1, qds.q2, qds.stats)
  rc < 0 && error(unsafe_string(ccall((:fpsq_last_error, libfpsq), Cstring, (Ptr{Cvoid},), qds.handle)))
  (rc & 1) != 0 && @warn "Failed solving 1st linear system lsqr in extra."
  (rc & 2) != 0 && @warn "Failed solving 2nd linear system minres in extra."
  return qds.q1, qds.q2
end


# ------------------------------------------------------------------------------------------------------------------
# Direct back-end for sparse banded Jacobians: the role of LDLtSolver (src/solve_two_systems_struct.jl:299-353,
# src/solve_linear_system.jl:142-252) on the MI355X -- include/fpsq.h, fpsq_band_*.  Same seam, same keyword names
# (ldlt_tol, ldlt_r1, ldlt_r2).     FletcherPenaltySolver.qdsolver_correspondence[:hip_ldlt] = HIPLDLtSolver

struct FpsqBandInfo
  n::Int64; m::Int64; nnz::Int64; nblocks::Int64; bandwidth_blocks::Int64; factor_bytes::Int64
  last_form_ms::Float64; last_chol_ms::Float64; last_solve_ms::Float64; regularized_pivots::Int64; reordered::Int64; chains::Int64
end

mutable struct HIPLDLtSolver{T, S} <: QDSolver
  handle::Ptr{Cvoid}
  nvar::Int
  ncon::Int
  coo_vals::S          # jac_coord! output (model order); sorted into the CSR slots ON THE DEVICE (fpsq_band_factorize_coo)
  p1::S; q1::S; p2::S; q2::S
  e1::S; e2::S; ep::S  # solve_two_extras has outputs of its own: hprod! Val(1) still reads p2 of the preceding
                       # solve_two_least_squares after it (src/model-Fletcherpenaltynlp.jl:593-611)
  factorized::Bool
  explicit_linear_constraints::Bool
end

function HIPLDLtSolver(nlp::AbstractNLPModel{T, S}, ::T; explicit_linear_constraints = false,
                       ldlt_tol = √eps(T), ldlt_r1 = √eps(T), ldlt_r2 = -√eps(T) #= struct.jl:314, the reference's default; ldlt_r2 = -1e200 (FPSQ_REG_DROP, include/fpsq.h) drops a vanishing pivot of M instead =#, kwargs...) where {T, S}
  T == Float64 || error("HIPLDLtSolver is fp64 only")
  nvar = nlp.meta.nvar
  ncon = explicit_linear_constraints ? nlp.meta.nnln : nlp.meta.ncon
  rows, cols = explicit_linear_constraints ? jac_nln_structure(nlp) : jac_structure(nlp)       # struct.jl:331-337
  h = Ref{Ptr{Cvoid}}(C_NULL)
  # symbolic phase = ldl_analyze (struct.jl:344); the COO triplets go over as they are (1-based, duplicates summed)
  rc = ccall((:fpsq_band_create_coo, libfpsq), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Int32),
             h, nvar, ncon, length(rows), Int64.(rows), Int64.(cols), 1, 0)
  rc == 0 || error(unsafe_string(ccall((:fpsq_band_last_error, libfpsq), Cstring, (Ptr{Cvoid},), C_NULL)))
  ccall((:fpsq_band_set_regularization, libfpsq), Cint, (Ptr{Cvoid}, Float64, Float64), h[], ldlt_tol, -ldlt_r2)  # :345-348
  qds = HIPLDLtSolver{T, S}(h[], nvar, ncon, S(undef, length(rows)), S(undef, nvar), S(undef, ncon), S(undef, nvar),
                            S(undef, ncon), S(undef, ncon), S(undef, ncon), S(undef, nvar), false, explicit_linear_constraints)
  finalizer(q -> ccall((:fpsq_band_destroy, libfpsq), Cint, (Ptr{Cvoid},), q.handle), qds)
  return qds
end

function _factorize!(qds::HIPLDLtSolver, nlp, x, δ)
  qds.explicit_linear_constraints ? jac_nln_coord!(nlp.nlp, x, qds.coo_vals) : jac_coord!(nlp.nlp, x, qds.coo_vals)  # :223-228
  info = Ref{Int32}(0)
  rc = ccall((:fpsq_band_factorize_coo, libfpsq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Ref{Int32}),
             qds.handle, qds.coo_vals, δ, info)                       # sparse(...) + ldl_factorize!, :233-234
  rc < 0 && error(unsafe_string(ccall((:fpsq_band_last_error, libfpsq), Cstring, (Ptr{Cvoid},), qds.handle)))
  qds.factorized = rc == 0
end

function _band_solve!(qds::HIPLDLtSolver, fn::Symbol, rhs1, rhs2, p1 = qds.p1, q1 = qds.q1, p2 = qds.p2, q2 = qds.q2)
  if !qds.factorized
    @warn "_solve_ldlt_factorization: failed _factorization"                        # linear_system.jl:196-198, :244-246
    return p1, q1, p2, q2
  end
  rc = fn == :mixed ?
    ccall((:fpsq_band_solve_two_mixed, libfpsq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
          qds.handle, rhs1, rhs2, p1, q1, p2, q2) :
    ccall((:fpsq_band_solve_two_least_squares, libfpsq), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
          qds.handle, rhs1, rhs2, p1, q1, p2, q2)
  rc < 0 && error(unsafe_string(ccall((:fpsq_band_last_error, libfpsq), Cstring, (Ptr{Cvoid},), qds.handle)))
  return p1, q1, p2, q2
end

function solve_two_mixed(nlp::FletcherPenaltyNLP{T, S, A, P, HIPLDLtSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  _factorize!(nlp.qdsolver, nlp, x, nlp.δ)                                                       # :206-252
  return _band_solve!(nlp.qdsolver, :mixed, rhs1, rhs2)
end

function solve_two_least_squares(nlp::FletcherPenaltyNLP{T, S, A, P, HIPLDLtSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  return _band_solve!(nlp.qdsolver, :least_squares, rhs1, rhs2)                                  # :161-204 (cached factor)
end

function solve_two_extras(nlp::FletcherPenaltyNLP{T, S, A, P, HIPLDLtSolver{T, S}}, x::AbstractVector, rhs1, rhs2) where {T, S, A, P}
  qds = nlp.qdsolver
  τ = max(nlp.δ, 1e-14)                                                                          # :148: tau, Jacobian at x
  _factorize!(qds, nlp, x, τ)
  # own output buffers: p1 / p2 of the preceding solve_two_least_squares stay intact (hprod! Val(1) reads p2 afterwards)
  _, q1, _, q2 = _band_solve!(qds, :mixed, rhs1, rhs2, qds.ep, qds.e1, qds.ep, qds.e2)
  qds.e2 .= .-q2
  # the reference's extras (cgls / minres on the operator) never touch the LDL' factors: leave the cached factor the one
  # of δ, which the next solve_two_least_squares re-uses (:194-195)
  τ != nlp.δ && _factorize!(qds, nlp, x, nlp.δ)
  return q1, qds.e2                                       # (A A' + tau I)^-1 A rhs1,  (A A' + tau I)^-1 rhs2
end
