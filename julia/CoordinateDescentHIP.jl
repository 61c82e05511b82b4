# CoordinateDescentHIP.jl -- the binding a CoordinateDescent.jl maintainer would add to route
# coordinateDescent! for CDLeastSquaresLoss / CDSqrtLassoLoss through libcdhip.so (include/cdhip.h).
#
# NOT EXECUTED in this repository's pipeline: there is no `julia` in the build image or on the GPU
# box.  It is deliberately a thin ccall shim with no arithmetic: every number comes from the
# library.  lasso(), sqrtLasso(), scaledLasso!, LassoPath, CDOptions, ProxL1 and SparseIterate
# are untouched; only the methods below are added (more specific than the generic ones in
# src/coordinate_descent.jl and src/cd_differentiable_function.jl).
module CoordinateDescentHIP

using CoordinateDescent, ProximalBase
using SparseArrays: nnz
import CoordinateDescent: coordinateDescent!, initialize!, gradient, descendCoordinate!,
                          numCoordinates, CDOptions, CDLeastSquaresLoss, CDSqrtLassoLoss

const libcdhip = get(ENV, "LIBCDHIP", "libcdhip.so")

const CDH_OK, CDH_DIM_MISMATCH, CDH_BAD_ARG, CDH_DOMAIN = Int32(0), Int32(1), Int32(2), Int32(3)

struct CdhOptions            # cdh_options: CDOptions field for field (src/utils.jl:7-13) + seed
  maxIter::Int64
  optTol::Float64
  randomize::Int32
  warmStart::Int32
  numSteps::Int64
  seed::UInt64
end
CdhOptions(o::CDOptions) = CdhOptions(o.maxIter, o.optTol, o.randomize, o.warmStart, o.numSteps, rand(UInt64))

mutable struct CdhStats
  passes::Int64; full_passes::Int64; visits::Int64
  converged::Int32; domain_error::Int32; maxH::Float64; lambda_max::Float64
  CdhStats() = new(0, 0, 0, 0, 0, 0.0, 0.0)
end

function check(h::Ptr{Cvoid}, st::Int32)
  st == CDH_OK && return
  msg = unsafe_string(ccall((:cdh_last_error, libcdhip), Cstring, (Ptr{Cvoid},), h))
  st == CDH_DIM_MISMATCH && throw(DimensionMismatch(msg))
  st == CDH_BAD_ARG && throw(ArgumentError(msg))
  st == CDH_DOMAIN && throw(DomainError(NaN, msg))
  error("cdhip status $st: $msg")
end

# A matrix whose columns live in HBM behind a cdh handle.  <: DenseMatrix so it satisfies
# lasso(X::StridedMatrix{T}, ...) (src/lasso.jl:27); getindex is for display only.
mutable struct HipMatrix{T<:Union{Float32,Float64}} <: DenseMatrix{T}
  handle::Ptr{Cvoid}
  n::Int
  p::Int
  loss::Int32
end
Base.size(X::HipMatrix) = (X.n, X.p)
Base.getindex(X::HipMatrix{T}, i::Int, j::Int) where {T} = begin
  col = Vector{T}(undef, X.n)
  check(X.handle, ccall((:cdh_get_X_cols, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64),
                        X.handle, j - 1, 1, col, X.n))
  col[i]
end

dtype_code(::Type{Float64}) = Int32(0)
dtype_code(::Type{Float32}) = Int32(1)

"Upload a host matrix and response; `loss` is 0 (LS) or 1 (SQRT).  Replaces the loss constructors'
`r = copy(y)` (src/cd_differentiable_function.jl:52-55, 211-214)."
function HipMatrix(X::StridedMatrix{T}, y::StridedVector{T}; loss::Integer=0, device::Integer=0) where {T}
  n, p = size(X)
  length(y) == n || throw(DimensionMismatch())
  ref = Ref{Ptr{Cvoid}}(C_NULL)
  check(C_NULL, ccall((:cdh_create, libcdhip), Int32,
                      (Ref{Ptr{Cvoid}}, Int32, Int32, Int64, Int64, Int64, Int64, Int32),
                      ref, dtype_code(T), loss, n, n, 0, p, device))
  h = ref[]
  GC.@preserve X y begin
    check(h, ccall((:cdh_set_X_cols, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64),
                   h, 0, p, X, stride(X, 2)))
    check(h, ccall((:cdh_set_y, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), h, y))
  end
  M = HipMatrix{T}(h, n, p, Int32(loss))
  finalizer(m -> ccall((:cdh_destroy, libcdhip), Int32, (Ptr{Cvoid},), m.handle), M)
  M
end

const HipLS{T}   = CDLeastSquaresLoss{T,<:Any,<:HipMatrix{T}}
const HipSqrt{T} = CDSqrtLassoLoss{T,<:Any,<:HipMatrix{T}}
const HipLoss{T} = Union{HipLS{T},HipSqrt{T}}

numCoordinates(f::HipLoss) = f.X.p

function push_iterate!(f::HipLoss, x::SparseIterate, rebuild::Bool)
  idx = Int64.(x.nzval2ind[1:nnz(x)])
  val = Float64.(x.nzval[1:nnz(x)])
  GC.@preserve idx val begin          # ccall needs a constant symbol: one call site per entry point
    st = rebuild ?
      ccall((:cdh_initialize, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
            f.X.handle, length(x), length(idx), idx, val) :
      ccall((:cdh_set_iterate, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
            f.X.handle, length(x), length(idx), idx, val)
    check(f.X.handle, st)
  end
end

function pull_iterate!(f::HipLoss{T}, x::SparseIterate{T}) where {T}
  p = f.X.p
  idx = Vector{Int64}(undef, p); nz = Ref{Int64}(0); beta = Vector{Float64}(undef, p)
  check(f.X.handle, ccall((:cdh_get_support, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), f.X.handle, idx, nz))
  check(f.X.handle, ccall((:cdh_get_beta, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), f.X.handle, beta))
  fill!(x, zero(T))
  for i in 1:nz[]                       # re-insert in the library's support order
    x[idx[i]] = T(beta[idx[i]])
  end
  # callers read f.r afterwards (src/lasso.jl:37): one device -> host copy per solve
  check(f.X.handle, ccall((:cdh_get_residual, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), f.X.handle, f.r))
  x
end

function set_penalty!(f::HipLoss, g::ProxL1)
  om = g.λ === nothing ? Float64[] : Vector{Float64}(g.λ)    # kept alive across the call
  GC.@preserve om check(f.X.handle,
    ccall((:cdh_set_penalty, libcdhip), Int32, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int64),
          f.X.handle, Float64(g.λ0), g.λ === nothing ? Ptr{Float64}(C_NULL) : pointer(om), length(om)))
end

# ---- the four-function operator interface (src/cd_differentiable_function.jl:1-35) ----------
initialize!(f::HipLoss, x::SparseIterate) = (push_iterate!(f, x, true); nothing)

function gradient(f::HipLoss{T}, x::SparseIterate{T}, k::Int64) where {T}
  push_iterate!(f, x, false)
  out = Ref{Float64}(0)
  check(f.X.handle, ccall((:cdh_gradient, libcdhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), f.X.handle, k, out))
  T(out[])
end

function descendCoordinate!(f::HipLoss{T}, g::ProxL1{T}, x::SparseIterate{T}, k::Int64) where {T}
  set_penalty!(f, g); push_iterate!(f, x, false)
  out = Ref{Float64}(0)
  check(f.X.handle, ccall((:cdh_descend, libcdhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), f.X.handle, k, out))
  pull_iterate!(f, x)
  T(out[])
end

# ---- the performance path: one ccall per solve (src/coordinate_descent.jl:7-39) ---------------
function coordinateDescent!(x::SparseIterate{T}, f::HipLoss{T}, g::ProxL1, options::CDOptions=CDOptions()) where {T}
  ProximalBase.numCoordinates(x) == numCoordinates(f) || throw(DimensionMismatch())
  set_penalty!(f, g)                  # checks length(g.λ) == p (coordinate_descent.jl:14-16)
  push_iterate!(f, x, false)
  opt = Ref(CdhOptions(options)); st = CdhStats()
  code = ccall((:cdh_coordinate_descent, libcdhip), Int32, (Ptr{Cvoid}, Ref{CdhOptions}, Ref{CdhStats}),
               f.X.handle, opt, st)
  check(f.X.handle, code)
  st.domain_error != 0 && throw(DomainError(NaN, "sqrt-lasso update"))
  pull_iterate!(f, x)
end

# Optional knobs (no reference counterpart): blocked sweep width, screened full passes, reuse of
# the carried residual by warm starts (what LassoPath wants: src/lasso.jl:250-252).
set_sweep_mode!(X::HipMatrix, blocked::Bool, block::Integer=32) =
  check(X.handle, ccall((:cdh_set_sweep_mode, libcdhip), Int32, (Ptr{Cvoid}, Int32, Int32), X.handle, blocked ? 1 : 0, block))
set_screening!(X::HipMatrix, on::Bool) =
  check(X.handle, ccall((:cdh_set_screening, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, on ? 1 : 0))
set_reuse_residual!(X::HipMatrix, on::Bool) =
  check(X.handle, ccall((:cdh_set_reuse_residual, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, on ? 1 : 0))

end # module
