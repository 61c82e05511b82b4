# CoordinateDescentHIP.jl -- the binding a CoordinateDescent.jl maintainer would add to route the
# coordinate-descent path through libcdhip.so (include/cdhip.h) for a design matrix that lives in HBM.
#
# NOT EXECUTED in this repository's pipeline: there is no `julia` in the build image or on the GPU
# box.  It is deliberately a thin ccall shim with no arithmetic beyond an s x s solve: every number
# comes from the library.  tests/test_binding_call_sequence.py replays, with plain ctypes and nothing
# else in between, the exact sequence of C calls each method below makes (function by function) and
# checks the results against the oracle -- that is the executable evidence for this file.
#
# lasso(), sqrtLasso(), scaledLasso!, LassoPath, CDOptions, IterLassoOptions, ProxL1, SparseIterate and
# the loss constructors are untouched: the user passes a HipMatrix where a Matrix went, and multiple
# dispatch picks the methods below (each strictly more specific than the reference's generic one).
#
#   front-end (src/lasso.jl)              methods of this file it ends up in
#   lasso(X, y, λ[, ω])        :26-53     coordinateDescent!
#   sqrtLasso(X, y, λ, ω)      :84-98     coordinateDescent!      (standardizeX=true is dead code in the
#   sqrtLasso(...; standardizeX=false)                             reference itself: `Array{T}(p)`, :73)
#   scaledLasso!(x, X, y, λ, ω) :107-144  scaledLasso!(x, X::HipMatrix, ...) below: same loop, σ from the device
#                                         (cdh_resid_moments), the residual copied back ONCE at the end
#   LassoPath(X, Y, λpath)     :229-260   LassoPath(X::HipMatrix, ...) below: same loop, nothing n-sized moves
#                                         between the λ (carried residual reused, gradient cache from the start)
# The generic front-ends still work on a HipMatrix through the operator methods (_findInitResiduals!,
# initialize!, coordinateDescent!, _stdX!), at the price of one n-sized device -> host copy per solve because
# their bodies read `f.r` on the host (lasso.jl:134); the two methods above exist to take that copy out of
# the σ loop and the λ loop.
module CoordinateDescentHIP

using CoordinateDescent, ProximalBase
using LinearAlgebra: Symmetric, eigen
using SparseArrays: nnz
using DataStructures: nlargest
import CoordinateDescent: coordinateDescent!, initialize!, gradient, descendCoordinate!,
                          numCoordinates, CDOptions, CDLeastSquaresLoss, CDSqrtLassoLoss, CDWeightedLSLoss,
                          _stdX!, _findInitResiduals!, _findLargestCorrelations,
                          scaledLasso!, LassoPath, LassoSolution, IterLassoOptions

const libcdhip = get(ENV, "LIBCDHIP", "libcdhip.so")

const CDH_OK, CDH_DIM_MISMATCH, CDH_BAD_ARG, CDH_DOMAIN = Int32(0), Int32(1), Int32(2), Int32(3)
const CDH_LS, CDH_SQRT, CDH_WLS = Int32(0), Int32(1), Int32(2)

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
# lasso(X::StridedMatrix{T}, ...) (src/lasso.jl:27); getindex is for display only.  The handle also
# holds y, r (and w) of whichever loss object used it last: `owner` IS that loss's residual vector
# (`f.r`, a fresh `copy(y)` per loss object, cd_differentiable_function.jl:54), held by reference and
# compared with `===`, so every method below can tell whether the device-side y / loss kind are still
# its own.  (A strong reference, not an objectid: the id of a Vector is its address, and the next
# `copy(y)` of the same length can land on the address of a collected one -- the handle would then
# solve against the previous y.  Holding the vector keeps it alive and unaliased; the cost is n
# elements of host memory per matrix until the next loss binds.)
mutable struct HipMatrix{T<:Union{Float32,Float64}} <: DenseMatrix{T}
  handle::Ptr{Cvoid}
  n::Int
  p::Int
  owner::Any
  # the iterate the handle holds, as last pushed to or pulled from it (support in slot order, values): an x that still
  # equals it is not pushed again -- cdh_set_iterate would declare the carried residual stale for nothing, and a
  # user-written pass over descendCoordinate! would upload p numbers per visit
  synced_idx::Vector{Int64}
  synced_val::Vector{Float64}
  synced::Bool
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

"Upload a host matrix once; every later lasso / sqrtLasso / scaledLasso! / LassoPath call on it runs in HBM."
function HipMatrix(X::StridedMatrix{T}; device::Integer=0) where {T}
  n, p = size(X)
  ref = Ref{Ptr{Cvoid}}(C_NULL)
  check(C_NULL, ccall((:cdh_create, libcdhip), Int32,
                      (Ref{Ptr{Cvoid}}, Int32, Int32, Int64, Int64, Int64, Int64, Int32),
                      ref, dtype_code(T), CDH_LS, n, n, 0, p, device))
  h = ref[]
  GC.@preserve X check(h, ccall((:cdh_set_X_cols, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64),
                                h, 0, p, X, stride(X, 2)))
  M = HipMatrix{T}(h, n, p, nothing, Int64[], Float64[], false)
  finalizer(m -> ccall((:cdh_destroy, libcdhip), Int32, (Ptr{Cvoid},), m.handle), M)
  M
end

const HipLS{T}   = CDLeastSquaresLoss{T,<:Any,<:HipMatrix{T}}
const HipSqrt{T} = CDSqrtLassoLoss{T,<:Any,<:HipMatrix{T}}
const HipWLS{T}  = CDWeightedLSLoss{T,<:Any,<:HipMatrix{T}}
const HipLoss{T} = Union{HipLS{T},HipSqrt{T},HipWLS{T}}

loss_kind(::HipLS) = CDH_LS
loss_kind(::HipSqrt) = CDH_SQRT
loss_kind(::HipWLS) = CDH_WLS

set_loss!(X::HipMatrix, kind::Int32) =
  check(X.handle, ccall((:cdh_set_loss, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, kind))

function upload_y!(X::HipMatrix{T}, y::AbstractVector{T}) where {T}
  length(y) == X.n || throw(DimensionMismatch())
  yy = y isa Vector{T} ? y : Vector{T}(y)            # contiguous host copy of a view / range
  GC.@preserve yy check(X.handle, ccall((:cdh_set_y, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), X.handle, yy))
end

function upload_w!(X::HipMatrix{T}, w::AbstractVector{T}) where {T}
  length(w) == X.n || throw(DimensionMismatch())
  ww = w isa Vector{T} ? w : Vector{T}(w)
  GC.@preserve ww check(X.handle, ccall((:cdh_set_obs_weights, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), X.handle, ww))
end

"Make the handle serve THIS loss object: its kind (the Julia type decides, never a flag given at upload
time), its y (and w).  A loss object is bound when first used and re-bound whenever another loss used
the same HipMatrix in between -- `lasso(Xh, y2, λ)` after `lasso(Xh, y1, λ)` solves against y2, and a
CDSqrtLassoLoss never runs the least-squares update.  (y mutated in place under a live loss object is
the one thing this cannot see: call `rebind!(f)`.)"
function bind!(f::HipLoss{T}) where {T}
  X = f.X
  X.owner === f.r && return f
  rebind!(f)
end
function rebind!(f::HipLoss{T}) where {T}
  X = f.X
  set_loss!(X, loss_kind(f))
  upload_y!(X, f.y)                                   # also r = copy(y), as the loss constructors do
  f isa HipWLS && upload_w!(X, f.w)
  X.owner = f.r
  f
end

numCoordinates(f::HipLoss) = f.X.p

"does the handle already hold exactly this iterate (same support order, same values)?"
function is_synced(X::HipMatrix, x::SparseIterate)
  X.synced || return false
  m = nnz(x)
  m == length(X.synced_idx) || return false
  @inbounds for i in 1:m
    (x.nzval2ind[i] == X.synced_idx[i] && Float64(x.nzval[i]) == X.synced_val[i]) || return false
  end
  true
end
function remember_iterate!(X::HipMatrix, x::SparseIterate)
  m = nnz(x)
  X.synced_idx = Int64.(x.nzval2ind[1:m]); X.synced_val = Float64.(x.nzval[1:m]); X.synced = true
  nothing
end

"x -> the handle.  `rebuild` = initialize! (r = y - X x is formed, cd_differentiable_function.jl:59-72); without it an
iterate the handle already holds is not sent again."
function push_iterate!(f::HipLoss, x::SparseIterate, rebuild::Bool)
  !rebuild && is_synced(f.X, x) && return nothing
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
  f.X.synced_idx = idx; f.X.synced_val = val; f.X.synced = true
  nothing
end

"callers read f.r afterwards (src/lasso.jl:37,134,143): one device -> host copy"
pull_residual!(f::HipLoss) =
  check(f.X.handle, ccall((:cdh_get_residual, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), f.X.handle, f.r))

"x <- the handle's iterate, support in the library's insertion order (p-sized; nothing n-sized moves)"
function pull_support!(f::HipLoss{T}, x::SparseIterate{T}) where {T}
  p = f.X.p
  idx = Vector{Int64}(undef, p); nz = Ref{Int64}(0); beta = Vector{Float64}(undef, p)
  check(f.X.handle, ccall((:cdh_get_support, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), f.X.handle, idx, nz))
  check(f.X.handle, ccall((:cdh_get_beta, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), f.X.handle, beta))
  fill!(x, zero(T))
  for i in 1:nz[]                       # re-insert in the library's support order
    x[idx[i]] = T(beta[idx[i]])
  end
  remember_iterate!(f.X, x)
  x
end

function pull_iterate!(f::HipLoss{T}, x::SparseIterate{T}) where {T}
  pull_support!(f, x)
  pull_residual!(f)
  x
end

function set_penalty!(f::HipLoss, g::ProxL1)
  om = g.λ === nothing ? Float64[] : Vector{Float64}(g.λ)    # kept alive across the call
  GC.@preserve om check(f.X.handle,
    ccall((:cdh_set_penalty, libcdhip), Int32, (Ptr{Cvoid}, Float64, Ptr{Float64}, Int64),
          f.X.handle, Float64(g.λ0), g.λ === nothing ? Ptr{Float64}(C_NULL) : pointer(om), length(om)))
end

# ---- the four-function operator interface (src/cd_differentiable_function.jl:1-35) ----------
function initialize!(f::HipLoss{T}, x::SparseIterate{T}) where {T}
  bind!(f); push_iterate!(f, x, true); pull_residual!(f)   # scaledLasso! :WarmStart reads std(f.r) next (:125-126)
  nothing
end

function gradient(f::HipLoss{T}, x::SparseIterate{T}, k::Int64) where {T}
  bind!(f); push_iterate!(f, x, false)
  out = Ref{Float64}(0)
  check(f.X.handle, ccall((:cdh_gradient, libcdhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), f.X.handle, k, out))
  T(out[])
end

"""descendCoordinate!(f, g, x, k) (cd_differentiable_function.jl:83-111) on the device.  x is refreshed (p-sized); the
residual is NOT copied back per visit -- n elements per call would be 80 MB per visit at n = 1e7 -- so inside a user-written
pass over this operator `f.r` is stale until `pull_residual!(f)` (or `initialize!`, or `coordinateDescent!`, which end
with one).  The reference's own `_cdPass!` (coordinate_descent.jl:102-107) never reads f.r between visits."""
function descendCoordinate!(f::HipLoss{T}, g::ProxL1{T}, x::SparseIterate{T}, k::Int64) where {T}
  bind!(f); set_penalty!(f, g); push_iterate!(f, x, false)
  out = Ref{Float64}(0)
  check(f.X.handle, ccall((:cdh_descend, libcdhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), f.X.handle, k, out))
  pull_support!(f, x)
  T(out[])
end

# ---- the performance path: one ccall per solve (src/coordinate_descent.jl:7-39) ---------------
"coordinateDescent! up to the point where results leave the device: x is refreshed, f.r is NOT"
function solve_resident!(x::SparseIterate{T}, f::HipLoss{T}, g::ProxL1, options::CDOptions) where {T}
  ProximalBase.numCoordinates(x) == numCoordinates(f) || throw(DimensionMismatch())
  bind!(f)
  set_penalty!(f, g)                  # checks length(g.λ) == p (coordinate_descent.jl:14-16)
  push_iterate!(f, x, false)
  opt = Ref(CdhOptions(options)); st = CdhStats()
  code = ccall((:cdh_coordinate_descent, libcdhip), Int32, (Ptr{Cvoid}, Ref{CdhOptions}, Ref{CdhStats}),
               f.X.handle, opt, st)
  check(f.X.handle, code)
  st.domain_error != 0 && throw(DomainError(NaN, "sqrt-lasso update"))
  pull_support!(f, x)
end

function coordinateDescent!(x::SparseIterate{T}, f::HipLoss{T}, g::ProxL1, options::CDOptions=CDOptions()) where {T}
  solve_resident!(x, f, g, options)
  pull_residual!(f)                   # callers read f.r next (lasso.jl:37,52,81,97)
  x
end

# ---- the dense helpers the front-ends call on X itself (src/utils.jl) ---------------------------
# Without these, LassoPath(standardizeX=true) and scaledLasso!(:Screening) would fall into the
# reference's generic loops over X[i, j] (a device round trip per element) and into BLAS / `\\` on a
# matrix that has no host memory.

"_stdX!(out, X) (utils.jl:127-138): out[j] = sqrt(sum_i X[i,j]^2 / n), one pass over X in HBM."
function _stdX!(out::Vector{T}, X::HipMatrix{T}) where {T<:AbstractFloat}
  length(out) == X.p || throw(DimensionMismatch())
  buf = Vector{Float64}(undef, X.p)
  check(X.handle, ccall((:cdh_col_rms, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), X.handle, buf))
  out .= T.(buf)
  out
end

"X'y for every column (the `mul!(storage, transpose(X), y)` of utils.jl:103): y goes to the device, r = y."
function xt_y(X::HipMatrix{T}, y::AbstractVector{T}) where {T}
  set_loss!(X, CDH_LS)
  upload_y!(X, y)
  X.owner = nothing                    # whatever loss was bound has lost its y: it re-binds on its next call
  check(X.handle, ccall((:cdh_initialize, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
                        X.handle, X.p, 0, C_NULL, C_NULL))
  X.synced = false                     # the handle's iterate is zero now, nobody's x
  out = Vector{Float64}(undef, X.p)
  check(X.handle, ccall((:cdh_xt_r, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), X.handle, out))
  out
end

"_findLargestCorrelations(X, y, s) (utils.jl:96-106): BitVector of the columns with |X_j'y| >= the s-th largest."
function _findLargestCorrelations(X::HipMatrix{T}, y::AbstractVector{T}, s::Int) where {T<:AbstractFloat}
  storage = abs.(xt_y(X, y))
  storage .>= nlargest(s, storage)[end]
end

"Xs \\ y (utils.jl:70; a QR in the reference) from the Gram block G = Xs'Xs and c = Xs'y: the minimum-norm solution of the
normal equations through the symmetric eigendecomposition of G (directions under 1e-13 of the largest eigenvalue are left
out, as a rank-revealing QR leaves them out), refined twice against the normal equations' own residual Xs'(y - Xs b) --
initialize! forms the residual on the device, cdh_xt_r_cols dots it with the s columns -- so that near-collinear
screening columns do not cost the squared condition number.  Leaves r = y - Xs b on the device."
function screening_ols!(X::HipMatrix, idx::Vector{Int64})
  m = length(idx)
  m <= 4096 || throw(ArgumentError("screening set larger than 4096 columns"))
  G = Matrix{Float64}(undef, m, m); c = Vector{Float64}(undef, m)
  check(X.handle, ccall((:cdh_gram, libcdhip), Int32,
                        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                        X.handle, m, idx, G, c, C_NULL))
  E = eigen(Symmetric(G))
  keep = E.values .> 1e-13 * max(E.values[end], 0.0)
  V = E.vectors[:, keep]
  pinvG = (V ./ E.values[keep]') * V'
  coef = pinvG * c
  res = Vector{Float64}(undef, m)
  for _ in 1:2
    check(X.handle, ccall((:cdh_initialize, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
                          X.handle, X.p, m, idx, coef))
    check(X.handle, ccall((:cdh_xt_r_cols, libcdhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}),
                          X.handle, m, idx, res))
    coef .+= pinvG * res
  end
  check(X.handle, ccall((:cdh_initialize, libcdhip), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
                        X.handle, X.p, m, idx, coef))
  X.synced = false                     # the handle's iterate is the OLS fit now, nobody's x
  coef
end

"_findInitResiduals!(X, y, s, storage) (utils.jl:65-77): storage = y - Xs (Xs \\ y), any s; the residual is formed on
the device (screening_ols!) and copied back once."
function _findInitResiduals!(X::HipMatrix{T}, y::AbstractVector{T}, s::Int, storage::Vector{T}) where {T<:AbstractFloat}
  S = _findLargestCorrelations(X, y, s)             # leaves y on the device and r = y
  screening_ols!(X, Int64.(findall(S)))
  check(X.handle, ccall((:cdh_get_residual, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), X.handle, storage))
  storage
end
# _findInitSigma!(X, y, s, storage) = std(_findInitResiduals!(X, y, s, storage)) (utils.jl:60-64) is
# generic and lands in the method above.

# ---- the σ loop and the λ loop with nothing n-sized crossing the bus (src/lasso.jl:107-144, 229-260) ------
"sum r, sum r^2 of the device-side residual (all shards)"
function resid_moments(X::HipMatrix)
  s = Ref{Float64}(0); ss = Ref{Float64}(0)
  check(X.handle, ccall((:cdh_resid_moments, libcdhip), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), X.handle, s, ss))
  s[], ss[]
end
"std(f.r) as Statistics.std computes it -- the mean, then the centred sum of squares (two passes on the device), Bessel-corrected"
function resid_std(X::HipMatrix)
  out = Ref{Float64}(0)
  check(X.handle, ccall((:cdh_resid_std, libcdhip), Int32, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}), X.handle, out, C_NULL))
  out[]
end

function gradient_cache_mode(X::HipMatrix)
  m = Ref{Int32}(0)
  check(X.handle, ccall((:cdh_get_gradient_cache, libcdhip), Int32, (Ptr{Cvoid}, Ref{Int32}), X.handle, m))
  m[]
end

"scaledLasso!(x, X, y, λ, ω, options) (lasso.jl:107-144) for a design in HBM: the reference's loop with σ taken
from the device-side residual; `f.r` comes back to the host once, for the LassoSolution."
function scaledLasso!(x::SparseIterate{T}, X::HipMatrix{T}, y::AbstractVector{T}, λ::T, ω::AbstractVector{T},
                      options::IterLassoOptions=IterLassoOptions()) where {T<:AbstractFloat}
  n = X.n
  f = CDLeastSquaresLoss(y, X)
  if options.initProcedure == :Screening
    # _findInitSigma! (utils.jl:60-64) without its copy back: scores, s x s normal equations, residual on the device
    S = _findLargestCorrelations(X, y, options.sinit)      # leaves y on the device, loss kind LS, r = y
    screening_ols!(X, Int64.(findall(S)))
    σ = T(resid_std(X))
    X.owner = f.r                     # the device holds exactly this loss's y and kind: no second upload
  elseif options.initProcedure == :InitStd
    σ = options.σinit
  elseif options.initProcedure == :WarmStart
    bind!(f); push_iterate!(f, x, true)
    σ = T(resid_std(X))
  else
    throw(ArgumentError("Incorrect initialization Symbol"))
  end
  g = ProxL1(λ * σ, ω)
  for iter = 1:options.maxIter
    solve_resident!(x, f, g, options.optionsCD)
    σnew = T(sqrt(resid_moments(X)[2] / n))                # sqrt(sum(abs2, f.r) / n), lasso.jl:134
    abs(σnew - σ) / σ < options.optTol && break
    σ = σnew
    g = ProxL1(λ * σ, ω)
  end
  pull_residual!(f)
  LassoSolution{T, typeof(g)}(x, f.r, g, T(resid_std(X)))
end

"LassoPath(X, Y, λpath, options; max_hat_s, standardizeX) (lasso.jl:229-260) for a design in HBM: one x and one f
for all λ as in the reference; warm starts keep the carried residual (it already equals Y - X x) and the gradient
cache is engaged from the first full pass for the duration of the path (whatever the caller set otherwise stays)."
function LassoPath(X::HipMatrix{T}, Y::StridedVector{T}, λpath::Vector{T}, options=CDOptions();
                   max_hat_s=Inf, standardizeX::Bool=true) where {T<:AbstractFloat}
  p = X.p
  stdX = Array{T}(undef, p)
  standardizeX ? _stdX!(stdX, X) : fill!(stdX, one(T))
  x = SparseIterate(T, p)
  f = CDLeastSquaresLoss(Y, X)
  numλ = length(λpath)
  βpath = Vector{SparseIterate{T}}(undef, numλ)
  mode_before = gradient_cache_mode(X)
  set_reuse_residual!(X, true)
  mode_before == 1 && set_gradient_cache!(X, 2)
  try
    for indλ = 1:numλ
      solve_resident!(x, f, ProxL1(λpath[indλ], stdX), options)
      βpath[indλ] = copy(x)
      if nnz(x) > max_hat_s
        resize!(λpath, indλ)
        break
      end
    end
  finally
    set_reuse_residual!(X, false)
    mode_before == 1 && set_gradient_cache!(X, 1)
  end
  LassoPath{T}(copy(λpath), βpath)
end

# Optional knobs (no reference counterpart): blocked sweep width, screened full passes, the gradient
# cache of repeated solves, hipGraph replay, reuse of the carried residual by warm starts (what LassoPath
# wants: src/lasso.jl:250-252).
set_sweep_mode!(X::HipMatrix, blocked::Bool, block::Integer=32) =
  check(X.handle, ccall((:cdh_set_sweep_mode, libcdhip), Int32, (Ptr{Cvoid}, Int32, Int32), X.handle, blocked ? 1 : 0, block))
set_screening!(X::HipMatrix, level::Integer) =
  check(X.handle, ccall((:cdh_set_screening, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, level))
"0 off, 1 (default) engages once it has paid for itself, 2 from the first full pass: ask for 2 before LassoPath"
set_gradient_cache!(X::HipMatrix, mode::Integer) =
  check(X.handle, ccall((:cdh_set_gradient_cache, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, mode))
set_use_graph!(X::HipMatrix, on::Bool) =
  check(X.handle, ccall((:cdh_set_use_graph, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, on ? 1 : 0))
set_reuse_residual!(X::HipMatrix, on::Bool) =
  check(X.handle, ccall((:cdh_set_reuse_residual, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, on ? 1 : 0))
"The one-launch solve of problems whose Gram matrix fits on chip (p ≤ 1024; on by default): the reference's own shapes"
set_onchip_solve!(X::HipMatrix, on::Bool) =
  check(X.handle, ccall((:cdh_set_onchip_solve, libcdhip), Int32, (Ptr{Cvoid}, Int32), X.handle, on ? 1 : 0))
"(solves run in one launch, Gram matrices built) on this design so far"
function onchip_stats(X::HipMatrix)
  out = zeros(Int64, 2)
  check(X.handle, ccall((:cdh_onchip_stats, libcdhip), Int32, (Ptr{Cvoid}, Ptr{Int64}), X.handle, out))
  (out[1], out[2])
end

end # module
