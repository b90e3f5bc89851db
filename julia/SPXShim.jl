# SPXShim.jl -- the reference-side binding a ShiftedProximalOperators.jl maintainer would add
# (e.g. as a package extension `ext/ShiftedProximalOperatorsSPXExt.jl` loaded when AMDGPU.jl is present).
#
# NOT EXECUTED IN THIS REPOSITORY'S PIPELINE: no Julia toolchain exists in the build image or on the GPU
# box.  It is the `ccall` counterpart of shiftedproximaloperators.jl_amd/_lib.py + shifted.py, which bind
# exactly the same C symbols (include/spx.h) and ARE exercised by tests/ on the MI355X.
#
# Idea: the reference's struct fields are type parameters `<: AbstractVector{R}` (src/shiftedNormL1Box.jl:3-21),
# so `shifted(h, xk::ROCVector{Float64}, ...)` already builds a ψ whose xk/sj/sol live in HBM -- the
# constructors need no change.  What cannot run on device arrays is the body of prox! (scalar-indexing loops).
# This file adds prox! methods that dispatch on ψ with ROCArray storage and forward to libspx.

module SPXShim

using ShiftedProximalOperators
using ShiftedProximalOperators:
  ShiftedNormL1, ShiftedNormL0, ShiftedRootNormLhalf, ShiftedNormL1Box, ShiftedNormL0Box,
  ShiftedRootNormLhalfBox, ShiftedIndBallL0, ShiftedIndBallL0BInf, ShiftedGroupNormL2, ShiftedGroupNormL2Binf
import ShiftedProximalOperators: prox!, iprox!
using AMDGPU  # ROCArray, AMDGPU.stream(), AMDGPU.device_id

const libspx = get(ENV, "LIBSPX", "libspx.so")
# Device vectors the C ABI takes as they are: a ROCVector, or a UNIT-STRIDE view of one (`view(x, 2:n)`: the reference builds
# operators on views, test/runtests.jl:196-209).  `pointer(v)` of such a view is the base pointer advanced by the offset
# (round 3: by pointer arithmetic, no copy, no CPU fallback); libspx handles vectors that start at any element (8-byte
# aligned: the kernels peel to the 16-byte boundary or take 8-byte accesses).  Strided views (`1:2:10`) still fall through to
# the reference's own method.
const DVec = Union{ROCVector{Float64}, SubArray{Float64, 1, <:ROCVector{Float64}, <:Tuple{AbstractUnitRange}, true}}

# ---------------------------------------------------------------------------------------------
# context: one per (device, HIP stream); enqueue on AMDGPU.jl's current stream so prox! is ordered with
# the solver's own broadcasts (R2 does `mν∇fk .= -ν .* ∇fk; prox!(s, ψ, mν∇fk, ν)`).
# ---------------------------------------------------------------------------------------------
const CTX = Dict{Tuple{Int, Ptr{Cvoid}}, Ptr{Cvoid}}()

# Status 7 (SPX_ERR_INTERNAL) is the device side reporting a failure of an EARLIER asynchronous call on this context (a kernel
# that synchronises inside one launch gave up waiting: another process or a graph replay held the CUs) -- libspx raises it on
# the next call, whichever it is; the results since then are NaN.  `ccall((:spx_sync, libspx), Cint, (Ptr{Cvoid},), ctx())`
# acknowledges it and resets the context.
function check(status::Cint)
  status == 0 && return
  msg = unsafe_string(ccall((:spx_last_error, libspx), Cstring, ()))
  error("libspx status $status: $msg")
end

function ctx()
  dev = AMDGPU.device_id(AMDGPU.device()) - 1
  st = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))      # hipStream_t of the task-local stream
  get!(CTX, (dev, st)) do
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:spx_ctx_create_on_stream, libspx), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, st, h))
    h[]
  end
end

dptr(v::DVec) = Ptr{Cdouble}(UInt(pointer(v)))
dptr(::Nothing) = Ptr{Cdouble}(C_NULL)
vec_or_nothing(b) = b isa Real ? nothing : b            # `isa(ψ.l, Real) ? ψ.l : ψ.l[i]`
scal(b) = b isa Real ? Float64(b) : 0.0
mptr(m) = m === nothing ? Ptr{UInt8}(C_NULL) : Ptr{UInt8}(UInt(pointer(m)))   # device byte mask or NULL
function with_value(f)                                   # entry points that return a scalar through a double*
  out = Ref{Cdouble}(0.0)
  check(f(out))
  out[]
end

# ---------------------------------------------------------------------------------------------
# separable, unboxed          src/shiftedNormL1.jl:40-54, shiftedNormL0.jl:38-55, shiftedRootNormLhalf.jl:41-63
# ---------------------------------------------------------------------------------------------
for (T, sym) in ((:ShiftedNormL1, :spx_prox_l1), (:ShiftedNormL0, :spx_prox_l0),
                 (:ShiftedRootNormLhalf, :spx_prox_lhalf))
  @eval function prox!(y::DVec, ψ::$T{Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64)
    n = length(ψ.xk)
    (length(y) == n && length(q) == n) || throw(BoundsError())
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble),
                ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ, σ))
    return y
  end
end

# ---------------------------------------------------------------------------------------------
# separable, boxed            src/shiftedNormL1Box.jl:89-125, shiftedNormL0Box.jl:89-131,
#                             shiftedRootNormLhalfBox.jl:86-120
# `selected` (any AbstractArray{<:Integer}, 1-based) -> device byte mask, built once per ψ and cached;
# the default 1:length(xk) needs no mask.
# ---------------------------------------------------------------------------------------------
const MASKS = IdDict{Any, Any}()   # ψ.selected (shared by a twice-shifted ψ) => ROCVector{UInt8} | nothing

function mask_for(ψ)
  n = length(ψ.xk)
  sel = ψ.selected
  (sel isa AbstractUnitRange && first(sel) <= 1 && last(sel) >= n) && return nothing
  get!(MASKS, sel) do
    idx0 = ROCVector{Int64}(collect(Int64, sel) .- 1)            # 0-based for the C ABI
    m = ROCVector{UInt8}(undef, n)
    check(ccall((:spx_build_mask, libspx), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64, Ptr{Int64}, Int64),
                ctx(), Ptr{UInt8}(UInt(pointer(m))), n, Ptr{Int64}(UInt(pointer(idx0))), length(idx0)))
    AMDGPU.synchronize()                                         # idx0 may be freed after this
    m
  end
end

for (T, sym) in ((:ShiftedNormL1Box, :spx_prox_l1_box), (:ShiftedNormL0Box, :spx_prox_l0_box),
                 (:ShiftedRootNormLhalfBox, :spx_prox_lhalf_box))
  @eval function prox!(y::DVec, ψ::$T{Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64)
    n = length(ψ.xk)
    (length(y) == n && length(q) == n) || throw(BoundsError())
    (ψ.l isa Real || ψ.l isa DVec) && (ψ.u isa Real || ψ.u isa DVec) ||
      return invoke(prox!, Tuple{AbstractVector{Float64}, $T{Float64}, AbstractVector{Float64}, Float64}, y, ψ, q, σ)
    m = mask_for(ψ)
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble,
                 Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{UInt8}),
                ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ, σ,
                dptr(vec_or_nothing(ψ.l)), dptr(vec_or_nothing(ψ.u)), scal(ψ.l), scal(ψ.u),
                mptr(m)))
    return y
  end
end

# ---------------------------------------------------------------------------------------------
# iprox!                      src/shiftedNormL1.jl:60-75, shiftedNormL0.jl:61-80, shiftedNormL1Box.jl:131-225,
#                             shiftedNormL0Box.jl:137-231
# ---------------------------------------------------------------------------------------------
for (T, sym) in ((:ShiftedNormL1, :spx_iprox_l1), (:ShiftedNormL0, :spx_iprox_l0))
  @eval function iprox!(y::DVec, ψ::$T{Float64, <:DVec, <:DVec, <:DVec}, g::DVec, d::DVec)
    n = length(ψ.xk)
    st = ccall(($(QuoteNode(sym)), libspx), Cint,
               (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cint),
               ctx(), dptr(y), dptr(g), dptr(d), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ, 1)
    st == 6 && throw(AssertionError("d[i] > 0"))   # SPX_ERR_ASSERT
    check(st)
    return y
  end
end
for (T, sym) in ((:ShiftedNormL1Box, :spx_iprox_l1_box), (:ShiftedNormL0Box, :spx_iprox_l0_box))
  @eval function iprox!(y::DVec, ψ::$T{Float64, <:DVec, <:DVec, <:DVec}, g::DVec, d::DVec)
    n = length(ψ.xk)
    m = mask_for(ψ)
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble,
                 Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{UInt8}),
                ctx(), dptr(y), dptr(g), dptr(d), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ,
                dptr(vec_or_nothing(ψ.l)), dptr(vec_or_nothing(ψ.u)), scal(ψ.l), scal(ψ.u),
                mptr(m)))
    return y
  end
end

# ---------------------------------------------------------------------------------------------
# top-r                       src/shiftedIndBallL0.jl:54-72, shiftedIndBallL0BInf.jl:73-95
# (ψ.p, the Vector{Int} permutation scratch, is not used: libspx selects, it does not sort)
# ---------------------------------------------------------------------------------------------
function prox!(y::DVec, ψ::ShiftedIndBallL0{<:Integer, Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64)
  n = length(ψ.xk)
  check(ccall((:spx_prox_indball_l0, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64),
              ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.r))
  return y
end

function prox!(y::DVec, ψ::ShiftedIndBallL0BInf{<:Integer, Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64)
  n = length(ψ.xk)
  check(ccall((:spx_prox_indball_l0_binf, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Cdouble),
              ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.r, ψ.Δ))
  return y
end

# ---------------------------------------------------------------------------------------------
# the unshifted value types the reference defines itself: prox!(y, h, x, γ) and h(x) on device arrays
#   RootNormLhalf  src/rootNormLhalf.jl:27-51   (returns h(y)),   GroupNormL2  src/groupNormL2.jl:33-58
# They run the shifted kernels with xk = sj = 0 (q + (0 + 0) = q and y - (0 + 0) = y are exact).
# ---------------------------------------------------------------------------------------------
const ZEROS = Dict{Int, ROCVector{Float64}}()
zeros_for(n) = get!(() -> AMDGPU.zeros(Float64, n), ZEROS, n)

function prox!(y::DVec, f::ShiftedProximalOperators.RootNormLhalf{Float64}, x::DVec, γ::Real = 1.0)
  n = length(x); z = zeros_for(n); out = Ref{Cdouble}(0.0)
  check(ccall((:spx_proxval_lhalf, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}),
              ctx(), dptr(y), dptr(x), dptr(z), dptr(z), n, f.lambda, Float64(γ), 1.0, out))
  return out[]                                                   # λ Σ sqrt|y_i|, as :50
end
function (f::ShiftedProximalOperators.RootNormLhalf{Float64})(x::DVec)
  z = zeros_for(length(x))
  with_value() do out
    ccall((:spx_obj_lhalf, libspx), Cint,
          (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Cdouble}),
          ctx(), dptr(x), dptr(z), dptr(z), length(x), f.lambda, out)
  end
end
# GroupNormL2: prox!(y, f, x, γ) = group_call on ψ = shifted(f, zeros) with q = x; its return value Σ λ_g ‖x_g‖ is
# spx_obj_group_l2 of the input (layout_for(f, n) supplies offsets / gather indices / weights, as in group_call).

# ---------------------------------------------------------------------------------------------
# prox! fused with h at the result (no counterpart in the reference; what R2 does in two steps:
# `prox!(s, ψ, mν∇fk, ν)` then `hkn = ψ(s)`): one pass over the vectors instead of two.  Headline operator shown;
# spx_proxval_l1 / l0 / lhalf / l0_box / lhalf_box follow the same pattern.
# ---------------------------------------------------------------------------------------------
function prox_value!(y::DVec, ψ::ShiftedNormL1Box{Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64;
                     q_scale::Float64 = 1.0)   # prox at q_scale .* q (R2: q = ∇fk, q_scale = -ν)
  n = length(ψ.xk)
  (length(y) == n && length(q) == n) || throw(BoundsError())
  m = mask_for(ψ)
  out = Ref{Cdouble}(0.0)
  check(ccall((:spx_proxval_l1_box, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble,
               Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{UInt8}, Cdouble, Ptr{Cdouble}),
              ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ, σ,
              dptr(vec_or_nothing(ψ.l)), dptr(vec_or_nothing(ψ.u)), scal(ψ.l), scal(ψ.u),
              mptr(m), q_scale, out))
  return y, out[]
end

# ---------------------------------------------------------------------------------------------
# l1 norm + l2-ball trust region      src/shiftedNormL1B2.jl:50-67   (χ = NormL2(χ.lambda))
# ---------------------------------------------------------------------------------------------
function prox!(y::DVec, ψ::ShiftedProximalOperators.ShiftedNormL1B2{Float64, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64)
  n = length(ψ.xk)
  (length(y) == n && length(q) == n) || throw(BoundsError())
  check(ccall((:spx_prox_l1_b2, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble, Cdouble, Cdouble),
              ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ψ.λ, σ, ψ.Δ, ψ.χ.lambda))
  return y
end

# ---------------------------------------------------------------------------------------------
# groups                      src/shiftedGroupNormL2.jl:52-79, shiftedGroupNormL2Binf.jl:67-119
# ψ.h.idx as consecutive contiguous ranges (UnitRanges or [:]) takes the CSR / uniform entry points; any other
# index sets (Vector{Int}s, gaps, overlaps, any order) take the gather entry points, which reproduce the reference's
# sequential loop literally.  Offsets / indices / weights are uploaded once per h and cached.
# ---------------------------------------------------------------------------------------------
const LAYOUTS = IdDict{Any, Any}()

function layout_for(h, n)
  get!(LAYOUTS, h) do
    # `[:]` is the whole vector; an explicit index vector that is a contiguous run (`collect(4:6)`, runtests.jl:290) is a range
    asrange(g) = g isa Colon ? (1:n) :
                 (g isa AbstractVector{<:Integer} && !(g isa AbstractUnitRange) && !isempty(g) &&
                  all(i -> g[i + 1] - g[i] == 1, 1:(length(g) - 1))) ? (first(g):last(g)) : g
    rngs = map(asrange, h.idx)
    lam = ROCVector{Float64}(collect(Float64, h.lambda))
    if !(all(g -> g isa AbstractUnitRange, rngs) &&
         all(i -> first(rngs[i + 1]) == last(rngs[i]) + 1, 1:(length(rngs) - 1)))
      ptr = Int64[0; cumsum(Int64[length(g) for g in rngs])]       # gather form: 0-based ptr / index
      index = Int64[i - 1 for g in rngs for i in g]
      all(i -> 0 <= i < n, index) || throw(BoundsError())
      return (gather = true, ptr = ROCVector{Int64}(ptr), index = ROCVector{Int64}(index), nnz = length(index),
              ngroups = length(rngs), lambda = lam)
    end
    off = Int64[first(rngs[1]) - 1; [last(g) for g in rngs]]      # 0-based CSR offsets
    sizes = diff(off)
    uniform = off[1] == 0 && off[end] == n && all(==(sizes[1]), sizes)
    # with offsets, gsize is the size bound that lets libspx use its register-tile / LDS-resident kernels
    (gather = false, offsets = uniform ? nothing : ROCVector{Int64}(off), gsize = uniform ? sizes[1] : maximum(sizes),
     ngroups = length(rngs), lambda = lam)
  end
end

function group_call(y, ψ, q, σ, extra...)   # extra = (Δ,) for the Binf form
  n = length(ψ.xk)
  L = layout_for(ψ.h, n)
  if L.gather
    ip(v) = Ptr{Int64}(UInt(pointer(v)))
    if isempty(extra)
      check(ccall((:spx_prox_group_l2_gather, libspx), Cint,
                  (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Ptr{Int64}, Int64,
                   Int64, Ptr{Cdouble}, Cdouble),
                  ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ip(L.ptr), ip(L.index), L.ngroups, L.nnz,
                  dptr(L.lambda), σ))
    else
      check(ccall((:spx_prox_group_l2_binf_gather, libspx), Cint,
                  (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Ptr{Int64}, Int64,
                   Int64, Ptr{Cdouble}, Cdouble, Cdouble),
                  ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, ip(L.ptr), ip(L.index), L.ngroups, L.nnz,
                  dptr(L.lambda), σ, extra[1]))
    end
    return y
  end
  offp = L.offsets === nothing ? Ptr{Int64}(C_NULL) : Ptr{Int64}(UInt(pointer(L.offsets)))
  if isempty(extra)
    check(ccall((:spx_prox_group_l2, libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Int64, Int64,
                 Ptr{Cdouble}, Cdouble),
                ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, offp, L.gsize, L.ngroups, dptr(L.lambda), σ))
  else
    check(ccall((:spx_prox_group_l2_binf, libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Int64, Int64,
                 Ptr{Cdouble}, Cdouble, Cdouble),
                ctx(), dptr(y), dptr(q), dptr(ψ.xk), dptr(ψ.sj), n, offp, L.gsize, L.ngroups, dptr(L.lambda), σ,
                extra[1]))
  end
  return y
end

function prox!(y::DVec, ψ::ShiftedGroupNormL2{Float64, RR, I, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64) where {RR, I}
  group_call(y, ψ, q, σ)
end

function prox!(y::DVec, ψ::ShiftedGroupNormL2Binf{Float64, RR, I, <:DVec, <:DVec, <:DVec}, q::DVec, σ::Float64) where {RR, I}
  group_call(y, ψ, q, σ, ψ.Δ)
end

# ---------------------------------------------------------------------------------------------
# ψ(y)                        src/ShiftedProximalOperators.jl:51-54, shiftedNormL1Box.jl:70-82 (idem L0Box, L½Box),
#                             shiftedIndBallL0BInf.jl:44-49, shiftedGroupNormL2Binf.jl:34-39
# ---------------------------------------------------------------------------------------------
# (ccall needs a literal symbol and a literal argument-type tuple: every entry point is spelled out)
for (T, sym) in ((:ShiftedNormL1, :spx_obj_l1), (:ShiftedNormL0, :spx_obj_l0), (:ShiftedRootNormLhalf, :spx_obj_lhalf))
  @eval (ψ::$T{Float64, <:DVec, <:DVec, <:DVec})(y::DVec) = with_value() do out
    ccall(($(QuoteNode(sym)), libspx), Cint,
          (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Cdouble}),
          ctx(), dptr(y), dptr(ψ.xk), dptr(ψ.sj), length(y), ψ.λ, out)
  end
end
for (T, sym) in ((:ShiftedNormL1Box, :spx_obj_l1_box), (:ShiftedNormL0Box, :spx_obj_l0_box),
                 (:ShiftedRootNormLhalfBox, :spx_obj_lhalf_box))
  @eval function (ψ::$T{Float64, <:DVec, <:DVec, <:DVec})(y::DVec)
    m = mask_for(ψ)
    with_value() do out
      ccall(($(QuoteNode(sym)), libspx), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble,
             Cdouble, Ptr{UInt8}, Ptr{Cdouble}),
            ctx(), dptr(y), dptr(ψ.xk), dptr(ψ.sj), length(y), ψ.λ, dptr(vec_or_nothing(ψ.l)),
            dptr(vec_or_nothing(ψ.u)), scal(ψ.l), scal(ψ.u), mptr(m), out)
    end
  end
end
(ψ::ShiftedIndBallL0{<:Integer, Float64, <:DVec, <:DVec, <:DVec})(y::DVec) = with_value() do out
  ccall((:spx_obj_indball_l0, libspx), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Ptr{Cdouble}),
        ctx(), dptr(y), dptr(ψ.xk), dptr(ψ.sj), length(y), ψ.r, out)
end
(ψ::ShiftedProximalOperators.ShiftedNormL1B2{Float64, <:DVec, <:DVec, <:DVec})(y::DVec) = with_value() do out  # src/shiftedNormL1B2.jl:32
  ccall((:spx_obj_l1_b2, libspx), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble, Ptr{Cdouble}),
        ctx(), dptr(y), dptr(ψ.xk), dptr(ψ.sj), length(y), ψ.λ, ψ.Δ, out)
end
(ψ::ShiftedIndBallL0BInf{<:Integer, Float64, <:DVec, <:DVec, <:DVec})(y::DVec) = with_value() do out
  ccall((:spx_obj_indball_l0_binf, libspx), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Cdouble, Ptr{Cdouble}),
        ctx(), dptr(y), dptr(ψ.xk), dptr(ψ.sj), length(y), ψ.r, ψ.Δ, out)
end
# (group forms: spx_obj_group_l2 / spx_obj_group_l2_binf with the layout_for(ψ.h, n) arguments, as in group_call)


# ---------------------------------------------------------------------------------------------
# Float32 (round 2): the reference is generic in R <: Real; spx_prox_*_f32 reproduce the NormL1 / NormL0 bodies bit for bit
# in fp32, on contiguous views that start at any element (src/shiftedNormL1Box.jl:89-94, test/runtests.jl:196-209).
# RootNormLhalf{Float32} keeps the reference's own method (its body computes in Float64).
# ---------------------------------------------------------------------------------------------
const DVec32 = Union{ROCVector{Float32}, SubArray{Float32, 1, <:ROCVector{Float32}, <:Tuple{AbstractUnitRange}, true}}
dptr32(v) = v === nothing ? Ptr{Cfloat}(C_NULL) : Ptr{Cfloat}(UInt(pointer(v)))
vec32_or_nothing(b) = b isa Real ? nothing : b
scal32(b) = b isa Real ? Float32(b) : 0.0f0

for (T, sym) in ((:ShiftedNormL1, :spx_prox_l1_f32), (:ShiftedNormL0, :spx_prox_l0_f32))
  @eval function prox!(y::DVec32, ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32}, q::DVec32, σ::Float32)
    n = length(ψ.xk)
    (length(y) == n && length(q) == n) || throw(BoundsError())
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat, Cfloat),
                ctx(), dptr32(y), dptr32(q), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.λ, σ))
    return y
  end
end
for (T, sym) in ((:ShiftedNormL1Box, :spx_prox_l1_box_f32), (:ShiftedNormL0Box, :spx_prox_l0_box_f32))
  @eval function prox!(y::DVec32, ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32}, q::DVec32, σ::Float32)
    n = length(ψ.xk)
    (length(y) == n && length(q) == n) || throw(BoundsError())
    (ψ.l isa Real || ψ.l isa DVec32) && (ψ.u isa Real || ψ.u isa DVec32) ||
      return invoke(prox!, Tuple{AbstractVector{Float32}, $T{Float32}, AbstractVector{Float32}, Float32}, y, ψ, q, σ)
    m = mask_for(ψ)
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat, Cfloat,
                 Ptr{Cfloat}, Ptr{Cfloat}, Cfloat, Cfloat, Ptr{UInt8}),
                ctx(), dptr32(y), dptr32(q), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.λ, σ,
                dptr32(vec32_or_nothing(ψ.l)), dptr32(vec32_or_nothing(ψ.u)), scal32(ψ.l), scal32(ψ.u), mptr(m)))
    return y
  end
end

# iprox! on Float32 vectors (round 3: spx_iprox_*_f32; src/shiftedNormL1.jl:60-75, shiftedNormL0.jl:61-80, shiftedNormL1Box.jl:131-225,
# shiftedNormL0Box.jl:137-231).  check_d = 1: the reference's `@assert d[i] > 0` (status 6 -> AssertionError).
for (T, sym) in ((:ShiftedNormL1, :spx_iprox_l1_f32), (:ShiftedNormL0, :spx_iprox_l0_f32))
  @eval function iprox!(y::DVec32, ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32}, g::DVec32, d::DVec32)
    n = length(ψ.xk)
    (length(y) == n && length(g) == n && length(d) == n) || throw(BoundsError())
    st = ccall(($(QuoteNode(sym)), libspx), Cint,
               (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat, Cint),
               ctx(), dptr32(y), dptr32(g), dptr32(d), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.λ, 1)
    st == 6 && throw(AssertionError("d[i] > 0"))
    check(st)
    return y
  end
end
for (T, sym) in ((:ShiftedNormL1Box, :spx_iprox_l1_box_f32), (:ShiftedNormL0Box, :spx_iprox_l0_box_f32))
  @eval function iprox!(y::DVec32, ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32}, g::DVec32, d::DVec32)
    n = length(ψ.xk)
    (length(y) == n && length(g) == n && length(d) == n) || throw(BoundsError())
    (ψ.l isa Real || ψ.l isa DVec32) && (ψ.u isa Real || ψ.u isa DVec32) ||
      return invoke(iprox!, Tuple{AbstractVector{Float32}, $T{Float32}, AbstractVector{Float32}, AbstractVector{Float32}}, y, ψ, g, d)
    m = mask_for(ψ)
    check(ccall(($(QuoteNode(sym)), libspx), Cint,
                (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat,
                 Ptr{Cfloat}, Ptr{Cfloat}, Cfloat, Cfloat, Ptr{UInt8}),
                ctx(), dptr32(y), dptr32(g), dptr32(d), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.λ,
                dptr32(vec32_or_nothing(ψ.l)), dptr32(vec32_or_nothing(ψ.u)), scal32(ψ.l), scal32(ψ.u), mptr(m)))
    return y
  end
end

# ψ(y) on Float32 vectors (spx_obj_*_f32): element operations in Float32 as in the reference, the sum in Float64, the result
# rounded to the Float32 the reference returns.  test/runtests.jl:196-209, 268-282, 346-360, 397-412, 524-550.
for (T, sym) in ((:ShiftedNormL1, :spx_obj_l1_f32), (:ShiftedNormL0, :spx_obj_l0_f32), (:ShiftedRootNormLhalf, :spx_obj_lhalf_f32))
  @eval (ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32})(y::DVec32) = Float32(with_value() do out
    ccall(($(QuoteNode(sym)), libspx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat, Ptr{Cdouble}),
          ctx(), dptr32(y), dptr32(ψ.xk), dptr32(ψ.sj), length(y), ψ.λ, out)
  end)
end
for (T, sym) in ((:ShiftedNormL1Box, :spx_obj_l1_box_f32), (:ShiftedNormL0Box, :spx_obj_l0_box_f32),
                 (:ShiftedRootNormLhalfBox, :spx_obj_lhalf_box_f32))
  @eval function (ψ::$T{Float32, <:DVec32, <:DVec32, <:DVec32})(y::DVec32)
    m = mask_for(ψ)
    v = with_value() do out
      ccall(($(QuoteNode(sym)), libspx), Cint,
            (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cfloat, Ptr{Cfloat}, Ptr{Cfloat}, Cfloat, Cfloat, Ptr{UInt8},
             Ptr{Cdouble}),
            ctx(), dptr32(y), dptr32(ψ.xk), dptr32(ψ.sj), length(y), ψ.λ, dptr32(vec32_or_nothing(ψ.l)),
            dptr32(vec32_or_nothing(ψ.u)), scal32(ψ.l), scal32(ψ.u), mptr(m), out)
    end
    return isinf(v) ? Inf : Float32(v)   # (the reference returns the Float64 literal Inf on an infeasible point, :78)
  end
end
(ψ::ShiftedIndBallL0{<:Integer, Float32, <:DVec32, <:DVec32, <:DVec32})(y::DVec32) = with_value() do out
  ccall((:spx_obj_indball_l0_f32, libspx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Ptr{Cdouble}),
        ctx(), dptr32(y), dptr32(ψ.xk), dptr32(ψ.sj), length(y), ψ.r, out)
end
(ψ::ShiftedIndBallL0BInf{<:Integer, Float32, <:DVec32, <:DVec32, <:DVec32})(y::DVec32) = with_value() do out
  ccall((:spx_obj_indball_l0_binf_f32, libspx), Cint,
        (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Cfloat, Ptr{Cdouble}),
        ctx(), dptr32(y), dptr32(ψ.xk), dptr32(ψ.sj), length(y), ψ.r, ψ.Δ, out)
end
# (group forms: spx_obj_group_l2_f32 / spx_obj_group_l2_binf_f32 with the layout_for(ψ.h, n) arguments and a Float32 λ vector)

# top-r on Float32 vectors (round 3: spx_prox_indball_l0[_binf]_f32; src/shiftedIndBallL0.jl:54-72, shiftedIndBallL0BInf.jl:73-95
# with R = Float32): v = (xk + sj) + q, the magnitude order and the final subtraction / clamp are Float32 operations.  One launch
# with the vector on chip up to 2^23 elements.
function prox!(y::DVec32, ψ::ShiftedIndBallL0{<:Integer, Float32, <:DVec32, <:DVec32, <:DVec32}, q::DVec32, σ::Float32)
  n = length(ψ.xk)
  (length(y) == n && length(q) == n) || throw(BoundsError())
  check(ccall((:spx_prox_indball_l0_f32, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64),
              ctx(), dptr32(y), dptr32(q), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.r))
  return y
end
function prox!(y::DVec32, ψ::ShiftedIndBallL0BInf{<:Integer, Float32, <:DVec32, <:DVec32, <:DVec32}, q::DVec32, σ::Float32)
  n = length(ψ.xk)
  (length(y) == n && length(q) == n) || throw(BoundsError())
  check(ccall((:spx_prox_indball_l0_binf_f32, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int64, Cfloat),
              ctx(), dptr32(y), dptr32(q), dptr32(ψ.xk), dptr32(ψ.sj), n, ψ.r, ψ.Δ))
  return y
end

# ShiftedGroupNormL2 on Float32 vectors, contiguous groups (round 3: spx_prox_group_l2_f32; src/shiftedGroupNormL2.jl:52-79 with
# R = Float32; index sets keep the reference's method).  λ travels as a Float32 device vector, cached per h (and per content).
# (ADVICE r3: the cache is keyed on the lambda ARRAY (mutable, so it can be held weakly: entries die with it) and on a hash of
#  its contents -- `h.lambda .= ...` in a continuation loop must not leave a stale device copy behind)
const LAMBDA32 = WeakKeyDict{Any, Any}()
function lambda32_for(h)
  lam = h.lambda
  key = hash(lam)
  ent = get(LAMBDA32, lam, nothing)
  if ent === nothing || ent[1] != key
    ent = (key, ROCVector{Float32}(collect(Float32, lam)))
    LAMBDA32[lam] = ent
  end
  return ent[2]
end
function prox!(y::DVec32, ψ::ShiftedGroupNormL2{Float32, <:Any, <:Any, <:DVec32, <:DVec32, <:DVec32}, q::DVec32, σ::Float32)
  n = length(ψ.xk)
  (length(y) == n && length(q) == n) || throw(BoundsError())
  L = layout_for(ψ.h, n)
  L.gather && return invoke(prox!, Tuple{AbstractVector{Float32}, typeof(ψ), AbstractVector{Float32}, Float32}, y, ψ, q, σ)  # index sets: the reference's own method
  lam32 = lambda32_for(ψ.h)
  offp = L.offsets === nothing ? Ptr{Int64}(C_NULL) : Ptr{Int64}(UInt(pointer(L.offsets)))
  check(ccall((:spx_prox_group_l2_f32, libspx), Cint,
              (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Ptr{Int64}, Int64, Int64, Ptr{Cfloat}, Cfloat),
              ctx(), dptr32(y), dptr32(q), dptr32(ψ.xk), dptr32(ψ.sj), n, offp, L.gsize, L.ngroups, dptr32(lam32), σ))
  return y
end

# ---------------------------------------------------------------------------------------------
# Device-resident values (round 2): `device_values(out::ROCVector{Float64}) do ... end` -- inside the block ψ(y) and prox_value!
# store their Float64 result in out[1] and return NaN; nothing is read back (spx_ctx_set_value_target).
# ---------------------------------------------------------------------------------------------
function device_values(f, out::ROCVector{Float64})
  check(ccall((:spx_ctx_set_value_target, libspx), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ctx(), dptr(out)))
  try
    return f()
  finally
    check(ccall((:spx_ctx_set_value_target, libspx), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ctx(), Ptr{Cdouble}(C_NULL)))
  end
end

# shift!, set_radius!, set_bounds!, prox (src/ShiftedProximalOperators.jl:72-111,189-190) need no methods:
# they are broadcasts / field updates on the stored (device) arrays and already work on ROCArrays.
# The Box constructors' `any(l .> u)` (src/shiftedNormL1Box.jl:33-35) is a device reduction via broadcasting.

end # module
