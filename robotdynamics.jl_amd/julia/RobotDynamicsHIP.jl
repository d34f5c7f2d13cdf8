# RobotDynamicsHIP.jl -- the Julia host side of librdhip: RobotDynamics.jl's batched dynamics / Jacobian operators over whole packed
# trajectories on MI355X, one `ccall` per batch.  Thin by design: argument checks and pointer plumbing, nothing numerical.
#
# STATUS: written against include/rdhip.h; NOT executed in the build image (no Julia toolchain there, see INTEGRATION.md).  What IS checked
# mechanically in that image (tests/test_julia_binding.py): every `rdhip_*` entry point of the header has a `ccall` here, with the header's
# argument count and C-compatible argument types.  The Python module `rdhip` next to this file is the same host mirror and is what the
# test-suite runs on the GPU.
#
# Two memory spaces, one set of methods:
#   * host   : plain `Array`s (`PackedTrajectory`, `Array{T,3}` Jacobians).  The library stages them through a per-context device workspace
#              and returns when the results are back (RDHIP_HOST).  Convenient, PCIe-bound (a 65 536-knot Satellite batch returns 136 MB).
#   * device : `DeviceArray`s (`DeviceTrajectory`, `DeviceArray{T,3}` Jacobians) allocated with rdhip_malloc and filled with
#              `copyto!`.  Calls are asynchronous on the context's HIP stream (RDHIP_DEVICE), results stay in HBM for the next call,
#              iterations can be captured into a HIP graph (`capture` / `launch`).  This is the path the benchmark numbers refer to.
#
#     using RobotDynamics, RobotDynamicsHIP
#     const RD = RobotDynamics; const RH = RobotDynamicsHIP
#     model  = RH.Satellite()                                   # <: RD.RigidBody{QuatRotation{Float64}}
#     dmodel = RD.DiscretizedDynamics{RD.RK4}(model)
#     Z  = RH.PackedTrajectory{13,6}(data, t, dt)               # (n+m) x N Matrix + t, dt; Z[k] is a KnotPoint over a view
#     Zd = RH.device(Z)                                         # the same trajectory resident in HBM
#     Jd = RH.DeviceArray{Float64}(undef, 13, 19, length(Z)); Yd = RH.DeviceArray{Float64}(undef, 13, length(Z))
#     RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dmodel, Jd, Yd, Zd)     # ONE asynchronous kernel launch for all k
#     Gd = RH.DeviceArray{Float64}(undef, 13, 12, length(Z)); RD.errstate_jacobian!(model, Gd, Zd; clear=true)
#     J = Array(Jd)                                             # synchronises and copies back
module RobotDynamicsHIP

using RobotDynamics
using ForwardDiff, FiniteDiff          # RD.@autodiff needs both in the local module (src/jacobian_gen.jl:21-24)
using Rotations
using StaticArrays
using LinearAlgebra
const RD = RobotDynamics

const librdhip = get(ENV, "RDHIP_LIB", joinpath(@__DIR__, "..", "lib", "librdhip.so"))

# ------------------------------------------------------------------------------------------------ enums of include/rdhip.h
const MODEL_CARTPOLE, MODEL_DOUBLE_INTEGRATOR, MODEL_LINEAR, MODEL_SATELLITE, MODEL_QUADROTOR, MODEL_SATELLITE_PAIR, MODEL_LIE_SATELLITE, MODEL_CUSTOM = 0:7
const ROT_QUAT, ROT_MRP, ROT_RP = 0:2
const FRAME_WORLD, FRAME_BODY = 0:1
const CONTINUOUS, EULER, RULE_RK2, RULE_RK3, RULE_RK4, IMPLICIT_MIDPOINT, DISCRETE = -1:5
const FORWARD_AD, FINITE_DIFF, USER_DEFINED = 0:2
const F64, F32 = 0:1
const DEVICE, HOST = 0:1
const STREAM_OWN, STREAM_BORROW = 0:1
const TRAJECTORY_MAJOR, TIME_MAJOR = 0:1
const GATHER_NONE, GATHER_ROOT, GATHER_ALL = 0:2
const COMM_ID_BYTES = 128

struct RdhipError <: Exception
    code::Int
    msg::String
end
Base.showerror(io::IO, e::RdhipError) = print(io, "rdhip error ", e.code, ": ", e.msg)

version() = Int(ccall((:rdhip_version, librdhip), Cint, ()))
function device_count()
    n = Ref{Cint}(0)
    ccall((:rdhip_device_count, librdhip), Cint, (Ref{Cint},), n)
    return Int(n[])
end

# ------------------------------------------------------------------------------------------------ contexts
"""
    Context(device=0; stream=C_NULL, borrow=false)

One HIP stream on one GPU (`rdhip_ctx`).  Not thread-safe: one context per task (the reference's model objects carry mutable caches and
are likewise not thread-safe, src/integration.jl:258-276).  `borrow=true` launches on a caller-owned `hipStream_t`.
"""
mutable struct Context
    ptr::Ptr{Cvoid}
    device::Int
    function Context(device::Integer=0; stream::Ptr{Cvoid}=C_NULL, borrow::Bool=false)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_ctx_create, librdhip), Cint, (Cint, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), device, stream, borrow ? STREAM_BORROW : STREAM_OWN, out)
        rc == 0 || throw(RdhipError(rc, last_error(C_NULL)))
        ctx = new(out[], Int(device))
        finalizer(c -> ccall((:rdhip_ctx_destroy, librdhip), Cint, (Ptr{Cvoid},), c.ptr), ctx)
        return ctx
    end
end
last_error(p::Ptr{Cvoid}) = unsafe_string(ccall((:rdhip_last_error, librdhip), Cstring, (Ptr{Cvoid},), p))
last_error(ctx::Context) = last_error(ctx.ptr)
"demangled name of the kernel this host thread launched last through the library (which specialisation a call was dispatched to; `@which` for kernels)"
function last_kernel_name()
    buf = zeros(UInt8, 1024)
    rc = GC.@preserve buf ccall((:rdhip_last_kernel_name, librdhip), Cint, (Ptr{UInt8}, Cint), pointer(buf), 1024)
    rc == 0 || throw(RdhipError(rc, last_error(C_NULL)))
    return GC.@preserve buf unsafe_string(pointer(buf))
end

"""
    tune(pairs::AbstractString)

Merge `"key=value,..."` into the dispatcher's tuning knobs (`rdhip_tune`; `"reset"`: back to what the environment variable `RDHIP_TUNE` says).
For tests and tuning scripts: no knob changes a result; `include/rdhip.h` lists the keys.
"""
function tune(pairs::AbstractString)
    rc = ccall((:rdhip_tune, librdhip), Cint, (Cstring,), pairs)
    rc == 0 || throw(RdhipError(rc, last_error(C_NULL)))
    return nothing
end

const _default_ctx = Ref{Union{Nothing,Context}}(nothing)
function default_context()
    if _default_ctx[] === nothing
        _default_ctx[] = Context()
    end
    return _default_ctx[]::Context
end

# status -> the exception the reference throws in the same situation (include/rdhip.h, rdhip_status)
function check(ctx::Union{Context,Nothing}, rc::Integer)
    rc == 0 && return nothing
    msg = last_error(ctx === nothing ? C_NULL : ctx.ptr)
    rc == -1 && throw(ArgumentError(msg))
    (rc == -2 || rc == -3) && throw(RD.NotImplementedError(msg))          # src/utils.jl:1-8
    rc == -6 && throw(DimensionMismatch(msg))
    throw(RdhipError(rc, msg))
end

synchronize(ctx::Context=default_context()) = check(ctx, ccall((:rdhip_ctx_synchronize, librdhip), Cint, (Ptr{Cvoid},), ctx.ptr))
"launch overlap: consecutive independent `jacobian!` calls on `ctx` alternate between two streams (ordering rule: include/rdhip.h)"
set_overlap!(ctx::Context, on::Bool=true) = check(ctx, ccall((:rdhip_ctx_set_overlap, librdhip), Cint, (Ptr{Cvoid}, Cint), ctx.ptr, on ? 1 : 0))
"order the context's stream behind everything issued so far (overlap mode keeps part of it on an internal stream)"
fence(ctx::Context=default_context()) = check(ctx, ccall((:rdhip_ctx_fence, librdhip), Cint, (Ptr{Cvoid},), ctx.ptr))
stream(ctx::Context) = ccall((:rdhip_ctx_stream, librdhip), Ptr{Cvoid}, (Ptr{Cvoid},), ctx.ptr)

# HIP-event stopwatch on the context's stream
timer_start(ctx::Context=default_context()) = check(ctx, ccall((:rdhip_timer_start, librdhip), Cint, (Ptr{Cvoid},), ctx.ptr))
function timer_stop(ctx::Context=default_context())
    ms = Ref{Cfloat}(0)
    check(ctx, ccall((:rdhip_timer_stop, librdhip), Cint, (Ptr{Cvoid}, Ref{Cfloat}), ctx.ptr, ms))
    return Float64(ms[])
end

# HIP graphs: record the device-memory calls of one solver iteration once, replay them with one launch (small batches are launch-bound).
#   g = capture(ctx) do ... batched calls on device arrays ... end ;  launch(ctx, g)
mutable struct Graph
    ptr::Ptr{Cvoid}
end
function capture(f::Function, ctx::Context=default_context())
    check(ctx, ccall((:rdhip_graph_begin_capture, librdhip), Cint, (Ptr{Cvoid},), ctx.ptr))
    ok = false
    try
        f()
        ok = true
    finally
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_graph_end_capture, librdhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.ptr, out)
        if ok
            check(ctx, rc)
            g = Graph(out[])
            finalizer(x -> ccall((:rdhip_graph_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), g)
            return g
        elseif rc == 0
            ccall((:rdhip_graph_destroy, librdhip), Cint, (Ptr{Cvoid},), out[])
        end
    end
end
launch(ctx::Context, g::Graph) = check(ctx, ccall((:rdhip_graph_launch, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, g.ptr))

# ------------------------------------------------------------------------------------------------ device memory
"""
    DeviceArray{T}(undef, dims...; ctx=default_context())

A dense column-major array in the HBM of `ctx`'s GPU (rdhip_malloc / rdhip_free).  `copyto!(d, a::Array)` uploads (asynchronous on the
context's stream), `copyto!(a::Array, d)` / `Array(d)` download (synchronise).  No scalar indexing: it is a buffer for the batched calls.
"""
mutable struct DeviceArray{T,N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N,Int}
    ctx::Context
    own::Bool
end
function DeviceArray{T}(::UndefInitializer, dims::Vararg{Integer,N}; ctx::Context=default_context()) where {T,N}
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nbytes = prod(dims) * sizeof(T)
    check(ctx, ccall((:rdhip_malloc, librdhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Culonglong), ctx.ptr, out, nbytes))
    d = DeviceArray{T,N}(out[], map(Int, dims), ctx, true)
    finalizer(unsafe_free!, d)
    return d
end
function unsafe_free!(d::DeviceArray)
    if d.own && d.ptr != C_NULL
        # (finalizers run in no particular order: the context this array was allocated through may have been destroyed already -- rdhip_free takes NULL)
        ccall((:rdhip_free, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), C_NULL, d.ptr)
        d.ptr = C_NULL
    end
    return nothing
end
Base.size(d::DeviceArray) = d.dims
Base.size(d::DeviceArray, i::Integer) = i <= length(d.dims) ? d.dims[i] : 1
Base.length(d::DeviceArray) = prod(d.dims)
Base.eltype(::DeviceArray{T}) where {T} = T
Base.ndims(::DeviceArray{T,N}) where {T,N} = N
Base.sizeof(d::DeviceArray{T}) where {T} = length(d) * sizeof(T)
Base.similar(d::DeviceArray{T}) where {T} = DeviceArray{T}(undef, d.dims...; ctx=d.ctx)
Base.show(io::IO, d::DeviceArray{T}) where {T} = print(io, join(d.dims, "x"), " DeviceArray{", T, "} on GPU ", d.ctx.device)
function Base.copyto!(d::DeviceArray{T}, a::Array{T}) where {T}
    length(a) == length(d) || throw(DimensionMismatch("copyto!: $(length(a)) elements into a DeviceArray of $(length(d))"))
    GC.@preserve a check(d.ctx, ccall((:rdhip_memcpy_h2d, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Culonglong), d.ctx.ptr, d.ptr, pointer(a), sizeof(a)))
    synchronize(d.ctx)          # the host array may be freed or rewritten right after this returns
    return d
end
function Base.copyto!(a::Array{T}, d::DeviceArray{T}) where {T}
    length(a) == length(d) || throw(DimensionMismatch("copyto!: a DeviceArray of $(length(d)) elements into $(length(a))"))
    GC.@preserve a check(d.ctx, ccall((:rdhip_memcpy_d2h, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Culonglong), d.ctx.ptr, pointer(a), d.ptr, sizeof(a)))
    return a
end
Base.Array(d::DeviceArray{T,N}) where {T,N} = copyto!(Array{T,N}(undef, d.dims), d)
DeviceArray(a::Array{T,N}; ctx::Context=default_context()) where {T,N} = copyto!(DeviceArray{T}(undef, size(a)...; ctx=ctx), a)
Base.fill!(d::DeviceArray{T}, v) where {T} = copyto!(d, fill(T(v), d.dims))
"A window onto the trailing-dimension range `r` of `d` (no copy, not owning): the block of knot points a shard works on."
function lastdim_view(d::DeviceArray{T,N}, r::UnitRange{Int}) where {T,N}
    (isempty(r) || (first(r) >= 1 && last(r) <= d.dims[N])) || throw(BoundsError(d, r))
    inner = N == 1 ? 1 : prod(d.dims[1:N-1])
    return DeviceArray{T,N}(d.ptr + (first(r) - 1) * inner * sizeof(T), (d.dims[1:N-1]..., length(r)), d.ctx, false)
end
"Device-to-device copy between two contexts' GPUs over xGMI (rdhip_memcpy_peer), ordered after `src`'s stream, executed on `dst`'s."
function copy_peer!(dst::DeviceArray{T}, src::DeviceArray{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copy_peer!: sizes differ"))
    check(dst.ctx, ccall((:rdhip_memcpy_peer, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Culonglong), dst.ctx.ptr, dst.ptr, src.ctx.ptr, src.ptr, sizeof(src)))
    return dst
end

const Dense{T,N} = Union{Array{T,N},DeviceArray{T,N}}
_p(a::Array) = convert(Ptr{Cvoid}, pointer(a))
_p(d::DeviceArray) = d.ptr
_p(::Nothing) = C_NULL
_space(::Array) = HOST
_space(::DeviceArray) = DEVICE
_space(::Nothing) = nothing
# all data arguments of one call live in one memory space; device arrays also fix the context
function _where(args...)
    sp = nothing
    ctx = nothing
    for a in args
        s = _space(a)
        s === nothing && continue
        sp === nothing || sp == s || throw(ArgumentError("all arrays of one batched call must live in the same memory space (host Arrays or DeviceArrays)"))
        sp = s
        if a isa DeviceArray
            ctx === nothing || ctx === a.ctx || throw(ArgumentError("device arrays of one call belong to different contexts"))
            ctx = a.ctx
        end
    end
    return something(sp, HOST), ctx
end
_ctx(given::Union{Nothing,Context}, found::Union{Nothing,Context}) = given !== nothing ? given : (found !== nothing ? found : default_context())

# ------------------------------------------------------------------------------------------------ models
# A compiled-in model is the reference's model type plus a native handle.  `dynamics` / `forces` / `moments` are also defined in Julia (they
# are the user's own model definitions, e.g. examples/single_satellite.jl), so every per-knot method of the reference keeps working on the
# CPU; the batched methods below go to the GPU.
mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(kind::Integer, rot::Integer, frame::Integer, params::Vector{Float64})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_model_create, librdhip), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}), kind, rot, frame, params, length(params), out)
        check(nothing, rc)
        h = new(out[])
        finalizer(x -> ccall((:rdhip_model_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
    # a user-defined model: `hip_source` is the body of  template <class S> __device__ void f(const S* x, const S* u, T t, S* xdot) const
    function Handle(n::Integer, m::Integer, hip_source::AbstractString, preamble::AbstractString, params::Vector{Float64}, rot::Integer, P::Vector{Cint})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_model_create_custom, librdhip), Cint, (Cint, Cint, Cstring, Cstring, Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
                   n, m, hip_source, preamble, params, length(params), rot, P, length(P), out)
        check(nothing, rc)
        h = new(out[])
        finalizer(x -> ccall((:rdhip_model_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
    # a user-defined DISCRETE model: `step_source` is the body of  template <class S> __device__ void step(const S* x, const S* u, T t, T h, S* xn) const,
    # `jac_source` (or nothing) the body of its own jacobian!:  __device__ void jac(const T* x, const T* u, T t, T h, T* J) const
    function Handle(::Val{:discrete}, n::Integer, m::Integer, step_source::AbstractString, jac_source::Union{Nothing,AbstractString}, preamble::AbstractString,
                    params::Vector{Float64}, rot::Integer, P::Vector{Cint})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_model_create_custom_discrete, librdhip), Cint, (Cint, Cint, Cstring, Cstring, Cstring, Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
                   n, m, step_source, jac_source === nothing ? C_NULL : jac_source, preamble, params, length(params), rot, P, length(P), out)
        check(nothing, rc)
        h = new(out[])
        finalizer(x -> ccall((:rdhip_model_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

rotcode(::Type{<:QuatRotation}) = ROT_QUAT
rotcode(::Type{<:MRP}) = ROT_MRP
rotcode(::Type{<:RodriguesParam}) = ROT_RP
_inertia9(J) = vec(Matrix{Float64}(J))

# --- Cartpole (examples/cartpole.jl:5-38) ---
RobotDynamics.@autodiff struct Cartpole{T} <: RobotDynamics.ContinuousDynamics
    mc::T
    mp::T
    l::T
    g::T
    handle::Handle
end
Cartpole(; mc=1.0, mp=0.2, l=0.5, g=9.81) = Cartpole{Float64}(mc, mp, l, g, Handle(MODEL_CARTPOLE, ROT_QUAT, FRAME_WORLD, Float64[mc, mp, l, g]))
RobotDynamics.state_dim(::Cartpole) = 4
RobotDynamics.control_dim(::Cartpole) = 1
function RobotDynamics.dynamics(model::Cartpole, x, u)
    mc, mp, l, g = model.mc, model.mp, model.l, model.g
    q = x[SA[1, 2]]
    qd = x[SA[3, 4]]
    s = sin(q[2])
    c = cos(q[2])
    H = @SMatrix [mc+mp mp*l*c; mp*l*c mp*l^2]
    C = @SMatrix [0 -mp*qd[2]*l*s; 0 0]
    G = @SVector [0, mp * g * l * s]
    B = @SVector [1, 0]
    qdd = -H \ (C * qd + G - B * u[1])
    return [qd; qdd]
end

# --- DoubleIntegrator{D} (test/double_integrator.jl:97-127) ---
RobotDynamics.@autodiff struct DoubleIntegrator{D} <: RobotDynamics.ContinuousDynamics
    handle::Handle
end
DoubleIntegrator(D::Integer) = DoubleIntegrator{Int(D)}(Handle(MODEL_DOUBLE_INTEGRATOR, ROT_QUAT, FRAME_WORLD, Float64[D]))
RobotDynamics.state_dim(::DoubleIntegrator{D}) where {D} = 2D
RobotDynamics.control_dim(::DoubleIntegrator{D}) where {D} = D
RobotDynamics.dynamics(::DoubleIntegrator{D}, x, u) where {D} = SVector{2D}(ntuple(i -> i <= D ? x[D+i] : u[i-D], 2D))
function RobotDynamics.jacobian!(::DoubleIntegrator{D}, J, y, x, u, t) where {D}
    J .= 0
    for i = 1:2D
        J[i, i+D] = 1.0
    end
    return nothing
end

# --- linear system xdot = A x + B u (the known-answer model of the test-suite) ---
RobotDynamics.@autodiff struct LinearModel{n,m} <: RobotDynamics.ContinuousDynamics
    A::Matrix{Float64}
    B::Matrix{Float64}
    handle::Handle
end
function LinearModel(A::AbstractMatrix, B::AbstractMatrix)
    n, m = size(B)
    size(A) == (n, n) || throw(DimensionMismatch("A must be n x n and B n x m"))
    LinearModel{n,m}(Matrix{Float64}(A), Matrix{Float64}(B), Handle(MODEL_LINEAR, ROT_QUAT, FRAME_WORLD, Float64[n; m; vec(Matrix{Float64}(A)); vec(Matrix{Float64}(B))]))
end
RobotDynamics.state_dim(::LinearModel{n}) where {n} = n
RobotDynamics.control_dim(::LinearModel{n,m}) where {n,m} = m
RobotDynamics.dynamics(model::LinearModel, x, u) = model.A * x + model.B * u
function RobotDynamics.jacobian!(model::LinearModel{n,m}, J, y, x, u, t) where {n,m}
    J[:, 1:n] .= model.A
    J[:, n+1:n+m] .= model.B
    return nothing
end

# --- Satellite: RigidBody{R} (examples/single_satellite.jl:7-35; `Body` of test/rigidbody_test.jl:23-54) ---
RobotDynamics.@autodiff struct Satellite{R} <: RobotDynamics.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    bodyframe::Bool
    handle::Handle
end
function Satellite{R}(; mass=1.0, J=SMatrix{3,3,Float64}(I), bodyframe::Bool=false) where {R}
    Js = SMatrix{3,3,Float64,9}(J)
    Satellite{R}(Float64(mass), Js, bodyframe, Handle(MODEL_SATELLITE, rotcode(R), bodyframe ? FRAME_BODY : FRAME_WORLD, Float64[mass; _inertia9(Js)]))
end
Satellite(; kwargs...) = Satellite{QuatRotation{Float64}}(; kwargs...)
"`Body{R}()` of test/rigidbody_test.jl:53-54: mass 2, J = Diagonal(2, 3, 1)"
Body(::Type{R}=QuatRotation{Float64}; bodyframe::Bool=false) where {R} = Satellite{R}(; mass=2.0, J=SMatrix{3,3,Float64}(Diagonal(SA[2.0, 3.0, 1.0])), bodyframe=bodyframe)
RobotDynamics.control_dim(::Satellite) = 6
RobotDynamics.mass(model::Satellite) = model.mass
RobotDynamics.inertia(model::Satellite) = model.J
RobotDynamics.velocity_frame(model::Satellite) = model.bodyframe ? :body : :world
RobotDynamics.forces(model::Satellite, x, u) = RD.orientation(model, x) * SA[u[1], u[2], u[3]]     # world frame
RobotDynamics.moments(model::Satellite, x, u) = SA[u[4], u[5], u[6]]                                # body frame

# --- Quadrotor: RigidBody{R} (test/quadrotor.jl:21-137) ---
RobotDynamics.@autodiff struct Quadrotor{R} <: RobotDynamics.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    gravity::SVector{3,Float64}
    motor_dist::Float64
    kf::Float64
    km::Float64
    bodyframe::Bool
    handle::Handle
end
function Quadrotor{R}(; mass=0.5, J=SMatrix{3,3,Float64}(Diagonal(SA[0.0023, 0.0023, 0.004])), gravity=SA[0.0, 0.0, -9.81], motor_dist=0.1750,
                      kf=1.0, km=0.0245, bodyframe::Bool=false) where {R}
    Js = SMatrix{3,3,Float64,9}(J)
    g = SVector{3,Float64}(gravity)
    p = Float64[mass; _inertia9(Js); g; motor_dist; kf; km]
    Quadrotor{R}(Float64(mass), Js, g, Float64(motor_dist), Float64(kf), Float64(km), bodyframe, Handle(MODEL_QUADROTOR, rotcode(R), bodyframe ? FRAME_BODY : FRAME_WORLD, p))
end
Quadrotor(; kwargs...) = Quadrotor{QuatRotation{Float64}}(; kwargs...)
RobotDynamics.control_dim(::Quadrotor) = 4
RobotDynamics.mass(model::Quadrotor) = model.mass
RobotDynamics.inertia(model::Quadrotor) = model.J
RobotDynamics.velocity_frame(model::Quadrotor) = model.bodyframe ? :body : :world
trim_controls(model::Quadrotor) = @SVector fill(-model.gravity[3] * model.mass / 4.0, 4)
function RobotDynamics.forces(model::Quadrotor, x, u)
    q = RD.orientation(model, x)
    kf = model.kf
    F1 = max(0, kf * u[1]); F2 = max(0, kf * u[2]); F3 = max(0, kf * u[3]); F4 = max(0, kf * u[4])
    return model.mass * model.gravity + q * SA[0.0, 0.0, F1 + F2 + F3 + F4]
end
function RobotDynamics.moments(model::Quadrotor, x, u)
    kf, km, L = model.kf, model.km, model.motor_dist
    F1 = max(0, kf * u[1]); F2 = max(0, kf * u[2]); F3 = max(0, kf * u[3]); F4 = max(0, kf * u[4])
    return SA[L * (F2 - F4), L * (F3 - F1), km * (u[1] - u[2] + u[3] - u[4])]
end

# --- SatellitePair: LieGroupModel with LieState(R, (0, 3, 3)) (examples/satellite_pair.jl:8-44) ---
RobotDynamics.@autodiff struct SatellitePair{R} <: RobotDynamics.LieGroupModel
    J1::SMatrix{3,3,Float64,9}
    J2::SMatrix{3,3,Float64,9}
    handle::Handle
end
function SatellitePair{R}(; J1=SMatrix{3,3,Float64}(I), J2=SMatrix{3,3,Float64}(2.0I)) where {R}
    A = SMatrix{3,3,Float64,9}(J1)
    B = SMatrix{3,3,Float64,9}(J2)
    SatellitePair{R}(A, B, Handle(MODEL_SATELLITE_PAIR, rotcode(R), FRAME_WORLD, Float64[_inertia9(A); _inertia9(B)]))
end
SatellitePair(; kwargs...) = SatellitePair{QuatRotation{Float64}}(; kwargs...)
RobotDynamics.control_dim(::SatellitePair) = 6
RobotDynamics.LieState(::SatellitePair{R}) where {R} = RD.LieState(R, (0, 3, 3))
function RobotDynamics.dynamics(model::SatellitePair, x, u)
    vs = RD.vec_states(model, x)
    qs = RD.rot_states(model, x)
    w1, w2 = vs[2], vs[3]
    q1, q2 = qs[1], qs[2]
    J1, J2 = model.J1, model.J2
    w1dot = J1 \ (u[SA[1, 2, 3]] - w1 × (J1 * w1))
    w2dot = J2 \ (u[SA[4, 5, 6]] - w2 × (J2 * w2))
    q1dot = Rotations.kinematics(q1, w1)
    q2dot = Rotations.kinematics(q2, w2)
    return SA[q1dot[1], q1dot[2], q1dot[3], q1dot[4], w1dot[1], w1dot[2], w1dot[3], q2dot[1], q2dot[2], q2dot[3], q2dot[4], w2dot[1], w2dot[2], w2dot[3]]
end

# --- LieSatellite: the `Satellite` LieGroupModel of test/liemodel.jl:1-22 (x = [w; q]) ---
RobotDynamics.@autodiff struct LieSatellite <: RobotDynamics.LieGroupModel
    J::Diagonal{Float64,SVector{3,Float64}}
    handle::Handle
end
LieSatellite(; Jd=SA[1.0, 1.0, 1.0]) = LieSatellite(Diagonal(SVector{3,Float64}(Jd)), Handle(MODEL_LIE_SATELLITE, ROT_QUAT, FRAME_WORLD, Float64[Jd[1], Jd[2], Jd[3]]))
RobotDynamics.control_dim(::LieSatellite) = 3
RobotDynamics.LieState(::LieSatellite) = RD.LieState(QuatRotation{Float64}, (3, 0))
function RobotDynamics.dynamics(model::LieSatellite, x, u)
    w = SA[x[1], x[2], x[3]]
    q = normalize(SA[x[4], x[5], x[6], x[7]])
    J = model.J
    wdot = J \ (u - w × (J * w))
    qdot = 0.5 * Rotations.lmult(q) * Rotations.hmat() * w
    return [wdot; qdot]
end

# --- user-defined models: `f` runs on the CPU (per-knot calls), `hip_source` is the same function as a HIP device-function body ---
"""
    CustomModel(n, m, f, hip_source; params=Float64[], preamble="")
    CustomLieModel(R, P, m, f, hip_source; params=Float64[], preamble="")

`f(x, u, t)` is the Julia `dynamics` (CPU, per knot point); `hip_source` the body of
`template <class S> __device__ void f(const S* x, const S* u, T t, S* xdot) const` for the batched GPU path (include/rdhip.h,
rdhip_model_create_custom; kernels are JIT-compiled with hipRTC on first use).  `CustomLieModel` carries a `LieState{R,P}`.
"""
struct CustomModel{F} <: RobotDynamics.ContinuousDynamics
    n::Int
    m::Int
    f::F
    handle::Handle
end
function CustomModel(n::Integer, m::Integer, f, hip_source::AbstractString; params=Float64[], preamble::AbstractString="", jacobian::Union{Nothing,AbstractString}=nothing)
    model = CustomModel(Int(n), Int(m), f, Handle(n, m, hip_source, preamble, collect(Float64, params), ROT_QUAT, Cint[]))
    jacobian === nothing || set_user_jacobian!(model, jacobian)
    return model
end
struct CustomLieModel{R,F} <: RobotDynamics.LieGroupModel
    P::Tuple{Vararg{Int}}
    m::Int
    f::F
    handle::Handle
end
function CustomLieModel(::Type{R}, P::Tuple{Vararg{Int}}, m::Integer, f, hip_source::AbstractString; params=Float64[], preamble::AbstractString="") where {R}
    n = length(RD.LieState(R, P))
    CustomLieModel{R,typeof(f)}(P, Int(m), f, Handle(n, m, hip_source, preamble, collect(Float64, params), rotcode(R), collect(Cint, P)))
end
"""
    set_user_jacobian!(model, hip_source)

The model's OWN `jacobian!(model, J, xdot, x, u, t)` (src/dynamics.jl:115-116) as the body of `__device__ void jac(const T* x, const T* u, T t, T* J) const`
(J: n x (n+m) column-major, zero on entry).  `UserDefined()` then evaluates these formulas: directly for the continuous model, through the integrators'
chain rule on the stage Jacobians (src/integration.jl:85-93, :149-177, :302-337) for `DiscretizedDynamics` of models with n + m <= 10
(rdhip_model_set_user_jacobian).  Without it `UserDefined()` is the matrix-free chain rule on forward duals of the dynamics.
"""
set_user_jacobian!(model, hip_source::AbstractString) = (check(nothing, ccall((:rdhip_model_set_user_jacobian, librdhip), Cint, (Ptr{Cvoid}, Cstring), handle(model), hip_source)); model)

"""
    CustomDiscreteModel(n, m, f, step_source; jacobian=nothing, params=Float64[], preamble="")

A user-defined `DiscreteDynamics` model (src/discrete_dynamics.jl:1-108; test/discrete_dynamics_test.jl:5-63): `f(x, u, t, dt)` is the Julia
`discrete_dynamics` (CPU, per knot point), `step_source` the body of `template <class S> __device__ void step(const S* x, const S* u, T t, T h, S* xn) const`
for the batched GPU path, `jacobian` the body of its own `jacobian!(model, J, xn, x, u, t, dt)` (what `UserDefined()` evaluates; without it the reference's
"User-defined discrete dynamics Jacobian not defined." error).  The default diff method is `ForwardAD()`, as `@autodiff` gives a discrete model.
"""
struct CustomDiscreteModel{F} <: RobotDynamics.DiscreteDynamics
    n::Int
    m::Int
    f::F
    handle::Handle
end
CustomDiscreteModel(n::Integer, m::Integer, f, step_source::AbstractString; jacobian::Union{Nothing,AbstractString}=nothing, params=Float64[], preamble::AbstractString="") =
    CustomDiscreteModel(Int(n), Int(m), f, Handle(Val(:discrete), n, m, step_source, jacobian, preamble, collect(Float64, params), ROT_QUAT, Cint[]))
RobotDynamics.state_dim(model::CustomDiscreteModel) = model.n
RobotDynamics.control_dim(model::CustomDiscreteModel) = model.m
RobotDynamics.discrete_dynamics(model::CustomDiscreteModel, x, u, t, dt) = model.f(x, u, t, dt)
RobotDynamics.discrete_dynamics!(model::CustomDiscreteModel, xn, x, u, t, dt) = (xn .= model.f(x, u, t, dt); nothing)
RobotDynamics.default_diffmethod(::CustomDiscreteModel) = RD.ForwardAD()
"`true` for a model whose rule is RDHIP_DISCRETE (rdhip_model_is_discrete)"
function is_discrete(model)
    r = Ref{Cint}(0)
    check(nothing, ccall((:rdhip_model_is_discrete, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}), handle(model), r))
    return r[] != 0
end
const AnyCustom = Union{CustomModel,CustomLieModel}
RobotDynamics.state_dim(model::CustomModel) = model.n
RobotDynamics.control_dim(model::AnyCustom) = model.m
RobotDynamics.LieState(model::CustomLieModel{R}) where {R} = RD.LieState(R, model.P)
RobotDynamics.dynamics(model::AnyCustom, x, u, t) = model.f(x, u, t)
RobotDynamics.dynamics(model::AnyCustom, x, u) = model.f(x, u, 0.0)
"Cost-balance the Jacobian columns of a custom model with n+m > 10 over the compute waves (rdhip_model_balance_columns); call before the first Jacobian."
balance_columns!(model::AnyCustom) = (check(nothing, ccall((:rdhip_model_balance_columns, librdhip), Cint, (Ptr{Cvoid},), handle(model))); model)
"""
    set_cascade!(model, state_block, control_block)

Declare the block structure of a custom model's dynamics (0-based block numbers; what the compiled-in rigid bodies state as ω | q | r, v,
`src/rigidbody.jl:213-236`): the time derivative of block `b` reads blocks `0:b` only, `control_block[j]` is the first block whose derivative reads
control `j`.  The model then gets the kernels that exploit it (FiniteDifference along the cascade, `rollout!` with one wavefront per block,
block-by-block Newton for `ImplicitMidpoint`); the declaration is verified against the Jacobian pattern on first GPU use.
"""
function set_cascade!(model::AnyCustom, state_block::AbstractVector{<:Integer}, control_block::AbstractVector{<:Integer})
    sb = Vector{Cint}(state_block); cb = Vector{Cint}(control_block)
    nb = Cint(max(maximum(sb; init=0), maximum(cb; init=0)) + 1)
    check(nothing, ccall((:rdhip_model_set_cascade, librdhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cint}), handle(model), nb, sb, cb))
    return model
end
"""
    rigid_body_source(forces_src, moments_src; bodyframe=false)

HIP body of `dynamics(model::RigidBody{QuatRotation}, x, u, t)` (`src/rigidbody.jl:213-236`) around user wrenches: `forces_src` fills `F[0..2]` (world
frame), `moments_src` fills `tau[0..2]` (body frame) -- the reference's `forces` / `moments` methods (`src/rigidbody.jl:244-258`) as HIP source; both may
read `q`, `r`, `v`, `w`, `x`, `u`, `t` and the user parameters `P[i]`.  Parameters of the generated model: mass, inertia (9), its inverse (9), then `P`.
"""
function rigid_body_source(forces_src::AbstractString, moments_src::AbstractString; bodyframe::Bool=false)
    lin = bodyframe ?
        "quat_rotate(q, v, xdot);\n    { S Fm[3] = {F[0] * im, F[1] * im, F[2] * im}; S a[3]; quat_rotate_inv(q, Fm, a); S cv[3]; cross3(w, v, cv);\n      for (int i = 0; i < 3; ++i) xdot[7 + i] = a[i] - cv[i]; }" :
        "for (int i = 0; i < 3; ++i) { xdot[i] = v[i]; xdot[7 + i] = F[i] * im; }"
    return """
    const S* r = x; const S* q = x + 3; const S* v = x + 7; const S* w = x + 10; const T* P = p + 19;
    (void)r; (void)P;
    S F[3], tau[3];
    { $forces_src }
    { $moments_src }
    quat_kinematics(q, w, xdot + 3);
    const T im = T(1) / p[0];
    $lin
    S Jw[3]; mat3vec(p + 1, w, Jw);
    S c[3]; cross3(w, Jw, c);
    const S tt[3] = {tau[0] - c[0], tau[1] - c[1], tau[2] - c[2]};
    mat3vec(p + 10, tt, xdot + 10);
"""
end
"""
    CustomRigidBody(m, f, forces_src, moments_src; mass=1.0, J=I, params=Float64[], bodyframe=false, cascade=true, force_only_controls=Int[])

A user-defined `RigidBody{QuatRotation}` for the GPU -- what `struct MyBody <: RigidBody{R}` with `forces` / `moments` methods is in the reference -- as a
`CustomLieModel` with `LieState{QuatRotation,(3,6)}`; `f(x, u, t)` is the Julia `dynamics` for per-knot CPU calls.  `cascade=true` declares the block
structure ω | q | r, v (`set_cascade!`; valid when the moments read neither q nor r, v and the forces neither r nor v; verified on first GPU use).
"""
function CustomRigidBody(m::Integer, f, forces_src::AbstractString, moments_src::AbstractString; mass=1.0, J=SMatrix{3,3,Float64}(I), params=Float64[],
                         bodyframe::Bool=false, preamble::AbstractString="", cascade::Bool=true, force_only_controls=Int[])
    Js = SMatrix{3,3,Float64,9}(J)
    p = Float64[mass; _inertia9(Js); _inertia9(inv(Js)); collect(Float64, params)]
    model = CustomLieModel(QuatRotation{Float64}, (3, 6), m, f, rigid_body_source(forces_src, moments_src; bodyframe=bodyframe); params=p, preamble=preamble)
    cascade && set_cascade!(model, [2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 0, 0, 0], [(j in force_only_controls) ? 2 : 0 for j in 0:m-1])
    return model
end
"""
    wrench_model(m, forces_src, moments_src; mass=1.0, J=I, params=Float64[], preamble="")

`wrenches(model, x, u, t)` / `wrench_jacobian!` (src/rigidbody.jl:244-248, :315-320) of a `CustomRigidBody`, for whole trajectories: a 13-output helper
model built from the SAME two snippets whose first six outputs are `[F; M]` (the rest zero) -- `dynamics!(helper, W13, Z)[1:6, :]` are the wrenches,
`jacobian!(StaticReturn(), ForwardAD(), helper, J, nothing, Z)[1:6, :, :]` is `d wrenches / d z`.  (The compiled-in satellite and quadrotor have
`wrenches!` / `wrench_jacobian!` entry points of their own.)
"""
function wrench_model(m::Integer, forces_src::AbstractString, moments_src::AbstractString; mass=1.0, J=SMatrix{3,3,Float64}(I), params=Float64[],
                      preamble::AbstractString="")
    Js = SMatrix{3,3,Float64,9}(J)
    p = Float64[mass; _inertia9(Js); _inertia9(inv(Js)); collect(Float64, params)]
    src = """
    const S* r = x; const S* q = x + 3; const S* v = x + 7; const S* w = x + 10; const T* P = p + 19;
    (void)r; (void)P; (void)q; (void)v; (void)w;
    S F[3], tau[3];
    { $forces_src }
    { $moments_src }
    for (int i = 0; i < 3; ++i) { xdot[i] = F[i]; xdot[3 + i] = tau[i]; }
    for (int i = 6; i < 13; ++i) xdot[i] = S(T(0));
"""
    return CustomModel(13, m, (x, u, t) -> error("wrench_model: GPU only"), src; params=p, preamble=preamble)
end
"JIT-compile one kernel of a custom model now (works without a GPU): `op` as in rdhip_model_precompile."
precompile_kernel(model::AnyCustom, op::Integer, rule::Integer, ::Type{T}=Float64) where {T} =
    check(nothing, ccall((:rdhip_model_precompile, librdhip), Cint, (Ptr{Cvoid}, Cint, Cint, Cint), handle(model), op, rule, dtypecode(T)))

handle(m) = m.handle.ptr
handle(m::RD.DiscretizedDynamics) = handle(m.continuous_dynamics)

# ------------------------------------------------------------------------------------------------ RK2 (v0.3 explicit midpoint; absent in v0.4.8)
"""
    RK2 <: RobotDynamics.Explicit

The second-order explicit midpoint rule of RobotDynamics v0.3 (README.md:128-136; commented test test/integration_tests.jl:42-45):
`k1 = f(x, u, t) h; k2 = f(x + k1/2, u, t + h/2) h; x + k2`.  `DiscretizedDynamics{RK2}(model)` works on the CPU (methods below) and on the GPU.
"""
struct RK2 <: RobotDynamics.Explicit
    RK2(n::Integer, m::Integer) = new()
end
function RobotDynamics.integrate(::RK2, model, x, u, t, h)
    k1 = RD.dynamics(model, x, u, t) * h
    k2 = RD.dynamics(model, x + k1 / 2, u, t + h / 2) * h
    return x + k2
end
function RobotDynamics.integrate!(::RK2, model, xn, x, u, t, h)
    k1 = similar(xn)
    RD.dynamics!(model, k1, x, u, t)
    @. xn = x + k1 * h / 2
    k2 = similar(xn)
    RD.dynamics!(model, k2, xn, u, t + h / 2)
    @. xn = x + k2 * h
    return nothing
end
# chain rule: A_d = I + A2 (I + A1 h/2) h,  B_d = B2 h + A2 B1 h^2 / 2   (the pattern of src/integration.jl:149-177)
function RobotDynamics.jacobian!(::RK2, sig::RD.FunctionSignature, model, J, xn, x, u, t, h)
    n, m = RD.dims(model)
    k1 = RD.dynamics(model, x, u, t) * h
    RD.jacobian!(model, J, xn, x, u, t)
    A1, B1 = J[1:n, 1:n], J[1:n, n+1:n+m]
    RD.jacobian!(model, J, xn, x + k1 / 2, u, t + h / 2)
    A2, B2 = J[1:n, 1:n], J[1:n, n+1:n+m]
    J[1:n, 1:n] .= I + A2 * (I + 0.5 * h * A1) * h
    J[1:n, n+1:n+m] .= B2 * h + 0.5 * A2 * B1 * h^2
    return nothing
end

rulecode(::RD.ContinuousDynamics) = CONTINUOUS
rulecode(::CustomDiscreteModel) = DISCRETE
rulecode(m::RD.DiscretizedDynamics) = rulecode(RD.integration(m))
rulecode(::RD.Euler) = EULER
rulecode(::RK2) = RULE_RK2
rulecode(::RD.RK3) = RULE_RK3
rulecode(::RD.RK4) = RULE_RK4
rulecode(::RD.ImplicitMidpoint) = IMPLICIT_MIDPOINT
diffcode(::RD.ForwardAD) = FORWARD_AD
diffcode(::RD.FiniteDifference) = FINITE_DIFF
diffcode(::RD.UserDefined) = USER_DEFINED
diffcode(::RD.ImplicitFunctionTheorem{D}) where {D} = diffcode(D())     # the C ABI takes the inner method for implicit rules
dtypecode(::Type{Float64}) = F64
dtypecode(::Type{Float32}) = F32

# implicit models: the reference only accepts ImplicitFunctionTheorem (src/discretized_dynamics.jl:243-249)
function _checkdiff(fun, diff)
    if rulecode(fun) == IMPLICIT_MIDPOINT
        diff isa RD.ImplicitFunctionTheorem || throw(ArgumentError("Must call jacobian! on an ImplicitDynamicsModel with the ImplicitFunctionTheorem diff method. Got $(typeof(diff))."))
    else
        diff isa RD.ImplicitFunctionTheorem && throw(ArgumentError("ImplicitFunctionTheorem needs an implicit integrator"))
    end
end
"setnewtoniters / setnewtontolerance for the GPU Newton solve (src/discretized_dynamics.jl:225-228; defaults 10, 1e-12)"
set_newton!(model, iters::Integer, tol::Real) = check(nothing, ccall((:rdhip_model_set_newton, librdhip), Cint, (Ptr{Cvoid}, Cint, Cdouble), handle(model), iters, tol))

# ------------------------------------------------------------------------------------------------ integer queries served by the library (bit-exact)
function native_dims(model)
    n, m, p, e = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    check(nothing, ccall((:rdhip_model_dims, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}), handle(model), n, m, p, e))
    return Int(n[]), Int(m[]), Int(p[]), Int(e[])
end
function native_statevectortype(model)
    r = Ref{Cint}(0)
    check(nothing, ccall((:rdhip_model_statevectortype, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}), handle(model), r))
    return r[] != 0 ? RD.RotationState() : RD.EuclideanState()
end
function native_liestate(model)
    rot, nP = Ref{Cint}(0), Ref{Cint}(0)
    P = zeros(Cint, 8)
    check(nothing, ccall((:rdhip_model_liestate, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ptr{Cint}), handle(model), rot, nP, P))
    return Int(rot[]), Tuple(Int.(P[1:nP[]]))
end
function native_gen_inds(model)
    first5, len5 = zeros(Cint, 5), zeros(Cint, 5)
    check(nothing, ccall((:rdhip_model_gen_inds, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}), handle(model), first5, len5))
    return NamedTuple{(:r, :q, :v, :ω, :u)}(ntuple(i -> first5[i]:(first5[i]+len5[i]-1), 5))
end
"wrench_sparsity(model) (src/rigidbody.jl:322-347) of a native RigidBody model: the 2 x 5 Bool pattern [∂F/∂r ∂F/∂q ∂F/∂v ∂F/∂ω ∂F/∂u; ∂M/∂…]"
function native_wrench_sparsity(model)
    js = zeros(Cint, 10)
    check(nothing, ccall((:rdhip_model_wrench_sparsity, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cint}), handle(model), js))
    return SMatrix{2,5,Bool,10}(js .!= 0)
end
RobotDynamics.wrench_sparsity(model::Union{Satellite,Quadrotor}) = native_wrench_sparsity(model)
"length, errstate_dim, first(rot_inds(R,P,i)), first(vec_inds(R,P,i)) of a LieState{R,P} (src/liestate.jl:121-131), from the library"
function liestate_query(::Type{R}, P::Tuple{Vararg{Int}}) where {R}
    nP = length(P)
    len, edim = Ref{Cint}(0), Ref{Cint}(0)
    rotf, vecf = zeros(Cint, max(nP - 1, 1)), zeros(Cint, nP)
    check(nothing, ccall((:rdhip_liestate_query, librdhip), Cint, (Cint, Ptr{Cint}, Cint, Ref{Cint}, Ref{Cint}, Ptr{Cint}, Ptr{Cint}),
                         rotcode(R), collect(Cint, P), nP, len, edim, rotf, vecf))
    return Int(len[]), Int(edim[]), Int.(rotf[1:nP-1]), Int.(vecf)
end
function shard_range(N::Integer, nshards::Integer, i::Integer)          # 0-based shard i -> 1-based knot range
    lo, cnt = Ref{Clonglong}(0), Ref{Clonglong}(0)
    check(nothing, ccall((:rdhip_shard_range, librdhip), Cint, (Clonglong, Cint, Cint, Ref{Clonglong}, Ref{Clonglong}), N, nshards, i, lo, cnt))
    return (Int(lo[])+1):(Int(lo[])+Int(cnt[]))
end

# ------------------------------------------------------------------------------------------------ packed trajectories
abstract type AbstractPackedTrajectory{n,m,T} <: RobotDynamics.AbstractTrajectory end
"""
    PackedTrajectory{n,m}(data::Matrix{T}, t, dt)

A dense trajectory in host memory: column k of `data` is `z_k = [x_k; u_k]` (the storage of `KnotPoint.z`, src/knotpoint.jl:146-153).
`Z[k]` is a `KnotPoint{n,m}` over a view of column k (legal: the constructor accepts any vector type, src/knotpoint.jl:154-176), so every
per-knot method of the reference accepts it and writes through to the packed storage.
"""
struct PackedTrajectory{n,m,T} <: AbstractPackedTrajectory{n,m,T}
    data::Matrix{T}
    t::Vector{Float64}
    dt::Vector{Float64}
    function PackedTrajectory{n,m}(data::Matrix{T}, t, dt) where {n,m,T}
        size(data, 1) == n + m || throw(DimensionMismatch("data must be (n+m) x N"))
        length(t) == length(dt) == size(data, 2) || throw(DimensionMismatch("t, dt must have one entry per knot point"))
        new{n,m,T}(data, collect(Float64, t), collect(Float64, dt))
    end
end
Base.length(Z::AbstractPackedTrajectory) = size(Z.data, 2)
Base.size(Z::AbstractPackedTrajectory) = (length(Z),)
Base.firstindex(Z::PackedTrajectory) = 1
Base.lastindex(Z::PackedTrajectory) = length(Z)
Base.getindex(Z::PackedTrajectory{n,m}, k::Integer) where {n,m} = RD.KnotPoint{n,m}(view(Z.data, :, k), Z.t[k], Z.dt[k])
Base.iterate(Z::PackedTrajectory, k::Int=1) = k > length(Z) ? nothing : (Z[k], k + 1)
RobotDynamics.dims(Z::AbstractPackedTrajectory{n,m}) where {n,m} = (n, m, length(Z))
RobotDynamics.getstate(Z::PackedTrajectory{n}, t::Real) where {n} = Z.data[1:n, searchsortedlast(Z.t, t)]
RobotDynamics.getcontrol(Z::PackedTrajectory{n,m}, t::Real) where {n,m} = Z.data[n+1:n+m, searchsortedlast(Z.t, t)]
RobotDynamics.getinitialtime(Z::PackedTrajectory) = Z.t[1]
RobotDynamics.getfinaltime(Z::PackedTrajectory) = Z.t[end]
states(Z::PackedTrajectory{n}) where {n} = view(Z.data, 1:n, :)
controls(Z::PackedTrajectory{n,m}) where {n,m} = view(Z.data, n+1:n+m, :)
PackedTrajectory(Z::RD.SampledTrajectory{n,m,T}) where {n,m,T} =
    PackedTrajectory{n,m}(reduce(hcat, [Vector{T}(RD.getdata(z)) for z in Z]), [RD.time(z) for z in Z], [RD.timestep(z) for z in Z])

"The same packed trajectory resident in HBM (`device(Z)`); `host(Zd)` copies it back."
struct DeviceTrajectory{n,m,T} <: AbstractPackedTrajectory{n,m,T}
    data::DeviceArray{T,2}
    t::DeviceArray{Float64,1}
    dt::DeviceArray{Float64,1}
end
function DeviceTrajectory{n,m,T}(N::Integer; ctx::Context=default_context()) where {n,m,T}
    DeviceTrajectory{n,m,T}(DeviceArray{T}(undef, n + m, N; ctx=ctx), DeviceArray{Float64}(undef, N; ctx=ctx), DeviceArray{Float64}(undef, N; ctx=ctx))
end
device(Z::PackedTrajectory{n,m,T}; ctx::Context=default_context()) where {n,m,T} =
    DeviceTrajectory{n,m,T}(DeviceArray(Z.data; ctx=ctx), DeviceArray(Z.t; ctx=ctx), DeviceArray(Z.dt; ctx=ctx))
host(Z::DeviceTrajectory{n,m,T}) where {n,m,T} = PackedTrajectory{n,m}(Array(Z.data), Array(Z.t), Array(Z.dt))
Base.copyto!(Zd::DeviceTrajectory, Z::PackedTrajectory) = (copyto!(Zd.data, Z.data); copyto!(Zd.t, Z.t); copyto!(Zd.dt, Z.dt); Zd)
"knot points `r` of a device trajectory (no copy): a shard's block"
knots(Z::DeviceTrajectory{n,m,T}, r::UnitRange{Int}) where {n,m,T} = DeviceTrajectory{n,m,T}(lastdim_view(Z.data, r), lastdim_view(Z.t, r), lastdim_view(Z.dt, r))

function _checkmodel(fun, ::AbstractPackedTrajectory{n,m}) where {n,m}
    (RD.state_dim(fun), RD.control_dim(fun)) == (n, m) || throw(DimensionMismatch("the trajectory holds (n, m) = ($n, $m) knot points, the model has ($(RD.state_dim(fun)), $(RD.control_dim(fun)))"))
end

# ------------------------------------------------------------------------------------------------ batched operators (host or device arrays)
# jacobian!(sig, diff, fun, J[:,:,k], y[:,k], Z[k]) for all k  (src/functionbase.jl:242; src/discretized_dynamics.jl:169,208-219)
function RobotDynamics.jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.AbstractModel, J::Dense{T,3}, Y::Union{Nothing,Dense{T,2}},
                                 Z::AbstractPackedTrajectory{n,m,T}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z); _checkdiff(fun, diff)
    N = length(Z)
    size(J) == (n, n + m, N) || throw(DimensionMismatch("J must be n x (n+m) x N"))
    Y === nothing || size(Y) == (n, N) || throw(DimensionMismatch("Y must be n x N"))
    sp, found = _where(Z.data, Z.t, Z.dt, J, Y)
    c = _ctx(ctx, found)
    GC.@preserve J Y Z begin
        rc = ccall((:rdhip_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(J), _p(Y), sp)
    end
    check(c, rc)
end

# wrenches(model, x, u, t) = [forces; moments] for all k (src/rigidbody.jl:244-248): W is 6 x N; and d wrenches / d z for all k -- what the default
# wrench_jacobian!(F, model, z) of src/rigidbody.jl:315-320 is written to return --: F is 6 x (n+m) x N.  Native RigidBody models only.
function wrenches!(W::Dense{T,2}, model::RD.AbstractModel, Z::AbstractPackedTrajectory{n,m,T}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(model, Z)
    N = length(Z)
    size(W) == (6, N) || throw(DimensionMismatch("W must be 6 x N"))
    sp, found = _where(Z.data, Z.t, W)
    c = _ctx(ctx, found)
    GC.@preserve W Z begin
        rc = ccall((:rdhip_wrenches, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(model), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(W), sp)
    end
    check(c, rc)
end
function RobotDynamics.wrench_jacobian!(F::Dense{T,3}, model::RD.AbstractModel, Z::AbstractPackedTrajectory{n,m,T}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(model, Z)
    N = length(Z)
    size(F) == (6, n + m, N) || throw(DimensionMismatch("F must be 6 x (n+m) x N"))
    sp, found = _where(Z.data, Z.t, F)
    c = _ctx(ctx, found)
    GC.@preserve F Z begin
        rc = ccall((:rdhip_wrench_jacobian, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(model), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(F), sp)
    end
    check(c, rc)
end

# discrete_dynamics!(model, xn[:,k], Z[k]) / dynamics!(model, xdot[:,k], Z[k]) for all k  (src/discrete_dynamics.jl:94, src/dynamics.jl:81-83)
function RobotDynamics.evaluate!(fun::RD.AbstractModel, Y::Dense{T,2}, Z::AbstractPackedTrajectory{n,m,T}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z)
    N = length(Z)
    size(Y) == (n, N) || throw(DimensionMismatch("Y must be n x N"))
    sp, found = _where(Z.data, Z.t, Z.dt, Y)
    c = _ctx(ctx, found)
    rule = rulecode(fun)
    GC.@preserve Y Z begin
        if rule == CONTINUOUS
            rc = ccall((:rdhip_dynamics, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                       c.ptr, handle(fun), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(Y), sp)
        else
            rc = ccall((:rdhip_discrete_dynamics, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                       c.ptr, handle(fun), rule, dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(Y), sp)
        end
    end
    check(c, rc)
end
RobotDynamics.discrete_dynamics!(fun::RD.DiscreteDynamics, Xn::Dense, Z::AbstractPackedTrajectory; kw...) = RD.evaluate!(fun, Xn, Z; kw...)
RobotDynamics.dynamics!(fun::RD.ContinuousDynamics, Xdot::Dense, Z::AbstractPackedTrajectory; kw...) = RD.evaluate!(fun, Xdot, Z; kw...)
RobotDynamics.evaluate(fun::RD.AbstractModel, Z::PackedTrajectory{n,m,T}; kw...) where {n,m,T} = (Y = Matrix{T}(undef, n, length(Z)); RD.evaluate!(fun, Y, Z; kw...); Y)
RobotDynamics.discrete_dynamics(fun::RD.DiscreteDynamics, Z::PackedTrajectory; kw...) = RD.evaluate(fun, Z; kw...)
RobotDynamics.dynamics(fun::RD.ContinuousDynamics, Z::PackedTrajectory; kw...) = RD.evaluate(fun, Z; kw...)

# dynamics_error(model, Z[k+1], Z[k]) = discrete_dynamics(model, Z[k]) - state(Z[k+1]) for all consecutive pairs (src/discrete_dynamics.jl:134-156)
function dynamics_error!(fun::RD.DiscretizedDynamics, E::Dense{T,2}, Z::AbstractPackedTrajectory{n,m,T}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z)
    N = length(Z)
    size(E) == (n, max(N - 1, 0)) || throw(DimensionMismatch("E must be n x (N-1)"))
    sp, found = _where(Z.data, Z.t, Z.dt, E)
    c = _ctx(ctx, found)
    GC.@preserve E Z begin
        rc = ccall((:rdhip_dynamics_error, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(E), sp)
    end
    check(c, rc)
end
RobotDynamics.dynamics_error(fun::RD.DiscretizedDynamics, Z::PackedTrajectory{n,m,T}; kw...) where {n,m,T} =
    (E = Matrix{T}(undef, n, max(length(Z) - 1, 0)); dynamics_error!(fun, E, Z; kw...); E)

# dynamics_error_jacobian!(sig, diff, model, J2[:,:,k], J1[:,:,k], y2[:,k], y1, Z[k+1], Z[k]) for all consecutive pairs
# (src/discrete_dynamics.jl:158-201): J1 = [A B] at Z[k], J2 = [-I 0], y2 = f(Z[k]) - state(Z[k+1])
function RobotDynamics.dynamics_error_jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.DiscretizedDynamics, J2::Union{Nothing,Dense{T,3}},
                                                J1::Union{Nothing,Dense{T,3}}, Y2::Union{Nothing,Dense{T,2}}, Y1, Z::AbstractPackedTrajectory{n,m,T};
                                                ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z)
    N = length(Z)
    for A in (J1, J2)
        A === nothing || size(A) == (n, n + m, max(N - 1, 0)) || throw(DimensionMismatch("J1, J2 must be n x (n+m) x (N-1)"))
    end
    Y2 === nothing || size(Y2) == (n, max(N - 1, 0)) || throw(DimensionMismatch("y2 must be n x (N-1)"))
    sp, found = _where(Z.data, Z.t, Z.dt, J2, J1, Y2)
    c = _ctx(ctx, found)
    d = diffcode(diff)
    GC.@preserve J2 J1 Y2 Z begin
        rc = ccall((:rdhip_dynamics_error_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), d, dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(J2), _p(J1), _p(Y2), sp)
    end
    check(c, rc)
end

# error-state expansion along a trajectory: E[:,:,k] = [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k] (what TrajectoryOptimization.jl's
# error_expansion! assembles per knot point); fused into the Jacobian kernel for LieGroupModels
function error_expansion!(fun::RD.DiscretizedDynamics, E::Dense{T,3}, Y::Union{Nothing,Dense{T,2}}, Z::AbstractPackedTrajectory{n,m,T};
                          ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z)
    N = length(Z); e = RD.errstate_dim(fun)
    size(E) == (e, e + m, max(N - 1, 0)) || throw(DimensionMismatch("E must be e x (e+m) x (N-1)"))
    Y === nothing || size(Y) == (n, max(N - 1, 0)) || throw(DimensionMismatch("Y must be n x (N-1)"))
    sp, found = _where(Z.data, Z.t, Z.dt, E, Y)
    c = _ctx(ctx, found)
    GC.@preserve E Y Z begin
        rc = ccall((:rdhip_error_expansion, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), dtypecode(T), N, _p(Z.data), _p(Z.t), _p(Z.dt), _p(E), _p(Y), sp)
    end
    check(c, rc)
end

# errstate_jacobian!(model, G[:,:,k], Z[k]) for all k (src/statevectortype.jl:119-122; src/liestate.jl:262-298).
# `clear=false` reproduces the reference: G is accumulated into, not cleared (RotationState models).  If you zero G before every call -- the reference's own
# pattern, src/trajectories.jl:428-433 -- pass `clear=true` instead and drop the zeroing: G leaves as one dense write of whole lines (63 us at 262 144
# quadrotor knots) where the untouched-entries mode pays a read-modify-write of every line for its 21 scattered values (131 us, plus your memset).
function RobotDynamics.errstate_jacobian!(fun::RD.AbstractModel, G::Dense{T,3}, Z::AbstractPackedTrajectory{n,m,T}; clear::Bool=false,
                                          ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Z)
    N = length(Z); e = RD.errstate_dim(fun)
    size(G) == (n, e, N) || throw(DimensionMismatch("G must be n x e x N"))
    sp, found = _where(Z.data, G)
    c = _ctx(ctx, found)
    GC.@preserve G Z begin
        rc = ccall((:rdhip_errstate_jacobian, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), dtypecode(T), N, _p(Z.data), n + m, clear, _p(G), sp)
    end
    check(c, rc)
end
# ... and over a bare n x N state matrix X
function RobotDynamics.errstate_jacobian!(fun::RD.AbstractModel, G::Dense{T,3}, X::Dense{T,2}; clear::Bool=false, ctx::Union{Nothing,Context}=nothing) where {T}
    n, e, N = RD.state_dim(fun), RD.errstate_dim(fun), size(X, 2)
    size(X, 1) == n || throw(DimensionMismatch("X must be n x N"))
    size(G) == (n, e, N) || throw(DimensionMismatch("G must be n x e x N"))
    sp, found = _where(X, G)
    c = _ctx(ctx, found)
    GC.@preserve G X begin
        rc = ccall((:rdhip_errstate_jacobian, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), dtypecode(T), N, _p(X), n, clear, _p(G), sp)
    end
    check(c, rc)
end

# the `errmap` argument of state_diff (src/liestate.jl:26-32): Rotations.jl's ErrorMap singletons -> rdhip_errmap codes (include/rdhip.h)
errmapcode(::Rotations.CayleyMap) = Cint(0)
errmapcode(::Rotations.ExponentialMap) = Cint(1)
errmapcode(::Rotations.MRPMap) = Cint(2)
errmapcode(::Rotations.QuatVecMap) = Cint(3)
errmapcode(::Rotations.IdentityMap) = Cint(4)      # three-parameter rotations only: RDHIP_ERR_UNSUPPORTED for QuatRotation states, where the reference throws as well

# state_diff!(model, dx[:,k], x[:,k], x0[:,k][, errmap]) for all k, default CayleyMap (src/statevectortype.jl:102-104,145-146; src/liestate.jl:26-32,210-260)
function RobotDynamics.state_diff!(fun::RD.AbstractModel, dX::Dense{T,2}, X::Dense{T,2}, X0::Dense{T,2}, errmap::Rotations.ErrorMap=Rotations.CayleyMap();
                                   ctx::Union{Nothing,Context}=nothing) where {T}
    n, e, N = RD.state_dim(fun), RD.errstate_dim(fun), size(X, 2)
    (size(X, 1) == n && size(X0) == size(X)) || throw(DimensionMismatch("X, X0 must be n x N"))
    size(dX) == (e, N) || throw(DimensionMismatch("dX must be e x N"))
    sp, found = _where(X, X0, dX)
    c = _ctx(ctx, found)
    GC.@preserve dX X X0 begin
        rc = ccall((:rdhip_state_diff_map, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), dtypecode(T), N, _p(X), n, _p(X0), n, errmapcode(errmap), _p(dX), sp)
    end
    check(c, rc)
end
RobotDynamics.state_diff(fun::RD.AbstractModel, X::Matrix{T}, X0::Matrix{T}, errmap::Rotations.ErrorMap=Rotations.CayleyMap(); kw...) where {T} =
    (dX = Matrix{T}(undef, RD.errstate_dim(fun), size(X, 2)); RD.state_diff!(fun, dX, X, X0, errmap; kw...); dX)
# state_diff!(model, dX, Z, Z0): states taken out of two packed trajectories (leading dimension n+m); the default CayleyMap through the plain entry point
function RobotDynamics.state_diff!(fun::RD.AbstractModel, dX::Dense{T,2}, Z::AbstractPackedTrajectory{n,m,T}, Z0::AbstractPackedTrajectory{n,m,T},
                                   errmap::Rotations.ErrorMap=Rotations.CayleyMap(); ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    N = length(Z)
    length(Z0) == N || throw(DimensionMismatch("the two trajectories differ in length"))
    size(dX) == (RD.errstate_dim(fun), N) || throw(DimensionMismatch("dX must be e x N"))
    sp, found = _where(Z.data, Z0.data, dX)
    c = _ctx(ctx, found)
    GC.@preserve dX Z Z0 begin
        rc = errmap isa Rotations.CayleyMap ?
            ccall((:rdhip_state_diff, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                  c.ptr, handle(fun), dtypecode(T), N, _p(Z.data), n + m, _p(Z0.data), n + m, _p(dX), sp) :
            ccall((:rdhip_state_diff_map, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                  c.ptr, handle(fun), dtypecode(T), N, _p(Z.data), n + m, _p(Z0.data), n + m, errmapcode(errmap), _p(dX), sp)
    end
    check(c, rc)
end

# ∇errstate_jacobian!(model, ∇G[:,:,k], x[:,k], x̄[:,k]) for all k (src/statevectortype.jl:137-147; src/liestate.jl:300-320)
function RobotDynamics.∇errstate_jacobian!(fun::RD.AbstractModel, ∇G::Dense{T,3}, X::Dense{T,2}, Xbar::Dense{T,2}; clear::Bool=false,
                                           ctx::Union{Nothing,Context}=nothing) where {T}
    n, e, N = RD.state_dim(fun), RD.errstate_dim(fun), size(X, 2)
    (size(X, 1) == n && size(Xbar) == size(X)) || throw(DimensionMismatch("X, Xbar must be n x N"))
    size(∇G) == (e, e, N) || throw(DimensionMismatch("∇G must be e x e x N"))
    sp, found = _where(X, Xbar, ∇G)
    c = _ctx(ctx, found)
    GC.@preserve ∇G X Xbar begin
        rc = ccall((:rdhip_nabla_errstate_jacobian, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), dtypecode(T), N, _p(X), n, _p(Xbar), n, clear, _p(∇G), sp)
    end
    check(c, rc)
end

# rollout!(sig, dmodel, Z, x0) (src/trajectories.jl:436-441): one packed trajectory, or M stacked trajectories of K knot points
# (Zs holds them back to back; `K` knot points each), sequential in k, parallel over trajectories.
# time_major = true: knot k of all M trajectories contiguous (index k M + j) instead of trajectory after trajectory (j K + k); explicit rules only.
function RobotDynamics.rollout!(::RD.FunctionSignature, fun::RD.DiscretizedDynamics, Zs::AbstractPackedTrajectory{n,m,T}, x0::Union{Nothing,Dense{T}}=nothing;
                                K::Integer=length(Zs), time_major::Bool=false, ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Zs)
    N = length(Zs); M = K == 0 ? 0 : div(N, K)
    M * K == N || throw(DimensionMismatch("the stack must hold whole trajectories of K knot points"))
    x0 === nothing || length(x0) == n * M || throw(DimensionMismatch("x0 must be n x M"))
    sp, found = _where(Zs.data, Zs.t, Zs.dt, x0)
    c = _ctx(ctx, found)
    GC.@preserve Zs x0 begin
        rc = time_major ?
             ccall((:rdhip_rollout_layout, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Clonglong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), dtypecode(T), K, M, Cint(1), _p(Zs.data), _p(Zs.t), _p(Zs.dt), _p(x0), sp) :
             ccall((:rdhip_rollout, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), dtypecode(T), K, M, _p(Zs.data), _p(Zs.t), _p(Zs.dt), _p(x0), sp)
    end
    check(c, rc)
end

# rollout! on split arrays (rdhip_rollout_split): X n x K x M receives the simulated states (knot 1 = x0[:, j], or what X holds there), U m x (K-1) x M the
# controls, t / dt K x M -- what `states(Z)` / `controls(Z)` / `gettimes(Z)` give for M trajectories (src/trajectories.jl:436-441).  time_major = true: the
# arrays are n x M x K, m x M x (K-1), M x K.  Reads U, t, dt once and writes X once: a big batch moves half the bytes of the packed stack's rollout.
function rollout_split!(::RD.FunctionSignature, fun::RD.DiscretizedDynamics, X::Dense{T,3}, U::Dense{T,3}, t::Dense{Float64,2}, dt::Dense{Float64,2},
                        x0::Union{Nothing,Dense{T}}=nothing; time_major::Bool=false, ctx::Union{Nothing,Context}=nothing) where {T}
    n, m = RD.state_dim(fun), RD.control_dim(fun)
    K, M = time_major ? (size(X, 3), size(X, 2)) : (size(X, 2), size(X, 3))
    size(X, 1) == n || throw(DimensionMismatch("X must be n x K x M"))
    size(U) == (time_major ? (m, M, max(K - 1, 0)) : (m, max(K - 1, 0), M)) || throw(DimensionMismatch("U must be m x (K-1) x M"))
    (size(t) == size(X)[2:3] && size(dt) == size(X)[2:3]) || throw(DimensionMismatch("t, dt: one entry per knot point of X"))
    x0 === nothing || length(x0) == n * M || throw(DimensionMismatch("x0 must be n x M"))
    sp, found = _where(X, U, t, dt, x0)
    c = _ctx(ctx, found)
    GC.@preserve X U t dt x0 begin
        rc = ccall((:rdhip_rollout_split, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Clonglong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), dtypecode(T), K, M, Cint(time_major), _p(X), _p(U), _p(t), _p(dt), _p(x0), sp)
    end
    check(c, rc)
end

# rollout! followed by jacobian! at every simulated knot point, one call (the forward pass + linearisation of a shooting method)
function rollout_jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.DiscretizedDynamics, J::Dense{T,3}, Zs::AbstractPackedTrajectory{n,m,T},
                           K::Integer, x0::Union{Nothing,Dense{T}}=nothing; time_major::Bool=false, ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkmodel(fun, Zs); _checkdiff(fun, diff)
    N = length(Zs); M = K == 0 ? 0 : div(N, K)
    M * K == N || throw(DimensionMismatch("the stack must hold whole trajectories of K knot points"))
    size(J) == (n, n + m, N) || throw(DimensionMismatch("J must be n x (n+m) x (K*M)"))
    sp, found = _where(Zs.data, Zs.t, Zs.dt, x0, J)
    c = _ctx(ctx, found)
    GC.@preserve J Zs x0 begin
        rc = ccall((:rdhip_rollout_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Clonglong, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   c.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), K, M, time_major ? TIME_MAJOR : TRAJECTORY_MAJOR,
                   _p(Zs.data), _p(Zs.t), _p(Zs.dt), _p(x0), _p(J), sp)
    end
    check(c, rc)
    k = Ref{Cint}(0)          # kernel launches issued: 1 = rollout and Jacobians overlapped in one launch
    check(c, ccall((:rdhip_ctx_last_rollout_jacobian_launches, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}), c.ptr, k))
    Int(k[])
end

# rollout! + jacobian! along the rollouts for several MODELS in one call (the mixed-model Monte-Carlo batch: even trajectories one model, odd ones
# another): every model on its own internal stream between one fork and one join, so the rollouts' latency chains run side by side
function rollout_jacobian_multi!(::RD.FunctionSignature, diff::RD.DiffMethod, funs::Vector{<:RD.DiscretizedDynamics}, Js::Vector, Zs::Vector, Ks::Vector{<:Integer},
                                 x0s::Union{Nothing,Vector}=nothing; time_major::Bool=false, ctx::Union{Nothing,Context}=nothing)
    nm = length(funs)
    length(Js) == nm && length(Zs) == nm && length(Ks) == nm && (x0s === nothing || length(x0s) == nm) || throw(DimensionMismatch("one entry per model"))
    nm == 0 && return nothing
    T = eltype(Zs[1].data)
    all_arrays = Any[]
    Ms = Clonglong[]
    for p = 1:nm
        _checkmodel(funs[p], Zs[p]); _checkdiff(funs[p], diff)
        eltype(Zs[p].data) == T || throw(ArgumentError("all stacks must share one element type"))
        N = length(Zs[p]); M = Ks[p] == 0 ? 0 : div(N, Ks[p])
        M * Ks[p] == N || throw(DimensionMismatch("stack $p must hold whole trajectories of K knot points"))
        push!(Ms, M)
        push!(all_arrays, Zs[p].data, Zs[p].t, Zs[p].dt, Js[p])
        x0s === nothing || x0s[p] === nothing || push!(all_arrays, x0s[p])
    end
    sp, found = _where(all_arrays...)
    c = _ctx(ctx, found)
    hs = Ptr{Cvoid}[handle(f) for f in funs]; rules = Cint[rulecode(f) for f in funs]; Kc = Clonglong[k for k in Ks]
    pZ = Ptr{Cvoid}[_p(Z.data) for Z in Zs]; pt = Ptr{Cvoid}[_p(Z.t) for Z in Zs]; pdt = Ptr{Cvoid}[_p(Z.dt) for Z in Zs]
    pX = Ptr{Cvoid}[(x0s === nothing ? C_NULL : _p(x0s[p])) for p = 1:nm]
    pJ = Ptr{Cvoid}[_p(J) for J in Js]
    GC.@preserve funs Js Zs x0s begin
        rc = ccall((:rdhip_rollout_jacobian_multi, librdhip), Cint,
                   (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Cint, Cint, Ptr{Clonglong}, Ptr{Clonglong}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}},
                    Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint),
                   c.ptr, nm, hs, rules, diffcode(diff), dtypecode(T), Kc, Ms, time_major ? TIME_MAJOR : TRAJECTORY_MAJOR, pZ, pt, pdt, pX, pJ, sp)
    end
    check(c, rc)
end

# jacobian! for several INDEPENDENT packed trajectories (different arrays) in ONE kernel launch: a trajectory of a few thousand knot points is
# launch-bound on its own, sixteen of them enumerated by one persistent tile loop cost what one 65 536-knot batch costs
function jacobian_multi!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.AbstractModel, Js::Vector{<:Dense{T,3}}, Ys::Union{Nothing,Vector},
                         Zs::Vector{<:AbstractPackedTrajectory{n,m,T}}; ctx::Union{Nothing,Context}=nothing) where {n,m,T}
    _checkdiff(fun, diff)
    nb = length(Zs)
    length(Js) == nb && (Ys === nothing || length(Ys) == nb) || throw(DimensionMismatch("Js, Ys and Zs must list one entry per trajectory"))
    all_arrays = Any[]
    for b = 1:nb
        _checkmodel(fun, Zs[b])
        size(Js[b]) == (n, n + m, length(Zs[b])) || throw(DimensionMismatch("Js[$b] must be n x (n+m) x N"))
        push!(all_arrays, Zs[b].data, Zs[b].t, Zs[b].dt, Js[b])
        Ys === nothing || Ys[b] === nothing || (size(Ys[b]) == (n, length(Zs[b])) ? push!(all_arrays, Ys[b]) : throw(DimensionMismatch("Ys[$b] must be n x N")))
    end
    sp, found = _where(all_arrays...)
    c = _ctx(ctx, found)
    Ns = Clonglong[length(Z) for Z in Zs]
    pZ = Ptr{Cvoid}[_p(Z.data) for Z in Zs]; pt = Ptr{Cvoid}[_p(Z.t) for Z in Zs]; pdt = Ptr{Cvoid}[_p(Z.dt) for Z in Zs]
    pJ = Ptr{Cvoid}[_p(J) for J in Js]
    pY = Ptr{Cvoid}[(Ys === nothing ? C_NULL : _p(Ys[b])) for b = 1:nb]
    GC.@preserve Js Ys Zs begin
        rc = ccall((:rdhip_jacobian_multi, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Clonglong}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint),
                   c.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), nb, Ns, pZ, pt, pdt, pJ, pY, sp)
    end
    check(c, rc)
    last_multi_launches(c)
end
"kernel launches the last `jacobian_multi!` on `ctx` issued: 1 = fused into one launch, 3 (5 for ImplicitMidpoint) = gathered into one stack (diff methods without a batch-table kernel), otherwise one (three) per run of adjacent trajectories"
function last_multi_launches(c::Context)
    k = Ref{Cint}(0)
    check(c, ccall((:rdhip_ctx_last_multi_launches, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}), c.ptr, k))
    Int(k[])
end

# ------------------------------------------------------------------------------------------------ several GPUs
# Knot points are independent: rank r of P owns block shard_range(N, P, r), evaluation never communicates.
"jacobian! on HOST arrays of the whole batch, block i on `ctxs[i]`, all blocks concurrently; the stacked result is complete in host memory on return"
function jacobian_sharded!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.AbstractModel, J::Array{T,3}, Y::Union{Nothing,Matrix{T}},
                           Z::PackedTrajectory{n,m,T}, ctxs::Vector{Context}) where {n,m,T}
    _checkmodel(fun, Z); _checkdiff(fun, diff)
    size(J) == (n, n + m, length(Z)) || throw(DimensionMismatch("J must be n x (n+m) x N"))
    ptrs = Ptr{Cvoid}[c.ptr for c in ctxs]
    GC.@preserve J Y Z ctxs begin
        rc = ccall((:rdhip_jacobian_sharded, librdhip), Cint,
                   (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                   ptrs, length(ptrs), handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), length(Z), _p(Z.data), _p(Z.t), _p(Z.dt), _p(J), _p(Y))
    end
    check(nothing, rc)
end

"""
    Comm(ctxs::Vector{Context})                       # one process drives all ranks (ncclCommInitAll)
    Comm(ctx, nranks, rank, id::Vector{UInt8})        # one process per GPU; `id = unique_id()` drawn by rank 0 and handed to the others (MPI, a file, ...)

The library's own RCCL communicator for the one exchange of the multi-GPU layout: stacking device-resident result shards (`gather!`).
"""
mutable struct Comm
    ptr::Ptr{Cvoid}
    ctxs::Vector{Context}
    function Comm(ctxs::Vector{Context})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        ptrs = Ptr{Cvoid}[c.ptr for c in ctxs]
        rc = ccall((:rdhip_comm_create, librdhip), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ref{Ptr{Cvoid}}), ptrs, length(ptrs), out)
        check(nothing, rc)
        C = new(out[], ctxs)
        finalizer(x -> ccall((:rdhip_comm_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), C)
        return C
    end
    function Comm(ctx::Context, nranks::Integer, rank::Integer, id::Vector{UInt8})
        length(id) == COMM_ID_BYTES || throw(ArgumentError("the communicator id has $COMM_ID_BYTES bytes"))
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_comm_create_rank, librdhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}), ctx.ptr, nranks, rank, id, out)
        check(ctx, rc)
        C = new(out[], [ctx])
        finalizer(x -> ccall((:rdhip_comm_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), C)
        return C
    end
end
function unique_id()
    id = zeros(UInt8, COMM_ID_BYTES)
    check(nothing, ccall((:rdhip_comm_unique_id, librdhip), Cint, (Ptr{UInt8},), id))
    return id
end
function Base.size(C::Comm)
    nr, nl = Ref{Cint}(0), Ref{Cint}(0)
    check(nothing, ccall((:rdhip_comm_size, librdhip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), C.ptr, nr, nl))
    return Int(nr[]), Int(nl[])
end
function rank_of(C::Comm, l::Integer)            # 0-based local slot -> global rank
    r = Ref{Cint}(0)
    cp = Ref{Ptr{Cvoid}}(C_NULL)
    check(nothing, ccall((:rdhip_comm_rank, librdhip), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ref{Ptr{Cvoid}}), C.ptr, l, r, cp))
    return Int(r[])
end
_gathercode(mode::Symbol) = mode === :none ? GATHER_NONE : mode === :root ? GATHER_ROOT : mode === :all ? GATHER_ALL : throw(ArgumentError("gather mode must be :none, :root or :all"))
"""
    gather!(comm, mode, shards, nbytes, outs; root=0)

`shards[l]` / `outs[l]`: the device buffers of the ranks this process drives; `nbytes[r]`: the shard size of EVERY rank.  `:root` stacks the
shards in rank order on rank `root` (grouped ncclSend / ncclRecv, every shard on its own xGMI link), `:all` on every rank, `:none` is a no-op.
Asynchronous on each rank's context stream.
"""
function gather!(C::Comm, mode::Symbol, shards::Vector{<:DeviceArray}, nbytes::Vector{<:Integer}, outs::Vector{<:DeviceArray}; root::Integer=0)
    ps = Ptr{Cvoid}[s.ptr for s in shards]
    po = Ptr{Cvoid}[o.ptr for o in outs]
    nb = Culonglong[b for b in nbytes]
    GC.@preserve shards outs begin
        rc = ccall((:rdhip_gather, librdhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Ptr{Culonglong}, Ptr{Ptr{Cvoid}}), C.ptr, _gathercode(mode), root, ps, nb, po)
    end
    check(C.ctxs[1], rc)
end
"jacobian! on every local rank's block (Zs[l], Js[l], Ys[l] are that BLOCK's device arrays; N = total knot count), then the stacked J by `gather`"
function jacobian_sharded_gather!(C::Comm, diff::RD.DiffMethod, fun::RD.AbstractModel, N::Integer, Zs::Vector{<:DeviceTrajectory{n,m,T}}, Js::Vector{<:DeviceArray{T,3}},
                                  Ys::Union{Nothing,Vector{<:DeviceArray{T,2}}}, gather::Symbol, Jstacks::Vector{<:DeviceArray{T,3}}; root::Integer=0) where {n,m,T}
    _checkdiff(fun, diff)
    pZ = Ptr{Cvoid}[Z.data.ptr for Z in Zs]; pt = Ptr{Cvoid}[Z.t.ptr for Z in Zs]; pdt = Ptr{Cvoid}[Z.dt.ptr for Z in Zs]
    pJ = Ptr{Cvoid}[J.ptr for J in Js]
    pY = Ys === nothing ? Ptr{Cvoid}[C_NULL for _ in Zs] : Ptr{Cvoid}[Y.ptr for Y in Ys]
    pS = Ptr{Cvoid}[S.ptr for S in Jstacks]
    GC.@preserve Zs Js Ys Jstacks begin
        rc = ccall((:rdhip_jacobian_sharded_gather, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Cint, Ptr{Ptr{Cvoid}}),
                   C.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), N, pZ, pt, pdt, pJ, pY, _gathercode(gather), root, pS)
    end
    check(C.ctxs[1], rc)
end
"errstate_jacobian! on every local rank's block of states (Xs[l]: n x N_l, leading dimension n), then the stacked G by `gather` (BASELINE configs[3])"
function errstate_jacobian_sharded_gather!(C::Comm, fun::RD.AbstractModel, N::Integer, Xs::Vector{<:DeviceArray{T,2}}, Gs::Vector{<:DeviceArray{T,3}},
                                           gather::Symbol, Gstacks::Vector{<:DeviceArray{T,3}}; clear::Bool=true, root::Integer=0) where {T}
    pX = Ptr{Cvoid}[X.ptr for X in Xs]
    pG = Ptr{Cvoid}[G.ptr for G in Gs]
    pS = Ptr{Cvoid}[S.ptr for S in Gstacks]
    GC.@preserve Xs Gs Gstacks begin
        rc = ccall((:rdhip_errstate_jacobian_sharded_gather, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Ptr{Cvoid}}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Cint, Ptr{Ptr{Cvoid}}),
                   C.ptr, handle(fun), dtypecode(T), N, pX, size(Xs[1], 1), clear, pG, _gathercode(gather), root, pS)
    end
    check(C.ctxs[1], rc)
end

# ------------------------------------------------------------------------------------------------ v0.3 spellings (README.md:73-82,128-136; SURVEY.md App. B)
discrete_dynamics(::Type{Q}, model::RD.ContinuousDynamics, Z::PackedTrajectory) where {Q<:RD.QuadratureRule} = RD.evaluate(RD.DiscretizedDynamics{Q}(model), Z)
discrete_jacobian!(::Type{Q}, D::Dense{T,3}, model::RD.ContinuousDynamics, Z::AbstractPackedTrajectory{n,m,T}) where {Q<:RD.QuadratureRule,n,m,T} =
    RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), RD.DiscretizedDynamics{Q}(model), D, nothing, Z)
jacobian!(D::Dense{T,3}, model::RD.ContinuousDynamics, Z::AbstractPackedTrajectory{n,m,T}) where {n,m,T} =
    RD.jacobian!(RD.default_signature(model), RD.default_diffmethod(model), model, D, nothing, Z)
state_diff_jacobian!(G::Dense{T,3}, model::RD.AbstractModel, Z::AbstractPackedTrajectory{n,m,T}) where {n,m,T} = RD.errstate_jacobian!(model, G, Z)
∇²differential!(∇G::Dense{T,3}, model::RD.AbstractModel, X::Dense{T,2}, dX::Dense{T,2}) where {T} = RD.∇errstate_jacobian!(model, ∇G, X, dX)
const Traj = PackedTrajectory
traj_size(Z::AbstractPackedTrajectory) = RD.dims(Z)

end # module
