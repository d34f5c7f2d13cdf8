# RobotDynamicsHIP.jl -- the Julia host side of librdhip: batched RobotDynamics.jl operators over a packed trajectory,
# one `ccall` per batch.  Thin by design: argument checks, pointer plumbing, nothing numerical.
#
# STATUS: written against include/rdhip.h but NOT executed in the build image (no Julia toolchain there, see
# INTEGRATION.md).  The Python module `rdhip` next to this file is the same host mirror, and is what the tests run.
#
# Usage (mirrors the per-knot loop of the reference, src/functionbase.jl:242):
#
#     using RobotDynamics, RobotDynamicsHIP
#     const RD = RobotDynamics; const RH = RobotDynamicsHIP
#     model  = RH.Satellite()                                   # <: RD.RigidBody{QuatRotation}
#     dmodel = RD.DiscretizedDynamics{RD.RK4}(model)
#     Z = RH.PackedTrajectory{13,6}(data, t, dt)                # (n+m) x N Matrix + t, dt; Z[k] is a KnotPoint view
#     J = zeros(13, 19, N); Y = zeros(13, N)
#     RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dmodel, J, Y, Z)       # ONE kernel launch for all k
#     G = zeros(13, 12, N); RD.errstate_jacobian!(model, G, Z)
module RobotDynamicsHIP

using RobotDynamics
using Rotations
using StaticArrays
const RD = RobotDynamics

const librdhip = get(ENV, "RDHIP_LIB", joinpath(@__DIR__, "..", "lib", "librdhip.so"))

# enums of include/rdhip.h
const MODEL_CARTPOLE, MODEL_DOUBLE_INTEGRATOR, MODEL_LINEAR, MODEL_SATELLITE, MODEL_QUADROTOR, MODEL_SATELLITE_PAIR, MODEL_LIE_SATELLITE = 0:6
const ROT_QUAT, ROT_MRP, ROT_RP = 0:2
const CONTINUOUS, EULER, RK2, RK3, RK4 = -1:3
const FORWARD_AD, FINITE_DIFF, USER_DEFINED = 0:2
const F64, F32 = 0:1
const DEVICE, HOST = 0:1

struct RdhipError <: Exception
    code::Int
    msg::String
end
Base.showerror(io::IO, e::RdhipError) = print(io, "rdhip error ", e.code, ": ", e.msg)

mutable struct Context
    ptr::Ptr{Cvoid}
    function Context(device::Integer=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_ctx_create, librdhip), Cint, (Cint, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), device, C_NULL, 0, out)
        rc == 0 || throw(RdhipError(rc, unsafe_string(ccall((:rdhip_last_error, librdhip), Cstring, (Ptr{Cvoid},), C_NULL))))
        ctx = new(out[])
        finalizer(c -> ccall((:rdhip_ctx_destroy, librdhip), Cint, (Ptr{Cvoid},), c.ptr), ctx)
    end
end
const _default_ctx = Ref{Union{Nothing,Context}}(nothing)
default_context() = something(_default_ctx[], (_default_ctx[] = Context(); _default_ctx[]))

function check(ctx::Context, rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:rdhip_last_error, librdhip), Cstring, (Ptr{Cvoid},), ctx.ptr))
    # map the status codes onto the exceptions the reference throws (include/rdhip.h, rdhip_status)
    rc == -1 && throw(ArgumentError(msg))
    (rc == -2 || rc == -3) && throw(RD.NotImplementedError(msg))
    rc == -6 && throw(DimensionMismatch(msg))
    throw(RdhipError(rc, msg))
end

# ---------------------------------------------------------------------------------------------- models
# A compiled-in model is the reference's model type plus a native handle.  `dynamics`/`wrenches` are also defined in
# Julia (copied from nowhere: they are the user's own model definitions, e.g. examples/single_satellite.jl), so that
# every per-knot method of the reference keeps working on the CPU; only the batched methods below go to the GPU.
mutable struct Handle
    ptr::Ptr{Cvoid}
    Handle(ptr::Ptr{Cvoid}) = new(ptr)
    function Handle(kind, rot, frame, params::Vector{Float64})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_model_create, librdhip), Cint, (Cint, Cint, Cint, Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}),
                   kind, rot, frame, params, length(params), out)
        rc == 0 || throw(ArgumentError(unsafe_string(ccall((:rdhip_last_error, librdhip), Cstring, (Ptr{Cvoid},), C_NULL))))
        h = new(out[])
        finalizer(x -> ccall((:rdhip_model_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), h)
    end
end

rotcode(::Type{<:QuatRotation}) = ROT_QUAT
rotcode(::Type{<:MRP}) = ROT_MRP
rotcode(::Type{<:RodriguesParam}) = ROT_RP

RD.@autodiff struct Satellite{R} <: RD.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    handle::Handle
end
function Satellite{R}(; mass=1.0, J=SMatrix{3,3}(1.0I)) where {R}
    Satellite{R}(mass, J, Handle(MODEL_SATELLITE, rotcode(R), 0, [mass; vec(Matrix(J))]))
end
Satellite(; kw...) = Satellite{QuatRotation{Float64}}(; kw...)
RD.control_dim(::Satellite) = 6
RD.mass(m::Satellite) = m.mass
RD.inertia(m::Satellite) = m.J
RD.forces(m::Satellite, x, u) = RD.orientation(m, x) * SA[u[1], u[2], u[3]]       # world frame (examples/single_satellite.jl:18-23)
RD.moments(m::Satellite, x, u) = SA[u[4], u[5], u[6]]

RD.@autodiff struct Cartpole <: RD.ContinuousDynamics
    mc::Float64; mp::Float64; l::Float64; g::Float64
    handle::Handle
end
Cartpole(; mc=1.0, mp=0.2, l=0.5, g=9.81) = Cartpole(mc, mp, l, g, Handle(MODEL_CARTPOLE, 0, 0, [mc, mp, l, g]))
RD.state_dim(::Cartpole) = 4
RD.control_dim(::Cartpole) = 1
function RD.dynamics(m::Cartpole, x, u)        # examples/cartpole.jl:15-34
    q, qd = x[SA[1, 2]], x[SA[3, 4]]
    s, c = sincos(q[2])
    H = @SMatrix [m.mc+m.mp m.mp*m.l*c; m.mp*m.l*c m.mp*m.l^2]
    C = @SMatrix [0 -m.mp*qd[2]*m.l*s; 0 0]
    G = @SVector [0, m.mp * m.g * m.l * s]
    B = @SVector [1, 0]
    qdd = -H \ (C * qd + G - B * u[1])
    return [qd; qdd]
end

# A user-defined model: the Julia `dynamics` method runs on the CPU (per-knot calls), `hip_source` is the same function written as a
# HIP device-function body for the batched GPU path (include/rdhip.h, rdhip_model_create_custom)
function custom_handle(n::Integer, m::Integer, hip_source::String; params=Float64[], preamble="", rotation=ROT_QUAT, P=Int[])
    out = Ref{Ptr{Cvoid}}(C_NULL)
    p = collect(Float64, params); Pc = collect(Cint, P)
    rc = ccall((:rdhip_model_create_custom, librdhip), Cint,
               (Cint, Cint, Cstring, Cstring, Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
               n, m, hip_source, preamble, p, length(p), rotation, Pc, length(Pc), out)
    rc == 0 || throw(ArgumentError(unsafe_string(ccall((:rdhip_last_error, librdhip), Cstring, (Ptr{Cvoid},), C_NULL))))
    h = Handle(out[])
    finalizer(x -> ccall((:rdhip_model_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), h)
end

handle(m) = m.handle.ptr
handle(m::RD.DiscretizedDynamics) = handle(m.continuous_dynamics)
rulecode(::RD.ContinuousDynamics) = CONTINUOUS
rulecode(m::RD.DiscretizedDynamics) = rulecode(RD.integration(m))
rulecode(::RD.Euler) = EULER
rulecode(::RD.RK3) = RK3
rulecode(::RD.RK4) = RK4
rulecode(::RD.ImplicitMidpoint) = 4          # RDHIP_IMPLICIT_MIDPOINT
diffcode(::RD.ForwardAD) = FORWARD_AD
diffcode(::RD.FiniteDifference) = FINITE_DIFF
diffcode(::RD.UserDefined) = USER_DEFINED
diffcode(::RD.ImplicitFunctionTheorem{D}) where {D} = diffcode(D())     # the C ABI takes the inner method for implicit rules
dtypecode(::Type{Float64}) = F64
dtypecode(::Type{Float32}) = F32

# ---------------------------------------------------------------------------------------------- packed trajectory
"""
    PackedTrajectory{n,m,T}(data::Matrix{T}, t::Vector{Float64}, dt::Vector{Float64})

A dense `AbstractTrajectory`: column k of `data` is `z_k = [x_k; u_k]` (the storage of `KnotPoint.z`,
src/knotpoint.jl:146-153).  `Z[k]` is a `KnotPoint{n,m}` over a view of column k (legal: the constructor accepts any
vector type, src/knotpoint.jl:154-176; `SubArray` is a `DataVector`, src/RobotDynamics.jl:40), so every per-knot method
of the reference accepts it.
"""
struct PackedTrajectory{n,m,T} <: RD.AbstractTrajectory{n,m,T}
    data::Matrix{T}
    t::Vector{Float64}
    dt::Vector{Float64}
    function PackedTrajectory{n,m}(data::Matrix{T}, t, dt) where {n,m,T}
        size(data, 1) == n + m || throw(DimensionMismatch("data must be (n+m) x N"))
        length(t) == length(dt) == size(data, 2) || throw(DimensionMismatch("t, dt must have one entry per knot point"))
        new{n,m,T}(data, collect(Float64, t), collect(Float64, dt))
    end
end
Base.length(Z::PackedTrajectory) = size(Z.data, 2)
Base.size(Z::PackedTrajectory) = (length(Z),)
Base.getindex(Z::PackedTrajectory{n,m}, k::Int) where {n,m} = RD.KnotPoint{n,m}(view(Z.data, :, k), Z.t[k], Z.dt[k])
Base.IndexStyle(::PackedTrajectory) = IndexLinear()
PackedTrajectory(Z::RD.SampledTrajectory{n,m,T}) where {n,m,T} =
    PackedTrajectory{n,m}(reduce(hcat, [Vector{T}(RD.getdata(z)) for z in Z]), RD.gettimes(Z), [RD.timestep(z) for z in Z])

# ---------------------------------------------------------------------------------------------- batched operators
# jacobian!(sig, diff, fun, J[:,:,k], y[:,k], Z[k]) for all k  (src/functionbase.jl:242; src/discretized_dynamics.jl:169,208-219)
function RD.jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.AbstractModel, J::Array{T,3}, Y::Union{Nothing,Matrix{T}},
                      Z::PackedTrajectory{n,m,T}; ctx::Context=default_context()) where {n,m,T}
    N = length(Z)
    size(J) == (n, n + m, N) || throw(DimensionMismatch("J must be n x (n+m) x N"))
    Y === nothing || size(Y) == (n, N) || throw(DimensionMismatch("Y must be n x N"))
    GC.@preserve J Y Z begin
        rc = ccall((:rdhip_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   ctx.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), N, Z.data, Z.t, Z.dt, J,
                   Y === nothing ? C_NULL : pointer(Y), HOST)
    end
    check(ctx, rc)
end

# discrete_dynamics!(model, xn[:,k], Z[k]) / dynamics!(model, xdot[:,k], Z[k]) for all k  (src/discrete_dynamics.jl:94, src/dynamics.jl:81-83)
function RD.evaluate!(fun::RD.AbstractModel, Y::Matrix{T}, Z::PackedTrajectory{n,m,T}; ctx::Context=default_context()) where {n,m,T}
    N = length(Z)
    size(Y) == (n, N) || throw(DimensionMismatch("Y must be n x N"))
    rule = rulecode(fun)
    GC.@preserve Y Z begin
        rc = rule == CONTINUOUS ?
            ccall((:rdhip_dynamics, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Cint),
                  ctx.ptr, handle(fun), dtypecode(T), N, Z.data, Z.t, Z.dt, Y, HOST) :
            ccall((:rdhip_discrete_dynamics, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Cint),
                  ctx.ptr, handle(fun), rule, dtypecode(T), N, Z.data, Z.t, Z.dt, Y, HOST)
    end
    check(ctx, rc)
end
RD.discrete_dynamics!(fun::RD.DiscreteDynamics, Xn::Matrix, Z::PackedTrajectory; kw...) = RD.evaluate!(fun, Xn, Z; kw...)
RD.dynamics!(fun::RD.ContinuousDynamics, Xdot::Matrix, Z::PackedTrajectory; kw...) = RD.evaluate!(fun, Xdot, Z; kw...)

# errstate_jacobian!(model, G[:,:,k], Z[k]) for all k (src/statevectortype.jl:119-122; src/liestate.jl:262-298).
# `clear=false` reproduces the reference: G is accumulated into, not cleared (RotationState models).
function RD.errstate_jacobian!(fun::RD.AbstractModel, G::Array{T,3}, Z::PackedTrajectory{n,m,T}; clear::Bool=false, ctx::Context=default_context()) where {n,m,T}
    N = length(Z); e = RD.errstate_dim(fun)
    size(G) == (n, e, N) || throw(DimensionMismatch("G must be n x e x N"))
    GC.@preserve G Z begin
        rc = ccall((:rdhip_errstate_jacobian, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Clonglong, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint),
                   ctx.ptr, handle(fun), dtypecode(T), N, Z.data, n + m, clear, G, HOST)
    end
    check(ctx, rc)
end

# dynamics_error_jacobian!(sig, diff, model, J2[:,:,k], J1[:,:,k], y2[:,k], y1, Z[k+1], Z[k]) for all consecutive pairs
# (src/discrete_dynamics.jl:158-201): J1 = [A B] at Z[k], J2 = [-I 0], y2 = f(Z[k]) - state(Z[k+1])
function RD.dynamics_error_jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.DiscretizedDynamics, J2::Array{T,3}, J1::Array{T,3},
                                     Y2::Union{Nothing,Matrix{T}}, Y1, Z::PackedTrajectory{n,m,T}; ctx::Context=default_context()) where {n,m,T}
    N = length(Z)
    size(J1) == size(J2) == (n, n + m, N - 1) || throw(DimensionMismatch("J1, J2 must be n x (n+m) x (N-1)"))
    GC.@preserve J2 J1 Y2 Z begin
        rc = ccall((:rdhip_dynamics_error_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   ctx.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), N, Z.data, Z.t, Z.dt, J2, J1,
                   Y2 === nothing ? C_NULL : pointer(Y2), HOST)
    end
    check(ctx, rc)
end

# error-state expansion along a trajectory: E[:,:,k] = [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k] (what TrajectoryOptimization.jl's
# error_expansion! assembles per knot point); fused into the Jacobian kernel for RigidBody models
function error_expansion!(fun::RD.DiscretizedDynamics, E::Array{T,3}, Y::Union{Nothing,Matrix{T}}, Z::PackedTrajectory{n,m,T}; ctx::Context=default_context()) where {n,m,T}
    N = length(Z); e = RD.errstate_dim(fun)
    size(E) == (e, e + m, N - 1) || throw(DimensionMismatch("E must be e x (e+m) x (N-1)"))
    GC.@preserve E Y Z begin
        rc = ccall((:rdhip_error_expansion, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Clonglong, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   ctx.ptr, handle(fun), rulecode(fun), dtypecode(T), N, Z.data, Z.t, Z.dt, E, Y === nothing ? C_NULL : pointer(Y), HOST)
    end
    check(ctx, rc)
end

# rollout!(sig, dmodel, Z, x0) of M stacked trajectories followed by jacobian! at every simulated knot point, one call
# (the forward pass + linearisation of a shooting method).  Zs holds the M trajectories back to back (K knot points each).
function rollout_jacobian!(::RD.FunctionSignature, diff::RD.DiffMethod, fun::RD.DiscretizedDynamics, J::Array{T,3}, Zs::PackedTrajectory{n,m,T},
                           K::Integer, x0::Union{Nothing,Matrix{T}}=nothing; ctx::Context=default_context()) where {n,m,T}
    N = length(Zs); M = div(N, K)
    M * K == N || throw(DimensionMismatch("the stack must hold whole trajectories of K knot points"))
    size(J) == (n, n + m, N) || throw(DimensionMismatch("J must be n x (n+m) x (K*M)"))
    GC.@preserve J Zs x0 begin
        rc = ccall((:rdhip_rollout_jacobian, librdhip), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Clonglong, Clonglong, Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                   ctx.ptr, handle(fun), rulecode(fun), diffcode(diff), dtypecode(T), K, M, #= RDHIP_TRAJECTORY_MAJOR =# 0,
                   Zs.data, Zs.t, Zs.dt, x0 === nothing ? C_NULL : pointer(x0), J, HOST)
    end
    check(ctx, rc)
end

# HIP graphs: record the device-memory calls of one solver iteration once, replay them with one launch (small batches are launch-bound).
#   g = capture(ctx) do ... batched calls on device arrays ... end ;  launch(ctx, g)
mutable struct Graph
    ptr::Ptr{Cvoid}
end
function capture(f::Function, ctx::Context)
    check(ctx, ccall((:rdhip_graph_begin_capture, librdhip), Cint, (Ptr{Cvoid},), ctx.ptr))
    try
        f()
    finally
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:rdhip_graph_end_capture, librdhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.ptr, out)
        check(ctx, rc)
        g = Graph(out[])
        finalizer(x -> ccall((:rdhip_graph_destroy, librdhip), Cint, (Ptr{Cvoid},), x.ptr), g)
        return g
    end
end
launch(ctx::Context, g::Graph) = check(ctx, ccall((:rdhip_graph_launch, librdhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, g.ptr))

# v0.3 spellings (README.md:73-82,128-136; SURVEY.md App. B)
discrete_jacobian!(::Type{Q}, D::Array{T,3}, model::RD.ContinuousDynamics, Z::PackedTrajectory{n,m,T}) where {Q<:RD.QuadratureRule,n,m,T} =
    RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), RD.DiscretizedDynamics{Q}(model), D, nothing, Z)
state_diff_jacobian!(G::Array{T,3}, model::RD.AbstractModel, Z::PackedTrajectory{n,m,T}) where {n,m,T} = RD.errstate_jacobian!(model, G, Z)

end # module
