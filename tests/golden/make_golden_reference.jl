#!/usr/bin/env julia
# make_golden_reference.jl -- REFERENCE-generated golden vectors: the real RobotDynamics.jl (v0.4.8) evaluated on the inputs of the
# committed fixtures, written under the same keys as `tests/golden/ref_<name>.npz`.
#
# Why this exists: the build image has no Julia, so `tests/golden/*.npz` come from an independent torch restatement (make_golden.py) and the
# parity claim of this repository is "restatement-derived" (DESIGN.md section 2).  Anyone with a Julia install closes that gap with ONE command:
#
#     julia --project=@. -e 'using Pkg; Pkg.add(["RobotDynamics", "Rotations", "StaticArrays", "ForwardDiff", "FiniteDiff", "NPZ"])'   # once
#     julia tests/golden/make_golden_reference.jl            # writes tests/golden/ref_*.npz next to this file
#     python -m pytest tests -q                              # every test that loads a fixture now prefers the ref_ file (tests/helpers.py)
#
# What it pins that the restatement cannot (SURVEY App. A, all [3P] = third-party arithmetic not vendored in the reference checkout):
#   * Rotations.jl: `q * r` / `q \ r` for NON-unit quaternions inside the RK stages, `kinematics`, `∇differential` (the 4x3 values of G, not
#     just their placement), `∇²differential`, `rotation_error(q, q0, CayleyMap())`, and the MRP / RodriguesParam kinematics, rotations and
#     error maps (keys of ref_mrp_body.npz / ref_rp_body.npz: the product SERVES RigidBody{MRP} / RigidBody{RodriguesParam} dynamics, Jacobians,
#     rollouts, errstate_jacobian! and state_diff on the formulas of SURVEY App. A, pinned so far only by their definitions -- these files are
#     what would confirm or refute them; since round 4 also ∇errstate_jacobian! -- ref_nab_*.npz, Rotations.jl's ∇²differential of MRP / RodriguesParam
#     against the Hessian of the composition map the product implements -- and the error-state expansion, ref_exp_*_{mrp,rp}_*.npz);
#   * the operators added since the first generator: ImplicitMidpoint (Newton solve + ImplicitFunctionTheorem Jacobian, ref_implicit_*.npz),
#     rollout! (ref_rollout_*.npz) and explicitly time-dependent models (stage times t, t + h/2, t + h: ref_tv_*.npz);
#   * ForwardDiff.jl: `max(0, x)` tie rule in the quadrotor wrench;
#   * FiniteDiff.jl: the forward-difference step rule (key `Jfd_rk4`; the reference's own tests only hold it to 1e-6).
#
# Model definitions below are the reference's example / test models restated with the parameters stored in each fixture (citations inline).
using RobotDynamics, Rotations, StaticArrays, LinearAlgebra, ForwardDiff, FiniteDiff, NPZ
const RD = RobotDynamics
const HERE = @__DIR__

# ------------------------------------------------------------------------------------------------ models
# examples/cartpole.jl:5-38
RD.@autodiff struct Cartpole{T} <: RobotDynamics.ContinuousDynamics
    mc::T; mp::T; l::T; g::T
end
RD.state_dim(::Cartpole) = 4
RD.control_dim(::Cartpole) = 1
function RD.dynamics(model::Cartpole, x, u)
    mc, mp, l, g = model.mc, model.mp, model.l, model.g
    q = x[SA[1, 2]]; qd = x[SA[3, 4]]
    s = sin(q[2]); c = cos(q[2])
    H = @SMatrix [mc+mp mp*l*c; mp*l*c mp*l^2]
    C = @SMatrix [0 -mp*qd[2]*l*s; 0 0]
    G = @SVector [0, mp * g * l * s]
    B = @SVector [1, 0]
    qdd = -H \ (C * qd + G - B * u[1])
    return [qd; qdd]
end

# test/cartpole_model.jl:57-96: the hand-written continuous Jacobian -- what UserDefined() evaluates (dyn_case: key Jud_rk4; user_jacobian_case below)
function RD.jacobian!(model::Cartpole, J, xdot, x, u, t)
    mc, mp, l, g = model.mc, model.mp, model.l, model.g
    q = x[SA[1, 2]]; qd = x[SA[3, 4]]
    s = sin(q[2]); c = cos(q[2])
    H = @SMatrix [mc+mp mp*l*c; mp*l*c mp*l^2]
    C = @SMatrix [0 -mp*qd[2]*l*s; 0 0]
    G = @SVector [0, mp * g * l * s]
    B = @SVector [1, 0]
    qdd = -H \ (C * qd + G - B * u[1])
    dH = SA[0 -mp*l*s*qdd[2] 0 0 0; 0 -mp*l*s*qdd[1] 0 0 0]
    dC = SA[0 -mp*l*c*qd[2]^2 0 -2*mp*l*qd[2]*s 0; 0 0 0 0 0]
    dG = SA[0 0 0 0 0; 0 mp*g*l*c 0 0 0]
    dB = SA[0 0 0 0 1; 0 0 0 0 0]
    J[1:2, :] .= 0
    J[1, 3] = 1.0
    J[2, 4] = 1.0
    J[3:4, :] .= H \ (-dH - dC - dG + dB)
    return nothing
end

# test/discrete_dynamics_test.jl:5-63: a model that IS a discrete map (no integrator) with its own Jacobian
struct DiscreteDI <: RobotDynamics.DiscreteDynamics end
RD.state_dim(::DiscreteDI) = 4
RD.control_dim(::DiscreteDI) = 2
RD.discrete_dynamics(::DiscreteDI, x, u, t, dt) = SA[x[1] + x[3] * dt, x[2] + x[4] * dt, x[3] + u[1] * dt, x[4] + u[2] * dt]
function RD.discrete_dynamics!(::DiscreteDI, xn, x, u, t, dt)
    xn[1] = x[1] + x[3] * dt; xn[2] = x[2] + x[4] * dt; xn[3] = x[3] + u[1] * dt; xn[4] = x[4] + u[2] * dt
    return nothing
end
function RD.jacobian!(::DiscreteDI, J, xn, x, u, t, dt)
    J .= 0
    J[1, 1] = 1; J[1, 3] = dt; J[2, 2] = 1; J[2, 4] = dt; J[3, 3] = 1; J[3, 5] = dt; J[4, 4] = 1; J[4, 6] = dt
    return nothing
end
function RD.jacobian!(::RD.StaticReturn, ::RD.ForwardAD, model::DiscreteDI, J, y, z)
    f(_z) = RD.evaluate(model, RD.getstate(z, _z), RD.getcontrol(z, _z), RD.getparams(z))
    J .= ForwardDiff.jacobian(f, RD.getdata(z))
end

# test/double_integrator.jl:97-107
RD.@autodiff struct DI{D} <: RobotDynamics.ContinuousDynamics end
RD.state_dim(::DI{D}) where {D} = 2D
RD.control_dim(::DI{D}) where {D} = D
RD.dynamics(::DI{D}, x, u) where {D} = SVector{2D}(ntuple(i -> i <= D ? x[D+i] : u[i-D], 2D))

# xdot = A x + B u
RD.@autodiff struct Lin{n,m} <: RobotDynamics.ContinuousDynamics
    A::SMatrix{n,n,Float64}
    B::SMatrix{n,m,Float64}
end
RD.state_dim(::Lin{n}) where {n} = n
RD.control_dim(::Lin{n,m}) where {n,m} = m
RD.dynamics(model::Lin, x, u) = model.A * x + model.B * u

# examples/single_satellite.jl:7-27 ; test/rigidbody_test.jl:23-54 (`Body`)
RD.@autodiff struct Sat{R} <: RobotDynamics.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    bodyframe::Bool
end
RD.control_dim(::Sat) = 6
RD.mass(model::Sat) = model.mass
RD.inertia(model::Sat) = model.J
RD.velocity_frame(model::Sat) = model.bodyframe ? :body : :world
RD.forces(model::Sat, x, u) = RD.orientation(model, x) * SA[u[1], u[2], u[3]]
RD.moments(model::Sat, x, u) = SA[u[4], u[5], u[6]]

# test/quadrotor.jl:21-137
RD.@autodiff struct Quad{R} <: RobotDynamics.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    gravity::SVector{3,Float64}
    motor_dist::Float64
    kf::Float64
    km::Float64
    bodyframe::Bool
end
RD.control_dim(::Quad) = 4
RD.mass(model::Quad) = model.mass
RD.inertia(model::Quad) = model.J
RD.velocity_frame(model::Quad) = model.bodyframe ? :body : :world
function RD.forces(model::Quad, x, u)
    q = RD.orientation(model, x)
    kf = model.kf
    F1 = max(0, kf * u[1]); F2 = max(0, kf * u[2]); F3 = max(0, kf * u[3]); F4 = max(0, kf * u[4])
    return model.mass * model.gravity + q * SA[0.0, 0.0, F1 + F2 + F3 + F4]
end
function RD.moments(model::Quad, x, u)
    kf, km, L = model.kf, model.km, model.motor_dist
    F1 = max(0, kf * u[1]); F2 = max(0, kf * u[2]); F3 = max(0, kf * u[3]); F4 = max(0, kf * u[4])
    return SA[L * (F2 - F4), L * (F3 - F1), km * u[1] - km * u[2] + km * u[3] - km * u[4]]
end

# examples/satellite_pair.jl:8-40
RD.@autodiff struct Pair{R} <: RobotDynamics.LieGroupModel
    J1::SMatrix{3,3,Float64,9}
    J2::SMatrix{3,3,Float64,9}
end
RD.control_dim(::Pair) = 6
RD.LieState(::Pair{R}) where {R} = RD.LieState(R, (0, 3, 3))
function RD.dynamics(model::Pair, x, u)
    vs = RD.vec_states(model, x); qs = RD.rot_states(model, x)
    w1, w2 = vs[2], vs[3]
    J1, J2 = model.J1, model.J2
    w1d = J1 \ (u[SA[1, 2, 3]] - w1 × (J1 * w1))
    w2d = J2 \ (u[SA[4, 5, 6]] - w2 × (J2 * w2))
    q1d = Rotations.kinematics(qs[1], w1); q2d = Rotations.kinematics(qs[2], w2)
    return SA[q1d[1], q1d[2], q1d[3], q1d[4], w1d[1], w1d[2], w1d[3], q2d[1], q2d[2], q2d[3], q2d[4], w2d[1], w2d[2], w2d[3]]
end

# test/liemodel.jl:1-22
RD.@autodiff struct LieSat <: RobotDynamics.LieGroupModel
    J::Diagonal{Float64,SVector{3,Float64}}
end
RD.control_dim(::LieSat) = 3
RD.LieState(::LieSat) = RD.LieState(QuatRotation{Float64}, (3, 0))
function RD.dynamics(model::LieSat, x, u)
    w = SA[x[1], x[2], x[3]]
    q = normalize(SA[x[4], x[5], x[6], x[7]])
    J = model.J
    wd = J \ (u - w × (J * w))
    qd = 0.5 * Rotations.lmult(q) * Rotations.hmat() * w
    return [wd; qd]
end

# a LieGroupModel that is nothing but its LieState{R,P} (for the err_* fixtures, test/liestate.jl:33-99)
struct LG{R,P} <: RobotDynamics.LieGroupModel end
RD.control_dim(::LG) = 1
RD.LieState(::LG{R,P}) where {R,P} = RD.LieState(R, P)

const QR = QuatRotation{Float64}
S33(A) = SMatrix{3,3,Float64,9}(A)
# the rotation type a fixture was generated for (key "rot": 0 QuatRotation, 1 MRP, 2 RodriguesParam; absent = QuatRotation)
function rot_type(d)
    r = haskey(d, "rot") ? Int(d["rot"][]) : 0
    return r == 1 ? MRP{Float64} : r == 2 ? RodriguesParam{Float64} : QR
end

function model_for(name::String, d)
    if startswith(name, "cartpole")
        p = d["params"]; return Cartpole{Float64}(p[1], p[2], p[3], p[4])
    elseif startswith(name, "di")
        return DI{Int(d["params"][1])}()
    elseif startswith(name, "linear")
        A, B = d["A"], d["B"]; n, m = size(B)
        return Lin{n,m}(SMatrix{n,n,Float64}(A), SMatrix{n,m,Float64}(B))
    elseif startswith(name, "satellite_") || startswith(name, "body_") || startswith(name, "satfull_")
        return Sat{rot_type(d)}(Float64(d["mass"][]), S33(d["J"]), d["frame"][] != 0)
    elseif startswith(name, "quadrotor")
        return Quad{rot_type(d)}(Float64(d["mass"][]), S33(d["J"]), SVector{3,Float64}(d["gravity"]), Float64(d["L"][]), Float64(d["kf"][]), Float64(d["km"][]), d["frame"][] != 0)
    elseif startswith(name, "satpair")
        return Pair{QR}(S33(d["J1"]), S33(d["J2"]))
    elseif startswith(name, "liesat")
        return LieSat(Diagonal(SVector{3,Float64}(d["Jd"])))
    end
    error("unknown fixture $name")
end

knot(Z, t, dt, k, n, m) = RD.KnotPoint{n,m}(SVector{n + m}(Z[k, :]), t[k], dt[k])

# ------------------------------------------------------------------------------------------------ dyn_* : dynamics, steps, Jacobians
function dyn_case(name)
    d = npzread(joinpath(HERE, "dyn_$name.npz"))
    model = model_for(name, d)
    n, m = RD.state_dim(model), RD.control_dim(model)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    N = size(Z, 1)
    out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt)
    out["xdot"] = zeros(N, n); out["Jc"] = zeros(N, n, n + m)
    rules = (("euler", RD.Euler), ("rk3", RD.RK3), ("rk4", RD.RK4))          # RK2 does not exist in v0.4.8 (SURVEY App. B): no reference value
    for (rn, _) in rules
        out["xn_$rn"] = zeros(N, n); out["Jd_$rn"] = zeros(N, n, n + m)
    end
    out["Jfd_rk4"] = zeros(N, n, n + m); out["Jud_rk4"] = fill(NaN, N, n, n + m)
    J = zeros(n, n + m); y = zeros(n)
    for k = 1:N
        z = knot(Z, t, dt, k, n, m)
        out["xdot"][k, :] = RD.dynamics(model, z)
        if model isa RD.RigidBody     # wrenches(model, x, u, t) and its ForwardDiff Jacobian (what the default wrench_jacobian! is written to return, src/rigidbody.jl:315-320)
            haskey(out, "W") || (out["W"] = zeros(N, 6); out["Fw"] = zeros(N, 6, n + m))
            out["W"][k, :] = RD.wrenches(model, RD.state(z), RD.control(z), t[k])
            out["Fw"][k, :, :] = ForwardDiff.jacobian(zz -> RD.wrenches(model, zz[SVector{n}(1:n)], zz[SVector{m}(n+1:n+m)], t[k]), Vector(z.z))
        end
        RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), model, J, y, z); out["Jc"][k, :, :] = J
        for (rn, Q) in rules
            dm = RD.DiscretizedDynamics{Q}(model)
            out["xn_$rn"][k, :] = RD.discrete_dynamics(dm, z)
            RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, z); out["Jd_$rn"][k, :, :] = J
        end
        dm4 = RD.DiscretizedDynamics{RD.RK4}(model)
        RD.jacobian!(RD.StaticReturn(), RD.FiniteDifference(), dm4, J, y, z); out["Jfd_rk4"][k, :, :] = J
        try        # UserDefined: only models that define jacobian!(model, J, y, x, u, t)
            RD.jacobian!(RD.StaticReturn(), RD.UserDefined(), dm4, J, y, z); out["Jud_rk4"][k, :, :] = J
        catch
        end
    end
    npzwrite(joinpath(HERE, "ref_dyn_$name.npz"), out)
end

# ------------------------------------------------------------------------------------------------ exp_* : dynamics error and error-state expansion
function exp_case(name)
    d = npzread(joinpath(HERE, "dyn_$name.npz"))
    model = model_for(name, d)
    n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    N = size(Z, 1)
    out = Dict{String,Any}()
    for (rn, Q) in (("rk3", RD.RK3), ("rk4", RD.RK4))
        dm = RD.DiscretizedDynamics{Q}(model)
        E = zeros(N - 1, e, e + m); err = zeros(N - 1, n)
        J = zeros(n, n + m); y = zeros(n)
        for k = 1:N-1
            z1, z2 = knot(Z, t, dt, k, n, m), knot(Z, t, dt, k + 1, n, m)
            err[k, :] = RD.dynamics_error(dm, z2, z1)
            RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, z1)
            G1 = zeros(n, e); G2 = zeros(n, e)
            RD.errstate_jacobian!(model, G1, z1); RD.errstate_jacobian!(model, G2, z2)
            E[k, :, 1:e] = G2' * J[:, 1:n] * G1
            E[k, :, e+1:e+m] = G2' * J[:, n+1:n+m]
        end
        out["E_$rn"] = E; out["err_$rn"] = err
    end
    npzwrite(joinpath(HERE, "ref_exp_$name.npz"), out)
end

# ------------------------------------------------------------------------------------------------ err_* : errstate_jacobian!, ∇errstate_jacobian!, state_diff
function err_case(tag)
    d = npzread(joinpath(HERE, "err_$tag.npz"))
    P = Tuple(Int.(d["P"]))
    model = LG{rot_type(d),P}()
    X, X0 = d["X"], d["X0"]
    Xb = haskey(d, "Xbar") ? d["Xbar"] : nothing          # (the MRP / RodriguesParam files of make_golden_rot3.py keep their second-order inputs in nab_*.npz)
    N, n = size(X); e = RD.errstate_dim(model)
    G = zeros(N, n, e); dG = zeros(N, e, e); dX = zeros(N, e)
    for k = 1:N
        x, x0 = SVector{n}(X[k, :]), SVector{n}(X0[k, :])
        Gk = zeros(n, e); RD.errstate_jacobian!(model, Gk, x); G[k, :, :] = Gk
        if Xb !== nothing
            Hk = zeros(e, e); RD.∇errstate_jacobian!(model, Hk, x, SVector{n}(Xb[k, :])); dG[k, :, :] = Hk
        end
        dX[k, :] = RD.state_diff(model, x, x0)
    end
    out = Dict{String,Any}("G" => G, "dX" => dX, "X" => X, "X0" => X0, "P" => d["P"])
    if Xb !== nothing
        out["dG"] = dG; out["Xbar"] = Xb
    end
    npzwrite(joinpath(HERE, "ref_err_$tag.npz"), out)
end

# ------------------------------------------------------------------------------------------------ errmap_* : state_diff(model, x, x0, errmap) (src/liestate.jl:26-32,216-217)
# tests/golden/make_golden_round6.py's inputs; keys dX_cayley, dX_exponential, dX_mrp, dX_quatvec, dX_identity (NaN where the reference throws: IdentityMap of a
# QuatRotation state does not fit a three-element RotationError)
function errmap_case(tag)
    d = npzread(joinpath(HERE, "errmap_$tag.npz"))
    P = Tuple(Int.(d["P"]))
    model = LG{rot_type(d),P}()
    X, X0 = d["X"], d["X0"]
    N, n = size(X); e = RD.errstate_dim(model)
    out = Dict{String,Any}("X" => X, "X0" => X0, "P" => d["P"], "rot" => d["rot"])
    for (key, emap) in (("cayley", Rotations.CayleyMap()), ("exponential", Rotations.ExponentialMap()), ("mrp", Rotations.MRPMap()), ("quatvec", Rotations.QuatVecMap()),
                        ("identity", Rotations.IdentityMap()))
        dX = fill(NaN, N, e)
        try
            for k = 1:N
                dX[k, :] = RD.state_diff(model, SVector{n}(X[k, :]), SVector{n}(X0[k, :]), emap)
            end
        catch err
            dX .= NaN
        end
        out["dX_$key"] = dX
    end
    npzwrite(joinpath(HERE, "ref_errmap_$tag.npz"), out)
end

# ------------------------------------------------------------------------------------------------ nab_* : ∇errstate_jacobian! of LieState{MRP | RodriguesParam, P} (src/liestate.jl:300-320)
function nab_case(tag)
    d = npzread(joinpath(HERE, "nab_$tag.npz"))
    P = Tuple(Int.(d["P"]))
    model = LG{rot_type(d),P}()
    X, Xb = d["X"], d["Xbar"]
    N, n = size(X); e = RD.errstate_dim(model)
    dG = zeros(N, e, e)
    for k = 1:N
        Hk = zeros(e, e); RD.∇errstate_jacobian!(model, Hk, SVector{n}(X[k, :]), SVector{n}(Xb[k, :])); dG[k, :, :] = Hk
    end
    npzwrite(joinpath(HERE, "ref_nab_$tag.npz"), Dict("dG" => dG, "X" => X, "Xbar" => Xb, "P" => d["P"]))
end

# ------------------------------------------------------------------------------------------------ RigidBody{MRP} / RigidBody{RodriguesParam} (unpinned [3P] formulas)
function three_param_case(::Type{R}, tag) where {R}
    d = npzread(joinpath(HERE, "dyn_body_world.npz"))
    Zq = d["Z"]; N = size(Zq, 1)
    model = Sat{R}(2.0, S33(Diagonal(SA[2.0, 3.0, 1.0])), false)
    n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)      # 12, 6, 12
    Z = zeros(N, n + m)
    for k = 1:N                                               # the same attitudes, re-parameterised
        q = QuatRotation(Zq[k, 4], Zq[k, 5], Zq[k, 6], Zq[k, 7])
        Z[k, :] = [Zq[k, 1:3]; Rotations.params(R(q)); Zq[k, 8:13]; Zq[k, 14:19]]
    end
    out = Dict{String,Any}("Z" => Z, "t" => d["t"], "dt" => d["dt"], "xdot" => zeros(N, n), "xn_rk4" => zeros(N, n), "Jd_rk4" => zeros(N, n, n + m),
                           "G" => zeros(N, n, e), "dX" => zeros(N, e))
    dm = RD.DiscretizedDynamics{RD.RK4}(model)
    J = zeros(n, n + m); y = zeros(n)
    for k = 1:N
        z = knot(Z, d["t"], d["dt"], k, n, m)
        out["xdot"][k, :] = RD.dynamics(model, z)
        out["xn_rk4"][k, :] = RD.discrete_dynamics(dm, z)
        RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, z); out["Jd_rk4"][k, :, :] = J
        Gk = zeros(n, e); RD.errstate_jacobian!(model, Gk, z); out["G"][k, :, :] = Gk
        k2 = k == N ? 1 : k + 1
        out["dX"][k, :] = RD.state_diff(model, SVector{n}(Z[k, 1:n]), SVector{n}(Z[k2, 1:n]))
    end
    npzwrite(joinpath(HERE, "ref_$(tag)_body.npz"), out)
end

# ------------------------------------------------------------------------------------------------ ImplicitMidpoint: Newton solve + IFT Jacobian
# src/integration.jl:422-463,517-538 (defaults: 10 iterations, tolerance 1e-12); keys xn_implicit, J_ift
function implicit_case(name)
    d = npzread(joinpath(HERE, "dyn_$name.npz"))
    model = model_for(name, d)
    n, m = RD.state_dim(model), RD.control_dim(model)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    N = size(Z, 1)
    dm = RD.DiscretizedDynamics{RD.ImplicitMidpoint}(model)
    out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt, "xn_implicit" => zeros(N, n), "J_ift" => zeros(N, n, n + m))
    J = zeros(n, n + m); y = zeros(n)
    for k = 1:N
        z = knot(Z, t, dt, k, n, m)
        RD.evaluate!(dm, y, z); out["xn_implicit"][k, :] = y
        RD.jacobian!(RD.InPlace(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dm, J, y, z); out["J_ift"][k, :, :] = J
    end
    npzwrite(joinpath(HERE, "ref_implicit_$name.npz"), out)
end

# ------------------------------------------------------------------------------------------------ rollout! (src/trajectories.jl:436-441)
# the first 8 knot points of a dyn_* fixture as ONE trajectory: controls, t, dt as stored, x0 = the first state; key X (8 x n)
function rollout_case(name, Q, rn)
    d = npzread(joinpath(HERE, "dyn_$name.npz"))
    model = model_for(name, d)
    n, m = RD.state_dim(model), RD.control_dim(model)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    K = min(8, size(Z, 1))
    dm = RD.DiscretizedDynamics{Q}(model)
    traj = RD.SampledTrajectory([knot(Z, t, dt, k, n, m) for k = 1:K])
    RD.rollout!(RD.StaticReturn(), dm, traj, SVector{n}(Z[1, 1:n]))
    X = zeros(K, n)
    for k = 1:K
        X[k, :] = RD.state(traj[k])
    end
    npzwrite(joinpath(HERE, "ref_rollout_$(name)_$rn.npz"), Dict("Z" => Z[1:K, :], "t" => t[1:K], "dt" => dt[1:K], "X" => X))
end

# ------------------------------------------------------------------------------------------------ explicitly time-dependent dynamics
# dynamics(model, x, u, t) (src/dynamics.jl:81-83): the two test models of oracle/rd_oracle.cpp (M_TV_PENDULUM, M_TV_SATELLITE)
RD.@autodiff struct TVPendulum <: RobotDynamics.ContinuousDynamics
    a::Float64; b::Float64; w::Float64; c::Float64
end
RD.state_dim(::TVPendulum) = 2
RD.control_dim(::TVPendulum) = 1
RD.dynamics(p::TVPendulum, x, u, t) = SA[x[2], -(p.a + p.b * cos(p.w * t)) * sin(x[1]) - p.c * x[2] + u[1] + p.b * sin(p.w * t)]
RD.@autodiff struct TVSat{R} <: RobotDynamics.RigidBody{R}
    mass::Float64
    J::SMatrix{3,3,Float64,9}
    fd::Float64; td::Float64; w::Float64
end
RD.control_dim(::TVSat) = 6
RD.mass(model::TVSat) = model.mass
RD.inertia(model::TVSat) = model.J
RD.wrenches(model::TVSat, x, u, t) = begin          # F += fd [cos wt, 0, sin wt] (world), tau += td [sin wt, cos wt, sin 2wt]
    q = RD.orientation(model, x); wt = model.w * t
    F = q * SA[u[1], u[2], u[3]] + model.fd * SA[cos(wt), 0.0, sin(wt)]
    T = SA[u[4], u[5], u[6]] + model.td * SA[sin(wt), cos(wt), sin(2wt)]
    return [F; T]
end
function tv_case()
    rngZ(N, nz) = [0.1 + 0.8 * ((7919 * (i + 31 * j)) % 1000) / 1000 for i = 1:N, j = 1:nz]
    for (tag, model) in (("pendulum", TVPendulum(2.0, 0.7, 3.0, 0.1)), ("satellite", TVSat{QR}(2.0, S33(Diagonal(SA[2.0, 3.0, 1.0])), 0.3, 0.2, 1.7)))
        n, m = RD.state_dim(model), RD.control_dim(model)
        N = 16
        Z = rngZ(N, n + m); t = collect(0:N-1) .* 0.13 .+ 0.4; dt = fill(0.05, N)
        if n == 13
            for k = 1:N
                Z[k, 4:7] = normalize(Z[k, 4:7] .- 0.5)
            end
        end
        out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt, "xdot" => zeros(N, n), "xn_rk4" => zeros(N, n), "Jd_rk4" => zeros(N, n, n + m), "xn_rk3" => zeros(N, n))
        J = zeros(n, n + m); y = zeros(n)
        for k = 1:N
            z = knot(Z, t, dt, k, n, m)
            out["xdot"][k, :] = RD.dynamics(model, z)
            out["xn_rk3"][k, :] = RD.discrete_dynamics(RD.DiscretizedDynamics{RD.RK3}(model), z)
            dm = RD.DiscretizedDynamics{RD.RK4}(model)
            out["xn_rk4"][k, :] = RD.discrete_dynamics(dm, z)
            RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, z); out["Jd_rk4"][k, :, :] = J
        end
        npzwrite(joinpath(HERE, "ref_tv_$tag.npz"), out)
    end
end

# ------------------------------------------------------------------------------------------------ round 5: discrete models, user Jacobians, NaN knot points
# discrete_di.npz / userjac_cartpole.npz / nan_satellite.npz hold the inputs (tests/golden/make_golden_round5.py); the product's plug-in bodies
# (rdhip_model_create_custom_discrete, rdhip_model_set_user_jacobian) and its NaN contract (include/rdhip.h) are tested against these quantities
function discrete_case()
    d = npzread(joinpath(HERE, "discrete_di.npz"))
    Z, t, dt = d["Z"], d["t"], d["dt"]; N = size(Z, 1)
    model = DiscreteDI()
    out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt, "xn" => zeros(N, 4), "J_user" => zeros(N, 4, 6), "J_ad" => zeros(N, 4, 6),
                           "e" => zeros(N - 1, 4), "J1" => zeros(N - 1, 4, 6), "J2" => zeros(N - 1, 4, 6))
    J = zeros(4, 6); J2 = zeros(4, 6); y = zeros(4); y2 = zeros(4)
    for k = 1:N
        z = knot(Z, t, dt, k, 4, 2)
        out["xn"][k, :] = RD.discrete_dynamics(model, z)
        RD.jacobian!(RD.StaticReturn(), RD.UserDefined(), model, J, y, z); out["J_user"][k, :, :] = J
        RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), model, J, y, z); out["J_ad"][k, :, :] = J
        if k < N
            z2 = knot(Z, t, dt, k + 1, 4, 2)
            out["e"][k, :] = RD.dynamics_error(model, z2, z)
            RD.dynamics_error_jacobian!(RD.StaticReturn(), RD.UserDefined(), model, J2, J, y2, y, z2, z)       # src/discrete_dynamics.jl:111-113, 160-176
            out["J1"][k, :, :] = J; out["J2"][k, :, :] = J2
        end
    end
    npzwrite(joinpath(HERE, "ref_discrete_di.npz"), out)
end
function user_jacobian_case()
    d = npzread(joinpath(HERE, "userjac_cartpole.npz"))
    Z, t, dt = d["Z"], d["t"], d["dt"]; N = size(Z, 1)
    p = d["params"]; model = Cartpole(p[1], p[2], p[3], p[4])
    out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt, "Jc_user" => zeros(N, 4, 5))
    J = zeros(4, 5); y = zeros(4)
    rules = (("euler", RD.Euler), ("rk3", RD.RK3), ("rk4", RD.RK4))
    for (rn, _) in rules
        out["Jd_user_$rn"] = zeros(N, 4, 5)
    end
    for k = 1:N
        z = knot(Z, t, dt, k, 4, 1)
        RD.jacobian!(RD.StaticReturn(), RD.UserDefined(), model, J, y, z); out["Jc_user"][k, :, :] = J
        for (rn, Q) in rules          # the integrators' chain rule on the stage Jacobians (src/integration.jl:149-177, 302-337)
            RD.jacobian!(RD.StaticReturn(), RD.UserDefined(), RD.DiscretizedDynamics{Q}(model), J, y, z); out["Jd_user_$rn"][k, :, :] = J
        end
    end
    npzwrite(joinpath(HERE, "ref_userjac_cartpole.npz"), out)
end
function nan_case()        # a NaN-filled knot point (src/trajectories.jl:83) and dt = Inf on the last one (:104-108) through jacobian! and discrete_dynamics
    d = npzread(joinpath(HERE, "nan_satellite.npz"))
    Z, t, dt = d["Z"], d["t"], d["dt"]; N = size(Z, 1)
    model = Sat{QR}(Float64(d["mass"][]), S33(d["J"]), false)
    dm = RD.DiscretizedDynamics{RD.RK4}(model)
    out = Dict{String,Any}("Z" => Z, "t" => t, "dt" => dt, "xn_rk4" => zeros(N, 13), "Jd_rk4" => zeros(N, 13, 19))
    J = zeros(13, 19); y = zeros(13)
    for k = 1:N
        z = knot(Z, t, dt, k, 13, 6)
        out["xn_rk4"][k, :] = RD.discrete_dynamics(dm, z)
        RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, z); out["Jd_rk4"][k, :, :] = J
    end
    npzwrite(joinpath(HERE, "ref_nan_satellite.npz"), out)
end

# ------------------------------------------------------------------------------------------------ the Julia host of this repository, once per wrapper
# (robotdynamics.jl_amd/julia/RobotDynamicsHIP.jl needs librdhip.so and a GPU: skipped with a message otherwise).  Every batched wrapper is
# called once on a small satellite batch and compared with the reference's per-knot loop computed above in this same session.
function host_smoke()
    jl = normpath(joinpath(HERE, "..", "..", "robotdynamics.jl_amd", "julia", "RobotDynamicsHIP.jl"))
    try
        include(jl)
    catch err
        println("RobotDynamicsHIP not exercised (", sprint(showerror, err), ")"); return
    end
    H = Main.RobotDynamicsHIP
    if H.device_count() == 0
        println("RobotDynamicsHIP loaded; no GPU: wrappers not called"); return
    end
    d = npzread(joinpath(HERE, "dyn_satellite_world.npz"))
    Z, t, dt = d["Z"], d["t"], d["dt"]; N = size(Z, 1)
    ref = Sat{QR}(Float64(d["mass"][]), S33(d["J"]), false)
    hm = H.Satellite(Float64(d["mass"][]), Matrix(d["J"]))
    dm = RD.DiscretizedDynamics{RD.RK4}(hm)
    P = H.PackedTrajectory{13,6,Float64}(permutedims(Z), t, dt)
    J = zeros(13, 19, N); Y = zeros(13, N)
    RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, P)
    Jr = zeros(13, 19); yr = zeros(13); worst = 0.0
    for k = 1:N
        RD.jacobian!(RD.StaticReturn(), RD.ForwardAD(), RD.DiscretizedDynamics{RD.RK4}(ref), Jr, yr, knot(Z, t, dt, k, 13, 6))
        worst = max(worst, maximum(abs.(J[:, :, k] .- Jr)), maximum(abs.(Y[:, k] .- yr)))
    end
    println("RobotDynamicsHIP jacobian! vs RobotDynamics per-knot loop: max |diff| = ", worst)
    Xd = zeros(13, N); RD.dynamics!(hm, Xd, P)
    Xn = zeros(13, N); RD.discrete_dynamics!(dm, Xn, P)
    G = zeros(13, 12, N); RD.errstate_jacobian!(hm, G, P)
    dX = zeros(12, N); RD.state_diff!(hm, dX, Z[:, 1:13]', circshift(Z[:, 1:13], (1, 0))')
    E = zeros(12, 18, N - 1); H.error_expansion!(dm, E, nothing, P)
    H.rollout_jacobian!(RD.StaticReturn(), RD.ForwardAD(), dm, J, P, N)
    println("RobotDynamicsHIP: dynamics!, discrete_dynamics!, errstate_jacobian!, state_diff!, error_expansion!, rollout_jacobian! ran")
end

function main()
    dyn = sort([replace(basename(f), r"^dyn_|\.npz$" => "") for f in readdir(HERE) if startswith(f, "dyn_") && endswith(f, ".npz")])
    foreach(dyn_case, dyn)
    foreach(exp_case, [replace(basename(f), r"^exp_|\.npz$" => "") for f in readdir(HERE) if startswith(f, "exp_") && endswith(f, ".npz")])
    foreach(err_case, [replace(basename(f), r"^err_|\.npz$" => "") for f in readdir(HERE) if startswith(f, "err_") && endswith(f, ".npz")])
    foreach(nab_case, [replace(basename(f), r"^nab_|\.npz$" => "") for f in readdir(HERE) if startswith(f, "nab_") && endswith(f, ".npz")])
    foreach(errmap_case, [replace(basename(f), r"^errmap_|\.npz$" => "") for f in readdir(HERE) if startswith(f, "errmap_") && endswith(f, ".npz")])
    three_param_case(MRP{Float64}, "mrp")
    three_param_case(RodriguesParam{Float64}, "rp")
    foreach(implicit_case, ["cartpole", "satellite_world", "quadrotor_world", "di2"])
    for (Q, rn) in ((RD.RK4, "rk4"), (RD.RK3, "rk3"), (RD.ImplicitMidpoint, "implicit")), name in ("cartpole", "satellite_world", "quadrotor_body")
        rollout_case(name, Q, rn)
    end
    tv_case()
    discrete_case(); user_jacobian_case(); nan_case()
    host_smoke()
    println("wrote reference-generated fixtures ref_*.npz to ", HERE, " (RobotDynamics ", pkgversion(RD), ")")
end

main()
