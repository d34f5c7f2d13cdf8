"""User-defined models for the hipRTC plug-in (rdhip.CustomModel): the reference's example models re-written as HIP device-function
bodies on a generic scalar S.  `x`, `u`, `xdot` are arrays of S; `p[i]` the parameters (type T).  Used by the tests and by
tools/jit_bench.py."""

CARTPOLE_SRC = """
    // examples/cartpole.jl:15-34 written as a HIP device-function body: p = {mc, mp, l, g}
    const S s = sin_(x[1]), c = cos_(x[1]);
    const T H11 = p[0] + p[1], H22 = p[1] * p[2] * p[2];
    const S H12 = c * (p[1] * p[2]);
    const S r0 = (x[3] * x[3]) * s * (-(p[1] * p[2])) - u[0];
    const S r1 = s * (p[1] * p[3] * p[2]);
    const S det = H12 * H12 * T(-1) + H11 * H22;
    xdot[0] = x[2]; xdot[1] = x[3];
    xdot[2] = -((r0 * H22 - H12 * r1) / det);
    xdot[3] = -((r1 * H11 - H12 * r0) / det);
"""

SATELLITE_SRC = """
    // src/rigidbody.jl:213-236 with the satellite wrench (examples/single_satellite.jl:18-27), world-frame velocity; p = {mass, Jxx, Jyy, Jzz}
    const S* q = x + 3; const S* v = x + 7; const S* w = x + 10;
    S F[3]; quat_rotate(q, u, F);
    quat_kinematics(q, w, xdot + 3);
    const T im = T(1) / p[0];
    for (int i = 0; i < 3; ++i) { xdot[i] = v[i]; xdot[7 + i] = F[i] * im; }
    const S Jw[3] = {w[0] * p[1], w[1] * p[2], w[2] * p[3]};
    S c[3]; cross3(w, Jw, c);
    xdot[10] = (u[3] - c[0]) / p[1]; xdot[11] = (u[4] - c[1]) / p[2]; xdot[12] = (u[5] - c[2]) / p[3];
"""

# Explicitly time-dependent dynamics: `t` (type T) is the time the dynamics are evaluated at -- the knot point's t for `dynamics`, the stage
# times t, t + h/2, t + h inside the Runge-Kutta rules (src/integration.jl:130-135,280-286), exactly what `dynamics(model, x, u, t)` receives
# in the reference.  These two are test models (the reference's own example models are time-invariant).
TV_PENDULUM_SRC = """
    // parametrically driven pendulum: thdd = -(a + b cos(w t)) sin(th) - c thd + u + b sin(w t);  p = {a, b, w, c}
    xdot[0] = x[1];
    xdot[1] = -(sin_(x[0]) * (p[0] + p[1] * cos_(p[2] * t))) - x[1] * p[3] + u[0] + p[1] * sin_(p[2] * t);
"""

TV_SATELLITE_SRC = """
    // the satellite of SATELLITE_SRC with a time-dependent disturbance wrench: F += fd [cos wt, 0, sin wt] (world frame),
    // tau += td [sin wt, cos wt, sin 2wt];  p = {mass, Jxx, Jyy, Jzz, fd, td, w}
    const S* q = x + 3; const S* v = x + 7; const S* w = x + 10;
    S F[3]; quat_rotate(q, u, F);
    const T wt = p[6] * t;
    F[0] = F[0] + p[4] * cos_(wt); F[2] = F[2] + p[4] * sin_(wt);
    quat_kinematics(q, w, xdot + 3);
    const T im = T(1) / p[0];
    for (int i = 0; i < 3; ++i) { xdot[i] = v[i]; xdot[7 + i] = F[i] * im; }
    const S Jw[3] = {w[0] * p[1], w[1] * p[2], w[2] * p[3]};
    S c[3]; cross3(w, Jw, c);
    xdot[10] = (u[3] + p[5] * sin_(wt) - c[0]) / p[1]; xdot[11] = (u[4] + p[5] * cos_(wt) - c[1]) / p[2]; xdot[12] = (u[5] + p[5] * sin_(T(2) * wt) - c[2]) / p[3];
"""

# The reference's way of defining a rigid body: subtype RigidBody{R} and write `forces` (world frame) and `moments` (body frame),
# src/rigidbody.jl:244-258 -- here the quadrotor of test/quadrotor.jl:56-96.  rdhip.CustomRigidBody(4, QUADROTOR_FORCES_SRC, QUADROTOR_MOMENTS_SRC,
# mass=0.5, J=diag(.0023, .0023, .004), params=(0, 0, -9.81, 0.175, 1.0, 0.0245)) generates the Newton-Euler dynamics around the two snippets,
# attaches LieState{QuatRotation,(3,6)} and declares the block structure w | q | r, v (verified on first GPU use).  In a snippet: q (attitude,
# 4 values), r, v, w, x, u, t; user parameters P[i] (here gravity 0-2, motor distance 3, kf 4, km 5); the mass is p[0].
QUADROTOR_FORCES_SRC = """
    const S F1 = max0_(u[0] * P[4]), F2 = max0_(u[1] * P[4]), F3 = max0_(u[2] * P[4]), F4 = max0_(u[3] * P[4]);
    const S fb[3] = {S(T(0)), S(T(0)), F1 + F2 + F3 + F4};
    S qF[3]; quat_rotate(q, fb, qF);
    for (int i = 0; i < 3; ++i) F[i] = qF[i] + p[0] * P[i];
"""
QUADROTOR_MOMENTS_SRC = """
    const S F1 = max0_(u[0] * P[4]), F2 = max0_(u[1] * P[4]), F3 = max0_(u[2] * P[4]), F4 = max0_(u[3] * P[4]);
    tau[0] = (F2 - F4) * P[3]; tau[1] = (F3 - F1) * P[3];
    tau[2] = (u[0] - u[1] + u[2] - u[3]) * P[5];
"""
