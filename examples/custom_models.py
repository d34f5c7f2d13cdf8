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
