// models.hpp -- the compiled-in model registry: continuous dynamics xdot = f(x,u) written ONCE as templates
// on the scalar type S (T or Dual<T,K>), plus the explicit Runge-Kutta rules.
// Reference formulas (file:line in RobotDynamics.jl):
//   Cartpole            examples/cartpole.jl:15-34 (== test/cartpole_model.jl:11-30)
//   DoubleIntegrator    test/double_integrator.jl:97-107
//   RigidBody           src/rigidbody.jl:213-236 ; quaternion is NEVER renormalised (src/rigidbody.jl:101-105)
//   Satellite wrench    examples/single_satellite.jl:18-27 ; Quadrotor wrench test/quadrotor.jl:56-96
//   SatellitePair       examples/satellite_pair.jl:14-38 ; LieSatellite test/liemodel.jl:14-22
//   Euler/RK3/RK4       src/integration.jl:73-83, :130-135, :280-286 ; RK2 = v0.3 explicit midpoint
// Rotations.jl arithmetic is restated in closed (formula) form, SURVEY.md App. A.
// Deviations from the reference's literal operation order (all << the 1e-10 parity tolerance, stated in DESIGN.md):
//   F ./ m  and  J \ v  are evaluated as multiplications by host-precomputed reciprocals / inverses;
//   hipcc contracts a*b+c into FMA.
#pragma once
#include "scalar.hpp"

namespace rd {

enum { RULE_CONTINUOUS = -1, RULE_EULER = 0, RULE_RK2 = 1, RULE_RK3 = 2, RULE_RK4 = 3, RULE_IMPLICIT_MIDPOINT = 4 };    // (4: the cascade rollout's implicit step, implicit.hpp)

// ---- small vector helpers -------------------------------------------------------------------------------
template <class S> RD_DEV void cross3(const S* a, const S* b, S* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
// y = A x, A 3x3 column-major plain-T matrix
template <class T, class S> RD_DEV void mat3vec(const T* A, const S* x, S* y) {
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = x[0] * A[i] + x[1] * A[i + 3] + x[2] * A[i + 6];
}
// q * r for a possibly non-unit quaternion q = [w, v]: (w^2 - v.v) r + 2 v (v.r) + 2 w (v x r)
template <class S> RD_DEV void quat_rotate(const S* q, const S* r, S* o) {
    using T = typename scalar_of<S>::type;
    const S vv = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const S vr = q[1] * r[0] + q[2] * r[1] + q[3] * r[2];
    const S a = q[0] * q[0] - vv;
    S c[3]; cross3(q + 1, r, c);
    const S vr2 = vr * T(2), w2 = q[0] * T(2);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a * r[i] + q[1 + i] * vr2 + w2 * c[i];
}
// q \ r = conj(q) * r
template <class S> RD_DEV void quat_rotate_inv(const S* q, const S* r, S* o) {
    const S qc[4] = {q[0], -q[1], -q[2], -q[3]};
    quat_rotate(qc, r, o);
}
// Rotations.kinematics(q, w) = 1/2 q (x) [0; w]
template <class S> RD_DEV void quat_kinematics(const S* q, const S* w, S* qd) {
    using T = typename scalar_of<S>::type;
    const T half = T(0.5);
    qd[0] = (-(q[1] * w[0]) - q[2] * w[1] - q[3] * w[2]) * half;
    qd[1] = (q[0] * w[0] + q[2] * w[2] - q[3] * w[1]) * half;
    qd[2] = (q[0] * w[1] + q[3] * w[0] - q[1] * w[2]) * half;
    qd[3] = (q[0] * w[2] + q[1] * w[1] - q[2] * w[0]) * half;
}

inline void host_inv3(const double* A, double* Ai) {
    const bool diag = A[1] == 0 && A[2] == 0 && A[3] == 0 && A[5] == 0 && A[6] == 0 && A[7] == 0;
    if (diag) {   // inv(Diagonal) = Diagonal(1 ./ d)
        for (int i = 0; i < 9; ++i) Ai[i] = 0;
        Ai[0] = 1.0 / A[0]; Ai[4] = 1.0 / A[4]; Ai[8] = 1.0 / A[8];
        return;
    }
    const double a = A[0], b = A[3], c = A[6], d = A[1], e = A[4], f = A[7], g = A[2], h = A[5], i = A[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    Ai[0] = (e * i - f * h) / det; Ai[3] = (c * h - b * i) / det; Ai[6] = (b * f - c * e) / det;
    Ai[1] = (f * g - d * i) / det; Ai[4] = (a * i - c * g) / det; Ai[7] = (c * d - a * f) / det;
    Ai[2] = (d * h - e * g) / det; Ai[5] = (b * g - a * h) / det; Ai[8] = (a * e - b * d) / det;
}

// ---- models ---------------------------------------------------------------------------------------------
template <class T_> struct Cartpole {
    using T = T_;
    static constexpr int NX = 4, NU = 1;
    static constexpr bool HAS_USER_JACOBIAN = true;     // test/cartpole_model.jl:32-55
    T mc, mp, l, g;
    static Cartpole make(const double* p) { return Cartpole{T(p[0]), T(p[1]), T(p[2]), T(p[3])}; }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
        const S s = sin_(x[1]), c = cos_(x[1]);
        const T H11 = mc + mp, H22 = mp * l * l;
        const S H12 = c * (mp * l);
        const S r0 = (x[3] * x[3]) * s * (-(mp * l)) - u[0];   // C*qd + G - B*u, row 1
        const S r1 = s * (mp * g * l);                           // row 2
        const S idet = recip_(H12 * H12 * T(-1) + H11 * H22);    // one fp64 division per evaluation instead of four (value and tangent of two quotients)
        const S y0 = (r0 * H22 - H12 * r1) * idet;
        const S y1 = (r1 * H11 - H12 * r0) * idet;
        xd[0] = x[2]; xd[1] = x[3]; xd[2] = -y0; xd[3] = -y1;
    }
};

template <class T_, int D> struct DoubleIntegrator {
    using T = T_;
    static constexpr int NX = 2 * D, NU = D;
    static constexpr bool HAS_USER_JACOBIAN = true;     // test/double_integrator.jl:119-127
    static DoubleIntegrator make(const double*) { return DoubleIntegrator{}; }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
#pragma unroll
        for (int i = 0; i < D; ++i) { xd[i] = x[D + i]; xd[D + i] = u[i]; }
    }
};

template <class T_, int N_, int M_> struct Linear {
    using T = T_;
    static constexpr int NX = N_, NU = M_;
    static constexpr bool HAS_USER_JACOBIAN = true;
    T A[N_ * N_], B[N_ * M_];
    static Linear make(const double* p) {
        Linear L;
        for (int i = 0; i < N_ * N_; ++i) L.A[i] = T(p[2 + i]);
        for (int i = 0; i < N_ * M_; ++i) L.B[i] = T(p[2 + N_ * N_ + i]);
        return L;
    }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
#pragma unroll
        for (int i = 0; i < N_; ++i) {
            S acc = x[0] * A[i];
#pragma unroll
            for (int j = 1; j < N_; ++j) acc = acc + x[j] * A[i + j * N_];
#pragma unroll
            for (int j = 0; j < M_; ++j) acc = acc + u[j] * B[i + j * N_];
            xd[i] = acc;
        }
    }
};

enum { WRENCH_SATELLITE = 0, WRENCH_QUADROTOR = 1 };

// RigidBody{QuatRotation}: x = [r(3) q(4) v(3) w(3)], forces in the world frame, moments in the body frame
// DIAGJ: the inertia is diagonal (the reference's examples hold it as a `Diagonal`, examples/single_satellite.jl:9,34;
// test/quadrotor.jl:40): J w and J \\ v are 3 multiplications each instead of 9 multiply-adds.  Chosen at dispatch time.
// GZ (body-frame quadrotor): gravity is along the world z axis (test/quadrotor.jl:41: gravity = [0, 0, -9.81]) -- q \ g is then g_z times the third
// row of the rotation matrix, three short expressions instead of a full rotation in every stage.  Chosen at dispatch time like DIAGJ.
template <class T_, int WRENCH, bool BODY, bool DIAGJ = false, bool GZ = false> struct RigidBody {
    using T = T_;
    static constexpr int NX = 13, NU = (WRENCH == WRENCH_SATELLITE) ? 6 : 4;
    static constexpr int NROT = 1;                       // LieState{QuatRotation,(3,6)}: one quaternion at state index 3 (src/rigidbody.jl:36-40)
    static constexpr int rot_start(int) { return 3; }
    static constexpr bool HAS_USER_JACOBIAN = true;     // src/rigidbody.jl:260-313 (analytic blocks)
    T mass, inv_mass;
    T J[9], Jinv[9];
    T grav[3];          // mass * gravity (quadrotor)
    T gacc[3];          // gravity (quadrotor; body-frame velocities)
    T L, kf, km;        // quadrotor
    static RigidBody make(const double* p) {
        RigidBody M; M.mass = T(p[0]); M.inv_mass = T(1.0 / p[0]);
        double Ji[9]; host_inv3(p + 1, Ji);
        for (int i = 0; i < 9; ++i) { M.J[i] = T(p[1 + i]); M.Jinv[i] = T(Ji[i]); }
        if (WRENCH == WRENCH_QUADROTOR) {
            for (int i = 0; i < 3; ++i) { M.grav[i] = T(p[0] * p[10 + i]); M.gacc[i] = T(p[10 + i]); }
            M.L = T(p[13]); M.kf = T(p[14]); M.km = T(p[15]);
        } else { M.grav[0] = M.grav[1] = M.grav[2] = T(0); M.gacc[0] = M.gacc[1] = M.gacc[2] = T(0); M.L = M.kf = M.km = T(0); }
        return M;
    }
    template <class S> RD_DEV void wrench(const S* q, const S* u, S* F, S* tau) const {
        if constexpr (WRENCH == WRENCH_SATELLITE) {
            quat_rotate(q, u, F);
            tau[0] = u[3]; tau[1] = u[4]; tau[2] = u[5];
        } else {
            const S F1 = max0_(u[0] * kf), F2 = max0_(u[1] * kf), F3 = max0_(u[2] * kf), F4 = max0_(u[3] * kf);
            const S fb[3] = {S(T(0)), S(T(0)), F1 + F2 + F3 + F4};
            S qF[3]; quat_rotate(q, fb, qF);
#pragma unroll
            for (int i = 0; i < 3; ++i) F[i] = qF[i] + grav[i];
            tau[0] = (F2 - F4) * L; tau[1] = (F3 - F1) * L;
            tau[2] = (u[0] - u[1] + u[2] - u[3]) * km;
        }
    }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
        const S* q = x + 3; const S* v = x + 7; const S* w = x + 10;
        S F[3], tau[3];
        quat_kinematics(q, w, xd + 3);
        if constexpr (!BODY) {
            wrench(q, u, F, tau);
#pragma unroll
            for (int i = 0; i < 3; ++i) { xd[i] = v[i]; xd[7 + i] = F[i] * inv_mass; }
        } else {
            // vdot = q \ (F / m) - w x v (src/rigidbody.jl:229-231) with F = q * f_b (+ m g): the force of both compiled-in wrenches is a BODY-frame
            // vector rotated to the world frame.  Rotations.jl's q * r on the stored (never renormalised) quaternion is |q|^2 R(q/|q|) r and q \ r its
            // transpose, so  q \ (q * f_b) = |q|^4 f_b  exactly -- evaluated that way: one squared norm instead of two rotations of a dual vector in
            // every stage (the literal form made a body-frame RK4 tangent 60 % more arithmetic than the world-frame one, 5 909 against 3 700 fp64
            // instructions per 64-knot tile, and the kernel compute-bound at 0.39 of the roofline).  Differs from the literal order by rounding only
            // (the oracle keeps the literal form; parity to 1e-10 as everywhere).
            quat_rotate(q, v, xd);
            const S s = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            const S s2m = s * s * inv_mass;
            S a[3];
            if constexpr (WRENCH == WRENCH_SATELLITE) {
                a[0] = u[0] * s2m; a[1] = u[1] * s2m; a[2] = u[2] * s2m;
                tau[0] = u[3]; tau[1] = u[4]; tau[2] = u[5];
            } else {
                const S F1 = max0_(u[0] * kf), F2 = max0_(u[1] * kf), F3 = max0_(u[2] * kf), F4 = max0_(u[3] * kf);
                if constexpr (GZ) {                              // q \ [0, 0, g_z] = g_z [2 (q1 q3 - q0 q2), 2 (q2 q3 + q0 q1), q0^2 - q1^2 - q2^2 + q3^2]
                    const T gz = gacc[2], gz2 = gz * T(2);
                    a[0] = (q[1] * q[3] - q[0] * q[2]) * gz2;
                    a[1] = (q[2] * q[3] + q[0] * q[1]) * gz2;
                    a[2] = (q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]) * gz;
                } else {
                    const S g[3] = {S(gacc[0]), S(gacc[1]), S(gacc[2])};
                    quat_rotate_inv(q, g, a);                    // q \ g
                }
                a[2] = a[2] + (F1 + F2 + F3 + F4) * s2m;
                tau[0] = (F2 - F4) * L; tau[1] = (F3 - F1) * L;
                tau[2] = (u[0] - u[1] + u[2] - u[3]) * km;
            }
            S c[3]; cross3(w, v, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) xd[7 + i] = a[i] - c[i];
        }
        S Jw[3];
        if constexpr (DIAGJ) { Jw[0] = w[0] * J[0]; Jw[1] = w[1] * J[4]; Jw[2] = w[2] * J[8]; }
        else mat3vec(J, w, Jw);
        S c[3]; cross3(w, Jw, c);
        S tt[3] = {tau[0] - c[0], tau[1] - c[1], tau[2] - c[2]};
        if constexpr (DIAGJ) { xd[10] = tt[0] * Jinv[0]; xd[11] = tt[1] * Jinv[4]; xd[12] = tt[2] * Jinv[8]; }
        else mat3vec(Jinv, tt, xd + 10);
    }
};
inline bool host_is_diag3(const double* A) { return A[1] == 0 && A[2] == 0 && A[3] == 0 && A[5] == 0 && A[6] == 0 && A[7] == 0; }

// ---- cascade structure of a model's dynamics (used by rollout_cascade_kernel, kernels.hpp) --------------------------------
// A partition of the state into NB ordered blocks such that the time derivative of block b depends only on blocks 0..b (and on u, t).
// A rigid body has one: the angular velocity drives itself (Euler's equation), the attitude follows from it (kinematics), the linear
// motion follows from both (src/rigidbody.jl:213-236: xdot = [v or q*v ; kinematics(q, w) ; F/m (- w x v) ; J \ (tau - w x J w)], with
// the wrench a function of q and u in every compiled-in rigid body).  NB = 0: no such structure declared (the plain rollout kernel).
// control_block(j): the first block whose derivative reads control j (a satellite's thrust only enters the linear acceleration, its torque
// Euler's equation; a rotor of the quadrotor produces both).
template <class Mdl> struct Cascade { static constexpr int NB = 0; static constexpr int block_of(int) { return 0; } static constexpr int control_block(int) { return 0; } };
template <class T, int WRENCH, bool BODY, bool DIAGJ, bool GZ> struct Cascade<RigidBody<T, WRENCH, BODY, DIAGJ, GZ>> {
    static constexpr int NB = 3;
    static constexpr int block_of(int i) { return i >= 10 ? 0 : (i >= 3 && i < 7) ? 1 : 2; }       // w | q | r, v
    static constexpr int control_block(int j) { return (WRENCH == WRENCH_SATELLITE && j < 3) ? 2 : 0; }
};

// RigidBody{MRP} / RigidBody{RodriguesParam}: x = [r(3) p(3) v(3) w(3)] (gen_inds, src/rigidbody.jl:80-94: q = 4:6, v = 7:9, w = 10:12), n = 12.
// src/rigidbody.jl:213-236 with the three-parameter attitude p:
//   rotation of a vector: Rotations.jl multiplies through the unit quaternion of p (MRP: q = [(1 - p.p), 2p] / (1 + p.p);
//                         RodriguesParam: q = [1, g] / sqrt(1 + g.g)), built WITHOUT renormalisation
//   kinematics:           MRP  pdot = 1/4 ((1 - p.p) w + 2 p x w + 2 p (p.w))      RodriguesParam  gdot = 1/2 (w + g x w + g (g.w))
// (the same matrices are the error-state blocks of errstate.hip).  The rotation kind and the velocity frame are run-time
// members -- uniform branches; these models are not on a BASELINE configuration and get the compact run-time-column kernels.
template <class T_, int WRENCH> struct RigidBody3 {
    using T = T_;
    static constexpr int NX = 12, NU = (WRENCH == WRENCH_SATELLITE) ? 6 : 4;
    static constexpr int NROT = 0;                       // (no quaternion in the state: the fused error-state expansion is not offered)
    static constexpr int rot_start(int) { return 3; }
    static constexpr bool HAS_USER_JACOBIAN = false;    // the reference's analytic Jacobian is for RigidBody{<:QuatRotation} only (src/rigidbody.jl:260)
    int rot;            // 1 = MRP, 2 = RodriguesParam (rdhip_rotation)
    int body;           // velocity frame (src/rigidbody.jl:258)
    T mass, inv_mass;
    T J[9], Jinv[9];
    T grav[3], L, kf, km;
    static RigidBody3 make(const double* p, int rot_, int body_) {
        RigidBody3 M; M.rot = rot_; M.body = body_; M.mass = T(p[0]); M.inv_mass = T(1.0 / p[0]);
        double Ji[9]; host_inv3(p + 1, Ji);
        for (int i = 0; i < 9; ++i) { M.J[i] = T(p[1 + i]); M.Jinv[i] = T(Ji[i]); }
        if (WRENCH == WRENCH_QUADROTOR) {
            for (int i = 0; i < 3; ++i) M.grav[i] = T(p[0] * p[10 + i]);
            M.L = T(p[13]); M.kf = T(p[14]); M.km = T(p[15]);
        } else { M.grav[0] = M.grav[1] = M.grav[2] = T(0); M.L = M.kf = M.km = T(0); }
        return M;
    }
    template <class S> RD_DEV void to_quat(const S* p, S* q) const {
        const S n = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        if (rot == 1) {
            const S d = recip_(n + T(1)), M = d * T(2);
            q[0] = (T(1) - n) * d; q[1] = p[0] * M; q[2] = p[1] * M; q[3] = p[2] * M;
        } else {
            const S M = recip_(sqrt_(n + T(1)));
            q[0] = M; q[1] = p[0] * M; q[2] = p[1] * M; q[3] = p[2] * M;
        }
    }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
        const S* p = x + 3; const S* v = x + 6; const S* w = x + 9;
        S q[4]; to_quat(p, q);
        S F[3], tau[3];
        if constexpr (WRENCH == WRENCH_SATELLITE) {
            quat_rotate(q, u, F);
            tau[0] = u[3]; tau[1] = u[4]; tau[2] = u[5];
        } else {
            const S F1 = max0_(u[0] * kf), F2 = max0_(u[1] * kf), F3 = max0_(u[2] * kf), F4 = max0_(u[3] * kf);
            const S fb[3] = {S(T(0)), S(T(0)), F1 + F2 + F3 + F4};
            S qF[3]; quat_rotate(q, fb, qF);
#pragma unroll
            for (int i = 0; i < 3; ++i) F[i] = qF[i] + grav[i];
            tau[0] = (F2 - F4) * L; tau[1] = (F3 - F1) * L;
            tau[2] = (u[0] - u[1] + u[2] - u[3]) * km;
        }
        // kinematics
        {
            const S pp = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], pw = p[0] * w[0] + p[1] * w[1] + p[2] * w[2];
            S c[3]; cross3(p, w, c);
            if (rot == 1) {
                const S a = T(1) - pp;
#pragma unroll
                for (int i = 0; i < 3; ++i) xd[3 + i] = (a * w[i] + (c[i] + p[i] * pw) * T(2)) * T(0.25);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) xd[3 + i] = (w[i] + c[i] + p[i] * pw) * T(0.5);
            }
        }
        if (!body) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { xd[i] = v[i]; xd[6 + i] = F[i] * inv_mass; }
        } else {
            quat_rotate(q, v, xd);
            S Fm[3] = {F[0] * inv_mass, F[1] * inv_mass, F[2] * inv_mass};
            S a[3]; quat_rotate_inv(q, Fm, a);
            S c[3]; cross3(w, v, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) xd[6 + i] = a[i] - c[i];
        }
        S Jw[3]; mat3vec(J, w, Jw);
        S c[3]; cross3(w, Jw, c);
        S tt[3] = {tau[0] - c[0], tau[1] - c[1], tau[2] - c[2]};
        mat3vec(Jinv, tt, xd + 9);
    }
};

// SatellitePair: x = [q1(4) w1(3) q2(4) w2(3)], u = [u1(3) u2(3)]
template <class T_> struct SatellitePair {
    using T = T_;
    static constexpr int NX = 14, NU = 6;
    static constexpr bool HAS_USER_JACOBIAN = false;    // examples/satellite_pair.jl defines none -> NotImplementedError (src/functionbase.jl:257)
    static constexpr int NROT = 2;                       // LieState{QuatRotation,(0,3,3)}: quaternions at state indices 0 and 7
    static constexpr int rot_start(int r) { return r == 0 ? 0 : 7; }
    T J[2][9], Jinv[2][9];
    static SatellitePair make(const double* p) {
        SatellitePair M;
        for (int b = 0; b < 2; ++b) {
            double Ji[9]; host_inv3(p + 9 * b, Ji);
            for (int i = 0; i < 9; ++i) { M.J[b][i] = T(p[9 * b + i]); M.Jinv[b][i] = T(Ji[i]); }
        }
        return M;
    }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const S* q = x + 7 * b; const S* w = x + 7 * b + 4; const S* ub = u + 3 * b;
            S Jw[3]; mat3vec(J[b], w, Jw);
            S c[3]; cross3(w, Jw, c);
            S tt[3] = {ub[0] - c[0], ub[1] - c[1], ub[2] - c[2]};
            mat3vec(Jinv[b], tt, xd + 7 * b + 4);
            quat_kinematics(q, w, xd + 7 * b);
        }
    }
};

// LieSatellite (test/liemodel.jl): x = [w(3) q(4)], q normalised inside the dynamics, J diagonal
template <class T_> struct LieSatellite {
    using T = T_;
    static constexpr int NX = 7, NU = 3;
    static constexpr bool HAS_USER_JACOBIAN = false;    // test/liemodel.jl defines none
    static constexpr int NROT = 1;                       // LieState{QuatRotation,(3,0)}: the quaternion at state index 3
    static constexpr int rot_start(int) { return 3; }
    T Jd[3], Jdinv[3];
    static LieSatellite make(const double* p) {
        LieSatellite M;
        for (int i = 0; i < 3; ++i) { M.Jd[i] = T(p[i]); M.Jdinv[i] = T(1.0 / p[i]); }
        return M;
    }
    template <class S> RD_DEV void f(const S* x, const S* u, S* xd) const {
        const S* w = x;
        // q = normalize(x[4:7]) (test/liemodel.jl:15): StaticArrays' normalize is inv(norm(a)) * a -- one reciprocal and four products, not four divisions
        const S nn = sqrt_(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
        const S inv = recip_(nn);
        const S q[4] = {x[3] * inv, x[4] * inv, x[5] * inv, x[6] * inv};
        const S Jw[3] = {w[0] * Jd[0], w[1] * Jd[1], w[2] * Jd[2]};
        S c[3]; cross3(w, Jw, c);
#pragma unroll
        for (int i = 0; i < 3; ++i) xd[i] = (u[i] - c[i]) * Jdinv[i];
        quat_kinematics(q, w, xd + 3);
    }
};

// ---- dynamics(model, x, u, t) ---------------------------------------------------------------------------------
// A model defines either the time-invariant form f(x, u, xdot) -- every model of the reference's examples/ and test/ does; the
// time argument is then dropped exactly as `dynamics(model, x, u, t) = dynamics(model, x, u)` drops it (src/dynamics.jl:83) -- or the
// time-dependent form f(x, u, t, xdot) with t a plain T (time is not differentiated: ForwardDiff.jacobian differentiates w.r.t. z only).
template <class M, class S, class T> RD_DEV auto model_f_(const M& m, const S* x, const S* u, T t, S* xd, int) -> decltype(m.f(x, u, t, xd), void()) { m.f(x, u, t, xd); }
template <class M, class S, class T> RD_DEV void model_f_(const M& m, const S* x, const S* u, T, S* xd, long) { m.f(x, u, xd); }
template <class M, class S, class T> RD_DEV void model_f(const M& m, const S* x, const S* u, T t, S* xd) { model_f_(m, x, u, t, xd, 0); }

// ---- DiscretizedDynamics{RULE}(model): `evaluate(fun, z)` -------------------------------------------------
// RULE_CONTINUOUS evaluates the continuous model.  The explicit rules follow the StaticReturn `integrate`
// methods (k_i = f(.) * h ; x + k1/2 ...), src/integration.jl:130-135 and :280-286, stage times t, t + h/2, t + h included.
template <class Mdl, int RULE> struct Fun {
    using T = typename Mdl::T;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = Mdl::NX, RULE_ID = RULE;
    // output tile of the Jacobian kernels: NY rows x NCOLS columns per knot point, NVAL function values; NEXT = 1 if the
    // kernel also needs the state of the following knot point
    static constexpr int NCOLS = Mdl::NX + Mdl::NU, NVAL = Mdl::NX, NEXT = 0;
    Mdl m;
    template <class S> RD_DEV void eval(const S* x, const S* u, T t, T h, S* y) const {
        if constexpr (RULE == RULE_CONTINUOUS) {
            model_f(m, x, u, t, y);
        } else if constexpr (RULE == RULE_EULER) {
            S k1[NX]; model_f(m, x, u, t, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) y[i] = x[i] + k1[i] * h;
        } else if constexpr (RULE == RULE_RK2) {
            S k[NX], xs[NX]; model_f(m, x, u, t, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = x[i] + k[i] * (h * T(0.5));
            model_f(m, xs, u, t + h * T(0.5), k);
#pragma unroll
            for (int i = 0; i < NX; ++i) y[i] = x[i] + k[i] * h;
        } else if constexpr (RULE == RULE_RK3) {
            // x + (k1 + 4 k2 + k3)/6 accumulated left to right as the stages finish (same summation order as
            // the reference expression, src/integration.jl:134); keeps only {k1, acc, k, xs} live.
            S k1[NX], k[NX], acc[NX], xs[NX];
            model_f(m, x, u, t, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k1[i] = k1[i] * h; xs[i] = x[i] + k1[i] * T(0.5); }
            model_f(m, xs, u, t + h * T(0.5), k);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k[i] = k[i] * h; acc[i] = k1[i] + k[i] * T(4); xs[i] = x[i] - k1[i] + k[i] * T(2); }
            model_f(m, xs, u, t + h, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) y[i] = x[i] + (acc[i] + k[i] * h) * T(1.0 / 6.0);
        } else {
            // x + (k1 + 2 k2 + 2 k3 + k4)/6 with k_i = h f_i (src/integration.jl:280-286), evaluated as
            // x + (h/6)(f1 + 2 f2 + 2 f3 + f4) with stage points x + (h/2) f_i: the same polynomial, 4*NX fewer multiplications
            // per tangent direction (differs from the literal order by rounding only); live set {acc, k, xs}
            S k[NX], acc[NX], xs[NX];
            const T h2 = h * T(0.5), h6 = h * T(1.0 / 6.0);
            model_f(m, x, u, t, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) { acc[i] = k[i]; xs[i] = x[i] + k[i] * h2; }
            model_f(m, xs, u, t + h2, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + k[i] * T(2); xs[i] = x[i] + k[i] * h2; }
            model_f(m, xs, u, t + h2, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + k[i] * T(2); xs[i] = x[i] + k[i] * h; }
            model_f(m, xs, u, t + h, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) y[i] = x[i] + (acc[i] + k[i]) * h6;
        }
    }
};

// The same explicit rules with a hook on every stage point: `hk.template at<S>(xs)` sees (and may overwrite) the point at which stage S
// evaluates the model.  rollout_cascade_kernel uses it to splice in the components other wavefronts own; the formulas are those of
// Fun::eval above, operation for operation.
template <int RULE> constexpr int rule_stages() { return (RULE == RULE_EULER || RULE == RULE_IMPLICIT_MIDPOINT) ? 1 : RULE == RULE_RK2 ? 2 : RULE == RULE_RK3 ? 3 : 4; }   // (implicit midpoint: the midpoint is its one stage point)
template <int RULE, class Mdl, class T, class Hook>
RD_DEV void rk_hooked(const Mdl& m, const T (&x)[Mdl::NX], const T* u, T t, T h, T (&y)[Mdl::NX], Hook& hk) {
    constexpr int NX = Mdl::NX;
    static_assert(RULE == RULE_EULER || RULE == RULE_RK2 || RULE == RULE_RK3 || RULE == RULE_RK4, "explicit rules only");
    T x0[NX], xs[NX], k[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x0[i] = x[i];
    hk.template at<0>(x0);
    if constexpr (RULE == RULE_EULER) {
        model_f(m, x0, u, t, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = x0[i] + k[i] * h;
    } else if constexpr (RULE == RULE_RK2) {
        model_f(m, x0, u, t, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[i] = x0[i] + k[i] * (h * T(0.5));
        hk.template at<1>(xs);
        model_f(m, xs, u, t + h * T(0.5), k);
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = x0[i] + k[i] * h;
    } else if constexpr (RULE == RULE_RK3) {
        T k1[NX], acc[NX];
        model_f(m, x0, u, t, k1);
#pragma unroll
        for (int i = 0; i < NX; ++i) { k1[i] = k1[i] * h; xs[i] = x0[i] + k1[i] * T(0.5); }
        hk.template at<1>(xs);
        model_f(m, xs, u, t + h * T(0.5), k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { k[i] = k[i] * h; acc[i] = k1[i] + k[i] * T(4); xs[i] = x0[i] - k1[i] + k[i] * T(2); }
        hk.template at<2>(xs);
        model_f(m, xs, u, t + h, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = x0[i] + (acc[i] + k[i] * h) * T(1.0 / 6.0);
    } else {
        T acc[NX];
        const T h2 = h * T(0.5), h6 = h * T(1.0 / 6.0);
        model_f(m, x0, u, t, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = k[i]; xs[i] = x0[i] + k[i] * h2; }
        hk.template at<1>(xs);
        model_f(m, xs, u, t + h2, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + k[i] * T(2); xs[i] = x0[i] + k[i] * h2; }
        hk.template at<2>(xs);
        model_f(m, xs, u, t + h2, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + k[i] * T(2); xs[i] = x0[i] + k[i] * h; }
        hk.template at<3>(xs);
        model_f(m, xs, u, t + h, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = x0[i] + (acc[i] + k[i]) * h6;
    }
}

// ---- error-state layout of a LieState{QuatRotation,P} model (src/liestate.jl:75-131) at compile time --------------------
// error index e <-> either a vector state index, or (rotation r, component j of its 3-vector)
struct ErrSlot { int rot; int j; int s; };      // rot < 0: vector part, state index s ; rot >= 0: quaternion starting at state index s
template <class Mdl> constexpr int err_dim() { return Mdl::NX - Mdl::NROT; }
template <class Mdl> constexpr ErrSlot err_slot(int e) {
    int s = 0, c = 0, r = 0;
    while (s < Mdl::NX) {
        if (r < Mdl::NROT && s == Mdl::rot_start(r)) {
            if (e >= c && e < c + 3) return ErrSlot{r, e - c, s};
            s += 4; c += 3; ++r;
        } else {
            if (e == c) return ErrSlot{-1, 0, s};
            ++s; ++c;
        }
    }
    return ErrSlot{-1, 0, 0};
}
// DiscretizedDynamics{RULE}(model) seen through the error state: rows and state columns are error-state coordinates.
// error_expansion! of TrajectoryOptimization.jl (the consumer named in SURVEY 8f-2):  [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k]
template <class Mdl, int RULE> struct ErrFun : Fun<Mdl, RULE> {
    static constexpr int NE = err_dim<Mdl>();
    static constexpr int NY = NE, NCOLS = NE + Mdl::NU, NVAL = Mdl::NX, NEXT = 1;
};

}  // namespace rd
